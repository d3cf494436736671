#!/bin/bash
# Reproduce round 5's SIGABRT: the driver's command, stderr kept, twice, then the A/B with the copy staging.
mkdir -p gpurun_out/r06a
export AMD_LOG_LEVEL=1
for i in 1 2; do
  timeout 1200 python -X faulthandler -m pytest tests/ -x -q -m gpu -p no:cacheprovider > gpurun_out/r06a/full_$i.out 2> gpurun_out/r06a/full_$i.err
  echo "run $i rc=$?" >> gpurun_out/r06a/summary.txt
  dmesg 2>/dev/null | tail -40 > gpurun_out/r06a/dmesg_$i.txt
done
tail -c 3000 gpurun_out/r06a/full_1.out; tail -c 3000 gpurun_out/r06a/full_1.err; cat gpurun_out/r06a/summary.txt
