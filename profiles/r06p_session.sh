#!/bin/bash
# Round 6, session p: stability — the driver's GPU command four more times in a row on the final tree (the round-5 abort was non-deterministic).
cd /root/repo; OUT=/root/repo/gpurun_out/r06p; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3 4; do
  timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu_$i.log 2> $OUT/pytest_gpu_$i.err; echo "run $i rc=$? $(grep -E 'passed|failed' $OUT/pytest_gpu_$i.log | tail -1)" | tee -a $OUT/summary.txt
done
