#!/bin/bash
# kernel-level view of the one-launch glacial step (rocprofv3 kernel trace, one step)
cd /root/repo; mkdir -p gpurun_out/r03ae
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_flow -o flow -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 1 > /root/repo/gpurun_out/r03ae/flow.log 2>&1
WO_CARVE_FLOW=0 WO_ICE_ROUNDS=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rounds -o rounds -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 1 > /root/repo/gpurun_out/r03ae/rounds.log 2>&1
cp $(find /tmp/prof_flow -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03ae/flow_kernel_stats.csv
cp $(find /tmp/prof_rounds -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03ae/rounds_kernel_stats.csv
tail -1 /root/repo/gpurun_out/r03ae/flow.log | cut -c1-300
