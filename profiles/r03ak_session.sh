#!/bin/bash
# granule carve: routes test, then kernel-trace timing of both one-launch forms
cd /root/repo; mkdir -p gpurun_out/r03ak
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "glacial_step_routes or golden or against_oracle_large or edge_cases or pipeline_matches" > gpurun_out/r03ak/tests.log 2>&1; tail -5 gpurun_out/r03ak/tests.log
cd /tmp
for M in 2 1; do
WO_CARVE_FLOW=$M timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m$M -o t -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 1 > /root/repo/gpurun_out/r03ak/bench_mode$M.log 2>&1
cp $(find /tmp/prof_m$M -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03ak/kernel_stats_mode$M.csv
echo "mode $M"; grep -E "k_carve|k_ice" /root/repo/gpurun_out/r03ak/kernel_stats_mode$M.csv | cut -c1-150
grep -o '"ms_per_step": [0-9.]*' /root/repo/gpurun_out/r03ak/bench_mode$M.log; grep -o '"parity": {[^}]*}' /root/repo/gpurun_out/r03ak/bench_mode$M.log; grep -o '"carve_flow_launches_with_leftovers": [0-9.]*' /root/repo/gpurun_out/r03ak/bench_mode$M.log
done
