#!/bin/bash
# batched solve setup: parity + kernel-trace timings
cd /root/repo; mkdir -p gpurun_out/r03ag
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r03ag/gpu_parity.log 2>&1; tail -3 gpurun_out/r03ag/gpu_parity.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o t -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 1 > /root/repo/gpurun_out/r03ag/bench.log 2>&1
cp $(find /tmp/prof_b -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03ag/kernel_stats.csv
grep -E "k_solve_setup|k_solve_coop" /root/repo/gpurun_out/r03ag/kernel_stats.csv | cut -c1-140
grep -o '"ms_per_step": [0-9.]*' /root/repo/gpurun_out/r03ag/bench.log; grep -o '"parity": {[^}]*}' /root/repo/gpurun_out/r03ag/bench.log
