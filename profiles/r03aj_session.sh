#!/bin/bash
# 11-bit elevation sort A/B (kernel trace) + the new glacial routes test
cd /root/repo; mkdir -p gpurun_out/r03aj
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "glacial_step_routes or golden or against_oracle_large" > gpurun_out/r03aj/tests.log 2>&1; tail -3 gpurun_out/r03aj/tests.log
cd /tmp
for B in 11 8; do
WO_SORT_BITS=$B timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s$B -o t -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 1 > /root/repo/gpurun_out/r03aj/bench_bits$B.log 2>&1
cp $(find /tmp/prof_s$B -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03aj/kernel_stats_bits$B.csv
echo "bits $B"; grep -E "rocprim|fillBuffer" /root/repo/gpurun_out/r03aj/kernel_stats_bits$B.csv | awk -F'",' '{print substr($1,1,40), $2}' | cut -c1-160
grep -o '"ms_per_step": [0-9.]*' /root/repo/gpurun_out/r03aj/bench_bits$B.log; grep -o '"parity": {[^}]*}' /root/repo/gpurun_out/r03aj/bench_bits$B.log
done
