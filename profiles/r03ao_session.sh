#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03ao
export TMPDIR=/tmp
timeout 600 python -m pytest /root/repo/tests/test_gpu_parity.py -m gpu -q -x -k "golden or against_oracle_large or ties_on_larger or edge_cases" 2>&1 | tail -2; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o t -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 1 > /root/repo/gpurun_out/r03ao/bench.log 2>&1
cp $(find /tmp/prof_r -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03ao/kernel_stats.csv
grep -E "k_rs_|k_sort_keys|rocprim|fillBuffer|k_rank" /root/repo/gpurun_out/r03ao/kernel_stats.csv | cut -c1-50,60-200 | cut -c1-170
grep -o "\"ms_per_step\": [0-9.]*" /root/repo/gpurun_out/r03ao/bench.log; grep -o "\"parity\": {[^}]*}" /root/repo/gpurun_out/r03ao/bench.log
