#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03bh
export TMPDIR=/tmp
cd /tmp
for V in keys nokeys; do
unset WO_KEYS_FROM_APPLY; [ $V = nokeys ] && export WO_KEYS_FROM_APPLY=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$V -o t -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 1 > /root/repo/gpurun_out/r03bh/$V.log 2>&1
cp $(find /tmp/prof_$V -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03bh/${V}_kernel_stats.csv
echo $V; grep -E "k_thermal|k_sort_keys|k_rs_" /root/repo/gpurun_out/r03bh/${V}_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}'
done
