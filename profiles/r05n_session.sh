#!/bin/bash
# Round 5, session n: where k_bucket_sort's time goes (passes capped: timing only, the fields are not the sort's)
cd /root/repo; OUT=/root/repo/gpurun_out/r05n; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
cd /tmp
for P in 0 1 2 9; do
rm -rf /tmp/kt$P; WO_X_BS_PASSES=$P timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$P -o t -- python /root/repo/bench.py --timed-only --steps 1 --warmup 1 --iters 40 > /dev/null 2> $OUT/kt$P.err
echo "passes cap $P"; python3 -c "
import csv,glob
f=glob.glob('/tmp/kt$P/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'bucket_sort' in r['Name'] or 'k_rs_scatter<true>' in r['Name'] or 'split_keys' in r['Name']: print(r['Name'].split('(')[0][-30:], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
"
done
