#!/bin/bash
# Round 5, session o: k_bucket_sort with its value gathers in flight together; splitter sort against the radix sort
cd /root/repo; OUT=/root/repo/gpurun_out/r05o; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sort_routes or golden or config3" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -3 $OUT/pytest_subset.log
for i in 1 2; do
python bench.py --timed-only --steps 5 --warmup 2 > $OUT/bench_split_$i.json 2> $OUT/bench_split.err
WO_SORT=radix python bench.py --timed-only --steps 5 --warmup 2 > $OUT/bench_radix_$i.json 2> $OUT/bench_radix.err
done
cd /tmp; rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 1 --warmup 1 > /dev/null 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python /root/repo/profiles/iteration_timeline.py /tmp/kt 120 > $OUT/iteration_timeline.txt 2>&1
cd /root/repo
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob("/root/repo/gpurun_out/r05o/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"]["sort"], d["stage_ms_last_step"]["priority_flood"], d["erode_stats"].get("sorts_by_splitters"), d["erode_stats"].get("sort_buckets_through_global_memory"), d["erode_stats"].get("calls_run_again_with_checks"))
    except Exception as ex: print(f, "ERR", ex)
for r in csv.DictReader(open("/root/repo/gpurun_out/r05o/kernel_stats.csv")):
    if any(k in r['Name'] for k in ('bucket_sort','split_keys','rs_scatter','rs_count')): print(r['Name'].split('(')[0][-30:], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
