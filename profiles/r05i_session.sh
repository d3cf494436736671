#!/bin/bash
# Round 5, session i: splitter sort (one partition pass + one LDS sort per bucket) against the four-pass radix sort
cd /root/repo; OUT=/root/repo/gpurun_out/r05i; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sort_routes or golden or against_oracle_large or config3 or flow_accumulation or mirror_layout or ties_on_larger or glacial_step or land_count or edge_cases or graph_replay" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -15 $OUT/pytest_subset.log
python bench.py --timed-only --steps 5 --warmup 2 > $OUT/bench_split.json 2> $OUT/bench_split.err
WO_SORT=radix python bench.py --timed-only --steps 5 --warmup 2 > $OUT/bench_radix.json 2> $OUT/bench_radix.err
cd /tmp; rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 1 --warmup 1 > /dev/null 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for n in 100 250; do python /root/repo/profiles/iteration_timeline.py /tmp/kt $n >> $OUT/iteration_timeline.txt 2>&1; done
cd /root/repo
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r05i/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"], d["erode_stats"].get("sorts_by_splitters"), d["erode_stats"].get("calls_run_again_with_checks"))
    except Exception as ex: print(f, "ERR", ex)
PY
