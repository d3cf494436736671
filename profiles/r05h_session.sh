#!/bin/bash
# Round 5, session h: where the basin layout forks off (after the first tile kernel / after the root links); default bench line for the record
cd /root/repo; OUT=/root/repo/gpurun_out/r05h; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
for i in 1 2; do
python bench.py --timed-only --steps 5 --warmup 2 > $OUT/bench_fork_tiles_$i.json 2> $OUT/bench_a.err
WO_X_FORK_AFTER_LINKS=1 python bench.py --timed-only --steps 5 --warmup 2 > $OUT/bench_fork_links_$i.json 2> $OUT/bench_b.err
done
cd /tmp; rm -rf /tmp/kt; WO_X_FORK_AFTER_LINKS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 1 --warmup 1 > /dev/null 2> $OUT/kt.err
python /root/repo/profiles/iteration_timeline.py /tmp/kt 150 > $OUT/iteration_timeline_fork_links.txt 2>&1
cd /root/repo
python bench.py --steps 5 --warmup 2 > $OUT/bench_full_line.json 2> $OUT/bench_full_line.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r05h/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
        if d.get("ensemble_in_flight"): print("   in flight", d["ensemble_in_flight"]["value"], "cpu", d["cpu_baseline"]["value"], "transfers", d["value_with_transfers"])
    except Exception as ex: print(f, "ERR", ex)
PY
