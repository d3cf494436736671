#!/bin/bash
# Round 5, session a: the tree as round 4 left it (+ bench hygiene): default bench line; rocprofv3 kernel stats + kernel trace of the
# TIMED REGION ONLY (bench.py --timed-only: warm-up + timed steps, nothing else); flood timing; sort displacement is in r03z.
cd /root/repo; OUT=/root/repo/gpurun_out/r05a; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 5 --warmup 1 > $OUT/bench_under_rocprof_timed_only.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_timed_region_only.csv
for n in 250 700 1150; do python /root/repo/profiles/iteration_timeline.py /tmp/kt $n >> $OUT/iteration_timeline.txt 2>&1 || true; done
cd /root/repo
WO_FLOOD_TIMING=1 python bench.py --timed-only --steps 2 --warmup 1 > $OUT/bench_flood_timing.json 2> $OUT/flood_timing_10m.txt
python - <<'PY'
import json
for f in ("bench_default","bench_under_rocprof_timed_only","bench_flood_timing"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r05a/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
        if d.get("roofline"): print("  roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
    except Exception as ex: print(f, "ERR", ex)
PY
