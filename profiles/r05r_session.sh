#!/bin/bash
# Round 5, session r: PMC FETCH_SIZE / WRITE_SIZE per kernel on ONE timed step (bench.py --timed-only --steps 1 --warmup 0: nothing but that step), 200 iterations;
# rocprofv3 kernel stats of the timed region only; the default bench line
cd /root/repo; OUT=/root/repo/gpurun_out/r05r; mkdir -p $OUT
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o pmc -- python bench.py --timed-only --steps 1 --warmup 0 --iters 200 > "$OUT/$C.log" 2>&1
    echo "$C rc=$?"
done
python profiles/summarize_pmc.py "$OUT" 200 > "$OUT/pmc_summary.json"
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
cp $OUT/pmc_summary.json profiles/r05_pmc_fetch_write_per_kernel_10m_200iters.json
cd /tmp; rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 5 --warmup 1 > $OUT/bench_under_rocprof_timed_only.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_timed_region_only.csv
python /root/repo/profiles/iteration_timeline.py /tmp/kt 700 > $OUT/iteration_timeline.txt 2>&1
cd /root/repo
python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -3 $OUT/bench_default.err
python - <<'PY'
import json
for f in ("bench_default","bench_under_rocprof_timed_only"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r05r/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
        r=d.get("roofline")
        if r: print("  roofline", r["kernel"], r["family"], r["frac"], r["avg_launch_us"], r["traffic"], {k:v["frac"] for k,v in r["passes"].items()}, r["whole_stack"])
    except Exception as ex: print(f, "ERR", ex)
PY
