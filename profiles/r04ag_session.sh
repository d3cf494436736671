#!/bin/bash
# Round 4, final tree of the round (+ every long range of the basin solve listed): default bench line, rocprofv3 kernel
# stats of the same command, PMC FETCH/WRITE per kernel (200 iterations, one planet), flood timing
cd /root/repo; OUT=/root/repo/gpurun_out/r04ag; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
WO_FLOOD_TIMING=1 python bench.py --no-cpu --no-profile --no-relaxed --in-flight 0 --steps 2 --warmup 1 > $OUT/bench_flood_timing.json 2> $OUT/flood_timing_10m.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_default_bench_command.csv
cd /root/repo
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o pmc -- python bench.py --no-cpu --no-profile --no-relaxed --in-flight 0 --steps 1 --warmup 0 --iters 200 > "$OUT/$C.log" 2>&1
    echo "$C rc=$?"
done
python profiles/summarize_pmc.py "$OUT" 200 > "$OUT/pmc_summary.json"
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
python - <<'PY'
import json
for f in ("bench_default","bench_flood_timing","bench_under_rocprof"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r04ag/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
    except Exception as ex: print(f, "ERR", ex)
PY
