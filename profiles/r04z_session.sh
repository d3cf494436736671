#!/bin/bash
# Round 4, final tree of the session (flood: chain-form carve pass, ring queue, replay with a stop level): default bench line, rocprofv3 kernel stats of
# the same command, flood timing at 10 M and 40 M cells, GPU test suite
cd /root/repo; OUT=/root/repo/gpurun_out/r04z; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
WO_FLOOD_TIMING=1 python bench.py --no-cpu --no-profile --no-relaxed --in-flight 0 --steps 2 --warmup 1 > $OUT/bench_flood_timing.json 2> $OUT/flood_timing_10m.txt
WO_FLOOD_TIMING=1 python bench.py --cells 40000000 --iters 20 --no-cpu --no-profile --no-relaxed --in-flight 0 --steps 1 --warmup 1 > $OUT/bench_40m_20iters.json 2> $OUT/flood_timing_40m.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_default_bench_command.csv
cd /root/repo
python -m pytest tests -m gpu -x -q > $OUT/gputests.txt 2>&1
grep -E "passed|failed|rror" $OUT/gputests.txt | tail -3
python - <<'PY'
import json
for f in ("bench_default","bench_flood_timing","bench_40m_20iters","bench_under_rocprof"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r04z/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
    except Exception as ex: print(f, "ERR", ex)
PY
