#!/bin/bash
# Round 4, after the chain form of the flood's carve pass: flood timing lines, the default bench line, the flood-related GPU tests
cd /root/repo; OUT=/root/repo/gpurun_out/r04x; mkdir -p $OUT
export TMPDIR=/tmp
WO_FLOOD_TIMING=1 python bench.py --no-cpu --no-profile --no-relaxed --in-flight 0 --steps 2 --warmup 1 > $OUT/bench_flood_timing.json 2> $OUT/flood_timing.txt
WO_FLOOD_TIMING=1 WO_FLOOD_CHAINS_MIN=0 python bench.py --no-cpu --no-profile --no-relaxed --in-flight 0 --steps 2 --warmup 1 > $OUT/bench_flood_timing_plain.json 2> $OUT/flood_timing_plain.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python -m pytest tests -m gpu -x -q > $OUT/gputests.txt 2>&1
tail -3 $OUT/gputests.txt
grep -E "round joined|walk of" $OUT/flood_timing.txt | tail -8
grep -E "round joined|walk of" $OUT/flood_timing_plain.txt | tail -8
python - <<'PY'
import json
for f in ("bench_flood_timing","bench_flood_timing_plain","bench_default"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r04x/{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
    except Exception as ex: print(f, "ERR", ex)
PY
