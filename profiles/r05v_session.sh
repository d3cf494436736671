#!/bin/bash
# Round 5, session v: the replay's decided-prefix rule (flood_host.cc: PopLog).  Full -m gpu suite (its 40 M-cell CRC tests go through the replay in
# nearly every flood call), the default bench line, and the 40 M-cell planet on one GPU with the flood's laps, prefix rule on and off.
cd /root/repo; OUT=/root/repo/gpurun_out/r05v; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
WO_FLOOD_TIMING=1 timeout 900 python bench.py --cells 40000000 --iters 20 --timed-only --steps 1 --warmup 1 > $OUT/bench_40m_20iters.json 2> $OUT/flood_timing_40m.txt
WO_FLOOD_PREFIX=0 WO_FLOOD_TIMING=1 timeout 900 python bench.py --cells 40000000 --iters 20 --timed-only --steps 1 --warmup 1 > $OUT/bench_40m_20iters_prefix_off.json 2> $OUT/flood_timing_40m_prefix_off.txt
python - <<'PY'
import json
for f in ("bench_default","bench_40m_20iters","bench_40m_20iters_prefix_off"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r05v/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), round(d["value"],1), d.get("parity",{}).get("parity_crc_ok"), d["stage_ms_last_step"])
    except Exception as ex: print(f, "failed", ex)
PY
grep -E "replay  |replay:|pipeline|resumed  |round 2" $OUT/flood_timing_40m.txt | tail -12
grep -E "replay  |replay:|pipeline|resumed  |round 2" $OUT/flood_timing_40m_prefix_off.txt | tail -12
