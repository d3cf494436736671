#!/bin/bash
# Round 5, session p: the tree after the splitter-sort experiment was taken out again: full -m gpu suite, bench
cd /root/repo; OUT=/root/repo/gpurun_out/r05p; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
timeout 2700 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -16 $OUT/pytest_gpu.log
python bench.py --timed-only --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
make -s -C planet_heightmap_generation_amd/csrc clean > /dev/null 2>&1; make -s -j16 -C planet_heightmap_generation_amd/csrc EXTRA="-DWO_EVENTS_N=4" > $OUT/make_ev4.log 2>&1
python bench.py --timed-only --steps 5 --warmup 2 > $OUT/bench_events4.json 2> $OUT/bench_events4.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r05p/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"], d["erode_stats"].get("calls_run_again_with_checks"))
    except Exception as ex: print(f, "ERR", ex)
PY
