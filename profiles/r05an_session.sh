#!/bin/bash
# Round 5, session an: the round's final tree: build() + smoke(), full -m gpu suite (80 tests: four more config-5 planets at the full 200 iterations), default bench line.
cd /root/repo; OUT=/root/repo/gpurun_out/r05an; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 2700 python -m pytest tests -x -q -m gpu --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -10 $OUT/pytest_gpu.log
python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r05an/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
print({k:r[k] for k in ('kernel','achieved','frac','launches','avg_launch_us','avg_launch_us_with_the_event_pair')}, r['event_pair_us']['taken_off_per_launch'])
PY
