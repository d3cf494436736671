#!/bin/bash
# Round 6, session e: after the prune (14 documented WO_* variables, losing routes and their kernels removed, test hooks in WO_TEST_HOOKS) and the cheaper new-terrain set-up:
# smoke, the full -m gpu suite, the new-terrain probe, the default bench.
cd /root/repo; OUT=/root/repo/gpurun_out/r06e; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 3300 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2> $OUT/pytest_gpu.err; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -14 $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.err
python profiles/new_terrain_probe.py > $OUT/new_terrain_probe.txt 2> $OUT/new_terrain_probe.err; echo "probe rc=$?"
cat $OUT/new_terrain_probe.txt; grep "mirror\]\|flood static" $OUT/new_terrain_probe.err | head -40
python bench.py > $OUT/bench_default_no_flags.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r06e/bench_default_no_flags.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["steps"], d["warmup"], round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
print("cold", d["cold_first_step_ms"], "new terrain", d["new_terrain"]["ms"])
print({k:r[k] for k in ('kernel','achieved','frac','launches','avg_launch_us')})
print("whole stack", r["whole_stack"]["frac"])
PY
