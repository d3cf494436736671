#!/bin/bash
# Round 6, session r: the final tree (after the flood-table change): the driver's GPU command, smoke, the driver's bench command.
cd /root/repo; OUT=/root/repo/gpurun_out/r06r; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2> $OUT/pytest_gpu.err; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -8 $OUT/pytest_gpu.log | grep -E "passed|failed|rc="
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-100
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r06r/bench_driver_command.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["steps"], d["warmup"], round(d["ms_per_step"],2), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
print("cold", round(d["cold_first_step_ms"],1), "new terrain", d["new_terrain"]["ms"], {k:r[k] for k in ('kernel','frac','avg_launch_us')}, "whole stack", round(r["whole_stack"]["frac"],4))
PY
