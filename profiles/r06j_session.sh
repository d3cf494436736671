#!/bin/bash
# Round 6, session j: the tree after the last clean-ups (unused families, helpers, arrays): smoke, the driver's GPU command, the driver's bench command.
cd /root/repo; OUT=/root/repo/gpurun_out/r06j; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-160
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2> $OUT/pytest_gpu.err; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -8 $OUT/pytest_gpu.log | grep -E "passed|failed|rc="
/usr/bin/time -v python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "bench rc=$?"; grep "Elapsed (wall" $OUT/bench_driver_command.err
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r06j/bench_driver_command.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["steps"], d["warmup"], round(d["ms_per_step"],2), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
print("cold", round(d["cold_first_step_ms"],1), "new terrain", d["new_terrain"]["ms"], "with transfers", round(d["value_with_transfers"],1))
print({k:r[k] for k in ('kernel','achieved','frac','launches','avg_launch_us','traffic')})
print("whole stack", r["whole_stack"]["frac"], "cpu", d["cpu_baseline"]["value"], "relaxed", {k:(round(v.get('ms_per_step',0),1) if isinstance(v,dict) else v) for k,v in (d.get("relaxed_mode") or {}).items()})
print("in flight", d.get("ensemble_in_flight"))
PY
