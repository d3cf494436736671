#!/bin/bash
# Round 6, session k: the driver's bench command on the final tree, its wall time, then the same under rocprofv3 --kernel-trace --stats (timed region only).
cd /root/repo; OUT=/root/repo/gpurun_out/r06k; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s.%N); python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "bench rc=$? wall $(echo "$(date +%s.%N) - $T0" | bc) s"
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r06k/bench_driver_command.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["steps"], d["warmup"], round(d["ms_per_step"],2), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
print("cold", round(d["cold_first_step_ms"],1), "new terrain", d["new_terrain"]["ms"], "with transfers", round(d["value_with_transfers"],1))
print({k:r[k] for k in ('kernel','achieved','frac','launches','avg_launch_us','traffic')})
print("whole stack", r["whole_stack"]["frac"], "cpu", d["cpu_baseline"]["value"], "relaxed", {k:(round(v.get('ms_per_step',0),1) if isinstance(v,dict) else v) for k,v in (d.get("relaxed_mode") or {}).items()})
print("in flight", d.get("ensemble_in_flight"))
print("passes", {k:(round(v["ms"],1), round(v["frac"],3)) for k,v in r["passes"].items()})
PY
cd /tmp; rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 5 --warmup 1 > $OUT/bench_under_rocprof_timed_only.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_timed_region_only.csv; head -6 $OUT/rocprofv3_kernel_stats_timed_region_only.csv | cut -c1-200
