#!/bin/bash
# Round 5, session aw: the round's final tree (after the host-side stalls of the step were removed): build() + smoke(), full -m gpu suite, plain bench, rocprofv3 kernel stats of the timed region.
cd /root/repo; OUT=/root/repo/gpurun_out/r05aw; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 2700 python -m pytest tests -x -q -m gpu --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -10 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_default_no_flags.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cd /tmp; rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 5 --warmup 1 > $OUT/bench_under_rocprof_timed_only.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_timed_region_only.csv
cd /root/repo
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r05aw/bench_default_no_flags.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["steps"], d["warmup"], round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
print({k:r[k] for k in ('kernel','achieved','frac','launches','avg_launch_us','avg_launch_us_with_the_event_pair')}, r['event_pair_us']['taken_off_per_launch'])
print("whole stack", r["whole_stack"]["frac"], "in flight", d.get("ensemble_in_flight"), "with transfers", d.get("value_with_transfers"))
print("relaxed", {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in (d.get("relaxed_mode") or {}).items()})
PY
