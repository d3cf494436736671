#!/bin/bash
# Round 5, session aj: the final tree of the round: build() + smoke(), full -m gpu suite, default bench line, rocprofv3 kernel stats of the timed region.
cd /root/repo; OUT=/root/repo/gpurun_out/r05aj; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 2700 python -m pytest tests -x -q -m gpu --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -10 $OUT/pytest_gpu.log
python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
cd /tmp; rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 5 --warmup 1 > $OUT/bench_under_rocprof_timed_only.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_timed_region_only.csv
cd /root/repo
python - <<'PY'
import json,csv
d=json.loads(open("/root/repo/gpurun_out/r05aj/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
print({k:r[k] for k in ('kernel','family','achieved','frac','launches','avg_launch_us','algorithmic_bytes_per_launch','traffic')})
tot=0;n=0
for row in csv.DictReader(open("/root/repo/gpurun_out/r05aj/rocprofv3_kernel_stats_timed_region_only.csv")):
    if 'k_rs_scatter' in row['Name'] or 'k_rs_count' in row['Name']: tot+=float(row['TotalDurationNs']); n+=int(row['Calls'])
print("rocprof avg per radix launch us", tot/n/1e3, "bench", r['avg_launch_us'])
print(d["cpu_baseline"])
PY
