#!/bin/bash
# Round 5, session al: the HIP-event pair's own share (2 x a pair around one empty kernel - a pair around two), measured live and taken off the family times:
# does avg_launch_us now agree with the rocprofv3 kernel trace of the same build, family by family?
cd /root/repo; OUT=/root/repo/gpurun_out/r05al; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"

cd /tmp; rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 5 --warmup 1 > $OUT/bench_under_rocprof_timed_only.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_timed_region_only.csv
cd /root/repo
python - <<'PY'
import json,csv
d=json.loads(open("/root/repo/gpurun_out/r05al/bench_default.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
print({k:r[k] for k in ('kernel','achieved','frac','launches','avg_launch_us','avg_launch_us_with_the_event_pair')}); print(r['event_pair_us'])
rows={}
for row in csv.DictReader(open("/root/repo/gpurun_out/r05al/rocprofv3_kernel_stats_timed_region_only.csv")):
    rows[row['Name']]=(float(row['TotalDurationNs']),int(row['Calls']),float(row['AverageNs']))
tot=sum(v[0] for k,v in rows.items() if 'k_rs_scatter' in k or 'k_rs_count' in k); n=sum(v[1] for k,v in rows.items() if 'k_rs_scatter' in k or 'k_rs_count' in k)
print("rocprof avg per radix launch us", tot/n/1e3, "bench net", r['avg_launch_us'], "ratio", r['avg_launch_us']/(tot/n/1e3))
fam=r['families']
def rp(sub): 
    v=[x for k,x in rows.items() if sub in k]; return sum(a for a,b,c in v)/max(1,sum(b for a,b,c in v))/1e3
for f,sub in (("receivers","k_receivers_flow_init"),("thermal_excess","k_thermal_excess"),("thermal_apply","k_thermal_apply_reg"),("solve_setup","k_solve_setup_batched"),("solve_basin","k_solve_flowing"),("flow_final","k_flow_final"),("sort_keys","k_sort_keys")):
    print(f, "bench net us", round(fam[f]['ms']*1e3/fam[f]['launches'],2), "rocprof us", round(rp(sub),2))
PY
