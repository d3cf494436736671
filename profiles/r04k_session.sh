#!/bin/bash
# Round 4, session k: kernel trace (timestamps) of 30 iterations for the per-iteration timeline
cd /root/repo; OUT=/root/repo/gpurun_out/r04k; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --no-cpu --no-profile --no-relaxed --in-flight 0 --steps 1 --warmup 1 --iters 60 > $OUT/kt.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_60iters.csv
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/kt/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# keep the last step: find the last k_warp
idx=[i for i,r in enumerate(rows) if 'k_warp' in r['Kernel_Name']]
rows=rows[idx[-1]:]
t0=int(rows[0]['Start_Timestamp'])
out=open('/root/repo/gpurun_out/r04k/timeline_last_step.csv','w')
out.write('start_us,end_us,dur_us,queue,kernel\n')
for r in rows:
    s=(int(r['Start_Timestamp'])-t0)/1e3; e=(int(r['End_Timestamp'])-t0)/1e3
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0]
    out.write(f"{s:.1f},{e:.1f},{e-s:.1f},{r.get('Queue_Id','')},{n}\n")
out.close()
PY
head -3 $OUT/timeline_last_step.csv; wc -l $OUT/timeline_last_step.csv
