#!/bin/bash
# kernel trace (timestamps) of one warm step, for the launch-gap analysis
cd /root/repo; mkdir -p gpurun_out/r03am
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o t -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 1 > /root/repo/gpurun_out/r03am/bench.log 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1); ls -la $f
python - "$f" <<'PY'
import csv, sys, gzip
rows=list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
# keep the second half (warm step) compactly: name(short), queue, start, end
out=open('/root/repo/gpurun_out/r03am/trace_compact.csv','w')
out.write('name,queue,stream,start,end\n')
for r in rows:
    n=r['Kernel_Name'].split('(')[0].replace('void ','').replace('wo::','').replace('(anonymous namespace)::','')[:40]
    if 'rocprim' in r['Kernel_Name']: n='rocprim'
    out.write(f"{n},{r.get('Queue_Id','')},{r.get('Stream_Id','')},{r['Start_Timestamp']},{r['End_Timestamp']}\n")
out.close()
PY
gzip -f /root/repo/gpurun_out/r03am/trace_compact.csv; ls -la /root/repo/gpurun_out/r03am/
