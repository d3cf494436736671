#!/bin/bash
# Round 5, session aq: the largest landmass's tree lists built with the idle workers (slices + tickets) and the stamp / path-mark clears folded into the gather sweep.
cd /root/repo; OUT=/root/repo/gpurun_out/r05aq; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
WO_FLOOD_TIMING=1 python bench.py --timed-only --steps 6 --warmup 2 > $OUT/bench_laps.json 2> $OUT/flood_laps.txt
for rep in 1 2 3; do python bench.py --timed-only --steps 12 --warmup 2 > $OUT/bench_$rep.json 2> /dev/null; done
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
python - <<'PY'
import json,re
t=open("/root/repo/gpurun_out/r05aq/flood_laps.txt").read()
w=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
st=[float(x) for x in re.findall(r"started ([\d.]+) ms into", t)]
l=[float(x) for x in re.findall(r"tree lists done at ([\d.]+) ms", t)]
j=[float(x) for x in re.findall(r"round joined at ([\d.]+) ms", t)]
h=[float(x) for x in re.findall(r"host passes\s+([\d.]+) ms", t)]
n=min(len(w),len(l),len(j),len(h))
print("walk", w[-n:][-8:]); print("lists done", l[-n:][-8:]); print("joined", j[-n:][-8:]); print("host passes", h[-n:][-8:])
print("lists - walk", [round(a-b,1) for a,b in zip(l[-8:],w[-8:])], "joined - lists", [round(a-b,1) for a,b in zip(j[-8:],l[-8:])], "host - joined", [round(a-b,1) for a,b in zip(h[-8:],j[-8:])])
for rep in (1,2,3):
    d=json.loads(open(f"/root/repo/gpurun_out/r05aq/bench_{rep}.json").read().strip().splitlines()[-1]); print(rep, round(d["ms_per_step"],1), d["stage_ms_last_step"]["priority_flood"], d["parity"]["parity_crc_ok"])
PY
