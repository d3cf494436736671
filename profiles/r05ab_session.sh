#!/bin/bash
# Round 5, session ab: where the flood stage's run-to-run spread (70-84 ms per step) comes from: the flood's laps over 3 x 6 timed steps.
cd /root/repo; OUT=/root/repo/gpurun_out/r05ab; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
lscpu | grep -E "Model name|Socket|Core|Thread|L2|L3|NUMA" > $OUT/lscpu.txt
for rep in 1 2 3; do
WO_FLOOD_TIMING=1 python bench.py --timed-only --steps 6 --warmup 2 > $OUT/bench_$rep.json 2> $OUT/flood_laps_$rep.txt
done
python - <<'PY'
import json,re
for rep in (1,2,3):
    d=json.loads(open(f"/root/repo/gpurun_out/r05ab/bench_{rep}.json").read().strip().splitlines()[-1])
    t=open(f"/root/repo/gpurun_out/r05ab/flood_laps_{rep}.txt").read()
    walks=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
    joined=[float(x) for x in re.findall(r"round joined at ([\d.]+) ms", t)]
    host=[float(x) for x in re.findall(r"host passes\s+([\d.]+) ms", t)]
    print(rep, round(d["ms_per_step"],1), d["stage_ms_last_step"]["priority_flood"])
    print("  walks", walks[-12:]); print("  joined", joined[-12:]); print("  host passes", host[-12:])
PY
cat $OUT/lscpu.txt
