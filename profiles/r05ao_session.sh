#!/bin/bash
# Round 5, session ao: flood worker count, longer A/B in the bench itself (3 x 12 timed steps each): 16 / 20 / 24 / 32.
cd /root/repo; OUT=/root/repo/gpurun_out/r05ao; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3; do for nt in 24 16 20 32; do
WO_FLOOD_THREADS=$nt python bench.py --timed-only --steps 12 --warmup 2 > $OUT/bench_t${nt}_$rep.json 2> /dev/null
done; done
python - <<'PY'
import json
for nt in (16,20,24,32):
    v=[]
    for rep in (1,2,3):
        d=json.loads(open(f"/root/repo/gpurun_out/r05ao/bench_t{nt}_{rep}.json").read().strip().splitlines()[-1]); v.append(round(d["ms_per_step"],1))
    print(nt, "flood threads:", v, "mean", round(sum(v)/3,1))
PY
