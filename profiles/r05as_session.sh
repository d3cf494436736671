#!/bin/bash
# Round 5, session as: the land lists of erodeComposite kept while the ocean mask stays the same (setup stage): full -m gpu suite, bench A/B against WO_NO_LAND_LIST_CACHE=1.
cd /root/repo; OUT=/root/repo/gpurun_out/r05as; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2700 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
for rep in 1 2 3; do
python bench.py --timed-only --steps 12 --warmup 2 > $OUT/bench_kept_$rep.json 2> /dev/null
WO_NO_LAND_LIST_CACHE=1 python bench.py --timed-only --steps 12 --warmup 2 > $OUT/bench_rebuilt_$rep.json 2> /dev/null
done
python - <<'PY'
import json
for tag in ("kept","rebuilt"):
    v=[]
    for rep in (1,2,3):
        d=json.loads(open(f"/root/repo/gpurun_out/r05as/bench_{tag}_{rep}.json").read().strip().splitlines()[-1]); v.append((round(d["ms_per_step"],1), d["stage_ms_last_step"]["setup"], d["parity"]["parity_crc_ok"]))
    print(tag, v, "mean", round(sum(a for a,b,c in v)/3,1))
PY
