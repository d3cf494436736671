#!/bin/bash
# Round 5, session av: the stage brackets of erodeComposite are turned into milliseconds when wo_last_stage_timing is asked, not at the end of the call: full -m gpu suite, bench.
cd /root/repo; OUT=/root/repo/gpurun_out/r05av; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2700 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
for rep in 1 2 3; do python bench.py --timed-only --steps 12 --warmup 2 > $OUT/bench_$rep.json 2> /dev/null; done
python - <<'PY'
import json
v=[]
for rep in (1,2,3):
    d=json.loads(open(f"/root/repo/gpurun_out/r05av/bench_{rep}.json").read().strip().splitlines()[-1]); v.append((round(d["ms_per_step"],1), d["stage_ms_last_step"]["setup"], d["parity"]["parity_crc_ok"]))
print(v, "mean", round(sum(a for a,b,c in v)/3,1))
PY
