#!/bin/bash
# Round 5, session aa: EXPERIMENT — the basin solve's task records kept by CELL (solve_setup writes consecutive records; the walking kernels find a
# slot's record through the layout's slot -> cell list) instead of scattered to the store index.  WO_TASK_BY_CELL=1.
cd /root/repo; OUT=/root/repo/gpurun_out/r05aa; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
WO_TASK_BY_CELL=1 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or headline or basin or routes or scramble or leftover or checked or oracle_large" > $OUT/pytest_task_by_cell.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_task_by_cell.log
tail -5 $OUT/pytest_task_by_cell.log
for rep in 1 2; do
WO_TASK_BY_CELL=1 python bench.py --timed-only --steps 6 --warmup 2 > $OUT/bench_task_by_cell_$rep.json 2> $OUT/err1.txt
python bench.py --timed-only --steps 6 --warmup 2 > $OUT/bench_task_by_slot_$rep.json 2> $OUT/err2.txt
done
python - <<'PY'
import json
for f in ("bench_task_by_cell_1","bench_task_by_slot_1","bench_task_by_cell_2","bench_task_by_slot_2"):
    d=json.loads(open(f"/root/repo/gpurun_out/r05aa/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
PY
