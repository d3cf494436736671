#!/bin/bash
# Round 5, session z: the thermal step leaves the next sort's keys behind (k_thermal_apply_reg: keysOut).  Route tests, default bench with and without.
cd /root/repo; OUT=/root/repo/gpurun_out/r05z; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python -m pytest tests -x -q -m gpu -k "sort or graph or relaxed or golden or headline" > $OUT/pytest_gpu_sort.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_sort.log
tail -6 $OUT/pytest_gpu_sort.log
for rep in 1 2; do
python bench.py --timed-only --steps 6 --warmup 2 > $OUT/bench_keys_from_thermal_$rep.json 2> $OUT/err1.txt
WO_SORT_KEYS=pass python bench.py --timed-only --steps 6 --warmup 2 > $OUT/bench_keys_by_pass_$rep.json 2> $OUT/err2.txt
done
python - <<'PY'
import json
for f in ("bench_keys_from_thermal_1","bench_keys_by_pass_1","bench_keys_from_thermal_2","bench_keys_by_pass_2"):
    d=json.loads(open(f"/root/repo/gpurun_out/r05z/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
PY
