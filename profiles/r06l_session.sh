#!/bin/bash
# Round 6, session l: the final tree (after the prepared-division experiment left the solve turn): smoke, the driver's GPU command, a short bench with CRC.
cd /root/repo; OUT=/root/repo/gpurun_out/r06l; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log | cut -c1-100
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2> $OUT/pytest_gpu.err; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -8 $OUT/pytest_gpu.log | grep -E "passed|failed|rc="
python bench.py --timed-only --steps 8 --warmup 2 > $OUT/bench_timed_only.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r06l/bench_timed_only.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
PY
