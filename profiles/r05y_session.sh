#!/bin/bash
# Round 5, session y: the tree after the lazy librccl binding and the one-flooding-rank exchange: build() + smoke(), full -m gpu suite, default bench line.
cd /root/repo; OUT=/root/repo/gpurun_out/r05y; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 2700 python -m pytest tests -x -q -m gpu --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -12 $OUT/pytest_gpu.log
python bench.py --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r05y/bench_default.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
PY
grep -c rccl /proc/self/maps; python - <<'PY'
import sys; sys.path.insert(0, "/root/repo")
from planet_heightmap_generation_amd import capi
capi.lib()
print("librccl mapped after loading libworogen:", any("rccl" in l for l in open("/proc/self/maps")))
PY
