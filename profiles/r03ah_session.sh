#!/bin/bash
# full GPU suite + default-shaped bench after: one-launch ice/carve, batched solve setup, 6-entry event lists
cd /root/repo; mkdir -p gpurun_out/r03ah
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03ah/gpu_tests.log 2>&1; tail -3 gpurun_out/r03ah/gpu_tests.log
timeout 600 python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03ah/bench.json 2> gpurun_out/r03ah/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03ah/bench.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],1), round(d["value"],1), d["parity"], {k:round(v,1) for k,v in d["stage_ms_last_step"].items()})
PY
