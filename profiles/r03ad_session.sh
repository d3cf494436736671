#!/bin/bash
# one-launch ice accumulation + one-launch carve: parity first, then timing A/B
cd /root/repo; mkdir -p gpurun_out/r03ad
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r03ad/gpu_parity.log 2>&1; tail -3 gpurun_out/r03ad/gpu_parity.log
timeout 600 python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03ad/bench_flow.json 2> gpurun_out/r03ad/bench_flow.err
WO_CARVE_FLOW=0 WO_ICE_ROUNDS=1 timeout 600 python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03ad/bench_rounds.json 2> gpurun_out/r03ad/bench_rounds.err
python - <<'PY'
import json
for n in ("flow","rounds"):
    try:
        d=json.loads(open(f"gpurun_out/r03ad/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"],1), d["parity"], {k:round(v,1) for k,v in d["stage_ms_last_step"].items()}, {k:v for k,v in d["erode_stats"].items() if "carve" in k or "ice" in k})
    except Exception as e: print(n, "failed", e)
PY
