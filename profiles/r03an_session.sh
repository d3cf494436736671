#!/bin/bash
# in-tree radix sort: parity subset, then bench A/B against the library sort
cd /root/repo; mkdir -p gpurun_out/r03an
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or against_oracle_large or ties_on_larger or edge_cases or glacial_step" > gpurun_out/r03an/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03an/tests.log
timeout 300 python bench.py --no-cpu --no-profile --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03an/bench_intree.json 2> gpurun_out/r03an/bench_intree.err; echo "bench rc=$?"
WO_SORT=hipcub timeout 300 python bench.py --no-cpu --no-profile --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03an/bench_hipcub.json 2> gpurun_out/r03an/bench_hipcub.err
python - <<'PY'
import json
for n in ("intree","hipcub"):
    try:
        d=json.loads(open(f"gpurun_out/r03an/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"],1), d["parity"], {k:round(v,1) for k,v in d["stage_ms_last_step"].items()})
    except Exception as e: print(n, "failed", e)
PY
