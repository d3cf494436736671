#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03ay
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or against_oracle_large or ties or edge_cases or sort_routes or basin_leftovers or glacial_step or flow_accumulation or mirror_layout" 2>&1 | tail -2
timeout 300 python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1 > gpurun_out/r03ay/b.json 2> gpurun_out/r03ay/b.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03ay/b.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], "solve stage", round(d["stage_ms_last_step"]["solve"],1), d["roofline"]["families"].get("solve_basin"))
PY
