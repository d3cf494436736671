#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03ba
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or against_oracle_large or ties or edge_cases or flow_accumulation or mirror_layout or basin_leftovers or sort_routes or glacial_step or pipeline_matches" 2>&1 | tail -2
for V in keys nokeys keys2 nokeys2; do
  unset WO_KEYS_FROM_APPLY; case $V in nokeys*) export WO_KEYS_FROM_APPLY=0;; esac
  timeout 300 python bench.py --no-cpu --no-profile --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03ba/$V.json 2> gpurun_out/r03ba/$V.err
  python - $V <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r03ba/{sys.argv[1]}.json").read().strip().splitlines()[-1])
st=d["stage_ms_last_step"]; print(sys.argv[1], round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], "flood", round(st["priority_flood"],1), "rest", round(d["ms_per_step"]-st["priority_flood"],1), {k:round(v,1) for k,v in st.items() if k in ("solve","thermal","sort","flow","receivers")})
PY
done
