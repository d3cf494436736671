#!/bin/bash
# solve launch: waves per workgroup x slots per range
cd /root/repo; mkdir -p gpurun_out/r03ax
export TMPDIR=/tmp
for V in "4 256" "8 512" "8 256" "16 1024" "16 512"; do
  set -- $V
  WO_BASIN_WAVES=$1 WO_BASIN_RANGE=$2 timeout 300 python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1 > gpurun_out/r03ax/w$1_r$2.json 2> gpurun_out/r03ax/w$1_r$2.err
  python - $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(f"gpurun_out/r03ax/w{sys.argv[1]}_r{sys.argv[2]}.json").read().strip().splitlines()[-1])
    f=d["roofline"]["families"].get("solve_basin",{})
    print(sys.argv[1:], round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], "solve stage", round(d["stage_ms_last_step"]["solve"],1), "coop", f)
except Exception as e: print(sys.argv[1:], "failed", e)
PY
done
