#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03au
export TMPDIR=/tmp
for V in pool nopool pool2; do
  unset WO_HOST_POOL; [ $V = nopool ] && export WO_HOST_POOL=0
  timeout 300 python bench.py --no-cpu --no-profile --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03au/$V.json 2> gpurun_out/r03au/$V.err
  python - $V <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r03au/{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],1), {k:round(v,1) for k,v in d["stage_ms_last_step"].items() if k in ("setup","priority_flood")}, d["erode_stats"]["flood_host_pass1_ms"])
PY
done
uptime
