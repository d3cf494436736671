#!/bin/bash
# six planets in flight: which of the host-side changes costs throughput there?
cd /root/repo; mkdir -p gpurun_out/r03be
export TMPDIR=/tmp
for V in default nopool nopoll sidelow checkpass; do
  unset WO_HOST_POOL WO_POLL_COUNTS WO_SIDE_PRIORITY WO_SOLVE_CHECK
  case $V in nopool) export WO_HOST_POOL=0;; nopoll) export WO_POLL_COUNTS=0;; sidelow) export WO_SIDE_PRIORITY=0;; checkpass) export WO_SOLVE_CHECK=pass;; esac
  timeout 300 python bench.py --no-cpu --no-profile --steps 1 --warmup 1 > gpurun_out/r03be/$V.json 2> gpurun_out/r03be/$V.err
  python - $V <<'PY'
import json,sys
d=json.loads(open(f"gpurun_out/r03be/{sys.argv[1]}.json").read().strip().splitlines()[-1])
e=d["ensemble_in_flight"]; print(sys.argv[1], "single", round(d["ms_per_step"],1), "in flight", round(e["value"],1), round(e["ms_per_planet"],1))
PY
done
uptime
