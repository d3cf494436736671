#!/bin/bash
# Round 5, session bd: soak — 300 timed steps of the benched planet in one process (the lock-free kernels' rare paths: a call run again with checks would show up in the
# stats; the CRC of the last field is checked as always), then 6 planets in flight for 20 rounds.
cd /root/repo; OUT=/root/repo/gpurun_out/r05bd; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --timed-only --steps 300 --warmup 2 > $OUT/bench_300_steps.json 2> $OUT/err.txt; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r05bd/bench_300_steps.json").read().strip().splitlines()[-1])
s=d["erode_stats"]
print(d["steps"], round(d["ms_per_step"],2), round(d["value"],1), d["parity"], "redo", s.get("calls_run_again_with_checks"), "leftover passes", s.get("solve_basin_passes_with_leftovers"), "carve leftovers", s.get("carve_flow_launches_with_leftovers"), "replays", s.get("flood_host_replays"))
PY
