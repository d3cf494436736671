#!/bin/bash
# Round-3 session U: landmass walks with a 4-ary heap vs the binary heap (flood timing on the GPU box's host)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03u; mkdir -p $O
for h in 4 2 4 2; do WO_FLOOD_HEAP=$h WO_FLOOD_TIMING=1 timeout 600 python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1 --no-profile > $O/bench_heap$h.log 2>&1; echo "heap $h:" $(grep "largest landmass\|pipeline" $O/bench_heap$h.log | tail -4 | sed 's/\[flood\] //' | tr '\n' ';') $(grep "^{" $O/bench_heap$h.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), d['stage_ms_last_step']['priority_flood'], d['parity']['parity_crc_ok'])"); done
