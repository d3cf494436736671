#!/bin/bash
# sweep of the late polling cap of the patch solve: bash research/ab/run_late.sh "from:cap" ...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab
for v in "$@"; do
  WO_SOLVE_LATE_FROM=${v%%:*} WO_SOLVE_LATE_SPINS=${v##*:} timeout 300 python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1 > gpurun_out/ab/bench_late_$v.log 2>&1
  grep "^{" gpurun_out/ab/bench_late_$v.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('late $v', round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], 'solve', d['stage_ms_last_step']['solve'], 'patch', d['roofline']['families']['solve_patch']['ms'], d['roofline']['families']['solve_patch']['launches'])
"
done
