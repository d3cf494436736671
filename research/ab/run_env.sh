#!/bin/bash
# sweep of one environment variable over the in-tree build: bash research/ab/run_env.sh VAR v1 v2 ...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab; VAR=$1; shift
for v in "$@"; do
  env $VAR=$v timeout 300 python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1 > gpurun_out/ab/bench_${VAR}_$v.log 2>&1
  grep "^{" gpurun_out/ab/bench_${VAR}_$v.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$VAR=$v', round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], 'solve', d['stage_ms_last_step']['solve'], 'patch', d['roofline']['families']['solve_patch']['ms'], d['roofline']['families']['solve_patch']['launches'])
"
done
