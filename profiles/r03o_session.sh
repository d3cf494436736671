#!/bin/bash
# Round-3 session O: basin layout on a side stream beside the flow accumulation; comm test; GPU suite
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03o; mkdir -p $O
B="python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1"
timeout 600 $B > $O/bench_overlap.log 2>&1

timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log
for f in $O/bench_*.log; do echo == $f; grep "^{" $f | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d.get('stage_ms_last_step'))
" || tail -5 $f; done
tail -4 $O/pytest_gpu.log
