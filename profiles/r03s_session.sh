#!/bin/bash
# Round-3 session S: carve rounds from packed per-task records; smoke, GPU suite, bench
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03s; mkdir -p $O
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-160
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log; grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
timeout 600 python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1 > $O/bench.log 2>&1
grep "^{" $O/bench.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d.get('stage_ms_last_step'))
"
