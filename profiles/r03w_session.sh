#!/bin/bash
# Round-3 session W: lean solve setup (outputs cleared by a memset, blocker hints only when a launch leaves tasks pending) + the leftover test
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03w; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log; grep -n "passed\|failed\|Error" $O/pytest_gpu.log | tail -4
timeout 600 python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1 > $O/bench.log 2>&1
grep "^{" $O/bench.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d.get('stage_ms_last_step')); print({k:(v['ms'],v['launches']) for k,v in d['roofline']['families'].items() if 'solve' in k})
"
