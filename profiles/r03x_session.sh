#!/bin/bash
# Round-3 session X: carve launches of several sub-rounds (block-local dependencies through LDS flags)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03x; mkdir -p $O
B="python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1"
for sub in 8 1 4 16; do WO_CARVE_SUB=$sub timeout 600 $B > $O/bench_sub$sub.log 2>&1; echo "sub $sub:" $(grep "^{" $O/bench_sub$sub.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d['stage_ms_last_step']['glacial'], d['erode_stats']['carve_rounds_total'], d['roofline']['families']['carve_round'])"); done
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log; grep -n "passed\|failed\|Error" $O/pytest_gpu.log | tail -4
