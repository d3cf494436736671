#!/bin/bash
# Round-3 session N: GPU suite + smoke with the eager static carve rounds (24 dependency slots, rows up to 12)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03n; mkdir -p $O
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log | cut -c1-200
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
