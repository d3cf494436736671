#!/bin/bash
# Round-3 session A: first contact of the basin-local solve (one gpurun call):  bash profiles/r03a_session.sh
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03a; mkdir -p $O
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1 > $O/bench_basin.log 2>&1
WO_BASIN=0 timeout 900 python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1 > $O/bench_patches.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log
for f in bench_basin bench_patches; do echo == $f; grep "^{" $O/$f.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity'], d.get('stage_ms_last_step')); fam=d['roofline']['families']; print({k:(v['ms'],v['launches']) for k,v in fam.items() if v['ms']>3}); print({k:v for k,v in d['erode_stats'].items() if 'solve' in k})
" || tail -5 $O/$f.log; done
tail -15 $O/pytest_gpu.log
