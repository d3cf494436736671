#!/bin/bash
# Round-3 session L: carve rounds over the static activation list with eager loads; GPU suite; bench
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03m; mkdir -p $O
B="python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1"
timeout 600 $B > $O/bench_static_rounds.log 2>&1


for f in $O/bench_*.log; do echo == $f; grep "^{" $f | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d.get('stage_ms_last_step')); fam=d['roofline']['families']; print({k:(v['ms'],v['launches']) for k,v in fam.items() if 'carve' in k or 'ice' in k}); print(d['roofline']['kernel'], d['roofline']['frac'], {k:v for k,v in d['erode_stats'].items() if 'carve' in k})
" || tail -5 $f; done

