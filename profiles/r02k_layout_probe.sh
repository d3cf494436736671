#!/bin/bash
# Layout probe (one gpurun call): the 10 M-cell bench step with the cells in Fibonacci index order (the reference's) against the
# same planet renumbered in Morton order (WO_BENCH_LAYOUT=morton; experiment only: ids enter the semantics, CRC differs by construction).
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r02k; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 1 > $O/$name.log 2>&1; python - $O/$name.log $name <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); st=d['stage_ms_last_step']; es=d['erode_stats']; fam=d['roofline']['families']
        print(sys.argv[2], 'ms/step %.0f'%d['ms_per_step'], 'crc', d['parity']['parity_crc_ok'], 'launches', es['solve_patch_launches_total'])
        print('  stages', {k: round(v) for k, v in st.items()})
        print('  families', {k: round(v['ms'],1) for k, v in fam.items() if v['ms'] > 3})
P
}
run fib WO_X=1
run morton WO_BENCH_LAYOUT=morton
