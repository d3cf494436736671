#!/bin/bash
# Patch-major mirror inside erodeComposite: smoke + 10 M bench (CRC vs oracle) with the mirror and with WO_LAYOUT=index
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r02l; mkdir -p $O
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 1 > $O/$name.log 2>&1; python - $O/$name.log $name <<'P'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        ok=True
        d=json.loads(l); st=d['stage_ms_last_step']; es=d['erode_stats']; fam=d['roofline']['families']
        print(sys.argv[2], 'ms/step %.0f'%d['ms_per_step'], 'crc', d['parity']['parity_crc_ok'], 'launches', es['solve_patch_launches_total'], 'cold', round(d['cold_first_step_ms']))
        print('  stages', {k: round(v) for k, v in st.items()})
        print('  families', {k: round(v['ms'],1) for k, v in fam.items() if v['ms'] > 3})
if not ok: print(open(sys.argv[1]).read()[-1500:])
P
}
run mirror WO_X=1
run index WO_LAYOUT=index
