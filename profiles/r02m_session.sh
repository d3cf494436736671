#!/bin/bash
# GPU suite + default bench with the patch-major mirror (erodeComposite + warp)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r02m; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
timeout 600 python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1 > $O/bench.log 2>&1
python - $O/bench.log <<'P'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); st=d['stage_ms_last_step']; es=d['erode_stats']; fam=d['roofline']['families']
        print('ms/step %.0f'%d['ms_per_step'], 'value %.0f'%d['value'], 'crc', d['parity']['parity_crc_ok'], 'launches', es['solve_patch_launches_total'], 'cold', round(d['cold_first_step_ms']))
        print('  stages', {k: round(v) for k, v in st.items()})
        print('  families', {k: round(v['ms'],1) for k, v in fam.items() if v['ms'] > 3})
P
