#!/bin/bash
# Under the patch-major mirror: rake rounds of the flow accumulation and polling cap of the patch solve (10 M cells, 200 iterations)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r02p; mkdir -p $O
(cd profiles/microbench && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off fastdiv.hip -o /tmp/fastdiv 2>/dev/null && timeout 120 /tmp/fastdiv) > $O/fastdiv.txt 2>&1; cat $O/fastdiv.txt
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 1 > $O/$name.log 2>&1; python - $O/$name.log $name <<'P'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        ok=True
        d=json.loads(l); st=d['stage_ms_last_step']; es=d['erode_stats']; fam=d['roofline']['families']
        print(sys.argv[2], 'ms/step %.0f'%d['ms_per_step'], 'crc', d['parity']['parity_crc_ok'], 'flow %.1f'%st['flow'], 'solve %.1f'%st['solve'], 'patch launches', es['solve_patch_launches_total'], 'patch_ms %.1f'%fam['solve_patch']['ms'], 'flow rounds', es['flow_rounds_total'])
if not ok: print(open(sys.argv[1]).read()[-1500:])
P
}
run base WO_X=1
run rake4 WO_FLOW_RAKE=4
run rake6 WO_FLOW_RAKE=6
run rake12 WO_FLOW_RAKE=12
run rake16 WO_FLOW_RAKE=16
run spins12 WO_SOLVE_SPINS=12
run spins24 WO_SOLVE_SPINS=24
