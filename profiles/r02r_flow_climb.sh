#!/bin/bash
# Flow accumulation: cap of the one-launch rake (k_flow_climb, packed count+total), 10 M cells x 200 iterations
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r02r; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 400 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 1 > $O/$name.log 2>&1; python - $O/$name.log $name <<'P'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        ok=True
        d=json.loads(l); st=d['stage_ms_last_step']; es=d['erode_stats']; fam=d['roofline']['families']
        print(sys.argv[2], 'ms/step %.0f'%d['ms_per_step'], 'crc', d['parity']['parity_crc_ok'], 'flow %.1f'%st['flow'], 'climb+snap %.1f (%d launches)'%(fam['flow_snap']['ms'], fam['flow_snap']['launches']), 'apply %.1f'%fam['flow_apply']['ms'], 'flow rounds', es['flow_rounds_total'])
if not ok: print(open(sys.argv[1]).read()[-1500:])
P
}
run cap128 WO_FLOW_CLIMB=128
run cap256 WO_FLOW_CLIMB=256
run cap1024 WO_FLOW_CLIMB=1024
run cap1M WO_FLOW_CLIMB=1000000
run cap128_again WO_FLOW_CLIMB=128
