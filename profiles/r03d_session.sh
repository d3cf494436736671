#!/bin/bash
# Round-3 session D: basin solve with the result store moved out of the level loop; per-phase clocks of the kernel
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03d; mkdir -p $O
B="python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1"
timeout 600 $B > $O/bench_default.log 2>&1
WO_BASIN_KEY_BITS=12 timeout 600 $B > $O/bench_keybits12.log 2>&1
for n in 2 30 120 190; do WO_BASIN_STATS=$n timeout 600 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 0 --no-profile 2>&1 | grep "basin stats" >> $O/basin_stats.txt; done
for f in bench_default bench_keybits12; do echo == $f; grep "^{" $O/$f.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d.get('stage_ms_last_step')); fam=d['roofline']['families']; print({k:(v['ms'],v['launches']) for k,v in fam.items() if 'basin' in k or 'solve' in k}); print({k:v for k,v in d['erode_stats'].items() if 'basin' in k})
" || tail -5 $O/$f.log; done
cat $O/basin_stats.txt
