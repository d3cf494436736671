#!/bin/bash
# Round-3 session F: basin solve as a template over the window size (records through LDS, aggregated counting sort,
# precomputed range starts): window x range matrix, per-phase clocks, GPU suite
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03f; mkdir -p $O
B="python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1"
for wr in 512:1024 512:512 256:256 256:1024 1024:1024; do
  w=${wr%%:*}; r=${wr##*:}
  WO_BASIN_WINDOW=$w WO_BASIN_RANGE=$r timeout 600 $B > $O/bench_w${w}_r${r}.log 2>&1
  WO_BASIN_WINDOW=$w WO_BASIN_RANGE=$r WO_BASIN_STATS=30 timeout 600 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 0 --no-profile 2>&1 | grep "basin stats" >> $O/basin_stats.txt
done
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log
for f in $O/bench_w*.log; do echo == $f; grep "^{" $f | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d.get('stage_ms_last_step')); fam=d['roofline']['families']; print({k:(v['ms'],v['launches']) for k,v in fam.items() if 'basin' in k or 'solve' in k}); print({k:v for k,v in d['erode_stats'].items() if 'basin' in k})
" || tail -5 $f; done
cat $O/basin_stats.txt; tail -4 $O/pytest_gpu.log
