#!/bin/bash
# Round-3 session J: the flood's replay of the single heap at 40 M cells (CRC tests at config 4's size and config 5's seeds), 40 M bench with flood timing
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03j; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "config4 or config5" 2>&1 | tail -15 > $O/pytest_new.log
WO_FLOOD_TIMING=1 timeout 900 python bench.py --cells 40000000 --iters 20 --no-cpu --in-flight 0 --steps 2 --warmup 1 > $O/bench_40m_20iters.log 2>&1
WO_FLOOD_TIMING=1 timeout 900 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 1 --no-profile > $O/bench_10m_flood_timing.log 2>&1
tail -15 $O/pytest_new.log
grep "^\[flood\]" $O/bench_40m_20iters.log | tail -24
grep "^{" $O/bench_40m_20iters.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity'], d.get('stage_ms_last_step')); print({k:v for k,v in d['erode_stats'].items() if 'flood' in k})
"
grep "^\[flood\]" $O/bench_10m_flood_timing.log | tail -8
