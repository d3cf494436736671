#!/bin/bash
# Round-3 session B: basin solve with hand-over phases, layout kernels in Morton order; knob A/B; per-kernel trace
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03b; mkdir -p $O
B="python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1"
timeout 600 $B > $O/bench_default.log 2>&1
WO_BASIN_COMPACT=0 timeout 600 $B > $O/bench_nocompact.log 2>&1
WO_BASIN_RANGE=4096 timeout 600 $B > $O/bench_range4096.log 2>&1
WO_BASIN_KEY_BITS=22 timeout 600 $B > $O/bench_keybits22.log 2>&1
WO_BASIN_KEY_BITS=12 timeout 600 $B > $O/bench_keybits12.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tr -- python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 0 --no-profile > $O/bench_trace.log 2>&1
python profiles/per_iteration_durations.py $O/prof 200 > $O/per_iteration_durations_10m.json 2>$O/per_iter.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_one_step.csv; rm -rf $O/prof
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log
for f in bench_default bench_nocompact bench_range4096 bench_keybits22 bench_keybits12; do echo == $f; grep "^{" $O/$f.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d.get('stage_ms_last_step')); fam=d['roofline']['families']; print({k:(v['ms'],v['launches']) for k,v in fam.items() if 'basin' in k or 'solve' in k}); print({k:v for k,v in d['erode_stats'].items() if 'basin' in k})
" || tail -5 $O/$f.log; done
head -30 $O/rocprofv3_kernel_stats_one_step.csv | cut -c1-150
tail -5 $O/pytest_gpu.log
