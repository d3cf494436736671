#!/bin/bash
# Round-3 session Q: neighbour windows staged in LDS (receivers, thermal_excess, thermal_apply); GPU suite; per-kernel stats
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03q; mkdir -p $O
B="python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1"
timeout 600 $B > $O/bench_lds_tiles.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tr -- python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 0 --no-profile > $O/bench_trace.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_one_step.csv; rm -rf $O/prof
for f in $O/bench_*.log; do echo == $f; grep "^{" $f | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d.get('stage_ms_last_step')); print({k:(v['ms'],v['frac']) for k,v in d['roofline']['passes'].items()})
" || tail -5 $f; done
grep -n "passed\|failed" $O/pytest_gpu.log | tail -2
grep "k_receivers\|k_thermal" $O/rocprofv3_kernel_stats_one_step.csv | cut -d, -f1-4 | cut -c1-120
