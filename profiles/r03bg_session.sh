#!/bin/bash
# Round-3 session K: PMC FETCH/WRITE per kernel at the full 200 iterations (one planet), the default bench command, rocprofv3 stats of it
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03bg; mkdir -p $O
bash profiles/collect_pmc.sh r03bg 200 > $O/collect_pmc.log 2>&1
cp gpurun_out/pmc_r03bg_summary.json $O/pmc_fetch_write_per_kernel_10m_200iters.json; cp gpurun_out/pmc_r03bg_summary.json profiles/r03_pmc_fetch_write_per_kernel_10m_200iters.json
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o st -- python bench.py --no-cpu --in-flight 0 > $O/bench_rocprof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_default_bench_command.csv; rm -rf $O/prof
tail -3 $O/collect_pmc.log
grep "^{" $O/bench_default.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d['roofline']; print(round(d['value'],1), round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d.get('stage_ms_last_step'))
    print('dominant', r['kernel'], r['frac'], r['achieved'], r['traffic'], r['algorithmic_bytes_per_launch'], r['avg_launch_us'])
    print({k:(v['ms'],v['frac']) for k,v in r['passes'].items()}); print(r['whole_stack']); print(d['cpu_baseline']); print(d['ensemble_in_flight'])
" || tail -20 $O/bench_default.log
