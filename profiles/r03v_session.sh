#!/bin/bash
# Round-3 final-ish measurement session (one gpurun call):  bash profiles/r03v_session.sh
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03v; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.log
bash profiles/collect_pmc.sh r03v 200 > $O/collect_pmc.log 2>&1
cp gpurun_out/pmc_r03v_summary.json $O/pmc_fetch_write_per_kernel_10m_200iters.json; cp gpurun_out/pmc_r03v_summary.json profiles/r03_pmc_fetch_write_per_kernel_10m_200iters.json
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o st -- python bench.py --no-cpu --in-flight 0 > $O/bench_rocprof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_default_bench_command.csv; rm -rf $O/prof
timeout 900 python bench.py --mode decomposed --shares 8 --steps 1 --warmup 1 --no-cpu --in-flight 0 > $O/bench_10m_decomposed_8shares_one_gpu.log 2>&1
timeout 900 python bench.py --cells 40000000 --iters 20 --no-cpu --in-flight 0 --steps 2 --warmup 1 > $O/bench_40m_20iters_single_gpu.log 2>&1
timeout 600 python bench.py --cells 1000000 --no-cpu --in-flight 0 --steps 2 --warmup 1 > $O/bench_1m.log 2>&1
tail -c 300 $O/pytest_gpu.log; for f in bench_default bench_10m_decomposed_8shares_one_gpu bench_40m_20iters_single_gpu bench_1m; do echo == $f; grep "^{" $O/$f.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); r=d.get('roofline') or {}; print(round(d['value'],1), round(d['ms_per_step'],1), d['parity'], d.get('stage_ms_last_step'), r.get('kernel'), r.get('frac'), r.get('traffic'), {k:v['frac'] for k,v in (r.get('passes') or {}).items()}, (d.get('decomposition') or {}).get('share_ms_last_step'), (d.get('decomposition') or {}).get('projected_speedup_one_gpu_per_share'), (d.get('cpu_baseline') or {}).get('value'), (d.get('ensemble_in_flight') or {}).get('value'))
" || tail -5 $O/$f.log; done
