#!/bin/bash
# Round-3 session P: cooperative streaming solve (4 / 2 / 1 waves per range), GPU suite, per-iteration trace
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03p; mkdir -p $O
B="python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1"
for w in 4 2 1; do WO_BASIN_WAVES=$w timeout 600 $B > $O/bench_waves$w.log 2>&1; done
WO_BASIN_WAVES=4 WO_BASIN_RANGE=1024 timeout 600 $B > $O/bench_waves4_r1024.log 2>&1
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o tr -- python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 0 --no-profile > $O/bench_trace.log 2>&1
python profiles/per_iteration_durations.py $O/prof 200 > $O/per_iteration_durations_10m.json 2>$O/per_iter.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_one_step.csv; rm -rf $O/prof
for f in $O/bench_w*.log; do echo == $f; grep "^{" $f | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d.get('stage_ms_last_step')); fam=d['roofline']['families']; print({k:(v['ms'],v['launches']) for k,v in fam.items() if 'basin' in k or 'solve' in k}); print({k:v for k,v in d['erode_stats'].items() if 'basin' in k})
" || tail -5 $f; done
tail -4 $O/pytest_gpu.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03p/per_iteration_durations_10m.json'))
for k,v in d.items():
    if 'solve' in k or 'carve' in k: print(k, v['us_min'], v['us_median'], v['us_max'], v['us_mean_by_tenth_of_the_run'], v['us_first_20'][:12])
PY
