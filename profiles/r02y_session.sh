#!/bin/bash
# Last round-2 measurement session (one gpurun call):  bash profiles/r02y_session.sh
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r02y; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o kt -- python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 0 > $O/bench_trace.log 2>&1
python profiles/per_iteration_durations.py $O/trace 200 > $O/per_iteration_durations_10m.json; rm -rf $O/trace
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o st -- python bench.py --no-cpu --in-flight 0 > $O/bench_rocprof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_default_bench_command.csv; rm -rf $O/prof
grep "^{" $O/bench_default.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity'], d.get('stage_ms_last_step'))
"
python - <<'P'
import json
d=json.load(open('gpurun_out/r02y/per_iteration_durations_10m.json'))
for k,v in d.items():
    if 'us_mean_by_tenth_of_the_run' in v: print(k, v['us_min'], v['us_median'], v['us_max'], v['us_mean_by_tenth_of_the_run'])
P
