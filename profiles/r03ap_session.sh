#!/bin/bash
# in-tree radix sort for both sorts, sampled stage timing: full GPU suite, bench, kernel trace
cd /root/repo; mkdir -p gpurun_out/r03ap
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03ap/gpu_tests.log 2>&1; tail -3 gpurun_out/r03ap/gpu_tests.log
timeout 600 python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03ap/bench.json 2> gpurun_out/r03ap/bench.err
WO_SORT=hipcub timeout 600 python bench.py --no-cpu --no-profile --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03ap/bench_hipcub.json 2> gpurun_out/r03ap/bench_hipcub.err
python - <<'PY'
import json
for n in ("bench","bench_hipcub"):
    d=json.loads(open(f"gpurun_out/r03ap/{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["ms_per_step"],1), round(d["value"],1), d["parity"], {k:round(v,1) for k,v in d["stage_ms_last_step"].items()})
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p -o t -- python /root/repo/bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 1 > /root/repo/gpurun_out/r03ap/trace_bench.log 2>&1
cp $(find /tmp/prof_p -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r03ap/kernel_stats.csv
