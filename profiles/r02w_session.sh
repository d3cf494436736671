#!/bin/bash
# Round-2 measurement session with the patch-major mirror (one gpurun call):  bash profiles/r02w_session.sh
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r02w; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.log
bash profiles/collect_pmc.sh r02w 6 > $O/collect_pmc.log 2>&1
cp gpurun_out/pmc_r02w_summary.json profiles/r02w_pmc_fetch_write_per_kernel_10m.json; cp gpurun_out/pmc_r02w_summary.json $O/pmc_fetch_write_per_kernel_10m.json
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o st -- python bench.py --no-cpu --in-flight 0 > $O/bench_rocprof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_default_bench_command.csv; rm -rf $O/prof
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --backend gloo --share-gpu --steps 2 --warmup 1 --no-cpu > $O/bench_decomposed_rehearsal_10m_2ranks_one_gpu.log 2>&1
timeout 900 python bench.py --mode decomposed --shares 8 --steps 1 --warmup 1 --no-cpu --in-flight 0 > $O/bench_10m_decomposed_8shares_one_gpu.log 2>&1
timeout 900 python bench.py --cells 40000000 --iters 20 --no-cpu --in-flight 0 --steps 2 --warmup 1 > $O/bench_40m_20iters_single_gpu.log 2>&1
timeout 600 python bench.py --cells 1000000 --no-cpu --in-flight 0 --steps 2 --warmup 1 > $O/bench_1m.log 2>&1
WO_LAYOUT=index timeout 600 python bench.py --no-cpu --in-flight 0 --steps 2 --warmup 1 > $O/bench_10m_index_layout.log 2>&1
tail -c 400 $O/pytest_gpu.log; for f in bench_default bench_40m_20iters_single_gpu bench_decomposed_rehearsal_10m_2ranks_one_gpu bench_10m_decomposed_8shares_one_gpu bench_1m bench_10m_index_layout; do echo == $f; grep "^{" $O/$f.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d['value'],1), round(d['ms_per_step'],1), d['parity'], d.get('stage_ms_last_step'), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'), (d.get('roofline') or {}).get('traffic'), d.get('decomposition'))
"; done
