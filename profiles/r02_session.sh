#!/bin/bash
# Round-2 measurement session on the GPU box (one gpurun call):  bash profiles/r02_session.sh
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o st -- python bench.py --no-cpu --in-flight 0 > $O/bench_rocprof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats_default_bench_command.csv; rm -rf $O/prof
bash profiles/collect_pmc.sh r02 6 > $O/collect_pmc.log 2>&1
bash profiles/collect_sq.sh r02 6 > $O/collect_sq.log 2>&1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --backend gloo --share-gpu --steps 2 --warmup 1 --no-cpu > $O/bench_decomposed_rehearsal_10m_2ranks_one_gpu.log 2>&1
timeout 900 python bench.py --cells 40000000 --iters 20 --no-cpu --in-flight 0 --steps 2 --warmup 1 > $O/bench_40m_20iters_single_gpu.log 2>&1
WO_FLOOD=device WO_FLOOD_TIES=id timeout 600 python profiles/flood_probe.py 10000000 200 > $O/flood_device_idorder_10m.log 2>&1
WO_FLOOD=device timeout 600 python profiles/flood_probe.py 1000000 200 > $O/flood_device_1m.log 2>&1
WO_SOLVE_STATS=60 timeout 300 python bench.py --steps 1 --warmup 0 --no-cpu --no-profile --in-flight 0 > /dev/null 2> $O/solve_visit_counters_iteration60.txt
WO_ELEV_TIMING=1 timeout 600 python profiles/time_elevation.py 10000000 > $O/time_elevation_10m.log 2>&1
tail -c 400 $O/pytest_gpu.log; for f in bench_default bench_40m_20iters_single_gpu bench_decomposed_rehearsal_10m_2ranks_one_gpu; do echo == $f; tail -c 600 $O/$f.log | head -c 600; echo; done
