#!/bin/bash
# Round-3 session AA: rehearsal of bench.py's N > 1 paths on a one-GPU box (2 ranks sharing the GPU, gloo): default (ensemble) and --mode decomposed
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03bi; mkdir -p $O
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --backend gloo --share-gpu --steps 2 --warmup 1 --no-cpu > $O/bench_rehearsal_ensemble_2ranks_one_gpu.log 2>&1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 2 --backend gloo --share-gpu --steps 2 --warmup 1 --no-cpu --mode decomposed > $O/bench_rehearsal_decomposed_2ranks_one_gpu.log 2>&1
for f in bench_rehearsal_ensemble_2ranks_one_gpu bench_rehearsal_decomposed_2ranks_one_gpu; do echo == $f; grep "^{" $O/$f.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['n_gpus'], d['scaling'], round(d['value'],1), round(d['ms_per_step'],1), d['parity'], d['config']['parallelism'][:80], d.get('decomposition'), d['host_threads'])
" || tail -8 $O/$f.log; done
