#!/bin/bash
# Round 5, session w: one rank floods the whole planet per undecided flood call and hands the land heights back (flood exchange phases 2 / 3).
# The multi-share / multi-process GPU tests, then the default --gpus 2 bench command rehearsed with two gloo ranks sharing this one GPU (not a measurement).
cd /root/repo; OUT=/root/repo/gpurun_out/r05w; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -x -q -m gpu -k "config4 or decomposed or exchange or two_ranks or shares" --durations=8 > $OUT/pytest_gpu_shares.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_shares.log
tail -14 $OUT/pytest_gpu_shares.log
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-gpu --steps 2 --warmup 1 > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2_rehearsal.err; echo "rehearsal rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("/root/repo/gpurun_out/r05w/bench_gpus2_rehearsal.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"])
o=d["one_planet"]; print(round(o["ms_per_step"],1), round(o["value"],1), o["parity"], o["per_rank"])
PY
tail -5 $OUT/bench_gpus2_rehearsal.err
