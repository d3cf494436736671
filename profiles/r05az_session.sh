#!/bin/bash
# Round 5, session az: the default --gpus 8 bench command rehearsed with eight gloo ranks sharing this one GPU (NOT a measurement: one device, one host): the ensemble line and
# the config-4 one-planet leg at 200 iterations with the round's flood exchange (one flooding rank per undecided call).
cd /root/repo; OUT=/root/repo/gpurun_out/r05az; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 8 --backend gloo --share-gpu --steps 2 --warmup 1 > $OUT/bench_gpus8_rehearsal.json 2> $OUT/bench_gpus8_rehearsal.err; echo "rehearsal rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("/root/repo/gpurun_out/r05az/bench_gpus8_rehearsal.json") if l.startswith("{")][-1])
print(round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], [ (s["seed"], s.get("parity_crc_ok")) for s in (d.get("ensemble_seeds") or [])][:16])
o=d["one_planet"]; print(round(o["ms_per_step"],1), round(o["value"],1), o["parity"]); 
for r in o["per_rank"]: print(r)
PY
tail -3 $OUT/bench_gpus8_rehearsal.err | cut -c1-200
