#!/bin/bash
# Round 6, session t: the N > 1 ensemble leg at full size with planets in flight, as a rehearsal (two gloo ranks sharing the one GPU; not a measurement): config 5's planets are NEW terrains inside the timed region.
cd /root/repo; OUT=/root/repo/gpurun_out/r06u; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --share-gpu --steps 6 --warmup 1 --seeds-per-rank 6 --planets-in-flight 3 --one-planet-cells 0 --no-cpu --no-profile --in-flight 0 > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2_rehearsal.err; echo "rc=$?"
tail -3 $OUT/bench_gpus2_rehearsal.err
python - <<'PY'
import json
d=json.loads([l for l in open("/root/repo/gpurun_out/r06u/bench_gpus2_rehearsal.json").read().splitlines() if l.startswith("{")][-1])
print(d["n_gpus"], d["steps"], round(d["ms_per_step"],1), round(d["value"],1), d["scaling"], d["ensemble_steps_are_new_terrain"], d["cold_first_step_ms"])
print([(e["seed"], e.get("parity_crc_ok"), e["steps"]) for e in d["ensemble_seeds"]])
PY
