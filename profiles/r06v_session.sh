#!/bin/bash
# Round 6, session v: after bench.py's last change (planets in flight for N > 1, default 1): the driver's bench command once more, the two-rank bench rehearsal test, smoke.
cd /root/repo; OUT=/root/repo/gpurun_out/r06v; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r06v/bench_driver_command.json").read().strip().splitlines()[-1])
print(d["metric"], d["unit"], d["n_gpus"], d["steps"], d["warmup"], round(d["ms_per_step"],2), round(d["value"],1), d["parity"]["parity_crc_ok"], d["dtype"], d["scaling"], d["vs_baseline"])
print(sorted(d.keys()))
PY
timeout 900 python -m pytest tests/test_bench_dist.py -q -m gpu 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-80
