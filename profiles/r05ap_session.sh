#!/bin/bash
# Round 5, session ap: the flood stage moves the land heights straight into / out of the flood's own (page-locked) array: flood-related GPU tests, bench A/B against WO_FLOOD_STAGING=copy.
cd /root/repo; OUT=/root/repo/gpurun_out/r05ap; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -x -q -m gpu  > $OUT/pytest_gpu_flood.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_flood.log
tail -4 $OUT/pytest_gpu_flood.log
for rep in 1; do
python bench.py --timed-only --steps 12 --warmup 2 > $OUT/bench_direct_$rep.json 2> /dev/null
WO_FLOOD_STAGING=copy python bench.py --timed-only --steps 12 --warmup 2 > $OUT/bench_copy_$rep.json 2> /dev/null
done
WO_FLOOD_TIMING=1 python bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_laps.json 2> $OUT/flood_laps.txt
python - <<'PY'
import json
for tag in ("direct","copy"):
    v=[]
    for rep in (1,):
        d=json.loads(open(f"/root/repo/gpurun_out/r05ap/bench_{tag}_{rep}.json").read().strip().splitlines()[-1]); v.append((round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"]))
    print(tag, v, "mean", round(sum(a for a,b in v)/1,1))
PY
grep -E "flood stage\]|writeback|pipeline  " $OUT/flood_laps.txt | tail -12
