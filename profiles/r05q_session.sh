#!/bin/bash
# Round 5, session q: does the basin solve's walk speed up with fewer workgroups per CU? (LDS padding: 10 -> 5 -> 2 workgroups of 2 waves per CU)
cd /root/repo; OUT=/root/repo/gpurun_out/r05q; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
run() { python bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_$1.json 2> $OUT/bench_$1.err; }
run pad0
for PAD in 16384 49152; do
  rm -f planet_heightmap_generation_amd/csrc/build/basin.hip.o
  make -s -j16 -C planet_heightmap_generation_amd/csrc EXTRA="-DWO_X_SOLVE_LDS_PAD=$PAD" > $OUT/make_$PAD.log 2>&1
  run pad$PAD
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r05q/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
    except Exception as ex: print(f, "ERR", ex)
PY
