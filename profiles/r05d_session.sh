#!/bin/bash
# Round 5, session d: k_flow_tiles with asynchronous local-root jumping and a one-round-trip root climb; tile sizes
cd /root/repo; OUT=/root/repo/gpurun_out/r05d; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flow_accumulation or golden or config3" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -3 $OUT/pytest_subset.log
run() { # name
  python bench.py --timed-only --steps 3 --warmup 2 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  (cd /tmp; rm -rf /tmp/kt_$1; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$1 -o t -- python /root/repo/bench.py --timed-only --steps 1 --warmup 1 > /dev/null 2> $OUT/kt_$1.err
   grep -i "k_flow\|basin_jump" $(find /tmp/kt_$1 -name "*kernel_stats.csv" | head -1) > $OUT/flow_kernels_$1.csv
   python /root/repo/profiles/iteration_timeline.py /tmp/kt_$1 150 > $OUT/iteration_timeline_$1.txt 2>&1)
}
run t1024
for cfg in "2048 256" "2048 512" "512 128" "4096 512"; do
  set -- $cfg
  make -s -C planet_heightmap_generation_amd/csrc build/planet.hip.o EXTRA="-DWO_FT_CELLS=$1 -DWO_FT_THREADS=$2" > $OUT/make_$1_$2.log 2>&1 && touch planet_heightmap_generation_amd/csrc/build/planet.hip.o
  rm -f planet_heightmap_generation_amd/csrc/build/planet.hip.o
  make -s -C planet_heightmap_generation_amd/csrc EXTRA="-DWO_FT_CELLS=$1 -DWO_FT_THREADS=$2" >> $OUT/make_$1_$2.log 2>&1
  run t$1_$2
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/root/repo/gpurun_out/r05d/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"]["flow"], d["stage_ms_last_step"]["receivers"])
    except Exception as ex: print(f, "ERR", ex)
PY
cat $OUT/flow_kernels_*.csv | cut -c1-140
