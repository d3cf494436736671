#!/bin/bash
# Round 5, session ae: every landmass's worker writes the start state of its own cells just before it walks them (flood_landmass_pipeline: init_runs).
# The flood alone (probe of session ad, same eroded state), then the bench, then the flood tests of the GPU suite.
cd /root/repo; OUT=/root/repo/gpurun_out/r05ae; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python research/flood/walk_spread_probe.py make > $OUT/make.txt 2>&1; tail -1 $OUT/make.txt
WO_FLOOD_TIMING=1 taskset -c 0-63,128-191 python research/flood/walk_spread_probe.py run node0_owner_init 24 > $OUT/probe.out 2> $OUT/probe.err
for rep in 1 2 3; do
python bench.py --timed-only --steps 8 --warmup 2 > $OUT/bench_$rep.json 2> /dev/null
done
WO_FLOOD_TIMING=1 python bench.py --timed-only --steps 6 --warmup 2 > $OUT/bench_laps.json 2> $OUT/flood_laps.txt
timeout 1500 python -m pytest tests -x -q -m gpu -k "flood or config4 or headline or exchange or golden or decomposed" > $OUT/pytest_gpu_flood.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu_flood.log
tail -4 $OUT/pytest_gpu_flood.log
python - <<'PY'
import json,re
t=open("/root/repo/gpurun_out/r05ae/probe.err").read()
w=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
j=[float(x) for x in re.findall(r"round joined at ([\d.]+) ms", t)]
print(open("/root/repo/gpurun_out/r05ae/probe.out").read().strip())
print("   walks: min %.1f median %.1f max %.1f  " % (min(w), sorted(w)[len(w)//2], max(w)), [round(x,1) for x in w])
print("   round joined: min %.1f median %.1f max %.1f" % (min(j), sorted(j)[len(j)//2], max(j)))
for rep in (1,2,3):
    d=json.loads(open(f"/root/repo/gpurun_out/r05ae/bench_{rep}.json").read().strip().splitlines()[-1])
    print(rep, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
t=open("/root/repo/gpurun_out/r05ae/flood_laps.txt").read()
w=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
h=[float(x) for x in re.findall(r"host passes\s+([\d.]+) ms", t)]
print("bench walks", w[-12:]); print("bench host passes", h[-12:])
PY
