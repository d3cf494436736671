#!/bin/bash
# Round 5, session ac: the largest walk on a CPU whose L3 no other flood worker uses (WO_FLOOD_PIN, default on) against WO_FLOOD_PIN=0: 3 x 6 timed steps each.
cd /root/repo; OUT=/root/repo/gpurun_out/r05ac; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3; do
WO_FLOOD_TIMING=1 python bench.py --timed-only --steps 6 --warmup 2 > $OUT/bench_pin_$rep.json 2> $OUT/flood_laps_pin_$rep.txt
WO_FLOOD_PIN=0 WO_FLOOD_TIMING=1 python bench.py --timed-only --steps 6 --warmup 2 > $OUT/bench_nopin_$rep.json 2> $OUT/flood_laps_nopin_$rep.txt
done
for rep in 1 2; do
python bench.py --timed-only --steps 8 --warmup 2 > $OUT/bench_quiet_pin_$rep.json 2> /dev/null
WO_FLOOD_PIN=0 python bench.py --timed-only --steps 8 --warmup 2 > $OUT/bench_quiet_nopin_$rep.json 2> /dev/null
done
python - <<'PY'
import json,re
for tag in ("pin","nopin"):
  for rep in (1,2,3):
    d=json.loads(open(f"/root/repo/gpurun_out/r05ac/bench_{tag}_{rep}.json").read().strip().splitlines()[-1])
    t=open(f"/root/repo/gpurun_out/r05ac/flood_laps_{tag}_{rep}.txt").read()
    walks=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
    host=[float(x) for x in re.findall(r"host passes\s+([\d.]+) ms", t)]
    print(tag, rep, round(d["ms_per_step"],1), d["stage_ms_last_step"]["priority_flood"], d["parity"]["parity_crc_ok"], "walks", walks[-12:], "host", [round(h) for h in host[-12:]])
for tag in ("quiet_pin","quiet_nopin"):
  for rep in (1,2):
    d=json.loads(open(f"/root/repo/gpurun_out/r05ac/bench_{tag}_{rep}.json").read().strip().splitlines()[-1])
    print(tag, rep, round(d["ms_per_step"],1), d["stage_ms_last_step"]["priority_flood"], d["parity"]["parity_crc_ok"])
PY
grep -m2 "largest walk stays" $OUT/flood_laps_pin_1.txt
