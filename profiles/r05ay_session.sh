#!/bin/bash
# Round 5, session ay: what the walk pays for the pop log of the replay's prefix rule (one 4-byte store per pop): flood alone, 72 calls each, WO_FLOOD_PREFIX=1 / 0.
cd /root/repo; OUT=/root/repo/gpurun_out/r05ay; mkdir -p $OUT
export TMPDIR=/tmp
python research/flood/walk_spread_probe.py make > $OUT/make.txt 2>&1; tail -1 $OUT/make.txt
for rep in 1 2 3; do
WO_FLOOD_PREFIX=1 WO_FLOOD_TIMING=1 taskset -c 0-63,128-191 python research/flood/walk_spread_probe.py run log 24 >> $OUT/log.out 2>> $OUT/log.err
WO_FLOOD_PREFIX=0 WO_FLOOD_TIMING=1 taskset -c 0-63,128-191 python research/flood/walk_spread_probe.py run nolog 24 >> $OUT/nolog.out 2>> $OUT/nolog.err
done
python - <<'PY'
import re
for tag in ("log","nolog"):
    t=open(f"/root/repo/gpurun_out/r05ay/{tag}.err").read()
    w=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
    j=[float(x) for x in re.findall(r"round joined at ([\d.]+) ms", t)]
    w2=sorted(w); j2=sorted(j)
    print(tag, "n %d walk min %.1f median %.1f mean %.1f | joined median %.1f mean %.1f" % (len(w), w2[0], w2[len(w2)//2], sum(w)/len(w), j2[len(j2)//2], sum(j)/len(j)))
PY
