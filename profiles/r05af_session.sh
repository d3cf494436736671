#!/bin/bash
# Round 5, session af: PROBE — do the idle flood workers' polls (sleep 25 us, look, sleep) disturb the largest walk?  Poll periods 25 / 200 / 2000 us, flood alone, node 0.
cd /root/repo; OUT=/root/repo/gpurun_out/r05af; mkdir -p $OUT
export TMPDIR=/tmp
python research/flood/walk_spread_probe.py make > $OUT/make.txt 2>&1; tail -1 $OUT/make.txt
for us in 25 200 2000 25 200 2000; do
WO_FLOOD_POLL_US=$us WO_FLOOD_TIMING=1 taskset -c 0-63,128-191 python research/flood/walk_spread_probe.py run poll_$us 24 >> $OUT/poll_$us.out 2>> $OUT/poll_$us.err
done
python - <<'PY'
import re
for us in (25,200,2000):
    t=open(f"/root/repo/gpurun_out/r05af/poll_{us}.err").read()
    w=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
    j=[float(x) for x in re.findall(r"round joined at ([\d.]+) ms", t)]
    print("poll", us, "us: walks min %.1f median %.1f mean %.1f max %.1f; round joined median %.1f" % (min(w), sorted(w)[len(w)//2], sum(w)/len(w), max(w), sorted(j)[len(j)//2]), [round(x) for x in w])
PY
