#!/bin/bash
# Round 5, session ah: PROBE — is the spread of the largest walk made by the other workers' allocator traffic (munmap / trim -> TLB shootdowns on every core of the process)?
cd /root/repo; OUT=/root/repo/gpurun_out/r05ah; mkdir -p $OUT
export TMPDIR=/tmp
python research/flood/walk_spread_probe.py make > $OUT/make.txt 2>&1; tail -1 $OUT/make.txt
for rep in 1 2; do
WO_FLOOD_TIMING=1 taskset -c 0-63,128-191 python research/flood/walk_spread_probe.py run base 24 >> $OUT/base.out 2>> $OUT/base.err
MALLOC_MMAP_MAX_=0 MALLOC_TRIM_THRESHOLD_=100000000000 MALLOC_TOP_PAD_=268435456 WO_FLOOD_TIMING=1 taskset -c 0-63,128-191 python research/flood/walk_spread_probe.py run nommap 24 >> $OUT/nommap.out 2>> $OUT/nommap.err
done
grep -c . /proc/interrupts | head -1
python - <<'PY'
import re
for tag in ("base","nommap"):
    t=open(f"/root/repo/gpurun_out/r05ah/{tag}.err").read()
    w=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
    j=[float(x) for x in re.findall(r"round joined at ([\d.]+) ms", t)]
    print(tag, "walks min %.1f median %.1f mean %.1f max %.1f; round joined median %.1f" % (min(w), sorted(w)[len(w)//2], sum(w)/len(w), max(w), sorted(j)[len(j)//2]), [round(x) for x in w])
PY
