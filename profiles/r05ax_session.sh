#!/bin/bash
# Round 5, session ax: the walk counts the cells of every drainage tree as it claims them (one pass less for the tree lists): does the walk pay for it?  Flood alone, 48 calls each.
cd /root/repo; OUT=/root/repo/gpurun_out/r05ax; mkdir -p $OUT
export TMPDIR=/tmp
python research/flood/walk_spread_probe.py make > $OUT/make.txt 2>&1; tail -1 $OUT/make.txt
for rep in 1 2; do
WO_FLOOD_TIMING=1 taskset -c 0-63,128-191 python research/flood/walk_spread_probe.py run counting 24 >> $OUT/counting.out 2>> $OUT/counting.err
WO_EMU_LIB=tests/emu/_build/libemu_before.so WO_FLOOD_TIMING=1 taskset -c 0-63,128-191 python research/flood/walk_spread_probe.py run before 24 >> $OUT/before.out 2>> $OUT/before.err
done
python - <<'PY'
import re
for tag in ("counting","before"):
    t=open(f"/root/repo/gpurun_out/r05ax/{tag}.err").read()
    w=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
    l=[float(x) for x in re.findall(r"tree lists done at ([\d.]+) ms", t)]
    j=[float(x) for x in re.findall(r"round joined at ([\d.]+) ms", t)]
    n=min(len(w),len(l),len(j)); d=sorted(a-b for a,b in zip(l[:n],w[:n])); w2=sorted(w); l2=sorted(l); j2=sorted(j)
    print(tag, "walk median %.1f mean %.1f | lists done median %.1f | lists - walk median %.2f | joined median %.1f mean %.1f" % (w2[len(w2)//2], sum(w)/len(w), l2[len(l2)//2], d[len(d)//2], j2[len(j2)//2], sum(j)/len(j)))
PY
