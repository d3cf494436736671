#!/bin/bash
# Round 5, session ad: PROBE — the spread of the largest landmass's walk (31-44 ms on the same input).  Eroded 10 M-cell state from the GPU, then the host flood alone,
# 24 calls per configuration: default, no pinning, one flood thread, the whole process inside one CCD, inside the other socket's node.
cd /root/repo; OUT=/root/repo/gpurun_out/r05ad; mkdir -p $OUT
export TMPDIR=/tmp
python research/flood/walk_spread_probe.py make > $OUT/make.txt 2>&1; tail -1 $OUT/make.txt
run() { tag=$1; shift; env "$@" WO_FLOOD_TIMING=1 python research/flood/walk_spread_probe.py run $tag 24 > $OUT/$tag.out 2> $OUT/$tag.err; }
run default X=1
run nopin WO_FLOOD_PIN=0
run one_thread WO_FLOOD_THREADS=1
run eight_threads WO_FLOOD_THREADS=8
WO_FLOOD_TIMING=1 taskset -c 0-7 python research/flood/walk_spread_probe.py run one_ccd 24 > $OUT/one_ccd.out 2> $OUT/one_ccd.err
WO_FLOOD_TIMING=1 taskset -c 0-63 python research/flood/walk_spread_probe.py run node0 24 > $OUT/node0.out 2> $OUT/node0.err
python - <<'PY'
import re
for tag in ("default","nopin","one_thread","eight_threads","one_ccd","node0"):
    t=open(f"/root/repo/gpurun_out/r05ad/{tag}.err").read()
    w=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
    j=[float(x) for x in re.findall(r"round joined at ([\d.]+) ms", t)]
    print(tag, open(f"/root/repo/gpurun_out/r05ad/{tag}.out").read().strip())
    if w: print("   walks: min %.1f median %.1f max %.1f  " % (min(w), sorted(w)[len(w)//2], max(w)), [round(x,1) for x in w])
    if j: print("   round joined: min %.1f median %.1f max %.1f" % (min(j), sorted(j)[len(j)//2], max(j)))
PY
