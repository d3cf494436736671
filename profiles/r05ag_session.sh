#!/bin/bash
# Round 5, session ag: PROBE — which thread walks the largest landmass, on which CPU, with which queue store, and how long it takes (flood alone, node 0, 48 calls).
cd /root/repo; OUT=/root/repo/gpurun_out/r05ag; mkdir -p $OUT
export TMPDIR=/tmp
python research/flood/walk_spread_probe.py make > $OUT/make.txt 2>&1; tail -1 $OUT/make.txt
WO_FLOOD_TIMING=1 taskset -c 0-63,128-191 python research/flood/walk_spread_probe.py run who 48 > $OUT/who.out 2> $OUT/who.err
grep -E "walker:|walk of the largest" $OUT/who.err | cut -c1-160 | paste - - | awk '{print}' | cut -c1-260
