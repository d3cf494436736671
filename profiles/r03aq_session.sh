#!/bin/bash
# flood walk variants on the GPU box's host (EPYC 9575F): default, no software prefetch, binary heap
cd /root/repo; mkdir -p gpurun_out/r03aq
export TMPDIR=/tmp
for V in default noprefetch binary binary_noprefetch; do
  unset WO_FLOOD_NOPREFETCH WO_FLOOD_HEAP
  case $V in noprefetch) export WO_FLOOD_NOPREFETCH=1;; binary) export WO_FLOOD_HEAP=2;; binary_noprefetch) export WO_FLOOD_HEAP=2 WO_FLOOD_NOPREFETCH=1;; esac
  WO_FLOOD_TIMING=1 timeout 300 python bench.py --no-cpu --no-profile --in-flight 0 --steps 2 --warmup 1 --iters 8 > gpurun_out/r03aq/$V.json 2> gpurun_out/r03aq/$V.txt
  echo "== $V"; grep -E "walk of the largest|pipeline" gpurun_out/r03aq/$V.txt | tail -8
done
