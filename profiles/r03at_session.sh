#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03at
export TMPDIR=/tmp
WO_FLOOD_TIMING=1 timeout 300 python bench.py --no-cpu --no-profile --in-flight 0 --steps 2 --warmup 1 --iters 8 > gpurun_out/r03at/b.json 2> gpurun_out/r03at/flood.txt
grep -E "\[flood" gpurun_out/r03at/flood.txt | tail -24
