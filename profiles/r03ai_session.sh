#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r03ai
export TMPDIR=/tmp
WO_FLOOD_TIMING=1 timeout 600 python bench.py --no-cpu --no-profile --in-flight 0 --steps 2 --warmup 1 > gpurun_out/r03ai/bench.json 2> gpurun_out/r03ai/flood_timing.txt
grep -c . gpurun_out/r03ai/flood_timing.txt; tail -40 gpurun_out/r03ai/flood_timing.txt
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket|L2|L3|MHz" 
