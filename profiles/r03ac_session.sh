#!/bin/bash
# land-first mirror: bench A/B, then the GPU suite
cd /root/repo; mkdir -p gpurun_out/r03ac
export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03ac/bench_landfirst.json 2> gpurun_out/r03ac/bench_landfirst.err
WO_MIRROR_LAND_FIRST=0 timeout 600 python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1 > gpurun_out/r03ac/bench_morton.json 2> gpurun_out/r03ac/bench_morton.err
tail -c 600 gpurun_out/r03ac/bench_landfirst.json; echo; tail -c 300 gpurun_out/r03ac/bench_morton.json
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r03ac/gpu_tests.log 2>&1; tail -3 gpurun_out/r03ac/gpu_tests.log
