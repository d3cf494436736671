#!/bin/bash
# Round 6, session c: where a new terrain's 200-600 ms of set-up go (probe with the library's laps), and the -m gpu suite again from the test that failed on its own assertion.
cd /root/repo; OUT=/root/repo/gpurun_out/r06c; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python profiles/new_terrain_probe.py > $OUT/new_terrain_probe.txt 2> $OUT/new_terrain_probe.err; echo "probe rc=$?"
cat $OUT/new_terrain_probe.txt; grep -v "^\[flood\] landmass\|tree of" $OUT/new_terrain_probe.err | head -150
timeout 3300 python -m pytest tests -x -q -m gpu --durations=8 > $OUT/pytest_gpu.log 2> $OUT/pytest_gpu.err; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -14 $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.err
