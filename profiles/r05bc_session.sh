#!/bin/bash
# Round 5, session bc: the fourteen config-5 planets pinned at the full 200 iterations (seeds 2-15) + the round's final full -m gpu suite.
cd /root/repo; OUT=/root/repo/gpurun_out/r05bc; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -x -q -m gpu --durations=5 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -9 $OUT/pytest_gpu.log
