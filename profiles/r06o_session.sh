#!/bin/bash
# Round 6, session o: the final tree with the seven new config-5 tests: the driver's GPU command, smoke.
cd /root/repo; OUT=/root/repo/gpurun_out/r06o; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu.log 2> $OUT/pytest_gpu.err; echo "pytest rc=$?" >> $OUT/pytest_gpu.log; tail -8 $OUT/pytest_gpu.log | grep -E "passed|failed|rc="
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-100
