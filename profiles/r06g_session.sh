#!/bin/bash
# Round 6, session g: the final tree, the driver's GPU command three times in a row in fresh processes (VERDICT r05 item 1's "done"), then the soak test on its own with its output kept.
cd /root/repo; OUT=/root/repo/gpurun_out/r06g; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for i in 1 2 3; do
  timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/pytest_gpu_$i.log 2> $OUT/pytest_gpu_$i.err; echo "run $i rc=$?" | tee -a $OUT/summary.txt
  tail -2 $OUT/pytest_gpu_$i.log | head -1 | tee -a $OUT/summary.txt
done
WO_TEST_CHILD=1 timeout 900 python -m pytest "tests/test_gpu_parity.py::test_soak_hundred_planets_in_one_process" -q -s -m gpu 2>&1 | grep -E "soak:|passed|failed" | tee -a $OUT/summary.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT/summary.txt
