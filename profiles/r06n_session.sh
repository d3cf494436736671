#!/bin/bash
# Round 6, session n: all 64 planets of BASELINE config 5 at the full 200 iterations on the HIP path, each against the oracle's CRC (seeds 16-64 pinned this round: 12 core-hours of the oracle),
# then the new GPU test.
cd /root/repo; OUT=/root/repo/gpurun_out/r06n; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python profiles/config5_all_seeds.py 1 64 > $OUT/config5_all_64_seeds.txt 2> $OUT/config5_all_64_seeds.err; echo "all seeds rc=$?"; tail -4 $OUT/config5_all_64_seeds.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "every_eighth" > $OUT/pytest_new.log 2>&1; tail -2 $OUT/pytest_new.log
