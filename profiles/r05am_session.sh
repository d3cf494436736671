#!/bin/bash
# Round 5, session am: differential run of the round's final tree (the replay's prefix rule, the carve pass's short peak scan, one flooding rank, ...) against the
# oracle, every cell: 12 seeds at 0.3-2 M cells and 8 seeds at 3-8 M cells that no earlier run used.
cd /root/repo; OUT=/root/repo/gpurun_out/r05am; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python profiles/differential_seeds.py 70 12 > $OUT/differential_seeds_12_cases.txt 2>&1; tail -3 $OUT/differential_seeds_12_cases.txt
timeout 2700 python profiles/differential_seeds.py 90 8 big > $OUT/differential_seeds_big_8_cases.txt 2>&1; tail -3 $OUT/differential_seeds_big_8_cases.txt
