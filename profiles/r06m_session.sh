#!/bin/bash
# Round 6, session m: differential run of the final (pruned) tree against the oracle, every cell, on seeds no earlier run used: 12 seeds at 0.3-2 M cells, 8 seeds at 3-8 M cells.
cd /root/repo; OUT=/root/repo/gpurun_out/r06m; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python profiles/differential_seeds.py 120 12 > $OUT/differential_seeds_12_cases.txt 2>&1; tail -4 $OUT/differential_seeds_12_cases.txt
timeout 3000 python profiles/differential_seeds.py 140 8 big > $OUT/differential_seeds_big_8_cases.txt 2>&1; tail -4 $OUT/differential_seeds_big_8_cases.txt
