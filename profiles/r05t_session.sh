#!/bin/bash
# Round 5, session t: differential run of the round-5 tree (two-level flow accumulation, layout start state from the tiles) against the oracle, every cell, 12 + 8 more seeds
cd /root/repo; OUT=/root/repo/gpurun_out/r05t; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python profiles/differential_seeds.py 30 12 > $OUT/differential_seeds_12_cases.txt 2>&1; tail -3 $OUT/differential_seeds_12_cases.txt
timeout 2400 python profiles/differential_seeds.py 50 8 big > $OUT/differential_seeds_big_8_cases.txt 2>&1; tail -3 $OUT/differential_seeds_big_8_cases.txt
