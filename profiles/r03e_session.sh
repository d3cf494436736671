#!/bin/bash
# Round-3 session E: per-phase clocks of k_solve_basin at four points of the run
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03e; mkdir -p $O
for n in 2 30 120 190; do WO_BASIN_STATS=$n timeout 600 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 0 --no-profile 2>&1 | grep "basin stats" >> $O/basin_stats.txt; done
for n in 30; do WO_BASIN_KEY_BITS=12 WO_BASIN_STATS=$n timeout 600 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 0 --no-profile 2>&1 | grep "basin stats" >> $O/basin_stats_keybits12.txt; done
cat $O/basin_stats.txt $O/basin_stats_keybits12.txt
