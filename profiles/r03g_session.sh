#!/bin/bash
# Round-3 session G: is k_solve_basin bound by its longest workgroup?  per-workgroup clocks at three points of the run
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03g; mkdir -p $O
for wr in 1024:1024 512:1024; do
  w=${wr%%:*}; r=${wr##*:}
  for n in 1 30 190; do WO_BASIN_WINDOW=$w WO_BASIN_RANGE=$r WO_BASIN_STATS=$n timeout 600 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 0 --no-profile 2>&1 | grep "basin stats" >> $O/basin_stats.txt; done
done
cat $O/basin_stats.txt
