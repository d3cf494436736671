#!/bin/bash
# A/B of alternative builds of libworogen.so on one box: bash research/ab/run_ab.sh <name> ...   (libs: research/ab/libworogen_<name>.so, made by
# build_variant.sh; "both" = the in-tree build).  Prints per build: ms per step, CRC check, the stage times of the last step and the kernel families
# named in FAMS (default: the solve's).  e.g.  FAMS="flow_snap flow_final" bash research/ab/run_ab.sh both climb8 both
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab; cp planet_heightmap_generation_amd/libworogen.so /tmp/libworogen_both.so
FAMS=${FAMS:-solve_setup solve_basin}
for v in "$@"; do
  if [ "$v" = both ]; then cp /tmp/libworogen_both.so planet_heightmap_generation_amd/libworogen.so; else cp research/ab/libworogen_$v.so planet_heightmap_generation_amd/libworogen.so; fi
  timeout 300 python bench.py --no-cpu --no-relaxed --in-flight 0 --steps 3 --warmup 1 > gpurun_out/ab/bench_$v.log 2>&1
  grep "^{" gpurun_out/ab/bench_$v.log | FAMS="$FAMS" python -c "
import json,sys,os
for l in sys.stdin:
    d=json.loads(l); f=d['roofline']['families']; print('$v', round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], d['stage_ms_last_step'], {k: f[k]['ms'] for k in os.environ['FAMS'].split() if k in f})
"
done
cp /tmp/libworogen_both.so planet_heightmap_generation_amd/libworogen.so     # leave the in-tree build in place
