#!/bin/bash
# A/B of alternative builds on one box, flow stage: bash research/ab/run_ab5.sh <name> ...   ("both" = the in-tree build)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out/ab; cp planet_heightmap_generation_amd/libworogen.so /tmp/libworogen_both.so
for v in "$@"; do
  if [ "$v" = both ]; then cp /tmp/libworogen_both.so planet_heightmap_generation_amd/libworogen.so; else cp research/ab/libworogen_$v.so planet_heightmap_generation_amd/libworogen.so; fi
  timeout 300 python bench.py --no-cpu --no-relaxed --in-flight 0 --steps 3 --warmup 1 > gpurun_out/ab/bench_$v.log 2>&1
  grep "^{" gpurun_out/ab/bench_$v.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); f=d['roofline']['families']; print('$v', round(d['ms_per_step'],1), d['parity']['parity_crc_ok'], 'flow stage', d['stage_ms_last_step']['flow'], 'glacial', d['stage_ms_last_step']['glacial'], 'flood', d['stage_ms_last_step']['priority_flood'], {k: f[k]['ms'] for k in ('flow_snap','flow_final','ice_round','carve_round') if k in f})
"
done
cp /tmp/libworogen_both.so planet_heightmap_generation_amd/libworogen.so     # leave the in-tree build in place
