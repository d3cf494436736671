#!/bin/bash
# Round 4, session e: depth of the slowest range's DAG and clocks per level (k_solve_flowing<4, true>); prepared divisions A/B
cd /root/repo; OUT=/root/repo/gpurun_out/r04e; mkdir -p $OUT
WO_BASIN_STATS=1 python bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters 40 > $OUT/bench_stats.json 2> $OUT/basin_stats.txt
grep "basin stats" $OUT/basin_stats.txt | head -42
B="python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1"
WO_LIBWOROGEN=/root/repo/planet_heightmap_generation_amd/libworogen_pd.so $B > $OUT/bench_prepared_div.json 2> $OUT/bench_prepared_div.err
$B > $OUT/bench_default.json 2> $OUT/bench_default.err
for f in prepared_div default; do python - $OUT/bench_$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fam=d["roofline"]["families"]
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], {k:round(v,1) for k,v in d["stage_ms_last_step"].items()}, {k:(fam[k]["ms"],fam[k]["launches"]) for k in ("solve_basin","solve_setup") if k in fam})
PY
done
