#!/bin/bash
# Round 4, session c: GPU suite; long-ranges-first dispatch of the basin solve A/B; phase clocks of the longest range; thermal pre-test
cd /root/repo; OUT=/root/repo/gpurun_out/r04c; mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/gputests.txt
B="python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1"
WO_BASIN_LONG_FIRST=0 $B > $OUT/bench_long_first_off.json 2> $OUT/bench_long_first_off.err
$B > $OUT/bench_long_first_on.json 2> $OUT/bench_long_first_on.err
WO_BASIN_STATS=1 python bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters 40 > $OUT/bench_stats.json 2> $OUT/basin_stats.txt
cat $OUT/gputests.txt
for f in off on; do python - $OUT/bench_long_first_$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fam=d["roofline"]["families"]
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], {k:round(v,1) for k,v in d["stage_ms_last_step"].items()}, {k:(fam[k]["ms"],fam[k]["launches"]) for k in ("solve_basin","thermal_apply","thermal_excess","solve_setup","receivers","flow_snap","flow_final") if k in fam})
PY
done
grep "basin stats" $OUT/basin_stats.txt | head -45
