#!/bin/bash
# Round 4, session f: straight-line polling loop of k_solve_flowing (solve_apply_flat)
cd /root/repo; OUT=/root/repo/gpurun_out/r04f; mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -5 > $OUT/gputests.txt
B="python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1"
$B > $OUT/bench_flat.json 2> $OUT/bench_flat.err
WO_BASIN_STATS=1 python bench.py --no-cpu --no-profile --in-flight 0 --steps 1 --warmup 0 --iters 12 > $OUT/bench_stats.json 2> $OUT/basin_stats.txt
cat $OUT/gputests.txt
python - $OUT/bench_flat.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fam=d["roofline"]["families"]
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], {k:round(v,1) for k,v in d["stage_ms_last_step"].items()}, {k:(fam[k]["ms"],fam[k]["launches"]) for k in ("solve_basin","solve_setup") if k in fam}, d["erode_stats"]["solve_basin_passes_with_leftovers"], d["erode_stats"]["calls_run_again_with_checks"])
PY
grep "basin stats" $OUT/basin_stats.txt | head -14
