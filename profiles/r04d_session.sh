#!/bin/bash
# Round 4, session d: barrier-free solve kernel (k_solve_flowing) against k_solve_coop
cd /root/repo; OUT=/root/repo/gpurun_out/r04d; mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -5 > $OUT/gputests.txt
B="python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1"
WO_BASIN_KERNEL=barrier $B > $OUT/bench_barrier.json 2> $OUT/bench_barrier.err
$B > $OUT/bench_flowing.json 2> $OUT/bench_flowing.err
cat $OUT/gputests.txt
for f in barrier flowing; do python - $OUT/bench_$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fam=d["roofline"]["families"]
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], {k:round(v,1) for k,v in d["stage_ms_last_step"].items()}, {k:(fam[k]["ms"],fam[k]["launches"]) for k in ("solve_basin","solve_setup","thermal_apply","thermal_excess") if k in fam}, d["erode_stats"]["solve_basin_passes_with_leftovers"], d["erode_stats"]["calls_run_again_with_checks"])
PY
done
