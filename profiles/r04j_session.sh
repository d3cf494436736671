#!/bin/bash
# Round 4, session j: hipGraph replay of the steady composite iteration, A/B
cd /root/repo; OUT=/root/repo/gpurun_out/r04j; mkdir -p $OUT
B="python bench.py --no-cpu --no-relaxed --in-flight 0 --steps 3 --warmup 1"
WO_GRAPH=0 $B > $OUT/bench_nograph.json 2> $OUT/bench_nograph.err
$B > $OUT/bench_graph.json 2> $OUT/bench_graph.err
tail -3 $OUT/bench_graph.err
for f in nograph graph; do python - $OUT/bench_$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], {k:round(v,1) for k,v in d["stage_ms_last_step"].items()}, d["erode_stats"].get("iterations_replayed_from_graph"))
PY
done
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -8
