#!/bin/bash
# Round 4, session h: GPU suite and bench after the prune (options struct, dead routes removed, in-tree selection)
cd /root/repo; OUT=/root/repo/gpurun_out/r04h; mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -8 > $OUT/gputests.txt
python bench.py --no-cpu --in-flight 0 --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/gputests.txt; tail -3 $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fam=d["roofline"]["families"]
print(round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], {k:round(v,1) for k,v in d["stage_ms_last_step"].items()}, {k:(fam[k]["ms"],fam[k]["launches"]) for k in fam if fam[k]["ms"]>3})
PY
