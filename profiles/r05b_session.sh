#!/bin/bash
# Round 5, session b: fixed-stride rows (load_row), pass tags instead of clearing the solve outputs, dead per-cell arrays of the receivers pass gone.
cd /root/repo; OUT=/root/repo/gpurun_out/r05b; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
python bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
WO_ROWS=csr python bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_rows_csr.json 2> $OUT/bench_rows_csr.err
python bench.py --no-cpu --no-relaxed --no-transfers --in-flight 0 --steps 2 --warmup 1 > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
python - <<'PY'
import json
for f in ("bench_default","bench_rows_csr","bench_profiled"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r05b/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
        if d.get("roofline"):
            print("  roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
            print("  families", {k:(v["ms"], v["launches"]) for k,v in d["roofline"]["families"].items() if v["ms"]>2})
    except Exception as ex: print(f, "ERR", ex)
PY
