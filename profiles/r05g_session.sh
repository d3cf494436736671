#!/bin/bash
# Round 5, session g: full -m gpu suite (new: 40 M x 200 CRC, 40 M as two gloo processes) on the two-level flow accumulation; bench + timeline
cd /root/repo; OUT=/root/repo/gpurun_out/r05g; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
tail -22 $OUT/pytest_gpu.log
python bench.py --timed-only --steps 5 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp; rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 2 --warmup 1 > /dev/null 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for n in 100 300 500; do python /root/repo/profiles/iteration_timeline.py /tmp/kt $n >> $OUT/iteration_timeline.txt 2>&1; done
cd /root/repo
python - <<'PY'
import json
for f in ("bench_default",):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r05g/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"], d["erode_stats"].get("calls_run_again_with_checks"))
    except Exception as ex: print(f, "ERR", ex)
PY
