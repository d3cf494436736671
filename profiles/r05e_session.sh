#!/bin/bash
# Round 5, session e: k_flow_tiles<false> also shortens the basin layout's start state; the layout starts after it
cd /root/repo; OUT=/root/repo/gpurun_out/r05e; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flow_accumulation or golden or config3 or mirror_layout or solve_kernel or basin_leftovers or ties_on_larger or edge_cases or against_oracle_large or graph_replay or land_count" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -3 $OUT/pytest_subset.log
python bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --no-cpu --no-relaxed --no-transfers --in-flight 0 --steps 2 --warmup 1 > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
cd /tmp; rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 2 --warmup 1 > /dev/null 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for n in 100 300 500; do python /root/repo/profiles/iteration_timeline.py /tmp/kt $n >> $OUT/iteration_timeline.txt 2>&1; done
cd /root/repo
python - <<'PY'
import json
for f in ("bench_default","bench_profiled"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r05e/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"], d["erode_stats"].get("calls_run_again_with_checks"))
        if d.get("roofline"): print("  families", {k:(v["ms"], v["launches"]) for k,v in d["roofline"]["families"].items() if v["ms"]>2})
    except Exception as ex: print(f, "ERR", ex)
PY
