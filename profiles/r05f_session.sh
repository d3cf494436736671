#!/bin/bash
# Round 5, session f: read-only chase for the layout's start state (no ring splits); + the events-stream route beside the tile kernels
cd /root/repo; OUT=/root/repo/gpurun_out/r05f; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flow_accumulation or golden or config3 or mirror_layout or solve_kernel or basin_leftovers or ties_on_larger or edge_cases or against_oracle_large or graph_replay or land_count" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -3 $OUT/pytest_subset.log
python bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
WO_FLOW_EVENTS_STREAM=1 python bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_events_stream.json 2> $OUT/bench_events_stream.err
python bench.py --no-cpu --no-relaxed --no-transfers --in-flight 0 --steps 2 --warmup 1 > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
cd /tmp; rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 2 --warmup 1 > /dev/null 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for n in 100 300 500; do python /root/repo/profiles/iteration_timeline.py /tmp/kt $n >> $OUT/iteration_timeline.txt 2>&1; done
rm -rf /tmp/kt2; WO_FLOW_EVENTS_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o t -- python /root/repo/bench.py --timed-only --steps 1 --warmup 1 > /dev/null 2> $OUT/kt2.err
python /root/repo/profiles/iteration_timeline.py /tmp/kt2 150 > $OUT/iteration_timeline_events_stream.txt 2>&1
cd /root/repo
python - <<'PY'
import json
for f in ("bench_default","bench_events_stream","bench_profiled"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r05f/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"], d["erode_stats"].get("calls_run_again_with_checks"))
        if d.get("roofline"): print("  families", {k:(v["ms"], v["launches"]) for k,v in d["roofline"]["families"].items() if v["ms"]>2})
    except Exception as ex: print(f, "ERR", ex)
PY
