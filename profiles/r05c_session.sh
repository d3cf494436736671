#!/bin/bash
# Round 5, session c: flow accumulation in two levels (k_flow_tiles) against the one-launch climb
cd /root/repo; OUT=/root/repo/gpurun_out/r05c; mkdir -p $OUT
export TMPDIR=/tmp WO_BENCH_ALLOW_STALE_PMC=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flow_accumulation or golden or against_oracle_large or mirror_layout or config3 or graph_replay or basin_leftovers or edge_cases or ties_on_larger" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_subset.log
tail -4 $OUT/pytest_subset.log
python bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_tiles.json 2> $OUT/bench_tiles.err
WO_FLOW=climb python bench.py --timed-only --steps 4 --warmup 2 > $OUT/bench_climb.json 2> $OUT/bench_climb.err
python bench.py --no-cpu --no-relaxed --no-transfers --in-flight 0 --steps 2 --warmup 1 > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 2 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
for n in 100 300 500; do python /root/repo/profiles/iteration_timeline.py /tmp/kt $n >> $OUT/iteration_timeline.txt 2>&1 || true; done
cd /root/repo
python - <<'PY'
import json
for f in ("bench_tiles","bench_climb","bench_profiled"):
    try:
        d=json.loads(open(f"/root/repo/gpurun_out/r05c/{f}.json").read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
        if d.get("roofline"):
            print("  families", {k:(v["ms"], v["launches"]) for k,v in d["roofline"]["families"].items() if v["ms"]>2})
    except Exception as ex: print(f, "ERR", ex)
PY
