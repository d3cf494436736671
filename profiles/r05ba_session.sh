#!/bin/bash
# Round 5, session ba: PMC FETCH_SIZE / WRITE_SIZE per kernel re-collected on the round's final tree (ONE timed step, 200 iterations), the iteration timeline of the
# final tree, and the default bench line quoting the new counter file.
cd /root/repo; OUT=/root/repo/gpurun_out/r05ba; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o pmc -- python bench.py --timed-only --steps 1 --warmup 0 --iters 200 > "$OUT/$C.log" 2>&1
    echo "$C rc=$?"
done
python profiles/summarize_pmc.py "$OUT" 200 > "$OUT/pmc_summary.json"
rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
cd /tmp; rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 5 --warmup 1 > $OUT/bench_under_trace.json 2> $OUT/kt.err
python /root/repo/profiles/iteration_timeline.py /tmp/kt 700 > $OUT/iteration_timeline.txt 2>&1
python /root/repo/profiles/step_idle_gaps.py /tmp/kt 100 > $OUT/step_idle_gaps.txt 2>&1
cd /root/repo
cp $OUT/pmc_summary.json profiles/r05_pmc_fetch_write_per_kernel_10m_200iters.json
python bench.py > $OUT/bench_default_no_flags.json 2> $OUT/bench_default.err; echo "bench rc=$?"; tail -2 $OUT/bench_default.err
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r05ba/bench_default_no_flags.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["steps"], d["warmup"], round(d["ms_per_step"],1), round(d["value"],1), d["parity"]["parity_crc_ok"], d["stage_ms_last_step"])
print({k:r[k] for k in ('kernel','achieved','frac','launches','avg_launch_us','traffic')})
PY
head -30 $OUT/iteration_timeline.txt
