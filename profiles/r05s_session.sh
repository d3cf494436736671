#!/bin/bash
# Round 5, session s: the whole relaxed mode (WO_RELAXED=full) measured: time per step, stages, distance from the exact field
cd /root/repo; OUT=/root/repo/gpurun_out/r05s; mkdir -p $OUT
export TMPDIR=/tmp
python bench.py --no-cpu --no-transfers --no-profile --in-flight 0 --steps 2 --warmup 1 > $OUT/bench_with_relaxed.json 2> $OUT/bench_with_relaxed.err; echo "rc=$?"; tail -3 $OUT/bench_with_relaxed.err
cd /tmp; rm -rf /tmp/kt; WO_RELAXED=full timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 1 --warmup 1 > $OUT/bench_relaxed_under_rocprof.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_relaxed.csv
python /root/repo/profiles/iteration_timeline.py /tmp/kt 150 k_receivers_flow_init > $OUT/iteration_timeline_relaxed.txt 2>&1
cd /root/repo
python - <<'PY'
import json
d=json.loads(open("/root/repo/gpurun_out/r05s/bench_with_relaxed.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"])
for r in d["relaxed_mode"]["runs"]: print(r)
print(d["relaxed_mode"]["full"])
PY
