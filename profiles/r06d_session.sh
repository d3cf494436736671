#!/bin/bash
# Round 6, session d: the elevation sort on the side stream beside the receivers pass + flow accumulation (default) against WO_SORT=serial: parity subset, A/B of the timed region, iteration timeline.
cd /root/repo; OUT=/root/repo/gpurun_out/r06d; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or sort_routes or ties or edge_cases or against_oracle_large or mirror_layout or flow_accumulation or land_count or config3_checksum or planets_in_flight" > $OUT/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.log
for i in 1 2 3; do
  python bench.py --timed-only --steps 8 --warmup 2 > $OUT/beside_$i.json 2>/dev/null
  WO_SORT=serial python bench.py --timed-only --steps 8 --warmup 2 > $OUT/serial_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for k in ("beside","serial"):
    v=[json.loads(open(f).read().strip().splitlines()[-1]) for f in sorted(glob.glob(f"/root/repo/gpurun_out/r06d/{k}_*.json"))]
    print(k, [(round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"]) for d in v], v[0]["stage_ms_last_step"])
PY
cd /tmp; rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 4 --warmup 1 > $OUT/bench_under_trace.json 2> $OUT/kt.err
python /root/repo/profiles/iteration_timeline.py /tmp/kt 500 > $OUT/iteration_timeline.txt 2>&1
python /root/repo/profiles/step_idle_gaps.py /tmp/kt 100 > $OUT/step_idle_gaps.txt 2>&1
cat $OUT/iteration_timeline.txt; cat $OUT/step_idle_gaps.txt
