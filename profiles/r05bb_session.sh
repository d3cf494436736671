#!/bin/bash
# Round 5, session bb: EXPERIMENT — the basin layout's slotOf[] written by k_basin_slots (after the sort) instead of by the sort's last scatter pass (WO_BASIN_SLOTS=pass): 3 x 12 steps each + timeline.
cd /root/repo; OUT=/root/repo/gpurun_out/r05bb; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2 3; do
WO_BASIN_SLOTS=pass python bench.py --timed-only --steps 12 --warmup 2 > $OUT/bench_pass_$rep.json 2> /dev/null
python bench.py --timed-only --steps 12 --warmup 2 > $OUT/bench_scatter_$rep.json 2> /dev/null
done
cd /tmp; rm -rf /tmp/kt; WO_BASIN_SLOTS=pass timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 3 --warmup 1 > $OUT/bench_under_trace.json 2> $OUT/kt.err
python /root/repo/profiles/iteration_timeline.py /tmp/kt 500 > $OUT/iteration_timeline_slots_pass.txt 2>&1
cd /root/repo
python - <<'PY'
import json
for tag in ("pass","scatter"):
    v=[]
    for rep in (1,2,3):
        d=json.loads(open(f"/root/repo/gpurun_out/r05bb/bench_{tag}_{rep}.json").read().strip().splitlines()[-1]); v.append((round(d["ms_per_step"],1), d["stage_ms_last_step"]["flow"], d["stage_ms_last_step"]["solve"], d["parity"]["parity_crc_ok"]))
    print(tag, v, "mean", round(sum(a for a,b,c,e in v)/3,1))
PY
sed -n 12,30p $OUT/iteration_timeline_slots_pass.txt
