#!/bin/bash
# Round 6, session f: compile-time A/B variants (research/ab/build_variants.sh: radix tile 2 048 / 8 192 pairs, thermal occupancy hint 4 / 8 waves, basin ranges of 128 / 512 slots) against the
# default; rocprofv3 kernel stats + iteration timeline + idle gaps of the timed region; PMC FETCH_SIZE / WRITE_SIZE per kernel of the round-6 tree.
cd /root/repo; OUT=/root/repo/gpurun_out/r06f; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for rep in 1 2; do
  for v in default rs8 rs32 tw8 tw4 br512 br128; do
    if [ $v = default ]; then unset WO_LIBWOROGEN; else export WO_LIBWOROGEN=/root/repo/research/ab/variants/libworogen_$v.so; fi
    python bench.py --timed-only --steps 8 --warmup 2 > $OUT/ab_${v}_$rep.json 2> $OUT/ab_${v}_$rep.err
  done
done
unset WO_LIBWOROGEN
python - <<'PY'
import json,glob
for v in ("default","rs8","rs32","tw8","tw4","br512","br128"):
    rows=[]
    for f in sorted(glob.glob(f"/root/repo/gpurun_out/r06f/ab_{v}_*.json")):
        try:
            d=json.loads(open(f).read().strip().splitlines()[-1]); rows.append((round(d["ms_per_step"],1), d["parity"]["parity_crc_ok"], {k:round(x,1) for k,x in d["stage_ms_last_step"].items() if k in ("sort","solve","thermal")}))
        except Exception as e:
            rows.append(("failed", str(e)[:80]))
    print(v, rows)
PY
for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $C --output-format csv -d "$OUT/$C" -o pmc -- python bench.py --timed-only --steps 1 --warmup 0 --iters 200 > "$OUT/$C.log" 2>&1
    echo "$C rc=$?"
done
python profiles/summarize_pmc.py "$OUT" 200 > "$OUT/pmc_summary.json"; rm -rf "$OUT/FETCH_SIZE" "$OUT/WRITE_SIZE"
cd /tmp; rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o t -- python /root/repo/bench.py --timed-only --steps 5 --warmup 1 > $OUT/bench_under_rocprof_timed_only.json 2> $OUT/kt.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/rocprofv3_kernel_stats_timed_region_only.csv
python /root/repo/profiles/iteration_timeline.py /tmp/kt 700 > $OUT/iteration_timeline.txt 2>&1
python /root/repo/profiles/step_idle_gaps.py /tmp/kt 100 > $OUT/step_idle_gaps.txt 2>&1
head -12 $OUT/rocprofv3_kernel_stats_timed_region_only.csv; cat $OUT/step_idle_gaps.txt
