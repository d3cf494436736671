#!/bin/bash
# Round 5, session ai: PROBE — fewer flood workers beside the largest walk (the other landmasses' tree passes are memory-hungry and share its memory system): 8 / 12 / 16 / 24.
cd /root/repo; OUT=/root/repo/gpurun_out/r05ai; mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python research/flood/walk_spread_probe.py make > $OUT/make.txt 2>&1; tail -1 $OUT/make.txt
for rep in 1 2; do for nt in 24 16 12 8; do
WO_FLOOD_THREADS=$nt WO_FLOOD_TIMING=1 taskset -c 0-63,128-191 python research/flood/walk_spread_probe.py run t$nt 24 >> $OUT/t$nt.out 2>> $OUT/t$nt.err
done; done
for rep in 1 2; do for nt in 24 16 12; do
WO_FLOOD_THREADS=$nt python bench.py --timed-only --steps 8 --warmup 2 > $OUT/bench_t${nt}_$rep.json 2> /dev/null
done; done
python - <<'PY'
import re, json
for nt in (24,16,12,8):
    t=open(f"/root/repo/gpurun_out/r05ai/t{nt}.err").read()
    w=[float(x) for x in re.findall(r"walk of the largest landmass \(\d+ cells\): ([\d.]+) ms", t)]
    j=[float(x) for x in re.findall(r"round joined at ([\d.]+) ms", t)]
    w2=sorted(w); j2=sorted(j)
    print(f"{nt} flood threads: walks min {w2[0]:.1f} median {w2[len(w2)//2]:.1f} mean {sum(w)/len(w):.1f} max {w2[-1]:.1f}; round joined median {j2[len(j2)//2]:.1f} mean {sum(j)/len(j):.1f}")
for nt in (24,16,12):
    for rep in (1,2):
        d=json.loads(open(f"/root/repo/gpurun_out/r05ai/bench_t{nt}_{rep}.json").read().strip().splitlines()[-1])
        print("bench", nt, rep, round(d["ms_per_step"],1), d["stage_ms_last_step"]["priority_flood"], d["parity"]["parity_crc_ok"])
PY
