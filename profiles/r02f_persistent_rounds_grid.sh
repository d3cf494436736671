O=gpurun_out/r02f; mkdir -p $O
for g in 32 64 128 256; do WO_ROUND_GRID=$g timeout 300 python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 1 > $O/bench_pc_$g.log 2>&1; python - $O/bench_pc_$g.log $g <<P
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); fam=d["roofline"]["families"]
        print("grid", sys.argv[2], "ms/step %.0f"%d["ms_per_step"], "crc", d["parity"]["parity_crc_ok"], "glacial %.0f"%d["stage_ms_last_step"]["glacial"], {k:(round(fam[k]["ms"],1),fam[k]["launches"]) for k in ("carve_round","ice_round")}, d["erode_stats"]["carve_rounds_total"])
P
done
