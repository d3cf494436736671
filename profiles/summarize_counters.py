"""Per-kernel averages of every counter in the rocprofv3 --pmc counter_collection CSVs under <root>/<pass>/ (one pass per
sub-directory): {kernel: {counter: per-launch average, "launches": n}} plus derived ratios."""
import csv, glob, json, sys
from collections import defaultdict
root, iters = sys.argv[1], int(sys.argv[2])
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        a = acc[name][row["Counter_Name"]]
        a[0] += 1; a[1] += float(row["Counter_Value"])
out = {}
for name, cs in acc.items():
    d = {c: round(t / max(1, n), 1) for c, (n, t) in cs.items()}
    d["launches"] = max(n for n, _ in cs.values())
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in d: d[k + "_frac_of_wave_cycles"] = round(d[k] / wc, 3)
    if d.get("TCC_HIT_sum") is not None and d.get("TCC_MISS_sum") is not None and d["TCC_HIT_sum"] + d["TCC_MISS_sum"] > 0:
        d["L2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 3)
    if d.get("SQ_WAVES") and d.get("SQ_INSTS_VALU") is not None:
        d["valu_insts_per_wave"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1)
    if d.get("SQ_WAVES") and d.get("SQ_INSTS_VMEM_RD") is not None:
        d["vmem_rd_insts_per_wave"] = round(d["SQ_INSTS_VMEM_RD"] / d["SQ_WAVES"], 1)
    out[name] = d
out["_meta"] = {"workload": f"bench.py --cells 10000000 --iters {iters} --steps 1 --warmup 0 --no-profile --no-cpu", "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"}
print(json.dumps(out, indent=1, sort_keys=True))
