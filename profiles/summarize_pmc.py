"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: {kernel: {COUNTER_KB: {launches, total, per_launch}}}.
FETCH_SIZE / WRITE_SIZE are reported in KB (MI355X_MICROARCH.md, HBM section: FETCH_SIZE halves wide coalesced
reads on gfx950 and is uncalibrated for narrow gathers — treat as a lower bound)."""
import csv
import glob
import json
import sys
from collections import defaultdict

root, iters = sys.argv[1], int(sys.argv[2])
out = defaultdict(dict)
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{root}/{counter}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            a = acc[name]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    for name, (n, tot) in acc.items():
        out[name][counter + "_KB"] = {"launches": n, "total": tot, "per_launch": tot / max(1, n)}
out["_meta"] = {"workload": f"bench.py --cells 10000000 --iters {iters} --steps 1 --warmup 0", "units": "KB as reported by rocprofv3"}
print(json.dumps(out, indent=1, sort_keys=True))
