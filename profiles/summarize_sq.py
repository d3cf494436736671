"""Per-kernel sums of rocprofv3 --pmc SQ counters: python profiles/summarize_sq.py <dir> -> JSON {kernel: {counter: total, launches}}.
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES per SE."""
import csv, glob, json, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[name].add((f, row.get("Dispatch_Id")))
out = {k: dict(v, launches=len(launches[k])) for k, v in acc.items()}
print(json.dumps(out, indent=1, sort_keys=True))
