import csv, glob, sys, collections
root = sys.argv[1]
rows = collections.defaultdict(dict)   # dispatch id -> counter -> value
names = {}
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_solve_patch" not in r["Kernel_Name"]: continue
        d = int(r["Dispatch_Id"]); rows[d][r["Counter_Name"]] = rows[d].get(r["Counter_Name"], 0) + float(r["Counter_Value"])
ids = sorted(rows)
cs = sorted({c for d in rows.values() for c in d})
print("n", len(ids)); print("idx " + " ".join(cs))
# last iteration: find last gap... just print the last 70 dispatches
for i, d in enumerate(ids[-70:]):
    print(i, " ".join(f"{rows[d].get(c,0):.0f}" for c in cs))
