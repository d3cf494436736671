"""Where the GPU sits idle inside a step, from a rocprofv3 --kernel-trace CSV: the union of all dispatches' [start, end] over the last step (from the last
launch of the step's first kernel, default k_warp, to the end of the trace), then every idle stretch of at least `min_us`, with the kernels on either side.
Usage: python profiles/step_idle_gaps.py <dir or csv> [min_us=50] [marker=k_warp]"""
import csv
import glob
import sys

root = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
marker = sys.argv[3] if len(sys.argv) > 3 else "k_warp"
f = root if root.endswith(".csv") else glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a = marks[-1]
rows = rows[a:]
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:48]
t0 = int(rows[0]["Start_Timestamp"])
busy_end, prev = int(rows[0]["End_Timestamp"]), rows[0]
idle_total, listed, busy = 0.0, [], 0.0
cur_start = t0
for r in rows[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        gap = (s - busy_end) / 1000
        idle_total += gap
        busy += (busy_end - cur_start) / 1000
        cur_start = s
        if gap >= min_us:
            listed.append(((busy_end - t0) / 1000, gap, name(prev), name(r)))
    if e > busy_end:
        busy_end, prev = e, r
busy += (busy_end - cur_start) / 1000
print(f"{f}: last step = {len(rows)} dispatches over {(busy_end - t0) / 1000:.0f} us; GPU busy {busy:.0f} us, idle {idle_total:.0f} us; idle stretches >= {min_us:g} us:")
for at, gap, p, n in listed:
    print(f"  at {at:9.0f} us: {gap:8.0f} us   after {p:48s} before {n}")
small = idle_total - sum(g for _, g, _, _ in listed)
print(f"  + {small:.0f} us in shorter gaps")
