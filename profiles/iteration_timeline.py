"""One composite iteration on the timeline, from a rocprofv3 --kernel-trace CSV: every dispatch between the n-th and the (n+1)-th
launch of the marker kernel (default k_sort_keys: the first kernel of an iteration), with start offset, duration and the gap to the
previous dispatch's end on the same queue.  Usage: python profiles/iteration_timeline.py <dir or csv> [n=500] [marker=k_sort_keys]"""
import csv
import glob
import sys

root = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
marker = sys.argv[3] if len(sys.argv) > 3 else "k_sort_keys"
f = root if root.endswith(".csv") else glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(marks) < n + 2:
    n = max(0, len(marks) - 2)
a, b = marks[n], marks[n + 1]
t0 = int(rows[a]["Start_Timestamp"])
print(f"{f}: dispatches {a}..{b - 1} (iteration marker #{n} of {len(marks)}), {(int(rows[b]['Start_Timestamp']) - t0) / 1000:.1f} us from marker to marker")
last_end = {}
print(f"{'start_us':>9} {'dur_us':>8} {'gap_us':>7} {'queue':>6}  kernel")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "0")
    gap = (s - last_end[q]) / 1000 if q in last_end else 0.0
    last_end[q] = e
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:70]
    print(f"{(s - t0) / 1000:9.1f} {(e - s) / 1000:8.1f} {gap:7.1f} {q:>6}  {name}")
