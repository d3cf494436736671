"""Per-launch durations of one kernel from a rocprofv3 --kernel-trace CSV, in launch order.
Usage: python profiles/kernel_sequence.py <dir> <kernel substring> [max rows]"""
import csv
import glob
import sys

root, pat = sys.argv[1], sys.argv[2]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 400
f = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows]
print(len(d), "launches; total", round(sum(d) / 1000, 2), "ms")
print(" ".join(f"{x:.0f}" for x in d[:lim]))
