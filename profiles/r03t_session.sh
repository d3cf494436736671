#!/bin/bash
# Round-3 session T: per-dispatch durations of the carve / ice rounds (kernel trace of one step at 20 iterations)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"; O=gpurun_out/r03t; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/prof -o tr -- python bench.py --no-cpu --in-flight 0 --steps 1 --warmup 0 --no-profile --iters 40 > $O/bench_trace.log 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r03t/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_carve_round_static" in r["Kernel_Name"]]
g = [ (int(rows[i+1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"])) / 1e3 for i in range(len(rows)-1) if "k_carve_round_static" in rows[i]["Kernel_Name"] and "k_carve_round_static" in rows[i+1]["Kernel_Name"]]
print("carve rounds", len(d), "mean", sum(d)/len(d), "gaps mean", sum(g)/max(1,len(g)))
print("first 40 durations", [round(x,1) for x in d[:40]])
print("durations 100..140", [round(x,1) for x in d[100:140]])
h = collections.Counter(int(x//2)*2 for x in d); print("histogram (us bucket: count)", sorted(h.items()))
i = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_ice_round" in r["Kernel_Name"]]
print("ice rounds", len(i), "mean", sum(i)/max(1,len(i)), [round(x,1) for x in i[:30]])
PY
rm -rf $O/prof
