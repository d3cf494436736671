"""Per-erosion-iteration durations of the once-per-iteration kernels, from a rocprofv3 --kernel-trace CSV of ONE bench
step (bench.py --steps 1 --warmup 0): for each kernel the n-th dispatch is iteration n, so a duration that depends on the
iteration (events left, drainage depth) shows as a trend and one that depends on the machine shows as noise.
usage: python profiles/per_iteration_durations.py <dir with *kernel_trace.csv> [iterations]"""
import csv, glob, json, sys
from collections import defaultdict
root = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
per = defaultdict(list)
for f in glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        per[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
out = {}
for name, v in per.items():
    if len(v) < iters or len(v) > 3 * iters + 16 or name == "wo::k_solve_patch": continue      # once-per-iteration kernels only (a cold step may precede the timed one)
    d = [x[1] / 1e3 for x in v[-iters:]]                          # us, iteration order
    dec = [round(sum(d[i * iters // 10:(i + 1) * iters // 10]) / (iters // 10), 1) for i in range(10)]
    s = sorted(d)
    out[name] = {"launches": len(v), "us_min": round(s[0], 1), "us_median": round(s[len(s) // 2], 1), "us_max": round(s[-1], 1),
                 "us_mean_by_tenth_of_the_run": dec, "us_first_20": [round(x, 1) for x in d[:20]]}
# the solve: launches per iteration and their durations by position in the pass
sp = per.get("wo::k_solve_patch") or []
if sp:
    setup = [x[0] for x in per.get("wo::k_solve_setup", [])]
    by_pos = defaultdict(list); k = 0; counts = []
    for i, t0 in enumerate(setup):
        t1 = setup[i + 1] if i + 1 < len(setup) else 1 << 62
        pos = 0
        while k < len(sp) and sp[k][0] < t0: k += 1
        while k < len(sp) and sp[k][0] < t1: by_pos[pos].append(sp[k][1] / 1e3); pos += 1; k += 1
        counts.append(pos)
    out["wo::k_solve_patch"] = {"launches_per_pass_mean": round(sum(counts) / max(1, len(counts)), 1),
                                "us_mean_by_position_in_pass": [round(sum(by_pos[q]) / len(by_pos[q]), 1) for q in sorted(by_pos) if len(by_pos[q]) >= len(setup) // 2]}
print(json.dumps(out, indent=1, sort_keys=True))
