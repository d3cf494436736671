"""EXPERIMENT (round 5): can the host flood be hidden behind device work by running ONE planet as two landmass shares on one GPU —
share 0 the largest landmass (its walk is the critical path of every flood call, 27-37 ms of the 36-40 ms), share 1 everything else —
each with its own host thread, context, stream and planet, WITHOUT the flood exchange (at 10 M cells no flood call is undecided, which
the script checks: 0 replays), so that share 1 starts iterating ~5 ms into a flood call while share 0's walk is still running?

Prints the wall time of erodeComposite + creep for the single planet and for the two shares started together (wall = until both are
done), whether the merged field is bit-identical, and each share's own time.  Usage: python research/ab/r05_two_shares_flood_overlap.py [cells]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import bench as B                                                                    # noqa: E402
from planet_heightmap_generation_amd import decomposed as D, terrain_post as TP      # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
params = dict(B.PARAMS)
args_e = (params["hIters"], params["K"], params["m"], params["dt"], params["tIters"], params["talusSlope"], params["kThermal"], params["gIters"], params["glacialStrength"])
mesh, xyz, nd, _ = B.build_inputs(cells, 1)
pl = TP.Planet(mesh, xyz, nd, device=0)
pl.synthetic_terrain(1)
pl.save_state()


def prepared():
    pl.restore_state(); pl.warp_terrain_resident(1, B.WARP); pl.ocean_from_elevation(); pl.sync()


single = []
for k in range(reps + 1):
    prepared()
    t0 = time.perf_counter()
    pl.erode_composite_resident(*args_e); pl.apply_soil_creep_resident(*B.CREEP); pl.sync()
    single.append((time.perf_counter() - t0) * 1e3)
ref = pl.download()
prepared()
field, oc = pl.download(), pl.download_ocean()
plan = D.plan_largest_apart(mesh, oc)
print("land cells per share", [int(v) for v in plan.load], "landmasses", plan.num_landmasses, flush=True)
shares = [TP.Planet(mesh, xyz, nd, ctx=TP.Context(0)) for _ in range(2)]
masks = [plan.rank_mask(k, oc) for k in range(2)]
two, own = [], []
for rep in range(reps + 1):
    for k in range(2):
        shares[k].upload(field, masks[k]); shares[k].sync()
    bar = threading.Barrier(3)
    secs, stats = [0.0, 0.0], [None, None]

    def work(k):
        bar.wait()
        t0 = time.perf_counter()
        shares[k].erode_composite_resident(*args_e); shares[k].apply_soil_creep_resident(*B.CREEP); shares[k].sync()
        secs[k] = (time.perf_counter() - t0) * 1e3
        stats[k] = shares[k].last_erode_stats()
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    two.append((time.perf_counter() - t0) * 1e3); own.append([round(v, 1) for v in secs])
    assert all(int(s["flood_host_replays"]) == 0 and int(s["flood_host_serial_pass1"]) == 0 for s in stats), "a flood call was undecided: the shares needed the exchange"
merged = field.copy()
for k in range(2):
    out = shares[k].download()
    merged[plan.cells[k]] = out[plan.cells[k]]
land = np.flatnonzero(oc == 0)
print("single planet ms", [round(v, 1) for v in single[1:]])
print("two shares  ms", [round(v, 1) for v in two[1:]], "per share", own[1:])
print("cells that differ on land", int((merged[land] != ref[land]).sum()), "stage ms of the shares", [shares[k].last_stage_timing() for k in range(2)])
