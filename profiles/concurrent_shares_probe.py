"""Probe: ONE planet's erosion stack as S landmass shares run CONCURRENTLY on one GPU (one context/stream and host thread per
share), merged afterwards, against the unpartitioned run.   python profiles/concurrent_shares_probe.py [cells] [S ...]"""
import json, sys, threading, time, zlib
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
import bench as B
from planet_heightmap_generation_amd import terrain_post as TP, decomposed as D

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
shares = [int(a) for a in sys.argv[2:]] or [2, 3, 4, 6, 8]
B.bind_to_gpu_numa_node(0)
mesh, xyz, nd, _ = B.build_inputs(cells, 1)
params = dict(B.PARAMS) if hasattr(B, "PARAMS") else None
p = params
pl = TP.Planet(mesh, xyz, nd, ctx=TP.Context(0))
pl.synthetic_terrain(1); pl.save_state()
def stack(q):
    q.erode_composite_resident(p["hIters"], p["K"], p["m"], p["dt"], p["tIters"], p["talusSlope"], p["kThermal"], p["gIters"], p["glacialStrength"])
    q.apply_soil_creep_resident(*B.CREEP)
    q.sync()
for rep in range(2):
    pl.restore_state(); pl.warp_terrain_resident(1, B.WARP); pl.ocean_from_elevation(); pl.sync()
    oc = pl.download_ocean(); warped = pl.download()
    t0 = time.perf_counter(); stack(pl); whole_ms = (time.perf_counter() - t0) * 1e3
ref = pl.download()
print(json.dumps(dict(cells=cells, unpartitioned_stack_ms=round(whole_ms, 1), crc=zlib.crc32(ref.tobytes()))), flush=True)
for S in shares:
    plan = D.plan_landmasses(mesh, oc, S)
    qs = [TP.Planet(mesh, xyz, nd, ctx=TP.Context(0)) for _ in range(S)]
    walls = []
    for rep in range(2):
        for k, q in enumerate(qs):
            q.upload(warped, plan.rank_mask(k, oc)); q.sync()
        t0 = time.perf_counter()
        th = [threading.Thread(target=stack, args=(q,)) for q in qs]
        [t.start() for t in th]; [t.join() for t in th]
        walls.append((time.perf_counter() - t0) * 1e3)
    merged = warped.copy()
    t1 = time.perf_counter()
    for k, q in enumerate(qs):
        part = q.download(); merged[plan.cells[k]] = part[plan.cells[k]]
    merge_ms = (time.perf_counter() - t1) * 1e3
    print(json.dumps(dict(shares=S, concurrent_stack_ms=[round(w, 1) for w in walls], host_merge_ms=round(merge_ms, 1), identical=bool(np.array_equal(merged, ref)),
                          ndiff=int((merged != ref).sum()), land_per_share=[int(v) for v in plan.load])), flush=True)
    for q in qs:
        q.close()
