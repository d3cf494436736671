"""Timing probe (GPU box): plate projection + assignElevation on a large sphere with synthetic plates (Voronoi plates
of random coarse seeds, random motion / ocean flags) — the reference's generatePlates is host logic we do not replace,
so this measures the per-cell stages at scale, not parity.  Usage: python profiles/time_elevation.py [cells]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from planet_heightmap_generation_amd import coarse_plates as CP, elevation as EL, sphere_mesh as S  # noqa: E402
from planet_heightmap_generation_amd.terrain_post import Planet  # noqa: E402

import bench  # noqa: E402  (NUMA pinning helper)

print("numa node:", bench.bind_to_gpu_numa_node(0))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
t = time.time(); mesh, xyz, nd = S.build_sphere(N, 0.75, 1); print(f"mesh {N}: {time.time() - t:.1f} s")
cm, cxyz, _ = S.build_sphere(20000, 0.75, 138)
rng = np.random.default_rng(1)
seeds = rng.choice(20001, 80, replace=False).astype(np.int32)
P3 = cxyz.reshape(-1, 3).astype(np.float64)
cplate = seeds[np.argmax(P3 @ P3[seeds].T, axis=1)].astype(np.int32)
pl = Planet(mesh, xyz, nd)
t = time.time(); rp = CP.project_coarse_plates(mesh, xyz, cm, cxyz, cplate, 1, 80, planet=pl); print(f"projectCoarsePlates: {(time.time() - t) * 1e3:.1f} ms")
t = time.time(); CP.smooth_and_reconnect_plates(mesh, rp, seeds, 3); print(f"smoothAndReconnectPlates: {(time.time() - t) * 1e3:.1f} ms")
plateVec, dens, ocean = {}, {}, set()
for s in seeds.tolist():
    v = rng.normal(size=3); v /= np.linalg.norm(v)
    plateVec[s] = {"pole": v.tolist(), "omega": float((0.5 + rng.random() * 1.5) * (1 if rng.random() < 0.5 else -1))}
    if rng.random() < 0.6:
        ocean.add(s)
    dens[s] = (3.0 if s in ocean else 2.4) + rng.random() * 0.5
for rep in range(2):
    t = time.time()
    res = EL.assign_elevation(mesh, xyz, ocean, rp, plateVec, seeds.tolist(), EL.SimplexNoise(1), 0.4, 1, 5, dens, None, planet=pl, debug=False)
    print(f"assignElevation: {(time.time() - t) * 1e3:.0f} ms; land fraction {(res['r_elevation'] > 0).mean():.3f}")
    for st in res["_timing"]:
        print(f"    {st['stage']:40s} {st['ms']:9.1f} ms")
