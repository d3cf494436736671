"""PROBE (round 5): why does the walk of the largest landmass take 31-44 ms on the same input?  Makes an eroded 10 M-cell state on the GPU, then runs the host flood
(test emulator library, same flood_host.cc) `reps` times per configuration in fresh processes and prints the walk times.  Usage: python research/flood/walk_spread_probe.py make|run <tag>"""
import ctypes as C
import os
import re
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, ".")
STATE = "/tmp/wo_probe_state.npy"
OCEAN = "/tmp/wo_probe_ocean.npy"
if sys.argv[1] == "make":
    import bench as B
    from planet_heightmap_generation_amd import terrain_post as TP
    mesh, xyz, nd, _ = B.build_inputs(10_000_000, 1)
    pl = TP.Planet(mesh, xyz, nd, device=0)
    pl.synthetic_terrain(1); pl.warp_terrain_resident(1, B.WARP); pl.ocean_from_elevation()
    oc = pl.download_ocean()
    p = dict(B.PARAMS)
    pl.erode_composite_resident(150, p["K"], p["m"], p["dt"], 150, p["talusSlope"], p["kThermal"], 10, p["glacialStrength"]); pl.sync()
    np.save(STATE, pl.download()); np.save(OCEAN, oc)
    print("state made")
    sys.exit(0)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
from planet_heightmap_generation_amd import sphere_mesh as S        # noqa: E402
mesh, xyz, nd = S.build_sphere(10_000_000, 0.75, 1)
e = np.load(STATE); oc = np.load(OCEAN)
L = C.CDLL(os.environ.get("WO_EMU_LIB", "tests/emu/_build/libemu.so")); p = C.c_void_p
L.emu_flood_host.argtypes = [C.c_int32, p, p, p, p, p, C.c_double, C.c_int32, C.c_int32, p]
P = lambda a: a.ctypes.data_as(p)                                   # noqa: E731
off = np.ascontiguousarray(mesh.adjOffset, np.int32); adj = np.ascontiguousarray(mesh.adjList, np.int32); xyz = np.ascontiguousarray(xyz, np.float32)
st = np.zeros(11); out = e.copy()
t0 = time.time()
L.emu_flood_host(off.size - 1, P(off), P(adj), P(xyz), P(out), P(oc), 0.85, 11, reps, P(st))
print(sys.argv[2], "total s", round(time.time() - t0, 2), "calls", st[0], "pass1 ms per call", round(st[7] / max(st[0], 1), 2), flush=True)
