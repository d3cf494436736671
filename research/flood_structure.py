"""Research driver: statistics of the reference flood's pop order on synthetic planets (see flood_structure.cc)."""
import ctypes as C, json, sys, time
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from oracle import pyoracle as O
from planet_heightmap_generation_amd import sphere_mesh as S

lib = C.CDLL("/tmp/libfloodres.so")
p = C.c_void_p
lib.flood_structure.argtypes = [C.c_int32, p, p, p, p, C.c_int32, p, p, p, p, p, p]
lib.flood_bf.argtypes = [C.c_int32, p, p, p, p, p, p, p, p, C.c_double, C.c_int32, C.c_int32]

def run(mesh, e, oc, deltas):
    d = np.asarray(deltas, np.float64)
    out = np.zeros(64 + 8 * d.size)
    a = lambda x: x.ctypes.data_as(p)
    N = mesh.numRegions
    T = np.empty(N, np.int32); par = np.empty(N, np.int32); Sx = np.empty(N, np.float32); Kx = np.empty(N, np.float32)
    lib.flood_structure(N, a(mesh.adjOffset), a(mesh.adjList), a(e), a(oc), d.size, a(d), a(out), a(T), a(par), a(Sx), a(Kx))
    for DELTA in (0.002, 0.01, 0.05, 10.0):
        bf = np.zeros(32)
        t0 = time.time()
        lib.flood_bf(N, a(mesh.adjOffset), a(mesh.adjList), a(e), a(oc), a(par), a(Sx), a(Kx), a(bf), DELTA, 1, 0)
        bfn = ["rounds", "evals", "changes", "maxDirty", "mismatchParent", "mismatchS", "mismatchK", "tieCompares", "deepCompares", "maxStackDepth", "overflow", "walks", "walkSteps", "walkReject", "maxWalk", "resets", "pendingTotal"]
        print("BF delta", DELTA, {n: int(bf[i]) for i, n in enumerate(bfn) if n not in ("walks", "walkSteps", "walkReject", "maxWalk")}, "sec %.2f" % (time.time() - t0), flush=True)
    names = ["L", "seeds", "popped", "maxHeap", "maxDepth", "units", "maxUnit", "units>8", "units>64", "units>1k", "units>16k", "cellsInUnits>64",
             "tauViolations", "consecutiveEqualKeyPops", "equalKeyPairsWithin2hops", "flooded", "maxHopInUnit", "tauMin", "tauMax"]
    res = {n: out[i] for i, n in enumerate(names)}
    res["windows"] = [dict(zip(["delta", "nWindows", "nonEmpty", "sumMaxChain+1", "maxChain", "maxPops"], out[64 + 8 * k: 64 + 8 * k + 6])) for k in range(d.size)]
    return res

if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    mesh, xyz, nd = S.build_sphere(N, 0.75, 1)
    print("mesh", time.time() - t0, flush=True)
    om = O.Mesh(mesh.adjOffset, mesh.adjList)
    e = O.synthetic_terrain(xyz, 1)
    e = O.warp_terrain(om, e, xyz, 1, 0.75)
    oc = (e <= 0).astype(np.uint8)
    print("terrain", time.time() - t0, flush=True)
    deltas = [0.0005, 0.001, 0.002, 0.005, 0.01, 0.02]
    print(json.dumps(run(mesh, e, oc, deltas), indent=1), flush=True)
    if iters > 0:
        e2 = O.erode_composite(om, e, xyz, oc, iters, 3e-4, 0.5, 1.0, iters, 1.16, 0.015, 10, 0.5, nd)
        print("eroded", time.time() - t0, flush=True)
        print(json.dumps(run(mesh, e2, oc, deltas), indent=1))
