"""Host flood timing on a CPU-built planet (no GPU): python research/flood_walk_bench.py [cells] [repeats]
Prints flood_host.cc's own timing lines (WO_FLOOD_TIMING): walk of the largest landmass, pipeline, write-back.
The planet is the oracle's synthetic terrain + warp; the flood runs through the test emulator library (tests/emu), which
compiles the same flood_host.cc as libworogen."""
import ctypes as C, os, subprocess, sys, time
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
os.environ["WO_FLOOD_TIMING"] = "1"
from planet_heightmap_generation_amd import sphere_mesh as S
from oracle import pyoracle as O
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
O.lib()
subprocess.run(["make", "-s", "-C", str(REPO / "tests" / "emu")], check=True)
L = C.CDLL(os.environ.get("WO_FWB_LIB", str(REPO / "tests" / "emu" / "_build" / "libemu.so")))    # WO_FWB_LIB: the same sources built another way (e.g. clang -O3, as libworogen is)
p = C.c_void_p
L.emu_flood_host.argtypes = [C.c_int32, p, p, p, p, p, C.c_double, C.c_int32, C.c_int32, p]
late = os.environ.get("WO_FWB_LATE")          # the planet after ~100 erosion iterations (research/make_late_terrain.py): what the second flood of a step meets
cache = Path(os.environ.get("WO_FWB_CACHE", "/tmp")) / (f"fwb_{cells}_late.npz" if late else f"fwb_{cells}.npz")
if cache.exists():                      # the planet of an earlier run of this script
    z = np.load(cache); xyz, e0 = z["xyz"], z["e0"]
    class M: pass
    mesh = M(); mesh.adjOffset, mesh.adjList = z["off"], z["adj"]
else:
    t = time.time(); mesh, xyz, nd = S.build_sphere(cells, 0.75, 1); print(f"mesh {time.time()-t:.1f} s", flush=True)
    t = time.time()
    try:                                   # on a GPU box the HIP path makes the (bit-identical) terrain in milliseconds
        from planet_heightmap_generation_amd import terrain_post as TP
        pl = TP.Planet(mesh, xyz, nd); pl.synthetic_terrain(1); pl.warp_terrain_resident(1, 0.75); e0 = pl.download(); pl.close()
    except Exception:
        om = O.Mesh(mesh.adjOffset, mesh.adjList)
        e0 = O.warp_terrain(om, O.synthetic_terrain(xyz, 1), xyz, 1, 0.75)
    print(f"terrain {time.time()-t:.1f} s", flush=True)
    np.savez(cache, xyz=xyz, e0=e0, off=mesh.adjOffset, adj=mesh.adjList)
oc = z["oc"] if (cache.exists() and "oc" in z.files) else (e0 <= 0).astype(np.uint8)
P = lambda a: a.ctypes.data_as(p)
e = e0.copy(); st = np.zeros(11)
t = time.time()
L.emu_flood_host(mesh.adjOffset.size - 1, P(mesh.adjOffset), P(mesh.adjList), P(xyz), P(e), P(oc), 0.5, 11, reps, P(st))
import zlib
print(f"{reps} floods {time.time()-t:.2f} s; stats {st.tolist()}; crc32 of the field {zlib.crc32(e.tobytes())}")
