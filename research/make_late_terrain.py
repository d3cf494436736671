"""A planet as the SECOND priority flood of a step meets it (after ~100 erosion iterations), for research/flood_walk_bench.py:
python research/make_late_terrain.py [cells] [iterations]  ->  /tmp/fwb_<cells>_late.npz
The oracle's erodeComposite (one host core, ~1.3 s per iteration at 10 M cells) on the cached planet of flood_walk_bench.py."""
import sys, time
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from planet_heightmap_generation_amd import sphere_mesh as S
from oracle import pyoracle as O
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
z = np.load(f"/tmp/fwb_{cells}.npz")
t = time.time(); mesh, xyz, nd = S.build_sphere(cells, 0.75, 1); print(f"mesh {time.time()-t:.1f} s", flush=True)
assert np.array_equal(mesh.adjOffset, z["off"])
e0 = z["e0"]; oc = (e0 <= 0).astype(np.uint8)
om = O.Mesh(mesh.adjOffset, mesh.adjList)
t = time.time()
e = O.erode_composite(om, e0, xyz, oc, iters, 3e-4, 0.5, 1.0, iters, 1.16, 0.015, 10, 0.5, nd)
print(f"erodeComposite x{iters}: {time.time()-t:.1f} s", flush=True)
np.savez(f"/tmp/fwb_{cells}_late.npz", xyz=z["xyz"], e0=e, off=z["off"], adj=z["adj"], oc=oc)
