"""GPU probe: one planet eroded unpartitioned and as W landmass shares (sequentially in this process, merged on the host):
cells that differ, RMS, and the same with the device flood in cell-id tie order (WO_FLOOD=device WO_FLOOD_TIES=id), whose
order does not depend on which cells share a heap."""
import json, os, sys, time, zlib
import numpy as np
sys.path.insert(0, ".")
from planet_heightmap_generation_amd import decomposed as D, sphere_mesh as S, terrain_post as TP

cells, iters, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mesh, xyz, nd = S.build_sphere(cells, 0.75, 1)
g = 10 if iters == 200 else min(10, max(1, iters // 20))
pl = TP.Planet(mesh, xyz, nd)
pl.synthetic_terrain(1); pl.warp_terrain_resident(1, 0.75); pl.ocean_from_elevation(); pl.save_state()
e0, oc = pl.download(), pl.download_ocean()

times = {}
def stack(mask, key=None):
    best = None
    for rep in range(2 if key is not None else 1):                     # second run: tables for this mask are cached (steady state, as bench.py times it)
        pl.restore_state(); pl.upload(None, mask); pl.sync()
        t0 = time.perf_counter()
        pl.erode_composite_resident(iters, 3e-4, 0.5, 1.0, iters, 1.16, 0.015, g, 0.5)
        pl.apply_soil_creep_resident(3, 0.1125); pl.sync()
        best = (time.perf_counter() - t0) * 1e3
    if key is not None:
        times[key] = round(best, 1)
    return pl.download()

plan = D.plan_landmasses(mesh, oc, world)
out = {}
for mode in ("host", "device_id"):
    if mode == "device_id":
        os.environ["WO_FLOOD"] = "device"; os.environ["WO_FLOOD_TIES"] = "id"
    full = stack(oc, mode + '_full')
    merged = e0.copy()
    for k in range(world):
        part = stack(plan.rank_mask(k, oc), f'{mode}_share{k}')
        merged[plan.cells[k]] = part[plan.cells[k]]
    d = full.astype(np.float64) - merged.astype(np.float64)
    bad = np.flatnonzero(full != merged)
    labs = D.land_components(mesh, oc)
    out[mode] = dict(crc_full=int(zlib.crc32(full.tobytes())), cells_differ=int(bad.size), rms=float(np.sqrt((d * d).mean())), max_abs=float(np.abs(d).max()),
                     landmasses_touched=int(np.unique(labs[bad]).size) if bad.size else 0)
shares = [times[f"host_share{k}"] for k in range(world)]
print(json.dumps(dict(cells=cells, iters=iters, world=world, **out, step_ms_unpartitioned=times["host_full"], step_ms_per_share=shares,
                      land_cells_per_share=[int(c.size) for c in plan.cells],
                      projected_speedup_if_one_gpu_per_share=round(times["host_full"] / max(shares), 2),
                      note="erodeComposite + creep per share, measured one after the other on ONE GPU (steady state); warp / mask / merge not included")))
