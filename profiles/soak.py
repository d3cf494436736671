"""Soak probe (GPU box): repeated planet create / full pipeline / destroy; device and host memory must stay flat."""
import os, sys, resource
sys.path.insert(0, ".")
import numpy as np, torch
from planet_heightmap_generation_amd import sphere_mesh as S, terrain_post as TP, coarse_plates as CP, climate_util as CU
mesh, xyz, nd = S.build_sphere(300000, 0.75, 1)
cm, cxyz, _ = S.build_sphere(20000, 0.75, 138)
cplate = (np.arange(cm.numRegions) % 40).astype(np.int32)
def used(): f, t = torch.cuda.mem_get_info(0); return (t - f) / 2**20
base = None
for it in range(40):
    ctx = TP.Context(0); pl = TP.Planet(mesh, xyz, nd, ctx=ctx)
    pl.synthetic_terrain(it); pl.warp_terrain_resident(it, 0.75); pl.ocean_from_elevation()
    pl.erode_composite_resident(6, 3e-4, 0.5, 1.0, 6, 1.16, 0.015, 2, 0.5); pl.apply_soil_creep_resident(3, 0.1)
    e = pl.download(); CU.smooth_field(mesh, e, 2, planet=pl)
    CP.project_coarse_plates(mesh, xyz, cm, cxyz, cplate, it, 40, planet=pl)
    pl.close(); ctx.close()
    if it in (4, 39):
        print(f"iter {it}: device used {used():.0f} MiB, host maxrss {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MiB", flush=True)
