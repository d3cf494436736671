"""GPU probe: step time and solve stage of the bench workload for a few settings of the patch solve's in-launch
revisits (child processes: the settings are read once per process)."""
import json, os, subprocess, sys
code = r'''
import sys, time, json, zlib
sys.path.insert(0, ".")
from planet_heightmap_generation_amd import sphere_mesh as S, terrain_post as TP
cells, iters = int(sys.argv[1]), int(sys.argv[2])
mesh, xyz, nd = S.build_sphere(cells, 0.75, 1)
pl = TP.Planet(mesh, xyz, nd); pl.synthetic_terrain(1); pl.save_state()
for rep in range(2):
    pl.restore_state(); t0 = time.perf_counter()
    pl.warp_terrain_resident(1, 0.75); pl.ocean_from_elevation()
    pl.erode_composite_resident(iters, 3e-4, 0.5, 1.0, iters, 1.16, 0.015, min(10, max(1, iters // 20)), 0.5)
    pl.apply_soil_creep_resident(3, 0.1125); pl.sync(); ms = (time.perf_counter() - t0) * 1e3
st = pl.last_stage_timing(); es = pl.last_erode_stats()
print(json.dumps(dict(step_ms=round(ms, 1), solve_ms=round(st["solve"], 1), launches=es["solve_patch_launches_total"], crc=int(zlib.crc32(pl.download().tobytes())))))
'''
cells, iters = sys.argv[1], sys.argv[2]
for rv, wt in ((0, 0), (2, 8), (4, 16), (6, 24), (8, 48), (12, 64), (16, 128)):
    env = dict(os.environ, WO_PATCH_REVISITS=str(rv), WO_PATCH_WAIT=str(wt))
    out = subprocess.run([sys.executable, "-c", code, cells, iters], env=env, capture_output=True, text=True)
    print("revisits", rv, "wait", wt, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
