"""Probe (GPU box): throughput of B planets in flight on ONE GPU (one host thread, context and stream per planet) —
the ensemble mode of BASELINE config 5 run 8 planets per GPU.  Usage: python profiles/ensemble_probe.py [cells] [iters]"""
import sys
import threading
import time

sys.path.insert(0, ".")
import bench  # noqa: E402
from planet_heightmap_generation_amd import terrain_post as TP  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
print("numa node:", bench.bind_to_gpu_numa_node(0))
params = dict(bench.PARAMS)
if iters != 200:
    params.update(hIters=iters, tIters=iters, gIters=min(10, max(1, iters // 20)))
mesh, xyz, nd, t_mesh = bench.build_inputs(cells, 1)
N = mesh.numRegions
for B in (1, 2, 3, 4, 6):
    planets = []
    for j in range(B):
        pl = TP.Planet(mesh, xyz, nd, ctx=TP.Context(0))
        pl.synthetic_terrain(1)
        pl.save_state()
        pl.sync()
        planets.append(pl)
    for pl in planets:                       # warm-up (flood statics, patch tables), sequentially
        bench.one_step(pl, 1, params)
        pl.sync()
    steps = 2
    def work(pl):
        for _ in range(steps):
            bench.one_step(pl, 1, params)
        pl.sync()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(pl,)) for pl in planets]
    for t in th: t.start()
    for t in th: t.join()
    wall = time.perf_counter() - t0
    print(f"B={B}: {wall / steps * 1e3:.0f} ms per round of {B} planets -> {N * iters * steps * B / wall / 1e6:.0f} Mcells·iter/s "
          f"({wall / steps / B * 1e3:.0f} ms per planet)", flush=True)
    for pl in planets:
        pl.close()
