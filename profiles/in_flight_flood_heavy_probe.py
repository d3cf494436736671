"""Planets in flight on one GPU when the host flood dominates (BASELINE config 5's usual case: profiles/r06n_*): B planets of the bench mesh, each with ANOTHER terrain
(seeds 101 ..), one warm-up step each, then the same B steps (i) one after the other and (ii) concurrently (one host thread, context and stream per planet).
python profiles/in_flight_flood_heavy_probe.py [B=6]"""
import sys, threading, time, zlib
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
from planet_heightmap_generation_amd import terrain_post as TP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mesh, xyz, nd, _ = bench.build_inputs(10_000_000, 1)
params = dict(bench.PARAMS)
planets = []
for i in range(B):
    q = TP.Planet(mesh, xyz, nd, ctx=TP.Context(0))
    q.synthetic_terrain(101 + i); q.save_state()
    bench.one_step(q, 101 + i, params); q.sync()
    planets.append(q)
def work(i):
    bench.one_step(planets[i], 101 + i, params); planets[i].sync()
each = []
t0 = time.perf_counter()
for i in range(B):
    t1 = time.perf_counter(); work(i); each.append((time.perf_counter() - t1) * 1e3)
seq = (time.perf_counter() - t0) * 1e3
crc_seq = [int(zlib.crc32(q.download().tobytes())) for q in planets]
for nb in sorted({B, max(2, B // 2)}):
    t0 = time.perf_counter()
    for g in range(0, B, nb):
        th = [threading.Thread(target=work, args=(i,)) for i in range(g, min(B, g + nb))]
        for t in th: t.start()
        for t in th: t.join()
    par = (time.perf_counter() - t0) * 1e3
    crc_par = [int(zlib.crc32(q.download().tobytes())) for q in planets]
    print(f"{B} flood-heavy planets: one after the other {seq:.0f} ms ({[round(v) for v in each]}; flood stage of the last {planets[-1].last_stage_timing().get('priority_flood', 0):.0f} ms), "
          f"{nb} in flight {par:.0f} ms = {seq / par:.2f}x, same fields: {crc_seq == crc_par}", flush=True)
for q in planets: q.close()
