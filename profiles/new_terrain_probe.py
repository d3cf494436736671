"""What a NEW terrain on a resident mesh costs (bench.py: new_terrain_step_ms): one warm step, then steps on other terrains with the
library's set-up laps (WO_FLOOD_TIMING: mirror, flood tables) on stderr and the stage times of each step."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
os.environ["WO_FLOOD_TIMING"] = "1"
import bench  # noqa: E402
from planet_heightmap_generation_amd import terrain_post as TP  # noqa: E402

cells = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
mesh, xyz, nd, _ = bench.build_inputs(cells, 1)
pl = TP.Planet(mesh, xyz, nd)
params = dict(bench.PARAMS)
for seed in (1, 1, 102, 103, 104, 104):
    pl.synthetic_terrain(seed); pl.save_state(); pl.sync()
    sys.stderr.write(f"==== seed {seed}\n"); sys.stderr.flush()
    t0 = time.perf_counter()
    pl.restore_state(); pl.warp_terrain_resident(seed, bench.WARP); pl.sync()
    t1 = time.perf_counter()
    pl.ocean_from_elevation(); pl.sync()
    t2 = time.perf_counter()
    pl.erode_composite_resident(params["hIters"], params["K"], params["m"], params["dt"], params["tIters"], params["talusSlope"], params["kThermal"], params["gIters"], params["glacialStrength"])
    pl.sync()
    t3 = time.perf_counter()
    pl.apply_soil_creep_resident(*bench.CREEP); pl.sync()
    t4 = time.perf_counter()
    st = pl.last_stage_timing()
    print(f"seed {seed}: step {(t4 - t0) * 1e3:.1f} ms = warp {(t1 - t0) * 1e3:.1f} + ocean {(t2 - t1) * 1e3:.1f} + erode {(t3 - t2) * 1e3:.1f} + creep {(t4 - t3) * 1e3:.1f}; stages {({k: round(v, 1) for k, v in st.items()})}", flush=True)
pl.close()
