"""Worker for tests/test_banded.py: every rank advances its index band of a Jacobi pass and exchanges halos over gloo.
Compute is injected: 'oracle' (CPU tests; the oracle is the checker's engine here, never the product's) or 'planet'
(the HIP kernels through the C ABI, on the GPU box)."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch.distributed as dist  # noqa: E402

from planet_heightmap_generation_amd import banded  # noqa: E402


def main():
    work, engine = Path(sys.argv[1]), sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    z = np.load(work / "case.npz")

    class M:
        adjOffset, adjList, numRegions = z["adjOffset"], z["adjList"], z["adjOffset"].size - 1
    plan = banded.BandPlan(M, world)
    part = plan.parts[rank]
    oc = plan.scatter(rank, z["isOcean"])
    results = {}
    if engine == "oracle":
        from oracle import pyoracle as O
        lm = O.Mesh(part.mesh.adjOffset, part.mesh.adjList)
        steps = {"smooth": lambda f, s: f.__setitem__(slice(None), O.smooth_elevation(lm, f, oc, 1, s)),
                 "creep": lambda f, s: f.__setitem__(slice(None), O.soil_creep(lm, f, oc, 1, s)),
                 "field": lambda f, s: f.__setitem__(slice(None), O.smooth_field(lm, f, 1))}
    else:
        from planet_heightmap_generation_amd import climate_util as CU, terrain_post as TP
        pl = TP.Planet(part.mesh, plan.scatter_xyz(rank, z["xyz"]), plan.scatter_slots(rank, z["neighborDist"]))
        steps = {"smooth": lambda f, s: pl.smooth_elevation(f, oc, 1, s),
                 "creep": lambda f, s: pl.apply_soil_creep(f, oc, 1, s),
                 "field": lambda f, s: CU.smooth_field(part.mesh, f, 1, planet=pl)}
    if engine == "resident":
        # the field stays in HBM; only halo values travel (pack kernel -> gloo -> unpack kernel)
        band = banded.ResidentBand(part, pl)
        for name, iters, strength in (("smooth", 4, 0.3), ("creep", 3, 0.1125)):
            pl.upload(plan.scatter(rank, z["elevation"]), oc)
            step = (lambda: pl.smooth_elevation_resident(1, strength)) if name == "smooth" else (lambda: pl.apply_soil_creep_resident(1, strength))
            band.jacobi(step, iters, dist)
            g = banded.gather_owned(plan, rank, pl.download(), dist)
            if rank == 0:
                results[name] = g
        if rank == 0:
            np.savez(work / "result.npz", **results)
        dist.barrier()
        dist.destroy_process_group()
        return
    for name, iters, strength in (("smooth", 4, 0.3), ("creep", 3, 0.1125), ("field", 5, 0.0)):
        f = plan.scatter(rank, z["elevation"]).copy()
        banded.banded_jacobi(part, f, iters, lambda fl: steps[name](fl, strength), dist)
        g = banded.gather_owned(plan, rank, f, dist)
        if rank == 0:
            results[name] = g
    if rank == 0:
        np.savez(work / "result.npz", **results)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
