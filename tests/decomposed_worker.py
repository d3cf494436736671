"""Worker for tests/test_decomposed.py: every rank erodes the planet with the other ranks' landmasses masked as ocean,
then the land elevations are merged over gloo.  Engine 'oracle' (CPU tests: the oracle is the checker's engine here,
never the product's), 'planet' (the HIP path through the C ABI, resident field, on the GPU box; host-staged exchange as
under gloo) or 'planet-device' (same, but the exchange takes the branch a RCCL run takes: device tensors packed / unpacked
through device pointers; the collective itself is stood in for by gloo behind a shim, since ranks sharing one GPU cannot
form a RCCL communicator)."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch.distributed as dist  # noqa: E402

from planet_heightmap_generation_amd import decomposed  # noqa: E402


class DeviceTensorGloo:
    """all_gather_into_tensor on device tensors with gloo underneath (stands in for RCCL when the ranks share one GPU)."""

    def all_gather_into_tensor(self, out, send):
        import torch
        torch.cuda.synchronize()
        host = [torch.empty(send.numel(), dtype=send.dtype) for _ in range(dist.get_world_size())]
        dist.all_gather(host, send.cpu())
        out.copy_(torch.cat(host).to(out.device))


def main():
    work, engine = Path(sys.argv[1]), sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    z = np.load(work / "case.npz")
    h, t, g = (int(v) for v in z["iters"])

    class M:
        adjOffset, adjList, numRegions = z["adjOffset"], z["adjList"], z["adjOffset"].size - 1
    e, oc, xyz, nd = z["elevation"].copy(), z["isOcean"], z["xyz"], z["neighborDist"]
    plan = decomposed.plan_landmasses(M, oc, world)
    mask = plan.rank_mask(rank, oc)
    if engine == "oracle":
        from oracle import pyoracle as O
        om = O.Mesh(M.adjOffset, M.adjList)
        e = O.erode_composite(om, e, xyz, mask, h, 3e-4, 0.5, 1.0, t, 1.16, 0.015, g, 0.5, nd)
        e = O.soil_creep(om, e, mask, 3, 0.1125)
        decomposed.merge_land(plan, rank, e, dist)
    else:
        from planet_heightmap_generation_amd import terrain_post as TP
        pl = TP.Planet(M, xyz, nd)
        pl.upload(e, mask)
        pl.set_flood_exchange(oc, decomposed.TorchFloodExchange(plan, rank, dist))      # equal flood keys that matter: the shares pool their heights
        pl.erode_composite_resident(h, 3e-4, 0.5, 1.0, t, 1.16, 0.015, g, 0.5)
        pl.apply_soil_creep_resident(3, 0.1125)
        if engine == "planet-device":
            decomposed.ResidentLandmass(plan, rank, pl).exchange(DeviceTensorGloo(), "cuda:0")
        else:
            decomposed.ResidentLandmass(plan, rank, pl).exchange(dist)
        e = pl.download()
        pl.close()
    np.save(work / f"result_{rank}.npy", e)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
