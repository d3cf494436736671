"""Worker of test_config4_two_processes_over_gloo (tests/test_gpu_parity.py): BASELINE config 4's planet as ONE PROCESS PER RANK.
Every rank builds the same planet (mesh builder, synthetic terrain and warp on the device), erodes its landmass share through the C ABI
with the flood exchange over torch.distributed (gloo: decomposed.TorchFloodExchange — the path `bench.py --gpus N` takes when the RCCL
communicator is not used), merges the land elevations, and rank 0 writes the merged field's CRC and every rank's exchange counters."""
import json
import os
import sys
import zlib
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch.distributed as dist  # noqa: E402

from planet_heightmap_generation_amd import decomposed as D  # noqa: E402
from planet_heightmap_generation_amd import sphere_mesh as S  # noqa: E402
from planet_heightmap_generation_amd import terrain_post as TP  # noqa: E402


def main():
    work, cells, seed, iters, g = Path(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    mesh, xyz, nd = S.build_sphere(cells, 0.75, seed)
    pl = TP.Planet(mesh, xyz, nd)
    pl.synthetic_terrain(seed)
    pl.warp_terrain_resident(seed, 0.75)
    pl.ocean_from_elevation()
    oc = pl.download_ocean()
    plan = D.plan_landmasses(mesh, oc, world)
    pl.upload(None, plan.rank_mask(rank, oc))
    pl.set_flood_exchange(oc, D.TorchFloodExchange(plan, rank, dist))
    link = D.ResidentLandmass(plan, rank, pl)
    pl.erode_composite_resident(iters, 3e-4, 0.5, 1.0, iters, 1.16, 0.015, g, 0.5)
    stats = pl.last_erode_stats()
    pl.apply_soil_creep_resident(3, 0.1125)
    link.exchange(dist)
    out = pl.download()
    pl.set_flood_exchange(None)
    pl.close()
    mine = dict(rank=rank, land_cells=int(stats["land_cells"]), flood_exchange_calls=int(stats["flood_exchange_calls"]), gathers=int(stats["flood_exchange_gathers"]),
                whole_planet_floods=int(stats["flood_exchange_whole_planet_floods"]), serial_pass1=int(stats["flood_host_serial_pass1"]),
                crc32=int(zlib.crc32(out.tobytes())), sum=float(out.astype(np.float64).sum()))
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    if rank == 0:
        (work / "result.json").write_text(json.dumps(allr))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
