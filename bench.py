#!/usr/bin/env python3
"""bench.py — Mcells·iter/s of the terrain-post erosion stack (BASELINE.json metric) on MI355X.

A *step* is one pass of the hot path over one planet: restore the resident synthetic terrain, then
warpTerrain(0.75) -> isOcean = elev <= 0 -> erodeComposite(hIters=200, K=3e-4, m=0.5, dt=1, tIters=200,
talus=1.16, kThermal=0.015, gIters=10, glacialStrength=0.5) -> applySoilCreep(3, 0.1125)
(BASELINE config 3: "10M cells, same stack + FBM domain warp + glacial carve, 1xMI355X").
Inputs (mesh, r_xyz, neighborDist, elevation) are resident in HBM before the timed region; the two
priority-flood calls per step run on the host by design and are inside the timed region.

  value = numRegions * 200 iterations * steps * n_gpus / wall / 1e6          [Mcells·iter/s, whole job]

N > 1, default (--mode ensemble = BASELINE config 5): one process per GPU, one independent planet per rank with seed = 1 + rank,
no collective on the data path (only the barriers and the MAX of the wall time) -> value = cells x 200 x steps x N / wall,
"scaling": "weak".  The erosion stack's units of work are whole planets; a production job (64 seeds) shards them like this.
--mode decomposed: ONE planet (seed 1) eroded by all ranks — landmass decomposition (planet_heightmap_generation_amd/
decomposed.py: every rank erodes the full mesh with the other ranks' landmasses masked as ocean; one all-gather of the land
elevations per step over RCCL, behind the C ABI: wo_planet_exchange_allgather; bit-identical to the unpartitioned run at
10 M cells) -> value = cells x 200 x steps / wall of that one planet, "scaling": "strong".  Exact mode does not scale well
this way (the largest landmass and the fixed per-iteration costs bound a share; projected 1.6x at 8 GPUs),
which is why it is not the default.
N = 1 with --mode decomposed [--shares S]: the partitioned code path on ONE GPU — the planet is split into S landmass shares
(the plan an S-GPU run would use), the shares are eroded one after the other with the same masks, merged, and checked against
the oracle's CRC; the line reports the time of every share (what each of S GPUs would spend) next to the sequential total.

Prints ONE JSON line on rank 0 with the `roofline` (dominant kernel family, HIP-event timed) and
`cpu_baseline` (the CPU oracle on a bounded sample of the same workload, rank 0, N=1 only) objects.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

# algorithmic bytes of one composite iteration = 243 B x land cells + 12 B x cells, made of five passes.  Every
# pass's budget is split over the kernels that implement it, so that the per-kernel figures ADD UP to the pass budget (a kernel
# that only exists because of how the pass is implemented here — the basin layout, its sort — gets 0 bytes: its time still
# counts in the pass).  (per land cell, per cell) per launch of the pass.
PASSES = {
    "sort":      {"budget": (8.0, 0.0),   "kernels": {"sort_keys": (2.0, 0.0), "sort_radix": (6.0, 0.0), "rank_scatter": (0.0, 0.0)}},
    "receivers": {"budget": (68.0, 4.0),  "kernels": {"receivers": (68.0, 4.0)}},
    "flow":      {"budget": (16.0, 4.0),  "kernels": {"flow_tiles": (8.0, 0.0), "flow_climb": (8.0, 0.0), "flow_final": (8.0, 4.0)}},
    "solve":     {"budget": (45.0, 0.0),  "kernels": {"solve_setup": (33.0, 0.0), "solve_basin": (8.0, 0.0), "solve_patch": (8.0, 0.0), "solve_round": (8.0, 0.0),
                                                       "solve_final": (4.0, 0.0), "basin_layout": (0.0, 0.0), "basin_sort": (0.0, 0.0)}},
    "thermal":   {"budget": (106.0, 4.0), "kernels": {"thermal_excess": (45.0, 0.0), "thermal_apply": (61.0, 4.0)}},
}
assert sum(v["budget"][0] for v in PASSES.values()) == 243.0 and sum(v["budget"][1] for v in PASSES.values()) == 12.0
ALTERNATIVES = ("solve_patch", "solve_round", "flow_climb")      # alternatives of solve_basin / flow_tiles: one of them runs in a pass
for _p in PASSES.values():
    assert (sum(v[0] for k, v in _p["kernels"].items() if k not in ALTERNATIVES), max(v[1] for v in _p["kernels"].values())) == _p["budget"], _p
# one-off stages and the glacial iterations: bytes per launch of the whole stage
ONE_OFF = {"soil_creep": (82.0 * 0.8, 4.0), "warp_terrain": (0.0, 20.0 + 100.0 * 24)}      # creep: 82 B per interior-land cell; warp: ~20 + 100 x hops B/cell, ~24 hops at 10M
GLACIAL_KERNELS = ("glac_index", "ice_receivers", "ice_round", "carve_setup", "carve_round", "moraine_fjord", "glacial_blend")      # ice_round: k_ice_climb, carve_round: k_carve_granules (one launch each per glacial step)
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE per kernel on this workload at the full iteration count (separate passes;
# profiles/collect_pmc.sh); used for roofline.traffic of the dominant kernel unless WO_BENCH_PMC=1 asks for a live collection.
PMC_FILE = REPO / "profiles" / "r06_pmc_fetch_write_per_kernel_10m_200iters.json"
FAMILY_KERNEL = {"solve_round": "wo::k_affine_jump", "solve_setup": ("void wo::k_solve_setup_batched<true>",), "sort_radix": ("wo::k_rs_scatter", "wo::k_rs_count"),
                 "thermal_apply": "void wo::k_thermal_apply_reg<16>", "solve_patch": "wo::k_solve_patch", "thermal_excess": "wo::k_thermal_excess", "receivers": "wo::k_receivers_flow_init",
                 "flow_final": "wo::k_flow_final", "flow_climb": "wo::k_flow_climb", "flow_tiles": ("void wo::k_flow_tiles", "wo::k_flow_root_climb", "wo::k_flow_root_links"), "carve_round": "wo::k_carve_granules",
                 "warp_terrain": "wo::k_warp", "soil_creep": "wo::k_creep", "solve_final": "wo::k_solve_final", "solve_basin": ("void wo::k_solve_flowing",)}
KERNEL_GROUPS = {"basin_sort": "sort_radix"}      # profile families that are call sites of the same HIP kernels
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)

PARAMS = dict(hIters=200, K=3e-4, m=0.5, dt=1.0, tIters=200, talusSlope=1.16, kThermal=0.015, gIters=10, glacialStrength=0.5)
WARP, CREEP = 0.75, (3, 0.1125)


# ---- multi-rank plumbing (one process per GPU; no collective on the data path) ---------------------------
def dist_env():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def dist_init(world: int, local_rank: int, backend: str = "nccl"):
    """Returns the torch.distributed module (initialised) or None for a single process."""
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend)
    return dist


def dist_max(dist, value: float, device: str) -> float:
    """MAX over ranks of a host scalar (the timed region's wall time)."""
    if dist is None:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def seed_for_rank(rank: int) -> int:
    return 1 + rank          # BASELINE config 5: one planet per GPU, seeds 1..


def whole_job_value(cells: int, iters: int, steps: int, world: int, wall_s: float) -> float:
    """Mcells·iter/s aggregated over all ranks (every rank processes `steps` planets of `cells` cells)."""
    return cells * iters * steps * world / wall_s / 1e6


def bind_to_gpu_numa_node(local_rank: int):
    """Pin this rank (and the threads it will create) to the CPUs of the NUMA node its GPU hangs off: the host-resident
    flood stage is memory-latency bound and loses 20-30 % when the OS places it on the far socket."""
    try:
        import torch
        node = os.environ.get("WO_BENCH_NUMA_NODE")
        if node is None:
            pr = torch.cuda.get_device_properties(local_rank)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            node = Path(f"/sys/bus/pci/devices/{bdf}/numa_node").read_text().strip()
        if int(node) < 0:
            return None
        cpus = set()
        for part in Path(f"/sys/devices/system/node/node{int(node)}/cpulist").read_text().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, cpus)
        return int(node)
    except Exception:
        return None


def build_inputs(cells: int, seed: int):
    from planet_heightmap_generation_amd import sphere_mesh as S
    t0 = time.time()
    mesh, xyz, nd = S.build_sphere(cells, 0.75, seed)
    if os.environ.get("WO_BENCH_LAYOUT") == "morton":
        mesh, xyz, nd = morton_relabel(mesh, xyz, nd)
    return mesh, xyz, nd, time.time() - t0


def morton_relabel(mesh, xyz, nd):
    """EXPERIMENT ONLY (WO_BENCH_LAYOUT=morton): the same planet with its cells renumbered in Morton order of their positions
    (rows keep their order).  Cell ids enter the reference's semantics (cellNoise, initial land order), so the field differs from
    the oracle's and parity_crc_ok is false by construction; the run only shows what the index-order kernels cost when
    neighbours are near each other in memory."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    N = mesh.numRegions
    p = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)

    def spread(v):
        v = v.astype(np.uint64) & 0x1fffff
        v = (v | (v << 32)) & 0x1f00000000ffff
        v = (v | (v << 16)) & 0x1f0000ff0000ff
        v = (v | (v << 8)) & 0x100f00f00f00f00f
        v = (v | (v << 4)) & 0x10c30c30c30c30c3
        v = (v | (v << 2)) & 0x1249249249249249
        return v
    q = np.clip(((p.astype(np.float64) + 1.0) * 1048575.5).astype(np.int64), 0, 2097151)
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    perm = np.argsort(key, kind="stable").astype(np.int64)          # new -> old
    inv = np.empty(N, dtype=np.int32); inv[perm] = np.arange(N, dtype=np.int32)
    off = mesh.adjOffset.astype(np.int64)
    deg = (off[1:] - off[:-1])[perm]
    noff = np.zeros(N + 1, dtype=np.int64); np.cumsum(deg, out=noff[1:])
    src = np.repeat(off[:-1][perm] - noff[:-1], deg) + np.arange(noff[-1], dtype=np.int64)
    m2 = S.SphereMesh(numRegions=N, triangles=mesh.triangles, halfedges=mesh.halfedges, adjOffset=noff.astype(np.int32),
                      adjList=inv[mesh.adjList[src]].astype(np.int32), adjTriList=mesh.adjTriList[src])
    return m2, np.ascontiguousarray(p[perm].reshape(-1)), np.ascontiguousarray(np.asarray(nd)[src])


def one_step(pl, seed, params, iters_scale=None):
    p = dict(params)
    pl.restore_state()
    pl.warp_terrain_resident(seed, WARP)
    pl.ocean_from_elevation()
    pl.erode_composite_resident(p["hIters"], p["K"], p["m"], p["dt"], p["tIters"], p["talusSlope"], p["kThermal"], p["gIters"],
                                p["glacialStrength"])
    pl.apply_soil_creep_resident(*CREEP)


class Decomposition:
    """Per-rank state of the landmass decomposition: the plan is a function of the ocean mask (recomputed when it changes)."""

    def __init__(self, mesh, pl, rank, world, dist, device, comm=None):
        self.mesh, self.pl, self.rank, self.world, self.dist, self.device, self.comm = mesh, pl, rank, world, dist, device, comm
        self.plan, self.mask, self.true_oc, self.link, self.exchange_ms, self.plan_ms = None, None, None, None, 0.0, 0.0

    def apply_mask(self):
        from planet_heightmap_generation_amd import decomposed as D
        oc = self.pl.download_ocean()
        if self.true_oc is None or not np.array_equal(oc, self.true_oc):
            t0 = time.perf_counter()
            self.plan = D.plan_landmasses(self.mesh, oc, self.world)
            self.mask = self.plan.rank_mask(self.rank, oc)
            self.true_oc = oc
            self.link = D.ResidentLandmass(self.plan, self.rank, self.pl)
            # equal flood keys that matter (every flood call at 40 M cells): the ranks pool their heights and the undecided rank
            # floods the whole planet (include/worogen.h: wo_planet_set_flood_exchange)
            if self.comm is not None:
                self.pl.set_flood_exchange(oc, comm=self.comm, counts=[int(c.size) for c in self.plan.cells], cells_by_rank=np.concatenate(self.plan.cells))
            else:
                self.pl.set_flood_exchange(oc, D.TorchFloodExchange(self.plan, self.rank, self.dist, self.device))
            self.plan_ms = (time.perf_counter() - t0) * 1e3
        self.pl.upload(None, self.mask)

    def merge(self):
        t0 = time.perf_counter()
        self.link.exchange(self.dist, self.device, self.comm)
        self.pl.sync()
        self.exchange_ms = (time.perf_counter() - t0) * 1e3

    def summary(self):
        L = max(1, int(self.plan.load.sum()))
        return dict(landmasses=self.plan.num_landmasses, largest_landmass_fraction=round(self.plan.largest / L, 4),
                    land_cells_per_rank=[int(v) for v in self.plan.load], speedup_bound_by_cell_count=round(L / max(1, int(self.plan.load.max())), 2),
                    plan_ms_when_mask_changes=round(self.plan_ms, 1), exchange_ms_last_step=round(self.exchange_ms, 2),
                    exchange="all-gather of 4 B per land cell per step (RCCL); no exchange inside the iteration loop",
                    exchange_path="C ABI (wo_planet_exchange_allgather: pack, ncclAllGather, unpack on the planet's stream)" if self.comm is not None else "torch.distributed")


def one_step_decomposed(pl, seed, params, dec: Decomposition):
    p = dict(params)
    pl.restore_state()
    pl.warp_terrain_resident(seed, WARP)
    pl.ocean_from_elevation()
    dec.apply_mask()
    pl.erode_composite_resident(p["hIters"], p["K"], p["m"], p["dt"], p["tIters"], p["talusSlope"], p["kThermal"], p["gIters"],
                                p["glacialStrength"])
    pl.apply_soil_creep_resident(*CREEP)
    dec.merge()


class VirtualShares:
    """--mode decomposed on one GPU: the S-rank plan executed share by share (same masks, same merge rule)."""

    def __init__(self, mesh, pl, shares):
        self.mesh, self.pl, self.S = mesh, pl, shares
        self.plan, self.true_oc, self.share_ms, self.prep_ms, self.merge_ms, self.plan_ms = None, None, [], 0.0, 0.0, 0.0

    def step(self, seed, params):
        from planet_heightmap_generation_amd import decomposed as D
        pl, p = self.pl, dict(params)
        t0 = time.perf_counter()
        pl.restore_state()
        pl.warp_terrain_resident(seed, WARP)
        pl.ocean_from_elevation()
        oc = pl.download_ocean()
        merged = pl.download()
        if self.true_oc is None or not np.array_equal(oc, self.true_oc):
            tp = time.perf_counter()
            self.plan, self.true_oc = D.plan_landmasses(self.mesh, oc, self.S), oc
            self.plan_ms = (time.perf_counter() - tp) * 1e3
        warped = merged.copy()
        self.prep_ms = (time.perf_counter() - t0) * 1e3
        self.share_ms = []
        tm = 0.0
        for k in range(self.S):
            pl.upload(warped, self.plan.rank_mask(k, oc))
            pl.sync()
            t1 = time.perf_counter()
            pl.erode_composite_resident(p["hIters"], p["K"], p["m"], p["dt"], p["tIters"], p["talusSlope"], p["kThermal"], p["gIters"],
                                        p["glacialStrength"])
            pl.apply_soil_creep_resident(*CREEP)
            pl.sync()
            self.share_ms.append((time.perf_counter() - t1) * 1e3)
            t2 = time.perf_counter()
            part = pl.download()
            merged[self.plan.cells[k]] = part[self.plan.cells[k]]
            tm += time.perf_counter() - t2
        pl.upload(merged, oc)
        pl.sync()
        self.merge_ms = tm * 1e3

    def summary(self, unpartitioned_ms):
        L = max(1, int(self.plan.load.sum()))
        worst = max(self.share_ms)
        return dict(shares=self.S, landmasses=self.plan.num_landmasses, largest_landmass_fraction=round(self.plan.largest / L, 4),
                    land_cells_per_share=[int(v) for v in self.plan.load], speedup_bound_by_cell_count=round(L / max(1, int(self.plan.load.max())), 2),
                    share_ms_last_step=[round(v, 1) for v in self.share_ms], unpartitioned_stack_ms=round(unpartitioned_ms, 1),
                    projected_speedup_one_gpu_per_share=round(unpartitioned_ms / worst, 2),
                    hand_off_rounds_per_iteration=0, exchange="one all-gather of 4 B per land cell per step; nothing inside the iteration loop",
                    plan_ms_when_mask_changes=round(self.plan_ms, 1), host_merge_ms_last_step=round(self.merge_ms, 1),
                    note="erodeComposite + creep per share, measured one after the other on ONE GPU, each including the mask-dependent tables a rank of a real run builds once (~0.1 s at 10 M cells); `value` is the sequential total "
                         "(what this one GPU did), the projection is unpartitioned / slowest share")


def ensemble_in_flight(TP, mesh, xyz, nd, seed, params, B: int, device: int, expect_crc=None):
    """Supplementary figure, NOT `value`: B independent planets in flight on one GPU (one host thread, context and
    stream per planet; BASELINE config 5 runs 8 planets per GPU).  The dependency-bound kernels of one planet leave
    most of the chip idle and its host flood leaves the GPU idle, so planets overlap; `value` above stays the
    one-planet-at-a-time rate.  One untimed step per planet (mask-dependent tables), then one timed step each."""
    import threading
    planets = []
    for _ in range(B):
        q = TP.Planet(mesh, xyz, nd, ctx=TP.Context(device))
        q.synthetic_terrain(seed)
        q.save_state()
        one_step(q, seed, params)
        q.sync()
        planets.append(q)

    def work(q):
        one_step(q, seed, params)
        q.sync()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(q,)) for q in planets]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    import zlib
    crcs = sorted({int(zlib.crc32(q.download().tobytes())) for q in planets})      # same seed: every planet must end on the same field
    for q in planets:
        q.close()
    iters = max(params["hIters"], params["tIters"], params["gIters"])
    return dict(planets_in_flight=B, value=mesh.numRegions * iters * B / wall / 1e6, unit="Mcells·iter/s", wall_ms=wall * 1e3,
                ms_per_planet=wall * 1e3 / B, crc32_of_the_fields=crcs,
                fields_equal_the_headline_field=(crcs == [expect_crc]) if expect_crc is not None else None,
                note="throughput of B concurrent independent planets on this GPU; not the headline value")


def cpu_baseline(mesh, xyz, nd, seed, budget_iters: int, iters: int, pl=None):
    """CPU oracle (oracle/*.c, single thread) on the same planet and stack, bounded (~30 s of CPU work).  The one-off stages
    (warp, creep) are timed on their own.  The per-iteration cost drifts over a run as the terrain settles (round 2's
    extrapolation from the first 4 iterations was off by 16 %), so erodeComposite is sampled at two points: 2 iterations from
    the warped field (with the glacial step) and 2 iterations from the field at 3/4 of the run — where the reference runs its
    second flood — produced by the HIP path (`pl`), which is bit-identical to the oracle.  A sample call of 2 iterations
    runs exactly one flood (the mid-run one needs iteration >= round(0.75 x 2) = 2); the flood is timed on the same state on
    its own and subtracted, and those two flood times are the two floods of the full run.  value = cells x iters / (warp +
    flood at 0 + flood at 3/4 + the per-iteration cost interpolated between the samples, summed over the iterations, + creep)."""
    from oracle import pyoracle as O
    om = O.Mesh(mesh.adjOffset, mesh.adjList)
    e0 = O.synthetic_terrain(xyz, seed)
    t0 = time.time(); e = O.warp_terrain(om, e0, xyz, seed, WARP); t_warp = time.time() - t0
    oc = (e <= 0).astype(np.uint8)
    P = PARAMS
    points = [0] + ([int(round(iters * 0.75))] if (pl is not None and iters >= 40) else [])
    costs, floods, e_last, t_cpu = [], [], e, t_warp
    for at in points:
        if at == 0:
            state = e
        else:                                   # the HIP path's field after `at` iterations of this stack (bit-identical to the oracle's)
            pl.restore_state(); pl.warp_terrain_resident(seed, WARP); pl.ocean_from_elevation()
            pl.erode_composite_resident(at, P["K"], P["m"], P["dt"], at, P["talusSlope"], P["kThermal"], min(P["gIters"], at), P["glacialStrength"])
            state = pl.download()
        t0 = time.time(); O.priority_flood_carve(om, state, oc, 0.5 if at == 0 else 0.85); t_flood = time.time() - t0
        t0 = time.time()
        e_last = O.erode_composite(om, state, xyz, oc, 2, P["K"], P["m"], P["dt"], 2, P["talusSlope"], P["kThermal"], 2 if at == 0 else 0, P["glacialStrength"], nd)
        t_call = time.time() - t0
        floods.append(t_flood); costs.append(max(1e-9, (t_call - t_flood) / 2)); t_cpu += t_flood + t_call
    t0 = time.time(); O.soil_creep(om, e_last, oc, *CREEP); t_creep = time.time() - t0
    t_cpu += t_creep
    per_iter_sum = float(np.interp(np.arange(iters), points, costs).sum()) if len(points) > 1 else costs[0] * iters
    est = t_warp + floods[0] + floods[-1] + per_iter_sum + t_creep
    return dict(value=mesh.numRegions * iters / est / 1e6, unit="Mcells·iter/s", cores=1, kind="port",
                seconds=dict(warp=round(t_warp, 2), flood_at={str(a): round(f, 2) for a, f in zip(points, floods)}, creep=round(t_creep, 2),
                             per_iteration_at={str(a): round(c, 3) for a, c in zip(points, costs)}, estimated_full_workload=round(est, 1)),
                sample=f"same {mesh.numRegions}-cell planet and stack on one host core, {t_cpu:.1f} s of CPU work: warp and creep timed on their own; at iteration "
                       f"{' and '.join(str(a) for a in points)} of the run: the priority flood, and erodeComposite for 2 composite iterations (the first sample with the glacial "
                       f"step; the later state produced by the HIP path, bit-identical to the oracle's); value = cells x {iters} / (warp + the two floods + the per-iteration "
                       f"cost interpolated between the samples, summed over {iters} iterations, + creep).  The complete {iters}-iteration oracle run, made once in the build "
                       f"container, is recorded in tests/golden/crc_config3.json (oracle_seconds).")


def collect_pmc_live(args):
    """WO_BENCH_PMC=1: FETCH_SIZE and WRITE_SIZE per kernel from two child runs of this script under rocprofv3 --pmc (the
    counters do not fit one pass; counter runs use no trace domains).  Returns {kernel: {COUNTER_KB: {launches, total, per_launch}}}."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    if shutil.which("rocprofv3") is None:
        return None
    out = defaultdict(dict)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="wo_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, str(REPO / "bench.py"), "--no-cpu", "--no-profile",
               "--in-flight", "0", "--steps", "1", "--warmup", "0", "--cells", str(args.cells), "--iters", str(args.iters)]
        env = dict(os.environ); env.pop("WO_BENCH_PMC", None)
        subprocess.run(cmd, check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, cwd=os.environ.get("TMPDIR", "/tmp"), timeout=1800)
        acc = defaultdict(lambda: [0, 0.0])
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == counter:
                    a = acc[row["Kernel_Name"].split("(")[0] if not row["Kernel_Name"].startswith("wo::(anonymous") else row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]]
                    a[0] += 1; a[1] += float(row["Counter_Value"])
        for name, (n, tot) in acc.items():
            out[name][counter + "_KB"] = {"launches": n, "total": tot, "per_launch": tot / max(1, n)}
        shutil.rmtree(d, ignore_errors=True)
    return dict(out) or None


def d2d_bandwidth_GBs(device: int) -> float:
    """Measured device-to-device copy rate (read + write bytes per second) next to the 8 TB/s spec."""
    import torch
    n = 256 * 1024 * 1024
    a = torch.empty(n, dtype=torch.float32, device=f"cuda:{device}"); b = torch.empty_like(a)
    b.copy_(a); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    del a, b
    return 2 * n * 4 / (ms / 1e3) / 1e9


def host_thread_usage():
    """Host threads the product itself uses (the `cores` of cpu_baseline is the oracle's, not these)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ht = int(os.environ.get("WO_HOST_THREADS", min(n, 64)))
    fl = int(os.environ.get("WO_FLOOD_THREADS", 24))
    serial = os.environ.get("WO_FLOOD_HOST") == "serial"
    return dict(available=n, mesh_builder_and_static_tables=ht, flood_workers=1 if serial else min(ht, fl), flood_pass2_pass3_tree_workers=min(ht, fl),
                note="inside the timed step the host runs only the priority flood (two calls per step): one heap per landmass, landmasses and their "
                     "drainage trees dealt to the flood workers; landmasses where equal keys matter are decided by a single-threaded replay of the "
                     "reference's heap (erode_stats.flood_host_replays: 0 at 10 M cells; flood_host_serial_pass1, round 2's serial walk, stays 0)")


def parity_crc(pl, cells: int, iters: int, seed: int = 1):
    """CRC32 of the field the timed steps produced against the oracle's for the same workload (tests/golden/crc_config3.json:
    any entry with the same cells / seed / iterations)."""
    import zlib
    f = REPO / "tests" / "golden" / "crc_config3.json"
    field = pl if isinstance(pl, np.ndarray) else pl.download()
    out = dict(crc32=int(zlib.crc32(np.ascontiguousarray(field, np.float32).tobytes())), parity_crc_ok=None)
    if f.exists():
        for gold in json.loads(f.read_text()).values():
            if gold["cells"] == cells and gold["iterations"] == iters and gold["seed"] == seed:
                out["parity_crc_ok"] = out["crc32"] == gold["crc32"]
                out["oracle_crc32"] = gold["crc32"]
                break
    return out


def with_transfers_ms(pl, seed, params):
    """the same step through the JS call surface's entry points, which take and return HOST arrays (every call
    copies r_elevation in and out and r_isOcean in: 2 x 4 B + 1 B per cell per call over PCIe).  One step, wall time in ms."""
    pl.restore_state()
    e = pl.download()
    pl.sync()
    t0 = time.perf_counter()
    pl.warp_terrain(e, seed, WARP)
    oc = (e <= 0).astype(np.uint8)
    pl.erode_composite(e, oc, params["hIters"], params["K"], params["m"], params["dt"], params["tIters"], params["talusSlope"], params["kThermal"],
                       params["gIters"], params["glacialStrength"])
    pl.apply_soil_creep(e, oc, *CREEP)
    return (time.perf_counter() - t0) * 1e3, e


def relaxed_mode_leg(pl, seed, params, exact_field, cells, ks=(4, 32, 1000000)):
    """SURVEY 7.3's question, measured: what does giving up the reference's visiting order buy?  WO_RELAXED_SORT_EVERY=K re-sorts
    landCells only every K-th iteration (K = 1000000: once); every pass then runs with a stale but consistent order — the
    reference's algorithm, not the reference's result.  One step per K on the same planet: time, and the field's distance from the
    exact field (which equals the oracle's bit for bit).  Labelled relaxed; never `value`, never parity."""
    out = []
    iters = max(params["hIters"], params["tIters"], params["gIters"])
    try:
        for k in ks:
            os.environ["WO_RELAXED_SORT_EVERY"] = str(k)
            one_step(pl, seed, params); pl.sync()               # (same tables; first relaxed step warms nothing new, kept for symmetry)
            t0 = time.perf_counter()
            one_step(pl, seed, params); pl.sync()
            ms = (time.perf_counter() - t0) * 1e3
            f = pl.download()
            d = f.astype(np.float64) - exact_field.astype(np.float64)
            out.append(dict(mode="relaxed", sort_every=k, sorts=int(pl.last_erode_stats().get("sorts", 0)), ms_per_step=round(ms, 1), value=round(cells * iters / (ms / 1e3) / 1e6, 1),
                            rms_vs_exact=float(np.sqrt((d * d).mean())), max_abs_vs_exact=float(np.abs(d).max()), cells_differing=int((f != exact_field).sum())))
    finally:
        os.environ.pop("WO_RELAXED_SORT_EVERY", None)
    # SURVEY 7.3's whole relaxed mode (WO_RELAXED=full): one sort per flood, the implicit solve as an affine recurrence composed by pointer jumping with the
    # deposition applied afterwards, the glacial carve from a snapshot (csrc/kernels_impl.h: k_affine_*, k_carve_jacobi); the flood and the thermal pass as they are
    full = None
    try:
        os.environ["WO_RELAXED"] = "full"
        one_step(pl, seed, params); pl.sync()
        t0 = time.perf_counter()
        one_step(pl, seed, params); pl.sync()
        ms = (time.perf_counter() - t0) * 1e3
        f = pl.download()
        d = f.astype(np.float64) - exact_field.astype(np.float64)
        full = dict(mode="relaxed", what="one sort per flood + affine pointer-jumping solve with deferred deposition + Jacobi glacial carve", ms_per_step=round(ms, 1),
                    value=round(cells * iters / (ms / 1e3) / 1e6, 1), rms_vs_exact=float(np.sqrt((d * d).mean())), max_abs_vs_exact=float(np.abs(d).max()),
                    cells_differing=int((f != exact_field).sum()), stage_ms={k: round(v, 2) for k, v in pl.last_stage_timing().items()})
    finally:
        os.environ.pop("WO_RELAXED", None)
    return dict(note="RELAXED MODE, not parity: landCells re-sorted every K-th iteration only (stale visiting order in between); north_star's bound is RMS < 1e-5; "
                     "everything else (flood, dependency-ordered solve / carve / thermal replay) is unchanged, so this isolates what the order itself costs.  `full`: "
                     "every order-defined pass of the iteration replaced by an order-free one (SURVEY 7.3's roofline mode)",
                runs=out, full=full)


def one_planet_leg(TP, args, rank, world, dist, local_rank):
    """North_star's multi-GPU workload (BASELINE config 4): ONE planet of --one-planet-cells cells (seed 1) eroded by all ranks —
    landmass decomposition with the flood exchange (decomposed.py), merged over RCCL — timed like the main region (barrier,
    max over ranks).  Returns the nested object of the bench line (rank 0) or None."""
    import torch
    cells, iters = args.one_planet_cells, args.one_planet_iters
    params = dict(PARAMS)
    if iters != 200:
        params.update(hIters=iters, tIters=iters, gIters=min(10, max(1, iters // 20)))
    mesh, xyz, nd, t_mesh = build_inputs(cells, 1)
    pl = TP.Planet(mesh, xyz, nd, device=local_rank)
    pl.synthetic_terrain(1)
    pl.save_state()
    pl.sync()
    comm = make_comm(TP, pl, args, rank, world, dist)
    dec = Decomposition(mesh, pl, rank, world, dist, f"cuda:{local_rank}" if args.backend == "nccl" else None, comm)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()
        pl.sync()
    # parity first, at the iteration count the build container's oracle could afford for this size (tests/golden/crc_config3.json:
    # 40 M cells x 20 iterations = 815 s of one core; 200 iterations would be ~2 h): the same partitioned path, CRC of the merged field
    parity_short = None
    if args.one_planet_parity_iters > 0:
        pi = args.one_planet_parity_iters
        pp = dict(PARAMS); pp.update(hIters=pi, tIters=pi, gIters=min(10, max(1, pi // 20)))
        one_step_decomposed(pl, 1, pp, dec)
        pl.sync()
        if rank == 0:
            parity_short = dict(iterations=pi, **parity_crc(pl, cells, pi))
        dist.barrier()
    for _ in range(max(1, args.one_planet_warmup)):
        one_step_decomposed(pl, 1, params, dec)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.one_planet_steps):
        one_step_decomposed(pl, 1, params, dec)
    barrier()
    wall = dist_max(dist, time.perf_counter() - t0, "cuda" if args.backend == "nccl" else "cpu")
    stats, stages = pl.last_erode_stats(), pl.last_stage_timing()
    busy = [None] * world
    dist.all_gather_object(busy, dict(rank=rank, land_cells=int(stats.get("land_cells", 0)), flood_stage_ms=round(stats.get("flood_stage_ms", 0.0), 1),
                                      whole_planet_floods=int(stats.get("flood_exchange_whole_planet_floods", 0)), exchange_gathers=int(stats.get("flood_exchange_gathers", 0)),
                                      floods_received=int(stats.get("flood_exchange_received", 0)),
                                      erode_ms=round(sum(stages.values()), 1)))
    out = None
    if rank == 0:
        N = mesh.numRegions
        out = dict(metric="Mcells·iter/s, terrain-post erosion stack, ONE planet over all GPUs", value=whole_job_value(N, iters, args.one_planet_steps, 1, wall), unit="Mcells·iter/s",
                   n_gpus=world, steps=args.one_planet_steps, ms_per_step=wall * 1e3 / args.one_planet_steps, scaling="strong",
                   config=dict(workload=f"BASELINE config 4: {N} cells (Fibonacci sphere {cells}+pole, jitter 0.75, seed 1), warp 0.75 + erodeComposite(h={params['hIters']},"
                                        f"t={params['tIters']},g={params['gIters']}) + creep x3, ONE planet over {world} GPUs", cells=N, iterations=iters,
                               parallelism=f"landmass decomposition x{world} with the flood exchange (flag all-reduce per flood call; heights pooled and the whole planet flooded "
                                           f"on the undecided rank when equal keys matter), one all-gather of the land elevations per step"),
                   parity=parity_crc(pl, cells, iters), parity_at_the_oracles_iteration_count=parity_short, decomposition=dec.summary(), per_rank=busy, mesh_build_s=round(t_mesh, 1),
                   note="strong scaling of one planet in exact mode is bounded by the largest landmass and by the flood (DESIGN.md section 7); the ensemble line above is how the path's units of work shard")
    if comm is not None:
        comm.close()
    pl.close()
    return out


def make_comm(TP, pl, args, rank, world, dist):
    """the exchange behind the C ABI: rank 0's RCCL id goes round through torch.distributed, every rank joins with it (None: torch.distributed carries the exchange)"""
    if args.backend != "nccl" or os.environ.get("WO_BENCH_TORCH_EXCHANGE") == "1":
        return None
    try:
        box = [TP.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = TP.Comm(pl.ctx, box[0], world, rank)
    except Exception as ex:          # never lose the run over the communicator: torch.distributed carries the exchange then
        print(f"[bench] rank {rank}: wo_comm unavailable ({ex}); exchanging through torch.distributed", file=sys.stderr)
        comm = None
    gathered = [None] * world
    dist.all_gather_object(gathered, comm is not None)
    if not all(gathered):
        if comm is not None:
            comm.close()
        comm = None
    return comm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=10_000_000)
    ap.add_argument("--iters", type=int, default=200, help="composite iterations per step (200 = BASELINE config)")
    ap.add_argument("--cpu-iters", type=int, default=4)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-relaxed", action="store_true", help="skip the relaxed-mode measurement (N = 1)")
    ap.add_argument("--no-transfers", action="store_true", help="skip the step through the host-array entry points (value_with_transfers)")
    ap.add_argument("--timed-only", action="store_true", help="nothing but warm-up and the timed steps (what a rocprofv3 trace of the timed region wants): "
                                                               "= --no-cpu --no-profile --no-relaxed --no-transfers --in-flight 0")
    ap.add_argument("--in-flight", type=int, default=6, help="planets in flight for the supplementary ensemble figure (0 = skip)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="gloo + --share-gpu: rehearse the multi-rank path on a one-GPU box (not a measurement)")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use device 0")
    ap.add_argument("--mode", choices=("auto", "decomposed", "ensemble"), default="auto",
                    help="N > 1: 'ensemble' = one planet per GPU (weak scaling, BASELINE config 5, default), 'decomposed' = one planet over all GPUs by landmass (strong); "
                         "N = 1 with 'decomposed': the partitioned code path, --shares landmass shares one after the other")
    ap.add_argument("--seeds-per-rank", type=int, default=0, help="N > 1, ensemble: distinct planets (seeds 1 + rank + k x N) a rank cycles through, one per step; "
                                                                   "0 = min(steps, 64 / N) (BASELINE config 5: 64 seeds per job), 1 = the same planet every step")
    ap.add_argument("--planets-in-flight", type=int, default=1, help="N > 1, ensemble: planets a rank keeps in flight (one host thread, context and stream each; the rank's flood / host "
                                                                      "threads are divided among them).  A config-5 planet's host flood (median 349 ms of its step over the 64 seeds) can hide "
                                                                      "behind other planets' device work — 1.75x with 3, 1.99x with 6 warm planets in ONE process on one GPU (profiles/r06s_*) — "
                                                                      "but with new terrains and ranks that share cores it was 4x SLOWER in the two-rank rehearsal (profiles/r06t_*): default 1")
    ap.add_argument("--shares", type=int, default=8, help="N = 1, --mode decomposed: number of landmass shares")
    ap.add_argument("--one-planet-cells", type=int, default=40_000_000, help="N > 1, default mode: after the ensemble region, ONE planet of this many cells over all GPUs "
                                                                              "(BASELINE config 4; nested object `one_planet` of the line); 0 = skip")
    ap.add_argument("--one-planet-iters", type=int, default=200)
    ap.add_argument("--one-planet-timeout", type=float, default=900.0, help="seconds after which the one-planet leg is given up and the line is printed without it")
    ap.add_argument("--one-planet-steps", type=int, default=1)
    ap.add_argument("--one-planet-parity-iters", type=int, default=20, help="one extra untimed step at this iteration count, whose CRC the committed oracle checksum covers (0 = skip)")
    ap.add_argument("--one-planet-warmup", type=int, default=1)
    args = ap.parse_args()
    if args.timed_only:
        args.no_cpu = args.no_profile = args.no_relaxed = args.no_transfers = True
        args.in_flight = 0

    rank, local_rank, world = dist_env()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    if args.share_gpu:
        local_rank = 0
    dist = dist_init(world, local_rank, args.backend)
    torch.cuda.set_device(local_rank)
    numa_node = bind_to_gpu_numa_node(local_rank)

    if world > 1:
        # several ranks share the host: the flood's workers (24 by default) and the mesh builder's threads (64) are scaled to this
        # rank's share of the cores it is pinned to (8 ranks x 24 flood workers + 8 x 64 builder threads would oversubscribe a node)
        try:
            mine = len(os.sched_getaffinity(0))
            nodes = max(1, len(list(Path("/sys/devices/system/node").glob("node[0-9]*")))) if numa_node is not None else 1
            per_rank = max(4, mine * nodes // world)
            per_planet = max(4, per_rank // max(1, args.planets_in_flight))          # planets in flight share the rank's cores
            os.environ.setdefault("WO_FLOOD_THREADS", str(min(24, per_planet)))
            os.environ.setdefault("WO_HOST_THREADS", str(min(64, per_planet)))
        except Exception:
            pass
    from planet_heightmap_generation_amd import terrain_post as TP
    params = dict(PARAMS)
    if args.iters != 200:
        params.update(hIters=args.iters, tIters=args.iters, gIters=min(10, max(1, args.iters // 20)))
    decomposed_mode = world > 1 and args.mode == "decomposed"
    seed = 1 if decomposed_mode else seed_for_rank(rank)
    mesh, xyz, nd, t_mesh = build_inputs(args.cells, seed)
    pl = TP.Planet(mesh, xyz, nd, device=local_rank)
    pl.synthetic_terrain(seed)
    pl.save_state()
    pl.sync()
    N = mesh.numRegions

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        pl.sync()

    comm = make_comm(TP, pl, args, rank, world, dist) if decomposed_mode else None
    dec = Decomposition(mesh, pl, rank, world, dist, f"cuda:{local_rank}" if args.backend == "nccl" else None, comm) if decomposed_mode else None
    step = (lambda: one_step_decomposed(pl, seed, params, dec)) if decomposed_mode else (lambda: one_step(pl, seed, params))
    virt = VirtualShares(mesh, pl, max(1, args.shares)) if (world == 1 and args.mode == "decomposed") else None
    unpart_ms = None
    if virt:
        one_step(pl, seed, params); pl.sync()                 # tables for the true mask, then the unpartitioned stack for reference
        pl.restore_state(); pl.warp_terrain_resident(seed, WARP); pl.ocean_from_elevation(); pl.sync()
        tu = time.perf_counter()
        pl.erode_composite_resident(params["hIters"], params["K"], params["m"], params["dt"], params["tIters"], params["talusSlope"], params["kThermal"],
                                    params["gIters"], params["glacialStrength"])
        pl.apply_soil_creep_resident(*CREEP); pl.sync()
        unpart_ms = (time.perf_counter() - tu) * 1e3
        step = lambda: virt.step(seed, params)
    # BASELINE config 5 (N > 1, ensemble): "10 M cells x 64 random seeds, one planet per GPU" — rank r erodes the planets of seeds
    # 1 + r + k x N (k = 0, 1, ...: round-robin over the ranks, 64 seeds per job at most), one planet per timed step, cycling when the
    # job has more steps than seeds.  Every planet is built and uploaded BEFORE the timed region (the one-time mesh upload SURVEY 8(d)
    # excludes) and warmed on another terrain; inside the region a step is the resident stack on the next planet of the cycle, its
    # terrain-dependent tables included the first time round.
    fleet, fleet_seeds = [pl], [seed]
    if world > 1 and not decomposed_mode and args.seeds_per_rank != 1:
        want = args.seeds_per_rank if args.seeds_per_rank > 0 else max(1, min(args.steps, 64 // world))
        for k in range(1, want):
            sk = 1 + rank + k * world
            mk, xk, nk, tk = build_inputs(args.cells, sk)
            q = TP.Planet(mk, xk, nk, ctx=pl.ctx if args.planets_in_flight <= 1 else TP.Context(local_rank))
            fleet.append(q); fleet_seeds.append(sk)
            t_mesh += tk
            del mk, xk, nk
        turn = [0]

        def step():
            i = turn[0] % len(fleet); turn[0] += 1
            one_step(fleet[i], fleet_seeds[i], params)
    cold_ms = None
    if len(fleet) > 1:
        # Config 5's planets are NEW TERRAINS inside the timed region: what is excluded is what SURVEY 8(d) excludes — the mesh and its
        # upload, and with it the planet's scratch allocations — so every planet is warmed on ANOTHER terrain (seed + 1000: other mask, other tables), then
        # given its own; its first timed step builds the mask-dependent tables (land-first mirror, flood tables, land lists) like any new planet's would.
        for i, (q, sk) in enumerate(zip(fleet, fleet_seeds)):
            q.synthetic_terrain(sk + 1000); q.save_state()
            tc = time.perf_counter()
            one_step(q, sk + 1000, params); q.sync()
            if i == 0:
                cold_ms = (time.perf_counter() - tc) * 1e3
            q.synthetic_terrain(sk); q.save_state(); q.sync()
    else:
        for w in range(args.warmup):
            tc = time.perf_counter()
            step()
            if w == 0:
                pl.sync()
                cold_ms = (time.perf_counter() - tc) * 1e3       # first step: mask-dependent tables, scratch allocation, launch-count prediction
    in_flight = min(max(1, args.planets_in_flight), len(fleet)) if len(fleet) > 1 else 1
    barrier()
    t0 = time.perf_counter()
    pl.timer_start()
    if in_flight > 1:
        # K planets in flight: worker j owns the planets j, j + K, ... of the rank's cycle and runs their turns of the job's `steps` planet-steps in order
        # (no planet is ever in two hands); the step count and what a step is are unchanged, only the host no longer waits for one planet's flood
        # before it feeds the device with the next planet's work
        import threading
        turns = [[t % len(fleet) for t in range(args.steps) if (t % len(fleet)) % in_flight == j] for j in range(in_flight)]
        failures = []

        def worker(mine):
            try:
                for i in mine:
                    one_step(fleet[i], fleet_seeds[i], params)
                    fleet[i].sync()
            except BaseException as ex:      # noqa: BLE001 — re-raised below, in the rank's main thread
                failures.append(ex)
        th = [threading.Thread(target=worker, args=(mine,)) for mine in turns if mine]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if failures:
            raise failures[0]
    else:
        for _ in range(args.steps):
            step()
    ev_ms = pl.timer_stop_ms()
    barrier()
    wall = time.perf_counter() - t0
    wall = dist_max(dist, wall, "cuda" if args.backend == "nccl" else "cpu")
    stats = pl.last_erode_stats()
    stages = pl.last_stage_timing()
    crc = parity_crc(pl, args.cells, max(params["hIters"], params["tIters"], params["gIters"])) if (rank == 0 and seed == 1) else None
    # config 5: which seeds this job ran, and for each the CRC of its field against the oracle's where tests/golden/crc_config3.json has one
    seeds_run = None
    if world > 1 and not decomposed_mode:
        it_ = max(params["hIters"], params["tIters"], params["gIters"])
        mine = [dict(rank=rank, seed=sk, steps=len([i for i in range(args.steps) if i % len(fleet) == j]), **parity_crc(q, args.cells, it_, sk))
                for j, (q, sk) in enumerate(zip(fleet, fleet_seeds))]
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        seeds_run = sorted((e for lst in allr for e in lst), key=lambda e: e["seed"])
    for q in fleet[1:]:
        own_ctx = q.ctx is not pl.ctx
        q.close()
        if own_ctx:
            q.ctx.close()
    # What a NEW planet costs on a resident mesh: the timed steps above re-run one terrain, whose mask-dependent tables (land-first
    # mirror, flood tables, land lists, launch-count prediction) are cached after the first step.  Three other terrains on the same mesh, one step each,
    # with those tables rebuilt inside the clock; then the benched terrain again (untimed) for the legs below.
    new_terrain = None
    if rank == 0 and world == 1 and not virt and not args.timed_only:
        ms_new = []
        for s2 in (seed + 101, seed + 102, seed + 103):
            pl.synthetic_terrain(s2); pl.save_state(); pl.sync()
            tn = time.perf_counter()
            one_step(pl, s2, params); pl.sync()
            ms_new.append((time.perf_counter() - tn) * 1e3)
        pl.synthetic_terrain(seed); pl.save_state()
        one_step(pl, seed, params); pl.sync()
        new_terrain = dict(ms=[round(v, 2) for v in ms_new], mean_ms=sum(ms_new) / len(ms_new), value=N * max(params["hIters"], params["tIters"], params["gIters"]) / (sum(ms_new) / len(ms_new) / 1e3) / 1e6,
                           note="one step each on three other synthetic terrains (seeds +101..+103) of the resident mesh: the terrain-dependent set-up (land-first mirror, flood tables, "
                                "land lists) is rebuilt inside the clock; `value` is the warm figure (same terrain every step), cold_first_step_ms additionally holds the planet's scratch allocations")
    L = int(stats.get("land_cells", 0))
    iters = max(params["hIters"], params["tIters"], params["gIters"])

    roofline = None
    if rank == 0 and not args.no_profile and not decomposed_mode and not virt:
        # separate pass with every launch bracketed by HIP events on the planet's stream
        pl.profile_reset()
        pl.profile_enable(True)
        one_step(pl, seed, params)
        pl.profile_enable(False)
        rep = pl.profile_report()
        # A family's time is the sum of one HIP-event pair per launch, and a pair adds ~3 us to the kernel it brackets (the library brackets nothing, one and two
        # kernels that do nothing, 64 times each, when profiling is switched on — wo_profile_enable; 2 x (one kernel) - (two kernels) is the pair's own share):
        # for launches of 5-25 us — the radix sort's — that is a seventh of the figure, and what made it disagree with a rocprofv3 kernel trace.  Every family's
        # time below is NET of launches x that share; the raw figures are kept beside them.
        cal = {k: rep.pop(k, None) for k in ("event_pair_empty", "event_pair_noop_kernel", "event_pair_two_noop_kernels")}
        cal_ms = {k: (v[0] / v[1]) if v and v[1] else None for k, v in cal.items()}
        pair_ms = 0.0
        if cal_ms["event_pair_noop_kernel"] is not None and cal_ms["event_pair_two_noop_kernels"] is not None:
            pair_ms = min(max(2 * cal_ms["event_pair_noop_kernel"] - cal_ms["event_pair_two_noop_kernels"], 0.0), cal_ms["event_pair_empty"] or 1e9)
        rep_raw = dict(rep)
        rep = {k: (max(kms - kl * pair_ms, 0.0), kl) for k, (kms, kl) in rep.items()}
        sorts = int(stats.get("sorts", iters))
        g_iters = params["gIters"]

        def passes_of(pass_name):
            return sorts if pass_name == "sort" else iters

        # per pass: budget bytes over the summed time of every kernel of the pass (overhead kernels included)
        kernel_pass = {k: pn for pn, pv in PASSES.items() for k in pv["kernels"]}
        per_pass = {}
        for pn, pv in PASSES.items():
            ms_sum = sum(rep[k][0] for k in pv["kernels"] if k in rep)
            if ms_sum <= 0:
                continue
            gb = (pv["budget"][0] * L + pv["budget"][1] * N) * passes_of(pn) / 1e9
            per_pass[pn] = dict(ms=round(ms_sum, 3), algorithmic_GB=round(gb, 3), achieved_GBs=round(gb / (ms_sum / 1e3), 1), frac=round(gb / (ms_sum / 1e3) / HBM_PEAK_GBS, 5),
                                kernels=[k for k in pv["kernels"] if k in rep])
        # glacial iterations: 86 B x land + 123 B x active carve cells + 27 B x cells per glacial iteration
        g_ms = sum(rep[k][0] for k in GLACIAL_KERNELS if k in rep)
        carve_active = float(stats.get("carve_active_total", 0.0))
        if g_ms > 0 and g_iters > 0:
            gb = ((86.0 * L + 27.0 * N) * g_iters + 123.0 * carve_active) / 1e9
            per_pass["glacial"] = dict(ms=round(g_ms, 3), algorithmic_GB=round(gb, 3), achieved_GBs=round(gb / (g_ms / 1e3), 1), frac=round(gb / (g_ms / 1e3) / HBM_PEAK_GBS, 5),
                                       kernels=[k for k in GLACIAL_KERNELS if k in rep], glacial_iterations=g_iters, active_carve_cells_total=carve_active)

        def kernel_bytes_total(k):
            """algorithmic bytes of all launches of kernel family k over the step (its share of its pass's budget), or None"""
            if k in kernel_pass:
                pl_, pc_ = PASSES[kernel_pass[k]]["kernels"][k]
                return (pl_ * L + pc_ * N) * passes_of(kernel_pass[k])
            if k in ONE_OFF:
                pl_, pc_ = ONE_OFF[k]
                return (pl_ * L + pc_ * N) * (3 if k == "soil_creep" else 1)
            if k == "carve_round":
                return 123.0 * carve_active
            return None
        fams = {}
        for k, (kms, kl) in sorted(rep.items(), key=lambda kv: -kv[1][0]):
            ent = {"ms": round(kms, 3), "launches": kl}
            kb = kernel_bytes_total(k)
            if kb is not None and kms > 0:
                ent.update(**({"pass": kernel_pass[k]} if k in kernel_pass else {}), algorithmic_GB=round(kb / 1e9, 3), achieved_GBs=round(kb / 1e9 / (kms / 1e3), 1),
                           frac=round(kb / 1e9 / (kms / 1e3) / HBM_PEAK_GBS, 5))
            fams[k] = ent
        # the dominant kernel: largest total time among the kernels that carry algorithmic bytes — by KERNEL, not by call site:
        # the in-tree radix sort (k_rs_count + k_rs_scatter) is launched by the elevation sort (family sort_radix, which carries the
        # sort pass's bytes) and by the basin layout (family basin_sort, 0 algorithmic bytes): one kernel, both call sites' time
        grouped = {}
        for k, (kms, kl) in rep.items():
            g = KERNEL_GROUPS.get(k, k)
            a = grouped.setdefault(g, [0.0, 0])
            a[0] += kms; a[1] += kl
        fam, (ms, launches) = max(((k, v) for k, v in grouped.items() if (kernel_bytes_total(k) or 0) > 0), key=lambda kv: kv[1][0])
        bytes_per_launch = kernel_bytes_total(fam) / launches
        avg_launch_s = ms / 1e3 / launches
        achieved = bytes_per_launch / avg_launch_s / 1e9
        traffic, traffic_note = None, None
        pmc_all = None
        if os.environ.get("WO_BENCH_PMC") == "1":
            pmc_all = collect_pmc_live(args)
            src = "collected by this run (rocprofv3 --pmc, one child process per counter, same workload and iteration count)"
        elif PMC_FILE.exists() and args.cells == 10_000_000 and iters == 200:
            pmc_all = json.loads(PMC_FILE.read_text())
            src = "committed file profiles/" + PMC_FILE.name + " (rocprofv3 --pmc, separate passes per counter, this workload at the full 200 iterations, one planet), not measured in this run"
        if pmc_all:
            # the counter file must describe THIS build's launches: every kernel family the profiled step ran (and that the table
            # below can name) has to be in it, else the file is stale — fail instead of quoting another build's traffic
            have = [k.replace("(anonymous namespace)::", "") for k, v in pmc_all.items() if isinstance(v, dict)]
            missing = []
            for f_, (ms_, _n) in rep.items():
                if ms_ > 0 and f_ in FAMILY_KERNEL and f_ in kernel_pass:
                    names_ = FAMILY_KERNEL[f_] if isinstance(FAMILY_KERNEL[f_], tuple) else (FAMILY_KERNEL[f_],)
                    if not any(h.startswith(names_) for h in have):
                        missing.append(f"{f_} ({' / '.join(names_)})")
            if missing and os.environ.get("WO_BENCH_ALLOW_STALE_PMC") != "1":
                raise SystemExit(f"bench.py: {PMC_FILE.name if os.environ.get('WO_BENCH_PMC') != '1' else 'the live PMC collection'} has no counters for kernels this run launched: "
                                 + ", ".join(missing) + " — re-collect with profiles/collect_pmc.sh (or WO_BENCH_PMC=1 / WO_BENCH_ALLOW_STALE_PMC=1)")
        if pmc_all and fam in FAMILY_KERNEL:
            names = FAMILY_KERNEL[fam] if isinstance(FAMILY_KERNEL[fam], tuple) else (FAMILY_KERNEL[fam],)
            hits = [v for k, v in pmc_all.items() if isinstance(v, dict) and k.replace("(anonymous namespace)::", "").startswith(names) and "FETCH_SIZE_KB" in v and "WRITE_SIZE_KB" in v]
            if hits:
                # a family of several kernels (the radix sort: count + scatter): all their bytes over all their launches, like `achieved`
                hits = hits if fam in KERNEL_GROUPS.values() else hits[:1]
                traffic = sum(v["FETCH_SIZE_KB"]["total"] + v["WRITE_SIZE_KB"]["total"] for v in hits) * 1024.0 / max(1, sum(v["FETCH_SIZE_KB"]["launches"] for v in hits))
                traffic_note = ("traffic_source: " + src + "; bytes per launch = (FETCH_SIZE + WRITE_SIZE) KB; FETCH_SIZE on gfx950 under-reports wide coalesced "
                                "reads by 2x and is uncalibrated for narrow gathers (MI355X_MICROARCH.md), so this is a lower bound")
        kn = FAMILY_KERNEL.get(fam, fam)
        kernel_name = " + ".join(kn) if (isinstance(kn, tuple) and fam == "sort_radix") else (kn[0] if isinstance(kn, tuple) else kn)
        roofline = dict(bound="hbm", kernel=kernel_name.replace("void ", ""), family=fam, call_sites=[k for k in rep if KERNEL_GROUPS.get(k, k) == fam], achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                        traffic=traffic, traffic_note=traffic_note, launches=launches, avg_launch_us=avg_launch_s * 1e6,
                        avg_launch_us_with_the_event_pair=sum(rep_raw[k][0] for k in rep_raw if KERNEL_GROUPS.get(k, k) == fam) * 1e3 / launches,
                        event_pair_us=dict(taken_off_per_launch=round(pair_ms * 1e3, 3), around_nothing=round(cal_ms["event_pair_empty"] * 1e3, 3) if cal_ms["event_pair_empty"] is not None else None,
                                           around_one_kernel_that_does_nothing=round(cal_ms["event_pair_noop_kernel"] * 1e3, 3) if cal_ms["event_pair_noop_kernel"] is not None else None,
                                           around_two_such_kernels=round(cal_ms["event_pair_two_noop_kernels"] * 1e3, 3) if cal_ms["event_pair_two_noop_kernels"] is not None else None,
                                           note="measured live, 64 pairs each on the planet's stream when profiling is switched on; taken_off_per_launch = 2 x (one kernel) - (two kernels); "
                                                "family times, avg_launch_us, achieved and frac are net of launches x that"),
                        algorithmic_bytes_per_launch=bytes_per_launch,
                        note="algorithmic bytes: the kernel's share of its pass's SURVEY 8(d) budget (the shares of a pass add up to the budget: PASSES in bench.py); "
                             "the solve walks the drainage DAG in dependency order and is bound by the latency of its chains, not by HBM (DESIGN.md section 5)",
                        passes=per_pass, families=fams,
                        whole_stack=dict(note="SURVEY 8(d): (243 B x land cells + 12 B x cells) per composite iteration over the wall time of a step",
                                         achieved_GBs=(243.0 * L + 12.0 * N) * iters / (wall / args.steps) / 1e9,
                                         frac=(243.0 * L + 12.0 * N) * iters / (wall / args.steps) / 1e9 / HBM_PEAK_GBS))
    ensemble = None
    if rank == 0 and world == 1 and args.in_flight > 1 and not virt:
        ensemble = ensemble_in_flight(TP, mesh, xyz, nd, seed, params, args.in_flight, local_rank, expect_crc=crc["crc32"] if crc else None)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu and not virt:
        cpu = cpu_baseline(mesh, xyz, nd, seed, args.cpu_iters, iters, pl)

    transfers = None
    if rank == 0 and world == 1 and not virt and not args.no_transfers:
        ms_t, field_t = with_transfers_ms(pl, seed, params)
        transfers = dict(value_with_transfers=N * iters / (ms_t / 1e3) / 1e6, ms_per_step=ms_t, crc32_equals_resident_run=(parity_crc(field_t, args.cells, iters, seed)["crc32"] == crc["crc32"]) if crc else None,
                         note="one step through the host-array entry points of the JS call surface (wo_warp_terrain, wo_erode_composite, wo_soil_creep: r_elevation H2D + D2H and r_isOcean H2D per call); `value` is the resident rate")
    relaxed = None
    if rank == 0 and world == 1 and not virt and not args.no_relaxed:
        one_step(pl, seed, params); pl.sync()
        relaxed = relaxed_mode_leg(pl, seed, params, pl.download(), N)
    # everything the line needs is worked out HERE, before the one-planet leg: its watchdog may only print (no torch / GPU call from the
    # timer thread: a collective that never gets its peers would block a device synchronisation for ever)
    d2d_GBs = round(d2d_bandwidth_GBs(local_rank), 1) if rank == 0 else None
    host_threads = host_thread_usage()

    def build_line(one_planet):
        value = whole_job_value(N, iters, args.steps, 1 if decomposed_mode else world, wall)
        out = {
            "metric": "Mcells·iter/s, terrain-post erosion stack", "value": value, "unit": "Mcells·iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "strong" if (decomposed_mode or virt) else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE config 3: {N} cells (Fibonacci sphere {args.cells}+pole, jitter 0.75, seed 1+rank), "
                                   f"warp 0.75 + erodeComposite(h={params['hIters']},t={params['tIters']},g={params['gIters']}) + creep x3, "
                                   + ("ONE planet over all GPUs" if decomposed_mode else (f"ONE planet as {virt.S} landmass shares, one after the other on one GPU" if virt else "one planet per GPU")), "cells": N, "land_cells": L, "iterations": iters,
                       "parallelism": (f"landmass decomposition x{world}: every rank erodes its share of the planet's landmasses, one all-gather of the land "
                                       f"elevations per step over RCCL (bit-identical to the unpartitioned run)") if decomposed_mode
                                      else (f"landmass decomposition, {virt.S} shares executed sequentially on one GPU (partitioned code path)" if virt
                                            else f"ensemble x{world} (no collective on the data path)")},
            "decomposition": dec.summary() if dec else (virt.summary(unpart_ms) if virt else None),
            "value_with_transfers": transfers["value_with_transfers"] if transfers else None, "with_transfers": transfers, "one_planet": one_planet, "relaxed_mode": relaxed,
            "roofline": roofline, "cpu_baseline": cpu, "ensemble_in_flight": ensemble,
            "parity": crc, "ensemble_seeds": seeds_run, "cold_first_step_ms": cold_ms, "new_terrain_step_ms": new_terrain["mean_ms"] if new_terrain else None, "new_terrain": new_terrain,
            "ensemble_steps_are_new_terrain": (len(fleet_seeds) > 1 and args.steps <= len(fleet_seeds)) if world > 1 and not decomposed_mode else None,
            "ensemble_planets_in_flight_per_rank": in_flight if world > 1 and not decomposed_mode else None, "host_threads": host_threads,
            "hbm_d2d_copy_GBs_measured": d2d_GBs,
            "stage_ms_last_step": {k: round(v, 2) for k, v in stages.items()},
            "erode_stats": stats, "mesh_build_s": round(t_mesh, 1), "host_numa_node": numa_node, "hip_event_ms_per_step": ev_ms / args.steps,
        }
        return out

    import threading
    printed = threading.Lock()          # the line is printed once: by the main thread or by the watchdog, whoever takes this first

    def emit(line):
        if printed.acquire(blocking=False):
            sys.stdout.write(line + "\n")
            sys.stdout.flush()

    one_planet = None
    if world > 1 and not decomposed_mode and args.mode == "auto" and args.one_planet_cells > 0:
        pl.close()
        # The one-planet leg exchanges between ranks inside the flood stage (a path that has never had real peers on this pool): should it not
        # come back, the ensemble measurement above must not be lost with it — after the timeout rank 0 prints the line without the leg and
        # every rank leaves.
        fallback_line = json.dumps(build_line({"error": f"the one-planet leg did not finish within {args.one_planet_timeout} s; the ensemble figures of this line are complete"})) if rank == 0 else None

        def give_up():          # timer thread: print the prepared string and leave; nothing here touches torch, HIP or the library
            if rank == 0:
                emit(fallback_line)
            sys.stderr.write(f"[bench] rank {rank}: one-planet leg timed out\n")
            sys.stderr.flush()
            os._exit(0)
        watchdog = threading.Timer(args.one_planet_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        one_planet = one_planet_leg(TP, args, rank, world, dist, local_rank)
        watchdog.cancel()
    if rank == 0:
        emit(json.dumps(build_line(one_planet)))
    pl.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
