"""Landmass decomposition: one planet's erosion stack spread over several GPUs, exactly.

Every pass of erodeComposite (js/terrain-post.js:369-707) and applySoilCreep (:758-794) couples a land cell only to
land cells it can reach over land: the priority flood grows from the coast inland (:118-147) and carves along drain
paths that end at the coast (:152-214); receivers, flow and the implicit solve + deposition follow the drainage forest,
whose trees end in ocean cells that are never written (:566-641); talus moves between land neighbours (:645-686); ice
flows downhill over land (:475-557); the sort's tie history only matters between cells that interact.  Rivers and
scree do not cross water.  So the connected components of the land cells ("landmasses") are independent problems, and
the decomposition needs NO exchange inside the iteration loop:

    rank k erodes the full mesh with   isOcean_k = isOcean  OR  (cell belongs to a landmass of another rank)

(its own cells see exactly the neighbours, ids, hash noise and ocean cells they see in the unpartitioned run; the other
landmasses are inert ocean cells to it), and after the stack the ranks exchange the elevations of their land cells
once (all-gather of 4 B per land cell: RCCL between GPUs, gloo in the CPU tests).  The result is bit-identical to the
unpartitioned run (tests/test_decomposed.py with the oracle as the per-rank engine, tests/test_gpu_parity.py with the
HIP path).  Landmasses are dealt to ranks largest first onto the least loaded rank; the speed-up is bounded by the
largest landmass (14.5 % of the land of the 10 M-cell bench planet: 6.9x at 8 ranks by cell count).  One subtlety: the
flood starts only from coasts of the OPEN ocean (the largest ocean component), so an island inside an inland sea is never
flooded; it travels with the landmass that encloses its sea (plan_landmasses), otherwise its rank would see that sea
open (found at 10 M cells: 746 cells in 4 lake islands differed by one flood EPS before this rule).

This complements banded.py (index bands with a one-ring halo per iteration), which is the decomposition of the Jacobi
passes that DO couple across water (smoothElevation over all cells, smoothField).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

from . import capi


def land_components(mesh, r_isOcean) -> np.ndarray:
    """label[r] = smallest region id of r's landmass, -1 for ocean cells (native host stage, wo_land_components)."""
    off = np.ascontiguousarray(mesh.adjOffset, np.int32)
    adj = np.ascontiguousarray(mesh.adjList, np.int32)
    oc = np.ascontiguousarray(r_isOcean, np.uint8)
    N = off.size - 1
    if oc.size != N:
        raise ValueError("r_isOcean length must equal mesh.numRegions")
    label = np.empty(N, np.int32)
    capi.check(capi.lib().wo_land_components(N, capi.ptr(off), capi.ptr(adj), capi.ptr(oc), capi.ptr(label)), "wo_land_components")
    return label


@dataclass
class LandmassPlan:
    world: int
    owner: np.ndarray            # int32 [N]: rank that erodes the cell, -1 for ocean cells
    cells: List[np.ndarray]      # cells[k]: region ids of rank k's land cells, ascending (int32)
    load: np.ndarray             # land cells per rank
    largest: int                 # cells of the largest landmass (bounds the speed-up)
    num_landmasses: int

    def rank_mask(self, rank: int, r_isOcean) -> np.ndarray:
        """isOcean as rank `rank` sees it: the true ocean plus every landmass it does not own."""
        return np.ascontiguousarray((np.asarray(r_isOcean) != 0) | (self.owner != rank), np.uint8)

    @property
    def max_cells(self) -> int:
        return int(max((c.size for c in self.cells), default=0))


def open_ocean(mesh, r_isOcean) -> np.ndarray:
    """The reference's open ocean (js/terrain-post.js:66-94): the largest connected component of the ocean cells, the one
    holding the smallest cell id among equally large ones.  Returns a bool mask."""
    oc = np.asarray(r_isOcean) != 0
    lab = land_components(mesh, (~oc).astype(np.uint8))          # components of the ocean cells (members = cells flagged 0)
    ids = lab[oc]
    if ids.size == 0:
        return np.zeros(oc.size, bool)
    labs, counts = np.unique(ids, return_counts=True)
    main = labs[np.flatnonzero(counts == counts.max())[0]]       # labels ascend: the first of the largest
    return lab == main


def plan_landmasses(mesh, r_isOcean, world: int) -> LandmassPlan:
    """Deal the landmasses to `world` ranks: largest first, each onto the least loaded rank (ties: lowest rank; equal
    sizes: lowest label first) — the same plan on every rank, no communication.

    The unit that is dealt is a landmass TOGETHER WITH the inland seas it encloses and the islands in them (a connected
    component of the cells that are not open ocean): the flood only starts from coasts of the open ocean (:118-128), so
    land inside an inland sea is never flooded — it must not see that sea become open because the enclosing landmass was
    masked away on its rank."""
    if world < 1:
        raise ValueError("world must be >= 1")
    oc = np.ascontiguousarray(r_isOcean, np.uint8)
    label = land_components(mesh, open_ocean(mesh, oc).astype(np.uint8))     # components of land + inland seas
    label = np.where(oc != 0, -1, label).astype(np.int32)                    # only land cells are owned
    land = np.flatnonzero(label >= 0).astype(np.int32)
    owner = np.full(label.size, -1, np.int32)
    load = np.zeros(world, np.int64)
    if land.size:
        labs, inv, counts = np.unique(label[land], return_inverse=True, return_counts=True)
        order = np.lexsort((labs, -counts))                     # size descending, label ascending
        rank_of = np.empty(labs.size, np.int32)
        for i in order:
            k = int(np.argmin(load))
            rank_of[i] = k
            load[k] += counts[i]
        owner[land] = rank_of[inv]
        largest, n = int(counts.max()), int(labs.size)
    else:
        largest, n = 0, 0
    cells = [np.ascontiguousarray(land[owner[land] == k], np.int32) for k in range(world)]
    return LandmassPlan(world, owner, cells, load, largest, n)


def plan_largest_apart(mesh, r_isOcean) -> LandmassPlan:
    """A two-share plan that is not balanced on purpose: share 0 is the largest landmass alone, share 1 everything else.  The walk
    of the largest landmass is the critical path of a flood call on the host: with this plan the rest of the
    planet can start iterating while that walk is still running (research/ab/r05_two_shares_flood_overlap.py)."""
    base = plan_landmasses(mesh, r_isOcean, 1)
    oc = np.ascontiguousarray(r_isOcean, np.uint8)
    label = land_components(mesh, open_ocean(mesh, oc).astype(np.uint8))
    label = np.where(oc != 0, -1, label).astype(np.int32)
    land = np.flatnonzero(label >= 0).astype(np.int32)
    owner = np.full(label.size, -1, np.int32)
    load = np.zeros(2, np.int64)
    if land.size:
        labs, counts = np.unique(label[land], return_counts=True)
        big = labs[np.flatnonzero(counts == counts.max())[0]]
        owner[land] = np.where(label[land] == big, 0, 1)
        load[0], load[1] = int(counts.max()), int(land.size - counts.max())
    cells = [np.ascontiguousarray(land[owner[land] == k], np.int32) for k in range(2)]
    return LandmassPlan(2, owner, cells, load, base.largest, base.num_landmasses)


def merge_land(plan: LandmassPlan, rank: int, field: np.ndarray, dist) -> None:
    """Host-array form of the exchange: every rank contributes the elevations of its own land cells, every rank ends up
    with the complete field (in place).  `dist` is the initialised torch.distributed module (None: single rank)."""
    if dist is None or plan.world == 1:
        return
    import torch
    n_max = max(1, plan.max_cells)
    send = torch.zeros(n_max, dtype=torch.float32)
    mine = plan.cells[rank]
    send[:mine.size] = torch.from_numpy(np.ascontiguousarray(field[mine]))
    out = [torch.empty(n_max, dtype=torch.float32) for _ in range(plan.world)]
    dist.all_gather(out, send)
    for j in range(plan.world):
        if j != rank and plan.cells[j].size:
            field[plan.cells[j]] = out[j][:plan.cells[j].size].numpy()


class ResidentLandmass:
    """The exchange for a field that stays in HBM: gather kernel over the rank's own land cells -> all-gather (device
    tensors go straight to RCCL; pinned host staging under gloo) -> scatter kernel into the other ranks' cells.  Uses the
    C ABI's halo pack / unpack entry points (wo_planet_set_halo / pack_halo / unpack_halo)."""

    def __init__(self, plan: LandmassPlan, rank: int, planet):
        self.plan, self.rank, self.planet = plan, rank, planet
        others = [plan.cells[j] for j in range(plan.world) if j != rank]
        recv = np.concatenate(others).astype(np.int32) if others else np.empty(0, np.int32)
        planet.set_halo(plan.cells[rank], recv)
        self.n_max = max(1, plan.max_cells)

    def exchange(self, dist, device=None, comm=None) -> None:
        """comm: a terrain_post.Comm — the whole exchange (pack, ncclAllGather, unpack) runs behind the C ABI on the planet's
        stream (wo_planet_exchange_allgather); otherwise torch.distributed carries it (device tensors over RCCL when `device` is
        given, pinned host staging under gloo)."""
        if self.plan.world == 1:
            return
        plan, rank = self.plan, self.rank
        sizes = [int(c.size) for c in plan.cells]
        if comm is not None:
            self.planet.exchange_allgather(comm, sizes)
            return
        if dist is None:
            return
        import torch
        if device is not None:
            # torch.empty, not torch.zeros: a fill kernel queued on torch's stream is not ordered against the pack, which
            # runs on the planet's own (non-blocking) stream and could be overwritten by a late fill.  The pack is
            # synchronous on the planet's stream; only the unused tail is zeroed, after it, on torch's stream.
            send = torch.empty(self.n_max, dtype=torch.float32, device=device)
            torch.cuda.current_stream(device).synchronize()
            if sizes[rank]:
                self.planet.pack_halo(device_ptr=send.data_ptr())          # synchronises the planet's stream
            if sizes[rank] < self.n_max:
                send[sizes[rank]:].zero_()
            out = torch.empty(plan.world * self.n_max, dtype=torch.float32, device=device)
            dist.all_gather_into_tensor(out, send)
            parts = [out[j * self.n_max: j * self.n_max + sizes[j]] for j in range(plan.world) if j != rank and sizes[j]]
            if parts:
                recv = torch.cat(parts).contiguous()
                torch.cuda.synchronize(device)
                self.planet.unpack_halo(device_ptr=recv.data_ptr())
        else:
            send = torch.zeros(self.n_max, dtype=torch.float32)
            if sizes[rank]:
                send[:sizes[rank]] = torch.from_numpy(self.planet.pack_halo())
            out = [torch.empty(self.n_max, dtype=torch.float32) for _ in range(plan.world)]
            dist.all_gather(out, send)
            parts = [out[j][:sizes[j]] for j in range(plan.world) if j != rank and sizes[j]]
            if parts:
                self.planet.unpack_halo(torch.cat(parts).numpy())


class TorchFloodExchange:
    """Planet.set_flood_exchange over torch.distributed (gloo in the CPU tests and rehearsals; host-staged, the data is a few MB
    and only moves when a rank's flood is undecided): the flag by all_reduce(MAX), the heights by all_gather of each rank's land
    cells (plan.cells)."""

    def __init__(self, plan: LandmassPlan, rank: int, dist, device=None):
        self.plan, self.rank, self.dist, self.device = plan, rank, dist, device      # device: tensors must live there (nccl backend)

    def allreduce_max(self, flag: int) -> int:
        import torch
        t = torch.tensor([int(flag)], dtype=torch.int32, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())

    def allgather(self, field: np.ndarray) -> None:
        if self.device is None:
            merge_land(self.plan, self.rank, field, self.dist)
            return
        import torch
        plan, rank = self.plan, self.rank
        n_max = max(1, plan.max_cells)
        send = torch.zeros(n_max, dtype=torch.float32)
        mine = plan.cells[rank]
        send[:mine.size] = torch.from_numpy(np.ascontiguousarray(field[mine]))
        out = torch.empty(plan.world * n_max, dtype=torch.float32, device=self.device)
        self.dist.all_gather_into_tensor(out, send.to(self.device))
        host = out.cpu().numpy()
        for j in range(plan.world):
            if j != rank and plan.cells[j].size:
                field[plan.cells[j]] = host[j * n_max: j * n_max + plan.cells[j].size]


    def broadcast(self, land: np.ndarray, sender: bool) -> None:
        """The land heights of the one rank that flooded the whole planet (it calls with sender=True), to every rank, in place."""
        import torch
        t = torch.tensor([self.rank if sender else -1], dtype=torch.int32, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        src = int(t.item())
        if src < 0:
            raise RuntimeError("flood exchange: no rank sent the flooded heights")
        if self.device is None:
            self.dist.broadcast(torch.from_numpy(land), src)          # gloo: in place on the host buffer
            return
        buf = torch.from_numpy(land).to(self.device) if sender else torch.empty(land.size, dtype=torch.float32, device=self.device)
        self.dist.broadcast(buf, src)
        if not sender:
            land[:] = buf.cpu().numpy()


class ThreadFloodExchange:
    """The same between the shares of ONE process (one host thread, context and planet per share: the partitioned code path
    rehearsed on a single GPU).  One instance per share from ThreadFloodExchange.group(plan)."""

    class _Shared:
        def __init__(self, world):
            import threading
            self.barrier = threading.Barrier(world)
            self.flags = [0] * world
            self.fields = [None] * world
            self.sent = None

    def __init__(self, plan: LandmassPlan, rank: int, shared):
        self.plan, self.rank, self.shared = plan, rank, shared

    @classmethod
    def group(cls, plan: LandmassPlan):
        sh = cls._Shared(plan.world)
        return [cls(plan, k, sh) for k in range(plan.world)]

    def abort(self):
        self.shared.barrier.abort()

    def allreduce_max(self, flag: int) -> int:
        sh = self.shared
        sh.flags[self.rank] = int(flag)
        sh.barrier.wait()
        out = max(sh.flags)
        sh.barrier.wait()                  # nobody overwrites its flag for the next call before everybody has read
        return out

    def allgather(self, field: np.ndarray) -> None:
        sh = self.shared
        sh.fields[self.rank] = field
        sh.barrier.wait()
        for j in range(self.plan.world):
            if j != self.rank and self.plan.cells[j].size:
                field[self.plan.cells[j]] = sh.fields[j][self.plan.cells[j]]
        sh.barrier.wait()


    def broadcast(self, land: np.ndarray, sender: bool) -> None:
        sh = self.shared
        if sender:
            sh.sent = land
        sh.barrier.wait()
        if not sender:
            land[:] = sh.sent
        sh.barrier.wait()


def erode_shares_concurrently(TP, mesh, r_xyz, neighborDist, field, r_isOcean, shares: int, erode_args, creep_args=None, device: int = 0,
                              planets=None, plan: "LandmassPlan | None" = None, exchange: bool = True):
    """The S-rank landmass decomposition executed by S host threads of ONE process on ONE GPU (a context, stream and planet per
    share): every share erodes `field` with the other shares' landmasses masked as ocean — erodeComposite(*erode_args), then
    applySoilCreep(*creep_args) — with the flood exchange between the shares (ThreadFloodExchange), and the land elevations are
    merged.  This is the partitioned code path of an S-GPU run, flood exchange included; shares of one planet cannot run one after
    the other once a flood call needs the other shares' CURRENT heights.  Returns (merged field, per-share erode stats,
    per-share wall seconds, plan).  planets: reuse a list of S planets of this mesh (else they are created and closed here).
    plan: a plan of `shares` shares other than plan_landmasses'.  exchange=False: no flood exchange — the shares never wait for
    each other (only exact while no flood call is undecided: every share's erode stats must then show 0 replays)."""
    import threading
    import time
    oc = np.ascontiguousarray(r_isOcean, np.uint8)
    if plan is None:
        plan = plan_landmasses(mesh, oc, shares)
    if plan.world != shares:
        raise ValueError("the plan is for another number of shares")
    own = planets is None
    if own:
        planets = [TP.Planet(mesh, r_xyz, neighborDist, ctx=TP.Context(device)) for _ in range(shares)]
    links = ThreadFloodExchange.group(plan)
    out, stats, secs, errors = [None] * shares, [None] * shares, [0.0] * shares, []

    def work(k):
        try:
            pl = planets[k]
            pl.upload(field, plan.rank_mask(k, oc))
            if exchange:
                pl.set_flood_exchange(oc, links[k])
            pl.sync()
            t0 = time.perf_counter()
            pl.erode_composite_resident(*erode_args)
            stats[k] = pl.last_erode_stats()
            if creep_args is not None:
                pl.apply_soil_creep_resident(*creep_args)
            pl.sync()
            secs[k] = time.perf_counter() - t0
            out[k] = pl.download()
            pl.set_flood_exchange(None)
        except Exception as e:              # a share that fails must not leave the others waiting at the barrier
            errors.append((k, e))
            links[k].abort()
    th = [threading.Thread(target=work, args=(k,)) for k in range(shares)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if own:
        for pl in planets:
            pl.close()
    if errors:
        raise RuntimeError(f"share {errors[0][0]} failed: {errors[0][1]!r}")
    merged = np.array(field, np.float32, copy=True)
    for k in range(shares):
        if plan.cells[k].size:
            merged[plan.cells[k]] = out[k][plan.cells[k]]
    return merged, stats, secs, plan
