"""Domain decomposition of the Jacobi passes over index bands with a one-ring halo exchange per iteration.

The per-cell passes whose new value depends on the previous field of the cell's neighbours only — smoothElevation
(js/terrain-post.js:317-354), applySoilCreep (:758-794), smoothField (js/climate-util.js:5-25) — shard by cell:
rank k owns the contiguous index band [N*k/W, N*(k+1)/W) (latitude bands about the Fibonacci spiral axis), keeps a
copy of the cells one hop outside it (the halo) and, after every iteration, receives the new values of its halo cells
from their owners (`torch.distributed` point-to-point: RCCL between GPUs, gloo in the CPU tests).  Owned cells always
see their complete neighbour rows in the original order, so the partitioned result is bit-identical to the
unpartitioned one; halo cells are recomputed locally with truncated rows and simply overwritten by the exchange.

The order-defined passes of erodeComposite (flood, flow, implicit solve, ice, carve) follow drainage paths across bands
and are NOT decomposed; whole planets per GPU are the scaling mode for the full stack.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List

import numpy as np


class _Mesh:
    def __init__(self, adjOffset, adjList):
        self.adjOffset, self.adjList, self.numRegions = adjOffset, adjList, adjOffset.size - 1


@dataclass
class BandPart:
    rank: int
    lo: int                      # owned global ids [lo, hi)
    hi: int
    local_ids: np.ndarray        # global id of every local cell, ascending (owned and halo interleaved by id)
    owned_pos: np.ndarray        # positions of the owned cells in the local arrays
    mesh: _Mesh                  # local CSR (rows of owned cells complete and in the original order)
    adj_slots: np.ndarray        # for every local adjacency slot, the global slot it came from (neighborDist etc.)
    send: List[np.ndarray]       # send[j]: local positions of owned cells that rank j keeps in its halo (ascending id)
    recv: List[np.ndarray]       # recv[j]: local positions of halo cells owned by rank j (ascending id)


class BandPlan:
    """Index-band partition of a CSR mesh for `world` ranks with a one-ring halo."""

    def __init__(self, mesh, world: int):
        off = np.ascontiguousarray(mesh.adjOffset, np.int64)
        adj = np.ascontiguousarray(mesh.adjList, np.int64)
        N = off.size - 1
        if world < 1 or world > N:
            raise ValueError("world must be between 1 and numRegions")
        self.world, self.N = world, N
        self.bounds = [N * k // world for k in range(world + 1)]
        owner = np.empty(N, np.int32)
        for k in range(world):
            owner[self.bounds[k]:self.bounds[k + 1]] = k
        rows = np.repeat(np.arange(N, dtype=np.int64), np.diff(off))
        self.parts: List[BandPart] = []
        halos = []
        for k in range(world):
            lo, hi = self.bounds[k], self.bounds[k + 1]
            nb = adj[off[lo]:off[hi]]
            halos.append(np.unique(nb[(nb < lo) | (nb >= hi)]))
        for k in range(world):
            lo, hi = self.bounds[k], self.bounds[k + 1]
            local = np.union1d(np.arange(lo, hi, dtype=np.int64), halos[k])
            pos_of = np.full(N, -1, np.int64)
            pos_of[local] = np.arange(local.size)
            # local CSR: every local cell's row restricted to local cells, original order kept
            sel = (pos_of[rows] >= 0) & (pos_of[adj] >= 0)
            l_rows, l_adj = pos_of[rows[sel]], pos_of[adj[sel]]
            l_off = np.zeros(local.size + 1, np.int64)
            np.add.at(l_off, l_rows + 1, 1)
            l_off = np.cumsum(l_off)
            order = np.argsort(l_rows, kind="stable")            # rows ascend with the global id, so this is the identity up to grouping
            send = [pos_of[halos[j][owner[halos[j]] == k]] if j != k else np.empty(0, np.int64) for j in range(world)]
            recv = [pos_of[halos[k][owner[halos[k]] == j]] if j != k else np.empty(0, np.int64) for j in range(world)]
            self.parts.append(BandPart(k, lo, hi, local, pos_of[lo:hi].copy(),
                                       _Mesh(l_off.astype(np.int32), l_adj[order].astype(np.int32)),
                                       np.flatnonzero(sel)[order], send, recv))

    def scatter(self, rank: int, field: np.ndarray) -> np.ndarray:
        """Local copy (owned + halo) of a global per-cell array."""
        return np.ascontiguousarray(field[self.parts[rank].local_ids])

    def scatter_slots(self, rank: int, per_slot: np.ndarray) -> np.ndarray:
        """Local copy of a per-adjacency-slot array (e.g. neighborDist)."""
        return np.ascontiguousarray(per_slot[self.parts[rank].adj_slots])

    def scatter_xyz(self, rank: int, r_xyz: np.ndarray) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(r_xyz).reshape(-1, 3)[self.parts[rank].local_ids].reshape(-1))


def exchange_halo(part: BandPart, field_local: np.ndarray, dist, device=None) -> None:
    """After an iteration: owners send the new values of the cells their neighbours keep as halo.  `dist` is the
    initialised torch.distributed module; with `device` (a torch cuda device) the buffers travel as device tensors (RCCL),
    otherwise as CPU tensors (gloo)."""
    import torch
    ops, rbufs = [], []
    for j in range(len(part.send)):
        if j == part.rank:
            continue
        if part.send[j].size:
            t = torch.from_numpy(np.ascontiguousarray(field_local[part.send[j]]))
            ops.append(dist.P2POp(dist.isend, t.to(device) if device is not None else t, j))
        if part.recv[j].size:
            r = torch.empty(part.recv[j].size, dtype=torch.from_numpy(field_local[:1]).dtype, device=device if device is not None else "cpu")
            rbufs.append((j, r))
            ops.append(dist.P2POp(dist.irecv, r, j))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for j, r in rbufs:
        field_local[part.recv[j]] = r.cpu().numpy()


def banded_jacobi(part: BandPart, field_local: np.ndarray, iterations: int, step: Callable[[np.ndarray], None], dist, device=None) -> None:
    """`iterations` sweeps of a one-hop Jacobi pass on this rank's band: `step(field_local)` advances the local field by
    ONE iteration in place (e.g. ``lambda f: planet.apply_soil_creep(f, oc_local, 1, strength)``), the halo is refreshed
    after each."""
    for _ in range(iterations):
        step(field_local)
        if dist is not None:
            exchange_halo(part, field_local, dist, device)


class ResidentBand:
    """A band whose field stays in HBM: after each resident iteration only the halo values move (pack kernel -> send /
    receive -> unpack kernel).  With `device` the pack / unpack buffers are torch device tensors that go straight to
    RCCL; without it they are staged through pinned host memory (gloo in the tests)."""

    def __init__(self, part: BandPart, planet):
        self.part, self.planet = part, planet
        self.peers = [j for j in range(len(part.send)) if j != part.rank and (part.send[j].size or part.recv[j].size)]
        self.send_sizes = [int(part.send[j].size) for j in self.peers]
        self.recv_sizes = [int(part.recv[j].size) for j in self.peers]
        cat = lambda xs: np.concatenate(xs).astype(np.int32) if xs else np.empty(0, np.int32)
        planet.set_halo(cat([part.send[j] for j in self.peers]), cat([part.recv[j] for j in self.peers]))

    def exchange(self, dist, device=None, comm=None) -> None:
        """comm: a terrain_post.Comm — pack, ncclSend / ncclRecv with the two chain neighbours and unpack behind the C ABI
        (wo_planet_exchange_neighbors).  Index bands only touch rank - 1 and rank + 1; the lists are [to prev | to next]."""
        if comm is not None:
            if any(abs(j - self.part.rank) != 1 for j in self.peers):
                raise ValueError("the C-ABI band exchange talks to rank - 1 and rank + 1 only")
            to_prev = sum(a for j, a in zip(self.peers, self.send_sizes) if j < self.part.rank)
            from_prev = sum(b for j, b in zip(self.peers, self.recv_sizes) if j < self.part.rank)
            self.planet.exchange_neighbors(comm, to_prev, from_prev)
            return
        import torch
        ns, nr = sum(self.send_sizes), sum(self.recv_sizes)
        if device is not None:
            sbuf = torch.empty(max(ns, 1), dtype=torch.float32, device=device)
            rbuf = torch.empty(max(nr, 1), dtype=torch.float32, device=device)
            self.planet.pack_halo(device_ptr=sbuf.data_ptr())
        else:
            sbuf = torch.from_numpy(self.planet.pack_halo()) if ns else torch.empty(0)
            rbuf = torch.empty(nr, dtype=torch.float32)
        ops, so, ro = [], 0, 0
        for j, a, b in zip(self.peers, self.send_sizes, self.recv_sizes):
            if a:
                ops.append(dist.P2POp(dist.isend, sbuf[so:so + a], j)); so += a
            if b:
                ops.append(dist.P2POp(dist.irecv, rbuf[ro:ro + b], j)); ro += b
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if nr:
            if device is not None:
                torch.cuda.synchronize(device)
                self.planet.unpack_halo(device_ptr=rbuf.data_ptr())
            else:
                self.planet.unpack_halo(rbuf.numpy())

    def jacobi(self, step_resident: Callable[[], None], iterations: int, dist, device=None, comm=None) -> None:
        """`step_resident()` advances the resident field by ONE iteration (e.g. ``lambda: planet.apply_soil_creep_resident(1, s)``)."""
        for _ in range(iterations):
            step_resident()
            if dist is not None or comm is not None:
                self.exchange(dist, device, comm)


def gather_owned(plan: BandPlan, rank: int, field_local: np.ndarray, dist) -> np.ndarray | None:
    """Assemble the global field on rank 0 from every rank's owned cells (testing / output)."""
    import torch
    own = np.ascontiguousarray(field_local[plan.parts[rank].owned_pos])
    if dist is None:
        return own
    bufs = [None] * plan.world
    dist.all_gather_object(bufs, own)
    return np.concatenate(bufs) if rank == 0 else None
