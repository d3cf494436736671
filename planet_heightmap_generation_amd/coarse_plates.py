"""Python mirror of the reference's plate projection step over the C ABI:

``project_coarse_plates(mesh, r_xyz, coarseMesh, coarse_xyz, coarse_r_plate, seed, numPlates)``
(js/coarse-plates.js:51-117) — HIP kernel, one thread per hi-res cell;
``smooth_and_reconnect_plates(mesh, r_plate, plateSeeds, numPasses)`` (js/plates.js:241-348) — native host stage
(order-defined in-place passes), mutates ``r_plate`` like the reference.

``generateCoarsePlates`` itself (plate seeds, motion, ocean/land on the fixed 20 000-cell mesh) is host logic of the
reference and stays there: its outputs are the inputs here.
"""
from __future__ import annotations

import numpy as np

from . import capi
from .terrain_post import Planet, _planet_for


def project_coarse_plates(mesh, r_xyz, coarseMesh, coarse_xyz, coarse_r_plate, seed, numPlates=None, planet: Planet | None = None) -> np.ndarray:
    pl = planet or _planet_for(mesh, r_xyz)
    c_off = np.ascontiguousarray(coarseMesh.adjOffset, np.int32)
    c_adj = np.ascontiguousarray(coarseMesh.adjList, np.int32)
    c_xyz = np.ascontiguousarray(coarse_xyz, np.float32)
    c_plate = np.ascontiguousarray(coarse_r_plate, np.int32)
    NC = int(coarseMesh.numRegions)
    if c_off.size != NC + 1 or c_xyz.size != 3 * NC or c_plate.size != NC:
        raise ValueError("coarse mesh / coarse_xyz / coarse_r_plate size mismatch")
    r_plate = np.empty(pl.numRegions, np.int32)
    capi.check(capi.lib().wo_project_coarse_plates(pl.handle, NC, capi.ptr(c_off), capi.ptr(c_adj), capi.ptr(c_xyz), capi.ptr(c_plate),
                                                   float(seed), -1 if numPlates is None else int(numPlates), capi.ptr(r_plate)),
               "wo_project_coarse_plates")
    return r_plate


def smooth_and_reconnect_plates(mesh, r_plate: np.ndarray, plateSeeds, numPasses: int) -> None:
    if not (isinstance(r_plate, np.ndarray) and r_plate.dtype == np.int32 and r_plate.flags.c_contiguous):
        raise TypeError("r_plate must be a contiguous int32 array (it is rewritten in place)")
    off = np.ascontiguousarray(mesh.adjOffset, np.int32)
    adj = np.ascontiguousarray(mesh.adjList, np.int32)
    if r_plate.size != off.size - 1:
        raise ValueError("r_plate length must equal mesh.numRegions")
    seeds = np.ascontiguousarray(list(plateSeeds), np.int32)
    capi.check(capi.lib().wo_smooth_reconnect_plates(int(mesh.numRegions), capi.ptr(off), capi.ptr(adj), capi.ptr(r_plate), capi.ptr(seeds),
                                                     int(seeds.size), int(numPasses)), "wo_smooth_reconnect_plates")
