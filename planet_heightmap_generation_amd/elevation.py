"""Python mirror of the reference's ``assignElevation`` (js/elevation.js:216-1391) over the C ABI.

``assign_elevation(mesh, r_xyz, plateIsOcean, r_plate, plateVec, plateSeeds, noise, noiseMag, seed, spread,
plateDensity, superPlateData)`` takes the reference's argument list (Sets -> Python sets / iterables in
insertion order, keyed objects -> dicts) and returns the same result object as a dict.  Per-cell work runs in
HIP kernels, order-defined graph traversals in native host code (csrc/elevation_host.cc).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .terrain_post import Planet, _planet_for

DEBUG_LAYERS = ("base", "tectonic", "noise", "interior", "coastal", "ocean", "hotspot", "tecActivity", "margins", "backArc",
                "foldRidge", "orogenicPower")


class PlateTable(C.Structure):
    _fields_ = [("numIds", C.c_int32), ("hasVec", C.c_void_p), ("pole", C.c_void_p), ("omega", C.c_void_p), ("isOcean", C.c_void_p),
                ("density", C.c_void_p)]


class SimplexNoise:
    """new SimplexNoise(seed): perm / pm12 tables (js/simplex-noise.js:6-15)."""

    def __init__(self, seed=0):
        self.seed = float(seed)
        self.perm = np.empty(512, np.uint8)
        self.pm12 = np.empty(512, np.uint8)
        capi.check(capi.lib().wo_noise_tables(self.seed, capi.ptr(self.perm), capi.ptr(self.pm12)), "wo_noise_tables")


def _table(ids_is_ocean, vec: dict, density: dict):
    """Dense-by-id arrays from the reference's keyed objects."""
    ids = set(vec) | set(density) | set(ids_is_ocean)
    n = (max(ids) + 1) if ids else 1
    has = np.zeros(n, np.uint8); pole = np.zeros(3 * n, np.float64); omega = np.zeros(n, np.float64)
    oc = np.zeros(n, np.uint8); dens = np.full(n, np.nan, np.float64)
    for pid, v in vec.items():
        has[pid] = 1; pole[3 * pid:3 * pid + 3] = v["pole"]; omega[pid] = v["omega"]
    for pid in ids_is_ocean:
        oc[pid] = 1
    for pid, d in density.items():
        dens[pid] = d
    t = PlateTable(n, has.ctypes.data, pole.ctypes.data, omega.ctypes.data, oc.ctypes.data, dens.ctypes.data)
    return t, (has, pole, omega, oc, dens)      # keep the arrays alive


def assign_elevation(mesh, r_xyz, plateIsOcean, r_plate, plateVec, plateSeeds, noise, noiseMag, seed, spread, plateDensity,
                     superPlateData=None, planet: Planet | None = None, debug=True):
    pl = planet or _planet_for(mesh, r_xyz)
    N = pl.numRegions
    r_plate = np.ascontiguousarray(r_plate, np.int32)
    seeds = np.ascontiguousarray(list(plateSeeds), np.int32)
    t, keep = _table(set(plateIsOcean), plateVec, plateDensity)
    ts, keep2, r_super = None, None, None
    if superPlateData is not None:
        r_super = np.ascontiguousarray(superPlateData["r_superPlate"], np.int32)
        ts, keep2 = _table(set(superPlateData["superPlateIsOcean"]), superPlateData["superPlateVec"], superPlateData["superPlateDensity"])
    e = np.empty(N, np.float32); st = np.empty(N, np.float32)
    dl = np.empty(12 * N, np.float32) if debug else None
    mo = np.empty(N, np.int32); co = np.empty(N, np.int32); oc = np.empty(N, np.int32); cnt = np.zeros(3, np.int32)
    capi.check(capi.lib().wo_assign_elevation(pl.handle, capi.ptr(r_plate), C.byref(t), capi.ptr(seeds), seeds.size, capi.ptr(r_super),
                                              C.byref(ts) if ts is not None else None, capi.ptr(noise.perm), capi.ptr(noise.pm12),
                                              float(noiseMag), float(seed), float(spread), capi.ptr(e), capi.ptr(st), capi.ptr(dl),
                                              capi.ptr(mo), capi.ptr(co), capi.ptr(oc), capi.ptr(cnt)), "assignElevation")
    layers = {name: dl[i * N:(i + 1) * N] for i, name in enumerate(DEBUG_LAYERS)} if debug else {}
    if superPlateData is not None and debug:
        layers["superPlates"] = r_super.astype(np.float32)
    return {"r_elevation": e, "mountain_r": mo[:cnt[0]].tolist(), "coastline_r": co[:cnt[1]].tolist(), "ocean_r": oc[:cnt[2]].tolist(),
            "r_stress": st, "debugLayers": layers, "_timing": [{"stage": k, "ms": v} for k, v in pl.last_stage_timing().items()]}
