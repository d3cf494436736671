"""ctypes loader for the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never from the product package.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_DIR = Path(__file__).resolve().parent
LIB_PATH = _DIR / "_build" / "liboracle.so"
_lib = None
_i32, _f64, _p, _i64 = C.c_int32, C.c_double, C.c_void_p, C.c_int64


def build() -> None:
    subprocess.run(["make", "-s", "-C", str(_DIR)], check=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            build()
        L = C.CDLL(str(LIB_PATH))
        L.wo_or_noise_init.argtypes = [_f64, _p, _p]
        L.wo_or_noise_batch.argtypes = [_f64, C.c_int, C.c_int, _f64, _f64, _f64, _i64, _p, _p]
        L.wo_or_synthetic_terrain.argtypes = [_i32, _p, _f64, _p]
        L.wo_or_rng_seed.argtypes = [_f64, _p]
        L.wo_or_rng_next.argtypes = [_p]
        L.wo_or_rng_next.restype = _f64
        L.wo_or_cell_noise_hash.argtypes = [_i32]
        L.wo_or_cell_noise_hash.restype = C.c_uint32
        L.wo_or_warp_terrain.argtypes = [_i32, _p, _p, _p, _p, _f64, _f64, _p]
        for f in ("wo_or_smooth_elevation", "wo_or_sharpen_ridges", "wo_or_soil_creep"):
            getattr(L, f).argtypes = [_i32, _p, _p, _p, _p, _i32, _f64]
        L.wo_or_priority_flood_carve.argtypes = [_i32, _p, _p, _p, _p, _f64]
        L.wo_or_erode_composite.argtypes = [_i32, _p, _p, _p, _p, _p, _i32, _f64, _f64, _f64, _i32, _f64, _f64, _i32, _f64, _p]
        L.wo_or_smooth_field.argtypes = [_i32, _p, _p, _p, _i32]
        L.wo_or_diffuse_ocean_warmth.argtypes = [_i32, _p, _p, _p, _p, _p, _i32, _p]
        L.wo_or_wind_convergence.argtypes = [_i32, _p, _p, _p, _p, _p, _p, _p]
        L.wo_or_advect_moisture.argtypes = [_i32, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p]
        L.wo_or_project_coarse_plates.argtypes = [_i32, _p, _i32, _p, _p, _p, _p, _f64, _i32, _p]
        L.wo_or_smooth_reconnect_plates.argtypes = [_i32, _p, _p, _p, _i32, _p, _i32]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(_p)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def rng_values(seed: float, count: int) -> np.ndarray:
    st = C.c_double(0.0)
    lib().wo_or_rng_seed(float(seed), C.byref(st))
    return np.array([lib().wo_or_rng_next(C.byref(st)) for _ in range(count)], dtype=np.float64)


def noise_tables(seed: float):
    p, m = np.empty(512, np.uint8), np.empty(512, np.uint8)
    lib().wo_or_noise_init(float(seed), _ptr(p), _ptr(m))
    return p, m


def noise_batch(seed, kind, xyz, octaves=5, p0=2.0 / 3.0, p1=0.5, p2=1.0) -> np.ndarray:
    xyz = _c(xyz, np.float64).reshape(-1, 3)
    out = np.empty(xyz.shape[0], np.float64)
    lib().wo_or_noise_batch(float(seed), int(kind), int(octaves), float(p0), float(p1), float(p2), xyz.shape[0], _ptr(xyz), _ptr(out))
    return out


def synthetic_terrain(r_xyz, seed) -> np.ndarray:
    r_xyz = _c(r_xyz, np.float32)
    out = np.empty(r_xyz.size // 3, np.float32)
    lib().wo_or_synthetic_terrain(out.size, _ptr(r_xyz), float(seed), _ptr(out))
    return out


class Mesh:
    """Minimal holder of the three members the path reads (js/sphere-mesh.js:144-145)."""

    def __init__(self, adjOffset, adjList):
        self.adjOffset = _c(adjOffset, np.int32)
        self.adjList = _c(adjList, np.int32)
        self.numRegions = self.adjOffset.size - 1


def warp_terrain(mesh, r_elevation, r_xyz, seed, strength, r_hotspot=None):
    e = _c(r_elevation, np.float32).copy()
    xyz = _c(r_xyz, np.float32)
    hot = None if r_hotspot is None else _c(r_hotspot, np.float32)
    lib().wo_or_warp_terrain(mesh.numRegions, _ptr(mesh.adjOffset), _ptr(mesh.adjList), _ptr(e), _ptr(xyz), float(seed), float(strength), _ptr(hot))
    return e


def _jacobi(fn, mesh, r_elevation, r_isOcean, iterations, strength):
    e = _c(r_elevation, np.float32).copy()
    oc = _c(r_isOcean, np.uint8)
    getattr(lib(), fn)(mesh.numRegions, _ptr(mesh.adjOffset), _ptr(mesh.adjList), _ptr(e), _ptr(oc), int(iterations), float(strength))
    return e


def smooth_elevation(mesh, e, oc, iterations, strength):
    return _jacobi("wo_or_smooth_elevation", mesh, e, oc, iterations, strength)


def sharpen_ridges(mesh, e, oc, iterations, strength):
    return _jacobi("wo_or_sharpen_ridges", mesh, e, oc, iterations, strength)


def soil_creep(mesh, e, oc, iterations, strength):
    return _jacobi("wo_or_soil_creep", mesh, e, oc, iterations, strength)


def priority_flood_carve(mesh, r_elevation, r_isOcean, carveStrength):
    e = _c(r_elevation, np.float32).copy()
    oc = _c(r_isOcean, np.uint8)
    lib().wo_or_priority_flood_carve(mesh.numRegions, _ptr(mesh.adjOffset), _ptr(mesh.adjList), _ptr(e), _ptr(oc), float(carveStrength))
    return e


def erode_composite(mesh, r_elevation, r_xyz, r_isOcean, hIters, K, m, dt, tIters, talusSlope, kThermal,
                    gIters, glacialStrength, neighborDist):
    e = _c(r_elevation, np.float32).copy()
    xyz, oc, nd = _c(r_xyz, np.float32), _c(r_isOcean, np.uint8), _c(neighborDist, np.float32)
    lib().wo_or_erode_composite(mesh.numRegions, _ptr(mesh.adjOffset), _ptr(mesh.adjList), _ptr(e), _ptr(xyz), _ptr(oc),
                                int(hIters), float(K), float(m), float(dt), int(tIters), float(talusSlope), float(kThermal),
                                int(gIters), float(glacialStrength), _ptr(nd))
    return e


def project_coarse_plates(mesh, r_xyz, coarse_mesh, coarse_xyz, coarse_r_plate, seed, numPlates=None) -> np.ndarray:
    """js/coarse-plates.js:51 projectCoarsePlates -> Int32Array r_plate."""
    N = mesh.numRegions
    out = np.empty(N, np.int32)
    xyz, cxyz, cp = _c(r_xyz, np.float32), _c(coarse_xyz, np.float32), _c(coarse_r_plate, np.int32)
    lib().wo_or_project_coarse_plates(N, _ptr(xyz), coarse_mesh.numRegions, _ptr(coarse_mesh.adjOffset), _ptr(coarse_mesh.adjList),
                                      _ptr(cxyz), _ptr(cp), float(seed), -1 if numPlates is None else int(numPlates), _ptr(out))
    return out


def smooth_reconnect_plates(mesh, r_plate, plate_seeds, numPasses) -> np.ndarray:
    """js/plates.js:241 smoothAndReconnectPlates (returns the updated copy)."""
    rp = _c(r_plate, np.int32).copy()
    seeds = _c(np.asarray(list(plate_seeds)), np.int32)
    lib().wo_or_smooth_reconnect_plates(mesh.numRegions, _ptr(mesh.adjOffset), _ptr(mesh.adjList), _ptr(rp), seeds.size, _ptr(seeds), int(numPasses))
    return rp


def smooth_field(mesh, field, passes) -> np.ndarray:
    """js/climate-util.js:5 smoothField (returns the smoothed copy)."""
    f = _c(field, np.float32).copy()
    lib().wo_or_smooth_field(mesh.numRegions, _ptr(mesh.adjOffset), _ptr(mesh.adjList), _ptr(f), int(passes))
    return f


def diffuse_ocean_warmth(mesh, r_oceanWarmth, r_isLand, r_plateContinentality, passes) -> np.ndarray:
    """js/temperature.js:19 diffuseOceanWarmth -> Float32Array."""
    out = np.empty(mesh.numRegions, np.float32)
    w = None if r_oceanWarmth is None else _c(r_oceanWarmth, np.float32)
    c = None if r_plateContinentality is None else _c(r_plateContinentality, np.float32)
    lib().wo_or_diffuse_ocean_warmth(mesh.numRegions, _ptr(mesh.adjOffset), _ptr(mesh.adjList), _ptr(w), _ptr(_c(r_isLand, np.uint8)), _ptr(c), int(passes), _ptr(out))
    return out


def wind_convergence(mesh, r_xyz, wx, wy, wz) -> np.ndarray:
    """js/precipitation.js:18 computeWindConvergence -> Float32Array."""
    out = np.empty(mesh.numRegions, np.float32)
    lib().wo_or_wind_convergence(mesh.numRegions, _ptr(mesh.adjOffset), _ptr(mesh.adjList), _ptr(_c(r_xyz, np.float32)), _ptr(_c(wx, np.float32)),
                                 _ptr(_c(wy, np.float32)), _ptr(_c(wz, np.float32)), _ptr(out))
    return out


def advect_moisture(mesh, r_xyz, r_heightKm, r_isLand, r_windE, r_windN, wx, wy, wz, r_oceanWarmth, r_coastDistLand, maxHops) -> np.ndarray:
    """js/precipitation.js:59 advectMoisture -> Float32Array."""
    out = np.empty(mesh.numRegions, np.float32)
    w = None if r_oceanWarmth is None else _c(r_oceanWarmth, np.float32)
    a = [_c(r_xyz, np.float32), _c(r_heightKm, np.float32), _c(r_isLand, np.uint8), _c(r_windE, np.float32), _c(r_windN, np.float32),
         _c(wx, np.float32), _c(wy, np.float32), _c(wz, np.float32)]
    cd = _c(r_coastDistLand, np.int32)
    lib().wo_or_advect_moisture(mesh.numRegions, _ptr(mesh.adjOffset), _ptr(mesh.adjList), *[_ptr(x) for x in a], _ptr(w), _ptr(cd), int(maxHops), _ptr(out))
    return out
