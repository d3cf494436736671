"""Python mirror of the climate stage's CSR-Jacobi sweeps over the C ABI: ``smoothField`` (js/climate-util.js:5-25, the
Laplacian smoothing applied to elevation, pressure, continentality, rain-shadow and precipitation fields),
``diffuseOceanWarmth`` (js/temperature.js:19-66), ``computeWindConvergence`` (js/precipitation.js:18-52) and
``advectMoisture`` (js/precipitation.js:59-195).  Same argument order and meaning as the reference's functions."""
from __future__ import annotations

import numpy as np

from . import capi
from .terrain_post import Planet, _planet_for


def smooth_field(mesh, field: np.ndarray, passes: int, r_xyz=None, planet: Planet | None = None) -> None:
    """Mutates ``field`` (contiguous float32, numRegions) in place like the reference; returns None."""
    if not (isinstance(field, np.ndarray) and field.dtype == np.float32 and field.flags.c_contiguous):
        raise TypeError("field must be a contiguous float32 array (it is rewritten in place)")
    pl = planet or _planet_for(mesh, r_xyz)
    if field.size != pl.numRegions:
        raise ValueError("field length must equal mesh.numRegions")
    capi.check(capi.lib().wo_smooth_field(pl.handle, capi.ptr(field), int(passes)), "wo_smooth_field")


def _arr(a, dt, n, what):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dt)
    if a.size != n:
        raise ValueError(f"{what} length must equal mesh.numRegions")
    return a


def diffuse_ocean_warmth(mesh, r_oceanWarmth, r_isLand, r_plateContinentality, passes: int, r_xyz=None, planet: Planet | None = None) -> np.ndarray:
    """diffuseOceanWarmth(mesh, r_oceanWarmth, r_isLand, r_plateContinentality, passes) -> Float32Array (js/temperature.js:19)."""
    pl = planet or _planet_for(mesh, r_xyz)
    n = pl.numRegions
    out = np.empty(n, np.float32)
    capi.check(capi.lib().wo_diffuse_ocean_warmth(pl.handle, capi.ptr(_arr(r_oceanWarmth, np.float32, n, "r_oceanWarmth")), capi.ptr(_arr(r_isLand, np.uint8, n, "r_isLand")),
                                                  capi.ptr(_arr(r_plateContinentality, np.float32, n, "r_plateContinentality")), int(passes), capi.ptr(out)),
               "wo_diffuse_ocean_warmth")
    return out


def compute_wind_convergence(mesh, r_xyz, r_wind3dX, r_wind3dY, r_wind3dZ, planet: Planet | None = None) -> np.ndarray:
    """computeWindConvergence(mesh, r_xyz, r_wind3dX, r_wind3dY, r_wind3dZ) -> Float32Array (js/precipitation.js:18)."""
    pl = planet or _planet_for(mesh, r_xyz)
    n = pl.numRegions
    out = np.empty(n, np.float32)
    capi.check(capi.lib().wo_wind_convergence(pl.handle, capi.ptr(_arr(r_wind3dX, np.float32, n, "r_wind3dX")), capi.ptr(_arr(r_wind3dY, np.float32, n, "r_wind3dY")),
                                              capi.ptr(_arr(r_wind3dZ, np.float32, n, "r_wind3dZ")), capi.ptr(out)), "wo_wind_convergence")
    return out


def advect_moisture(mesh, r_xyz, r_heightKm, r_isLand, r_windE, r_windN, r_wind3dX, r_wind3dY, r_wind3dZ, r_oceanWarmth, r_coastDistLand,
                    maxHops: int, avgEdgeKm: float = 0.0, planet: Planet | None = None) -> np.ndarray:
    """advectMoisture(...) -> Float32Array (js/precipitation.js:59); avgEdgeKm is accepted and unused, as in the reference."""
    pl = planet or _planet_for(mesh, r_xyz)
    n = pl.numRegions
    out = np.empty(n, np.float32)
    f = lambda a, w: capi.ptr(_arr(a, np.float32, n, w))  # noqa: E731
    capi.check(capi.lib().wo_advect_moisture(pl.handle, f(r_heightKm, "r_heightKm"), capi.ptr(_arr(r_isLand, np.uint8, n, "r_isLand")), f(r_windE, "r_windE"),
                                             f(r_windN, "r_windN"), f(r_wind3dX, "r_wind3dX"), f(r_wind3dY, "r_wind3dY"), f(r_wind3dZ, "r_wind3dZ"),
                                             f(r_oceanWarmth, "r_oceanWarmth"), capi.ptr(_arr(r_coastDistLand, np.int32, n, "r_coastDistLand")), int(maxHops), capi.ptr(out)),
               "wo_advect_moisture")
    return out
