"""Deterministic inputs for the climate sweeps (diffuseOceanWarmth, computeWindConvergence, advectMoisture): built from a
mesh, its positions and an elevation field, shared by the golden generator (oracle/ref_harness/make_golden_climate.py)
and the tests, so that only the reference's OUTPUTS are stored."""
import numpy as np


def sweep_inputs(adjOffset, adjList, xyz, elevation):
    N = adjOffset.size - 1
    p = np.asarray(xyz, np.float64).reshape(-1, 3)
    e = np.asarray(elevation, np.float64)
    isLand = (e > 0).astype(np.uint8)
    warmth = (0.6 * np.sin(3.0 * p[:, 1]) + 0.4 * np.cos(5.0 * p[:, 0] + 2.0 * p[:, 2]) - 0.1).astype(np.float32)
    warmth[isLand == 1] = 0.0
    plateCont = np.clip(0.5 + 0.6 * np.sin(2.5 * p[:, 0] + 1.3) * np.cos(1.7 * p[:, 2]), 0.0, 1.0).astype(np.float32)
    plateCont[isLand == 0] *= np.float32(0.3)
    # a tangent wind field: mostly zonal with a meridional wobble; a few calm cells
    east = np.stack([-p[:, 2], np.zeros(N), p[:, 0]], 1)
    el = np.linalg.norm(east, axis=1); el[el < 1e-9] = 1.0
    east /= el[:, None]
    north = np.cross(p, east)
    we = (0.8 * np.cos(4.0 * p[:, 1]) + 0.1).astype(np.float32)
    wn = (0.35 * np.sin(6.0 * p[:, 0] + p[:, 1])).astype(np.float32)
    calm = (np.arange(N) % 97) == 0
    we[calm] = 0.0; wn[calm] = 0.0
    w3 = (we.astype(np.float64)[:, None] * east + wn.astype(np.float64)[:, None] * north).astype(np.float32)
    heightKm = (np.maximum(e, 0.0) * 6.0).astype(np.float32)
    # land cells that touch the ocean: coast distance 0, every other cell -1 (only `=== 0` is read by the sweep)
    rows = np.repeat(np.arange(N), np.diff(adjOffset))
    touches = np.zeros(N, bool)
    np.logical_or.at(touches, rows, isLand[adjList] == 0)
    coastDist = np.where((isLand == 1) & touches, 0, -1).astype(np.int32)
    return dict(isLand=isLand, oceanWarmth=warmth, plateContinentality=plateCont, windE=we, windN=wn,
                wind3dX=np.ascontiguousarray(w3[:, 0]), wind3dY=np.ascontiguousarray(w3[:, 1]), wind3dZ=np.ascontiguousarray(w3[:, 2]),
                heightKm=heightKm, coastDistLand=coastDist)


SWEEP_CASES = dict(diffuse_passes=(0, 1, 6), diffuse_no_cont_passes=3, advect_hops=(8, 13))
