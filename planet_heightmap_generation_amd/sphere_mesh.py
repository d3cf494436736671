"""Python mirror of the reference's mesh producer (js/sphere-mesh.js) over the C ABI.

``build_sphere(N, jitter, seed)`` is the counterpart of ``buildSphere(N, jitter, makeRng(seed))``
(js/sphere-mesh.js:174-186) plus ``computeNeighborDist`` (js/sphere-mesh.js:191-203); the returned
``SphereMesh`` exposes the three members the hot path reads: ``numRegions``, ``adjOffset``, ``adjList``
(js/sphere-mesh.js:144-145).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import capi


@dataclass
class SphereMesh:
    numRegions: int
    triangles: np.ndarray    # int32 [numSides]
    halfedges: np.ndarray    # int32 [numSides]
    adjOffset: np.ndarray    # int32 [numRegions + 1]
    adjList: np.ndarray      # int32 [E]
    adjTriList: np.ndarray   # int32 [E]

    @property
    def numSides(self) -> int:
        return int(self.triangles.shape[0])

    @property
    def numTriangles(self) -> int:
        return self.numSides // 3


def fibonacci_sphere(N: int, jitter: float, seed: float) -> np.ndarray:
    """generateFibonacciSphere + appended pole: float32 [3*(N+1)] (js/sphere-mesh.js:9-37,179-181)."""
    xyz = np.empty(3 * (N + 1), dtype=np.float32)
    capi.check(capi.lib().wo_fib_sphere_points(N, float(jitter), float(seed), capi.ptr(xyz)), "wo_fib_sphere_points")
    return xyz


def sphere_mesh_from_points(r_xyz: np.ndarray) -> SphereMesh:
    r_xyz = np.ascontiguousarray(r_xyz, dtype=np.float32).reshape(-1)
    V = r_xyz.size // 3
    ns = 3 * (2 * V - 4)
    tri = np.empty(ns, dtype=np.int32)
    he = np.empty(ns, dtype=np.int32)
    L = capi.lib()
    capi.check(L.wo_sphere_delaunay(V, capi.ptr(r_xyz), capi.ptr(tri), capi.ptr(he)), "wo_sphere_delaunay")
    return sphere_mesh_from_triangles(tri, he, V)


def sphere_mesh_from_triangles(tri: np.ndarray, he: np.ndarray, numRegions: int) -> SphereMesh:
    """new SphereMesh(triangles, halfedges, numRegions) (js/sphere-mesh.js:94-146)."""
    tri = np.ascontiguousarray(tri, dtype=np.int32)
    he = np.ascontiguousarray(he, dtype=np.int32)
    ns = tri.size
    off = np.empty(numRegions + 1, dtype=np.int32)
    adj = np.empty(ns, dtype=np.int32)
    adjt = np.empty(ns, dtype=np.int32)
    capi.check(capi.lib().wo_mesh_csr(numRegions, ns, capi.ptr(tri), capi.ptr(he), capi.ptr(off), capi.ptr(adj), capi.ptr(adjt)),
               "wo_mesh_csr")
    E = int(off[-1])
    return SphereMesh(numRegions, tri, he, off, adj[:E].copy(), adjt[:E].copy())


def compute_neighbor_dist(mesh: SphereMesh, r_xyz: np.ndarray) -> np.ndarray:
    out = np.empty(mesh.adjList.size, dtype=np.float32)
    r_xyz = np.ascontiguousarray(r_xyz, dtype=np.float32)
    capi.check(capi.lib().wo_neighbor_dist(mesh.numRegions, capi.ptr(mesh.adjOffset), capi.ptr(mesh.adjList),
                                           capi.ptr(r_xyz), capi.ptr(out)), "wo_neighbor_dist")
    return out


def build_sphere(N: int, jitter: float, seed: float):
    """Returns (mesh, r_xyz, neighborDist) for N requested cells (numRegions = N + 1)."""
    xyz = fibonacci_sphere(N, jitter, seed)
    mesh = sphere_mesh_from_points(xyz)
    return mesh, xyz, compute_neighbor_dist(mesh, xyz)


def triangle_elevations(mesh: SphereMesh, r_elevation: np.ndarray) -> np.ndarray:
    out = np.empty(mesh.numTriangles, dtype=np.float32)
    r_elevation = np.ascontiguousarray(r_elevation, dtype=np.float32)
    capi.check(capi.lib().wo_triangle_elevations(mesh.numTriangles, capi.ptr(mesh.triangles), capi.ptr(r_elevation),
                                                 capi.ptr(out)), "wo_triangle_elevations")
    return out
