// Counterpart of the reference's js/sphere-mesh.js on the native mesh producer (no Delaunator needed):
// buildSphere / SphereMesh / computeNeighborDist keep their names, argument order and result shapes
// (js/sphere-mesh.js:94-146,174-203).  `rngSeed` is the number the caller would have passed to makeRng().
import addon from './native.js';

export class SphereMesh {
    constructor(triangles, halfedges, numRegions) {
        this.triangles = triangles;
        this.halfedges = halfedges;
        this.numRegions = numRegions;
        this.numSides = triangles.length;
        this.numTriangles = (triangles.length / 3) | 0;
        const csr = addon.meshCsr(numRegions, triangles, halfedges);
        this._adjOffset = csr.adjOffset;
        this._adjList = csr.adjList;
        this._adjTriList = csr.adjTriList;
        this.adjOffset = this._adjOffset;
        this.adjList = this._adjList;
    }
    _next(s)     { return (s % 3 === 2) ? s - 2 : s + 1; }
    s_begin_r(s) { return this.triangles[s]; }
    s_end_r(s)   { return this.triangles[this._next(s)]; }
    s_inner_t(s) { return (s / 3) | 0; }
    s_outer_t(s) { return (this.halfedges[s] / 3) | 0; }
    r_circulate_r(out, r) {
        const start = this._adjOffset[r], len = this._adjOffset[r + 1] - start;
        out.length = len;
        for (let i = 0; i < len; i++) out[i] = this._adjList[start + i];
        return out;
    }
    r_circulate_t(out, r) {
        const start = this._adjOffset[r], len = this._adjOffset[r + 1] - start;
        out.length = len;
        for (let i = 0; i < len; i++) out[i] = this._adjTriList[start + i];
        return out;
    }
}

export function generateFibonacciSphere(N, jitter, rngSeed) {
    return addon.fibSpherePoints(N, jitter, rngSeed).subarray(0, 3 * N);
}

// buildSphere(N, jitter, makeRng(seed)) in the reference; here the seed itself is passed.
export function buildSphere(N, jitter, rngSeed) {
    const r_xyz = addon.fibSpherePoints(N, jitter, rngSeed);         // pole already appended at index N
    const { triangles, halfedges } = addon.sphereDelaunay(r_xyz);
    return { mesh: new SphereMesh(triangles, halfedges, N + 1), r_xyz };
}

export function computeNeighborDist(mesh, r_xyz) {
    return addon.neighborDist(mesh.adjOffset, mesh.adjList, r_xyz);
}

export function computeTriangleElevations(mesh, r_elevation) {       // js/planet-worker.js:29-37
    return addon.triangleElevations(mesh.triangles, r_elevation);
}
