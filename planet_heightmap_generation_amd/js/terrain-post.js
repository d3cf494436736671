// Drop-in for the reference's js/terrain-post.js: the same five exports with the same signatures, in-place
// mutation of r_elevation and `undefined` return value (js/terrain-post.js:233,317,369,713,758).  Every pass
// runs in the HIP kernels of libworogen through the N-API shim.
//
// The mesh / r_xyz / neighborDist are uploaded once per mesh object and stay resident (mirrors W in
// js/planet-worker.js).  smoothElevation / sharpenRidges / applySoilCreep do not receive r_xyz in the
// reference's signature; in the worker they are always preceded by warpTerrain or erodeComposite on the same
// mesh (js/planet-worker.js:45-94) which binds it — or call bindMesh(mesh, r_xyz, neighborDist) once.
import addon, { planetFor, bindMesh } from './native.js';

export { bindMesh };

export function warpTerrain(mesh, r_elevation, r_xyz, seed, strength, r_hotspot) {
    if (strength <= 0) return;                                         // js/terrain-post.js:234
    addon.warpTerrain(planetFor(mesh, r_xyz), r_elevation, seed, strength, r_hotspot || null);
}

export function smoothElevation(mesh, r_elevation, r_isOcean, iterations, strength) {
    addon.smoothElevation(planetFor(mesh), r_elevation, r_isOcean, iterations, strength);
}

export function erodeComposite(mesh, r_elevation, r_xyz, r_isOcean,
    hIters, K, m, dt,
    tIters, talusSlope, kThermal,
    gIters, glacialStrength,
    neighborDist)
{
    gIters = gIters || 0;                                              // js/terrain-post.js:375-376
    glacialStrength = glacialStrength || 0;
    if (!neighborDist) {
        // the reference dereferences neighborDist[j] and throws a TypeError when it is missing
        // (js/terrain-post.js:517,599; the fallback path of js/generate.js:777 hits exactly that)
        throw new TypeError('erodeComposite: neighborDist is required');
    }
    addon.erodeComposite(planetFor(mesh, r_xyz, neighborDist), r_elevation, r_isOcean,
        hIters, K, m, dt, tIters, talusSlope, kThermal, gIters, glacialStrength);
}

export function sharpenRidges(mesh, r_elevation, r_isOcean, iterations, strength) {
    addon.sharpenRidges(planetFor(mesh), r_elevation, r_isOcean, iterations, strength);
}

export function applySoilCreep(mesh, r_elevation, r_isOcean, iterations, strength) {
    addon.applySoilCreep(planetFor(mesh), r_elevation, r_isOcean, iterations, strength);
}
