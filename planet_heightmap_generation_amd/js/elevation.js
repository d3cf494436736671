// Drop-in for the reference's js/elevation.js `assignElevation` (js/elevation.js:216-1391): same argument list
// and the same result object { r_elevation, mountain_r:Set, coastline_r:Set, ocean_r:Set, r_stress, debugLayers,
// _timing:[{stage, ms}] }.  Per-cell work runs in HIP kernels, the order-defined graph traversals in native host
// code; this module only converts the reference's keyed objects / Sets into dense tables and back.
import addon, { planetFor } from './native.js';

const LAYERS = ['base', 'tectonic', 'noise', 'interior', 'coastal', 'ocean', 'hotspot', 'tecActivity', 'margins', 'backArc', 'foldRidge', 'orogenicPower'];

function denseTable(isOceanSet, vec, density) {
    let maxId = -1;
    for (const k of Object.keys(vec)) maxId = Math.max(maxId, +k);
    for (const k of Object.keys(density)) maxId = Math.max(maxId, +k);
    for (const k of isOceanSet) maxId = Math.max(maxId, +k);
    const n = maxId + 1;
    const t = { numIds: n, hasVec: new Uint8Array(n), pole: new Float64Array(3 * n), omega: new Float64Array(n),
                isOcean: new Uint8Array(n), density: new Float64Array(n).fill(NaN) };
    for (const k of Object.keys(vec)) { const id = +k, v = vec[k]; t.hasVec[id] = 1; t.pole.set(v.pole, 3 * id); t.omega[id] = v.omega; }
    for (const k of isOceanSet) t.isOcean[+k] = 1;
    for (const k of Object.keys(density)) t.density[+k] = density[k];
    return t;
}

export function assignElevation(mesh, r_xyz, plateIsOcean, r_plate, plateVec, plateSeeds, noise, noiseMag, seed, spread, plateDensity, superPlateData) {
    const p = planetFor(mesh, r_xyz);
    const plates = denseTable(plateIsOcean, plateVec, plateDensity);
    const hasSuper = superPlateData != null;
    const sup = hasSuper ? denseTable(superPlateData.superPlateIsOcean, superPlateData.superPlateVec, superPlateData.superPlateDensity) : null;
    const res = addon.assignElevation(p, r_plate, plates, Int32Array.from(plateSeeds), hasSuper ? superPlateData.r_superPlate : null, sup,
                                      noise.perm, noise.pm12, noiseMag, seed, spread, true);
    const N = mesh.numRegions;
    const debugLayers = {};
    LAYERS.forEach((name, i) => { debugLayers[name] = res.debugLayers.subarray(i * N, (i + 1) * N); });
    if (hasSuper) debugLayers.superPlates = new Float32Array(superPlateData.r_superPlate);
    return { r_elevation: res.r_elevation, mountain_r: new Set(res.mountain), coastline_r: new Set(res.coastline), ocean_r: new Set(res.ocean),
             r_stress: res.r_stress, debugLayers, _timing: addon.lastStageTiming(p) };
}
