// Node counterpart of runPostProcessing (js/planet-worker.js:40-102): same arguments, same slider ->
// parameter mapping, same { dl_erosionDelta, postTiming:[{stage, ms}] } result, r_elevation mutated in place.
// The field stays resident in HBM across the passes (one upload, one download), which is the "reapply"
// pattern of js/planet-worker.js:341-440.
import { performance } from 'perf_hooks';
import addon, { planetFor } from './native.js';

export function runPostProcessing(mesh, r_xyz, r_elevation, params, neighborDist, seed, r_hotspot) {
    const p = planetFor(mesh, r_xyz, neighborDist);
    addon.planetUpload(p, r_elevation, null);
    if (r_hotspot) addon.planetUploadHotspot(p, r_hotspot);
    return runPostProcessingResident(p, mesh.numRegions, r_elevation, params, seed, !!r_hotspot);
}

// The same on a field that is ALREADY resident in HBM (the worker's "reapply": js/planet-worker.js:341-440 restores the
// pre-erosion field on the device and calls this; nothing is uploaded).  r_elevation receives the result.
export function runPostProcessingResident(p, N, r_elevation, params, seed, useHotspot) {
    const { smoothing, glacialErosion, hydraulicErosion, thermalErosion, ridgeSharpening, terrainWarp } = params;
    const timing = [];
    const r_hotspot = useHotspot;
    const timed = (stage, fn) => { const t0 = performance.now(); fn(); addon.planetSync(p); timing.push({ stage, ms: performance.now() - t0 }); };

    if (terrainWarp > 0) {
        timed(`Terrain warp (strength=${terrainWarp.toFixed(2)})`, () => addon.warpTerrainResident(p, seed, terrainWarp, r_hotspot ? 1 : 0));
    }
    addon.planetOceanFromElevation(p);                                 // r_isOcean = elev <= 0, after the warp (:51-54)
    const preErosion = new Float32Array(N);
    addon.planetDownload(p, preErosion);

    if (smoothing > 0) {
        const smoothIters = Math.round(1 + smoothing * 4), smoothStr = 0.2 + smoothing * 0.5;
        timed(`Smoothing (${smoothIters} iters, str=${smoothStr.toFixed(2)})`, () => addon.smoothElevationResident(p, smoothIters, smoothStr));
    }
    if (glacialErosion > 0 || hydraulicErosion > 0 || thermalErosion > 0) {
        const gIters = Math.round(glacialErosion * 10), hIters = Math.round(hydraulicErosion * 20), hK = hydraulicErosion * 0.0006;
        const tIters = Math.round(thermalErosion * 10), talusSlope = 1.2 - thermalErosion * 0.4, kThermal = thermalErosion * 0.15;
        timed(`Erosion composite (h=${hIters}, t=${tIters}, g=${gIters})`,
              () => addon.erodeCompositeResident(p, hIters, hK, 0.5, 1.0, tIters, talusSlope, kThermal, gIters, glacialErosion));
    }
    if (ridgeSharpening > 0) {
        const rsIters = Math.round(1 + ridgeSharpening * 3), rsStr = ridgeSharpening * 0.08;
        timed(`Ridge sharpening (${rsIters} iters)`, () => addon.sharpenRidgesResident(p, rsIters, rsStr));
    }
    timed('Soil creep (3 iters)', () => addon.applySoilCreepResident(p, 3, 0.1125));

    addon.planetDownload(p, r_elevation);
    const dl_erosionDelta = new Float32Array(N);
    for (let r = 0; r < N; r++) dl_erosionDelta[r] = r_elevation[r] - preErosion[r];
    return { dl_erosionDelta, postTiming: timing, deviceStages: addon.lastStageTiming(p) };
}
