// worker_threads counterpart of the reference's Web Worker (js/planet-worker.js) for the part of its message protocol
// that is the device path: the retained state W (:277-292), `reapply` (:341-440) and the dispatcher (:944-954).
//
//   cmd 'retain'   { mesh: { numRegions, adjOffset, adjList, triangles? }, r_xyz, neighborDist?, prePostElev, seed, r_hotspot? }
//                  What `generate` leaves in W for later reapplies, handed over by the caller (plate generation, ocean /
//                  land assignment and the climate modules are the reference's own host code and stay where they are).
//                  The mesh, positions and the pre-erosion field go to HBM ONCE and stay there.
//                  -> { type: 'retained', numRegions }
//   cmd 'reapply'  { terrainWarp, smoothing, glacialErosion, hydraulicErosion, thermalErosion, ridgeSharpening, skipClimate? }
//                  restore the pre-erosion field on the device (no upload), runPostProcessing resident, triangle
//                  elevations; the climate stages are not run here (skipClimate is reported as true, the reference's own
//                  behaviour above 300 k cells).  Same result message as the reference, typed arrays transferred:
//                  -> { type: 'reapplyDone', skipClimate: true, r_elevation, t_elevation, erosionDelta, _reapplyTiming, _postTiming }
//   cmd 'dispose'  frees the retained state -> { type: 'disposed' }
//   progress / errors exactly as the reference posts them: { type: 'progress', pct, label }, { type: 'error', message, stack };
//   an unknown command answers `Unknown command: <cmd>` (:952).
//
// Usage (Node >= 12):  const w = new Worker(new URL('./planet-worker.js', import.meta.url)); w.postMessage({ cmd: 'retain', ... })
import { parentPort } from 'worker_threads';
import { performance } from 'perf_hooks';
import addon, { defaultContext } from './native.js';
import { runPostProcessingResident } from './post-processing.js';

let W = null;          // retained state (js/planet-worker.js:22)

function progress(pct, label) { parentPort.postMessage({ type: 'progress', pct, label }); }

// the retained planet holds several GB of device memory at 40 M cells and its JS handle is a few bytes: free it explicitly
function releaseRetained() { if (W && W.planet) addon.planetDestroy(W.planet); W = null; }

function handleRetain(data) {
    try {
        const { mesh, r_xyz, neighborDist, prePostElev, seed, r_hotspot } = data;
        if (!mesh || !(mesh.adjOffset instanceof Int32Array) || !(mesh.adjList instanceof Int32Array)) throw new TypeError('retain: mesh.adjOffset / mesh.adjList must be Int32Arrays');
        if (!(r_xyz instanceof Float32Array) || !(prePostElev instanceof Float32Array)) throw new TypeError('retain: r_xyz and prePostElev must be Float32Arrays');
        releaseRetained();                               // a second retain replaces the first: its device memory goes now, not at the next GC
        const planet = addon.planetCreate(defaultContext(), mesh.numRegions, mesh.adjOffset, mesh.adjList, r_xyz, neighborDist || null);
        addon.planetUpload(planet, prePostElev, null);
        if (r_hotspot) addon.planetUploadHotspot(planet, r_hotspot);
        addon.planetSaveState(planet);                  // W.prePostElev, device copy
        W = { planet, numRegions: mesh.numRegions, triangles: mesh.triangles || null, seed, hasHotspot: !!r_hotspot };
        parentPort.postMessage({ type: 'retained', numRegions: mesh.numRegions });
    } catch (err) {
        parentPort.postMessage({ type: 'error', message: err.message, stack: err.stack });
    }
}

function handleReapply(data) {
    if (!W) { parentPort.postMessage({ type: 'error', message: 'No retained state for reapply' }); return; }
    try {
        const tTotal0 = performance.now();
        progress(0, 'Reapplying terrain…');
        let t0 = performance.now();
        addon.planetRestoreState(W.planet);             // r_elevation = new Float32Array(W.prePostElev), on the device
        const r_elevation = new Float32Array(W.numRegions);
        const tClone = performance.now() - t0;

        progress(20, 'Eroding terrain…');
        t0 = performance.now();
        const { dl_erosionDelta, postTiming } = runPostProcessingResident(W.planet, W.numRegions, r_elevation, data, W.seed, W.hasHotspot);
        const tPost = performance.now() - t0;

        progress(70, 'Computing triangle elevations…');
        t0 = performance.now();
        const t_elevation = W.triangles ? addon.triangleElevations(W.triangles, r_elevation) : new Float32Array(0);
        const tTriElev = performance.now() - t0;

        const result = {
            type: 'reapplyDone',
            skipClimate: true,
            r_elevation,
            t_elevation,
            erosionDelta: dl_erosionDelta,
            _reapplyTiming: { clone: tClone, postProcessing: tPost, wind: 0, ocean: 0, precipitation: 0, temperature: 0,
                              triangleElevations: tTriElev, workerTotal: performance.now() - tTotal0 },
            _postTiming: postTiming
        };
        parentPort.postMessage(result, [r_elevation.buffer, t_elevation.buffer, dl_erosionDelta.buffer]);
    } catch (err) {
        parentPort.postMessage({ type: 'error', message: err.message, stack: err.stack });
    }
}

parentPort.on('message', (data) => {
    const { cmd } = data;
    switch (cmd) {
        case 'retain': handleRetain(data); break;
        case 'reapply': handleReapply(data); break;
        case 'dispose': releaseRetained(); parentPort.postMessage({ type: 'disposed' }); break;
        case 'generate': case 'editRecompute': case 'computeClimate': case 'importHeightmap':
            parentPort.postMessage({ type: 'error', message: `Command not served by the device worker (host stages of the reference): ${cmd}` });
            break;
        default: parentPort.postMessage({ type: 'error', message: `Unknown command: ${cmd}` });
    }
});
