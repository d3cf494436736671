// Loads the N-API shim (../worogen.node -> libworogen.so).  ES module, Node >= 12.
// There is no JavaScript fallback for the device passes: if the addon or a HIP device is missing the
// calls throw, and the worker's try/catch reports it (js/planet-worker.js:336-338 in the reference).
import { createRequire } from 'module';
import { fileURLToPath } from 'url';
import path from 'path';

const require = createRequire(import.meta.url);
const here = path.dirname(fileURLToPath(import.meta.url));

let addon;
try {
    addon = require(path.join(here, '..', 'worogen.node'));
} catch (e) {
    throw new Error('orogen-hip: cannot load worogen.node (build it with `python -c "import __graft_entry__ as g; g.build()"`): ' + e.message);
}

let ctx = null;
export function defaultContext(device = 0) {
    if (!ctx) ctx = addon.ctxCreate(device);          // throws when no HIP device is usable
    return ctx;
}

// One device-resident planet per mesh object (the worker keeps one mesh in W, js/planet-worker.js:277-292).
// The planet is rebuilt when a later call brings a different r_xyz / neighborDist array for the same mesh object (or
// the mesh's own arrays were replaced): the device copy must never silently disagree with what the caller passes.
const planets = new WeakMap();
export function planetFor(mesh, r_xyz, neighborDist) {
    let ent = planets.get(mesh);
    const stale = ent && ((r_xyz && ent.r_xyz !== r_xyz) || (neighborDist && ent.neighborDist && ent.neighborDist !== neighborDist) ||
                          (neighborDist && !ent.neighborDist) || ent.adjList !== mesh.adjList || ent.numRegions !== mesh.numRegions);
    if (!ent || stale) {
        const xyz = r_xyz || (ent && ent.r_xyz);
        if (!xyz) throw new Error('orogen-hip: first call for this mesh needs r_xyz (call bindMesh(mesh, r_xyz, neighborDist))');
        const nd = neighborDist || (ent && ent.neighborDist) || null;
        const p = addon.planetCreate(defaultContext(), mesh.numRegions, mesh.adjOffset, mesh.adjList, xyz, nd);
        ent = { p, r_xyz: xyz, neighborDist: nd, adjList: mesh.adjList, numRegions: mesh.numRegions };
        planets.set(mesh, ent);
    }
    return ent.p;
}
export function bindMesh(mesh, r_xyz, neighborDist) { return planetFor(mesh, r_xyz, neighborDist); }

export default addon;
