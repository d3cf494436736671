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
const planets = new WeakMap();
export function planetFor(mesh, r_xyz, neighborDist) {
    let p = planets.get(mesh);
    if (!p) {
        if (!r_xyz) throw new Error('orogen-hip: first call for this mesh needs r_xyz (call bindMesh(mesh, r_xyz, neighborDist))');
        p = addon.planetCreate(defaultContext(), mesh.numRegions, mesh.adjOffset, mesh.adjList, r_xyz, neighborDist || null);
        planets.set(mesh, p);
    }
    return p;
}
export function bindMesh(mesh, r_xyz, neighborDist) { return planetFor(mesh, r_xyz, neighborDist); }

export default addon;
