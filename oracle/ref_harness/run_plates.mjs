// Golden-vector generator for the plate projection step (js/coarse-plates.js:51-117 projectCoarsePlates,
// js/plates.js:241-348 smoothAndReconnectPlates): runs the REFERENCE's own JavaScript (scratch copy of
// /root/reference/js) through the sequence of handleGenerate (js/planet-worker.js:150-173) with a stub Delaunay
// provider that returns the build's triangulations, and dumps the coarse plate table and both hi-res results.
// Build container only (Node 12, no GPU).  Test infrastructure.
//
//   node run_plates.mjs <refJsDir> <job.json>
import fs from 'fs';
import path from 'path';
import { performance } from 'perf_hooks';
import { pathToFileURL } from 'url';

globalThis.performance = performance;
const refDir = process.argv[2];
const job = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));

function readArr(file, Type) {
    const buf = fs.readFileSync(file);
    return new Type(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength));
}
function writeArr(file, arr) { fs.writeFileSync(file, Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength)); }

async function main() {
    const imp = (f) => import(pathToFileURL(path.join(refDir, f)).href);
    const RNG = await imp('rng.js');
    const SM = await imp('sphere-mesh.js');
    const CP = await imp('coarse-plates.js');
    const PL = await imp('plates.js');
    const tris = {};
    for (const t of job.triangulations) tris[t.n] = { triangles: readArr(t.triangles, Int32Array), halfedges: readArr(t.halfedges, Int32Array) };
    class StubDelaunator {
        constructor(flat) {
            const t = tris[flat.length / 2];
            if (!t) throw new Error('no triangulation for n=' + flat.length / 2);
            this.triangles = new Uint32Array(t.triangles); this.halfedges = t.halfedges;
        }
    }
    SM.setDelaunator(StubDelaunator);
    const { N, P, jitter, numContinents, seed, passes } = job;
    const out = job.out;
    const { mesh, r_xyz } = SM.buildSphere(N, jitter, RNG.makeRng(seed));
    const co = CP.generateCoarsePlates(seed, P, numContinents, 0, 0.3);
    let t0 = performance.now();
    const r_plate = CP.projectCoarsePlates(mesh, r_xyz, co.coarseMesh, co.coarse_xyz, co.coarse_r_plate, seed, P);
    const msProject = performance.now() - t0;
    writeArr(out + 'r_plate_projected.bin', r_plate);
    t0 = performance.now();
    PL.smoothAndReconnectPlates(mesh, r_plate, co.coarsePlateSeeds, passes);
    const msSmooth = performance.now() - t0;
    writeArr(out + 'r_plate_smoothed.bin', r_plate);
    writeArr(out + 'coarse_r_plate.bin', co.coarse_r_plate);
    writeArr(out + 'coarse_xyz.bin', co.coarse_xyz);
    writeArr(out + 'coarse_adjOffset.bin', co.coarseMesh.adjOffset);
    writeArr(out + 'coarse_adjList.bin', co.coarseMesh.adjList);
    writeArr(out + 'plateSeeds.bin', Int32Array.from(co.coarsePlateSeeds));
    writeArr(out + 'xyz.bin', r_xyz); writeArr(out + 'adjOffset.bin', mesh.adjOffset); writeArr(out + 'adjList.bin', mesh.adjList);
    fs.writeFileSync(out + 'meta.json', JSON.stringify({ numRegions: mesh.numRegions, coarseRegions: co.coarseMesh.numRegions, N, P, seed, passes, msProject, msSmooth }));
}
main().catch((e) => { console.error(e.stack || e); process.exit(1); });
