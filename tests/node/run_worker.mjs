// Test driver for planet_heightmap_generation_amd/js/planet-worker.js: retain -> reapply (twice) -> error paths.
//   node run_worker.mjs <dir>   (reads <dir>/worker_job.json, writes <dir>/worker_result.json and the result arrays)
import fs from 'fs';
import path from 'path';
import { fileURLToPath } from 'url';
import { Worker } from 'worker_threads';

const here = path.dirname(fileURLToPath(import.meta.url));
const workerFile = path.join(here, '..', '..', 'planet_heightmap_generation_amd', 'js', 'planet-worker.js');
const dir = process.argv[2];
const job = JSON.parse(fs.readFileSync(path.join(dir, 'worker_job.json'), 'utf8'));
function readArr(file, Type) {
    const buf = fs.readFileSync(path.join(dir, file));
    return new Type(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength));
}
function writeArr(file, arr) { fs.writeFileSync(path.join(dir, file), Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength)); }

const w = new Worker(workerFile);
const log = [];
let waiting = null;
w.on('message', (m) => {
    if (m.type === 'progress') { log.push({ type: 'progress', pct: m.pct, label: m.label }); return; }
    if (waiting) { const f = waiting; waiting = null; f(m); }
});
w.on('error', (e) => { console.error(e.stack || e); process.exit(1); });
const ask = (msg, transfer) => new Promise((resolve) => { waiting = resolve; w.postMessage(msg, transfer || []); });

async function main() {
    const out = {};
    out.beforeRetain = await ask({ cmd: 'reapply', ...job.params });
    out.unknown = await ask({ cmd: 'frobnicate' });
    out.hostStage = await ask({ cmd: 'generate' });
    const mesh = { numRegions: job.numRegions, adjOffset: readArr(job.adjOffset, Int32Array), adjList: readArr(job.adjList, Int32Array), triangles: readArr(job.triangles, Int32Array) };
    const retained = await ask({ cmd: 'retain', mesh, r_xyz: readArr(job.xyz, Float32Array), neighborDist: readArr(job.neighborDist, Float32Array),
                                 prePostElev: readArr(job.elevation, Float32Array), seed: job.seed, r_hotspot: job.hotspot ? readArr(job.hotspot, Float32Array) : null });
    out.retained = retained;
    const r1 = await ask({ cmd: 'reapply', ...job.params });
    out.first = { type: r1.type, skipClimate: r1.skipClimate, keys: Object.keys(r1).sort(), postTiming: (r1._postTiming || []).map((s) => s.stage),
                  timingKeys: Object.keys(r1._reapplyTiming || {}).sort(), n: r1.r_elevation ? r1.r_elevation.length : 0, nt: r1.t_elevation ? r1.t_elevation.length : 0 };
    if (r1.type === 'reapplyDone') { writeArr('w_elev1.bin', r1.r_elevation); writeArr('w_tri1.bin', r1.t_elevation); writeArr('w_delta1.bin', r1.erosionDelta); }
    const r2 = await ask({ cmd: 'reapply', ...job.params2 });                // other sliders, same retained state: nothing is re-uploaded
    if (r2.type === 'reapplyDone') writeArr('w_elev2.bin', r2.r_elevation);
    const r3 = await ask({ cmd: 'reapply', ...job.params });                 // and back: must reproduce the first result
    if (r3.type === 'reapplyDone') writeArr('w_elev3.bin', r3.r_elevation);
    out.reapplyMs = [r1, r2, r3].map((r) => r._reapplyTiming ? r._reapplyTiming.workerTotal : null);
    out.disposed = await ask({ cmd: 'dispose' });
    out.afterDispose = await ask({ cmd: 'reapply', ...job.params });
    out.progress = log;
    fs.writeFileSync(path.join(dir, 'worker_result.json'), JSON.stringify(out));
    await w.terminate();
}
main().catch((e) => { console.error(e.stack || e); process.exit(1); });
