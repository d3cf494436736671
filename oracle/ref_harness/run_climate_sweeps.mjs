// Golden-vector generator for the climate sweeps: diffuseOceanWarmth (js/temperature.js:19-66), computeWindConvergence
// (js/precipitation.js:18-52) and advectMoisture (js/precipitation.js:59-195).  Runs the REFERENCE's own JavaScript
// (scratch copy of /root/reference/js; make_golden_climate.py appends `export { ... }` lines for these module-private
// functions to the scratch copies).  Build container only.  Test infrastructure.
//   node run_climate_sweeps.mjs <refJsDir> <job.json>
import fs from 'fs';
import path from 'path';
import { pathToFileURL } from 'url';
import { createRequire } from 'module';

const require = createRequire(import.meta.url);
globalThis.performance = require('perf_hooks').performance;
const refDir = process.argv[2];
const job = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));
function readArr(file, Type) {
    const buf = fs.readFileSync(file);
    return new Type(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength));
}
function writeArr(file, a) { fs.writeFileSync(file, Buffer.from(a.buffer, a.byteOffset, a.byteLength)); }
async function main() {
    const T = await import(pathToFileURL(path.join(refDir, 'temperature.js')).href);
    const P = await import(pathToFileURL(path.join(refDir, 'precipitation.js')).href);
    const i = job.in;
    const mesh = { numRegions: job.numRegions, adjOffset: readArr(i.adjOffset, Int32Array), adjList: readArr(i.adjList, Int32Array) };
    const xyz = readArr(i.xyz, Float32Array), isLand = readArr(i.isLand, Uint8Array), warmth = readArr(i.oceanWarmth, Float32Array);
    const cont = readArr(i.plateContinentality, Float32Array), we = readArr(i.windE, Float32Array), wn = readArr(i.windN, Float32Array);
    const wx = readArr(i.wind3dX, Float32Array), wy = readArr(i.wind3dY, Float32Array), wz = readArr(i.wind3dZ, Float32Array);
    const hk = readArr(i.heightKm, Float32Array), cd = readArr(i.coastDistLand, Int32Array);
    for (const c of job.cases) {
        let out;
        if (c.fn === 'diffuseOceanWarmth') out = T.diffuseOceanWarmth(mesh, c.noWarmth ? null : warmth, isLand, c.noCont ? null : cont, c.passes);
        else if (c.fn === 'computeWindConvergence') out = P.computeWindConvergence(mesh, xyz, wx, wy, wz);
        else if (c.fn === 'advectMoisture') out = P.advectMoisture(mesh, xyz, hk, isLand, we, wn, wx, wy, wz, c.noWarmth ? null : warmth, cd, c.maxHops, 0);
        else throw new Error('unknown fn ' + c.fn);
        if (!(out instanceof Float32Array)) throw new Error(c.fn + ' did not return a Float32Array');
        writeArr(c.out, out);
    }
}
main().catch((e) => { console.error(e.stack || e); process.exit(1); });
