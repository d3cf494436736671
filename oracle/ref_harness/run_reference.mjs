// Golden-vector generator: runs the REFERENCE's own JavaScript (imported from a scratch copy of
// /root/reference/js made by make_golden.py — never from inside this repository) on inputs written by
// make_golden.py and dumps the outputs as raw little-endian typed arrays.
//
// Runs only in the build container (Node 12, no GPU).  Test infrastructure, not product code.
//
//   node run_reference.mjs <refJsDir> <jobFile.json>
//
// Job file: { "jobs": [ { "op": ..., ...params, "in": {name: path}, "out": {name: path} } ] }
import fs from 'fs';
import path from 'path';
import { performance } from 'perf_hooks';
import { pathToFileURL } from 'url';

globalThis.performance = performance;   // js/elevation.js:220 expects the browser global

const refDir = process.argv[2];
const jobFile = process.argv[3];

function readArr(file, Type) {
    const buf = fs.readFileSync(file);
    const ab = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
    return new Type(ab);
}
function writeArr(file, arr) {
    fs.writeFileSync(file, Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength));
}

async function main() {
    const imp = (f) => import(pathToFileURL(path.join(refDir, f)).href);
    const RNG = await imp('rng.js');
    const SN = await imp('simplex-noise.js');
    const SM = await imp('sphere-mesh.js');
    const TP = await imp('terrain-post.js');           // the scratch copy additionally exports priorityFloodCarve
    const { jobs } = JSON.parse(fs.readFileSync(jobFile, 'utf8'));

    const meshCache = {};
    function loadMesh(j) {
        const key = j.in.triangles;
        if (!meshCache[key]) {
            const tri = readArr(j.in.triangles, Int32Array), he = readArr(j.in.halfedges, Int32Array);
            meshCache[key] = new SM.SphereMesh(tri, he, j.numRegions);
        }
        return meshCache[key];
    }

    for (const j of jobs) {
        const t0 = performance.now();
        switch (j.op) {
        case 'rng': {
            const r = RNG.makeRng(j.seed), out = new Float64Array(j.count);
            for (let i = 0; i < j.count; i++) out[i] = r();
            writeArr(j.out.values, out);
            const ri = RNG.makeRandInt(j.seed), oi = new Int32Array(j.count);
            for (let i = 0; i < j.count; i++) oi[i] = ri(j.n);
            writeArr(j.out.ints, oi);
            break;
        }
        case 'noise': {
            const n = new SN.SimplexNoise(j.seed);
            writeArr(j.out.perm, n.perm); writeArr(j.out.pm12, n.pm12);
            const p = readArr(j.in.points, Float64Array), cnt = p.length / 3;
            const o = { noise3D: new Float64Array(cnt), fbm5: new Float64Array(cnt), fbm4h: new Float64Array(cnt),
                        ridged6: new Float64Array(cnt), ridged3h: new Float64Array(cnt), fbm2: new Float64Array(cnt),
                        ridged4: new Float64Array(cnt) };
            for (let i = 0; i < cnt; i++) {
                const x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
                o.noise3D[i] = n.noise3D(x, y, z);
                o.fbm5[i] = n.fbm(x, y, z);
                o.fbm4h[i] = n.fbm(x, y, z, 4, 0.5);
                o.fbm2[i] = n.fbm(x, y, z, 2);
                o.ridged6[i] = n.ridgedFbm(x, y, z);
                o.ridged3h[i] = n.ridgedFbm(x, y, z, 3, 0.5);
                o.ridged4[i] = n.ridgedFbm(x, y, z, 4);
            }
            for (const k of Object.keys(o)) writeArr(j.out[k], o[k]);
            break;
        }
        case 'points': {
            const xyz = SM.generateFibonacciSphere(j.N, j.jitter, RNG.makeRng(j.seed));
            writeArr(j.out.xyz, xyz);
            break;
        }
        case 'csr': {       // SphereMesh constructor + computeNeighborDist on the build's triangulation
            const mesh = loadMesh(j);
            writeArr(j.out.adjOffset, mesh.adjOffset); writeArr(j.out.adjList, mesh.adjList);
            writeArr(j.out.adjTriList, mesh._adjTriList);
            const xyz = readArr(j.in.xyz, Float32Array);
            writeArr(j.out.neighborDist, SM.computeNeighborDist(mesh, xyz));
            break;
        }
        case 'synthetic': { // SURVEY 8(d) bench terrain
            const n = new SN.SimplexNoise(j.seed);
            const xyz = readArr(j.in.xyz, Float32Array), N = xyz.length / 3;
            const e = new Float32Array(N);
            for (let r = 0; r < N; r++) {
                const x = xyz[3 * r], y = xyz[3 * r + 1], z = xyz[3 * r + 2];
                const f = n.fbm(x * 1.5, y * 1.5, z * 1.5, 5);
                e[r] = 0.9 * f - 0.12 + 0.25 * n.ridgedFbm(x * 3, y * 3, z * 3, 4) * Math.max(0, f);
            }
            writeArr(j.out.elevation, e);
            break;
        }
        case 'post': {      // one terrain-post export (or the scratch-exported priorityFloodCarve)
            const mesh = loadMesh(j);
            const xyz = readArr(j.in.xyz, Float32Array);
            const elev = readArr(j.in.elevation, Float32Array);
            const isOcean = j.in.isOcean ? readArr(j.in.isOcean, Uint8Array) : null;
            const nd = j.in.neighborDist ? readArr(j.in.neighborDist, Float32Array) : null;
            const hot = j.in.hotspot ? readArr(j.in.hotspot, Float32Array) : undefined;
            const a = j.args;
            switch (j.fn) {
            case 'warpTerrain': TP.warpTerrain(mesh, elev, xyz, a.seed, a.strength, hot); break;
            case 'smoothElevation': TP.smoothElevation(mesh, elev, isOcean, a.iterations, a.strength); break;
            case 'sharpenRidges': TP.sharpenRidges(mesh, elev, isOcean, a.iterations, a.strength); break;
            case 'applySoilCreep': TP.applySoilCreep(mesh, elev, isOcean, a.iterations, a.strength); break;
            case 'erodeComposite':
                TP.erodeComposite(mesh, elev, xyz, isOcean, a.hIters, a.K, a.m, a.dt, a.tIters, a.talusSlope,
                                  a.kThermal, a.gIters, a.glacialStrength, nd);
                break;
            case 'priorityFloodCarve': TP.priorityFloodCarve(mesh, elev, isOcean, a.carveStrength); break;
            default: throw new Error('unknown fn ' + j.fn);
            }
            writeArr(j.out.elevation, elev);
            break;
        }
        default: throw new Error('unknown op ' + j.op);
        }
        if (j.label) console.error(`[ref] ${j.label}: ${(performance.now() - t0).toFixed(1)} ms`);
    }
}
main().catch((e) => { console.error(e.stack || e); process.exit(1); });
