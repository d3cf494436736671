// Golden-vector generator for assignElevation / the config-1 pipeline: runs the REFERENCE's own
// JavaScript (scratch copy of /root/reference/js) through the sequence of handleGenerate
// (js/planet-worker.js:136-225) with a stub Delaunay provider that returns the build's triangulation
// (planar part: pole triangles removed, hull half-edges = -1), and dumps inputs + outputs.
// Build container only (Node 12, no GPU).  Test infrastructure.
//
//   node run_elevation.mjs <refJsDir> <job.json>
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
    const SN = await imp('simplex-noise.js');
    const SM = await imp('sphere-mesh.js');
    const CP = await imp('coarse-plates.js');
    const PL = await imp('plates.js');
    const SP = await imp('super-plates.js');
    const EL = await imp('elevation.js');
    const TP = await imp('terrain-post.js');

    // stub Delaunay: keyed by point count
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

    const { N, P, jitter, nMag, numContinents, seed, params } = job;
    const spread = 5;
    const out = job.out;
    const { mesh, r_xyz } = SM.buildSphere(N, jitter, RNG.makeRng(seed));
    const neighborDist = SM.computeNeighborDist(mesh, r_xyz);
    const co = CP.generateCoarsePlates(seed, P, numContinents, 0, 0.3);
    const r_plate = CP.projectCoarsePlates(mesh, r_xyz, co.coarseMesh, co.coarse_xyz, co.coarse_r_plate, seed, P);
    PL.smoothAndReconnectPlates(mesh, r_plate, co.coarsePlateSeeds, 3);
    const plateSeeds = co.coarsePlateSeeds, plateVec = co.coarsePlateVec, plateIsOcean = co.coarsePlateIsOcean;
    const plateDensity = {};
    for (const r of plateSeeds) {
        const drng = RNG.makeRng(r + 777);
        const dO = 3.0 + drng() * 0.5, dL = 2.4 + drng() * 0.5;
        plateDensity[r] = plateIsOcean.has(r) ? dO : dL;
    }
    const noise = new SN.SimplexNoise(seed);
    let superPlateData = null;
    if (P >= 8 && job.superPlates) superPlateData = SP.buildSuperPlates(mesh, r_plate, plateSeeds, plateVec, plateIsOcean, plateDensity);

    const res = EL.assignElevation(mesh, r_xyz, plateIsOcean, r_plate, plateVec, plateSeeds, noise, nMag, seed, spread, plateDensity, superPlateData);

    // ---- dump inputs ----
    writeArr(out + 'triangles.bin', mesh.triangles); writeArr(out + 'halfedges.bin', mesh.halfedges);
    writeArr(out + 'xyz.bin', r_xyz); writeArr(out + 'neighborDist.bin', neighborDist);
    writeArr(out + 'adjOffset.bin', mesh.adjOffset); writeArr(out + 'adjList.bin', mesh.adjList);
    writeArr(out + 'r_plate.bin', r_plate);
    const seedsArr = Int32Array.from(plateSeeds);
    writeArr(out + 'plateSeeds.bin', seedsArr);
    const pv = new Float64Array(4 * seedsArr.length), dens = new Float64Array(seedsArr.length), isOc = new Uint8Array(seedsArr.length);
    seedsArr.forEach((id, i) => { const v = plateVec[id]; pv.set([v.pole[0], v.pole[1], v.pole[2], v.omega], 4 * i); dens[i] = plateDensity[id]; isOc[i] = plateIsOcean.has(id) ? 1 : 0; });
    writeArr(out + 'plateVec.bin', pv); writeArr(out + 'plateDensity.bin', dens); writeArr(out + 'plateIsOcean.bin', isOc);
    const meta = { numRegions: mesh.numRegions, hasSuper: !!superPlateData, seed, nMag, spread, P };
    if (superPlateData) {
        const ns = superPlateData.numSuperPlates;
        meta.numSuperPlates = ns;
        writeArr(out + 'r_superPlate.bin', superPlateData.r_superPlate);
        const spv = new Float64Array(4 * ns), sd = new Float64Array(ns), so = new Uint8Array(ns);
        for (let s = 0; s < ns; s++) { const v = superPlateData.superPlateVec[s]; spv.set([v.pole[0], v.pole[1], v.pole[2], v.omega], 4 * s); sd[s] = superPlateData.superPlateDensity[s]; so[s] = superPlateData.superPlateIsOcean.has(s) ? 1 : 0; }
        writeArr(out + 'superPlateVec.bin', spv); writeArr(out + 'superPlateDensity.bin', sd); writeArr(out + 'superPlateIsOcean.bin', so);
    }
    // ---- dump outputs ----
    writeArr(out + 'ref_elevation.bin', res.r_elevation); writeArr(out + 'ref_stress.bin', res.r_stress);
    writeArr(out + 'ref_mountain.bin', Int32Array.from(res.mountain_r)); writeArr(out + 'ref_coastline.bin', Int32Array.from(res.coastline_r));
    writeArr(out + 'ref_ocean.bin', Int32Array.from(res.ocean_r));
    const layers = ['base', 'tectonic', 'noise', 'interior', 'coastal', 'ocean', 'hotspot', 'tecActivity', 'margins', 'backArc', 'foldRidge', 'orogenicPower'];
    meta.layers = layers;
    for (const l of layers) writeArr(out + 'ref_dl_' + l + '.bin', res.debugLayers[l]);
    // pair-intensity known answers are exercised through stress; also export a few intermediate-free checks
    // ---- config-1 end to end: runPostProcessing mapping (js/planet-worker.js:40-102) ----
    if (params) {
        const e = new Float32Array(res.r_elevation);
        const { smoothing, glacialErosion, hydraulicErosion, thermalErosion, ridgeSharpening, terrainWarp } = params;
        if (terrainWarp > 0) TP.warpTerrain(mesh, e, r_xyz, seed, terrainWarp, res.debugLayers.hotspot);
        const oc = new Uint8Array(mesh.numRegions);
        for (let r = 0; r < mesh.numRegions; r++) if (e[r] <= 0) oc[r] = 1;
        if (smoothing > 0) TP.smoothElevation(mesh, e, oc, Math.round(1 + smoothing * 4), 0.2 + smoothing * 0.5);
        if (glacialErosion > 0 || hydraulicErosion > 0 || thermalErosion > 0)
            TP.erodeComposite(mesh, e, r_xyz, oc, Math.round(hydraulicErosion * 20), hydraulicErosion * 0.0006, 0.5, 1.0, Math.round(thermalErosion * 10),
                              1.2 - thermalErosion * 0.4, thermalErosion * 0.15, Math.round(glacialErosion * 10), glacialErosion, neighborDist);
        if (ridgeSharpening > 0) TP.sharpenRidges(mesh, e, oc, Math.round(1 + ridgeSharpening * 3), ridgeSharpening * 0.08);
        TP.applySoilCreep(mesh, e, oc, 3, 0.1125);
        writeArr(out + 'ref_final_elevation.bin', e); writeArr(out + 'ref_final_isOcean.bin', oc);
    }
    meta.timing = res._timing;
    fs.writeFileSync(out + 'meta.json', JSON.stringify(meta));
}
main().catch((e) => { console.error(e.stack || e); process.exit(1); });
