// Test driver for the JavaScript host: runs jobs described in <dir>/jobs.json through the drop-in modules
// (planet_heightmap_generation_amd/js/*.js) exactly the way the reference's worker calls terrain-post.js.
import fs from 'fs';
import path from 'path';
import { fileURLToPath, pathToFileURL } from 'url';

const here = path.dirname(fileURLToPath(import.meta.url));
const jsDir = path.join(here, '..', '..', 'planet_heightmap_generation_amd', 'js');
const imp = (f) => import(pathToFileURL(path.join(jsDir, f)).href);
const dir = process.argv[2];

function readArr(file, Type) {
    const buf = fs.readFileSync(path.join(dir, file));
    return new Type(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength));
}
function writeArr(file, arr) { fs.writeFileSync(path.join(dir, file), Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength)); }

async function main() {
    const { jobs } = JSON.parse(fs.readFileSync(path.join(dir, 'jobs.json'), 'utf8'));
    const SN = await imp('simplex-noise.js');
    const SM = await imp('sphere-mesh.js');
    const native = (await imp('native.js')).default;
    const result = {};
    let TP = null, PP = null, mesh = null, xyz = null, nd = null;
    for (const j of jobs) {
        switch (j.op) {
        case 'exports':
            result.exports = Object.keys(native).sort();
            result.deviceCount = native.deviceCount();
            break;
        case 'noise_scalar': {
            const n = new SN.SimplexNoise(j.seed), p = readArr(j.points, Float64Array), cnt = p.length / 3;
            const o = new Float64Array(4 * cnt);
            for (let i = 0; i < cnt; i++) {
                const x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
                o[4 * i] = n.noise3D(x, y, z); o[4 * i + 1] = n.fbm(x, y, z); o[4 * i + 2] = n.ridgedFbm(x, y, z); o[4 * i + 3] = n.ridgedFbm(x, y, z, 3, 0.5);
            }
            writeArr(j.out, o); writeArr(j.out + '.perm', n.perm);
            break;
        }
        case 'noise_batch': {
            const n = new SN.SimplexNoise(j.seed), p = readArr(j.points, Float64Array);
            writeArr(j.out, n.evalBatch(j.kind, p, j.octaves, j.p0, j.p1, j.p2));
            break;
        }
        case 'build_sphere': {
            const b = SM.buildSphere(j.N, j.jitter, j.seed);
            writeArr(j.out + '.xyz', b.r_xyz); writeArr(j.out + '.tri', b.mesh.triangles); writeArr(j.out + '.he', b.mesh.halfedges);
            writeArr(j.out + '.off', b.mesh.adjOffset); writeArr(j.out + '.adj', b.mesh.adjList);
            writeArr(j.out + '.nd', SM.computeNeighborDist(b.mesh, b.r_xyz));
            break;
        }
        case 'device_must_throw':
            try { native.ctxCreate(0); result.threw = null; } catch (e) { result.threw = e.message; }
            break;
        case 'load_mesh':
            mesh = new SM.SphereMesh(readArr(j.tri, Int32Array), readArr(j.he, Int32Array), j.numRegions);
            xyz = readArr(j.xyz, Float32Array); nd = readArr(j.nd, Float32Array);
            TP = await imp('terrain-post.js'); PP = await imp('post-processing.js');
            break;
        case 'post': {
            const e = readArr(j.elevation, Float32Array);
            const oc = j.isOcean ? readArr(j.isOcean, Uint8Array) : null;
            const hot = j.hotspot ? readArr(j.hotspot, Float32Array) : undefined;
            const a = j.args;
            let ret;
            if (j.fn === 'warpTerrain') ret = TP.warpTerrain(mesh, e, xyz, a.seed, a.strength, hot);
            else if (j.fn === 'smoothElevation') ret = TP.smoothElevation(mesh, e, oc, a.iterations, a.strength);
            else if (j.fn === 'sharpenRidges') ret = TP.sharpenRidges(mesh, e, oc, a.iterations, a.strength);
            else if (j.fn === 'applySoilCreep') ret = TP.applySoilCreep(mesh, e, oc, a.iterations, a.strength);
            else if (j.fn === 'erodeComposite') ret = TP.erodeComposite(mesh, e, xyz, oc, a.hIters, a.K, a.m, a.dt, a.tIters, a.talusSlope, a.kThermal, a.gIters, a.glacialStrength, nd);
            else throw new Error('unknown fn ' + j.fn);
            if (ret !== undefined) throw new Error(j.fn + ' must return undefined');
            writeArr(j.out, e);
            break;
        }
        case 'pipeline': {
            const e = readArr(j.elevation, Float32Array);
            const hot = j.hotspot ? readArr(j.hotspot, Float32Array) : undefined;
            const r = PP.runPostProcessing(mesh, xyz, e, j.params, nd, j.seed, hot);
            writeArr(j.out, e); writeArr(j.out + '.delta', r.dl_erosionDelta);
            result.postTiming = r.postTiming.map((t) => t.stage);
            break;
        }
        case 'assign_elevation': {
            const EL = await imp('elevation.js');
            const ids = Array.from(readArr(j.plateSeeds, Int32Array));
            const pv = readArr(j.plateVec, Float64Array), pd = readArr(j.plateDensity, Float64Array), po = readArr(j.plateIsOcean, Uint8Array);
            const plateVec = {}, plateDensity = {}; const plateIsOcean = new Set();
            ids.forEach((id, i) => { plateVec[id] = { pole: [pv[4 * i], pv[4 * i + 1], pv[4 * i + 2]], omega: pv[4 * i + 3] }; plateDensity[id] = pd[i]; if (po[i]) plateIsOcean.add(id); });
            let sup = null;
            if (j.r_superPlate) {
                const sv = readArr(j.superPlateVec, Float64Array), sd = readArr(j.superPlateDensity, Float64Array), so = readArr(j.superPlateIsOcean, Uint8Array);
                sup = { r_superPlate: readArr(j.r_superPlate, Int32Array), superPlateVec: {}, superPlateDensity: {}, superPlateIsOcean: new Set() };
                for (let s2 = 0; s2 < sd.length; s2++) { sup.superPlateVec[s2] = { pole: [sv[4 * s2], sv[4 * s2 + 1], sv[4 * s2 + 2]], omega: sv[4 * s2 + 3] }; sup.superPlateDensity[s2] = sd[s2]; if (so[s2]) sup.superPlateIsOcean.add(s2); }
            }
            const r = EL.assignElevation(mesh, xyz, plateIsOcean, readArr(j.r_plate, Int32Array), plateVec, new Set(ids), new SN.SimplexNoise(j.seed), j.nMag, j.seed, j.spread, plateDensity, sup);
            writeArr(j.out + '.elev', r.r_elevation); writeArr(j.out + '.stress', r.r_stress); writeArr(j.out + '.hotspot', r.debugLayers.hotspot);
            writeArr(j.out + '.mountain', Int32Array.from(r.mountain_r)); writeArr(j.out + '.coastline', Int32Array.from(r.coastline_r)); writeArr(j.out + '.ocean', Int32Array.from(r.ocean_r));
            result.elevTiming = r._timing.map((t) => t.stage);
            result.elevKeys = Object.keys(r).sort(); result.layerKeys = Object.keys(r.debugLayers).sort();
            break;
        }
        case 'smooth_plates': {           // host stage: no GPU needed
            const PL = await imp('plates.js');
            const m = { numRegions: j.numRegions, adjOffset: readArr(j.off, Int32Array), adjList: readArr(j.adj, Int32Array) };
            const rp = readArr(j.r_plate, Int32Array);
            const ret = PL.smoothAndReconnectPlates(m, rp, new Set(Array.from(readArr(j.seeds, Int32Array))), j.passes);
            if (ret !== undefined) throw new Error('smoothAndReconnectPlates must return undefined');
            writeArr(j.out, rp);
            break;
        }
        case 'project_plates': {
            const CP = await imp('coarse-plates.js');
            const m = { numRegions: j.numRegions, adjOffset: readArr(j.off, Int32Array), adjList: readArr(j.adj, Int32Array) };
            const cm = { numRegions: j.coarseRegions, adjOffset: readArr(j.coff, Int32Array), adjList: readArr(j.cadj, Int32Array) };
            const rp = CP.projectCoarsePlates(m, readArr(j.xyz, Float32Array), cm, readArr(j.cxyz, Float32Array), readArr(j.cplate, Int32Array), j.seed, j.P);
            if (!(rp instanceof Int32Array)) throw new Error('projectCoarsePlates must return an Int32Array');
            writeArr(j.out, rp);
            break;
        }
        case 'smooth_field': {
            const CU = await imp('climate-util.js');
            const fld = readArr(j.field, Float32Array);
            const ret = CU.smoothField(mesh, fld, j.passes);
            if (ret !== undefined) throw new Error('smoothField must return undefined');
            writeArr(j.out, fld);
            break;
        }
        case 'climate_sweeps': {
            const CS = await imp('climate-sweeps.js');
            const rd = (k, T) => readArr(j.in[k], T);
            const land = rd('isLand', Uint8Array), warmth = rd('oceanWarmth', Float32Array), cont = rd('plateContinentality', Float32Array);
            const wx = rd('wind3dX', Float32Array), wy = rd('wind3dY', Float32Array), wz = rd('wind3dZ', Float32Array);
            writeArr(j.out.diffuse, CS.diffuseOceanWarmth(mesh, warmth, land, cont, j.passes));
            writeArr(j.out.diffuseNulls, CS.diffuseOceanWarmth(mesh, null, land, null, j.passesNulls));
            writeArr(j.out.convergence, CS.computeWindConvergence(mesh, xyz, wx, wy, wz));
            writeArr(j.out.advect, CS.advectMoisture(mesh, xyz, rd('heightKm', Float32Array), land, rd('windE', Float32Array), rd('windN', Float32Array),
                wx, wy, wz, warmth, rd('coastDistLand', Int32Array), j.maxHops, 123.0));
            const errs = [];
            try { CS.computeWindConvergence(mesh, xyz, wx.subarray(1), wy, wz); errs.push(null); } catch (ex) { errs.push(ex.constructor.name); }
            result.climateErrors = errs;
            break;
        }
        case 'comm_single_rank': {          // the RCCL communicator and both exchange shapes through the addon (one rank: RCCL refuses two on one GPU)
            const NT = await imp('native.js');
            const planet = NT.planetFor(mesh, xyz, nd);
            const e = readArr(j.elevation, Float32Array);
            native.planetUpload(planet, e, null);
            const id = native.commUniqueId();
            const comm = native.commCreate(NT.defaultContext(), id, 1, 0);
            const send = new Int32Array(1000).map((_, k) => 3 * k);
            native.planetSetHalo(planet, send, new Int32Array(0));
            native.planetExchangeAllgather(planet, comm, new Int32Array([send.length]));
            const errs = [];
            try { native.planetExchangeAllgather(planet, comm, new Int32Array([send.length, 1])); errs.push(null); } catch (ex) { errs.push(ex.constructor.name); }
            try { native.planetExchangeNeighbors(planet, comm, 0, 0); errs.push(null); } catch (ex) { errs.push(ex.constructor.name); }
            native.planetSetHalo(planet, new Int32Array(0), new Int32Array(0));
            native.planetExchangeNeighbors(planet, comm, 0, 0);
            const back = new Float32Array(e.length);
            native.planetDownload(planet, back);
            result.comm = { idBytes: id.length, idNonZero: id.some((b) => b !== 0), errors: errs, unchanged: back.every((v, k) => v === e[k]) };
            break;
        }
        case 'planet_destroy': {            // explicit release of a planet's device memory: the handle stays, dead
            const NT = await imp('native.js');
            const planet = native.planetCreate(NT.defaultContext(), mesh.numRegions, mesh.adjOffset, mesh.adjList, xyz, nd);
            const e = readArr(j.elevation, Float32Array);
            native.planetUpload(planet, e, null);
            const back = new Float32Array(e.length);
            native.planetDownload(planet, back);
            native.planetDestroy(planet);
            let after = null;
            try { native.planetDownload(planet, back); } catch (ex) { after = ex.constructor.name; }
            native.planetDestroy(planet);                                   // twice is harmless
            result.destroy = { roundTrip: back.every((v, k) => v === e[k]), afterDestroy: after };
            break;
        }
        case 'error_paths': {
            const errs = [];
            const e = readArr(j.elevation, Float32Array), oc = readArr(j.isOcean, Uint8Array);
            try { TP.erodeComposite(mesh, e, xyz, oc, 1, 3e-4, 0.5, 1, 0, 1.16, 0.015, 0, 0); errs.push(null); } catch (ex) { errs.push(ex.constructor.name); }
            try { TP.smoothElevation(mesh, new Float64Array(e.length), oc, 1, 0.5); errs.push(null); } catch (ex) { errs.push(ex.constructor.name); }
            try { TP.smoothElevation(mesh, e, oc.subarray(1), 1, 0.5); errs.push(null); } catch (ex) { errs.push(ex.constructor.name); }
            // a field from another (smaller) mesh: must be refused, not read and written out of bounds
            try { TP.applySoilCreep(mesh, e.subarray(1), oc.subarray(1), 1, 0.1); errs.push(null); } catch (ex) { errs.push(ex.constructor.name); }
            try { TP.warpTerrain(mesh, e, xyz, 1, 0.5, e.subarray(2)); errs.push(null); } catch (ex) { errs.push(ex.constructor.name); }
            result.errors = errs;
            break;
        }
        default: throw new Error('unknown op ' + j.op);
        }
    }
    fs.writeFileSync(path.join(dir, 'result.json'), JSON.stringify(result));
}
main().catch((e) => { console.error(e.stack || e); process.exit(1); });
