// Golden-vector generator for smoothField (js/climate-util.js:5-25): runs the REFERENCE's own JavaScript (scratch copy
// of /root/reference/js) on a field and mesh handed in as raw arrays.  Build container only.  Test infrastructure.
//   node run_smooth_field.mjs <refJsDir> <job.json>
import fs from 'fs';
import path from 'path';
import { pathToFileURL } from 'url';

const refDir = process.argv[2];
const job = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));
function readArr(file, Type) {
    const buf = fs.readFileSync(file);
    return new Type(buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength));
}
async function main() {
    const CU = await import(pathToFileURL(path.join(refDir, 'climate-util.js')).href);
    const mesh = { numRegions: job.numRegions, adjOffset: readArr(job.adjOffset, Int32Array), adjList: readArr(job.adjList, Int32Array) };
    for (const c of job.cases) {
        const f = readArr(job.field, Float32Array);
        const ret = CU.smoothField(mesh, f, c.passes);
        if (ret !== undefined) throw new Error('smoothField returned a value');
        fs.writeFileSync(c.out, Buffer.from(f.buffer, f.byteOffset, f.byteLength));
    }
}
main().catch((e) => { console.error(e.stack || e); process.exit(1); });
