// SimplexNoise with the reference's constructor and methods (js/simplex-noise.js:5-54), all evaluated by the native
// library: the tables come from wo_noise_tables (bit-identical to the reference's Fisher-Yates over makeRng(seed)), the
// point-at-a-time methods the out-of-scope callers use (js/wind.js:394, js/coarse-plates.js:57) go to wo_noise_point
// on the host, and `evalBatch` evaluates many points on the device — one arithmetic (csrc/noise.h) everywhere.
import addon, { defaultContext } from './native.js';

const KIND = { noise3D: 0, fbm: 1, ridgedFbm: 2 };

export class SimplexNoise {
    constructor(seed = 0) {
        this.seed = seed;
        const t = addon.noiseTables(seed);
        this.perm = t.perm;
        this.pm12 = t.pm12;
    }

    noise3D(x, y, z) { return addon.noisePoint(this.perm, this.pm12, 0, 0, 0, 0, 0, x, y, z); }

    fbm(x, y, z, octaves = 5, persistence = 2 / 3) { return addon.noisePoint(this.perm, this.pm12, 1, octaves, persistence, 0, 0, x, y, z); }

    ridgedFbm(x, y, z, octaves = 6, lacunarity = 2.0, gain = 0.5, offset = 1.0) {
        return addon.noisePoint(this.perm, this.pm12, 2, octaves, lacunarity, gain, offset, x, y, z);
    }

    // kind: 'noise3D' | 'fbm' | 'ridgedFbm'; points: Float64Array of xyz triples -> Float64Array (device)
    evalBatch(kind, points, octaves, a, b, c) {
        const k = KIND[kind];
        if (k === undefined) throw new Error('evalBatch: unknown kind ' + kind);
        const oct = octaves === undefined ? (k === 2 ? 6 : 5) : octaves;
        const p0 = a === undefined ? (k === 2 ? 2.0 : 2 / 3) : a;
        return addon.noiseEval(defaultContext(), this.seed, k, oct, p0, b === undefined ? 0.5 : b, c === undefined ? 1.0 : c, points);
    }
}
