// SimplexNoise with the reference's constructor and methods (js/simplex-noise.js:5-54).  The permutation
// tables come from the native library (bit-identical to the reference's Fisher-Yates over makeRng(seed));
// `evalBatch` evaluates many points on the device with the same arithmetic the HIP passes use.  The scalar
// methods stay in JavaScript because the out-of-scope callers (js/wind.js:394, js/coarse-plates.js:57)
// call them one point at a time — host code stays JavaScript (north_star).
import addon, { defaultContext } from './native.js';

const F3 = 1 / 3, H = 1 / 6;

export class SimplexNoise {
    constructor(seed = 0) {
        this.seed = seed;
        this.G = [[1,1,0],[-1,1,0],[1,-1,0],[-1,-1,0],[1,0,1],[-1,0,1],[1,0,-1],[-1,0,-1],[0,1,1],[0,-1,1],[0,1,-1],[0,-1,-1]];
        const t = addon.noiseTables(seed);
        this.perm = t.perm;
        this.pm12 = t.pm12;
    }

    // kind: 'noise3D' | 'fbm' | 'ridgedFbm'; points: Float64Array of xyz triples -> Float64Array (device)
    evalBatch(kind, points, octaves, a, b, c) {
        const k = { noise3D: 0, fbm: 1, ridgedFbm: 2 }[kind];
        if (k === undefined) throw new Error('evalBatch: unknown kind ' + kind);
        const oct = octaves === undefined ? (k === 2 ? 6 : 5) : octaves;
        const p0 = a === undefined ? (k === 2 ? 2.0 : 2 / 3) : a;
        return addon.noiseEval(defaultContext(), this.seed, k, oct, p0, b === undefined ? 0.5 : b, c === undefined ? 1.0 : c, points);
    }

    noise3D(x, y, z) {
        const s = (x + y + z) * F3;
        const i = Math.floor(x + s), j = Math.floor(y + s), k = Math.floor(z + s);
        const t = (i + j + k) * H, x0 = x - i + t, y0 = y - j + t, z0 = z - k + t;
        let i1, j1, k1, i2, j2, k2;
        if (x0 >= y0) {
            if (y0 >= z0) { i1 = 1; j1 = 0; k1 = 0; i2 = 1; j2 = 1; k2 = 0; }
            else if (x0 >= z0) { i1 = 1; j1 = 0; k1 = 0; i2 = 1; j2 = 0; k2 = 1; }
            else { i1 = 0; j1 = 0; k1 = 1; i2 = 1; j2 = 0; k2 = 1; }
        } else {
            if (y0 < z0) { i1 = 0; j1 = 0; k1 = 1; i2 = 0; j2 = 1; k2 = 1; }
            else if (x0 < z0) { i1 = 0; j1 = 1; k1 = 0; i2 = 0; j2 = 1; k2 = 1; }
            else { i1 = 0; j1 = 1; k1 = 0; i2 = 1; j2 = 1; k2 = 0; }
        }
        const P = this.perm, M = this.pm12, g = this.G;
        const ii = i & 255, jj = j & 255, kk = k & 255;
        const corner = (gi, cx, cy, cz) => {
            let a = 0.6 - cx * cx - cy * cy - cz * cz;
            if (a <= 0) return 0;
            a *= a;
            const v = g[M[gi]];
            return a * a * (v[0] * cx + v[1] * cy + v[2] * cz);
        };
        const n0 = corner(ii + P[jj + P[kk]], x0, y0, z0);
        const n1 = corner(ii + i1 + P[jj + j1 + P[kk + k1]], x0 - i1 + H, y0 - j1 + H, z0 - k1 + H);
        const n2 = corner(ii + i2 + P[jj + j2 + P[kk + k2]], x0 - i2 + 2 * H, y0 - j2 + 2 * H, z0 - k2 + 2 * H);
        const n3 = corner(ii + 1 + P[jj + 1 + P[kk + 1]], x0 - 1 + 3 * H, y0 - 1 + 3 * H, z0 - 1 + 3 * H);
        return 32 * (n0 + n1 + n2 + n3);
    }

    fbm(x, y, z, octaves = 5, persistence = 2 / 3) {
        let sum = 0, max = 0, amp = 1;
        for (let o = 0; o < octaves; o++) {
            const f = 1 << o;
            sum += amp * this.noise3D(x * f, y * f, z * f);
            max += amp;
            amp *= persistence;
        }
        return sum / max;
    }

    ridgedFbm(x, y, z, octaves = 6, lacunarity = 2.0, gain = 0.5, offset = 1.0) {
        let sum = 0, freq = 1, amp = 1, prev = 1, maxVal = 0;
        for (let o = 0; o < octaves; o++) {
            let n = offset - Math.abs(this.noise3D(x * freq, y * freq, z * freq));
            n = n * n;
            sum += n * amp * prev;
            maxVal += amp;
            prev = Math.min(n, 1);
            freq *= lacunarity;
            amp *= gain;
        }
        return sum / maxVal;
    }
}
