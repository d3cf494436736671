/*
 * ORACLE (test infrastructure only — see wo_oracle.h): serial C restatement of js/rng.js and
 * js/simplex-noise.js.  Parity pinned against tests/golden/noise_seed*.npz and rng_seed*.npz.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "wo_oracle.h"

/* js/rng.js:3-6: s = (|floor(seed*9301+49297)| % 2147483646) + 1; s = s*16807 % 2147483647 */
void wo_or_rng_seed(double seed, double* state) {
    *state = fmod(fabs(floor(seed * 9301.0 + 49297.0)), 2147483646.0) + 1.0;
}
double wo_or_rng_next(double* state) {
    *state = fmod(*state * 16807.0, 2147483647.0);
    return (*state - 1.0) / 2147483646.0;
}

/* js/simplex-noise.js:7 */
static const double GRAD[12][3] = {
    {1, 1, 0}, {-1, 1, 0}, {1, -1, 0}, {-1, -1, 0}, {1, 0, 1}, {-1, 0, 1},
    {1, 0, -1}, {-1, 0, -1}, {0, 1, 1}, {0, -1, 1}, {0, 1, -1}, {0, -1, -1}};

/* js/simplex-noise.js:8-14 */
void wo_or_noise_init(double seed, uint8_t* perm, uint8_t* pm12) {
    double st;
    uint8_t p[256];
    int i;
    wo_or_rng_seed(seed, &st);
    for (i = 0; i < 256; i++) p[i] = (uint8_t)i;
    for (i = 255; i > 0; i--) {
        int j = (int)floor(wo_or_rng_next(&st) * (i + 1));
        uint8_t t = p[i]; p[i] = p[j]; p[j] = t;
    }
    for (i = 0; i < 512; i++) { perm[i] = p[i & 255]; pm12[i] = perm[i] % 12; }
}

/* js/simplex-noise.js:17-32 */
double wo_or_noise3d(const uint8_t* P, const uint8_t* M, double x, double y, double z) {
    const double F = 1.0 / 3.0, H = 1.0 / 6.0, s = (x + y + z) * F;
    const double i = floor(x + s), j = floor(y + s), k = floor(z + s);
    const double t = (i + j + k) * H, x0 = x - i + t, y0 = y - j + t, z0 = z - k + t;
    int i1, j1, k1, i2, j2, k2;
    if (x0 >= y0) {
        if (y0 >= z0) { i1 = 1; j1 = 0; k1 = 0; i2 = 1; j2 = 1; k2 = 0; }
        else if (x0 >= z0) { i1 = 1; j1 = 0; k1 = 0; i2 = 1; j2 = 0; k2 = 1; }
        else { i1 = 0; j1 = 0; k1 = 1; i2 = 1; j2 = 0; k2 = 1; }
    } else {
        if (y0 < z0) { i1 = 0; j1 = 0; k1 = 1; i2 = 0; j2 = 1; k2 = 1; }
        else if (x0 < z0) { i1 = 0; j1 = 1; k1 = 0; i2 = 0; j2 = 1; k2 = 1; }
        else { i1 = 0; j1 = 1; k1 = 0; i2 = 1; j2 = 1; k2 = 0; }
    }
    {
        const double x1 = x0 - i1 + H, y1 = y0 - j1 + H, z1 = z0 - k1 + H;
        const double x2 = x0 - i2 + 2 * H, y2 = y0 - j2 + 2 * H, z2 = z0 - k2 + 2 * H;
        const double x3 = x0 - 1 + 3 * H, y3 = y0 - 1 + 3 * H, z3 = z0 - 1 + 3 * H;
        /* ToInt32(i) & 255 */
        const int ii = (int)((long long)i & 255), jj = (int)((long long)j & 255), kk = (int)((long long)k & 255);
        double n0 = 0, n1 = 0, n2 = 0, n3 = 0;
        double a = 0.6 - x0 * x0 - y0 * y0 - z0 * z0;
        double b, c, d;
        if (a > 0) { const double* v = GRAD[M[ii + P[jj + P[kk]]]]; a *= a; n0 = a * a * (v[0] * x0 + v[1] * y0 + v[2] * z0); }
        b = 0.6 - x1 * x1 - y1 * y1 - z1 * z1;
        if (b > 0) { const double* v = GRAD[M[ii + i1 + P[jj + j1 + P[kk + k1]]]]; b *= b; n1 = b * b * (v[0] * x1 + v[1] * y1 + v[2] * z1); }
        c = 0.6 - x2 * x2 - y2 * y2 - z2 * z2;
        if (c > 0) { const double* v = GRAD[M[ii + i2 + P[jj + j2 + P[kk + k2]]]]; c *= c; n2 = c * c * (v[0] * x2 + v[1] * y2 + v[2] * z2); }
        d = 0.6 - x3 * x3 - y3 * y3 - z3 * z3;
        if (d > 0) { const double* v = GRAD[M[ii + 1 + P[jj + 1 + P[kk + 1]]]]; d *= d; n3 = d * d * (v[0] * x3 + v[1] * y3 + v[2] * z3); }
        return 32 * (n0 + n1 + n2 + n3);
    }
}

/* js/simplex-noise.js:34-38 */
double wo_or_fbm(const uint8_t* P, const uint8_t* M, double x, double y, double z, int octaves, double persistence) {
    double sum = 0, max = 0, amp = 1;
    int o;
    for (o = 0; o < octaves; o++) {
        const double f = (double)(1 << o);
        sum += amp * wo_or_noise3d(P, M, x * f, y * f, z * f);
        max += amp;
        amp *= persistence;
    }
    return sum / max;
}

/* js/simplex-noise.js:40-53 */
double wo_or_ridged(const uint8_t* P, const uint8_t* M, double x, double y, double z, int octaves,
                    double lacunarity, double gain, double offset) {
    double sum = 0, freq = 1, amp = 1, prev = 1, maxVal = 0;
    int o;
    for (o = 0; o < octaves; o++) {
        double n = wo_or_noise3d(P, M, x * freq, y * freq, z * freq);
        n = offset - fabs(n);
        n = n * n;
        sum += n * amp * prev;
        maxVal += amp;
        prev = n < 1 ? n : 1;
        freq *= lacunarity;
        amp *= gain;
    }
    return sum / maxVal;
}

void wo_or_noise_batch(double seed, int kind, int octaves, double p0, double p1, double p2,
                       int64_t n, const double* xyz, double* out) {
    uint8_t P[512], M[512];
    int64_t i;
    wo_or_noise_init(seed, P, M);
    for (i = 0; i < n; i++) {
        const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (kind == 0) out[i] = wo_or_noise3d(P, M, x, y, z);
        else if (kind == 1) out[i] = wo_or_fbm(P, M, x, y, z, octaves, p0);
        else out[i] = wo_or_ridged(P, M, x, y, z, octaves, p0, p1, p2);
    }
}

/* e = 0.9*fbm(1.5p,5) - 0.12 + 0.25*ridged(3p,4)*max(0,fbm(1.5p,5)), SimplexNoise(seed) */
void wo_or_synthetic_terrain(int32_t N, const float* xyz, double seed, float* e) {
    uint8_t P[512], M[512];
    int32_t r;
    wo_or_noise_init(seed, P, M);
    for (r = 0; r < N; r++) {
        const double x = xyz[3 * r], y = xyz[3 * r + 1], z = xyz[3 * r + 2];
        const double f = wo_or_fbm(P, M, x * 1.5, y * 1.5, z * 1.5, 5, 2.0 / 3.0);
        const double rg = wo_or_ridged(P, M, x * 3, y * 3, z * 3, 4, 2.0, 0.5, 1.0);
        e[r] = (float)(0.9 * f - 0.12 + 0.25 * rg * (f > 0 ? f : 0));
    }
}
