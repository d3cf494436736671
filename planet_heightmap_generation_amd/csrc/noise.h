// SimplexNoise device/host function library (reference: js/simplex-noise.js:5-54, js/rng.js:3-6).
// All arithmetic is IEEE double in the reference's left-to-right evaluation order; build with
// -ffp-contract=off so no multiply-add is fused.  P / M point at perm[512] / permMod12[512]
// (LDS on the device).
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define WO_HD __host__ __device__
#else
#define WO_HD
#endif

namespace wo {

// host: constructor tables (js/simplex-noise.js:8-14)
void noise_tables(double seed, uint8_t* perm512, uint8_t* pm12_512);

// gradient components in the order listed at js/simplex-noise.js:7, packed 2 bits each (0:-1 1:0 2:+1)
WO_HD inline void simplex_grad(int g, double& gx, double& gy, double& gz) {
    // [1,1,0],[-1,1,0],[1,-1,0],[-1,-1,0],[1,0,1],[-1,0,1],[1,0,-1],[-1,0,-1],[0,1,1],[0,-1,1],[0,1,-1],[0,-1,-1]
    const unsigned xs = 0x552222u;  // per-gradient x code, 2 bits each, g = 0 in the low bits
    const unsigned ys = 0x22550au;
    const unsigned zs = 0x0a0a55u;
    gx = (double)((int)((xs >> (2 * g)) & 3u) - 1);
    gy = (double)((int)((ys >> (2 * g)) & 3u) - 1);
    gz = (double)((int)((zs >> (2 * g)) & 3u) - 1);
}

WO_HD inline double simplex_corner(const uint8_t* M, int gi_index, double x, double y, double z) {
    double a = 0.6 - x * x - y * y - z * z;
    if (a > 0) {
        a *= a;
        double gx, gy, gz;
        simplex_grad(M[gi_index], gx, gy, gz);
        return a * a * (gx * x + gy * y + gz * z);
    }
    return 0.0;
}

WO_HD inline double noise3d(const uint8_t* P, const uint8_t* M, double x, double y, double z) {
    const double F = 1.0 / 3.0, H = 1.0 / 6.0;
    const double s = (x + y + z) * F;
    const double i = floor(x + s), j = floor(y + s), k = floor(z + s);
    const double t = (i + j + k) * H;
    const double x0 = x - i + t, y0 = y - j + t, z0 = z - k + t;
    int i1, j1, k1, i2, j2, k2;
    if (x0 >= y0) {
        if (y0 >= z0)      { i1 = 1; j1 = 0; k1 = 0; i2 = 1; j2 = 1; k2 = 0; }
        else if (x0 >= z0) { i1 = 1; j1 = 0; k1 = 0; i2 = 1; j2 = 0; k2 = 1; }
        else               { i1 = 0; j1 = 0; k1 = 1; i2 = 1; j2 = 0; k2 = 1; }
    } else {
        if (y0 < z0)       { i1 = 0; j1 = 0; k1 = 1; i2 = 0; j2 = 1; k2 = 1; }
        else if (x0 < z0)  { i1 = 0; j1 = 1; k1 = 0; i2 = 0; j2 = 1; k2 = 1; }
        else               { i1 = 0; j1 = 1; k1 = 0; i2 = 1; j2 = 1; k2 = 0; }
    }
    const double x1 = x0 - i1 + H, y1 = y0 - j1 + H, z1 = z0 - k1 + H;
    const double x2 = x0 - i2 + 2 * H, y2 = y0 - j2 + 2 * H, z2 = z0 - k2 + 2 * H;
    const double x3 = x0 - 1 + 3 * H, y3 = y0 - 1 + 3 * H, z3 = z0 - 1 + 3 * H;
    // i & 255 with JS ToInt32 semantics (|i| stays far below 2^31 on this path)
    const int ii = ((int)(long long)i) & 255, jj = ((int)(long long)j) & 255, kk = ((int)(long long)k) & 255;
    const double n0 = simplex_corner(M, ii + P[jj + P[kk]], x0, y0, z0);
    const double n1 = simplex_corner(M, ii + i1 + P[jj + j1 + P[kk + k1]], x1, y1, z1);
    const double n2 = simplex_corner(M, ii + i2 + P[jj + j2 + P[kk + k2]], x2, y2, z2);
    const double n3 = simplex_corner(M, ii + 1 + P[jj + 1 + P[kk + 1]], x3, y3, z3);
    return 32 * (n0 + n1 + n2 + n3);
}

WO_HD inline double fbm(const uint8_t* P, const uint8_t* M, double x, double y, double z,
                        int octaves = 5, double persistence = 2.0 / 3.0) {
    double sum = 0, mx = 0, amp = 1;
    for (int o = 0; o < octaves; ++o) {
        const double f = (double)(1 << o);
        sum += amp * noise3d(P, M, x * f, y * f, z * f);
        mx += amp;
        amp *= persistence;
    }
    return sum / mx;
}

WO_HD inline double ridged_fbm(const uint8_t* P, const uint8_t* M, double x, double y, double z,
                               int octaves = 6, double lacunarity = 2.0, double gain = 0.5, double offset = 1.0) {
    double sum = 0, freq = 1, amp = 1, prev = 1, maxVal = 0;
    for (int o = 0; o < octaves; ++o) {
        double n = noise3d(P, M, x * freq, y * freq, z * freq);
        n = offset - fabs(n);
        n = n * n;
        sum += n * amp * prev;
        maxVal += amp;
        prev = n < 1 ? n : 1;   // Math.min(n, 1) (n is never NaN here)
        freq *= lacunarity;
        amp *= gain;
    }
    return sum / maxVal;
}

}  // namespace wo
