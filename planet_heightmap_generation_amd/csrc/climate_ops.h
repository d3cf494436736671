// Per-cell bodies of the climate sweeps — the CSR-Jacobi passes that follow the terrain path:
//   diffuseOceanWarmth      js/temperature.js:19-66      (one pass: tmp = mean of self + neighbours, deep-interior cells kept)
//   computeWindConvergence  js/precipitation.js:18-52
//   advectMoisture          js/precipitation.js:59-195   (start state, then maxHops upwind-gather sweeps, ping-pong)
// Double arithmetic on float32 loads, float32 stores, neighbours in adjacency order: bit-exact against the reference.
#pragma once
#include <cstdint>

#include "noise.h"

namespace wo {

struct ClimateMesh { int32_t N; const int32_t* off; const int32_t* adj; const float* xyz; };

WO_HD inline double cl_max(double a, double b) { return (a != a || b != b) ? (a + b) : (a > b ? a : b); }   // Math.max (NaN-propagating)
WO_HD inline double cl_min(double a, double b) { return (a != a || b != b) ? (a + b) : (a < b ? a : b); }

// js/temperature.js:27-31
WO_HD inline float warmth_seed_cell(const float* r_oceanWarmth, const uint8_t* r_isLand, int32_t r) {
    return (!r_isLand[r] && r_oceanWarmth) ? r_oceanWarmth[r] : 0.0f;
}
// js/temperature.js:36-50, one pass
WO_HD inline float warmth_diffuse_cell(const ClimateMesh& M, const float* coastal, const float* r_plateContinentality, int32_t r) {
    if (r_plateContinentality && (double)r_plateContinentality[r] >= 0.95) return coastal[r];
    double sum = coastal[r];
    int32_t count = 1;
    for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni) { sum += (double)coastal[M.adj[ni]]; ++count; }
    return (float)(sum / count);
}

// js/precipitation.js:24-49
WO_HD inline float wind_convergence_cell(const ClimateMesh& M, const float* wx, const float* wy, const float* wz, int32_t r) {
    const double wdx = wx[r], wdy = wy[r], wdz = wz[r];
    const double px = M.xyz[3 * r], py = M.xyz[3 * r + 1], pz = M.xyz[3 * r + 2];
    double conv = 0; int32_t count = 0;
    for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni) {
        const int32_t nb = M.adj[ni];
        const double dx = (double)M.xyz[3 * nb] - px, dy = (double)M.xyz[3 * nb + 1] - py, dz = (double)M.xyz[3 * nb + 2] - pz;
        conv -= ((double)wx[nb] + wdx) * dx + ((double)wy[nb] + wdy) * dy + ((double)wz[nb] + wdz) * dz;
        ++count;
    }
    return count > 0 ? (float)(conv / count) : 0.0f;
}

// js/precipitation.js:70-119: start moisture of cell r
WO_HD inline float moisture_seed_cell(const ClimateMesh& M, const uint8_t* r_isLand, const float* wx, const float* wy, const float* wz,
                                      const float* r_oceanWarmth, const int32_t* r_coastDistLand, int32_t r) {
    if (!r_isLand[r]) {
        const double warmth = r_oceanWarmth ? (double)r_oceanWarmth[r] : 0;
        return (float)(0.4 + 0.35 * cl_max(0, warmth));
    }
    if (r_coastDistLand[r] != 0) return 0.0f;
    double warmthSum = 0, ox = 0, oy = 0, oz = 0; int32_t oceanCount = 0;
    const double px = M.xyz[3 * r], py = M.xyz[3 * r + 1], pz = M.xyz[3 * r + 2];
    for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni) {
        const int32_t nb = M.adj[ni];
        if (!r_isLand[nb]) {
            ++oceanCount;
            if (r_oceanWarmth) warmthSum += (double)r_oceanWarmth[nb];
            ox += (double)M.xyz[3 * nb] - px; oy += (double)M.xyz[3 * nb + 1] - py; oz += (double)M.xyz[3 * nb + 2] - pz;
        }
    }
    if (oceanCount == 0) return 0.0f;
    const double avgWarmth = warmthSum / oceanCount;
    const double windDotOcean = (double)wx[r] * ox + (double)wy[r] * oy + (double)wz[r] * oz;
    const double onshore = windDotOcean < 0 ? 1.0 : 0.25;
    const double warmthFactor = 0.5 + 0.5 * cl_max(-0.8, cl_min(1, avgWarmth));
    return (float)(onshore * warmthFactor);
}

// js/precipitation.js:129-182: one sweep, cell r
WO_HD inline float moisture_advect_cell(const ClimateMesh& M, const float* src, const float* r_heightKm, const uint8_t* r_isLand,
                                        const float* r_windE, const float* r_windN, const float* wx, const float* wy, const float* wz,
                                        int32_t maxHops, double depletionBase, int32_t r) {
    if (!r_isLand[r]) return src[r];
    const double we = r_windE[r], wn = r_windN[r];
    if (we * we + wn * wn < 1e-6) return src[r];
    double upwindMoisture = 0, upwindWeight = 0, upwindHeightSum = 0;
    const double heightHere = r_heightKm[r];
    const double px = M.xyz[3 * r], py = M.xyz[3 * r + 1], pz = M.xyz[3 * r + 2];
    for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni) {
        const int32_t nb = M.adj[ni];
        const double dx = px - (double)M.xyz[3 * nb], dy = py - (double)M.xyz[3 * nb + 1], dz = pz - (double)M.xyz[3 * nb + 2];
        const double dot = (double)wx[nb] * dx + (double)wy[nb] * dy + (double)wz[nb] * dz;
        if (dot > 0) {
            upwindMoisture += (double)src[nb] * dot;
            upwindHeightSum += (double)r_heightKm[nb] * dot;
            upwindWeight += dot;
        }
    }
    if (!(upwindWeight > 0)) return src[r];
    const double incoming = upwindMoisture / upwindWeight;
    const double upwindHeight = upwindHeightSum / upwindWeight;
    const double heightGain = cl_max(0, heightHere - upwindHeight);
    const double normalizedGain = heightGain * maxHops;
    const double elevDepletion = cl_min(0.8, normalizedGain * 0.55);
    const double depletion = depletionBase + elevDepletion;
    const double carried = incoming * cl_max(0, 1 - depletion);
    return (float)cl_max((double)src[r], carried);
}

}  // namespace wo
