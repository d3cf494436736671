/*
 * ORACLE (test infrastructure only — see wo_oracle.h): serial C restatement of the climate sweeps
 *   diffuseOceanWarmth      js/temperature.js:19-66
 *   computeWindConvergence  js/precipitation.js:18-52
 *   advectMoisture          js/precipitation.js:59-195
 * All arithmetic in double on float32 loads, every Float32Array store a (float) cast, loops in the reference's order.
 * Parity pinned against tests/golden/climate_sweeps_N10000_s1.npz (bit-exact).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "wo_oracle.h"

/* js/temperature.js:19-66; r_oceanWarmth / r_plateContinentality may be NULL */
void wo_or_diffuse_ocean_warmth(int32_t N, const int32_t* off, const int32_t* adj, const float* r_oceanWarmth, const uint8_t* r_isLand,
                                const float* r_plateContinentality, int32_t passes, float* coastal) {
    float* tmp = (float*)malloc(sizeof(float) * (size_t)N);
    int32_t r, pass, ni;
    for (r = 0; r < N; r++) coastal[r] = (!r_isLand[r] && r_oceanWarmth) ? r_oceanWarmth[r] : 0.0f;      /* :27-31 */
    for (pass = 0; pass < passes; pass++) {
        memcpy(tmp, coastal, sizeof(float) * (size_t)N);                                                 /* :35 */
        for (r = 0; r < N; r++) {
            double sum; int32_t count = 1;
            if (r_plateContinentality && (double)r_plateContinentality[r] >= 0.95) continue;             /* :38 */
            sum = coastal[r];
            for (ni = off[r]; ni < off[r + 1]; ni++) { sum += (double)coastal[adj[ni]]; count++; }
            tmp[r] = (float)(sum / count);
        }
        memcpy(coastal, tmp, sizeof(float) * (size_t)N);                                                 /* :52 */
    }
    free(tmp);
}

/* js/precipitation.js:18-52 */
void wo_or_wind_convergence(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, const float* wx, const float* wy,
                            const float* wz, float* convergence) {
    int32_t r, ni;
    for (r = 0; r < N; r++) {
        const double wdx = wx[r], wdy = wy[r], wdz = wz[r];
        double conv = 0; int32_t count = 0;
        for (ni = off[r]; ni < off[r + 1]; ni++) {
            const int32_t nb = adj[ni];
            const double dx = (double)xyz[3 * nb] - (double)xyz[3 * r];
            const double dy = (double)xyz[3 * nb + 1] - (double)xyz[3 * r + 1];
            const double dz = (double)xyz[3 * nb + 2] - (double)xyz[3 * r + 2];
            conv -= ((double)wx[nb] + wdx) * dx + ((double)wy[nb] + wdy) * dy + ((double)wz[nb] + wdz) * dz;
            count++;
        }
        convergence[r] = count > 0 ? (float)(conv / count) : 0.0f;
    }
}

/* js/precipitation.js:59-195; r_oceanWarmth may be NULL.  Returns the buffer the last iteration wrote (copied to out). */
void wo_or_advect_moisture(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, const float* r_heightKm,
                           const uint8_t* r_isLand, const float* r_windE, const float* r_windN, const float* wx, const float* wy,
                           const float* wz, const float* r_oceanWarmth, const int32_t* r_coastDistLand, int32_t maxHops, float* out) {
    float* a = (float*)calloc((size_t)N, sizeof(float));
    float* b = (float*)calloc((size_t)N, sizeof(float));
    float *src = a, *dst = b;
    int32_t r, ni, iter;
    const double depletionBase = 1 - pow(0.78, 1.0 / maxHops);                                           /* :123 */
    for (r = 0; r < N; r++) {                                                                            /* :70-119 */
        double warmthSum = 0, oceanDirX = 0, oceanDirY = 0, oceanDirZ = 0; int32_t oceanCount = 0;
        if (!r_isLand[r]) {
            const double warmth = r_oceanWarmth ? (double)r_oceanWarmth[r] : 0;
            a[r] = (float)(0.4 + 0.35 * (warmth > 0 ? warmth : 0));
            continue;
        }
        if (r_coastDistLand[r] != 0) continue;
        for (ni = off[r]; ni < off[r + 1]; ni++) {
            const int32_t nb = adj[ni];
            if (!r_isLand[nb]) {
                oceanCount++;
                if (r_oceanWarmth) warmthSum += (double)r_oceanWarmth[nb];
                oceanDirX += (double)xyz[3 * nb] - (double)xyz[3 * r];
                oceanDirY += (double)xyz[3 * nb + 1] - (double)xyz[3 * r + 1];
                oceanDirZ += (double)xyz[3 * nb + 2] - (double)xyz[3 * r + 2];
            }
        }
        if (oceanCount == 0) continue;
        {
            const double avgWarmth = warmthSum / oceanCount;
            const double windDotOcean = (double)wx[r] * oceanDirX + (double)wy[r] * oceanDirY + (double)wz[r] * oceanDirZ;
            const double onshore = windDotOcean < 0 ? 1.0 : 0.25;
            double cl = avgWarmth < 1 ? avgWarmth : 1;                  /* Math.min(1, avgWarmth) */
            double warmthFactor;
            if (!(cl > -0.8)) cl = (cl != cl) ? cl : -0.8;              /* Math.max(-0.8, .) */
            warmthFactor = 0.5 + 0.5 * cl;
            a[r] = (float)(onshore * warmthFactor);
        }
    }
    for (iter = 0; iter < maxHops; iter++) {                                                             /* :128-187 */
        for (r = 0; r < N; r++) {
            double we, wn, upwindMoisture = 0, upwindWeight = 0, upwindHeightSum = 0, heightHere;
            if (!r_isLand[r]) { dst[r] = src[r]; continue; }
            we = r_windE[r]; wn = r_windN[r];
            if (we * we + wn * wn < 1e-6) { dst[r] = src[r]; continue; }
            heightHere = r_heightKm[r];
            for (ni = off[r]; ni < off[r + 1]; ni++) {
                const int32_t nb = adj[ni];
                const double dx = (double)xyz[3 * r] - (double)xyz[3 * nb];
                const double dy = (double)xyz[3 * r + 1] - (double)xyz[3 * nb + 1];
                const double dz = (double)xyz[3 * r + 2] - (double)xyz[3 * nb + 2];
                const double dot = (double)wx[nb] * dx + (double)wy[nb] * dy + (double)wz[nb] * dz;
                if (dot > 0) {
                    upwindMoisture += (double)src[nb] * dot;
                    upwindHeightSum += (double)r_heightKm[nb] * dot;
                    upwindWeight += dot;
                }
            }
            if (upwindWeight > 0) {
                const double incoming = upwindMoisture / upwindWeight;
                const double upwindHeight = upwindHeightSum / upwindWeight;
                double heightGain = heightHere - upwindHeight, normalizedGain, elevDepletion, depletion, keep, carried, cur;
                if (!(heightGain > 0)) heightGain = (heightGain != heightGain) ? heightGain : 0;         /* Math.max(0, .) */
                normalizedGain = heightGain * maxHops;
                elevDepletion = normalizedGain * 0.55;
                if (!(elevDepletion < 0.8)) elevDepletion = (elevDepletion != elevDepletion) ? elevDepletion : 0.8;   /* Math.min(0.8, .) */
                depletion = depletionBase + elevDepletion;
                keep = 1 - depletion;
                if (!(keep > 0)) keep = (keep != keep) ? keep : 0;
                carried = incoming * keep;
                cur = src[r];
                dst[r] = (float)((carried > cur || carried != carried) ? carried : cur);                  /* Math.max(src[r], carried) */
                if (cur != cur) dst[r] = (float)cur;
            } else dst[r] = src[r];
        }
        { float* t = src; src = dst; dst = t; }
    }
    memcpy(out, src, sizeof(float) * (size_t)N);
    free(a); free(b);
}
