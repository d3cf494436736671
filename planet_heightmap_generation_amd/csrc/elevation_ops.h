// Per-cell bodies of assignElevation (reference: js/elevation.js).  Same contract as erode_ops.h: each
// function is the body of one HIP thread and is also driven by the test-only emulator.  Arithmetic in
// double, every Float32Array store of the reference is a (float) narrowing here, evaluation order follows
// the JavaScript; build with -ffp-contract=off.
//
// Split of js/elevation.js:216-1391:
//   device, per cell : findCollisions (:27-122), main uplift loop (:638-973), coastal roughening (:977-1050),
//                      island-arc uplift (:1088-1106), hotspot uplift (:1264-1372), compression (:1378-1382)
//   host, serial     : Set ordering, blends, propagateStress, assignDistanceField x5, attribute-carrying BFS
//                      fields, the 97th-percentile stress normaliser, the hotspot dome list (elevation_host.cc)
#pragma once
#include <cmath>
#include <cstdint>

#include "noise.h"

namespace wo {

constexpr double EL_PI = 3.141592653589793;
constexpr int EL_MAX_DOMES = 96;          // 5 hotspots x (1 + <= 11 chain members)
constexpr int EL_MAX_RIFTS = 3;

// plate tables: dense by plate id
struct PlateTable {
    int32_t numIds;
    const uint8_t* hasVec;     // [numIds]
    const double* pole;        // [3*numIds]
    const double* omega;       // [numIds]
    const uint8_t* isOcean;    // [numIds]
    const double* density;     // [numIds]
};

// one SimplexNoise instance = perm[512] + pm12[512]
struct NoiseTab { const uint8_t* P; const uint8_t* M; };

struct CollisionOut {          // per cell, one layer (small plates or super plates)
    float* stress; float* subduct; int8_t* btype; uint8_t* bothOcean; uint8_t* hasOcean; uint8_t* setCode;
};

struct Dome {
    double x, y, z, strength, baseStrength, sigma;
    int32_t chainIndex, chainLength;
    double ux, uy, uz, vx, vy, vz;
    double riftAngles[EL_MAX_RIFTS]; int32_t numRifts;
    double cosThreshPeak, invS2, swellStrength, cosThreshSwell, invS2Swell, driftStretch, calderaDepth, invS2Caldera, ageFactor;
    int32_t hasCaldera;
};

struct ElevParams {
    int32_t N;
    double scaleFactor, maxStress, noiseMag;
    int32_t warpOctaves, interiorBand, tectonicReach, plateauStart;
    int32_t riftHalfWidth, ridgeHalfWidth, fractureHalfWidth, baStart, baPeak, baEnd;
    int32_t coastRoughenDist, islandDist, maxArcDist, numDomes;
};

struct ElevFields {
    const float* xyz; const int32_t* plate; const uint8_t* isOcean;      // isOcean BY PLATE (js/elevation.js:397-400)
    const float* stress; const float* subduct; const int8_t* btype;
    const float* distMountain; const float* distOcean; const float* distCoastline; const float* distCoast; const float* distCoastLand;
    const float* dBdry; const float* coastStressMax; const float* coastSubductMax; const uint8_t* coastConvergent;
    const float* riftDist; const float* ridgeDist; const float* fractureDist; const float* backArcDist; const float* backArcStress;
    const float* arcDist; const float* arcStress;
    float* elev;
    float* dl;                 // 12 debug layers, layer-major [12][N], or nullptr
};
enum { DL_BASE = 0, DL_TECTONIC, DL_NOISE, DL_INTERIOR, DL_COASTAL, DL_OCEAN, DL_HOTSPOT, DL_TECACT, DL_MARGINS, DL_BACKARC, DL_FOLD, DL_ORO, DL_COUNT };

WO_HD inline double js_round(double x) { return floor(x + 0.5); }            // Math.round for the values on this path
WO_HD inline double js_min(double a, double b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }
WO_HD inline double js_max(double a, double b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }

// getPairIntensity (js/elevation.js:44-53): JS Number arithmetic, int32 bit operations
WO_HD inline double pair_intensity(int32_t a, int32_t b) {
    const int32_t lo = a < b ? a : b, hi = a < b ? b : a;
    const double p1 = (double)lo * 16807.0, p2 = (double)hi * 48271.0;
    uint32_t h = (uint32_t)((int32_t)(int64_t)p1 ^ (int32_t)(int64_t)p2);   // ToInt32 ^ ToInt32, then >>> 0
    const int32_t hs = (int32_t)h;
    const int32_t x = (hs >> 16) ^ hs;                                        // signed shift
    const double q = (double)x * 73244475.0;                                 // 0x45d9f3b, rounded as a double
    h = (uint32_t)(int64_t)q;                                                 // >>> 0
    return 0.5 + (double)(h % 10001u) / 10000.0;
}

// findCollisions, one cell (js/elevation.js:57-120)
WO_HD inline void collision_cell(int32_t r, int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, const int32_t* plate,
                                 const PlateTable& T, NoiseTab noise, const CollisionOut& O) {
    const double dt = 1e-2 / js_max(1, sqrt((double)N / 10000));
    const int32_t undulOctaves = N > 200000 ? 2 : 3;
    const int32_t myPlate = plate[r];
    double bestComp = -INFINITY, bestNormalComp = 0;
    int32_t best = -1;
    const double rx = xyz[3 * r], ry = xyz[3 * r + 1], rz = xyz[3 * r + 2];
    for (int32_t ni = off[r]; ni < off[r + 1]; ++ni) {
        const int32_t nb = adj[ni];
        const int32_t np = plate[nb];
        if (myPlate == np) continue;
        const double nx = xyz[3 * nb], ny = xyz[3 * nb + 1], nz = xyz[3 * nb + 2];
        const double dx = rx - nx, dy = ry - ny, dz = rz - nz;
        const double dBefore = sqrt(dx * dx + dy * dy + dz * dz);
        const double* p1 = T.pole + 3 * myPlate; const double o1 = T.omega[myPlate];
        const double* p2 = T.pole + 3 * np;      const double o2 = T.omega[np];
        const double v1x = o1 * (p1[1] * rz - p1[2] * ry), v1y = o1 * (p1[2] * rx - p1[0] * rz), v1z = o1 * (p1[0] * ry - p1[1] * rx);
        const double v2x = o2 * (p2[1] * nz - p2[2] * ny), v2y = o2 * (p2[2] * nx - p2[0] * nz), v2z = o2 * (p2[0] * ny - p2[1] * nx);
        const double ax = rx + v1x * dt, ay = ry + v1y * dt, az = rz + v1z * dt;
        const double bx = nx + v2x * dt, by = ny + v2y * dt, bz = nz + v2z * dt;
        const double adx = ax - bx, ady = ay - by, adz = az - bz;
        const double dAfter = sqrt(adx * adx + ady * ady + adz * adz);
        const double comp = dBefore - dAfter;
        if (comp > bestComp) {
            bestComp = comp; best = nb;
            const double rvx = v1x - v2x, rvy = v1y - v2y, rvz = v1z - v2z;
            const double bnLen = (dBefore == 0 || dBefore != dBefore) ? 1 : dBefore;
            bestNormalComp = -(rvx * dx + rvy * dy + rvz * dz) / bnLen;
        }
    }
    float stress = 0.0f, subduct = 0.5f;
    int8_t bt = 0; uint8_t both = 0, has = 0, code = 0;
    if (best != -1) {
        const int32_t np = plate[best];
        const bool collided = bestComp > 0.75 * dt;
        const bool rOcean = T.isOcean[myPlate] != 0, nOcean = T.isOcean[np] != 0;
        both = (rOcean && nOcean) ? 1 : 0;
        has = (rOcean || nOcean) ? 1 : 0;
        const double thresh = 0.3 * dt;
        bt = bestNormalComp > thresh ? 1 : (bestNormalComp < -thresh ? 2 : 3);
        if (collided) stress = (float)((bestComp / dt) * pair_intensity(myPlate, np));
        const double densityDiff = T.density[myPlate] - T.density[np];
        const double baseFactor = 0.5 + 0.5 * tanh(densityDiff * 8);
        const double undulationStrength = exp(-fabs(densityDiff) * 12);
        const double undulation = fbm(noise.P, noise.M, rx * 6, ry * 6, rz * 6, undulOctaves) * 0.4 * undulationStrength;
        subduct = (float)js_max(0, js_min(1, baseFactor + undulation));
        // Set membership (:109-118): 1 mountain, 2 coastline, 3 ocean
        if (rOcean && nOcean) code = collided ? 2 : 3;
        else if (!rOcean && !nOcean) { if (collided) code = ((double)subduct < 0.55) ? 1 : 2; }
        else code = collided ? 1 : 2;
    }
    O.stress[r] = stress; O.subduct[r] = subduct; O.btype[r] = bt; O.bothOcean[r] = both; O.hasOcean[r] = has; O.setCode[r] = code;
}

// back-arc bell (js/elevation.js:732-753 and :945-965, identical on land and ocean)
WO_HD inline bool back_arc_effect(const ElevFields& F, const ElevParams& Q, int32_t r, double& baEffect) {
    const double bad = F.backArcDist[r];
    if (!(bad != INFINITY && bad >= Q.baStart)) return false;
    const double dMtn = F.distMountain[r];
    const double orogenyFactor = (dMtn != INFINITY && dMtn < bad) ? js_max(0, dMtn / bad) : 1.0;
    baEffect = 0;
    if (bad <= Q.baPeak) {
        const double t = (bad - Q.baStart) / js_max(1, Q.baPeak - Q.baStart);
        const double s = t * t * (3 - 2 * t);
        baEffect = -0.10 * (double)F.backArcStress[r] * s * orogenyFactor;
    } else if (bad <= Q.baEnd) {
        const double t = (bad - Q.baPeak) / js_max(1, Q.baEnd - Q.baPeak);
        const double s = t * t * (3 - 2 * t);
        baEffect = -0.10 * (double)F.backArcStress[r] * (1 - s) * orogenyFactor;
    }
    return true;
}

#define EL_DL(layer) F.dl[(size_t)(layer) * (size_t)Q.N + (size_t)r]
#define EL_ADD(e, x) (e) = (float)((double)(e) + (x))

// main per-cell loop (js/elevation.js:638-973).  Returns the new r_elevation[r].
WO_HD inline float elevation_main_cell(const ElevFields& F, const ElevParams& Q, const PlateTable& T, int32_t r,
                                       NoiseTab noise, NoiseTab riftNoise, NoiseTab foldNoise) {
    const bool dl = F.dl != nullptr;
    const bool ocean_plate = F.isOcean[r] != 0;
    const double eps = 1e-3, warpScale = 0.4, S = Q.scaleFactor, noiseMag = Q.noiseMag;
    const double sub_f0 = F.subduct[r];
    const double slab_skew = 1.0 + (sub_f0 - 0.5) * 0.8;
    const double a = (double)F.distMountain[r] * slab_skew + eps;
    const double b = (double)F.distOcean[r] + eps;
    const double c = (double)F.distCoastline[r] + eps;
    const double kBaseScale = 0.6;
    float e;
    if (a == INFINITY && b == INFINITY) e = (float)(0.1 * kBaseScale);
    else e = (float)((1 / a - 1 / b) / (1 / a + 1 / b + 1 / c) * kBaseScale);
    if (dl) EL_DL(DL_BASE) = e;

    const double stress_rel = js_min(1, (double)F.stress[r] / Q.maxStress);
    const int32_t btype = F.btype[r];
    const double x = F.xyz[3 * r], y = F.xyz[3 * r + 1], z = F.xyz[3 * r + 2];
    const double wx = x + warpScale * fbm(noise.P, noise.M, x + 5.3, y + 1.7, z + 3.1, Q.warpOctaves);
    const double wy = y + warpScale * fbm(noise.P, noise.M, x + 8.1, y + 2.9, z + 7.3, Q.warpOctaves);
    const double wz = z + warpScale * fbm(noise.P, noise.M, x + 1.4, y + 6.2, z + 4.8, Q.warpOctaves);

    const double oro_raw = noise3d(noise.P, noise.M, x * 1.5 + 33.7, y * 1.5 + 11.2, z * 1.5 + 22.9);
    const double oro_shaped = oro_raw >= 0 ? sqrt(oro_raw) : -sqrt(-oro_raw);
    const double oro_gain = js_max(0, js_min(1, 0.5 + 0.5 * oro_shaped));
    if (dl) EL_DL(DL_ORO) = (float)(oro_gain - 0.5);

    if (!ocean_plate) {
        const double sf = F.subduct[r];
        const float e_prev = e;
        if (sf > 0.5 && e > 0) {
            const double slab_damp = (sf - 0.5) * 2;
            e = (float)((double)e * (1 - slab_damp * 0.42));
        }
        if (stress_rel > 0.01) {
            const double stress_amp = stress_rel * stress_rel * 0.55 * oro_gain;
            const double uplift = stress_amp * (1 - sf);
            const double sag = stress_amp * 0.4 * sf;
            const double relief_var = 0.60 + 0.8 * fbm(noise.P, noise.M, x * 8 + 13.7, y * 8 + 9.2, z * 8 + 4.5, 3);
            EL_ADD(e, (uplift - sag) * relief_var);
        }
        if (stress_rel > 0 && stress_rel < 0.10) {
            const double foreland_u = stress_rel / 0.10;
            e = (float)((double)e - 0.06 * (1 - foreland_u));
        }
        {   // rift graben (:698-727)
            const double rd = F.riftDist[r];
            if (rd != INFINITY) {
                const double floor_edge = js_max(1, js_round(1.5 * S));
                const double shoulder_edge = js_max(2, js_round(2.5 * S));
                double rift_drop = 0;
                if (rd <= 0.5) {
                    rift_drop = -0.15;
                    rift_drop += ridged_fbm(riftNoise.P, riftNoise.M, x * 8, y * 8, z * 8, 3) * 0.04;
                } else if (rd <= floor_edge) {
                    const double t = rd / floor_edge;
                    rift_drop = -0.12 * (1 - t * 0.3);
                    rift_drop += ridged_fbm(riftNoise.P, riftNoise.M, x * 8, y * 8, z * 8, 3) * 0.03 * (1 - t);
                } else if (rd <= shoulder_edge) {
                    const double t = (rd - floor_edge) / (shoulder_edge - floor_edge);
                    rift_drop = 0.03 * (1 - t);
                } else if (Q.riftHalfWidth > shoulder_edge) {
                    const double t = (rd - shoulder_edge) / (Q.riftHalfWidth - shoulder_edge);
                    const double fadeT = js_min(1, t);
                    const double fade = fadeT * fadeT * (3 - 2 * fadeT);
                    rift_drop = 0.03 * (1 - fade) * 0.2;
                }
                EL_ADD(e, rift_drop);
            }
        }
        {   // back-arc
            double backarc_add;
            if (back_arc_effect(F, Q, r, backarc_add)) { EL_ADD(e, backarc_add); if (dl) EL_DL(DL_BACKARC) = (float)backarc_add; }
        }
        if (dl) EL_DL(DL_TECTONIC) = (float)((double)e - (double)e_prev);

        const double dMtn = F.distMountain[r];
        const double near_raw = (dMtn == INFINITY || dMtn >= Q.tectonicReach) ? 0 : (1 - dMtn / Q.tectonicReach);
        const double tect_level = js_max(stress_rel, near_raw * near_raw);
        if (dl) EL_DL(DL_TECACT) = (float)tect_level;

        {   // fold ridges (:770-799)
            const int32_t pid = F.plate[r];
            const double fold_level = tect_level * tect_level;
            if (pid >= 0 && pid < T.numIds && T.hasVec[pid] && fold_level > 0.01) {
                const double ppx = T.pole[3 * pid], ppy = T.pole[3 * pid + 1], ppz = T.pole[3 * pid + 2];
                const double u = x * ppx + y * ppy + z * ppz;
                const double phase_bend = fbm(foldNoise.P, foldNoise.M, x * 3 + 55.3, y * 3 + 33.7, z * 3 + 17.2, 2) * 0.08;
                const double phase = (u + phase_bend) * 30 * EL_PI;
                const double ridge = 1 - fabs(sin(phase));
                const double fold_mid = ridge - 0.36;
                const double amp_scale = 0.6 + 0.4 * fbm(foldNoise.P, foldNoise.M, x * 4 + 88.1, y * 4 + 62.3, z * 4 + 41.7, 2);
                const double e_gain = 1 + 4 * js_max(0, (double)e);
                const double fold_gain = fold_level * js_max(0, 1 - sf * 1.5) * noiseMag * 0.8 * e_gain;
                const double fold_add = fold_mid * fold_gain * amp_scale;
                EL_ADD(e, fold_add);
                if (dl) EL_DL(DL_FOLD) = (float)fold_add;
            }
        }
        const bool on_plateau = sf < 0.45 && dMtn != INFINITY && dMtn > Q.plateauStart;
        const double blend = js_min(1, stress_rel * 3);
        const double noise_smooth = fbm(noise.P, noise.M, wx, wy, wz) * noiseMag;
        const double ridged_raw = ridged_fbm(noise.P, noise.M, wx, wy, wz) * noiseMag * 1.5;
        const double noise_raw = noise_smooth * (1 - blend) + ridged_raw * blend;
        const double noise_detail = fbm(noise.P, noise.M, wx * 4 + 22.1, wy * 4 + 6.8, wz * 4 + 15.4, 4, 0.5) * noiseMag * 0.5;
        const double noise_level = js_min(1, stress_rel * 4);
        const double plateau_damp = on_plateau ? js_max(0.30, 1 - tect_level * 0.60) : 1.0;
        const double noise_freq = (0.25 + 0.75 * noise_level) * plateau_damp;
        const double noise_fine = fbm(noise.P, noise.M, wx * 8 + 41.7, wy * 8 + 13.2, wz * 8 + 27.9, 3, 0.5) * noiseMag * 0.25;
        const double fine_freq = sqrt(noise_freq);
        const double noise_sum = (noise_raw + noise_detail) * noise_freq + noise_fine * fine_freq;
        EL_ADD(e, noise_sum);
        float dlNoise = (float)noise_sum;
        {   // dissection (:829-842)
            const double e_now = e;
            if (e_now > 0.12) {
                const double e_over = e_now - 0.12;
                const double incise_raw = fbm(noise.P, noise.M, wx * 16 + 71.3, wy * 16 + 44.8, wz * 16 + 29.1, 3, 0.5);
                const double incise_amp = sqrt(e_over) * stress_rel * noiseMag * 0.4;
                const double incise_add = incise_raw * incise_amp;
                EL_ADD(e, incise_add);
                EL_ADD(dlNoise, incise_add);
            }
        }
        {   // summits (:848-863): ridgedFbm(x, y, z, 3, 0.5) -> lacunarity 0.5, as written in the reference
            const double e_now = e;
            if (e_now > 0.65 && stress_rel > 0.2) {
                const double excess = e_now - 0.65;
                const double peak_raw = ridged_fbm(noise.P, noise.M, wx * 24 + 91.3, wy * 24 + 55.7, wz * 24 + 38.2, 3, 0.5);
                const double spike = js_max(0, peak_raw - 0.45);
                const double peak_add = spike * excess * stress_rel * 1.2;
                EL_ADD(e, peak_add);
                EL_ADD(dlNoise, peak_add);
            }
        }
        if (dl) EL_DL(DL_NOISE) = dlNoise;
        float dlInterior = 0.0f;
        const double lcd = F.distCoastLand[r];
        if (lcd < INFINITY) {
            const double tDown = js_min(lcd / Q.interiorBand, 1);
            const double sDown = tDown * tDown * (3 - 2 * tDown);
            const double tUp = js_min(lcd / (Q.interiorBand * 0.4), 1);
            const double sUp = tUp * tUp * (3 - 2 * tUp);
            const double craton_lift = 0.06 + tect_level * 0.16;
            const double base_shift = -0.08 * (1 - sDown) + craton_lift * sUp;
            const double mod = 1.0 + 0.2 * fbm(noise.P, noise.M, x * 2 + 19.3, y * 2 + 7.6, z * 2 + 13.1, 2);
            const double bias = base_shift * mod;
            EL_ADD(e, bias);
            dlInterior = (float)bias;
        }
        if (on_plateau && tect_level > 0.1) {
            const double plateau_gain = 0.025 * tect_level * (1 - sf);
            EL_ADD(e, plateau_gain);
            EL_ADD(dlInterior, plateau_gain);
        }
        if (dl) EL_DL(DL_INTERIOR) = dlInterior;
    } else {
        const double dc = F.distCoast[r];
        double abyss_base;
        if (dc < 5) abyss_base = -0.04 - 0.06 * (dc / 5);
        else if (dc < 12) abyss_base = -0.10 - 0.25 * ((dc - 5) / 7);
        else abyss_base = -0.35 + fbm(noise.P, noise.M, x * 2, y * 2, z * 2, 3) * 0.03;
        e = (float)js_min((double)e, abyss_base);
        if (dl) EL_DL(DL_OCEAN) = e;
        if (dl) {
            float mg = F.coastConvergent[r] == 1 ? 0.8f : 0.2f;
            if (F.ridgeDist[r] != INFINITY && F.ridgeDist[r] <= Q.ridgeHalfWidth) mg = 1.0f;
            if (F.fractureDist[r] != INFINITY && F.fractureDist[r] <= Q.fractureHalfWidth) mg = -0.5f;
            EL_DL(DL_MARGINS) = mg;
        }
        const float e_prev_oc = e;
        const double rd = F.ridgeDist[r];
        if (rd != INFINITY && rd <= Q.ridgeHalfWidth) {
            const double t = rd / Q.ridgeHalfWidth;
            const double ridge_taper = (1 - t) * (1 - t);
            const double ridge_raw = ridged_fbm(noise.P, noise.M, x * 3, y * 3, z * 3, 4);
            const double ridge_lift = (0.12 * ridge_raw + 0.06) * ridge_taper;
            EL_ADD(e, ridge_lift);
        }
        const double fd = F.fractureDist[r];
        if (fd != INFINITY && fd <= Q.fractureHalfWidth) {
            const double ft = fd / Q.fractureHalfWidth;
            e = (float)((double)e - 0.03 * (1 - ft));
        }
        if (btype == 1) e = (float)((double)e - (0.15 + 0.15 * stress_rel));
        {
            double backarc_add;
            if (back_arc_effect(F, Q, r, backarc_add)) { EL_ADD(e, backarc_add); if (dl) EL_DL(DL_BACKARC) = (float)backarc_add; }
        }
        if (dl) EL_DL(DL_TECTONIC) = (float)((double)e - (double)e_prev_oc);
        const double abyss_noise = fbm(noise.P, noise.M, wx, wy, wz) * noiseMag * 0.3;
        EL_ADD(e, abyss_noise);
        if (dl) EL_DL(DL_NOISE) = (float)abyss_noise;
    }
    return e;
}

// coastal roughening (js/elevation.js:983-1049); returns the new elevation
WO_HD inline float coastal_cell(const ElevFields& F, const ElevParams& Q, int32_t r, float e, NoiseTab noise, NoiseTab cNoise,
                                NoiseTab cNoise2, NoiseTab cNoise3) {
    if ((double)F.dBdry[r] > Q.coastRoughenDist) return e;
    const double x = F.xyz[3 * r], y = F.xyz[3 * r + 1], z = F.xyz[3 * r + 2];
    const double t = (double)F.dBdry[r] / Q.coastRoughenDist;
    const double sn = js_min(1, js_max((double)F.coastStressMax[r], (double)F.stress[r] / Q.maxStress));
    const bool isOc = F.isOcean[r] != 0, conv = F.coastConvergent[r] != 0;
    const bool isSubductingOcean = isOc && conv && (double)F.coastSubductMax[r] > 0.45;
    const double subSup = isSubductingOcean ? js_min(1, ((double)F.coastSubductMax[r] - 0.45) / 0.55) : 0;
    const float elevBeforeCoast = e;
    const bool isPassiveCoast = !conv;
    const double falloff1 = (1 - t) * (1 - t);
    const double stressAmp1 = 1 + sn * 5;
    const double coastFreq = isPassiveCoast ? 12 : 18;
    const double coastAmp = isPassiveCoast ? 0.08 : 0.12;
    const double n1 = fbm(cNoise.P, cNoise.M, x * coastFreq + 3.7, y * coastFreq + 7.1, z * coastFreq + 2.3, 5, 0.55);
    double coastNoise1 = n1 * coastAmp * falloff1 * stressAmp1;
    if (subSup > 0 && coastNoise1 > 0) coastNoise1 *= (1 - subSup);
    EL_ADD(e, coastNoise1);
    const double warpReach = isPassiveCoast ? 1.2 : 1.5;
    const double falloffW = js_max(0, 1 - t * warpReach);
    if (falloffW > 0) {
        const double warpAmt = 0.35 * falloffW * (1 + sn * 2);
        const double dwx = fbm(cNoise3.P, cNoise3.M, x * 6 + 11.3, y * 6 + 4.7, z * 6 + 8.2, 3, 0.6) * warpAmt;
        const double dwy = fbm(cNoise3.P, cNoise3.M, x * 6 + 2.9, y * 6 + 9.4, z * 6 + 1.6, 3, 0.6) * warpAmt;
        const double dwz = fbm(cNoise3.P, cNoise3.M, x * 6 + 7.5, y * 6 + 0.3, z * 6 + 5.9, 3, 0.6) * warpAmt;
        const double origN = fbm(noise.P, noise.M, x, y, z) * Q.noiseMag;
        const double warpN = fbm(noise.P, noise.M, x + dwx, y + dwy, z + dwz) * Q.noiseMag;
        double warpDelta = (warpN - origN) * falloffW;
        if (subSup > 0 && warpDelta > 0) warpDelta *= (1 - subSup);
        EL_ADD(e, warpDelta);
    }
    if (isOc && F.dBdry[r] > 0 && (double)F.dBdry[r] <= Q.islandDist && subSup < 0.3) {
        const double islandN = fbm(cNoise2.P, cNoise2.M, x * 35 + 5.1, y * 35 + 9.3, z * 35 + 2.7, 4, 0.5);
        const double threshold = 0.25 - sn * 0.2;
        if (islandN > threshold) {
            const double excess = (islandN - threshold) / (1 - threshold);
            const double distFade = 1 - ((double)F.dBdry[r] / Q.islandDist);
            double bump = excess * excess * 0.18 * (1 + sn * 2) * distFade;
            bump *= (1 - subSup / 0.3);
            EL_ADD(e, bump);
        }
    }
    if (F.dl) { float& d = EL_DL(DL_COASTAL); d = (float)((double)d + ((double)e - (double)elevBeforeCoast)); }
    return e;
}

// island-arc uplift (js/elevation.js:1088-1106)
WO_HD inline float arc_cell(const ElevFields& F, const ElevParams& Q, int32_t r, float e, NoiseTab arcNoise) {
    const double d = F.arcDist[r];
    if (d < 1 || d > Q.maxArcDist) return e;
    const double x = F.xyz[3 * r], y = F.xyz[3 * r + 1], z = F.xyz[3 * r + 2];
    const double peakDist = js_max(1.5, 1.5 * Q.scaleFactor), sigma = js_max(1.5, 1.5 * Q.scaleFactor);
    const double q = (d - peakDist) / sigma;
    const double distWeight = exp(-0.5 * (q * q));
    const double n = ridged_fbm(arcNoise.P, arcNoise.M, x * 4, y * 4, z * 4, 4, 2.0, 0.5, 1.0);
    if (n > 0.30) {
        const double excess = (n - 0.30) / (1 - 0.30);
        const double uplift = excess * excess * 0.55 * distWeight * (0.5 + (double)F.arcStress[r]);
        EL_ADD(e, uplift);
        if (F.dl) { float& dd = EL_DL(DL_COASTAL); dd = (float)((double)dd + uplift); }
    }
    return e;
}

// hotspot uplift (js/elevation.js:1264-1372)
WO_HD inline float hotspot_cell(const ElevFields& F, const ElevParams& Q, int32_t r, float e, const Dome* domes,
                                NoiseTab hsNoise, NoiseTab hsNoise2) {
    const double rx = F.xyz[3 * r], ry = F.xyz[3 * r + 1], rz = F.xyz[3 * r + 2];
    bool nearSwell = false, nearPeak = false;
    for (int32_t d = 0; d < Q.numDomes; ++d) {
        const Dome& dm = domes[d];
        const double cdot = dm.x * rx + dm.y * ry + dm.z * rz;
        if (cdot > dm.cosThreshSwell) { nearSwell = true; if (cdot > dm.cosThreshPeak) { nearPeak = true; break; } }
    }
    if (!nearSwell) return e;
    double shapeWarpSq = 1.0;
    if (nearPeak) {
        const double ws = 8;
        const double wx = fbm(hsNoise2.P, hsNoise2.M, rx * ws + 5.1, ry * ws + 3.7, rz * ws + 9.2, 2, 0.5) * 0.4;
        const double wy = fbm(hsNoise2.P, hsNoise2.M, rx * ws + 11.3, ry * ws + 7.1, rz * ws + 2.9, 2, 0.5) * 0.4;
        const double wz = fbm(hsNoise2.P, hsNoise2.M, rx * ws + 1.7, ry * ws + 13.5, rz * ws + 6.4, 2, 0.5) * 0.4;
        const double shapeWarp = 1.0 + 0.40 * fbm(hsNoise.P, hsNoise.M, (rx + wx) * 20 + 3.2, (ry + wy) * 20 + 7.8, (rz + wz) * 20 + 1.5, 4, 0.5);
        shapeWarpSq = shapeWarp * shapeWarp;
    }
    double totalUplift = 0, totalSwellUplift = 0, weightedAge = 0, ageWeightSum = 0;
    for (int32_t d = 0; d < Q.numDomes; ++d) {
        const Dome& dm = domes[d];
        const double dot = dm.x * rx + dm.y * ry + dm.z * rz;
        if (dot > dm.cosThreshSwell) {
            const double swAngleSq = 2 * (1 - dot);
            totalSwellUplift += dm.swellStrength * exp(swAngleSq * dm.invS2Swell);
        }
        if (dot < dm.cosThreshPeak) continue;
        const double offX = rx - dot * dm.x, offY = ry - dot * dm.y, offZ = rz - dot * dm.z;
        const double parComp = offX * dm.ux + offY * dm.uy + offZ * dm.uz;
        const double perpComp = offX * dm.vx + offY * dm.vy + offZ * dm.vz;
        const double stretchedParSq = (parComp * dm.driftStretch) * (parComp * dm.driftStretch);
        const double angleSq = stretchedParSq + perpComp * perpComp;
        double gauss = exp(angleSq * shapeWarpSq * dm.invS2);
        if (dm.numRifts > 0 && gauss > 0.01) {
            const double angle = atan2(perpComp, parComp);
            double maxRift = 0;
            for (int32_t ri = 0; ri < dm.numRifts; ++ri) {
                double da = angle - dm.riftAngles[ri];
                da = da - js_round(da / (2 * EL_PI)) * 2 * EL_PI;
                const double c2 = cos(da);
                const double riftFactor = c2 * c2 * c2 * c2;
                if (riftFactor > maxRift) maxRift = riftFactor;
            }
            gauss *= (1.0 + 0.5 * maxRift);
        }
        const double peakUplift = dm.strength * gauss;
        totalUplift += peakUplift;
        weightedAge += dm.ageFactor * peakUplift;
        ageWeightSum += peakUplift;
        if (dm.hasCaldera) totalUplift -= dm.calderaDepth * exp(angleSq * dm.invS2Caldera);
    }
    const double combinedUplift = totalSwellUplift + totalUplift;
    if (combinedUplift > 0.001) {
        const double age = ageWeightSum > 0 ? weightedAge / ageWeightSum : 0;
        const double texBase = 0.7 * ridged_fbm(hsNoise.P, hsNoise.M, rx * 12, ry * 12, rz * 12, 4, 2.0, 0.5, 1.0);
        const double texDetail = 0.3 * ridged_fbm(hsNoise.P, hsNoise.M, rx * 30, ry * 30, rz * 30, 3, 2.0, 0.5, 1.0);
        const double texRaw = texBase + texDetail;
        const double texMin = 0.4 + age * 0.3, texMax = 1.2 - age * 0.2;
        const double volc = texMin + (texMax - texMin) * texRaw;
        const double uplift = totalSwellUplift + js_max(0, totalUplift) * volc;
        EL_ADD(e, uplift);
        if (F.dl) EL_DL(DL_HOTSPOT) = (float)uplift;
    }
    return e;
}

// peak compression (js/elevation.js:1378-1382)
WO_HD inline float compress_cell(float e) { return e > 0 ? (float)pow((double)e, 0.92) : e; }

#undef EL_DL
#undef EL_ADD

}  // namespace wo
