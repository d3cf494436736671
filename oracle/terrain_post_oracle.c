/*
 * ORACLE (test infrastructure only — see wo_oracle.h): serial C restatement of js/terrain-post.js.
 * Every loop keeps the reference's visiting order; every Float32Array store is a (float) cast; all
 * arithmetic is double.  Parity pinned against tests/golden/post_*.npz (bit-exact).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "wo_oracle.h"

#define PI 3.141592653589793

/* ------------------------------------------------------------------------------------------------
 * Stable sort of cell ids by a float32 key, ascending or descending — the observable behaviour of
 * V8's stable Array.prototype.sort with comparator (a,b)=>key[b]-key[a] (js/terrain-post.js:471,563)
 * or key[a]-key[b] (:204): ties (including -0 vs +0) keep their previous relative order.
 * Implemented as an LSD radix sort, which is stable by construction.
 * ---------------------------------------------------------------------------------------------- */
static uint32_t key_bits(float f, int descending) {
    uint32_t u;
    if (f == 0.0f) f = 0.0f;                 /* -0 compares equal to +0 */
    memcpy(&u, &f, 4);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   /* monotone ascending map */
    return descending ? ~u : u;
}

static void stable_sort_by_key(int32_t* cells, int32_t n, const float* key, int descending,
                               int32_t* tmpCells, uint32_t* k0, uint32_t* k1) {
    int32_t i;
    int pass;
    for (i = 0; i < n; i++) k0[i] = key_bits(key[cells[i]], descending);
    for (pass = 0; pass < 4; pass++) {
        uint32_t cnt[257];
        const int sh = pass * 8;
        memset(cnt, 0, sizeof(cnt));
        for (i = 0; i < n; i++) cnt[((k0[i] >> sh) & 255) + 1]++;
        for (i = 0; i < 256; i++) cnt[i + 1] += cnt[i];
        for (i = 0; i < n; i++) {
            uint32_t d = cnt[(k0[i] >> sh) & 255]++;
            k1[d] = k0[i]; tmpCells[d] = cells[i];
        }
        { uint32_t* t = k0; k0 = k1; k1 = t; }
        { int32_t j; for (j = 0; j < n; j++) cells[j] = tmpCells[j]; }
    }
}

/* js/terrain-post.js:100-105.  JS Number semantics: products are doubles (rounded above 2^53) and are
 * reduced mod 2^32 only afterwards; `^` yields a signed int32. */
static double cell_noise(int32_t r) {
    double p = (double)r * 2654435761.0;
    uint32_t h = (uint32_t)(uint64_t)p;
    int32_t x = (int32_t)((h >> 16) ^ h);
    double q = (double)x * 73244475.0;               /* 0x45d9f3b */
    h = (uint32_t)(int64_t)q;
    h = (h >> 16) ^ h;
    return ((double)h / 4294967295.0) * 0.01;
}

/* exposed for the known-answer test */
uint32_t wo_or_cell_noise_hash(int32_t r) {
    double p = (double)r * 2654435761.0;
    uint32_t h = (uint32_t)(uint64_t)p;
    int32_t x = (int32_t)((h >> 16) ^ h);
    double q = (double)x * 73244475.0;
    h = (uint32_t)(int64_t)q;
    return (h >> 16) ^ h;
}

/* ------------------------------------------------------------------------------------------------
 * MinHeap (js/terrain-post.js:12-47)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int32_t* data; int32_t size; const float* key; } heap_t;
#ifdef WO_ORACLE_TIE_BY_ID   /* research only: equal keys ordered by cell id (what the device flood's label order does) */
#define HEAP_LT(h, a, b) ((h)->key[a] < (h)->key[b] || ((h)->key[a] == (h)->key[b] && (a) < (b)))
#endif

static void heap_push(heap_t* h, int32_t cell) {
    int32_t i = h->size++;
    h->data[i] = cell;
    while (i > 0) {
        int32_t parent = (i - 1) >> 1;
        int32_t t;
#ifdef WO_ORACLE_TIE_BY_ID
        if (!HEAP_LT(h, h->data[i], h->data[parent])) break;
#elif defined(WO_ORACLE_TIE_VARIANT)    /* research only: a different (equally arbitrary) order among EQUAL keys, to measure how much the
                                    result depends on the reference heap's tie mechanics */
        if (h->key[h->data[i]] > h->key[h->data[parent]]) break;
#else
        if (h->key[h->data[i]] >= h->key[h->data[parent]]) break;
#endif
        t = h->data[i]; h->data[i] = h->data[parent]; h->data[parent] = t;
        i = parent;
    }
}

static int32_t heap_pop(heap_t* h) {
    int32_t top = h->data[0];
    int32_t last = h->data[--h->size];
    if (h->size > 0) {
        int32_t i = 0;
        const int32_t n = h->size;
        h->data[0] = last;
        for (;;) {
            int32_t smallest = i, t;
            const int32_t l = 2 * i + 1, r = 2 * i + 2;
#ifdef WO_ORACLE_TIE_BY_ID
            if (l < n && HEAP_LT(h, h->data[l], h->data[smallest])) smallest = l;
            if (r < n && HEAP_LT(h, h->data[r], h->data[smallest])) smallest = r;
#else
            if (l < n && h->key[h->data[l]] < h->key[h->data[smallest]]) smallest = l;
            if (r < n && h->key[h->data[r]] < h->key[h->data[smallest]]) smallest = r;
#endif
            if (smallest == i) break;
            t = h->data[i]; h->data[i] = h->data[smallest]; h->data[smallest] = t;
            i = smallest;
        }
    }
    return top;
}

/* js/terrain-post.js:59-215 */
void wo_or_priority_flood_carve(int32_t N, const int32_t* adjOffset, const int32_t* adjList,
                                float* e, const uint8_t* isOcean, double carveStrength) {
    const double EPS = 1e-7;
    int32_t* oceanLabel = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int32_t* stack = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int32_t* compSize = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N + 1));
    uint8_t* isOpenOcean = (uint8_t*)calloc((size_t)N, 1);
    float* surface = (float*)malloc(sizeof(float) * (size_t)N);
    int32_t* drainTo = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    uint8_t* visited = (uint8_t*)calloc((size_t)N, 1);
    float* key = (float*)malloc(sizeof(float) * (size_t)N);
    int32_t* path = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int32_t nComp = 0, mainLabel = 0, r, i;
    heap_t heap;

    /* :66-94 ocean components, largest (first on ties) is the open ocean */
    for (r = 0; r < N; r++) oceanLabel[r] = -1;
    for (r = 0; r < N; r++) {
        int32_t sp = 0, size = 0, label;
        if (!isOcean[r] || oceanLabel[r] >= 0) continue;
        label = nComp;
        stack[sp++] = r; oceanLabel[r] = label;
        while (sp > 0) {
            const int32_t cur = stack[--sp];
            size++;
            for (i = adjOffset[cur]; i < adjOffset[cur + 1]; i++) {
                const int32_t nb = adjList[i];
                if (isOcean[nb] && oceanLabel[nb] < 0) { oceanLabel[nb] = label; stack[sp++] = nb; }
            }
        }
        compSize[nComp++] = size;
    }
    for (i = 1; i < nComp; i++) if (compSize[i] > compSize[mainLabel]) mainLabel = i;
    for (r = 0; r < N; r++) if (isOcean[r] && oceanLabel[r] == mainLabel) isOpenOcean[r] = 1;

    /* :107-113 */
    for (r = 0; r < N; r++) { surface[r] = e[r]; drainTo[r] = -1; key[r] = (float)((double)e[r] + cell_noise(r)); }

    heap.data = stack; heap.size = 0; heap.key = key;   /* the DFS stack is free again */

    /* :118-128 seeds */
    for (r = 0; r < N; r++) {
        if (isOcean[r]) { visited[r] = 1; continue; }
        for (i = adjOffset[r]; i < adjOffset[r + 1]; i++) {
            if (isOpenOcean[adjList[i]]) { visited[r] = 1; drainTo[r] = adjList[i]; heap_push(&heap, r); break; }
        }
    }

    /* :131-147 pass 1 */
    while (heap.size > 0) {
        const int32_t c = heap_pop(&heap);
        const double surfR = surface[c];
        for (i = adjOffset[c]; i < adjOffset[c + 1]; i++) {
            const int32_t nb = adjList[i];
            if (visited[nb]) continue;
            visited[nb] = 1;
            drainTo[nb] = c;
            if ((double)e[nb] < surfR + EPS) {
                surface[nb] = (float)(surfR + EPS);
                key[nb] = (float)((double)surface[nb] + cell_noise(nb));
            }
            heap_push(&heap, nb);
        }
    }

    /* :152-196 pass 2 (sequential, reads the already-carved elevations) */
    for (r = 0; r < N; r++) {
        double deficit, peakElev, carveAmount, kernelSum, fillAmount;
        int32_t len = 0, peakIdx = -1, cur, radius, startIdx, endIdx, k;
        if (isOcean[r]) continue;
        deficit = (double)surface[r] - (double)e[r];
        if (deficit <= EPS) continue;
        peakElev = -INFINITY;
        cur = r;
        while (cur >= 0 && !isOcean[cur]) {
            path[len++] = cur;
            if ((double)e[cur] > peakElev) { peakElev = e[cur]; peakIdx = len - 1; }
            cur = drainTo[cur];
        }
        if (peakIdx < 0 || len == 0) continue;
        carveAmount = deficit * carveStrength;
        { double rr = ceil((double)len * 0.3); radius = rr > 3 ? (int32_t)rr : 3; }
        startIdx = peakIdx - radius > 0 ? peakIdx - radius : 0;
        endIdx = peakIdx + radius < len - 1 ? peakIdx + radius : len - 1;
        kernelSum = 0;
        for (k = startIdx; k <= endIdx; k++) {
            const double dist = fabs((double)(k - peakIdx));
            kernelSum += 1 - dist / (radius + 1);
        }
        if (kernelSum > 0) {
            for (k = startIdx; k <= endIdx; k++) {
                const double dist = fabs((double)(k - peakIdx));
                const double weight = (1 - dist / (radius + 1)) / kernelSum;
                e[path[k]] = (float)((double)e[path[k]] - carveAmount * weight);
                if (e[path[k]] < 0) e[path[k]] = 0;
            }
        }
        fillAmount = deficit * (1 - carveStrength);
        e[r] = (float)((double)e[r] + fillAmount);
    }

    /* :200-214 pass 3: ascending surface (stable), enforce monotone drainage */
    {
        int32_t nLand = 0;
        int32_t* order = oceanLabel;           /* reuse */
        uint32_t* k0 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)N);
        uint32_t* k1 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)N);
        for (r = 0; r < N; r++) if (!isOcean[r]) order[nLand++] = r;
        stable_sort_by_key(order, nLand, surface, 0, path, k0, k1);
        for (i = 0; i < nLand; i++) {
            const int32_t c = order[i], target = drainTo[c];
            double targetElev;
            if (target < 0) continue;
            targetElev = isOcean[target] ? 0 : (double)e[target];
            if ((double)e[c] <= targetElev) e[c] = (float)(targetElev + EPS);
        }
        free(k0); free(k1);
    }
    free(oceanLabel); free(stack); free(compSize); free(isOpenOcean); free(surface); free(drainTo);
    free(visited); free(key); free(path);
}

/* js/terrain-post.js:233-309 */
void wo_or_warp_terrain(int32_t N, const int32_t* adjOffset, const int32_t* adjList, float* e,
                        const float* xyz, double seed, double strength, const float* hot) {
    uint8_t P[512], M[512];
    const double freq = 4;
    const int octaves = 5;
    double maxAmp, warpBias;
    float* out;
    int32_t r;
    if (strength <= 0) return;
    wo_or_noise_init(seed + 9999, P, M);
    maxAmp = 0.12 * strength;
    out = (float*)malloc(sizeof(float) * (size_t)N);
    memcpy(out, e, sizeof(float) * (size_t)N);
    for (r = 0; r < N; r++) {
        const double px = xyz[3 * r], py = xyz[3 * r + 1], pz = xyz[3 * r + 2];
        double ex = -pz, ey = 0, ez = px;
        const double elen = sqrt(ex * ex + ez * ez);
        double nx, ny, nz, nlen, nnx, nny, nnz, d1, d2, wx, wy, wz, wlen, bestDot;
        int32_t cur, i;
        if (elen > 1e-10) { ex /= elen; ez /= elen; } else { ex = 1; ez = 0; }
        nx = py * ez; ny = pz * ex - px * ez; nz = -py * ex;
        nlen = sqrt(nx * nx + ny * ny + nz * nz);
        if (nlen == 0 || nlen != nlen) nlen = 1;        /* `|| 1` */
        nnx = nx / nlen; nny = ny / nlen; nnz = nz / nlen;
        d1 = wo_or_fbm(P, M, px * freq, py * freq, pz * freq, octaves, 2.0 / 3.0) * maxAmp;
        d2 = wo_or_fbm(P, M, px * freq + 31.7, py * freq + 47.3, pz * freq + 19.1, octaves, 2.0 / 3.0) * maxAmp;
        wx = px + ex * d1 + nnx * d2;
        wy = py + ey * d1 + nny * d2;
        wz = pz + ez * d1 + nnz * d2;
        wlen = sqrt(wx * wx + wy * wy + wz * wz);
        if (wlen == 0 || wlen != wlen) wlen = 1;
        wx /= wlen; wy /= wlen; wz /= wlen;
        cur = r;
        bestDot = wx * px + wy * py + wz * pz;
        for (;;) {
            int moved = 0;
            const int32_t iEnd = adjOffset[cur + 1];
            for (i = adjOffset[cur]; i < iEnd; i++) {
                const int32_t nb = adjList[i];
                const double dot = wx * xyz[3 * nb] + wy * xyz[3 * nb + 1] + wz * xyz[3 * nb + 2];
                if (dot > bestDot) { bestDot = dot; cur = nb; moved = 1; }
            }
            if (!moved) break;
        }
        out[r] = e[cur];
    }
    warpBias = 0.25 + 0.5 * strength;
    for (r = 0; r < N; r++) {
        const double orig = e[r], warped = out[r];
        double bias = warpBias;
        if (hot) {
            double den = fabs(orig);
            double hf;
            if (den == 0) den = 1;
            hf = fabs((double)hot[r]) / den;
            if (hf > 1) hf = 1;                          /* Math.min(1, .) */
            bias *= 1 - 0.8 * hf;
        }
        if (warped > orig) e[r] = (float)(orig + (warped - orig) * bias);
        else e[r] = (float)(warped + (orig - warped) * (1 - bias));
    }
    free(out);
}

/* js/terrain-post.js:317-354 */
void wo_or_smooth_elevation(int32_t N, const int32_t* adjOffset, const int32_t* adjList, float* e,
                            const uint8_t* isOcean, int32_t iterations, double strength) {
    float* tmp = (float*)malloc(sizeof(float) * (size_t)N);
    uint8_t* locked = (uint8_t*)calloc((size_t)N, 1);
    int32_t r, i, iter;
    for (r = 0; r < N; r++) {
        if (isOcean[r]) continue;
        for (i = adjOffset[r]; i < adjOffset[r + 1]; i++) if (isOcean[adjList[i]]) { locked[r] = 1; break; }
    }
    for (iter = 0; iter < iterations; iter++) {
        for (r = 0; r < N; r++) {
            double h, wSum = 0, hSum = 0;
            if (locked[r]) { tmp[r] = e[r]; continue; }
            h = e[r];
            for (i = adjOffset[r]; i < adjOffset[r + 1]; i++) {
                const double nh = e[adjList[i]];
                const double diff = fabs(nh - h);
                const double w = 1 / (1 + diff * 8);
                wSum += w;
                hSum += nh * w;
            }
            if (wSum > 0) { const double avg = hSum / wSum; tmp[r] = (float)(h + (avg - h) * strength); }
            else tmp[r] = (float)h;
        }
        for (r = 0; r < N; r++) e[r] = tmp[r];
    }
    free(tmp); free(locked);
}

static double smoothstep(double x, double e0, double e1) {
    double t = (x - e0) / (e1 - e0);
    if (!(t < 1)) t = (t != t) ? t : 1;     /* Math.min(1, t) */
    if (!(t > 0)) t = (t != t) ? t : 0;     /* Math.max(0, .) */
    return t * t * (3 - 2 * t);
}

/* js/terrain-post.js:369-707 */
void wo_or_erode_composite(int32_t N, const int32_t* adjOffset, const int32_t* adjList, float* e,
                           const float* xyz, const uint8_t* isOcean,
                           int32_t hIters, double K, double m, double dt,
                           int32_t tIters, double talusSlope, double kThermal,
                           int32_t gIters, double glacialStrength, const float* neighborDist) {
    int32_t totalIters, landCount = 0, r, i, j, iter, midFloodIter, midFloodDone = 0;
    int32_t *landCells, *tmpCells, *drainTarget, *iceTarget = NULL, *excNb;
    uint32_t *k0, *k1;
    float *cellDist, *flow, *delta, *glacIdx = NULL, *iceFlow = NULL, *excVal;
    uint8_t* numIceUpstream = NULL;
    double gScale, gCarveRate, gConvergenceBonus, gDepositAmount, gFjordCarve;
    const double gFlowThreshold = 0.1, gFjordThreshold = 0.5;
    int32_t maxDeg = 0;

    if (gIters < 0) gIters = 0;
    if (glacialStrength != glacialStrength) glacialStrength = 0;
    totalIters = hIters > tIters ? hIters : tIters;
    if (gIters > totalIters) totalIters = gIters;
    if (totalIters <= 0) return;

    landCells = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    for (r = 0; r < N; r++) if (!isOcean[r]) landCells[landCount++] = r;
    if (landCount == 0) { free(landCells); return; }
    tmpCells = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    k0 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)N);
    k1 = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)N);
    drainTarget = (int32_t*)calloc((size_t)N, sizeof(int32_t));
    cellDist = (float*)calloc((size_t)N, sizeof(float));
    flow = (float*)calloc((size_t)N, sizeof(float));
    delta = (float*)calloc((size_t)N, sizeof(float));

    if (hIters > 0) wo_or_priority_flood_carve(N, adjOffset, adjList, e, isOcean, 0.5);

    /* :410-433 glacial precomputation */
    if (gIters > 0 && glacialStrength > 0) {
        const double thresholdLat = PI / 2 - glacialStrength * PI / 4.5;
        glacIdx = (float*)calloc((size_t)N, sizeof(float));
        for (r = 0; r < N; r++) {
            double y, polarDist, latFactor, elevFactor, latScale, a, b;
            if (isOcean[r]) continue;
            y = xyz[3 * r + 1];
            if (y > 1) y = 1;
            if (y < -1) y = -1;
            polarDist = fabs(asin(y));
            latFactor = smoothstep(polarDist, thresholdLat, PI / 2);
            elevFactor = smoothstep(e[r], 0.5, 0.9);
            latScale = smoothstep(polarDist, PI / 8, PI / 3);
            a = latFactor; b = elevFactor * 0.3 * (0.3 + 0.7 * latScale);
            glacIdx[r] = (float)((a > b ? a : b) * glacialStrength);
        }
        iceTarget = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
        iceFlow = (float*)malloc(sizeof(float) * (size_t)N);
        numIceUpstream = (uint8_t*)malloc((size_t)N);
    }
    gScale = gIters > 0 ? 1.0 / gIters : 0;
    gCarveRate = 0.02 * gScale;
    gConvergenceBonus = 0.01 * gScale;
    gDepositAmount = 0.005 * gScale;
    gFjordCarve = 0.015 * gScale;

    midFloodIter = (int32_t)floor(totalIters * 0.75 + 0.5);   /* Math.round */

    for (r = 0; r < N; r++) { const int32_t d = adjOffset[r + 1] - adjOffset[r]; if (d > maxDeg) maxDeg = d; }
    excNb = (int32_t*)malloc(sizeof(int32_t) * (size_t)(maxDeg + 1));
    excVal = (float*)malloc(sizeof(float) * (size_t)(maxDeg + 1));

    for (iter = 0; iter < totalIters; iter++) {
        int glacialThisIter, hydraulicThisIter;
        if (!midFloodDone && iter >= midFloodIter) {
            midFloodDone = 1;
            wo_or_priority_flood_carve(N, adjOffset, adjList, e, isOcean, 0.85);
        }
        glacialThisIter = (iter < gIters) && glacIdx != NULL;
        hydraulicThisIter = iter < hIters;
        if (glacialThisIter || hydraulicThisIter) stable_sort_by_key(landCells, landCount, e, 1, tmpCells, k0, k1);

        /* ---- glacial :475-557 ---- */
        if (glacialThisIter) {
            for (r = 0; r < N; r++) { iceTarget[r] = -1; numIceUpstream[r] = 0; }
            for (i = 0; i < landCount; i++) {
                int32_t bestNb = -1;
                double h, bestDrop = 0;
                r = landCells[i];
                if (glacIdx[r] <= 0) continue;
                h = e[r];
                for (j = adjOffset[r]; j < adjOffset[r + 1]; j++) {
                    const int32_t nb = adjList[j];
                    const double drop = h - (double)e[nb];
                    if (drop > bestDrop) { bestDrop = drop; bestNb = nb; }
                }
                if (bestNb >= 0) iceTarget[r] = bestNb;
            }
            for (r = 0; r < N; r++) iceFlow[r] = glacIdx[r];
            for (i = 0; i < landCount; i++) {
                int32_t target;
                r = landCells[i];
                target = iceTarget[r];
                if (target >= 0 && iceFlow[r] > 0) {
                    iceFlow[target] = (float)((double)iceFlow[target] + (double)iceFlow[r]);
                    numIceUpstream[target]++;
                }
            }
            for (i = 0; i < landCount; i++) {
                double deepening;
                r = landCells[i];
                if (iceFlow[r] <= gFlowThreshold) continue;
                deepening = gCarveRate * pow(iceFlow[r], 0.6) * glacialStrength;
                e[r] = (float)((double)e[r] - deepening);
                for (j = adjOffset[r]; j < adjOffset[r + 1]; j++) {
                    const int32_t nb = adjList[j];
                    double d, slope, f;
                    if (isOcean[nb]) continue;
                    d = neighborDist[j];
                    if (d == 0 || d != d) d = 1e-6;
                    slope = fabs((double)e[r] - (double)e[nb]) / d;
                    f = 1 - slope;
                    if (!(f > 0)) f = (f != f) ? f : 0;
                    e[nb] = (float)((double)e[nb] - deepening * 0.4 * f);
                }
                if (numIceUpstream[r] >= 2) e[r] = (float)((double)e[r] - gConvergenceBonus * pow(iceFlow[r], 0.4));
            }
            for (i = 0; i < landCount; i++) {
                int32_t target;
                r = landCells[i];
                if (iceFlow[r] <= gFlowThreshold) continue;
                target = iceTarget[r];
                if (target < 0 || isOcean[target]) continue;
                if ((double)glacIdx[target] < (double)glacIdx[r] * 0.3)
                    e[target] = (float)((double)e[target] + gDepositAmount * pow(iceFlow[r], 0.3));
            }
            for (r = 0; r < N; r++) {
                int isCoastal = 0;
                if (isOcean[r]) continue;
                if (glacIdx[r] <= 0.2 || iceFlow[r] <= gFjordThreshold) continue;
                for (j = adjOffset[r]; j < adjOffset[r + 1]; j++) if (isOcean[adjList[j]]) { isCoastal = 1; break; }
                if (isCoastal) {
                    e[r] = (float)((double)e[r] - gFjordCarve * pow(iceFlow[r], 0.5));
                    if (e[r] < 0) e[r] = 0;
                }
            }
            for (r = 0; r < N; r++) if (!isOcean[r] && e[r] < 0) e[r] = 0;
        }

        /* ---- hydraulic :560-642 ---- */
        if (hydraulicThisIter) {
            if (glacialThisIter) stable_sort_by_key(landCells, landCount, e, 1, tmpCells, k0, k1);
            for (r = 0; r < N; r++) drainTarget[r] = -1;
            for (i = 0; i < landCount; i++) {
                int32_t bestNb = -1, bestJ = -1;
                double h, bestDrop = -INFINITY;
                r = landCells[i];
                h = e[r];
                for (j = adjOffset[r]; j < adjOffset[r + 1]; j++) {
                    const int32_t nb = adjList[j];
                    const double drop = h - (double)e[nb];
                    if (drop > bestDrop) { bestDrop = drop; bestNb = nb; bestJ = j; }
                }
                if (bestDrop <= 0) {
                    double minAscent = INFINITY;
                    for (j = adjOffset[r]; j < adjOffset[r + 1]; j++) {
                        const int32_t nb = adjList[j];
                        const double ascent = (double)e[nb] - h;
                        if (ascent < minAscent) { minAscent = ascent; bestNb = nb; bestJ = j; }
                    }
                }
                if (bestNb >= 0) {
                    float d = neighborDist[bestJ];
                    drainTarget[r] = bestNb;
                    cellDist[r] = (d == 0 || d != d) ? (float)1e-6 : d;
                }
            }
            for (r = 0; r < N; r++) flow[r] = 0;
            for (i = 0; i < landCount; i++) flow[landCells[i]] = 1;
            for (i = 0; i < landCount; i++) {
                int32_t target;
                r = landCells[i];
                target = drainTarget[r];
                if (target >= 0) flow[target] = (float)((double)flow[target] + (double)flow[r]);
            }
            for (i = landCount - 1; i >= 0; i--) {
                int32_t target;
                double factor, h_receiver, h_new, eroded;
                r = landCells[i];
                target = drainTarget[r];
                if (target < 0 || cellDist[r] <= 0) continue;
                factor = K * pow(flow[r], m) * dt / (double)cellDist[r];
                h_receiver = e[target] > 0 ? (double)e[target] : 0;
                h_new = ((double)e[r] + factor * h_receiver) / (1 + factor);
                if (h_new < h_receiver) h_new = h_receiver;
                if (h_new < 0) h_new = 0;
                eroded = (double)e[r] - h_new;
                if (eroded > 0 && !isOcean[target]) {
                    const int32_t drainOfTarget = drainTarget[target];
                    double receiverSlope = 0, depositFrac, deposit;
                    if (drainOfTarget >= 0 && cellDist[target] > 0)
                        receiverSlope = fabs((double)e[target] - (double)e[drainOfTarget]) / (double)cellDist[target];
                    depositFrac = 0.5 / (1 + receiverSlope * 50);
                    deposit = eroded * depositFrac;
                    e[target] = (float)((double)e[target] + deposit);
                    if ((double)e[target] > h_new) e[target] = (float)h_new;
                }
                e[r] = (float)h_new;
            }
        }

        /* ---- thermal :645-686 ---- */
        if (iter < tIters) {
            for (r = 0; r < N; r++) delta[r] = 0;
            for (i = 0; i < landCount; i++) {
                double h, totalExcess = 0, transfer;
                int32_t excCount = 0, k;
                r = landCells[i];
                h = e[r];
                for (j = adjOffset[r]; j < adjOffset[r + 1]; j++) {
                    const int32_t nb = adjList[j];
                    double nh, d, slope;
                    if (isOcean[nb]) continue;
                    nh = e[nb];
                    if (nh >= h) continue;
                    d = neighborDist[j];
                    if (d == 0 || d != d) d = 1e-6;
                    slope = (h - nh) / d;
                    if (slope > talusSlope) {
                        const double excess = (slope - talusSlope) * d;
                        excNb[excCount] = nb;
                        excVal[excCount] = (float)excess;
                        excCount++;
                        totalExcess += excess;
                    }
                }
                if (totalExcess <= 0) continue;
                transfer = kThermal * totalExcess * 0.5;
                for (k = 0; k < excCount; k++) {
                    const double share = ((double)excVal[k] / totalExcess) * transfer;
                    delta[r] = (float)((double)delta[r] - share);
                    delta[excNb[k]] = (float)((double)delta[excNb[k]] + share);
                }
            }
            for (i = 0; i < landCount; i++) {
                r = landCells[i];
                e[r] = (float)((double)e[r] + (double)delta[r]);
            }
        }
    }

    /* :690-706 */
    if (glacIdx) {
        float* tmp = (float*)malloc(sizeof(float) * (size_t)N);
        memcpy(tmp, e, sizeof(float) * (size_t)N);
        for (r = 0; r < N; r++) {
            double sum = 0;
            int32_t count = 0;
            if (isOcean[r] || glacIdx[r] <= 0) continue;
            for (j = adjOffset[r]; j < adjOffset[r + 1]; j++)
                if (!isOcean[adjList[j]]) { sum += e[adjList[j]]; count++; }
            if (count > 0) { const double avg = sum / count; tmp[r] = (float)((double)e[r] + (avg - (double)e[r]) * 0.3); }
        }
        for (r = 0; r < N; r++) if (!isOcean[r] && glacIdx[r] > 0) e[r] = tmp[r];
        free(tmp);
    }
    free(landCells); free(tmpCells); free(k0); free(k1); free(drainTarget); free(cellDist); free(flow); free(delta);
    free(glacIdx); free(iceTarget); free(iceFlow); free(numIceUpstream); free(excNb); free(excVal);
}

/* js/terrain-post.js:713-751 */
void wo_or_sharpen_ridges(int32_t N, const int32_t* adjOffset, const int32_t* adjList, float* e,
                          const uint8_t* isOcean, int32_t iterations, double strength) {
    float* tmp = (float*)calloc((size_t)N, sizeof(float));
    float* original = (float*)malloc(sizeof(float) * (size_t)N);
    int32_t r, i, iter;
    memcpy(original, e, sizeof(float) * (size_t)N);
    for (iter = 0; iter < iterations; iter++) {
        for (r = 0; r < N; r++) {
            double h, sum = 0, avg;
            int32_t count;
            if (isOcean[r]) continue;
            h = e[r];
            count = adjOffset[r + 1] - adjOffset[r];
            for (i = adjOffset[r]; i < adjOffset[r + 1]; i++) sum += e[adjList[i]];
            if (count == 0) { tmp[r] = (float)h; continue; }
            avg = sum / count;
            if (h > avg) {
                double h_new = h + (h - avg) * strength;
                const double cap = (double)original[r] * 1.5;
                if (h_new > cap) h_new = cap;
                tmp[r] = (float)h_new;
            } else tmp[r] = (float)h;
        }
        for (r = 0; r < N; r++) if (!isOcean[r]) e[r] = tmp[r];
    }
    free(tmp); free(original);
}

/* js/terrain-post.js:758-794 */
void wo_or_soil_creep(int32_t N, const int32_t* adjOffset, const int32_t* adjList, float* e,
                      const uint8_t* isOcean, int32_t iterations, double strength) {
    int32_t* interior = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    float* tmp = (float*)calloc((size_t)N, sizeof(float));
    int32_t il = 0, r, i, li, iter;
    for (r = 0; r < N; r++) {
        int coastal = 0;
        if (isOcean[r]) continue;
        for (i = adjOffset[r]; i < adjOffset[r + 1]; i++) if (isOcean[adjList[i]]) { coastal = 1; break; }
        if (!coastal) interior[il++] = r;
    }
    for (iter = 0; iter < iterations; iter++) {
        for (li = 0; li < il; li++) {
            double h, sum = 0;
            int32_t count = 0;
            r = interior[li];
            h = e[r];
            for (i = adjOffset[r]; i < adjOffset[r + 1]; i++)
                if (!isOcean[adjList[i]]) { sum += e[adjList[i]]; count++; }
            if (count == 0) { tmp[r] = (float)h; continue; }
            tmp[r] = (float)(h + (sum / count - h) * strength);
        }
        for (li = 0; li < il; li++) e[interior[li]] = tmp[interior[li]];
    }
    free(interior); free(tmp);
}


/* js/climate-util.js:5-25 smoothField: `passes` Jacobi sweeps of (self + neighbours) / (1 + degree), Float32Array ping-pong */
void wo_or_smooth_field(int32_t numRegions, const int32_t* adjOffset, const int32_t* adjList, float* field, int32_t passes) {
    float* tmp = (float*)malloc(sizeof(float) * (size_t)numRegions);
    float *src = field, *dst = tmp;
    for (int32_t pass = 0; pass < passes; ++pass) {
        for (int32_t r = 0; r < numRegions; ++r) {
            double sum = src[r];
            int32_t count = 1;
            const int32_t end = adjOffset[r + 1];
            for (int32_t ni = adjOffset[r]; ni < end; ++ni) { sum += src[adjList[ni]]; count++; }
            dst[r] = (float)(sum / count);
        }
        float* sw = src; src = dst; dst = sw;
    }
    if (src != field) memcpy(field, src, sizeof(float) * (size_t)numRegions);
    free(tmp);
}
