/*
 * plates_oracle.c — CPU ORACLE (test infrastructure only, see wo_oracle.h) for the plate projection step:
 *   projectCoarsePlates       js/coarse-plates.js:51-117
 *   smoothAndReconnectPlates  js/plates.js:241-348
 * Serial, operation-for-operation restatement.  Parity status: PINNED against reference outputs
 * (tests/golden/plates_*.npz via oracle/ref_harness/make_golden_plates.py; tests/test_oracle_golden.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "wo_oracle.h"

/* js/coarse-plates.js:51-117; numPlates < 0 stands for `numPlates == null` */
void wo_or_project_coarse_plates(int32_t numRegions, const float* r_xyz, int32_t coarseRegions, const int32_t* cOff,
                                 const int32_t* cAdj, const float* coarse_xyz, const int32_t* coarse_r_plate, double seed,
                                 int32_t numPlates, int32_t* r_plate) {
    uint8_t perm[512], pm12[512];
    wo_or_noise_init(seed + 999, perm, pm12);                                     /* :57 */
    const double coarseEdgeRad = 3.141592653589793 / sqrt((double)coarseRegions);  /* :58 */
    double lowPlateT = 0;                                                           /* :59 */
    if (numPlates >= 0) { lowPlateT = (80 - numPlates) / 60.0; if (lowPlateT > 1) lowPlateT = 1; if (lowPlateT < 0) lowPlateT = 0; }
    const double perturbAmp = coarseEdgeRad * (1.5 + 1.0 * lowPlateT);              /* :60 */
    const double BASE_FREQ = 8;
    const int32_t NC = coarseRegions;
    const int32_t MAX_WALK = (int32_t)ceil(sqrt((double)NC));                       /* :64 */
    int32_t cur = 0;                                                                /* :65 warm start */
    for (int32_t r = 0; r < numRegions; ++r) {
        const double ox = r_xyz[3 * r], oy = r_xyz[3 * r + 1], oz = r_xyz[3 * r + 2];
        double dx = 0, dy = 0, dz = 0, amp = perturbAmp, freq = BASE_FREQ;
        for (int oct = 0; oct < 4; ++oct) {                                         /* :73-79 */
            dx += wo_or_noise3d(perm, pm12, ox * freq, oy * freq, oz * freq) * amp;
            dy += wo_or_noise3d(perm, pm12, ox * freq + 100, oy * freq + 100, oz * freq + 100) * amp;
            dz += wo_or_noise3d(perm, pm12, ox * freq + 200, oy * freq + 200, oz * freq + 200) * amp;
            amp *= 0.5; freq *= 2;
        }
        double px = ox + dx, py = oy + dy, pz = oz + dz;                            /* :82-84 */
        double len = sqrt(px * px + py * py + pz * pz);
        if (len == 0 || len != len) len = 1;                                        /* `|| 1` */
        px /= len; py /= len; pz /= len;
        double bestDot = px * coarse_xyz[3 * cur] + py * coarse_xyz[3 * cur + 1] + pz * coarse_xyz[3 * cur + 2];
        int improved = 1; int32_t steps = 0;
        while (improved && steps < MAX_WALK) {                                      /* :91-103 */
            improved = 0; ++steps;
            const int32_t iEnd = cOff[cur + 1];
            for (int32_t i = cOff[cur]; i < iEnd; ++i) {                            /* bounds fixed at loop entry; cur may move */
                const int32_t nb = cAdj[i];
                const double d = px * coarse_xyz[3 * nb] + py * coarse_xyz[3 * nb + 1] + pz * coarse_xyz[3 * nb + 2];
                if (d > bestDot) { bestDot = d; cur = nb; improved = 1; }
            }
        }
        if (steps >= MAX_WALK) {                                                    /* :106-111 */
            for (int32_t c = 0; c < NC; ++c) {
                const double d = px * coarse_xyz[3 * c] + py * coarse_xyz[3 * c + 1] + pz * coarse_xyz[3 * c + 2];
                if (d > bestDot) { bestDot = d; cur = c; }
            }
        }
        r_plate[r] = coarse_r_plate[cur];
    }
}

/* js/plates.js:241-348.  plateSeeds in the Set's iteration order. */
void wo_or_smooth_reconnect_plates(int32_t numRegions, const int32_t* adjOffset, const int32_t* adjList, int32_t* r_plate,
                                   int32_t numSeeds, const int32_t* plateSeeds, int32_t numPasses) {
    const int32_t N = numRegions;
    uint8_t* isSeed = (uint8_t*)calloc((size_t)N, 1);
    for (int32_t i = 0; i < numSeeds; ++i) {                                        /* :252-255 */
        const int32_t pid = plateSeeds[i];
        if (pid >= 0 && pid < N && r_plate[pid] == pid) isSeed[pid] = 1;
    }
    int32_t maxDeg = 0;
    for (int32_t r = 0; r < N; ++r) { const int32_t deg = adjOffset[r + 1] - adjOffset[r]; if (deg > maxDeg) maxDeg = deg; }
    int32_t* cntPlates = (int32_t*)malloc(sizeof(int32_t) * (size_t)(maxDeg + 1));
    uint8_t* cntValues = (uint8_t*)malloc((size_t)(maxDeg + 1));
    for (int32_t pass = 0; pass < numPasses; ++pass) {                              /* :265-287 in place, ascending r */
        const double threshold = pass == 0 ? 0.4 : 0.5;
        for (int32_t r = 0; r < N; ++r) {
            const int32_t rStart = adjOffset[r], rEnd = adjOffset[r + 1], deg = rEnd - rStart;
            int32_t nDistinct = 0;
            for (int32_t j = rStart; j < rEnd; ++j) {
                const int32_t p = r_plate[adjList[j]];
                int found = 0;
                for (int32_t k = 0; k < nDistinct; ++k) if (cntPlates[k] == p) { cntValues[k]++; found = 1; break; }
                if (!found) { cntPlates[nDistinct] = p; cntValues[nDistinct] = 1; nDistinct++; }
            }
            int32_t bestPlate = r_plate[r], bestCount = 0;
            for (int32_t k = 0; k < nDistinct; ++k) if (cntValues[k] > bestCount) { bestCount = cntValues[k]; bestPlate = cntPlates[k]; }
            if ((double)bestCount > deg * threshold && !isSeed[r]) r_plate[r] = bestPlate;
        }
    }
    /* reconnect (:292-347): largest component per plate, first found wins ties */
    uint8_t* visited = (uint8_t*)calloc((size_t)N, 1);
    int32_t* comp = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);      /* component ordinal of each region */
    int32_t* bfs = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int32_t* compSize = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int32_t* compPlate = (int32_t*)malloc(sizeof(int32_t) * (size_t)N);
    int32_t nComp = 0;
    for (int32_t r = 0; r < N; ++r) {
        if (visited[r]) continue;
        const int32_t pid = r_plate[r];
        int32_t qn = 0;
        bfs[qn++] = r; visited[r] = 1; comp[r] = nComp;
        for (int32_t qi = 0; qi < qn; ++qi)
            for (int32_t ni = adjOffset[bfs[qi]]; ni < adjOffset[bfs[qi] + 1]; ++ni) {
                const int32_t nb = adjList[ni];
                if (!visited[nb] && r_plate[nb] == pid) { visited[nb] = 1; comp[nb] = nComp; bfs[qn++] = nb; }
            }
        compSize[nComp] = qn; compPlate[nComp] = pid; ++nComp;
    }
    /* best component per plate id: plate ids are region indices of the coarse mesh or arbitrary ints; use a small open hash */
    int32_t cap = 1; while (cap < 4 * nComp + 16) cap <<= 1;
    int32_t* hkey = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    int32_t* hval = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
    uint8_t* hused = (uint8_t*)calloc((size_t)cap, 1);
    for (int32_t c = 0; c < nComp; ++c) {                                /* ascending discovery order = ascending first r */
        uint32_t h = ((uint32_t)compPlate[c] * 2654435761u) & (uint32_t)(cap - 1);
        while (hused[h] && hkey[h] != compPlate[c]) h = (h + 1) & (uint32_t)(cap - 1);
        if (!hused[h]) { hused[h] = 1; hkey[h] = compPlate[c]; hval[h] = c; }
        else if (compSize[c] > compSize[hval[h]]) hval[h] = c;           /* strictly larger replaces (:311) */
    }
    uint8_t* inMain = (uint8_t*)calloc((size_t)N, 1);
    for (int32_t r = 0; r < N; ++r) {
        uint32_t h = ((uint32_t)r_plate[r] * 2654435761u) & (uint32_t)(cap - 1);
        while (hkey[h] != r_plate[r] || !hused[h]) h = (h + 1) & (uint32_t)(cap - 1);
        if (hval[h] == comp[r]) inMain[r] = 1;
    }
    int32_t qn = 0;                                                      /* :324-335, in place, ascending r */
    for (int32_t r = 0; r < N; ++r) {
        if (inMain[r]) continue;
        for (int32_t ni = adjOffset[r]; ni < adjOffset[r + 1]; ++ni)
            if (inMain[adjList[ni]]) { r_plate[r] = r_plate[adjList[ni]]; inMain[r] = 1; bfs[qn++] = r; break; }
    }
    for (int32_t qi = 0; qi < qn; ++qi) {                                /* :336-346 */
        const int32_t r = bfs[qi];
        for (int32_t ni = adjOffset[r]; ni < adjOffset[r + 1]; ++ni) {
            const int32_t nb = adjList[ni];
            if (!inMain[nb]) { r_plate[nb] = r_plate[r]; inMain[nb] = 1; bfs[qn++] = nb; }
        }
    }
    free(isSeed); free(cntPlates); free(cntValues); free(visited); free(comp); free(bfs); free(compSize); free(compPlate);
    free(hkey); free(hval); free(hused); free(inMain);
}
