// The FIFO breadth-first fields of assignElevation on the device (reference: js/elevation.js:464-631, 1059-1086):
//   coast-boundary distance with the attributes of the strongest boundary cell (:464-509), rift (:511-538), ridge
//   (:542-568), fracture (:570-596), back-arc with its carried stress (:598-631), island-arc with its carried stress
//   (:1059-1086).
//
// The reference walks one FIFO queue per field (seeds in ascending id).  The DISTANCES are plain hop counts over the
// admissible edges and do not depend on the queue order: the fields without attributes run as level-synchronous
// claims (atomicMin on the float bits, one launch per level).  The CARRIED ATTRIBUTES do: a cell takes them from the
// parent that relaxed it first, i.e. the parent earliest in the queue, and the queue order of a level is defined by
// the order of the level before it — a child is appended when its first parent is popped, in that parent's adjacency
// order.  So a level is four launches that reproduce the queue exactly:
//   push    every frontier entry i offers itself to its admissible unreached neighbours: atomicMin(pushPos[nb], i)
//           (coast: atomicMax of {strongest stress, earliest i} as well — the reference replaces the attributes when a
//           later parent of the same distance is strictly stronger, :500-505)
//   count   entry i counts the neighbours it won, in adjacency order
//   scan    exclusive prefix sum over the frontier in queue order (one workgroup)
//   assign  entry i writes its children to next[base + k] (their queue positions), their distance and attributes
// Bodies only (WO_HD); kernels below; the emulator (tests/emu) drives the same bodies against the host walk.
#pragma once
#include <cstdint>

#include "noise.h"

namespace wo {

enum BfsMode : int32_t { BFS_COAST = 0, BFS_RIFT = 1, BFS_RIDGE = 2, BFS_FRACTURE = 3, BFS_BACKARC = 4, BFS_ARC = 5 };

struct BfsCtx {
    int32_t N;
    const int32_t* off; const int32_t* adj;
    const uint8_t* isOcean;            // by plate (js/elevation.js:396-399)
    const int32_t* plate;
};

// admission of neighbour nr reached from r (the `pass` lambdas of elevation_host.cc, :511-631, 1059-1086)
WO_HD inline bool bfs_admit(const BfsCtx& B, int32_t mode, int32_t nr, int32_t r) {
    switch (mode) {
        case BFS_COAST: return true;
        case BFS_RIFT: return B.plate[nr] == B.plate[r] && !B.isOcean[nr];
        case BFS_RIDGE: case BFS_FRACTURE: return B.isOcean[nr] != 0;
        case BFS_BACKARC: return B.plate[nr] == B.plate[r];
        default: return B.plate[nr] == B.plate[r] && B.isOcean[nr] != 0;      // BFS_ARC
    }
}

WO_HD inline uint32_t bfs_f32_bits(float f) { union { float f; uint32_t u; } v; v.f = f; return v.u; }
WO_HD inline float bfs_bits_f32(uint32_t u) { union { float f; uint32_t u; } v; v.u = u; return v.f; }
// {stress of the parent (non-negative float: bits are monotone), earliest queue position}
WO_HD inline unsigned long long bfs_attr_key(float stress, int32_t pos) { return ((unsigned long long)bfs_f32_bits(stress) << 32) | (uint32_t)(0x7fffffff - pos); }
WO_HD inline int32_t bfs_attr_pos(unsigned long long k) { return 0x7fffffff - (int32_t)(uint32_t)k; }

}  // namespace wo

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
namespace wo {

struct BfsLists { int32_t* cur; int32_t* next; int32_t* curCount; int32_t* nextCount; };

// dist <- init everywhere; carried attributes cleared
__global__ __launch_bounds__(WO_BLOCK) void k_bfs_init(float* dist, float init, float* a0, float* a1, uint8_t* a2, int32_t* pushPos, unsigned long long* attrKey, int32_t N) {
    WO_GRID_STRIDE(r, N) {
        dist[r] = init;
        if (a0) a0[r] = 0.0f;
        if (a1) a1[r] = 0.0f;
        if (a2) a2[r] = 0;
        if (pushPos) pushPos[r] = 0x7fffffff;
        if (attrKey) attrKey[r] = 0ull;
    }
}
// level 0: the seed list (ascending id, built by the host stage) with its start attributes
//   coast:   stressMax = min(1, stress / maxStress), subductMax = subduct, convergent = (btype == 1)      (:478-483)
//   backarc / arc: carried stress = min(1, stress / maxStress)                                             (:607, 1068)
__global__ __launch_bounds__(WO_BLOCK) void k_bfs_seed(int32_t mode, const int32_t* seeds, int32_t n, float* dist, float* a0, float* a1, uint8_t* a2,
                                                        const float* stress, const float* subduct, const int8_t* btype, double maxStress) {
    WO_GRID_STRIDE(i, n) {
        const int32_t r = seeds[i];
        dist[r] = 0.0f;
        if (mode == BFS_COAST || mode == BFS_BACKARC || mode == BFS_ARC) {
            const double v = (double)stress[r] / maxStress;
            a0[r] = (float)(v < 1.0 ? v : 1.0);
        }
        if (mode == BFS_COAST) { a1[r] = subduct[r]; a2[r] = btype[r] == 1 ? 1 : 0; }
    }
}

// fields without attributes: one launch per level, claims by atomicMin on the float bits (all claims of a level write
// the same distance, so exactly one claimant sees the old value and appends the cell)
__global__ __launch_bounds__(WO_BLOCK) void k_bfs_plain(BfsCtx B, int32_t mode, float* dist, const int32_t* cur, const int32_t* curCount, int32_t* next,
                                                         int32_t* nextCount, int32_t* zeroCount, int32_t level) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *zeroCount = 0;
    const int32_t n = *curCount;
    const uint32_t ndBits = bfs_f32_bits((float)level);
    WO_BLOCK_STRIDE(i, valid, n) {
        const int32_t r = valid ? cur[i] : 0;
        const int32_t b = valid ? B.off[r] : 0, deg = valid ? B.off[r + 1] - b : 0;
        for (int k = 0; __any(k < deg); ++k) {
            bool add = false; int32_t nr = -1;
            if (k < deg) {
                nr = B.adj[b + k];
                if (bfs_bits_f32(ndBits) < dist[nr] && bfs_admit(B, mode, nr, r))
                    add = atomicMin(reinterpret_cast<uint32_t*>(dist) + nr, ndBits) > ndBits;
            }
            wave_append(add, nr, next, nextCount);
        }
    }
}

// fields with attributes, phase 1
__global__ __launch_bounds__(WO_BLOCK) void k_bfs_push(BfsCtx B, int32_t mode, const float* dist, const int32_t* cur, const int32_t* curCount,
                                                        int32_t* pushPos, unsigned long long* attrKey, const float* a0, int32_t level) {
    const int32_t n = *curCount;
    const float nd = (float)level;
    WO_GRID_STRIDE(i, n) {
        const int32_t r = cur[i];
        for (int32_t j = B.off[r]; j < B.off[r + 1]; ++j) {
            const int32_t nr = B.adj[j];
            if (!(nd <= dist[nr])) continue;                        // reached at an earlier level
            if (!bfs_admit(B, mode, nr, r)) continue;
            if (nd < dist[nr]) atomicMin(&pushPos[nr], i);          // nd == dist[nr] cannot happen here: distances of this level are written by k_bfs_assign
            if (attrKey) atomicMax(&attrKey[nr], bfs_attr_key(a0[r], i));
        }
    }
}
// phase 2: children per frontier entry (adjacency order)
__global__ __launch_bounds__(WO_BLOCK) void k_bfs_count(BfsCtx B, const float* dist, const int32_t* cur, const int32_t* curCount, const int32_t* pushPos,
                                                         int32_t* cnt, int32_t level) {
    const int32_t n = *curCount;
    const float nd = (float)level;
    WO_GRID_STRIDE(i, n) {
        const int32_t r = cur[i];
        int32_t c = 0;
        for (int32_t j = B.off[r]; j < B.off[r + 1]; ++j) { const int32_t nr = B.adj[j]; if (pushPos[nr] == i && nd < dist[nr]) ++c; }
        cnt[i] = c;
    }
}
// phase 3: exclusive prefix sum over the frontier in queue order (one workgroup of 1024 threads)
__global__ __launch_bounds__(1024) void k_bfs_scan(const int32_t* cnt, const int32_t* curCount, int32_t* base, int32_t* nextCount) {
    __shared__ int32_t s_part[1024];
    const int32_t n = *curCount;
    const int tid = threadIdx.x;
    const int32_t per = (n + 1023) / 1024;
    const int32_t lo = min(n, tid * per), hi = min(n, lo + per);
    int32_t sum = 0;
    for (int32_t i = lo; i < hi; ++i) sum += cnt[i];
    s_part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                              // Hillis-Steele inclusive scan of the 1024 partial sums
        const int32_t v = tid >= o ? s_part[tid - o] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int32_t run = s_part[tid] - sum;
    for (int32_t i = lo; i < hi; ++i) { base[i] = run; run += cnt[i]; }
    if (tid == 1023) *nextCount = s_part[1023];
}
// phase 4: queue positions, distances and attributes of the new level
__global__ __launch_bounds__(WO_BLOCK) void k_bfs_assign(BfsCtx B, int32_t mode, float* dist, const int32_t* cur, const int32_t* curCount, const int32_t* pushPos,
                                                          const unsigned long long* attrKey, const int32_t* base, int32_t* next, float* a0, float* a1, uint8_t* a2,
                                                          int32_t level) {
    const int32_t n = *curCount;
    const float nd = (float)level;
    WO_GRID_STRIDE(i, n) {
        const int32_t r = cur[i];
        int32_t k = base[i];
        for (int32_t j = B.off[r]; j < B.off[r + 1]; ++j) {
            const int32_t nr = B.adj[j];
            if (!(pushPos[nr] == i && nd < dist[nr])) continue;
            next[k++] = nr;
            int32_t src = r;                                           // attributes: the first parent, or (coast) the strongest parent
            if (mode == BFS_COAST) src = cur[bfs_attr_pos(attrKey[nr])];
            if (a0) a0[nr] = a0[src];
            if (a1) a1[nr] = a1[src];
            if (a2) a2[nr] = a2[src];
            dist[nr] = nd;
        }
    }
}

}  // namespace wo
#endif
