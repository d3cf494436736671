// Stable descending sort of landCells by float32 elevation (reference: the in-place
// landCells.sort((a,b) => r_elevation[b] - r_elevation[a]) at js/terrain-post.js:471,563 — V8's sort is
// stable, so ties keep the previous iteration's order).  A stable LSD radix sort of (key, cell) pairs taken
// in the previous order reproduces that exactly; keys map -0 and +0 to the same value (erode_ops.h).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdio>
#include <string>
#include <cstdlib>

#include "device.h"

namespace wo {

__global__ __launch_bounds__(WO_BLOCK) void k_sort_keys(const float* __restrict__ e, const int32_t* __restrict__ land,
                                                         uint32_t* __restrict__ keys, int32_t L) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x)
        keys[i] = desc_key(e[land[i]]);
}

__global__ __launch_bounds__(WO_BLOCK) void k_rank_scatter(const int32_t* __restrict__ land, int32_t* __restrict__ rank, int32_t L) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x)
        rank[land[i]] = i;
}

// rank[] for the current (e.g. initial ascending-r) order without sorting
void rank_from_land(wo_planet* p) {
    launch(p, FAM_RANK, k_rank_scatter, blocks_for(p->L, 4096), WO_BLOCK, (const int32_t*)p->d_land[p->landCur], p->d_rank, p->L);
}

// ---- solve scheduling: land cells grouped by the round in which they completed last iteration ----
__global__ __launch_bounds__(WO_BLOCK) void k_level_keys(const int32_t* __restrict__ level, const int32_t* __restrict__ landIdx,
                                                          uint32_t* __restrict__ keys, int32_t L, int32_t maxLevel) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) {
        int32_t l = level[landIdx[i]];
        if (l < 1) l = 1;
        if (l > maxLevel) l = maxLevel;
        keys[i] = (uint32_t)l;
    }
}
// start[l] = first position in the sorted key array whose key is >= l  (l = 0 .. maxLevel + 1)
__global__ void k_level_bounds(const uint32_t* __restrict__ keys, int32_t L, int32_t* __restrict__ start, int32_t maxLevel) {
    for (int32_t l = blockIdx.x * blockDim.x + threadIdx.x; l <= maxLevel + 1; l += gridDim.x * blockDim.x) {
        int32_t lo = 0, hi = L;
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (keys[mid] < (uint32_t)l) lo = mid + 1; else hi = mid; }
        start[l] = lo;
    }
}

void sort_by_level(wo_planet* p) {
    const int32_t L = p->L;
    hipStream_t s = p->ctx->stream;
    launch(p, FAM_LEVEL_SORT, k_level_keys, blocks_for(L, 4096), WO_BLOCK, (const int32_t*)p->d_level, (const int32_t*)p->d_landIdx,
           p->d_keys[0], L, (int32_t)WO_MAX_LEVEL);
    size_t bytes = p->sortTempBytes;
    hipEvent_t a = nullptr, b = nullptr;
    if (p->profiling) { a = profile_event(p); b = profile_event(p); WO_HIP(hipEventRecord(a, s)); }
    WO_HIP(hipcub::DeviceRadixSort::SortPairs(p->d_sortTemp, bytes, (const uint32_t*)p->d_keys[0], p->d_keys[1],
                                             (const int32_t*)p->d_landIdx, p->d_byLevel, L, 0, WO_LEVEL_BITS, s));
    if (p->profiling) { WO_HIP(hipEventRecord(b, s)); p->pending.push_back({FAM_LEVEL_SORT, a, b}); }
    launch(p, FAM_LEVEL_SORT, k_level_bounds, blocks_for(WO_MAX_LEVEL + 2, 64), WO_BLOCK, (const uint32_t*)p->d_keys[1], L,
           p->d_levelStart, (int32_t)WO_MAX_LEVEL);
    WO_HIP(hipMemcpyAsync(p->h_levelStart, p->d_levelStart, (WO_MAX_LEVEL + 2) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    WO_HIP(hipStreamSynchronize(s));
}

// The active carve tasks in landCells order: the cells of the current order whose arank is set, order kept (stable selection).
// In-tree compaction, three launches: flagged entries per tile of 2 048, one workgroup scans the tile counts, every tile writes
// its flagged entries behind the tiles before it (a thread takes 8 consecutive entries, so the order inside a tile is kept too).
constexpr int WO_SEL_PER_THREAD = 8, WO_SEL_TILE = WO_BLOCK * WO_SEL_PER_THREAD;
__device__ inline int32_t block_exclusive_scan(int32_t v, int32_t* s_wave, int32_t& total) {      // WO_BLOCK threads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WO_BLOCK / 64; ++w) { const int32_t c = s_wave[w]; if (w < wave) before += c; tot += c; }
    total = tot;
    __syncthreads();
    return before + incl - v;
}
__global__ __launch_bounds__(WO_BLOCK) void k_sel_count(const int32_t* __restrict__ land, const int32_t* __restrict__ arank, int32_t L, int32_t* __restrict__ tileCnt) {
    __shared__ int32_t s_wave[WO_BLOCK / 64];
    const int32_t base = blockIdx.x * WO_SEL_TILE + threadIdx.x * WO_SEL_PER_THREAD;
    int32_t c = 0;
#pragma unroll
    for (int q = 0; q < WO_SEL_PER_THREAD; ++q) { const int32_t i = base + q; if (i < L && arank[land[i]] != WO_NOT_DONE) ++c; }
    int32_t total;
    (void)block_exclusive_scan(c, s_wave, total);
    if (threadIdx.x == 0) tileCnt[blockIdx.x] = total;
}
__global__ __launch_bounds__(WO_BLOCK) void k_sel_scan(int32_t* tileCnt, int32_t nTiles, int32_t* outCount) {      // one workgroup
    __shared__ int32_t s_wave[WO_BLOCK / 64];
    int32_t run = 0;
    for (int32_t b0 = 0; b0 < nTiles; b0 += WO_BLOCK) {
        const int32_t i = b0 + threadIdx.x;
        const int32_t v = i < nTiles ? tileCnt[i] : 0;
        int32_t total;
        const int32_t ex = block_exclusive_scan(v, s_wave, total);
        if (i < nTiles) tileCnt[i] = run + ex;
        run += total;
    }
    if (threadIdx.x == 0) *outCount = run;
}
__global__ __launch_bounds__(WO_BLOCK) void k_sel_scatter(const int32_t* __restrict__ land, const int32_t* __restrict__ arank, int32_t L, const int32_t* __restrict__ tileStart, int32_t* __restrict__ out) {
    __shared__ int32_t s_wave[WO_BLOCK / 64];
    const int32_t base = blockIdx.x * WO_SEL_TILE + threadIdx.x * WO_SEL_PER_THREAD;
    int32_t cell[WO_SEL_PER_THREAD]; bool on[WO_SEL_PER_THREAD];
    int32_t c = 0;
#pragma unroll
    for (int q = 0; q < WO_SEL_PER_THREAD; ++q) { const int32_t i = base + q; cell[q] = i < L ? land[i] : 0; on[q] = i < L && arank[cell[q]] != WO_NOT_DONE; c += on[q] ? 1 : 0; }
    int32_t total;
    int32_t at = tileStart[blockIdx.x] + block_exclusive_scan(c, s_wave, total);
#pragma unroll
    for (int q = 0; q < WO_SEL_PER_THREAD; ++q) if (on[q]) out[at++] = cell[q];
}
void select_active_by_rank(wo_planet* p, const int32_t* arank, int32_t* out, int32_t* outCount) {
    const int32_t L = p->L;
    const int nTiles = (int)(((int64_t)L + WO_SEL_TILE - 1) / WO_SEL_TILE);
    int32_t* tileCnt = reinterpret_cast<int32_t*>(p->d_sortTemp);
    const int32_t* land = p->d_land[p->landCur];
    launch(p, FAM_CARVE_SETUP, k_sel_count, nTiles, WO_BLOCK, land, arank, L, tileCnt);
    launch(p, FAM_CARVE_SETUP, k_sel_scan, 1, WO_BLOCK, tileCnt, (int32_t)nTiles, outCount);
    launch(p, FAM_CARVE_SETUP, k_sel_scatter, nTiles, WO_BLOCK, land, arank, L, (const int32_t*)tileCnt, out);
}

size_t sort_temp_bytes(int32_t n) {
    size_t bytes = 0;
    hipcub::DoubleBuffer<uint32_t> k(nullptr, nullptr);
    hipcub::DoubleBuffer<int32_t> v(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, v, n, 0, 32, nullptr);
    size_t bytes2 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes2, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                                             (int32_t*)nullptr, n, 0, WO_LEVEL_BITS, nullptr);
    const size_t bytes3 = ((size_t)n / WO_SEL_TILE + 2) * sizeof(int32_t);      // select_active_by_rank: tile counts
    return std::max(bytes, std::max(bytes2, bytes3));
}

void sort_land_by_elevation(wo_planet* p) {
    const int32_t L = p->L;
    const int cur = p->landCur;
    // (measured in round 4: the keys made inside the first counting pass of the in-tree sort instead of by a launch of their own — sort stage 34.6 ms
    // per step against 33.1: the counting pass's 679 workgroups gather the heights more slowly than this grid does.  Round 5: the keys written by the
    // thermal step of the previous iteration, keys[rank[cell]] — sort 33.1 -> 28.3 ms per step, thermal 37.6 -> 44.2: profiles/r05z_*; removed)
    launch(p, FAM_SORT_KEYS, k_sort_keys, blocks_for(L, 4096), WO_BLOCK, (const float*)p->d_e, (const int32_t*)p->d_land[cur],
           p->d_keys[0], L);
    hipStream_t s = p->ctx->stream;
    // WO_SORT=hipcub: the library sort + the rank scatter of earlier builds; default: the in-tree sort (radix.hip), whose last pass
    // also writes rank[cell] = position
    const bool library = p->opt.sortLibrary;
    bool rankWritten = false;
    if (!library) {
        uint32_t* const kb[2] = {p->d_keys[0], p->d_keys[1]};
        int32_t* const vb[2] = {p->d_land[cur], p->d_land[cur ^ 1]};
        const int r = radix_sort_pairs(p, FAM_SORT_RADIX, kb, vb, L, 0, 32, p->d_rank, radix_scratch(p, 0), p->N, p->rsFlip[0]);
        p->landCur = r == 0 ? cur : (cur ^ 1);
        rankWritten = true;
    } else {
        hipcub::DoubleBuffer<uint32_t> k(p->d_keys[0], p->d_keys[1]);
        hipcub::DoubleBuffer<int32_t> v(p->d_land[cur], p->d_land[cur ^ 1]);
        hipEvent_t a = nullptr, b = nullptr;
        if (p->profiling) { a = profile_event(p); b = profile_event(p); WO_HIP(hipEventRecord(a, s)); }
        size_t bytes = p->sortTempBytes;
        // (11-bit digits — three passes instead of four, rocPRIM's onesweep with the `match` ranking — were measured: 180 us per pass
        // against 32 us, profiles/r03aj_*)
        WO_HIP(hipcub::DeviceRadixSort::SortPairs(p->d_sortTemp, bytes, k, v, L, 0, 32, s));
        if (p->profiling) { WO_HIP(hipEventRecord(b, s)); p->pending.push_back({FAM_SORT_RADIX, a, b}); }
        p->landCur = (v.Current() == p->d_land[cur]) ? cur : (cur ^ 1);
    }
    if (!rankWritten) launch(p, FAM_RANK, k_rank_scatter, blocks_for(L, 4096), WO_BLOCK, (const int32_t*)p->d_land[p->landCur], p->d_rank, L);
}

}  // namespace wo
