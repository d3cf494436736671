// Stable descending sort of landCells by float32 elevation (reference: the in-place
// landCells.sort((a,b) => r_elevation[b] - r_elevation[a]) at js/terrain-post.js:471,563 — V8's sort is
// stable, so ties keep the previous iteration's order).  A stable LSD radix sort of (key, cell) pairs taken
// in the previous order reproduces that exactly; keys map -0 and +0 to the same value (erode_ops.h).
#include <hip/hip_runtime.h>

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

// scratch of select_active_by_rank: tile counts
size_t sort_temp_bytes(int32_t n) { return ((size_t)n / WO_SEL_TILE + 2) * sizeof(int32_t); }

void sort_land_by_elevation(wo_planet* p) {
    const int32_t L = p->L;
    const int cur = p->landCur;
    // (measured in round 4: the keys made inside the first counting pass of the sort instead of by a launch of their own — sort stage 34.6 ms per step against
    // 33.1: the counting pass's 679 workgroups gather the heights more slowly than this grid does.  Round 5: the keys written by the thermal step of the
    // previous iteration, keys[rank[cell]] — sort 33.1 -> 28.3 ms per step, thermal 37.6 -> 44.2: profiles/r05z_*; removed.  The library sort
    // (hipcub::DeviceRadixSort, 253 us per sort against 150) was a cross-check route until round 6; 11-bit digits with rocPRIM's onesweep: 180 us per pass
    // against 32, profiles/r03aj_*.)
    launch(p, FAM_SORT_KEYS, k_sort_keys, blocks_for(L, 4096), WO_BLOCK, (const float*)p->d_e, (const int32_t*)p->d_land[cur], p->d_keys[0], L);
    // the in-tree stable sort (radix.hip); its last pass also writes rank[cell] = position
    uint32_t* const kb[2] = {p->d_keys[0], p->d_keys[1]};
    int32_t* const vb[2] = {p->d_land[cur], p->d_land[cur ^ 1]};
    const int r = radix_sort_pairs(p, FAM_SORT_RADIX, kb, vb, L, 0, 32, p->d_rank, radix_scratch(p, 0), p->N, p->rsFlip[0]);
    p->landCur = r == 0 ? cur : (cur ^ 1);
}

}  // namespace wo
