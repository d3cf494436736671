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

// the active carve tasks in landCells order: the cells of the current order whose arank is set, order kept (stable selection)
struct CarveActive {
    const int32_t* arank;
    __device__ bool operator()(const int32_t& r) const { return arank[r] != WO_NOT_DONE; }
};
void select_active_by_rank(wo_planet* p, const int32_t* arank, int32_t* out, int32_t* outCount) {
    size_t bytes = p->sortTempBytes;
    hipStream_t s = p->ctx->stream;
    hipEvent_t a = nullptr, b = nullptr;
    if (p->profiling) { a = profile_event(p); b = profile_event(p); WO_HIP(hipEventRecord(a, s)); }
    WO_HIP(hipcub::DeviceSelect::If(p->d_sortTemp, bytes, (const int32_t*)p->d_land[p->landCur], out, outCount, p->L, CarveActive{arank}, s));
    if (p->profiling) { WO_HIP(hipEventRecord(b, s)); p->pending.push_back({FAM_CARVE_SETUP, a, b}); }
}

size_t sort_temp_bytes(int32_t n) {
    size_t bytes = 0;
    hipcub::DoubleBuffer<uint32_t> k(nullptr, nullptr);
    hipcub::DoubleBuffer<int32_t> v(nullptr, nullptr);
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, v, n, 0, 32, nullptr);
    size_t bytes2 = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes2, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const int32_t*)nullptr,
                                             (int32_t*)nullptr, n, 0, WO_LEVEL_BITS, nullptr);
    size_t bytes3 = 0;
    (void)hipcub::DeviceSelect::If(nullptr, bytes3, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, n, CarveActive{nullptr}, nullptr);
    return std::max(bytes, std::max(bytes2, bytes3));
}

// diagnostic (WO_SORT_STATS=1): how far does a cell move in the order from one sort to the next?  rank[] still holds the previous
// positions when this runs.  hist[b]: cells whose displacement d has floor(log2(d + 1)) == b; hist[32]: the maximum.
__global__ __launch_bounds__(WO_BLOCK) void k_sort_displacement(const int32_t* __restrict__ land, const int32_t* __restrict__ rankOld, int32_t L, unsigned long long* hist) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) {
        const int32_t d = abs(i - rankOld[land[i]]);
        atomicAdd(&hist[31 - __clz((uint32_t)d + 1u)], 1ull);
        atomicMax(&hist[32], (unsigned long long)d);
    }
}

void sort_land_by_elevation(wo_planet* p) {
    const int32_t L = p->L;
    const int cur = p->landCur;
    launch(p, FAM_SORT_KEYS, k_sort_keys, blocks_for(L, 4096), WO_BLOCK, (const float*)p->d_e, (const int32_t*)p->d_land[cur],
           p->d_keys[0], L);
    hipStream_t s = p->ctx->stream;
    // WO_SORT=hipcub: the library sort + the rank scatter of earlier builds; default: the in-tree sort (radix.hip), whose last pass
    // also writes rank[cell] = position
    const bool library = getenv("WO_SORT") && std::string(getenv("WO_SORT")) == "hipcub";     // read per sort (tests switch it)
    static const bool stats0 = getenv("WO_SORT_STATS") != nullptr;
    bool rankWritten = false;
    if (!library) {
        uint32_t* const kb[2] = {p->d_keys[0], p->d_keys[1]};
        int32_t* const vb[2] = {p->d_land[cur], p->d_land[cur ^ 1]};
        const int r = radix_sort_pairs(p, FAM_SORT_RADIX, kb, vb, L, 0, 32, stats0 ? (int32_t*)nullptr : p->d_rank, radix_scratch(p, 0), p->N, p->rsFlip[0]);
        p->landCur = r == 0 ? cur : (cur ^ 1);
        rankWritten = !stats0;
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
    static const bool stats = getenv("WO_SORT_STATS") != nullptr;
    if (stats) {
        unsigned long long* d_h = nullptr; unsigned long long h[33];
        WO_HIP(hipMalloc((void**)&d_h, sizeof(h))); WO_HIP(hipMemsetAsync(d_h, 0, sizeof(h), s));
        hipLaunchKernelGGL(k_sort_displacement, dim3(blocks_for(L, 4096)), dim3(WO_BLOCK), 0, s, (const int32_t*)p->d_land[p->landCur], (const int32_t*)p->d_rank, L, d_h);
        WO_HIP(hipMemcpyAsync(h, d_h, sizeof(h), hipMemcpyDeviceToHost, s)); WO_HIP(hipStreamSynchronize(s)); WO_HIP(hipFree(d_h));
        unsigned long long moved = 0; for (int b = 1; b < 32; ++b) moved += h[b];
        fprintf(stderr, "[sort] displacement vs the previous order: %llu of %d cells moved, max %llu; cells by floor(log2(d+1)):", moved, L, h[32]);
        for (int b = 0; b < 24; ++b) fprintf(stderr, " %llu", h[b]);
        fprintf(stderr, "\n");
    }
    if (!rankWritten) launch(p, FAM_RANK, k_rank_scatter, blocks_for(L, 4096), WO_BLOCK, (const int32_t*)p->d_land[p->landCur], p->d_rank, L);
}

}  // namespace wo
