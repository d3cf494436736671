// Stable LSD radix sort of (u32 key, i32 value) pairs for the two sorts of an erosion iteration (the order of landCells by elevation,
// js/terrain-post.js:471,563 — sort.hip; the group-major store order of the basin-local solve — basin.hip).
//
// Both sorts are SMALL for this chip (2.8 M pairs = 22 MB at the benched size) and sit on the critical chain of the iteration, so
// what they cost is launches and dependent round trips, not bytes.  The library sort (hipCUB -> rocPRIM onesweep) runs 4 scatter
// passes of ~32 us for the 32-bit keys, but around them a histogram launch, a scan launch and NINE hipMemsetAsync launches (digit
// counters, and per pass the look-back states and the tile counter: ~5.5 us each) — 253 us of stream time per elevation sort with
// the key and rank kernels, of which the passes are half (profiles/r03am_*).  A first in-tree onesweep (tickets + decoupled
// look-back, no memsets) was no faster: handing 679 tiles their numbers through one returning atomic is 13 us of serialised adds
// per pass, and every tile of a pass is resident at once, so a late tile walks back over hundreds of tiles that are all only
// PARTIAL (48 us per pass, profiles/r03ao_*).  What is here has no ticket, no spin and no flag — nothing in it depends on dispatch
// order or residency — at the price of reading the (L2-resident) keys twice per pass:
//   k_rs_count       per pass: the digit counts of every tile of 4096 pairs -> counts[tile][256], and their sums over groups of 32
//                    tiles -> groupTot[tile / 32][256] (atomics on 256 x #groups addresses; two such buffers alternate, the scatter of
//                    one pass clears the buffer of the next: no memset).
//   k_rs_scatter     per pass: tile t ranks its keys per wave by digit matching (8 ballots per key), thread d adds up digit d's pairs in
//                    the earlier tiles (whole groups from groupTot, the rest of its own group from counts: <= 21 + 31 independent loads
//                    at the benched size), the pass's histogram is the sum of ALL group totals and the digit's global base its exclusive scan
//                    (each tile works the 256 numbers out itself: no histogram or scan launch), the tile is put in destination order in LDS and written out in runs.  The
//                    LAST pass also writes value -> position (the rank array), which was a launch of its own.
// Stability: a pair's destination is base[digit] + (pairs with that digit in earlier tiles) + (in earlier waves of the tile) + (earlier
// in the wave, in index order) — the order of equal keys is the input order, as the reference's stable Array.prototype.sort requires.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "device.h"

namespace wo {

namespace {
#ifndef WO_RS_ITEMS
#define WO_RS_ITEMS 16
#endif
constexpr int RS_THREADS = 256, RS_WAVES = 4, RS_ITEMS = WO_RS_ITEMS, RS_TILE = RS_THREADS * RS_ITEMS, RS_DIGITS = 256;

constexpr int RS_GROUP = 32;          // tiles per group of the two-level prefix

__global__ __launch_bounds__(RS_THREADS) void k_rs_count(const uint32_t* __restrict__ keys, int32_t n, int shift, uint32_t* __restrict__ counts, uint32_t* groupTot) {
    __shared__ uint32_t s_c[RS_WAVES][RS_DIGITS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) s_c[w][threadIdx.x] = 0;
    const int32_t tile = blockIdx.x;
    const int32_t first = tile * RS_TILE + wave * (64 * RS_ITEMS) + lane;
    uint32_t key[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) { const int32_t i = first + j * 64; key[j] = i < n ? keys[i] : 0u; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) if (first + j * 64 < n) atomicAdd(&s_c[wave][(key[j] >> shift) & 255u], 1u);
    __syncthreads();
    const uint32_t c = s_c[0][threadIdx.x] + s_c[1][threadIdx.x] + s_c[2][threadIdx.x] + s_c[3][threadIdx.x];
    counts[(size_t)tile * RS_DIGITS + threadIdx.x] = c;
    if (c) atomicAdd(&groupTot[(size_t)(tile / RS_GROUP) * RS_DIGITS + threadIdx.x], c);
}

// posOut (may be null): posOut[value] = destination, written by the last pass
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint32_t* __restrict__ keysIn, const int32_t* __restrict__ valsIn, uint32_t* __restrict__ keysOut,
                                                          int32_t* __restrict__ valsOut, int32_t n, int shift, const uint32_t* __restrict__ counts,
                                                          const uint32_t* __restrict__ groupTot, int32_t groups, uint32_t* clearNext, int32_t* __restrict__ posOut) {
    __shared__ uint32_t s_cnt[RS_WAVES][RS_DIGITS];       // per wave: pairs with the digit (running while the wave ranks its keys), then the wave's base inside the digit
    __shared__ uint32_t s_off[RS_DIGITS];                 // destination of the pair at tile position q with this digit: s_off[digit] + q
    __shared__ uint32_t s_start[RS_DIGITS];               // first tile position of the digit
    __shared__ uint32_t s_wsum[2][RS_WAVES];
    __shared__ uint32_t s_keys[RS_TILE];
    __shared__ int32_t s_vals[RS_TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int32_t tile = blockIdx.x;
    const int32_t tileFirst = tile * RS_TILE;
    const int32_t first = tileFirst + wave * (64 * RS_ITEMS) + lane;
    uint32_t key[RS_ITEMS]; int32_t val[RS_ITEMS]; uint32_t rnk[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int32_t i = first + j * 64;
        const bool ok = i < n;
        key[j] = ok ? keysIn[i] : 0xffffffffu;
        val[j] = ok ? valsIn[i] : 0;
    }
    // thread d: digit d's pairs in the earlier tiles — whole groups, then the earlier tiles of this tile's group (independent loads) —
    // and in all tiles (the pass's histogram is the sum of the group totals)
    uint32_t excl = 0, histD = 0;
    {
        const int d = threadIdx.x;
        const int32_t g = tile / RS_GROUP;
        uint32_t v[16];
        for (int32_t q0 = 0; q0 < groups; q0 += 16) {                  // sixteen loads in flight (one by one they would be dependent round trips)
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (q0 + u < groups) ? groupTot[(size_t)(q0 + u) * RS_DIGITS + d] : 0u;
#pragma unroll
            for (int u = 0; u < 16; ++u) { histD += v[u]; if (q0 + u < g) excl += v[u]; }
        }
#pragma unroll
        for (int h = 0; h < RS_GROUP; h += 16) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int32_t t = g * RS_GROUP + h + u; v[u] = (t < tile) ? counts[(size_t)t * RS_DIGITS + d] : 0u; }
#pragma unroll
            for (int u = 0; u < 16; ++u) excl += v[u];
        }
        if (tile < groups) clearNext[(size_t)tile * RS_DIGITS + d] = 0;        // the other group-total buffer, for the next pass (of this sort or the next)
    }
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) s_cnt[w][threadIdx.x] = 0;
    // exclusive scan of the pass's histogram over the digits: the digit's global base (every tile computes the same 256 numbers)
    uint32_t hbase;
    {
        const uint32_t h = histD;
        uint32_t incl = h;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) s_wsum[0][wave] = incl;
        __syncthreads();                                    // also: s_cnt zeroed
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) if (w < wave) before += s_wsum[0][w];
        hbase = before + incl - h;
    }
    // rank inside the wave, in index order: item j of lane l is index first + 64 j + l
    uint32_t* cnt = s_cnt[wave];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const bool ok = first + j * 64 < n;
        const uint32_t d = (key[j] >> shift) & 255u;
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) { const bool bit = (d >> b) & 1u; const unsigned long long bal = __ballot(bit); m &= bit ? bal : ~bal; }
        const uint32_t before = ok ? cnt[d] : 0;                         // every lane of the group reads before its leader adds
        rnk[j] = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        __builtin_amdgcn_wave_barrier();
        if (ok && lane == __ffsll((long long)m) - 1) cnt[d] = before + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // thread d: bases of the waves inside the digit, the digit's first position in the tile (scan over the digits)
    {
        const int d = threadIdx.x;
        uint32_t c[RS_WAVES], total = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) { c[w] = s_cnt[w][d]; }
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) { s_cnt[w][d] = total; total += c[w]; }
        uint32_t incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o); if (lane >= o) incl += t; }
        if (lane == 63) s_wsum[1][wave] = incl;
        __syncthreads();
        uint32_t before = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) if (w < wave) before += s_wsum[1][w];
        const uint32_t start = before + incl - total;
        s_start[d] = start;
        s_off[d] = hbase + excl - start;
    }
    __syncthreads();
    // the tile in destination order in LDS, then out in runs: consecutive threads write consecutive destinations of a digit
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        if (first + j * 64 >= n) continue;
        const uint32_t d = (key[j] >> shift) & 255u;
        const uint32_t q = s_start[d] + s_cnt[wave][d] + rnk[j];
        s_keys[q] = key[j];
        s_vals[q] = val[j];
    }
    __syncthreads();
    const int32_t have = min((int32_t)RS_TILE, n - tileFirst);
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const int32_t q = j * RS_THREADS + (int32_t)threadIdx.x;
        if (q >= have) continue;
        const uint32_t k = s_keys[q];
        const int32_t v = s_vals[q];
        const uint32_t dst = s_off[(k >> shift) & 255u] + (uint32_t)q;
        keysOut[dst] = k;
        valsOut[dst] = v;
        if (posOut) posOut[v] = (int32_t)dst;
    }
}
}  // namespace

static inline int rs_tiles(int32_t n) { return (int)(((int64_t)n + RS_TILE - 1) / RS_TILE); }

static inline int rs_groups(int32_t n) { return (rs_tiles(n) + RS_GROUP - 1) / RS_GROUP; }
// scratch: [2 group-total buffers: groups x 256 each][tile counts: tiles x 256]
size_t radix_scratch_words(int32_t nMax) { return (size_t)2 * rs_groups(nMax) * RS_DIGITS + (size_t)rs_tiles(nMax) * RS_DIGITS; }

// Sorts n pairs by key bits [beginBit, endBit) (whole 8-bit digits).  keys/vals: two buffers each, input in [0]; returns the index of the
// buffer that holds the result.  posOut (nullable): posOut[value] = final position.  scratch: radix_scratch_words(nMax) u32, ALL ZERO
// before the first call (afterwards every pass clears the group totals the next pass adds into); flip: parity of the passes run so far
// on this scratch, kept by the caller.
int radix_sort_pairs(wo_planet* p, int family, uint32_t* const keys[2], int32_t* const vals[2], int32_t n, int beginBit, int endBit, int32_t* posOut,
                     uint32_t* scratch, int32_t nMax, int& flip) {
    const int passes = (endBit - beginBit + 7) / 8;
    if (passes <= 0 || n <= 0) return 0;
    // every pass clears the group totals the NEXT pass adds into, up to this sort's group count: with an even number of passes per
    // sort the buffer a sort starts on is always the one its predecessor's last pass cleared; an odd count would swap the roles and
    // let a later, larger sort start on totals a smaller one left behind
    if (passes & 1) throw HipError{"radix_sort_pairs: the number of digits must be even"};
    if (n > nMax) throw HipError{"radix_sort_pairs: more pairs than the scratch was sized for"};
    hipStream_t s = cur_stream(p);
    const int tiles = rs_tiles(n), groups = rs_groups(n);
    const size_t gtWords = (size_t)rs_groups(nMax) * RS_DIGITS;
    uint32_t* counts = scratch + 2 * gtWords;
    // profiled runs bracket every kernel on its own (as wo::launch does): a family's time is then the sum of its kernels' durations, comparable
    // with a rocprofv3 kernel trace, and not the length of the whole sort with the gaps a side stream waits in
    auto bracketed = [&](auto&& launchIt) {
        hipEvent_t a = nullptr, b = nullptr;
        if (p->profiling) { a = profile_event(p); b = profile_event(p); WO_HIP(hipEventRecord(a, s)); }
        launchIt();
        if (p->profiling) { WO_HIP(hipEventRecord(b, s)); p->pending.push_back({family, a, b}); if (p->pending.size() >= 4096) profile_resolve(p); }
    };
    int cur = 0;
    for (int q = 0; q < passes; ++q, ++flip) {
        uint32_t* gt = scratch + (size_t)(flip & 1) * gtWords;
        uint32_t* gtOther = scratch + (size_t)((flip + 1) & 1) * gtWords;
        bracketed([&] { hipLaunchKernelGGL(k_rs_count, dim3(tiles), dim3(RS_THREADS), 0, s, (const uint32_t*)keys[cur], n, beginBit + 8 * q, counts, gt); });
        bracketed([&] { hipLaunchKernelGGL(k_rs_scatter, dim3(tiles), dim3(RS_THREADS), 0, s, (const uint32_t*)keys[cur], (const int32_t*)vals[cur], keys[cur ^ 1], vals[cur ^ 1], n,
                           beginBit + 8 * q, (const uint32_t*)counts, (const uint32_t*)gt, (int32_t)groups, gtOther, (q == passes - 1) ? posOut : (int32_t*)nullptr); });
        cur ^= 1;
    }
    WO_HIP(hipGetLastError());
    return cur;
}

uint32_t* radix_scratch(wo_planet* p, int which) {
    if (!p->d_rs[which]) {
        const size_t words = radix_scratch_words(p->N);
        WO_HIP(hipMalloc((void**)&p->d_rs[which], words * 4));
        WO_HIP(hipMemsetAsync(p->d_rs[which], 0, words * 4, cur_stream(p)));
        p->rsFlip[which] = 0;
    }
    return p->d_rs[which];
}

}  // namespace wo
