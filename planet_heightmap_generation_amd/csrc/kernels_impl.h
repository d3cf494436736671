// gfx950 kernels of the terrain-post path.  Each kernel is a grid-stride wrapper around one body of
// erode_ops.h (which documents the reference lines it restates); all launches go through wo::launch so the
// HIP-event profiler sees them.  Wave64, 256-thread workgroups, no MFMA (no dense contraction anywhere on
// this path): the passes are CSR gathers bounded by HBM/L2 bandwidth or, for the dependency rounds, by
// launch latency.
// (Included by planet.hip only: kernels and their host-side launches live in one translation unit.)
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>

#include "device.h"
#include "plates_ops.h"
#include "climate_ops.h"

namespace wo {

#define WO_GRID_STRIDE(i, n) for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += gridDim.x * blockDim.x)

// XCD-aware cell loop for index-order passes.  Workgroup b is observed to run on XCD b % 8 (MI355X_MICROARCH.md);
// with the natural mapping the eight L2s each fetch the same +-4.9*sqrt(N) neighbour windows (measured: ~8x the
// algorithmic bytes in FETCH_SIZE).  Here the blocks are cut into tiles of F.xcdTile consecutive blocks and whole
// tiles are dealt round-robin to the XCDs: an L2 then sees contiguous index ranges (tile + halo), while land and
// ocean bands still spread evenly over the eight dies (one contiguous eighth per XCD was measured slower: the land
// fraction varies with latitude).  Launch with xcd_grid(N, tile) blocks; placement only affects speed, never results.
#define WO_XCD_CELLS(r, n)                                                                                           \
    for (int32_t xi_ = (int32_t)(blockIdx.x >> 3), xt_ = F.xcdTile,                                                  \
                 r = (int32_t)((((xi_ / xt_) * 8 + (int32_t)(blockIdx.x & 7u)) * xt_ + (xi_ % xt_)) * (int32_t)blockDim.x + (int32_t)threadIdx.x), \
                 once_ = 1;                                                                                          \
         once_ && r < (n); once_ = 0)

// The same over the ascending land list: i = position in F.landIdx, r = the cell.  The erosion passes touch land cells
// only (28 % of a synthetic planet); the per-ocean-cell constants they used to rewrite every iteration are set once per
// erodeComposite by k_erode_ocean_init.
#define WO_XCD_LAND(i, r)                                                                                            \
    for (int32_t xi_ = (int32_t)(blockIdx.x >> 3), xt_ = F.xcdTileL,                                                 \
                 i = (int32_t)((((xi_ / xt_) * 8 + (int32_t)(blockIdx.x & 7u)) * xt_ + (xi_ % xt_)) * (int32_t)blockDim.x + (int32_t)threadIdx.x), \
                 once_ = 1;                                                                                          \
         once_ && i < F.L; once_ = 0)                                                                                \
        for (int32_t r = F.landIdx ? F.landIdx[i] : i, once2_ = 1; once2_; once2_ = 0)          /* landIdx == nullptr: the land cells ARE the ids 0 .. L-1 (land-first mirror): one dependent load less per thread */

// Block-uniform strided loop: every thread of the workgroup runs the same number of trips (needed around
// block_append's barriers); `valid` tells whether index i is in range.
#define WO_BLOCK_STRIDE(i, valid, n)                                                              \
    for (int32_t base_ = blockIdx.x * blockDim.x, i = base_ + threadIdx.x, valid = (i < (n));   \
         base_ < (n); base_ += gridDim.x * blockDim.x, i = base_ + threadIdx.x, valid = (i < (n)))

// Append `value` of every thread with `flag` to out[] with ONE global atomic per workgroup (wave ballot +
// LDS scan of the per-wave counts) instead of one per wave on a single hot counter.
__device__ inline void block_append(bool flag, int32_t value, int32_t* out, int32_t* outCount) {
    __shared__ int32_t s_wave[WO_BLOCK / 64];
    __shared__ int32_t s_base;
    const unsigned long long mask = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int prefix = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wave] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        int32_t tot = 0;
        for (int w = 0; w < WO_BLOCK / 64; ++w) { const int32_t c = s_wave[w]; s_wave[w] = tot; tot += c; }
        s_base = tot ? atomicAdd(outCount, tot) : 0;
    }
    __syncthreads();
    if (flag) out[s_base + s_wave[wave] + prefix] = value;
    __syncthreads();
}

// ---------------------------------------------------------------- fields / Jacobi ---------------
__global__ __launch_bounds__(WO_BLOCK) void k_coast(Fields F, uint8_t* coast) {
    WO_XCD_CELLS(r, F.N) coast[r] = coast_flag(F, r);
}
__global__ __launch_bounds__(WO_BLOCK) void k_ocean_from_elev(const float* e, uint8_t* ocean, int32_t N) {
    WO_GRID_STRIDE(r, N) ocean[r] = (e[r] <= 0.0f) ? 1 : 0;
}
__global__ __launch_bounds__(WO_BLOCK) void k_smooth(Fields F, const float* in, float* out, double strength) {
    WO_XCD_CELLS(r, F.N) out[r] = smooth_cell(F, in, r, strength);
}
__global__ __launch_bounds__(WO_BLOCK) void k_sharpen(Fields F, const float* in, const float* orig, float* out, double strength) {
    WO_XCD_CELLS(r, F.N) out[r] = sharpen_cell(F, in, orig, r, strength);
}
__global__ __launch_bounds__(WO_BLOCK) void k_creep(Fields F, const float* in, float* out, double strength) {
    WO_XCD_CELLS(r, F.N) out[r] = creep_cell(F, in, r, strength);
}
__global__ __launch_bounds__(WO_BLOCK) void k_glacial_blend(Fields F, const float* in, float* out) {
    WO_XCD_CELLS(r, F.N) out[r] = glacial_blend_cell(F, in, r);
}

// ---------------------------------------------------------------- noise -------------------------
__device__ inline void load_tables(const uint8_t* tables, uint8_t* sP, uint8_t* sM) {
    for (int i = threadIdx.x; i < 512; i += blockDim.x) { sP[i] = tables[i]; sM[i] = tables[512 + i]; }
    __syncthreads();
}

__global__ __launch_bounds__(WO_BLOCK) void k_noise_eval(const uint8_t* tables, int32_t kind, int32_t octaves, double p0,
                                                          double p1, double p2, int64_t n, const double* xyz, double* out) {
    __shared__ uint8_t sP[512], sM[512];
    load_tables(tables, sP, sM);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        double v;
        if (kind == 0) v = noise3d(sP, sM, x, y, z);
        else if (kind == 1) v = fbm(sP, sM, x, y, z, octaves, p0);
        else v = ridged_fbm(sP, sM, x, y, z, octaves, p0, p1, p2);
        out[i] = v;
    }
}

// SURVEY 8(d) synthetic terrain: e = 0.9*fbm(1.5p,5) - 0.12 + 0.25*ridged(3p,4)*max(0,fbm(1.5p,5))
__global__ __launch_bounds__(WO_BLOCK) void k_synthetic(const uint8_t* tables, const float* xyz, float* e, uint8_t* ocean, int32_t N) {
    __shared__ uint8_t sP[512], sM[512];
    load_tables(tables, sP, sM);
    WO_GRID_STRIDE(r, N) {
        const double x = xyz[3 * r], y = xyz[3 * r + 1], z = xyz[3 * r + 2];
        const double f = fbm(sP, sM, x * 1.5, y * 1.5, z * 1.5, 5);
        const double rg = ridged_fbm(sP, sM, x * 3, y * 3, z * 3, 4);
        const float v = (float)(0.9 * f - 0.12 + 0.25 * rg * (f > 0 ? f : 0));
        e[r] = v;
        ocean[r] = (v <= 0.0f) ? 1 : 0;
    }
}

__global__ __launch_bounds__(WO_BLOCK) void k_warp(Fields F, const uint8_t* tables, const float* in, float* out, double maxAmp,
                                                    double warpBias, const float* hot) {
    __shared__ uint8_t sP[512], sM[512];
    load_tables(tables, sP, sM);
    WO_XCD_CELLS(r, F.N) {
        const int32_t src = warp_source_cell(F, sP, sM, r, maxAmp);
        out[r] = warp_blend(in[r], in[src], warpBias, hot != nullptr, hot ? hot[r] : 0.0f);
    }
}

// ---------------------------------------------------------------- halo pack / unpack (banded Jacobi) ----
__global__ __launch_bounds__(WO_BLOCK) void k_halo_pack(const float* e, const int32_t* idx, int32_t n, float* out) { WO_GRID_STRIDE(i, n) out[i] = e[idx[i]]; }
__global__ __launch_bounds__(WO_BLOCK) void k_halo_unpack(float* e, const int32_t* idx, int32_t n, const float* in) { WO_GRID_STRIDE(i, n) e[idx[i]] = in[i]; }

// ---------------------------------------------------------------- climate-util smoothField ------
__global__ __launch_bounds__(WO_BLOCK) void k_smooth_field(Fields F, const float* src, float* dst) {
    WO_XCD_CELLS(r, F.N) dst[r] = smooth_field_cell(F, src, r);
}

// ---------------------------------------------------------------- climate sweeps (js/temperature.js, js/precipitation.js) ----
__global__ __launch_bounds__(WO_BLOCK) void k_warmth_seed(const float* warmth, const uint8_t* isLand, float* out, int32_t N) {
    WO_GRID_STRIDE(r, N) out[r] = warmth_seed_cell(warmth, isLand, r);
}
__global__ __launch_bounds__(WO_BLOCK) void k_warmth_diffuse(Fields F, ClimateMesh M, const float* src, const float* cont, float* dst) {
    WO_XCD_CELLS(r, F.N) dst[r] = warmth_diffuse_cell(M, src, cont, r);
}
__global__ __launch_bounds__(WO_BLOCK) void k_wind_convergence(Fields F, ClimateMesh M, const float* wx, const float* wy, const float* wz, float* out) {
    WO_XCD_CELLS(r, F.N) out[r] = wind_convergence_cell(M, wx, wy, wz, r);
}
__global__ __launch_bounds__(WO_BLOCK) void k_moisture_seed(Fields F, ClimateMesh M, const uint8_t* isLand, const float* wx, const float* wy, const float* wz,
                                                             const float* warmth, const int32_t* coastDist, float* out) {
    WO_XCD_CELLS(r, F.N) out[r] = moisture_seed_cell(M, isLand, wx, wy, wz, warmth, coastDist, r);
}
__global__ __launch_bounds__(WO_BLOCK) void k_moisture_advect(Fields F, ClimateMesh M, const float* src, const float* heightKm, const uint8_t* isLand,
                                                               const float* windE, const float* windN, const float* wx, const float* wy, const float* wz,
                                                               int32_t maxHops, double depletionBase, float* dst) {
    WO_XCD_CELLS(r, F.N) dst[r] = moisture_advect_cell(M, src, heightKm, isLand, windE, windN, wx, wy, wz, maxHops, depletionBase, r);
}

// ---------------------------------------------------------------- plate projection --------------
// js/coarse-plates.js:51-117: one thread per hi-res cell (12 noise3D from LDS tables, then a greedy ascent over the
// 20 k-cell coarse mesh, which stays in L2); the start cells come from a small (z, longitude) bucket grid.
__global__ __launch_bounds__(WO_BLOCK) void k_plate_grid(CoarsePlates C, int32_t* grid) {
    WO_GRID_STRIDE(b, C.gridZ * C.gridLon) grid[b] = plate_grid_cell(C, b);
}
__global__ __launch_bounds__(WO_BLOCK) void k_plate_project(CoarsePlates C, const uint8_t* tables, const float* r_xyz, int32_t N, double perturbAmp,
                                                             int32_t* r_plate) {
    __shared__ uint8_t sP[512], sM[512];
    load_tables(tables, sP, sM);
    WO_GRID_STRIDE(r, N) r_plate[r] = plate_project_cell(C, sP, sM, r_xyz, r, perturbAmp);
}

// ---------------------------------------------------------------- land list ---------------------
__global__ __launch_bounds__(WO_BLOCK) void k_init_rank(int32_t* rank, int32_t N) { WO_GRID_STRIDE(r, N) rank[r] = -1; }

// ---------------------------------------------------------------- hydraulic ---------------------
// receivers + the start state of the flow accumulation (k_flow_init) in one pass over the cells
// per-ocean-cell constants of the hydraulic and thermal passes (the ocean mask is fixed during an erodeComposite)
__global__ __launch_bounds__(WO_BLOCK) void k_erode_ocean_init(Fields F) {
    WO_GRID_STRIDE(r, F.N) {
        if (!F.ocean[r]) continue;
        if (F.target) F.target[r] = -1;
        TargetRank z; z.target = -1; z.rank = -1; F.tr[r] = z;
        if (F.accA) F.accA[r] = 0;
        F.jumpA[r] = -1; F.flow[r] = 0.0f; F.totalExcess[r] = 0.0; F.me[r] = INFINITY;
    }
}
// ---- neighbour cells staged into LDS ----
// Under the patch-major mirror the land list is in Morton order of all cells: the 256 land cells of a workgroup span ~1 500
// consecutive cell ids, and 94 % of their neighbours lie inside that span widened by 256 ids on either side (measured on the
// 10 M-cell bench planet: 87 % with no margin, 97 % with 1 024).  So the workgroup copies that stretch of the field into LDS with
// coalesced loads (~8 KB) and the ~1 800 neighbour gathers of the tile become LDS reads; the few neighbours outside take the
// global load.  A tile whose span does not fit (index layout, WO_LAYOUT=index: neighbours are +-15 000 ids away) is not staged.
// MEASURED (10 M cells, profiles/r03q_*): no gain — receivers 98 -> 100 us per launch, thermal_excess 82 -> 86, thermal_apply 160 -> 181:
// under the mirror those gathers already hit the vector cache, and the passes wait on their chain of dependent loads (offsets ->
// row -> values) and, for thermal_apply, on f64 arithmetic, not on the gathers; the staging adds a barrier and ~8 KB of loads per
// workgroup.  Off by default (Fields::tileLds, WO_TILE_LDS=1 switches it on; results identical either way).
constexpr int WO_TILE_WIN = 4096;                // floats of LDS per workgroup
constexpr int WO_TILE_MARGIN = 256;
struct TileWindow {
    const float* g; const float* s; int32_t lo, hi;      // s[k] = g[lo + k] for lo <= lo + k <= hi; empty: hi < lo
    __device__ inline float operator()(int32_t c) const { return (c >= lo && c <= hi) ? s[c - lo] : g[c]; }
};
// all threads of the workgroup call this (barrier inside); i0 = the workgroup's first index into the land list
__device__ inline TileWindow stage_tile(const Fields& F, const float* field, float* s_buf, int32_t i0) {
    TileWindow W; W.g = field; W.s = s_buf; W.lo = 1; W.hi = 0;
    if (!F.tileLds) return W;                                     // block-uniform
    if (i0 < F.L) {
        const int32_t i1 = min(i0 + (int32_t)blockDim.x, F.L) - 1;
        const int32_t lo = max((F.landIdx ? F.landIdx[i0] : i0) - WO_TILE_MARGIN, 0), hi = min((F.landIdx ? F.landIdx[i1] : i1) + WO_TILE_MARGIN, F.N - 1);
        if (hi - lo + 1 <= WO_TILE_WIN) {
            for (int32_t k = threadIdx.x; k <= hi - lo; k += blockDim.x) s_buf[k] = field[lo + k];
            W.lo = lo; W.hi = hi;
        }
    }
    __syncthreads();
    return W;
}
// the workgroup's first index into the land list under WO_XCD_LAND's block-to-tile mapping
#define WO_XCD_LAND_BASE() ((int32_t)(((((int32_t)(blockIdx.x >> 3) / F.xcdTileL) * 8 + (int32_t)(blockIdx.x & 7u)) * F.xcdTileL + ((int32_t)(blockIdx.x >> 3) % F.xcdTileL)) * (int32_t)blockDim.x))

__global__ __launch_bounds__(WO_BLOCK) void k_receivers_flow_init(Fields F, int32_t* donorCnt) {
    __shared__ float s_tile[WO_TILE_WIN];
    const TileWindow E = stage_tile(F, F.e, s_tile, WO_XCD_LAND_BASE());
    WO_XCD_LAND(i, r) {
        const int32_t t = receiver_cell_t(F, r, E);
        int32_t j = -1; const uint32_t a = 1;
        if (t >= 0 && !F.ocean[t] && F.rank[r] < F.rank[t]) j = t;      // flow_forward_target
        if (F.accA) F.accA[r] = a;                           // (the pointer doubling's accumulator: nullptr on the default route, where k_flow_climb retires every cell)
        F.jumpA[r] = j; F.accCnt[r] = 1ull;
        if (j >= 0 && donorCnt) atomicAdd(&donorCnt[j], 1);  // donorCnt is all zero on entry (k_flow_final leaves it so); nullptr: the tile route counts a cell's donors itself (k_flow_tiles)
        if (F.basinJ) {                                      // start state of the drainage-component search (basin.hip: k_basin_init's job, one launch less on the layout's chain)
            const bool landT = t >= 0 && !F.ocean[t];
            const int32_t sr = F.basinMslot ? F.basinMslot[r] : r;
            F.basinJ[sr] = landT ? (F.basinMslot ? F.basinMslot[t] : t) : sr;
        }
    }
}
// Flow accumulation on the planet's own cell order (WO_LAYOUT=index; under the land-first mirror: k_flow_tiles below).  Subtree sizes are integers, so any
// order of the additions is exact.  ONE launch: every leaf hands its total to its receiver, and the thread whose hand-over completes a receiver (last
// donor in) carries on with that receiver.  A cell's running total and the number of donors that have arrived share one 64-bit word (accCnt), so ONE
// returning atomic both delivers a total and tells the deliverer whether it was the last — then old total + its own contribution IS the cell's total: no
// ordering between two atomics to arrange, one memory round trip per step.  (Round 2's rake rounds + pointer doubling, 108 ms of flow stage per step
// against 45, and the capped climb between them were cross-check routes until round 6.)
__global__ __launch_bounds__(WO_BLOCK) void k_flow_climb(Fields F, const int32_t* donorCnt, int32_t cap) {
    WO_XCD_LAND(i, r) {
        int32_t d = r;
        if (!(donorCnt[d] == 0 && F.jumpA[d] >= 0)) continue;
        unsigned long long v = 1;                                 // a leaf's total
        int32_t j = F.jumpA[d];
        for (int32_t step = 0; step < cap; ++step) {
            // a hop is two round trips: {the receiver's donor count, ITS receiver, the hand-over atomic} go out together (the first two
            // are fixed since the receivers pass: reading them before the hand-over is decided costs nothing), then the next hop's
            const int32_t donors = donorCnt[j];
            const int32_t jj = F.jumpA[j];                        // only -2 ("retired") is ever written here, by the thread that retires j: not before this hop's atomic
            const unsigned long long old = atomicAdd(&F.accCnt[j], (1ull << 32) | v);
            F.jumpA[d] = -2;                                      // retired
            if ((int32_t)(old >> 32) + 1 != donors) break;        // other donors of j are still out
            if (jj < 0) break;                                    // a root keeps the sum
            d = j;
            j = jj;
            v = (old & 0xffffffffull) + v;
        }
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// Flow accumulation in two levels (default under the land-first mirror; WO_FLOW=climb: k_flow_climb alone).
// k_flow_climb pays one returning device-scope atomic — a round trip to the memory side, ~0.5 us under load — per cell of the longest
// flow path (a few hundred cells at 10 M: ~200 us per launch with the chip idle).  The land cells are numbered in Morton order, so a TILE
// of FT_CELLS consecutive ids is a compact patch and ~95 % of the forward edges stay inside one.  Hence:
//   1. k_flow_tiles<false>  one workgroup per tile: the forest restricted to the tile's own edges, in LDS — child counts, the local root
//                           of every cell (pointer jumping), and the same last-arriver climb as k_flow_climb on LDS atomics (~0.1 us per
//                           hop).  Out: lr[c] = local root of c, acc[R] = cells of R's local tree for every local root R.
//   2. k_flow_root_links    a local root whose receiver lies in another tile is a child of THAT cell's local root: arrival counts.
//   3. k_flow_root_climb    k_flow_climb on the forest of local roots (its depth is the number of tiles a river crosses: tens, not
//                           hundreds); every root's total W is also added to inflow[x] of the cell x it drains into.
//   4. k_flow_tiles<true>   the local climb again with the weight 1 + inflow[c] per cell: the forward total of EVERY cell (-> accCnt).
// Integer sums over the same forest: any order of the additions is exact (test_flow_accumulation_routes_agree).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef WO_FT_CELLS
#define WO_FT_CELLS 1024
#endif
#ifndef WO_FT_THREADS
#define WO_FT_THREADS 256
#endif
constexpr int FT_CELLS = WO_FT_CELLS, FT_THREADS = WO_FT_THREADS, FT_PER = FT_CELLS / FT_THREADS;
static_assert(FT_CELLS % FT_THREADS == 0 && FT_CELLS < 0xffff, "tile cells are 16-bit indices");
struct FlowTiles {
    int32_t* lr;                    // [L] local root of a land cell
    unsigned long long* rootAcc;    // [L] per local root: {arrived children roots, running total}
    uint32_t* inflow;               // [L] totals of the roots of other tiles that drain into this cell (zero between passes)
    int32_t* extCnt;                // [L] per local root: roots of other tiles that drain into its tree (zero between passes)
    int32_t* parent;                // [L] per local root: the local root of the cell it drains into, -1: none
    int32_t* basinJ;                // nullable: start state of the basin layout's component search (basin.hip), shortened in place by k_flow_tiles<false>
};
template <bool FINAL>
__global__ __launch_bounds__(FT_THREADS) void k_flow_tiles(Fields F, FlowTiles T) {
    __shared__ uint32_t s_pc[FT_CELLS];                // low 16 bits: parent inside the tile (0xffff: none / another tile), high 16: children inside the tile
    __shared__ unsigned long long s_acc[FT_CELLS];     // {arrived children, running total} as in k_flow_climb
    __shared__ uint16_t s_lr[FINAL ? 1 : FT_CELLS];
    __shared__ uint16_t s_b[FINAL ? 1 : FT_CELLS];       // basin layout: receiver of the cell inside the tile (any receiver edge, not only forward ones), or the cell itself
    __shared__ int32_t s_bext[FINAL ? 1 : FT_CELLS];     //    and what J held for the cell
    const int tid = threadIdx.x;
    const int32_t base = (int32_t)blockIdx.x * FT_CELLS;
    const int32_t n = min((int32_t)FT_CELLS, F.L - base);
#pragma unroll
    for (int q = 0; q < FT_PER; ++q) {
        const int i = tid + q * FT_THREADS;
        if (i >= n) continue;
        const int32_t j = F.jumpA[base + i];
        const bool local = j >= base && j < base + n;
        s_pc[i] = local ? (uint32_t)(j - base) : 0xffffu;
        uint32_t w = 1u;
        if (FINAL) { w += T.inflow[base + i]; T.inflow[base + i] = 0u; T.extCnt[base + i] = 0; }
        s_acc[i] = (unsigned long long)w;
        if (!FINAL) s_lr[i] = local ? (uint16_t)(j - base) : (uint16_t)i;
        if (!FINAL && T.basinJ) {
            const int32_t jb = T.basinJ[base + i];
            s_b[i] = (jb >= base && jb < base + n) ? (uint16_t)(jb - base) : (uint16_t)i;
            s_bext[i] = jb;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FT_PER; ++q) {
        const int i = tid + q * FT_THREADS;
        if (i >= n) continue;
        const uint32_t p = s_pc[i] & 0xffffu;
        if (p != 0xffffu) atomicAdd(&s_pc[p], 1u << 16);
    }
    __syncthreads();
    // the climb: a cell without children hands its total to its parent; whoever completes a parent carries on with it
#pragma unroll
    for (int q = 0; q < FT_PER; ++q) {
        const int i = tid + q * FT_THREADS;
        if (i >= n) continue;
        const uint32_t pc0 = s_pc[i];
        if ((pc0 >> 16) != 0u || (pc0 & 0xffffu) == 0xffffu) continue;
        uint32_t p = pc0 & 0xffffu;
        uint32_t v = (uint32_t)s_acc[i];
        for (;;) {
            const uint32_t pcp = s_pc[p];                         // fixed since the barrier above
            const unsigned long long old = atomicAdd(&s_acc[p], (1ull << 32) | (unsigned long long)v);
            if ((uint32_t)(old >> 32) + 1u != (pcp >> 16)) break; // other children of p are still out
            v += (uint32_t)old;
            p = pcp & 0xffffu;
            if (p == 0xffffu) break;                              // p was a local root
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < FT_PER; ++q) {
        const int i = tid + q * FT_THREADS;
        if (i >= n) continue;
        const uint32_t total = (uint32_t)s_acc[i];
        if (FINAL) F.accCnt[base + i] = (unsigned long long)total;
        else {
            // local root: asynchronous pointer jumping in place, no barrier (whatever a thread reads in s_lr is an ancestor of the cell it
            // stands on, so concurrent chases only shorten each other's paths; a local forest has no cycle: forward edges follow the ranks)
            uint16_t r = s_lr[i];
            for (int it = 0; it < FT_CELLS; ++it) {
                const uint16_t rr = s_lr[r];
                if (rr == r) break;
                s_lr[i] = rr;                                    // (16-bit LDS stores: a reader sees the old or the new ancestor)
                r = rr;
            }
            T.lr[base + i] = base + (int32_t)r;
            if ((s_pc[i] & 0xffffu) == 0xffffu) T.rootAcc[base + i] = (unsigned long long)total;
            if (T.basinJ) {
                // The layout's component search (basin.hip: basin_root) accepts in J[x] ANY ancestor of x and chases from there.  Every forward
                // edge is an edge of the layout's forest too, so x's path there runs through its local root r found above; what r points at
                // decides: a cell of another tile (or r itself: a root) -> J[x] <- that.  If r's receiver lies inside the tile (an edge that
                // is not a forward one: a donor ranked after its receiver, ~1 %) the walk goes on READ-ONLY (halving the pointers in place
                // would split a ring of four cells draining into each other — they occur on flats — into two rings of two) until it
                // leaves the tile; a walk that comes back to r (a ring) or is not over after 64 steps ends at r: J[x] <- r, and r keeps its own.
                const uint16_t b0 = s_b[r];
                int32_t out = s_bext[r];
                if (b0 != r) {
                    int32_t res = -1;
                    uint16_t c = b0;
                    for (int it = 0; it < 64; ++it) {
                        const uint16_t cc = s_b[c];
                        if (cc == c) { res = s_bext[c]; break; }
                        if (cc == r) break;
                        c = cc;
                    }
                    out = res >= 0 ? res : ((uint16_t)i == r ? s_bext[i] : base + (int32_t)r);
                }
                T.basinJ[base + i] = out;
            }
        }
    }
}
__global__ __launch_bounds__(WO_BLOCK) void k_flow_root_links(Fields F, FlowTiles T) {
    WO_GRID_STRIDE(i, F.L) {
        if (T.lr[i] != i) continue;
        const int32_t x = F.jumpA[i];
        int32_t P = -1;
        if (x >= 0) { P = T.lr[x]; atomicAdd(&T.extCnt[P], 1); }
        T.parent[i] = P;                                          // the local root this root's tree drains into (-1: none)
    }
}
__global__ __launch_bounds__(WO_BLOCK) void k_flow_root_climb(Fields F, FlowTiles T) {
    WO_GRID_STRIDE(i, F.L) {
        if (T.lr[i] != i || T.extCnt[i] != 0) continue;
        int32_t P = T.parent[i];
        if (P < 0) continue;
        int32_t x = F.jumpA[i];
        uint32_t v = (uint32_t)T.rootAcc[i];
        for (;;) {
            // one round trip per hop, as in k_flow_climb: everything the NEXT hop needs of P (fixed since k_flow_root_links) goes out with the hand-over
            const int32_t need = T.extCnt[P];
            const int32_t PP = T.parent[P];
            const int32_t xx = F.jumpA[P];
            atomicAdd(&T.inflow[x], v);
            const unsigned long long old = atomicAdd(&T.rootAcc[P], (1ull << 32) | (unsigned long long)v);
            if ((int32_t)(old >> 32) + 1 != need) break;
            if (PP < 0) break;
            v += (uint32_t)old;
            P = PP; x = xx;
        }
    }
}


__global__ __launch_bounds__(WO_BLOCK) void k_flow_final(Fields F, int32_t* donorCnt, SolveOut* clearOut) {
    WO_XCD_LAND(i, c) {
        if (donorCnt) donorCnt[c] = 0;                      // for the next iteration's receivers pass (only land cells are counted into)
        if (clearOut) { SolveOut z; z.self.v = 0; z.self.tag = 0; z.dep.v = 0; z.dep.tag = 0; clearOut[i] = z; }      // the solve's output tags of this pass (store index i: one coalesced sweep, was a memset launch)
        flow_final_cell(F, c);                              // + the event list of c for the solve
    }
}


// The same setup, written for the memory pipeline.  solve_setup_cell decides what to load next from what it has just loaded (has a
// receiver? is it land? did a list overflow?), so the compiler must wait for every load before the branch that follows it: the ISA
// was load, s_waitcnt vmcnt(0), branch, load, ... — ~85 loads per wave one after the other, 56 000 cycles per wave with the
// vector ALU 4 % busy (profiles/r03af_sq_counters_*.json), and the row-scan path of the ~3 % of tasks with an overflowed event list
// sat inside 86 % of the waves.  Here every level of the chain r -> receiver -> its receiver -> predecessors' slots is loaded
// unconditionally from a clamped index (the task's own cell where there is no such cell), so each level's loads are in flight
// together and nothing is decided before the end; tasks that need the row scans are put on a list and set up by
// k_solve_setup_deferred, in their own waves.  Same record, field for field, as solve_setup_cell.
template <bool SLOT>
__device__ inline bool solve_setup_cell_batched(const Fields& F, int32_t r) {
    // level 1: the task's own cell
    const TargetRank trr = F.tr[r];
    const EventList Er = F.ev[r];
    const float e0r = F.e[r], flow = F.flow[r], cdR = F.cellDist[r];
    const int32_t si = SLOT ? F.slotOf[r] : r;
    const int32_t t = trr.target, rr = trr.rank;
    const int32_t tc = t >= 0 ? t : r;
    // level 2: the receiver; the slot of the latest earlier event on r
    const TargetRank trt = F.tr[tc];
    const EventList EtL = F.ev[tc];
    const float cdTL = F.cellDist[tc], e0tL = F.e[tc];
    const int32_t p0 = event_before(Er, r, rr);
    const int32_t p0c = p0 >= 0 ? p0 : r;
    // everything else a list is needed for is taken from it as soon as it is there, so that no list stays live across the levels below
    const bool overflowR = Er.rank[0] == -2;
    int32_t lastR = -1;
#pragma unroll
    for (int q = 0; q < WO_EVENTS; ++q) if (Er.rank[q] >= 0) lastR = Er.cell[q];
    const int32_t s0 = SLOT ? F.slotOf[p0c] : p0c;
    const bool tLand = t >= 0 && trt.rank >= 0;                     // ocean cells carry rank -1
    const float cdT = tLand ? cdTL : 0.0f;
    const int32_t t2 = (tLand && trt.target >= 0 && cdT > 0) ? trt.target : -1;
    const int32_t t2c = t2 >= 0 ? t2 : r;
    // level 3: the receiver's receiver; the slot of the latest earlier event on t
    const TargetRank trt2 = F.tr[t2c];
    const EventList Et2L = F.ev[t2c];
    const float e0t2L = F.e[t2c];
    const int32_t p1 = tLand ? event_before(EtL, r, rr) : -1;
    const int32_t p1c = p1 >= 0 ? p1 : r;
    const bool overflowT = tLand && EtL.rank[0] == -2;
    int32_t lastT = -1;
#pragma unroll
    for (int q = 0; q < WO_EVENTS; ++q) if (tLand && EtL.rank[q] >= 0) lastT = EtL.cell[q];
    const int32_t s1 = SLOT ? F.slotOf[p1c] : p1c;
    const bool t2Land = t2 >= 0 && trt2.rank >= 0;
    // level 4: the slot of the latest earlier event on t2
    const int32_t p2 = t2Land ? event_before(Et2L, r, rr) : -1;
    const int32_t p2c = p2 >= 0 ? p2 : r;
    const int32_t s2 = SLOT ? F.slotOf[p2c] : p2c;
    if (overflowR || overflowT || (t2Land && Et2L.rank[0] == -2)) return false;     // row scans: deferred
    SolveTask T;
    T.predSelf = p0 >= 0 ? 2 * s0 + (p0 == r ? 0 : 1) : -1;
    T.predT = -1; T.predT2 = -1; T.flags = 0; T.pad_[0] = T.pad_[1] = 0;
    T.e0r = e0r; T.e0t = 0; T.e0t2 = 0; T.cellDistT = 0;
    T.factor = solve_factor_of(flow, cdR, F.solveK, F.solveM, F.solveDt);
    if (t >= 0) {
        T.flags |= 4u;
        T.e0t = e0tL;
        if (!tLand) T.flags |= 1u;
        else {
            T.predT = p1 >= 0 ? 2 * s1 + (p1 == t ? 0 : 1) : -1;
            T.cellDistT = cdT;
            if (t2 >= 0) {
                T.flags |= 8u;
                T.e0t2 = e0t2L;
                if (!t2Land) T.flags |= 2u;
                else T.predT2 = p2 >= 0 ? 2 * s2 + (p2 == t2 ? 0 : 1) : -1;
            }
        }
    }
    if (F.solveFinals) {
        // which task leaves the last event on r and on t: the last valid entry of the (descending-rank) lists (lastR, lastT above)
        T.pad_[0] = r; T.pad_[1] = tLand ? t : -1;
        if (lastR == r) T.flags |= 16u;
        if (tLand && lastT == r) T.flags |= 32u;
        if (lastR < 0) { F.e2[r] = e0r; F.me[r] = e0r; }                  // no event on r at all: its height stays
    }
    F.task[si] = T;
    if (!F.solveLean) {
        SolveOut z; z.self.v = 0; z.self.tag = 0; z.dep.v = 0; z.dep.tag = 0;
        F.out[si] = z;
        if (F.blk) F.blk[si] = T.predT >= 0 ? T.predT : (T.predSelf >= 0 ? T.predSelf : T.predT2);
    }
    return true;
}
// the same finality flags for a task set up by the row scans (deferred tasks: an event list involved overflowed)
__device__ inline void solve_setup_finals_by_rows(const Fields& F, int32_t r) {
    const int32_t si = F.slotOf ? F.slotOf[r] : r;
    SolveTask T = F.task[si];
    const int32_t t = F.tr[r].target;
    const bool tLand = t >= 0 && F.tr[t].rank >= 0;
    const int32_t lastR = latest_event_cell(F, r), lastT = tLand ? latest_event_cell(F, t) : -1;
    T.pad_[0] = r; T.pad_[1] = tLand ? t : -1;
    if (lastR == r) T.flags |= 16u;
    if (tLand && lastT == r) T.flags |= 32u;
    if (lastR < 0) { const float e0 = F.e[r]; F.e2[r] = e0; F.me[r] = e0; }
    F.task[si] = T;
}
// Tasks that need the row scans are collected per workgroup (LDS) and set up by the workgroup's first lanes once its main pass is
// over: the slow path then runs in one partly filled wave per workgroup instead of inside nearly every wave, and no global
// counter is involved (one returning atomic per wave on a single word was measured to serialise the whole launch: 374 us).
// (occupancy hints of 5 / 6 / 8 waves per SIMD make this kernel spill on its hot path: 32.4 / 33.3 / 38.3 ms per step against 30.0 at the 114 VGPRs it takes by itself)
#ifndef WO_SETUP_WAVES
#define WO_SETUP_WAVES 1
#endif
template <bool SLOT>
__global__ __launch_bounds__(WO_BLOCK, WO_SETUP_WAVES) void k_solve_setup_batched(Fields F, int32_t* zeroA, int32_t nA, int32_t* zeroB, int32_t nB) {
    __shared__ int32_t s_deferred[WO_BLOCK];
    __shared__ int32_t s_n;
    // the solve launch's counters (tasks left pending per patch, per launch) start at zero: cleared here instead of by two launches of their own
    for (int32_t z = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x); z < nA + nB; z += (int32_t)(gridDim.x * blockDim.x)) { if (z < nA) zeroA[z] = 0; else zeroB[z - nA] = 0; }
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int32_t i = WO_XCD_LAND_BASE() + (int32_t)threadIdx.x;
    int32_t r = -1;
    bool defer = false;
    if (i < F.L) { r = F.landIdx ? F.landIdx[i] : i; defer = !solve_setup_cell_batched<SLOT>(F, r); }
    const unsigned long long m = __ballot(defer);
    if (m) {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
        int32_t base = 0;
        if (lane == leader) base = atomicAdd(&s_n, (int32_t)__popcll(m));
        base = __shfl(base, leader);
        if (defer) s_deferred[base + __popcll(m & ((1ull << lane) - 1ull))] = r;
    }
    __syncthreads();
    if ((int32_t)threadIdx.x < s_n) { solve_setup_cell_rows(F, s_deferred[threadIdx.x]); if (F.solveFinals) solve_setup_finals_by_rows(F, s_deferred[threadIdx.x]); }
}

// The tail of the solve DAG: once only river main stems are left (a few hundred tasks per round for hundreds of
// rounds) a kernel launch per round is all latency.  One 1024-thread workgroup then runs the remaining rounds
// itself: same tasks, same "consume only earlier rounds" rule, but a round boundary is a __syncthreads()
// (workgroup-scope visibility of the granules written through to L2) instead of a kernel boundary.
// Pending lists live in LDS.  stats[0] = last round run, stats[1] = 1 if the LDS lists overflowed (the host then
// rebuilds the pending list from the round tags and goes back to one launch per round).
constexpr int WO_TAIL_THREADS = 1024;
constexpr int WO_TAIL_CAP = 14336;          // 2 lists x 56 KiB of the CU's 160 KiB LDS

// ---------------------------------------------------------------------------------------------------------
// Patch-local solve.  Land cells are grouped in spatial patches of WO_PATCH cells (Morton order); one workgroup
// owns one patch and keeps the {value, tag} granules of its tasks in LDS.  In a launch a workgroup runs every task
// whose predecessors are (a) outside the patch and produced by an EARLIER launch, or (b) inside the patch and already
// produced — by an earlier launch or earlier in this visit (visible through LDS).  So a chain segment that stays inside a
// patch advances at LDS latency and only patch-crossing edges wait for a kernel boundary.  The dataflow is
// single-assignment, so the bits do not depend on the schedule (same results as the level-synchronous rounds; checked by
// the parity tests).  patchPending[p] = tasks of patch p still waiting.
//
// A visit, in order:
//  1. settle.  F.blk remembers, per task, one granule that was unresolved when the task last failed (initially the receiver's
//     event).  From those words alone every thread decides whether its task is certainly still blocked: it is if the
//     remembered blocker is an external granule that is not there yet, or a patch-local task that is itself certainly
//     blocked (chains settle by polling LDS).  Blocked tasks mark their granules BLOCKED (-1, LDS only).
//  2. hand-out.  The other pending tasks ("runnable") are handed to threads 0..n-1 through an LDS list, and every wave that
//     got none RETURNS: at 10 M cells a visit finds 36-50 runnable tasks among the patch's (250 in the first launch), and a
//     finished wave frees its slot for the next workgroup.  No barrier follows this point.
//  3. run.  A runnable task loads its 48-byte record, checks its external predecessors (not there: it blocks and remembers
//     which), then polls its patch-local predecessors in LDS and runs as soon as they are there.  A granule is one aligned
//     64-bit LDS word {value, tag}, published and read with single accesses: no fence sits on the chain.  A task that finds a
//     BLOCKED predecessor blocks itself; a runnable task's granules stay open (tag 0) until it has run or blocked.
//  4. the last wave to finish (LDS counter) writes the patch's pending count.
// The polling loop is capped (spinCap passes): what is still open then simply stays pending.  A launch lasts as long as its
// slowest visit, and a few patches hold chains of ~300 steps while the typical visit needs 14, so short visits and a few more
// launches are faster than long ones (10 M cells, per 200 iterations: no cap 513 ms / 13.6 k launches, cap 16: 420 ms /
// 15.1 k, cap 4: 617 ms / 29 k; with 1024-cell patches and cap 16: 358 ms / 13.0 k).
// (Also tried in round 2, measured slower, not kept: (a) granules published / read with agent-scope accesses so that a
// workgroup consumes what another one produced earlier in the SAME launch, plus bounded in-launch revisits: 6-27 launches per
// iteration instead of 60, but 670 ms; (b) narrower workgroups settling several tasks per thread (64 / 128 / 256 threads for
// 512 tasks: 840-1200 / 612 / 490 ms); (c) 256-cell patches: 669 ms; 2048-cell patches with two tasks per thread: 493 ms.)
// ---------------------------------------------------------------------------------------------------------
constexpr int WO_PATCH_THREADS = WO_PATCH;                 // one task per thread
constexpr int WO_PATCH_SPIN_LIMIT = 1 << 16;               // settle loop bound (never reached: every chain ends at a settled task)
constexpr int WO_PATCH_SPIN_CAP = 12;                      // default spinCap (sweep with the log-depth settle: 8 / 12 / 16 / 20 / 24 -> 307 / 298 / 303 / 312 / 321 ms)
#ifndef WO_PATCH_LATE_FROM
#define WO_PATCH_LATE_FROM (1 << 30)      // launch of a pass from which the polling cap is WO_PATCH_LATE_SPIN_CAP
#define WO_PATCH_LATE_SPIN_CAP 12
#endif
#ifndef WO_SETTLE_JUMP
#define WO_SETTLE_JUMP 1
#endif
#ifndef WO_PATCH_SLEEP
#define WO_PATCH_SLEEP 1
#endif
// dbg (diagnostic, may be null), per launch: [0] visits that found nothing runnable, [1] visits that ran, [2] runnable tasks,
// [3] completed tasks, [4] largest number of polling passes of a wave
__global__ __launch_bounds__(WO_PATCH_THREADS) void k_solve_patch(Fields F, int32_t L, int32_t launchTag, int32_t* patchPending, int32_t* totalPending,
                                                                   double K, double m, double dt, int32_t* dbg, int32_t spinCap) {
    __shared__ Granule s_out[2 * WO_PATCH];
    __shared__ int32_t s_st[WO_PATCH];                      // 0 unsettled, 1 certainly blocked, 2 candidate or done
    __shared__ int32_t s_cand[WO_PATCH];
#if WO_SETTLE_JUMP
    __shared__ int32_t s_wait[WO_PATCH];                    // settle: the local task whose state this one is waiting for
#endif
    __shared__ int32_t s_ncand, s_pend, s_done, s_fin;
    const int p = blockIdx.x, tid = threadIdx.x;
    if (patchPending[p] == 0) return;                       // block-uniform
    const Granule* G = reinterpret_cast<const Granule*>(F.out);
    const int32_t s0 = p * WO_PATCH;
    const int32_t base = 2 * s0;
    auto is_local = [&](int32_t g) { return (uint32_t)(g - base) < (uint32_t)(2 * WO_PATCH); };
    auto ext_ready = [&](int32_t g, double& v) {
        const Granule q = G[g];
        if (q.tag == 0 || q.tag >= launchTag) return false;
        v = q.v; return true;
    };
    if (tid == 0) { s_ncand = 0; s_pend = 0; s_done = 0; s_fin = 0; }
    // ---- every task: granules to LDS, remembered blocker settled
    {
        const int32_t sm = s0 + tid;
        SolveOut mine; mine.self.v = 0; mine.self.tag = 1; mine.dep.v = 0; mine.dep.tag = 1;      // beyond L: counts as done
        int32_t b = -1;
        if (sm < L) { mine = F.out[sm]; b = F.blk[sm]; }
        const bool pending = mine.self.tag == 0;
        int32_t st = 2, waitOn = -1;
        if (pending && b >= 0) {
            if (is_local(b)) { st = 0; waitOn = (b - base) >> 1; }
            else { double unused; st = ext_ready(b, unused) ? 2 : 1; }
        }
        s_st[tid] = st;
#if WO_SETTLE_JUMP
        s_wait[tid] = waitOn;
#endif
        s_out[2 * tid] = mine.self; s_out[2 * tid + 1] = mine.dep;
        __syncthreads();
        volatile int32_t* vst = s_st;
#if WO_SETTLE_JUMP
        // a task's state is the state of the head of its chain of remembered blockers; a task that finds its blocker unsettled
        // takes over the blocker's own blocker (any ancestor will do), so a chain of d tasks settles in ~log2 d polls, not d
        volatile int32_t* vw = s_wait;
        for (int spin = 0; spin < WO_PATCH_SPIN_LIMIT && __any(st == 0); ++spin) {
            if (st == 0) {
                const int32_t w = vst[waitOn];
                if (w != 0) { st = w; vst[tid] = w; }
                else { const int32_t up = vw[waitOn]; if (up >= 0) { waitOn = up; vw[tid] = up; } }
            }
        }
#else
        for (int spin = 0; spin < WO_PATCH_SPIN_LIMIT && __any(st == 0); ++spin) {
            if (st == 0) { const int32_t w = vst[waitOn]; if (w != 0) { st = w; vst[tid] = w; } }
        }
#endif
        if (pending) {
            if (st == 2) s_cand[atomicAdd(&s_ncand, 1)] = tid;      // (one atomic per wave through a ballot: measured 1.5 % slower)
            else { s_out[2 * tid].tag = -1; s_out[2 * tid + 1].tag = -1; }
        }
        const unsigned long long pm = __ballot(pending);
        if ((tid & 63) == 0 && pm) atomicAdd(&s_pend, __popcll(pm));
    }
    __syncthreads();
    const int32_t ncand = s_ncand;
    if (ncand == 0) {                                       // nothing can move in this patch: same pending count as before
        if (tid == 0) { atomicAdd(totalPending, s_pend); if (dbg) atomicAdd(&dbg[0], 1); }
        return;
    }
    if ((tid & ~63) >= ncand) return;                       // this wave got no task
    const int32_t aliveWaves = (ncand + 63) >> 6;
    if (dbg && tid == 0) { atomicAdd(&dbg[1], 1); atomicAdd(&dbg[2], ncand); }
    // ---- runnable tasks: record, external predecessors
    const unsigned long long BLOCKED = 0xffffffff00000000ull;              // tag -1
    volatile unsigned long long* vs = reinterpret_cast<volatile unsigned long long*>(s_out);
    SolveTask T;
    double er = 0, et = 0, et2 = 0;
    SolvePrepared pre;
    bool unresolved = tid < ncand;
    int32_t t = 0, s = 0, ran = 0;
    if (unresolved) {
        t = s_cand[tid]; s = s0 + t;
        T = F.task[s];
        er = T.e0r; et = T.e0t; et2 = T.e0t2;
        int32_t fail = -1;
        if (T.predSelf >= 0 && !is_local(T.predSelf) && !ext_ready(T.predSelf, er)) fail = T.predSelf;
        if (T.predT >= 0 && !is_local(T.predT) && !ext_ready(T.predT, et)) fail = T.predT;
        if (T.predT2 >= 0 && !is_local(T.predT2) && !ext_ready(T.predT2, et2)) fail = T.predT2;
        if (fail >= 0) { F.blk[s] = fail; unresolved = false; vs[2 * t] = BLOCKED; vs[2 * t + 1] = BLOCKED; }
        else pre = solve_prepare(T, K, m, dt);
    }
    // ---- run: poll the patch-local predecessors
    auto pack = [](Granule g) { return (unsigned long long)__float_as_uint(g.v) | ((unsigned long long)(uint32_t)g.tag << 32); };
    int spins = 0;
    for (int spin = 0; spin < spinCap && __any(unresolved); ++spin) {
        if (spin && WO_PATCH_SLEEP) __builtin_amdgcn_s_sleep(WO_PATCH_SLEEP);
        ++spins;
        if (!unresolved) continue;
        int32_t open = 0, fail = -1;
        double a = er, b = et, c = et2;
        auto rd = [&](int32_t g, double& v) {
            if (g < 0 || !is_local(g)) return;
            const unsigned long long w = vs[g - base];
            const int32_t tg = (int32_t)(w >> 32);
            if (tg < 0) fail = g;
            else if (tg == 0) open = 1;
            else v = __uint_as_float((uint32_t)w);
        };
        rd(T.predSelf, a); rd(T.predT, b); rd(T.predT2, c);
        if (fail >= 0) {
            vs[2 * t] = BLOCKED; vs[2 * t + 1] = BLOCKED;
            F.blk[s] = fail;
            unresolved = false;
        } else if (!open) {
            const SolveOut o = solve_apply(T, pre, a, b, c, launchTag);
            vs[2 * t] = pack(o.self); vs[2 * t + 1] = pack(o.dep);
            F.out[s] = o;
            unresolved = false; ran = 1;
        }
    }
    // ---- the last wave to finish publishes the patch's pending count
    const unsigned long long rm = __ballot(ran != 0);
    if (dbg && (tid & 63) == 0) atomicMax(&dbg[4], spins);
    if ((tid & 63) == 0) {
        if (rm) atomicAdd(&s_done, __popcll(rm));
        __threadfence_block();
        if (atomicAdd(&s_fin, 1) == aliveWaves - 1) {
            const int32_t done = atomicAdd(&s_done, 0);
            const int32_t left = s_pend - done;
            patchPending[p] = left;
            if (left) atomicAdd(totalPending, left);
            if (dbg) atomicAdd(&dbg[3], done);
        }
    }
}

// blocker hints of the tasks a basin launch left pending (solve_setup's choice: any unresolved predecessor will do)
__global__ __launch_bounds__(WO_BLOCK) void k_solve_blk_init(Fields F, int32_t L) {
    WO_GRID_STRIDE(s, L) {
        if (F.out[s].self.tag != 0) continue;
        const SolveTask T = F.task[s];
        F.blk[s] = T.predT >= 0 ? T.predT : (T.predSelf >= 0 ? T.predSelf : T.predT2);
    }
}
__global__ __launch_bounds__(WO_BLOCK) void k_slot_scatter(const int32_t* patchOrder, int32_t* slotOf, int32_t L) {
    WO_GRID_STRIDE(s, L) slotOf[patchOrder[s]] = s;
}


// `masked` (may be null): the thermal step's masked elevation of the new field, written in the same pass
__global__ __launch_bounds__(WO_BLOCK) void k_solve_final(Fields F, float* out, float* masked) {
    WO_XCD_LAND(i, r) {
        const float v = solve_final_cell(F, r);
        out[r] = v;
        if (masked) masked[r] = v;
    }
}
__global__ __launch_bounds__(WO_BLOCK) void k_fill_i32(int32_t* a, int32_t v, int32_t n) { WO_GRID_STRIDE(i, n) a[i] = v; }

// ---------------------------------------------------------------- thermal -----------------------
__global__ __launch_bounds__(WO_BLOCK) void k_masked_elev(Fields F) { WO_XCD_LAND(i, r) F.me[r] = F.e[r]; }
__global__ __launch_bounds__(WO_BLOCK) void k_thermal_excess(Fields F, double talus) {
    __shared__ float s_tile[WO_TILE_WIN];
    const TileWindow M = stage_tile(F, F.me, s_tile, WO_XCD_LAND_BASE());
    WO_XCD_LAND(i, r) thermal_excess_cell_t(F, r, talus, M);
}
// meshes whose largest degree is <= 16 (jittered Fibonacci spheres: 10-11 at 10^4..10^6 cells, 13 at 10^7) keep the event
// lists of the rare rows of more than WO_ROW neighbours in private arrays; larger degrees use the LDS form.
// Occupancy: left alone the compiler promotes the arrays to registers (103 / 115 VGPRs, 4 waves per SIMD, no scratch); asked for
// >= 5 waves it keeps them in scratch (62 VGPRs, 272 B per lane, 8 waves per SIMD) — and since 99.9 % of the rows take
// thermal_apply_row, which never touches them, the launch is 27 % SHORTER that way (10 M cells: 28.0 -> 20.6 ms per step, round 4;
// the kernel waits on its gathers 37 % of the time and issues f64 arithmetic the rest: twice the waves hide more of both).
#ifndef WO_THERMAL_WAVES
#define WO_THERMAL_WAVES 6
#endif
template <int MAXIN>
__global__ __launch_bounds__(WO_BLOCK, WO_THERMAL_WAVES) void k_thermal_apply_reg(Fields F, float* out, double talus, double kThermal) {
    __shared__ float s_tile[WO_TILE_WIN];
    const TileWindow M = stage_tile(F, F.me, s_tile, WO_XCD_LAND_BASE());
    WO_XCD_LAND(i, r) {
        double inShare[MAXIN], outShare[MAXIN]; int32_t inRank[MAXIN];
        out[r] = thermal_apply_cell_t(F, r, talus, kThermal, inShare, inRank, 1, outShare, M);
    }
}
__global__ __launch_bounds__(WO_BLOCK) void k_thermal_apply(Fields F, float* out, double talus, double kThermal, int32_t maxDeg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [maxDeg][256] doubles, then [maxDeg][256] ints
    double* inShare = reinterpret_cast<double*>(smem) + threadIdx.x;
    int32_t* inRank = reinterpret_cast<int32_t*>(smem + (size_t)maxDeg * WO_BLOCK * sizeof(double)) + threadIdx.x;
    WO_XCD_LAND(i, r) out[r] = thermal_apply_cell(F, r, talus, kThermal, inShare, inRank, WO_BLOCK);
}

// ---------------------------------------------------------------- glacial -----------------------
__global__ __launch_bounds__(WO_BLOCK) void k_glac_index(Fields F, double strength) { WO_XCD_CELLS(r, F.N) F.glac[r] = glac_index_cell(F, r, strength); }
__global__ __launch_bounds__(WO_BLOCK) void k_ice_receivers(Fields F) {
    WO_XCD_CELLS(r, F.N) { ice_receiver_cell(F, r); F.blocker[r] = 0; if (F.ocean[r]) { F.iceFlow[r] = 0.0f; F.iceUp[r] = 0; } }      // blocker: k_ice_climb's arrival counts (carve_setup_cell resets it)
}
// ---- in-launch hand-offs between workgroups (MI355X: a CU's L1 is never refreshed by other CUs' stores and the eight XCD L2s are not
// coherent with each other): every word that crosses between threads inside one launch is written AND read at agent scope
// (global_store / global_load ... sc1: write-through, L1 bypassed), and a flag or an arrival count follows its payload only after
// the payload's stores have been acknowledged (s_waitcnt vmcnt(0), as inline asm so that no pass can drop it). ----
__device__ inline float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline int32_t ld_agent(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Ice accumulation (js/terrain-post.js:495-503) as ONE launch.  A cell's ice flow is glacIdx plus its donors' flows added in
// landCells order, each add rounded to f32 — the donors must all be final first, and the forest of ice receivers is ~80 cells
// deep: as synchronous rounds that was ~80 launches of 16 us per glacial step.  Here every cell without donors runs its task and
// hands over to its receiver through one returning atomic on the receiver's arrival count; the thread whose arrival completes a
// receiver (last donor in) runs the receiver's task and carries on upwards — no thread ever waits, so nothing depends on dispatch
// order or residency.  The sum is still taken over the donors in rank order by the one thread that runs the task: the same
// operations on the same values as ice_accumulate_task.  The receiver's donor list is static during the launch and is read while
// the cell's own store drains.
// rows of up to WO_ROW neighbours stay in registers (no indexed arrays: those would live in scratch memory, on the critical path)
struct IceRow { int32_t nb[WO_ROW]; int32_t rank[WO_ROW]; uint32_t donors; int nd; int deg; };
__device__ inline void ice_row_of(const Fields& F, int32_t t, IceRow& D) {
    int32_t b;
    D.deg = load_row(F, t, b, D.nb);
    D.donors = 0; D.nd = 0;
    if (D.deg > WO_ROW) {                                       // long row: only the count here, the task walks the row itself
        for (int32_t j = b; j < b + D.deg; ++j) if (F.iceTarget[F.adj[j]] == t) ++D.nd;
        return;
    }
    int32_t tg[WO_ROW];
#pragma unroll
    for (int k = 0; k < WO_ROW; ++k) { tg[k] = F.iceTarget[D.nb[k]]; D.rank[k] = F.rank[D.nb[k]]; }
#pragma unroll
    for (int k = 0; k < WO_ROW; ++k) if (k < D.deg && tg[k] == t) { D.donors |= 1u << k; ++D.nd; }
}
__device__ inline float ice_task_long_row(const Fields& F, int32_t t, int& up) {     // ice_accumulate_task's walk, heights of the donors at agent scope
    int32_t dn[WO_MAX_DEG]; int nd = 0;
    for (int32_t j = F.off[t]; j < F.off[t + 1]; ++j) { const int32_t n = F.adj[j]; if (F.iceTarget[n] == t) dn[nd++] = n; }
    float acc = F.glac[t];
    int32_t last = -1;
    up = 0;
    for (int k = 0; k < nd; ++k) {
        int pick = -1; int32_t pr = 0x7fffffff;
        for (int q = 0; q < nd; ++q) { const int32_t rk = F.rank[dn[q]]; if (rk > last && rk < pr) { pr = rk; pick = q; } }
        const float df = ld_agent(&F.iceFlow[dn[pick]]);
        if (df > 0) { acc = (float)((double)acc + (double)df); ++up; }
        last = pr;
    }
    return acc;
}
__global__ __launch_bounds__(WO_BLOCK) void k_ice_climb(Fields F, int32_t* arrived) {
    WO_XCD_LAND(i, r) {
        IceRow D;
        ice_row_of(F, r, D);
        if (D.nd > 0) continue;                                 // its last donor carries on with it
        int32_t cur = r;
        for (;;) {
            float acc;
            int up = 0;
            const int32_t tgt = F.iceTarget[cur];
            if (D.deg > WO_ROW) acc = ice_task_long_row(F, cur, up);
            else {
                float fl[WO_ROW];
#pragma unroll
                for (int k = 0; k < WO_ROW; ++k) fl[k] = ((D.donors >> k) & 1u) ? ld_agent(&F.iceFlow[D.nb[k]]) : 0.0f;
                acc = F.glac[cur];
                int32_t last = -1;
#pragma unroll
                for (int step = 0; step < WO_ROW; ++step) {     // donors in landCells order (ascending rank)
                    int32_t pr = 0x7fffffff; float df = 0.0f;
#pragma unroll
                    for (int q = 0; q < WO_ROW; ++q) if (((D.donors >> q) & 1u) && D.rank[q] > last && D.rank[q] < pr) { pr = D.rank[q]; df = fl[q]; }
                    if (pr == 0x7fffffff) break;
                    if (df > 0) { acc = (float)((double)acc + (double)df); ++up; }
                    last = pr;
                }
            }
            st_agent(&F.iceFlow[cur], acc);
            F.iceUp[cur] = (uint8_t)up;
            if (tgt < 0 || F.ocean[tgt]) break;                 // ocean cells can be targets, they are not tasks
            ice_row_of(F, tgt, D);                              // static during the launch: read while the store drains
            drain_stores();
            if (atomicAdd(&arrived[tgt], 1) + 1 != D.nd) break; // other donors of tgt are still out
            cur = tgt;
        }
    }
}

__global__ __launch_bounds__(WO_BLOCK) void k_carve_setup_cells(Fields F) {          // the list comes from select_active_by_rank
    WO_XCD_CELLS(r, F.N) carve_setup_cell(F, r);
}
__global__ __launch_bounds__(WO_BLOCK) void k_carve_deps(Fields F, const int32_t* list, const int32_t* count, int32_t* carveSlot) {
    const int32_t n = *count;
    WO_GRID_STRIDE(i, n) { const int32_t r = list[i]; carveSlot[r] = i; carve_deps_cell(F, r, i); }
}
// per-task records of the static rounds (after k_carve_deps) and their slot-indexed round tags
__global__ __launch_bounds__(WO_BLOCK) void k_carve_records(Fields F, const int32_t* list, const int32_t* count, CarveRec* recs, int32_t* slotDone, double gCarve, double gConv, double gStrength,
                                                             int32_t withDeps, int32_t resetDone) {
    const int32_t n = *count;
    WO_GRID_STRIDE(i, n) { carve_record_cell(F, list[i], i, recs, gCarve, gConv, gStrength, withDeps); if (resetDone) slotDone[i] = WO_NOT_DONE; }
}
// Carve rounds over the STATIC activation list: one thread per active task in every round, no pending lists, no counters on the
// chain.  A finished task leaves after one load; an open one issues all its loads at once (carve_task_eager) and runs when its
// dependencies finished in earlier launches.  done: tasks finished so far (one atomic per wave that finished any), read back
// by the driver every few rounds.
// (Also measured in round 3 and dropped, profiles/r03x_*: launches of several SUB-ROUNDS — dependencies inside the workgroup's 256
// tasks satisfied through LDS flags between barriers.  3 248 launches became 1 344 with 8 sub-rounds, but a launch then lasts 35 us:
// glacial stage 58-81 ms against 60; the chains of the carve DAG leave a workgroup's patch after two or three levels.)
__global__ __launch_bounds__(WO_BLOCK) void k_carve_round_static(Fields F, const CarveRec* __restrict__ recs, int32_t* slotDone, const int32_t* __restrict__ count, int32_t round,
                                                                  double gCarve, double gConv, double gStrength, int32_t* done) {
    const int32_t n = *count;
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool ran = false;
    if (i < n && slotDone[i] == WO_NOT_DONE) {
        const CarveRec R = recs[i];
        ran = carve_task_rec(F, R, i, round, gCarve, gConv, gStrength);
        if (ran) slotDone[i] = round;
    }
    const unsigned long long m = __ballot(ran);
    if (m && (threadIdx.x & 63) == 0) atomicAdd(done, __popcll(m));
}
// ---- the carve turns as ONE launch, the heights as self-validating granules ------------------------------------------------------
// A carve turn (js/terrain-post.js:506-526) may run once every lower-ranked active cell within two hops has had its turn; the DAG of
// those waits is ~330 turns deep on a 10 M-cell planet, and as synchronous rounds every level cost a launch (11 us each, 3.7 ms
// per glacial step for ~3 us of actual work per level).  Here every active task has its own thread and takes its turn as soon as what
// it waits for is there: a level costs a hand-off between CUs (~2 us).
//  * The activation list is in RANK order (select_active_by_rank), so a task's dependencies all sit at lower positions; thread g
//    takes positions g, g + G, g + 2G, ... in turn with G = all the launch's threads.  The lowest unfinished position then always
//    belongs to a thread that has nothing earlier left, so some task can always run provided the G threads are resident: the grid
//    is sized by the occupancy query (minus a margin, planet.hip), never by the task count.
//  * A wave is a set of 64 independent little state machines inside ONE wave-uniform loop: a lane that finds its task ready takes
//    the turn inside the loop body, so no lane ever sits at a reconvergence point waiting for another lane's poll (whose task may
//    depend on it).
//  * Every spin is bounded: a lane that has waited `budget` ticks of the 100 MHz wall clock gives up and leaves its task to the
//    synchronous rounds (k_carve_round_static), which take over from whatever state the launch left.
//  * WHAT a task waits for is the heights themselves (round 3's first form polled per-task done words and paid four memory round trips per
//    level: the finished task's stores acknowledged, the word seen, the heights loaded, the new ones stored; removed in round 6).  During the
//    carve every cell's height lives in an 8-byte granule {height, tag}, written by one store, where tag = 1 + rank of the task that wrote it
//    (0: not written yet in this glacial step).  Which task writes a cell x last before task T's turn is static — the highest-ranked active
//    cell below T among x and x's neighbours (the turns that touch x are the active cells of that set, and they touch x in rank order) —
//    so T knows, for each of the <= 13 cells it reads, the tag it must see (k_carve_expect).  T polls those granules; when all carry their
//    expected tags it already HAS the values, takes its turn and stores the new granules: a level is one store seen by one poll.
//  * Nothing can overwrite a granule before T has read it: a task U that writes x after T is an active cell of the same set with
//    a higher rank, so U itself waits for x to carry T's tag (or a later one) — and T writes x only after reading it.
//  * An 8-byte naturally aligned store is seen whole (MI355X_MICROARCH.md: hand-off granules), so no ordering is needed at all.
//  * A task that gives up leaves its slot open, the heights go back into the field (k_carve_unpack) and the synchronous rounds finish the step.
// The arithmetic is carve_task_rec's / carve_task's, statement for statement.
__device__ inline unsigned long long granule_pack(float v, int32_t tag) { return ((unsigned long long)(uint32_t)tag << 32) | (unsigned long long)__float_as_uint(v); }
__device__ inline float granule_value(unsigned long long g) { return __uint_as_float((uint32_t)g); }
__device__ inline int32_t granule_tag(unsigned long long g) { return (int32_t)(g >> 32); }
__device__ inline unsigned long long ld_agent(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// tag of the last turn that writes x before the turn of the task with rank myRank (0: none)
__device__ inline int32_t carve_expected_tag(const Fields& F, int32_t x, int32_t myRank) {
    int32_t m = -1;
    { const int32_t a = F.arank[x]; if (a < myRank) m = a; }
    for (int32_t j = F.off[x]; j < F.off[x + 1]; ++j) { const int32_t a = F.arank[F.adj[j]]; if (a < myRank && a > m) m = a; }
    return m + 1;
}
__global__ __launch_bounds__(WO_BLOCK) void k_carve_expect(Fields F, const CarveRec* __restrict__ recs, const int32_t* __restrict__ count, CarveExpect* ex) {
    const int32_t n = *count;
    WO_GRID_STRIDE(i, n) {
        const int32_t r = recs[i].r, deg = recs[i].deg, myRank = F.arank[r];
        CarveExpect X;
        X.myTag = myRank + 1; X.pad_[0] = X.pad_[1] = 0;
        for (int k = 0; k <= WO_EAGER_ROW; ++k) X.tag[k] = -1;
        if (deg <= WO_EAGER_ROW) {
            const int32_t b = F.off[r];
            for (int k = 0; k < deg; ++k) { const int32_t x = F.adj[b + k]; if (!F.ocean[x]) X.tag[k] = carve_expected_tag(F, x, myRank); }
            X.tag[WO_EAGER_ROW] = carve_expected_tag(F, r, myRank);
        }
        ex[i] = X;
    }
}
__global__ __launch_bounds__(WO_BLOCK) void k_carve_pack(const float* __restrict__ e, unsigned long long* __restrict__ G, int32_t n) { WO_GRID_STRIDE(i, n) G[i] = granule_pack(e[i], 0); }
__global__ __launch_bounds__(WO_BLOCK) void k_carve_unpack(const unsigned long long* __restrict__ G, float* __restrict__ e, int32_t n) { WO_GRID_STRIDE(i, n) e[i] = granule_value(G[i]); }
// rows longer than WO_EAGER_ROW: one non-blocking sweep (expected tags worked out on the spot); true: the turn was taken
__device__ inline bool carve_granule_turn_long_row(const Fields& F, unsigned long long* G, int32_t r, double deepening, double bonus, int32_t up) {
    const int32_t myRank = F.arank[r], myTag = myRank + 1;
    if (granule_tag(ld_agent(&G[r])) != carve_expected_tag(F, r, myRank)) return false;
    for (int32_t j = F.off[r]; j < F.off[r + 1]; ++j) {
        const int32_t nb = F.adj[j];
        if (!F.ocean[nb] && granule_tag(ld_agent(&G[nb])) != carve_expected_tag(F, nb, myRank)) return false;
    }
    float er = (float)((double)granule_value(ld_agent(&G[r])) - deepening);
    for (int32_t j = F.off[r]; j < F.off[r + 1]; ++j) {
        const int32_t nb = F.adj[j];
        if (F.ocean[nb]) continue;
        const double d = nd_or_eps(F.dist[j]);
        const float en = granule_value(ld_agent(&G[nb]));
        const double slope = fabs((double)er - (double)en) / d;
        double f = 1 - slope;
        if (!(f > 0)) f = (f != f) ? f : 0;
        st_agent(&G[nb], granule_pack((float)((double)en - deepening * 0.4 * f), myTag));
    }
    if (up >= 2) er = (float)((double)er - bonus);
    st_agent(&G[r], granule_pack(er, myTag));
    return true;
}
__global__ __launch_bounds__(WO_BLOCK, 4) void k_carve_granules(Fields F, const CarveRec* __restrict__ recs, const CarveExpect* __restrict__ expect, unsigned long long* G,
                                                              int32_t* slotDone, const int32_t* __restrict__ count, int32_t* done, long long budget) {
    const int32_t n = *count;
    const int32_t stride = (int32_t)(gridDim.x * blockDim.x);
    int32_t i = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
    const long long t0 = wall_clock64();
    int32_t finished = 0;
    bool active = i < n;
    // the task's record, the part a turn needs (the dependency list is the rounds')
    int32_t r = 0, deg = 0, up = 0, nbs[WO_EAGER_ROW]; float dist[WO_EAGER_ROW]; double deepening = 0, bonus = 0;
    CarveExpect X;
    float vals[WO_EAGER_ROW + 1];
    uint32_t pend = 0;
    auto fetch = [&]() {
        const CarveRec& R = recs[i];
        r = R.r; deg = R.deg; up = R.up; deepening = R.deepening; bonus = R.bonus;
#pragma unroll
        for (int k = 0; k < WO_EAGER_ROW; ++k) { nbs[k] = R.nbs[k]; dist[k] = R.dist[k]; }
        X = expect[i];
        pend = 0;
#pragma unroll
        for (int k = 0; k <= WO_EAGER_ROW; ++k) if (X.tag[k] >= 0) pend |= 1u << k;
    };
    if (active) fetch();
    for (;;) {
        if (active) {
            bool ran = false;
            if (deg > WO_EAGER_ROW) ran = carve_granule_turn_long_row(F, G, r, deepening, bonus, up);
            else {
                unsigned long long g[WO_EAGER_ROW + 1];
#pragma unroll
                for (int k = 0; k <= WO_EAGER_ROW; ++k) g[k] = ((pend >> k) & 1u) ? ld_agent(&G[k < WO_EAGER_ROW ? nbs[k] : r]) : 0ull;
#pragma unroll
                for (int k = 0; k <= WO_EAGER_ROW; ++k)
                    if (((pend >> k) & 1u) && granule_tag(g[k]) == X.tag[k]) { vals[k] = granule_value(g[k]); pend &= ~(1u << k); }
                if (pend == 0) {
                    float er = (float)((double)vals[WO_EAGER_ROW] - deepening);
#pragma unroll
                    for (int k = 0; k < WO_EAGER_ROW; ++k) {
                        if (X.tag[k] < 0) continue;                 // past the row's end, or an ocean neighbour
                        const double d = nd_or_eps(dist[k]);
                        const double slope = fabs((double)er - (double)vals[k]) / d;
                        double f = 1 - slope;
                        if (!(f > 0)) f = (f != f) ? f : 0;
                        st_agent(&G[nbs[k]], granule_pack((float)((double)vals[k] - deepening * 0.4 * f), X.myTag));
                    }
                    if (up >= 2) er = (float)((double)er - bonus);
                    st_agent(&G[r], granule_pack(er, X.myTag));
                    ran = true;
                }
            }
            if (ran) {
                F.doneAt[r] = 1;                                    // for the rounds, should they have to finish the step (read after the launch)
                slotDone[i] = 1;
                ++finished;
                i += stride;
                active = i < n;
                if (active) fetch();
            } else if (wall_clock64() - t0 > budget) {
                active = false;                                     // left to the synchronous rounds
            }
        }
        if (!__any(active)) break;
        __builtin_amdgcn_s_sleep(2);
    }
    for (int o = 32; o > 0; o >>= 1) finished += __shfl_down(finished, o);
    if ((threadIdx.x & 63) == 0 && finished) atomicAdd(done, finished);
}
__global__ __launch_bounds__(WO_BLOCK) void k_moraine_fjord(Fields F, double gDep, double gFjord) {
    WO_XCD_CELLS(r, F.N) moraine_fjord_cell(F, r, gDep, gFjord);
}

// ---------------------------------------------------------------------------------------------------------------------
// RELAXED MODE (WO_RELAXED=full; SURVEY 7.3's "roofline mode") — NOT the reference's semantics, never used for parity or for `value`.
// The order-defined parts of an iteration are replaced by order-free ones: landCells is sorted once per flood (the passes keep comparing
// the stale ranks pairwise), the implicit solve becomes an affine recurrence composed by pointer jumping with the deposition applied
// afterwards from the old slopes, the glacial carve reads a snapshot (Jacobi).  bench.py reports its time and its distance from the
// exact field as `relaxed_mode`.
// ---------------------------------------------------------------------------------------------------------------------
// js/terrain-post.js:614-627 without the order: h' = (h + f * hr) / (1 + f) with hr = max(h'(t), 0) is affine in h'(t) along a forward edge that
// descends; an ocean receiver, a late edge (the receiver's turn comes later: it still has its old height) and a cell without a receiver are
// final at once; a forward edge that does not descend gives h' = max(.., hr) = hr
__global__ __launch_bounds__(WO_BLOCK) void k_affine_init(Fields F, Affine* X) {
    WO_XCD_LAND(i, r) {
        const TargetRank trr = F.tr[r];
        const int32_t t = trr.target;
        const double h = F.e[r];
        Affine A; A.pad = 0; A.j = -1; A.b = 0.0f;
        if (t < 0) A.a = (float)h;
        else {
            const double f = solve_factor_of(F.flow[r], F.cellDist[r], F.solveK, F.solveM, F.solveDt);
            const TargetRank trt = F.tr[t];
            const double et = F.e[t];
            const bool forward = trt.rank >= 0 && trr.rank < trt.rank;            // (ocean cells carry rank -1)
            if (!forward) {
                const double hr = et > 0 ? et : 0;
                double hn = (h + f * hr) / (1 + f);
                hn = hn < hr ? hr : hn; hn = hn < 0 ? 0 : hn;
                A.a = (float)hn;
            } else if (et >= h) { A.a = 0.0f; A.b = 1.0f; A.j = t; }
            else { A.a = (float)(h / (1 + f)); A.b = (float)(f / (1 + f)); A.j = t; }
        }
        X[r] = A;
    }
}
__global__ __launch_bounds__(WO_BLOCK) void k_affine_jump(Fields F, const Affine* __restrict__ in, Affine* __restrict__ out) {
    WO_XCD_LAND(i, r) {
        Affine A = in[r];
        if (A.j >= 0) { const Affine B = in[A.j]; A.a = (float)((double)A.a + (double)A.b * (double)B.a); A.b = (float)((double)A.b * (double)B.b); A.j = B.j; }
        out[r] = A;
    }
}
// the new height of every land cell: its solved height plus what its donors deposit on it (js/terrain-post.js:628-640, deposits taken from the OLD slope of the cell
// and applied in one sum; the cap of a deposit at the donor's new height is dropped with the order)
__global__ __launch_bounds__(WO_BLOCK) void k_affine_apply(Fields F, const Affine* __restrict__ X, float* __restrict__ out) {
    WO_XCD_LAND(i, c) {
        double hc = X[c].a; if (hc < 0) hc = 0;
        const TargetRank trc = F.tr[c];
        double slope = 0;
        if (trc.target >= 0) { const float cd = F.cellDist[c]; if (cd > 0) slope = fabs((double)F.e[c] - (double)F.e[trc.target]) / (double)cd; }
        const double frac = 0.5 / (1 + slope * 50);
        double dep = 0;
        for (int32_t jn = F.off[c]; jn < F.off[c + 1]; ++jn) {
            const int32_t d = F.adj[jn];
            if (F.tr[d].target != c) continue;                                   // (ocean cells carry target -1)
            double hd = X[d].a; if (hd < 0) hd = 0;
            const double eroded = (double)F.e[d] - hd;
            if (eroded > 0) dep += eroded * frac;
        }
        const float v = (float)(hc + dep);
        out[c] = v; F.me[c] = v;
    }
}
// the glacial carve from a snapshot (js/terrain-post.js:506-526 as a Jacobi pass): every active cell's own deepening and convergence bonus, and the widening every
// active neighbour would apply to this cell, all from the heights before the pass
__global__ __launch_bounds__(WO_BLOCK) void k_carve_jacobi(Fields F, const float* __restrict__ in, float* __restrict__ out, double gCarveRate, double gConvergenceBonus, double glacialStrength) {
    WO_XCD_LAND(i, c) {
        const double h = in[c];
        double e = h;
        const double flc = F.iceFlow[c];
        if (flc > 0.1) { e -= gCarveRate * pow(flc, 0.6) * glacialStrength; if (F.iceUp[c] >= 2) e -= gConvergenceBonus * pow(flc, 0.4); }
        for (int32_t jn = F.off[c]; jn < F.off[c + 1]; ++jn) {
            const int32_t nb = F.adj[jn];
            if (F.ocean[nb]) continue;
            const double fl = F.iceFlow[nb];
            if (!(fl > 0.1)) continue;
            const double deep = gCarveRate * pow(fl, 0.6) * glacialStrength;
            const double slope = fabs(((double)in[nb] - deep) - h) / nd_or_eps(F.dist[jn]);
            const double w = 1 - slope;
            if (w > 0) e -= deep * 0.4 * w;
        }
        out[c] = (float)e;
    }
}

// ---- patch-major mirror of the mesh for erodeComposite (planet.hip: Mirror).  perm: mirror id -> cell id, inv: the inverse ----
__global__ __launch_bounds__(WO_BLOCK) void k_mirror_gather_f32(const float* src, const int32_t* perm, float* dst, int32_t n) { WO_GRID_STRIDE(i, n) dst[i] = src[perm[i]]; }
__global__ __launch_bounds__(WO_BLOCK) void k_mirror_scatter_f32(const float* src, const int32_t* perm, float* dst, int32_t n) { WO_GRID_STRIDE(i, n) dst[perm[i]] = src[i]; }
__global__ __launch_bounds__(WO_BLOCK) void k_mirror_gather_u8(const uint8_t* src, const int32_t* perm, uint8_t* dst, int32_t n) { WO_GRID_STRIDE(i, n) dst[i] = src[perm[i]]; }
__global__ __launch_bounds__(WO_BLOCK) void k_mirror_map_i32(const int32_t* in, const int32_t* map, int32_t* out, int32_t n) { WO_GRID_STRIDE(i, n) out[i] = map[in[i]]; }
__global__ __launch_bounds__(WO_BLOCK) void k_mirror_invert(const int32_t* perm, int32_t* inv, int32_t n) { WO_GRID_STRIDE(i, n) inv[perm[i]] = i; }
// rows keep their order (the reference's adjacency order is part of its semantics); only the ids are renamed
__global__ __launch_bounds__(WO_BLOCK) void k_mirror_rows(const int32_t* off, const int32_t* adj, const float* dist, const float* xyz, const int32_t* perm,
                                                           const int32_t* inv, const int32_t* moff, int32_t* madj, float* mdist, float* mxyz, int32_t n) {
    WO_GRID_STRIDE(i, n) {
        const int32_t r = perm[i];
        const int32_t b = off[r], deg = off[r + 1] - b, mb = moff[i];
        for (int32_t k = 0; k < deg; ++k) { madj[mb + k] = inv[adj[b + k]]; mdist[mb + k] = dist[b + k]; }
        mxyz[3 * i] = xyz[3 * r]; mxyz[3 * i + 1] = xyz[3 * r + 1]; mxyz[3 * i + 2] = xyz[3 * r + 2];
    }
}

// {serial, value} of a device counter into a host-mapped word: the host polls the word instead of paying a copy, a stream
// synchronisation and its wake-up for one integer (planet.hip: publish_and_wait)
__global__ void k_publish_count(const int32_t* __restrict__ src, unsigned long long* hostWord, uint32_t serial) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        __hip_atomic_store(hostWord, ((unsigned long long)serial << 32) | (unsigned long long)(uint32_t)*src, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_set_counters(int32_t* c, int32_t v0, int32_t v1, int32_t v2, int32_t v3) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { c[0] = v0; c[1] = v1; c[2] = v2; c[3] = v3; }
}

}  // namespace wo
