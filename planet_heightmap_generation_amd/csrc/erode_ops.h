// Per-cell / per-task bodies of the terrain-post kernels (reference: js/terrain-post.js).
//
// Every function here is the body of one HIP kernel thread.  They are written against raw pointers in
// a `Fields` struct so that the same code is (a) launched by the gfx950 kernels in kernels_*.hip and
// (b) driven one thread at a time by the test-only emulator (tests/emu), which is how the parallel
// re-formulations below were validated against the serial oracle in a container without a GPU.
//
// Numeric contract: arithmetic in double, stores narrow to float; -ffp-contract=off.
//
// How the order-defined reference loops are made parallel *without changing any result bit*:
//
//  * sort      landCells is kept as a device array; a stable LSD radix sort of (descending key, cell) run on
//              the previous order IS V8's stable sort (ties keep their previous order).  rank[c] = position.
//  * flow      flow values are integers (exact in f32), so only the *set* of contributions matters.
//              F(c) = 1 + sum F(d) over donors ranked before c is a forest subtree size, computed by pointer
//              doubling (A_{k+1}[c] = A_k[c] + sum_{jump_k[d]=c} A_k[d]); donors ranked after their receiver
//              ("late" edges, ~1 %) are added once, un-forwarded, exactly as the serial pass leaves them.
//  * solve     every turn of the ascending-order implicit solve is a task whose three inputs (own height,
//              receiver height, receiver's receiver height) are each "the value left by the latest earlier
//              event on that cell".  Events are single-assignment (selfOut / tOut per task), so the pass is
//              a pure dataflow DAG executed in synchronous rounds: a task runs in round k when all (<=3) of
//              its predecessor tasks finished in rounds < k.
//  * thermal   delta[] is an f32 accumulator whose rounding depends on visiting order; each cell replays, in
//              rank order, exactly the additions the serial loop would have applied to it.
//  * glacial   ice accumulation = per-receiver f32 sums over donors in rank order (dataflow rounds); the
//              in-place carve runs a task when no unfinished active cell within two hops has a lower rank,
//              which makes concurrently running tasks touch disjoint cells and preserves every
//              read-after-write / write-after-read order of the serial loop.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "noise.h"

namespace wo {

constexpr int WO_MAX_DEG = 24;          // planets with a larger vertex degree are rejected at creation
constexpr int32_t WO_NOT_DONE = 0x7fffffff;

// ---- solve dataflow records (one 48-byte record in, one 16-byte record out per task) ----
// Everything a turn of the implicit solve needs that does not depend on other turns is gathered once, in
// index order, by solve_setup; a round then costs: task record -> <=3 predecessor granules.
// Records are stored at the task's *store index*: the cell id, or — for the patch-local solve — the cell's slot in
// the Morton-ordered patch list, so that a patch's records are contiguous (Fields::slotOf, store_index()).
#ifndef WO_TASK_ALIGN
#define WO_TASK_ALIGN 16                // 64 (whole blocks per scattered write) measured no faster at 10 M cells
#endif
struct alignas(WO_TASK_ALIGN) SolveTask {
    int32_t predSelf, predT, predT2;   // granule index 2*store + (0: that task's own turn, 1: its deposit on its receiver), -1: none
    uint32_t flags;                    // bit0: target is ocean, bit1: t2 is ocean, bit2: has a target, bit3: has a t2
    float e0r, e0t, e0t2;              // heights before the pass (used where there is no predecessor event)
    float cellDistT;                   // cellDist[target]
    double factor;                     // K * flow[r]^m * dt / cellDist[r] (js/terrain-post.js:621-622): depends on nothing the pass changes, so
                                       // solve_setup works it out once per task instead of every visit of the task's patch before it may poll
    int32_t pad_[2];                   // with Fields::solveFinals: {the task's cell r, its land receiver t or -1}; flags bit4: r's own turn is the LAST event on r, bit5: r's
                                       // deposit is the last event on t — the solve launch then writes those heights itself and the final pass is not run
};
static_assert(sizeof(SolveTask) == 48, "one 48-byte record per task");
#ifndef WO_PATCH_CELLS
#define WO_PATCH_CELLS 1024
#endif
constexpr int WO_PATCH = WO_PATCH_CELLS;   // land cells (= solve tasks) per spatial patch / workgroup
// {value, round tag}: tag 0 = not produced yet.  Written once per pass, consumed only by later rounds.
struct alignas(8) Granule { float v; int32_t tag; };
struct alignas(16) SolveOut { Granule self, dep; };

// {drainTarget, rank} of a cell in one 8-byte word: the solve's setup/final passes ask both of every neighbour
struct alignas(8) TargetRank { int32_t target; int32_t rank; };
// Events on one location x during the solve pass, in processing order (descending rank): x's own turn and the deposits
// of its donors.  Built once per pass by the flow accumulation's last kernel (which has x's row and the {target, rank} of
// its neighbours in registers anyway); solve_setup then finds a task's <=3 predecessor events in the 32-byte lists of
// r, its receiver and the receiver's receiver instead of re-reading three neighbour rows (24 gathers), and solve_final
// reads the latest event off the end of the list.  Locations with more than WO_EVENTS events (own turn + >3 donors:
// ~1 % of the cells) carry the overflow mark and take the row scans.
#ifndef WO_EVENTS_N
#define WO_EVENTS_N 6
#endif
constexpr int WO_EVENTS = WO_EVENTS_N;
struct alignas(16) EventList { int32_t cell[WO_EVENTS]; int32_t rank[WO_EVENTS]; };     // unused slots: rank -1; overflow: rank[0] == -2
constexpr int WO_CARVE_DEPS = 24;       // dependency slots per active carve task (a task with more takes the scanning form, and a round lasts as long as its slowest task)
struct alignas(16) Affine { float a, b; int32_t j; int32_t pad; };       // relaxed mode (kernels_impl.h): h'(r) = a + b * h'(j); j < 0: h'(r) = a
struct Fields {
    int32_t N;                 // numRegions
    int32_t xcdTile;           // blocks per XCD tile for index-order kernels (device.h: xcd_tile)
    int32_t tileLds;           // 1: the index-order passes over land stage their tile's neighbour window in LDS (kernels_impl.h: stage_tile; WO_TILE_LDS=1)
    const int32_t* off;        // adjOffset [N+1]
    const int32_t* adj;        // adjList   [E]
    const float* dist;         // neighborDist [E]
    const float* xyz;          // r_xyz [3N]
    const uint8_t* ocean;      // r_isOcean [N]
    const uint8_t* coast;      // land cell with >=1 ocean neighbour [N]
    float* e;                  // r_elevation [N] (current)
    float* e2;                 // second elevation buffer (Jacobi ping-pong)
    float* me;                 // thermal: elevation with ocean cells at +inf (one gather per neighbour instead of e + isOcean)
    // erodeComposite scratch
    int32_t L;                 // land cells
    int32_t* land;             // landCells in current order [L]
    const int32_t* landIdx;    // land cells in ascending id (the order of the index-order passes over land) [L]
    int32_t xcdTileL;          // blocks per XCD tile for passes over the land list
    int32_t* rank;             // rank[c] = index of c in land, -1 for ocean [N]
    int32_t* target;           // drainTarget [N]
    TargetRank* tr;            // {target, rank} written by the receivers pass [N] (ocean: {-1, -1})
    float* cellDist;           // [N]
    float* flow;               // [N]
    uint32_t* accA; uint32_t* accB;     // pointer-doubling accumulators [N]
    unsigned long long* accCnt;         // rake (k_flow_climb): {donors arrived, running total} of a cell in one word [N]
    int32_t* jumpA; int32_t* jumpB;     // pointer-doubling ancestors [N]
    // solve dataflow
    SolveTask* task;                    // per-land-cell task record built by solve_setup [N], at the store index
    const int32_t* slotOf;              // position of a land cell in the Morton-ordered patch list, -1 for ocean [N]; nullptr: store index = cell
    SolveOut* out;                      // per-task event outputs {own turn, deposit on receiver} [N], at the store index
    double solveK, solveM, solveDt;     // the pass's constants (K, m, dt): solve_setup folds them into SolveTask::factor
    int32_t solveFinals;                // 1: records carry the finality flags and cells (SolveTask::pad_), cells without any event get their height copied by the setup
    int32_t solveLean;                  // 1: solve_setup writes the task record only — the outputs were cleared by a memset and the blocker hints are
                                        //    made from the records if a launch ever leaves tasks pending (basin-local solve: 20 of the 68 scattered bytes per task)
    int32_t* basinJ;                    // basin layout (basin.hip): the receivers pass leaves the start state of the component search here — J[slot of r] = slot of
    const int32_t* basinMslot;          //    r's land receiver, or r's own slot; basinMslot: Morton slot of a cell, nullptr: slot == cell id (land-first mirror).  nullptr: off
    int32_t* blk;                       // patch solve: granule that was seen unresolved when the task last failed, or -1 [N], at the store index
    EventList* ev;                      // events per location [N] (land entries written by flow_final_cell); nullptr: row scans
    int32_t* doneAt;                    // glacial rounds: round in which the task finished, WO_NOT_DONE before [N]
    // thermal
    double* totalExcess;                // [N]
    // glacial
    float* glac;               // glacIdx [N]
    int32_t* iceTarget;        // [N]
    float* iceFlow;            // [N]
    uint8_t* iceUp;            // numIceUpstream [N]
    int32_t* arank;            // carve: rank if the cell is an active carve task else WO_NOT_DONE [N]
    const int32_t* carveSlot;  // carve: position of an active cell in the activation list [N] (valid for active cells)
    int32_t* carveDeps;        // carve: the lower-ranked active cells within two hops of each active task, WO_CARVE_DEPS per slot
    int32_t* carveDepCnt;      // carve: how many (or -1: more than fit, use the full scan); entries before carveDepPos are finished
    int32_t* carveDepPos;
    int32_t* blocker;          // carve: the unfinished lower-ranked active cell that blocked the task at its last full scan, or -1 [N]
};

WO_HD inline double nd_or_eps(float d) { return (d == 0.0f || d != d) ? 1e-6 : (double)d; }   // `x || 1e-6`

// Neighbour rows are read in one batch: the index-order passes are bound by the latency of dependent loads (offset ->
// neighbour id -> neighbour state, once per neighbour when written as a plain loop), not by bytes, so a thread first
// issues the loads of the whole row, then the gathers of every neighbour's state, and only then computes.  Rows longer
// than WO_ROW (0.1 % of the cells of a jittered Fibonacci sphere) take the plain loops.  Slots past the degree hold the
// cell itself: a valid index whose value is never used.
constexpr int WO_ROW = 8;
// On the device a row comes in as two 16-byte loads instead of eight 4-byte ones (rows are 4-byte aligned; the mesh arrays are
// allocated WO_ROW entries longer than E, so the tail of the last rows stays inside the allocation): a quarter of the load
// instructions and address computations per wave for the same lines.
struct alignas(4) RowI4 { int32_t v[4]; };
struct alignas(4) RowF4 { float v[4]; };
// MEASURED AND REMOVED (round 5, profiles/r05b_*): a fixed-stride copy of the rows (8 neighbour slots + 8 distances per cell at 8 * r, so that
// a thread needs no adjOffset[r] before it can ask for its row: two dependent load levels instead of three).  Bit-identical, and no faster:
// receivers 21.8 against 20.0 ms per step, thermal 40.9 against 38.6 — these passes are not bound by the depth of their load chain but by the
// lines they move (a 32-byte slot per row is 1.3x the bytes of a six-neighbour CSR row).
WO_HD inline int load_row(const Fields& F, int32_t r, int32_t& b, int32_t (&nb)[WO_ROW]) {
    b = F.off[r];
    const int deg = F.off[r + 1] - b;
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(WO_ROW == 8, "two RowI4 per row");
    const RowI4 lo = *reinterpret_cast<const RowI4*>(F.adj + b), hi = *reinterpret_cast<const RowI4*>(F.adj + b + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { nb[k] = (k < deg) ? lo.v[k] : r; nb[k + 4] = (k + 4 < deg) ? hi.v[k] : r; }
#else
#pragma unroll
    for (int k = 0; k < WO_ROW; ++k) nb[k] = (k < deg) ? F.adj[b + k] : r;
#endif
    return deg;
}
// neighborDist of the row starting at b (entries beyond deg: 1.0f, never used)
WO_HD inline void load_row_dist(const Fields& F, int32_t b, int deg, float (&dd)[WO_ROW]) {
#if defined(__HIP_DEVICE_COMPILE__)
    const RowF4 lo = *reinterpret_cast<const RowF4*>(F.dist + b), hi = *reinterpret_cast<const RowF4*>(F.dist + b + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) { dd[k] = (k < deg) ? lo.v[k] : 1.0f; dd[k + 4] = (k + 4 < deg) ? hi.v[k] : 1.0f; }
#else
#pragma unroll
    for (int k = 0; k < WO_ROW; ++k) dd[k] = (k < deg) ? F.dist[b + k] : 1.0f;
#endif
}

// js/climate-util.js:13-21 smoothField: (self + neighbours) / (1 + degree), double sum in adjacency order, f32 store
WO_HD inline float smooth_field_cell(const Fields& F, const float* src, int32_t r) {
    double sum = src[r];
    int32_t count = 1;
    for (int32_t j = F.off[r]; j < F.off[r + 1]; ++j) { sum += src[F.adj[j]]; ++count; }
    return (float)(sum / count);
}

// sort key: descending elevation, -0 == +0 (comparator (a,b)=>e[b]-e[a], js/terrain-post.js:471)
WO_HD inline uint32_t desc_key(float f) {
    if (f == 0.0f) f = 0.0f;
    union { float f; uint32_t u; } v; v.f = f;
    uint32_t u = (v.u & 0x80000000u) ? ~v.u : (v.u | 0x80000000u);
    return ~u;
}

// ------------------------------------------------------------------------------------------------
// Jacobi family (js/terrain-post.js:317-354, 713-751, 758-794, 690-706)
// ------------------------------------------------------------------------------------------------
WO_HD inline uint8_t coast_flag(const Fields& F, int32_t r) {
    if (F.ocean[r]) return 0;
    for (int32_t i = F.off[r]; i < F.off[r + 1]; ++i) if (F.ocean[F.adj[i]]) return 1;
    return 0;
}

// smoothElevation, one iteration, cell r: returns the new value (js/terrain-post.js:332-350)
WO_HD inline float smooth_cell(const Fields& F, const float* e, int32_t r, double strength) {
    if (F.coast[r]) return e[r];
    const double h = e[r];
    double wSum = 0, hSum = 0;
    for (int32_t i = F.off[r]; i < F.off[r + 1]; ++i) {
        const double nh = e[F.adj[i]];
        const double diff = fabs(nh - h);
        const double w = 1 / (1 + diff * 8);
        wSum += w;
        hSum += nh * w;
    }
    if (wSum > 0) { const double avg = hSum / wSum; return (float)(h + (avg - h) * strength); }
    return (float)h;
}

// sharpenRidges, one iteration (js/terrain-post.js:728-747); ocean cells keep their value
WO_HD inline float sharpen_cell(const Fields& F, const float* e, const float* original, int32_t r, double strength) {
    if (F.ocean[r]) return e[r];
    const double h = e[r];
    double sum = 0;
    const int32_t count = F.off[r + 1] - F.off[r];
    for (int32_t i = F.off[r]; i < F.off[r + 1]; ++i) sum += e[F.adj[i]];
    if (count == 0) return (float)h;
    const double avg = sum / count;
    if (h > avg) {
        double hn = h + (h - avg) * strength;
        const double cap = (double)original[r] * 1.5;
        if (hn > cap) hn = cap;
        return (float)hn;
    }
    return (float)h;
}

// applySoilCreep, one iteration (js/terrain-post.js:777-791); only interior land moves
WO_HD inline float creep_cell(const Fields& F, const float* e, int32_t r, double strength) {
    if (F.ocean[r] || F.coast[r]) return e[r];
    const double h = e[r];
    double sum = 0; int32_t count = 0;
    for (int32_t i = F.off[r]; i < F.off[r + 1]; ++i) {
        const int32_t nb = F.adj[i];
        if (!F.ocean[nb]) { sum += e[nb]; ++count; }
    }
    if (count == 0) return (float)h;
    return (float)(h + (sum / count - h) * strength);
}

// post-loop glacial blend (js/terrain-post.js:690-706)
WO_HD inline float glacial_blend_cell(const Fields& F, const float* e, int32_t r) {
    if (F.ocean[r] || !(F.glac[r] > 0)) return e[r];
    double sum = 0; int32_t count = 0;
    for (int32_t j = F.off[r]; j < F.off[r + 1]; ++j) {
        const int32_t nb = F.adj[j];
        if (!F.ocean[nb]) { sum += e[nb]; ++count; }
    }
    if (count > 0) { const double avg = sum / count; return (float)((double)e[r] + (avg - (double)e[r]) * 0.3); }
    return e[r];
}

// ------------------------------------------------------------------------------------------------
// warpTerrain (js/terrain-post.js:245-308): source cell of the greedy walk, then the blend
// ------------------------------------------------------------------------------------------------
WO_HD inline int32_t warp_source_cell(const Fields& F, const uint8_t* P, const uint8_t* M, int32_t r, double maxAmp) {
    const double px = F.xyz[3 * r], py = F.xyz[3 * r + 1], pz = F.xyz[3 * r + 2];
    double ex = -pz, ey = 0, ez = px;
    const double elen = sqrt(ex * ex + ez * ez);
    if (elen > 1e-10) { ex /= elen; ez /= elen; } else { ex = 1; ez = 0; }
    const double nx = py * ez, ny = pz * ex - px * ez, nz = -py * ex;
    double nlen = sqrt(nx * nx + ny * ny + nz * nz);
    if (nlen == 0 || nlen != nlen) nlen = 1;
    const double nnx = nx / nlen, nny = ny / nlen, nnz = nz / nlen;
    const double freq = 4;
    const double d1 = fbm(P, M, px * freq, py * freq, pz * freq, 5) * maxAmp;
    const double d2 = fbm(P, M, px * freq + 31.7, py * freq + 47.3, pz * freq + 19.1, 5) * maxAmp;
    double wx = px + ex * d1 + nnx * d2;
    double wy = py + ey * d1 + nny * d2;
    double wz = pz + ez * d1 + nnz * d2;
    double wlen = sqrt(wx * wx + wy * wy + wz * wz);
    if (wlen == 0 || wlen != wlen) wlen = 1;
    wx /= wlen; wy /= wlen; wz /= wlen;
    int32_t cur = r;
    double bestDot = wx * px + wy * py + wz * pz;
    for (;;) {
        bool moved = false;
        const int32_t iEnd = F.off[cur + 1];
        for (int32_t i = F.off[cur]; i < iEnd; ++i) {
            const int32_t nb = F.adj[i];
            const double dot = wx * F.xyz[3 * nb] + wy * F.xyz[3 * nb + 1] + wz * F.xyz[3 * nb + 2];
            if (dot > bestDot) { bestDot = dot; cur = nb; moved = true; }
        }
        if (!moved) break;
    }
    return cur;
}

WO_HD inline float warp_blend(float origF, float warpedF, double warpBias, bool useHot, float hotF) {
    const double orig = origF, warped = warpedF;
    double bias = warpBias;
    if (useHot) {
        double den = fabs(orig); if (den == 0) den = 1;
        double hf = fabs((double)hotF) / den; if (hf > 1) hf = 1;
        bias *= 1 - 0.8 * hf;
    }
    if (warped > orig) return (float)(orig + (warped - orig) * bias);
    return (float)(warped + (orig - warped) * (1 - bias));
}

// ------------------------------------------------------------------------------------------------
// Hydraulic: receivers (js/terrain-post.js:566-601)
// ------------------------------------------------------------------------------------------------
// elev(c): the elevation of cell c — F.e[c], or the workgroup's LDS copy of its tile's neighbourhood (kernels_impl.h: TileWindow)
template <class Elev>
WO_HD inline int32_t receiver_cell_t(const Fields& F, int32_t r, Elev elev) {
    if (F.ocean[r]) { if (F.target) F.target[r] = -1; TargetRank z; z.target = -1; z.rank = -1; F.tr[r] = z; return -1; }
    const double h = elev(r);
    int32_t bestNb = -1, bestJ = -1;
    double bestDrop = -INFINITY;
    int32_t b, nbs[WO_ROW];
    const int deg = load_row(F, r, b, nbs);
    const int32_t en = b + deg;
    if (deg <= WO_ROW) {
        float eh[WO_ROW];
#pragma unroll
        for (int k = 0; k < WO_ROW; ++k) eh[k] = elev(nbs[k]);
#pragma unroll
        for (int k = 0; k < WO_ROW; ++k) {
            const double drop = h - (double)eh[k];
            if (k < deg && drop > bestDrop) { bestDrop = drop; bestNb = nbs[k]; bestJ = b + k; }
        }
        if (bestDrop <= 0) {
            double minAscent = INFINITY;
#pragma unroll
            for (int k = 0; k < WO_ROW; ++k) {
                const double ascent = (double)eh[k] - h;
                if (k < deg && ascent < minAscent) { minAscent = ascent; bestNb = nbs[k]; bestJ = b + k; }
            }
        }
    } else {
        for (int32_t j = b; j < en; ++j) {
            const int32_t nb = F.adj[j];
            const double drop = h - (double)elev(nb);
            if (drop > bestDrop) { bestDrop = drop; bestNb = nb; bestJ = j; }
        }
        if (bestDrop <= 0) {
            double minAscent = INFINITY;
            for (int32_t j = b; j < en; ++j) {
                const int32_t nb = F.adj[j];
                const double ascent = (double)elev(nb) - h;
                if (ascent < minAscent) { minAscent = ascent; bestNb = nb; bestJ = j; }
            }
        }
    }
    if (F.target) F.target[r] = bestNb;          // (the device passes read the target out of tr[]: Fields::target is nullptr there)
    { TargetRank v; v.target = bestNb; v.rank = F.rank[r]; F.tr[r] = v; }
    if (bestNb >= 0) { const float d = F.dist[bestJ]; F.cellDist[r] = (d == 0.0f || d != d) ? (float)1e-6 : d; }
    return bestNb;
}

WO_HD inline int32_t receiver_cell(const Fields& F, int32_t r) { return receiver_cell_t(F, r, [&](int32_t c) { return F.e[c]; }); }

// Flow (js/terrain-post.js:604-611).  fwd edge: receiver is land and ranked after the donor.
WO_HD inline int32_t flow_forward_target(const Fields& F, int32_t r) {
    const int32_t t = F.tr[r].target;
    if (t < 0 || F.ocean[t]) return -1;
    return (F.rank[r] < F.rank[t]) ? t : -1;
}

// final flow of land cell c (js/terrain-post.js:604-611): its forwarded total plus the totals of its late donors (ranked
// after c: the serial loop adds them to flow[c] after c has already passed its own total on), and the event list of c.
WO_HD inline void event_insert(EventList& E, int& n, bool& over, int32_t cell, int32_t rank) {
    if (n == WO_EVENTS) { over = true; return; }
    ++n;
    // descending rank, unused slots hold rank -1: the new event sinks in from the top and pushes the smaller ones down
    // (static indices only: the list stays in registers; a dynamically indexed array went to scratch memory)
#pragma unroll
    for (int q = 0; q < WO_EVENTS; ++q) {
        const bool up = rank > E.rank[q];
        const int32_t c2 = E.cell[q], r2 = E.rank[q];
        E.cell[q] = up ? cell : c2; E.rank[q] = up ? rank : r2;
        cell = up ? c2 : cell; rank = up ? r2 : rank;
    }
}
// forward total of a cell after the flow accumulation: the packed {arrived, total} word when the one-launch rake retired every
// cell (Fields::accCnt set by the driver), else the pointer doubling's accumulator
WO_HD inline uint32_t flow_total(const Fields& F, int32_t c) { return F.accCnt ? (uint32_t)F.accCnt[c] : F.accA[c]; }
WO_HD inline void flow_final_cell(const Fields& F, int32_t c) {
    uint32_t f = flow_total(F, c);
    const TargetRank trc = F.tr[c];
    const int32_t rc = trc.rank;
    EventList E;
#pragma unroll
    for (int q = 0; q < WO_EVENTS; ++q) { E.cell[q] = -1; E.rank[q] = -1; }
    int n = 0; bool over = false;
    if (trc.target >= 0) event_insert(E, n, over, c, rc);
    int32_t b, nbs[WO_ROW];
    const int deg = load_row(F, c, b, nbs);
    if (deg <= WO_ROW) {
        TargetRank q[WO_ROW];
#pragma unroll
        for (int k = 0; k < WO_ROW; ++k) q[k] = F.tr[nbs[k]];                        // ocean cells carry target -1
#pragma unroll
        for (int k = 0; k < WO_ROW; ++k) {
            if (k < deg && q[k].target == c) {
                if (q[k].rank > rc) f += flow_total(F, nbs[k]);                              // late donor, un-forwarded
                event_insert(E, n, over, nbs[k], q[k].rank);
            }
        }
    } else {
        for (int32_t j = b; j < b + deg; ++j) {
            const int32_t nb = F.adj[j];
            const TargetRank v = F.tr[nb];
            if (v.target != c) continue;
            if (v.rank > rc) f += flow_total(F, nb);
            event_insert(E, n, over, nb, v.rank);
        }
    }
    F.flow[c] = (float)f;
    if (F.ev) { if (over) E.rank[0] = -2; F.ev[c] = E; }
}
// latest event on the list's location strictly before the turn of task r (rank rr), r's own events excluded; -1: none
WO_HD inline int32_t event_before(const EventList& E, int32_t r, int32_t rr) {
    int32_t best = -1;
#pragma unroll
    for (int q = 0; q < WO_EVENTS; ++q) if (E.rank[q] > rr && E.cell[q] != r) best = E.cell[q];     // descending ranks: the last hit is the closest
    return best;
}

// ------------------------------------------------------------------------------------------------
// Hydraulic: implicit solve + deposition as dataflow (js/terrain-post.js:614-641)
// Processing time of a land cell: tau = -rank (the serial loop walks landCells backwards), so
// "earlier" == larger rank.
// ------------------------------------------------------------------------------------------------
// latest event on location x strictly before the turn of task r (exclusive of r itself):
// events on x are x's own turn and the turns of its donors (neighbours n with target[n]==x).
WO_HD inline int32_t latest_event_before(const Fields& F, int32_t x, int32_t r) {
    const int32_t rr = F.tr[r].rank;
    int32_t best = -1, bestRank = -1;          // earlier == larger rank; want the smallest rank that is > rr
    if (x != r) { const TargetRank v = F.tr[x]; if (v.target >= 0 && v.rank > rr) { best = x; bestRank = v.rank; } }
    for (int32_t j = F.off[x]; j < F.off[x + 1]; ++j) {
        const int32_t n = F.adj[j];
        if (n == r) continue;
        const TargetRank v = F.tr[n];          // ocean cells carry target -1, never == x
        if (v.target != x) continue;
        if (v.rank > rr && (best < 0 || v.rank < bestRank)) { best = n; bestRank = v.rank; }
    }
    return best;
}

// the same with x's row and the {target, rank} of its neighbours already loaded (trx = tr[x])
WO_HD inline int32_t latest_event_before_row(int32_t x, int32_t r, int32_t rr, TargetRank trx, const int32_t (&nb)[WO_ROW], const TargetRank (&trn)[WO_ROW], int deg) {
    int32_t best = -1, bestRank = -1;
    if (x != r && trx.target >= 0 && trx.rank > rr) { best = x; bestRank = trx.rank; }
#pragma unroll
    for (int k = 0; k < WO_ROW; ++k) {
        const TargetRank v = trn[k];
        if (k < deg && nb[k] != r && v.target == x && v.rank > rr && (best < 0 || v.rank < bestRank)) { best = nb[k]; bestRank = v.rank; }
    }
    return best;
}

WO_HD inline int32_t store_index(const Fields& F, int32_t cell) { return F.slotOf ? F.slotOf[cell] : cell; }
// granule index of the event task p leaves on location x (p == x: own turn, else deposit)
WO_HD inline int32_t granule_index(const Fields& F, int32_t x, int32_t p) { return p < 0 ? -1 : 2 * store_index(F, p) + (p == x ? 0 : 1); }

WO_HD inline double solve_factor_of(float flow, float cellDist, double K, double m, double dt) {
    const double fl = flow;
    const double pw = (m == 0.5) ? sqrt(fl) : pow(fl, m);
    return K * pw * dt / (double)cellDist;
}
WO_HD inline void solve_setup_cell_plain(const Fields& F, int32_t r);
WO_HD inline void solve_setup_cell_rows(const Fields& F, int32_t r);
// setup from the event lists (flow_final_cell); any list involved overflowed -> the row scans
WO_HD inline void solve_setup_cell(const Fields& F, int32_t r) {
    if (F.ocean[r]) return;
    if (!F.ev) { solve_setup_cell_rows(F, r); return; }
    const TargetRank trr = F.tr[r];
    const int32_t t = trr.target, rr = trr.rank;
    const EventList Er = F.ev[r];
    TargetRank trt; trt.target = -1; trt.rank = -1;
    if (t >= 0) trt = F.tr[t];
    const bool tLand = t >= 0 && trt.rank >= 0;                     // ocean cells carry rank -1
    EventList Et, Et2; Et.rank[0] = -1; Et2.rank[0] = -1;
    float cdT = 0.0f;
    if (tLand) { Et = F.ev[t]; cdT = F.cellDist[t]; }
    const int32_t t2 = (tLand && trt.target >= 0 && cdT > 0) ? trt.target : -1;
    TargetRank trt2; trt2.target = -1; trt2.rank = -1;
    if (t2 >= 0) trt2 = F.tr[t2];
    const bool t2Land = t2 >= 0 && trt2.rank >= 0;
    if (t2Land) Et2 = F.ev[t2];
    if (Er.rank[0] == -2 || Et.rank[0] == -2 || Et2.rank[0] == -2) { solve_setup_cell_plain(F, r); return; }
    SolveTask T;
    T.predSelf = granule_index(F, r, event_before(Er, r, rr));
    T.predT = -1; T.predT2 = -1; T.flags = 0; T.pad_[0] = T.pad_[1] = 0;
    T.e0r = F.e[r]; T.e0t = 0; T.e0t2 = 0; T.cellDistT = 0;
    T.factor = solve_factor_of(F.flow[r], F.cellDist[r], F.solveK, F.solveM, F.solveDt);
    if (t >= 0) {
        T.flags |= 4u;
        T.e0t = F.e[t];
        if (!tLand) T.flags |= 1u;
        else {
            T.predT = granule_index(F, t, event_before(Et, r, rr));
            T.cellDistT = cdT;
            if (t2 >= 0) {
                T.flags |= 8u;
                T.e0t2 = F.e[t2];
                if (!t2Land) T.flags |= 2u;
                else T.predT2 = granule_index(F, t2, event_before(Et2, r, rr));
            }
        }
    }
    const int32_t si = store_index(F, r);
    F.task[si] = T;
    SolveOut z; z.self.v = 0; z.self.tag = 0; z.dep.v = 0; z.dep.tag = 0;
    if (!F.solveLean) F.out[si] = z;
    if (F.blk && !F.solveLean) F.blk[si] = T.predT >= 0 ? T.predT : (T.predSelf >= 0 ? T.predSelf : T.predT2);
}
WO_HD inline void solve_setup_cell_rows(const Fields& F, int32_t r) {
    if (F.ocean[r]) return;
    // batched form: rows of r, of its receiver t and of t's receiver t2 first, then every neighbour's {target, rank}
    const TargetRank trr = F.tr[r];
    const int32_t t = trr.target, rr = trr.rank;
    int32_t bR, nbR[WO_ROW], bT = 0, nbT[WO_ROW], bT2 = 0, nbT2[WO_ROW];
    const int degR = load_row(F, r, bR, nbR);
    TargetRank trt; trt.target = -1; trt.rank = -1;
    int degT = 0, degT2 = 0;
    if (t >= 0) { trt = F.tr[t]; degT = load_row(F, t, bT, nbT); }
    const bool tLand = t >= 0 && trt.rank >= 0;                     // ocean cells carry rank -1
    const float cdT = tLand ? F.cellDist[t] : 0.0f;
    const int32_t t2 = (tLand && trt.target >= 0 && cdT > 0) ? trt.target : -1;
    TargetRank trt2; trt2.target = -1; trt2.rank = -1;
    if (t2 >= 0) { trt2 = F.tr[t2]; degT2 = load_row(F, t2, bT2, nbT2); }
    if (degR > WO_ROW || degT > WO_ROW || degT2 > WO_ROW) { solve_setup_cell_plain(F, r); return; }
    TargetRank qR[WO_ROW], qT[WO_ROW], qT2[WO_ROW];
#pragma unroll
    for (int k = 0; k < WO_ROW; ++k) qR[k] = F.tr[nbR[k]];
    if (tLand) {
#pragma unroll
        for (int k = 0; k < WO_ROW; ++k) qT[k] = F.tr[nbT[k]];
    }
    const bool t2Land = t2 >= 0 && trt2.rank >= 0;
    if (t2Land) {
#pragma unroll
        for (int k = 0; k < WO_ROW; ++k) qT2[k] = F.tr[nbT2[k]];
    }
    SolveTask T;
    T.predSelf = granule_index(F, r, latest_event_before_row(r, r, rr, trr, nbR, qR, degR));
    T.predT = -1; T.predT2 = -1; T.flags = 0; T.pad_[0] = T.pad_[1] = 0;
    T.e0r = F.e[r]; T.e0t = 0; T.e0t2 = 0; T.cellDistT = 0;
    T.factor = solve_factor_of(F.flow[r], F.cellDist[r], F.solveK, F.solveM, F.solveDt);
    if (t >= 0) {
        T.flags |= 4u;
        T.e0t = F.e[t];
        if (!tLand) T.flags |= 1u;
        else {
            T.predT = granule_index(F, t, latest_event_before_row(t, r, rr, trt, nbT, qT, degT));
            T.cellDistT = cdT;
            if (t2 >= 0) {
                T.flags |= 8u;
                T.e0t2 = F.e[t2];
                if (!t2Land) T.flags |= 2u;
                else T.predT2 = granule_index(F, t2, latest_event_before_row(t2, r, rr, trt2, nbT2, qT2, degT2));
            }
        }
    }
    const int32_t si = store_index(F, r);
    F.task[si] = T;
    SolveOut z; z.self.v = 0; z.self.tag = 0; z.dep.v = 0; z.dep.tag = 0;
    if (!F.solveLean) F.out[si] = z;
    if (F.blk && !F.solveLean) F.blk[si] = T.predT >= 0 ? T.predT : (T.predSelf >= 0 ? T.predSelf : T.predT2);
}
WO_HD inline void solve_setup_cell_plain(const Fields& F, int32_t r) {
    if (F.ocean[r]) return;
    SolveTask T;
    const int32_t t = F.tr[r].target;
    T.predSelf = granule_index(F, r, latest_event_before(F, r, r));
    T.predT = -1; T.predT2 = -1; T.flags = 0; T.pad_[0] = T.pad_[1] = 0;
    T.e0r = F.e[r]; T.e0t = 0; T.e0t2 = 0; T.cellDistT = 0;
    T.factor = solve_factor_of(F.flow[r], F.cellDist[r], F.solveK, F.solveM, F.solveDt);
    if (t >= 0) {
        T.flags |= 4u;
        T.e0t = F.e[t];
        if (F.ocean[t]) T.flags |= 1u;
        else {
            T.predT = granule_index(F, t, latest_event_before(F, t, r));
            const int32_t t2 = F.tr[t].target;
            T.cellDistT = F.cellDist[t];
            if (t2 >= 0 && T.cellDistT > 0) {
                T.flags |= 8u;
                T.e0t2 = F.e[t2];
                if (F.ocean[t2]) T.flags |= 2u;
                else T.predT2 = granule_index(F, t2, latest_event_before(F, t2, r));
            }
        }
    }
    const int32_t si = store_index(F, r);
    F.task[si] = T;
    SolveOut z; z.self.v = 0; z.self.tag = 0; z.dep.v = 0; z.dep.tag = 0;
    if (!F.solveLean) F.out[si] = z;
    // any unresolved predecessor is a valid first blocker; the receiver's event usually resolves last
    if (F.blk && !F.solveLean) F.blk[si] = T.predT >= 0 ? T.predT : (T.predSelf >= 0 ? T.predSelf : T.predT2);
}

// (Divisions by per-task constants taken off the dependency chain by preparing the divisor's reciprocal before the task starts to wait — the hardware division
// sequence split at the point where it stops depending on the divisor alone — were built for the patch solve in round 2: same bits, measured slower, 345 -> 361-391 ms
// per step, profiles/r02x_prepared_division_ab.txt; removed in round 6.  The basin walk's own form of the idea — reciprocals refined while the record arrives,
// the turn keeps the three dependent operations of each division — is in basin.hip: recip_refined / div_tail.)
// one turn of the implicit solve + deposition (js/terrain-post.js:616-640) given its three inputs.
// solve_prepare is the part that does not depend on the predecessors (callers on a dependency chain hoist it).
struct SolvePrepared { double factor, onePlusFactor, cellDistT; };
WO_HD inline double solve_factor(const SolveTask& T, double, double, double) { return T.factor; }
WO_HD inline SolvePrepared solve_prepare(const SolveTask& T, double K, double m, double dt) {
    SolvePrepared S;
    S.factor = solve_factor(T, K, m, dt);
    S.onePlusFactor = 1 + S.factor;
    S.cellDistT = (double)T.cellDistT;
    return S;
}
WO_HD inline SolveOut solve_apply(const SolveTask& T, const SolvePrepared& S, double er, double et, double et2, int32_t tag) {
    SolveOut o;
    o.self.tag = tag; o.dep.tag = tag;
    if (!(T.flags & 4u)) {       // isolated cell: the serial loop skips it (cellDist is > 0 by construction otherwise)
        o.self.v = (float)er; o.dep.v = 0; return o;
    }
    const double hr = et > 0 ? et : 0;
    double hn = (er + S.factor * hr) / S.onePlusFactor;
    if (hn < hr) hn = hr;
    if (hn < 0) hn = 0;
    const double eroded = er - hn;
    float tval = (float)et;
    if (eroded > 0 && !(T.flags & 1u)) {
        double slope = 0;
        if (T.flags & 8u) slope = fabs(et - et2) / S.cellDistT;
        const double depositFrac = 0.5 / (1 + slope * 50);
        const double deposit = eroded * depositFrac;
        tval = (float)(et + deposit);
        if ((double)tval > hn) tval = (float)hn;
    }
    o.self.v = (float)hn; o.dep.v = tval;
    return o;
}
// The same turn without a branch, for a wave that runs it for whichever lanes are ready (basin.hip: k_solve_flowing): every
// expression of solve_apply is evaluated, on the same operands, and the conditions select among the results — an unused
// quotient may be inf / NaN (a missing t2 has cellDistT 0), it is never selected.
WO_HD inline SolveOut solve_apply_flat(const SolveTask& T, const SolvePrepared& S, double er, double et, double et2, int32_t tag) {
    const bool hasT = (T.flags & 4u) != 0, tOcean = (T.flags & 1u) != 0, hasT2 = (T.flags & 8u) != 0;
    const double hr = et > 0 ? et : 0;
    double hn = (er + S.factor * hr) / S.onePlusFactor;
    hn = hn < hr ? hr : hn;
    hn = hn < 0 ? 0 : hn;
    const double eroded = er - hn;
    const double sl = fabs(et - et2) / S.cellDistT;
    const double slope = hasT2 ? sl : 0.0;
    const double depositFrac = 0.5 / (1 + slope * 50);
    const double deposit = eroded * depositFrac;
    float tv = (float)(et + deposit);
    tv = ((double)tv > hn) ? (float)hn : tv;
    const bool deposits = eroded > 0 && !tOcean;
    SolveOut o;
    o.self.tag = tag; o.dep.tag = tag;
    o.self.v = hasT ? (float)hn : (float)er;
    o.dep.v = hasT ? (deposits ? tv : (float)et) : 0.0f;
    return o;
}
WO_HD inline SolveOut solve_compute(const SolveTask& T, double er, double et, double et2, int32_t tag, double K, double m, double dt) {
    return solve_apply(T, solve_prepare(T, K, m, dt), er, et, et2, tag);
}

// Returns true when the task ran (all predecessors were produced in rounds < round).  Level-round schedule only
// (store index = cell id).
WO_HD inline bool solve_task(const Fields& F, int32_t r, int32_t round, double K, double m, double dt) {
    const SolveTask T = F.task[r];
    const Granule* G = reinterpret_cast<const Granule*>(F.out);
    double er = T.e0r, et = T.e0t, et2 = T.e0t2;
    if (T.predSelf >= 0) { const Granule g = G[T.predSelf]; if (g.tag == 0 || !(g.tag < round)) return false; er = g.v; }
    if (T.predT >= 0)    { const Granule g = G[T.predT];    if (g.tag == 0 || !(g.tag < round)) return false; et = g.v; }
    if (T.predT2 >= 0)   { const Granule g = G[T.predT2];   if (g.tag == 0 || !(g.tag < round)) return false; et2 = g.v; }
    F.out[r] = solve_compute(T, er, et, et2, round, K, m, dt);
    return true;
}

// the task that leaves the LAST event on land cell x in a pass (x itself: its own turn; a donor: its deposit), by the row scan; -1: no event
WO_HD inline int32_t latest_event_cell(const Fields& F, int32_t x) {
    int32_t best = -1, bestRank = 0x7fffffff;      // latest == smallest rank
    { const TargetRank v = F.tr[x]; if (v.target >= 0) { best = x; bestRank = v.rank; } }
    int32_t b, nbs[WO_ROW];
    const int deg = load_row(F, x, b, nbs);
    if (deg <= WO_ROW) {
        TargetRank q[WO_ROW];
#pragma unroll
        for (int k = 0; k < WO_ROW; ++k) q[k] = F.tr[nbs[k]];
#pragma unroll
        for (int k = 0; k < WO_ROW; ++k) if (k < deg && q[k].target == x && q[k].rank < bestRank) { best = nbs[k]; bestRank = q[k].rank; }
    } else {
        for (int32_t j = b; j < b + deg; ++j) {
            const int32_t n = F.adj[j];
            const TargetRank v = F.tr[n];              // ocean cells carry target -1, never == x
            if (v.target != x) continue;
            if (v.rank < bestRank) { best = n; bestRank = v.rank; }
        }
    }
    return best;
}
// final height of land cell x after the pass = value left by the latest event on x
WO_HD inline float solve_final_cell(const Fields& F, int32_t x) {
    if (F.ocean[x]) return F.e[x];
    if (F.ev) {
        const EventList E = F.ev[x];
        if (E.rank[0] != -2) {
            int32_t last = -1;
#pragma unroll
            for (int q = 0; q < WO_EVENTS; ++q) if (E.rank[q] >= 0) last = E.cell[q];           // descending ranks: the last one is the latest
            if (last < 0) return F.e[x];
            return (last == x) ? F.out[store_index(F, x)].self.v : F.out[store_index(F, last)].dep.v;
        }
    }
    const int32_t best = latest_event_cell(F, x);
    if (best < 0) return F.e[x];
    return (best == x) ? F.out[store_index(F, x)].self.v : F.out[store_index(F, best)].dep.v;
}

// ------------------------------------------------------------------------------------------------
// Thermal (js/terrain-post.js:645-686) in exact gather form
// ------------------------------------------------------------------------------------------------
// F.me[r] = isOcean ? +inf : e.  An ocean neighbour then fails `nh < h`, and in the `nh > h` branch it contributes
// nothing because its totalExcess is 0 — exactly the effect of the reference's `if (r_isOcean[nb]) continue`.
WO_HD inline float masked_elev_cell(const Fields& F, int32_t r) { return F.ocean[r] ? INFINITY : F.e[r]; }

// Cheap exact filter for `(x / d) > talus` (x >= 0, d > 0, doubles): when x <= talus * d * (1 - 1e-12) the real quotient is
// below talus, and since rounding is monotone and talus is a double the rounded quotient cannot exceed it — the division
// (an f64 division is ~30 VALU operations, and the thermal kernels are bound by them: 110 M VALU instructions per launch
// at 10 M cells, profiles/r04b_*) is only evaluated where the test can still come out true.  After the first ~80 iterations
// few slopes exceed the talus angle, so whole waves skip every division of the pass.  Same results bit for bit: the filter
// only decides whether the exact expression is evaluated.
WO_HD inline bool talus_may_exceed(double x, double d, double talus) { return x > talus * d * (1.0 - 1e-12); }

// masked(c): F.me[c], or the workgroup's LDS copy of it (kernels_impl.h: TileWindow)
template <class Masked>
WO_HD inline void thermal_excess_cell_t(const Fields& F, int32_t r, double talus, Masked masked) {
    double total = 0;
    if (!F.ocean[r]) {
        const double h = masked(r);                            // a land cell's masked height is its height
        int32_t b, nbs[WO_ROW];
        const int deg = load_row(F, r, b, nbs);
        if (deg <= WO_ROW) {
            float mh[WO_ROW], dd[WO_ROW];
#pragma unroll
            for (int k = 0; k < WO_ROW; ++k) mh[k] = masked(nbs[k]);
            load_row_dist(F, b, deg, dd);
#pragma unroll
            for (int k = 0; k < WO_ROW; ++k) {
                const double nh = mh[k];
                if (k >= deg || nh >= h) continue;
                const double d = nd_or_eps(dd[k]);
                if (!talus_may_exceed(h - nh, d, talus)) continue;
                const double slope = (h - nh) / d;
                if (slope > talus) total += (slope - talus) * d;
            }
        } else {
            for (int32_t j = b; j < b + deg; ++j) {
                const double nh = masked(F.adj[j]);
                if (nh >= h) continue;
                const double d = nd_or_eps(F.dist[j]);
                const double slope = (h - nh) / d;
                if (slope > talus) total += (slope - talus) * d;
            }
        }
    }
    F.totalExcess[r] = total;
}

WO_HD inline void thermal_excess_cell(const Fields& F, int32_t r, double talus) { thermal_excess_cell_t(F, r, talus, [&](int32_t c) { return F.me[c]; }); }

// New height of cell c (reads F.e, the pre-thermal field).  inShare / inRank are caller-provided scratch for the
// <= degree incoming events, element k at [k * stride] (LDS columns on the device: a per-thread array indexed at
// run time would live in scratch memory and tripled this kernel's HBM traffic).
// thermal_apply_cell for rows of at most WO_ROW neighbours (99.9 % of the cells), with every per-neighbour quantity in a
// statically indexed slot: the replay of the serial loop's additions (senders in ascending rank, c's own sends at c's turn
// in adjacency order) selects by scanning the 8 slots instead of indexing a list, so nothing lives in scratch memory.
template <class Masked>
WO_HD inline float thermal_apply_row(const Fields& F, double h, int32_t myRank, double myTotal, int32_t b, const int32_t (&nbs)[WO_ROW], int deg,
                                     double talus, double kThermal, Masked masked) {
    float mh[WO_ROW], dd[WO_ROW];
#pragma unroll
    for (int k = 0; k < WO_ROW; ++k) mh[k] = masked(nbs[k]);
    load_row_dist(F, b, deg, dd);
    // One slope per neighbour, whichever way it points: (h - nh) / d and (nh - h) / d are the same quotient up to the sign, so the
    // serial loop's `slope` of the higher cell of the pair is |nh - h| / d either way, and the share it moves is
    // (f32(excess) / total) * (kThermal * total * 0.5) with the SENDER's total — c's own (c sends on its own turn) or the
    // neighbour's (it sends on its turn).  Two divisions per steep pair instead of five in the wave's instruction stream; the
    // exact pre-test (talus_may_exceed) skips both where the slope cannot exceed the talus angle.  An ocean neighbour's masked
    // height is +inf: never a sender (its totalExcess is 0), and never lower than c.
    double inSh[WO_ROW], outSh[WO_ROW]; int32_t inRk[WO_ROW]; bool outOn[WO_ROW];
    bool hasOut = false, hasIn = false;
#pragma unroll
    for (int k = 0; k < WO_ROW; ++k) {
        inRk[k] = 0x7fffffff; inSh[k] = 0; outSh[k] = 0; outOn[k] = false;
        const double nh = mh[k], d = nd_or_eps(dd[k]);
        const double gap = fabs(nh - h);
        if (!(k < deg && nh != (double)INFINITY && talus_may_exceed(gap, d, talus))) continue;
        const double slope = gap / d;
        if (!(slope > talus)) continue;
        const bool sends = nh < h;                 // c is the higher cell: it sends to nb on its own turn; else nb sends to c on nb's turn
        if (!sends && !(nh > h)) continue;
        const double tot = sends ? myTotal : F.totalExcess[nbs[k]];
        if (!(tot > 0)) continue;
        const float excess = (float)((slope - talus) * d);        // excVal is a Float32Array
        const double share = ((double)excess / tot) * (kThermal * tot * 0.5);
        if (sends) { hasOut = true; outOn[k] = true; outSh[k] = share; }
        else { hasIn = true; inSh[k] = share; inRk[k] = F.rank[nbs[k]]; }
    }
    if (!hasIn && !hasOut) return (float)(h + 0.0);   // e += delta with delta == 0 (also maps -0 -> +0 like the f32 add)
    float delta = 0.0f;
    int32_t last = -1;
    bool ownPending = hasOut;
    for (int it = 0; it < WO_ROW + 2; ++it) {
        int32_t pr = 0x7fffffff; double v = 0;
#pragma unroll
        for (int k = 0; k < WO_ROW; ++k) { const bool hit = inRk[k] > last && inRk[k] < pr; pr = hit ? inRk[k] : pr; v = hit ? inSh[k] : v; }
        if (ownPending && myRank < pr) {        // c's own turn comes before the next sender's: its sends, in adjacency order (js/terrain-post.js:676-680)
#pragma unroll
            for (int k = 0; k < WO_ROW; ++k) if (outOn[k]) delta = (float)((double)delta - outSh[k]);
            ownPending = false;
            continue;
        }
        if (pr == 0x7fffffff) break;
        delta = (float)((double)delta + v);
        last = pr;
    }
    return (float)(h + (double)delta);
}

template <class Masked>
WO_HD inline float thermal_apply_cell_t(const Fields& F, int32_t c, double talus, double kThermal, double* inShare, int32_t* inRank, int stride,
                                        double* outShare, Masked masked) {
    if (F.ocean[c]) return F.e[c];
    const double h = F.e[c];
    const int32_t myRank = F.rank[c];
    const double myTotal = F.totalExcess[c];
    const double myTransfer = kThermal * myTotal * 0.5;
    int nIn = 0, nOut = 0; bool hasOut = false;
    int32_t b, nbs[WO_ROW];
    const int deg = load_row(F, c, b, nbs);
    if (deg <= WO_ROW) return thermal_apply_row(F, h, myRank, myTotal, b, nbs, deg, talus, kThermal, masked);
    {
        for (int32_t j = b; j < b + deg; ++j) {
            const int32_t nb = F.adj[j];
            const double nh = F.me[nb];         // +inf for ocean neighbours: falls into the branch below and finds totalExcess 0
            const double d = nd_or_eps(F.dist[j]);
            if (nh < h) {                       // c sends to nb on c's own turn
                const double slope = (h - nh) / d;
                if (slope > talus && myTotal > 0) {
                    hasOut = true;
                    if (outShare) { const float excess = (float)((slope - talus) * d); outShare[nOut++ * stride] = ((double)excess / myTotal) * myTransfer; }
                }
            } else if (nh > h) {                // nb may send to c on nb's turn
                const double slope = (nh - h) / d;
                if (slope > talus) {
                    const double tot = F.totalExcess[nb];
                    if (tot > 0) {
                        const float excess = (float)((slope - talus) * d);        // excVal is a Float32Array
                        inShare[nIn * stride] = ((double)excess / tot) * (kThermal * tot * 0.5);
                        inRank[nIn * stride] = F.rank[nb];
                        ++nIn;
                    }
                }
            }
        }
    }
    if (nIn == 0 && !hasOut) return (float)(h + 0.0);   // e += delta with delta == 0 (also maps -0 -> +0 like the f32 add)
    float delta = 0.0f;
    int32_t last = -1;
    for (int phase = 0; phase < 2; ++phase) {
        // phase 0: senders whose turn precedes c's own turn; phase 1: the ones after it
        for (;;) {
            int pick = -1; int32_t pr = 0x7fffffff;
            for (int k = 0; k < nIn; ++k) { const int32_t rk = inRank[k * stride]; if (rk > last && rk < pr) { pr = rk; pick = k; } }
            if (pick < 0 || (phase == 0 && pr > myRank)) break;
            delta = (float)((double)delta + inShare[pick * stride]);
            last = pr;
        }
        if (phase == 0 && hasOut && outShare) {      // c's own turn: its sends, in adjacency order (js/terrain-post.js:676-680)
            for (int k = 0; k < nOut; ++k) delta = (float)((double)delta - outShare[k * stride]);
        } else if (phase == 0 && hasOut) {
            for (int32_t j = b; j < b + deg; ++j) {
                const double nh = (double)F.me[F.adj[j]];
                if (!(nh < h)) continue;
                const double d = nd_or_eps(F.dist[j]);
                const double slope = (h - nh) / d;
                if (slope > talus) {
                    const float excess = (float)((slope - talus) * d);
                    delta = (float)((double)delta - ((double)excess / myTotal) * myTransfer);
                }
            }
        }
    }
    return (float)(h + (double)delta);
}

WO_HD inline float thermal_apply_cell(const Fields& F, int32_t c, double talus, double kThermal, double* inShare, int32_t* inRank, int stride,
                                      double* outShare = nullptr) {
    return thermal_apply_cell_t(F, c, talus, kThermal, inShare, inRank, stride, outShare, [&](int32_t x) { return F.me[x]; });
}

// ------------------------------------------------------------------------------------------------
// Glacial (js/terrain-post.js:410-433, 475-557)
// ------------------------------------------------------------------------------------------------
WO_HD inline double smoothstep_js(double x, double e0, double e1) {
    double t = (x - e0) / (e1 - e0);
    if (!(t < 1)) t = (t != t) ? t : 1;
    if (!(t > 0)) t = (t != t) ? t : 0;
    return t * t * (3 - 2 * t);
}

WO_HD inline float glac_index_cell(const Fields& F, int32_t r, double glacialStrength) {
    if (F.ocean[r]) return 0.0f;
    const double PI_ = 3.141592653589793;
    const double thresholdLat = PI_ / 2 - glacialStrength * PI_ / 4.5;
    double y = F.xyz[3 * r + 1];
    if (y > 1) y = 1;
    if (y < -1) y = -1;
    const double polarDist = fabs(asin(y));
    const double latFactor = smoothstep_js(polarDist, thresholdLat, PI_ / 2);
    const double elevFactor = smoothstep_js(F.e[r], 0.5, 0.9);
    const double latScale = smoothstep_js(polarDist, PI_ / 8, PI_ / 3);
    const double a = latFactor, b = elevFactor * 0.3 * (0.3 + 0.7 * latScale);
    return (float)((a > b ? a : b) * glacialStrength);
}

// ice receivers (js/terrain-post.js:481-492); also resets the per-iteration state of the cell
WO_HD inline void ice_receiver_cell(const Fields& F, int32_t r) {
    int32_t bestNb = -1;
    if (!F.ocean[r] && F.glac[r] > 0) {
        const double h = F.e[r];
        double bestDrop = 0;
        for (int32_t j = F.off[r]; j < F.off[r + 1]; ++j) {
            const int32_t nb = F.adj[j];
            const double drop = h - (double)F.e[nb];
            if (drop > bestDrop) { bestDrop = drop; bestNb = nb; }
        }
    }
    F.iceTarget[r] = bestNb;
    F.doneAt[r] = WO_NOT_DONE;
}

// ice accumulation task for cell t (js/terrain-post.js:495-503): iceFlow[t] = glac[t] (+ donors in rank
// order, each add rounded to f32).  Ocean cells can be targets too.  Runs when every donor is done.
WO_HD inline bool ice_accumulate_task(const Fields& F, int32_t t, int32_t round) {
    int32_t dn[WO_MAX_DEG]; int nd = 0;
    for (int32_t j = F.off[t]; j < F.off[t + 1]; ++j) {
        const int32_t n = F.adj[j];
        if (F.iceTarget[n] != t) continue;
        if (!(F.doneAt[n] < round)) return false;
        dn[nd++] = n;
    }
    float acc = F.glac[t];
    // donors in landCells order (ascending rank)
    int32_t last = -1;
    for (int k = 0; k < nd; ++k) {
        int pick = -1; int32_t pr = 0x7fffffff;
        for (int q = 0; q < nd; ++q) { const int32_t rk = F.rank[dn[q]]; if (rk > last && rk < pr) { pr = rk; pick = q; } }
        // a donor forwards only if its own iceFlow > 0 (always true: glac > 0 for cells with a target)
        const float df = F.iceFlow[dn[pick]];
        if (df > 0) acc = (float)((double)acc + (double)df);
        last = pr;
    }
    int up = 0;
    for (int k = 0; k < nd; ++k) if (F.iceFlow[dn[k]] > 0) ++up;
    F.iceFlow[t] = acc;
    F.iceUp[t] = (uint8_t)up;
    F.doneAt[t] = round;
    return true;
}

// carve task activation: active iff land and iceFlow > 0.1 (js/terrain-post.js:508)
WO_HD inline void carve_setup_cell(const Fields& F, int32_t r) {
    const bool active = !F.ocean[r] && ((double)F.iceFlow[r] > 0.1);
    F.arank[r] = active ? F.rank[r] : WO_NOT_DONE;
    F.doneAt[r] = WO_NOT_DONE;
    F.blocker[r] = -1;
}

// in-place carve of cell r (js/terrain-post.js:506-526).  Ready when no unfinished active cell within
// two hops has a lower rank.
// the cells a carve task must wait for: lower-ranked active cells within two hops, listed once per glacial step so
// that a round does not walk the two-hop neighbourhood again (a round lasts as long as its slowest task, and the
// nested adjacency walk is five dependent loads deep)
WO_HD inline void carve_deps_cell(const Fields& F, int32_t r, int32_t slot) {
    const int32_t myRank = F.arank[r];
    int32_t n = 0;
    int32_t* deps = F.carveDeps + (size_t)slot * WO_CARVE_DEPS;
    auto note = [&](int32_t c) {
        if (!(F.arank[c] < myRank)) return;
        for (int i = 0; i < n && i < WO_CARVE_DEPS; ++i) if (deps[i] == c) return;       // two-hop walks meet cells twice
        if (n < WO_CARVE_DEPS) deps[n] = c;
        ++n;
    };
    for (int32_t j = F.off[r]; j < F.off[r + 1]; ++j) {
        const int32_t nb = F.adj[j];
        note(nb);
        for (int32_t q = F.off[nb]; q < F.off[nb + 1]; ++q) { const int32_t m = F.adj[q]; if (m != r) note(m); }
    }
    F.carveDepCnt[slot] = n <= WO_CARVE_DEPS ? n : -1;
    F.carveDepPos[slot] = 0;
}

WO_HD inline bool carve_task(const Fields& F, int32_t r, int32_t round, double gCarveRate, double gConvergenceBonus,
                             double glacialStrength) {
    const int32_t myRank = F.arank[r];
    int32_t depCnt = -1, slot = -1;
    if (F.carveDeps) { slot = F.carveSlot[r]; depCnt = F.carveDepCnt[slot]; }
    if (depCnt >= 0) {
        // listed dependencies: resume at the first one that was still open last time
        const int32_t* deps = F.carveDeps + (size_t)slot * WO_CARVE_DEPS;
        // all of them must be looked at before the task may run: issue the loads together instead of one
        // dependent round trip per entry (this walk is on the critical path of the round)
        const int32_t pos = F.carveDepPos[slot];
        int32_t dv[WO_CARVE_DEPS], da[WO_CARVE_DEPS];
#pragma unroll
        for (int k = 0; k < WO_CARVE_DEPS; ++k) dv[k] = (k >= pos && k < depCnt) ? deps[k] : -1;
#pragma unroll
        for (int k = 0; k < WO_CARVE_DEPS; ++k) da[k] = dv[k] >= 0 ? F.doneAt[dv[k]] : -1;
        int32_t firstOpen = depCnt;
#pragma unroll
        for (int k = WO_CARVE_DEPS - 1; k >= 0; --k) if (dv[k] >= 0 && !(da[k] < round)) firstOpen = k;
        if (firstOpen < depCnt) { F.carveDepPos[slot] = firstOpen; return false; }
    } else {
        // cheap test first: the cell that blocked us at the last full scan (1-2 loads instead of ~40)
        const int32_t b0 = F.blocker[r];
        if (b0 >= 0 && !(F.doneAt[b0] < round)) return false;
        // full 2-hop scan; remember the unfinished blocker closest to us in rank (it tends to finish last)
        int32_t blk = -1, blkRank = -1;
        for (int32_t j = F.off[r]; j < F.off[r + 1]; ++j) {
            const int32_t n = F.adj[j];
            { const int32_t a = F.arank[n]; if (a < myRank && a > blkRank && !(F.doneAt[n] < round)) { blk = n; blkRank = a; } }
            for (int32_t q = F.off[n]; q < F.off[n + 1]; ++q) {
                const int32_t mcell = F.adj[q];
                if (mcell == r) continue;
                const int32_t a = F.arank[mcell];
                if (a < myRank && a > blkRank && !(F.doneAt[mcell] < round)) { blk = mcell; blkRank = a; }
            }
        }
        if (blk >= 0) { F.blocker[r] = blk; return false; }
    }
    const double fl = F.iceFlow[r];
    const double deepening = gCarveRate * pow(fl, 0.6) * glacialStrength;
    float er = (float)((double)F.e[r] - deepening);
    F.e[r] = er;
    // A round lasts as long as its slowest task.  The neighbours are distinct cells and none of them is r (checked at
    // planet creation), so their updates are independent: fetch every neighbour's inputs before the first store — the
    // compiler cannot prove that the stores do not alias the next iteration's loads and would serialise the round trips.
    const int32_t jb = F.off[r], deg = F.off[r + 1] - jb;
    if (deg <= 8) {
        int32_t nbs[8]; float en[8], dn[8]; uint8_t on[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { nbs[k] = k < deg ? F.adj[jb + k] : r; dn[k] = k < deg ? F.dist[jb + k] : 1.0f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { on[k] = F.ocean[nbs[k]]; en[k] = F.e[nbs[k]]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k >= deg || on[k]) continue;
            const double d = nd_or_eps(dn[k]);
            const double slope = fabs((double)er - (double)en[k]) / d;
            double f = 1 - slope;
            if (!(f > 0)) f = (f != f) ? f : 0;
            F.e[nbs[k]] = (float)((double)en[k] - deepening * 0.4 * f);
        }
    } else {
        for (int32_t j = jb; j < jb + deg; ++j) {
            const int32_t nb = F.adj[j];
            if (F.ocean[nb]) continue;
            const double d = nd_or_eps(F.dist[j]);
            const double slope = fabs((double)er - (double)F.e[nb]) / d;
            double f = 1 - slope;
            if (!(f > 0)) f = (f != f) ? f : 0;
            F.e[nb] = (float)((double)F.e[nb] - deepening * 0.4 * f);
        }
    }
    if (F.iceUp[r] >= 2) F.e[r] = (float)((double)F.e[r] - gConvergenceBonus * pow(fl, 0.4));
    F.doneAt[r] = round;
    return true;
}

// The same turn with every load issued before anything is decided: the dependency list, the row, the distances and the
// task's own inputs first, then the dependencies' round tags together with the neighbours' heights — two levels of dependent
// loads in all.  A carve round lasts as long as its slowest thread's chain of loads (a round's arithmetic is nothing), and the
// form above walks list -> tags -> row -> neighbours one after the other.  Values read for a task that turns out not to be
// ready are dropped; values read for a ready task are current, because everything it depends on finished in an earlier
// launch.  slot: the task's position in the activation list.  Tasks with more dependencies than fit the list, or a row
// longer than WO_EAGER_ROW, take the form above.
constexpr int WO_EAGER_ROW = 12;         // longest row the eager form keeps in registers
WO_HD inline bool carve_task_eager(const Fields& F, int32_t r, int32_t slot, int32_t round, double gCarveRate, double gConvergenceBonus,
                                   double glacialStrength) {
    const int32_t depCnt = F.carveDepCnt[slot];
    const int32_t jb = F.off[r], deg = F.off[r + 1] - jb;
    if (depCnt < 0 || deg > WO_EAGER_ROW) return carve_task(F, r, round, gCarveRate, gConvergenceBonus, glacialStrength);
    const int32_t* deps = F.carveDeps + (size_t)slot * WO_CARVE_DEPS;
    int32_t dv[WO_CARVE_DEPS], da[WO_CARVE_DEPS];
#pragma unroll
    for (int k = 0; k < WO_CARVE_DEPS; ++k) dv[k] = deps[k];
    int32_t nbs[WO_EAGER_ROW]; float en[WO_EAGER_ROW], dn[WO_EAGER_ROW]; uint8_t on[WO_EAGER_ROW];
#pragma unroll
    for (int k = 0; k < WO_EAGER_ROW; ++k) { nbs[k] = k < deg ? F.adj[jb + k] : r; dn[k] = k < deg ? F.dist[jb + k] : 1.0f; }
    const double fl = F.iceFlow[r];
    const uint8_t up = F.iceUp[r];
    const float e0 = F.e[r];
#pragma unroll
    for (int k = 0; k < WO_CARVE_DEPS; ++k) da[k] = k < depCnt ? F.doneAt[dv[k]] : -1;
#pragma unroll
    for (int k = 0; k < WO_EAGER_ROW; ++k) { on[k] = F.ocean[nbs[k]]; en[k] = F.e[nbs[k]]; }
    bool ready = true;
#pragma unroll
    for (int k = 0; k < WO_CARVE_DEPS; ++k) if (k < depCnt && !(da[k] < round)) ready = false;
    if (!ready) return false;
    const double deepening = gCarveRate * pow(fl, 0.6) * glacialStrength;
    float er = (float)((double)e0 - deepening);
#pragma unroll
    for (int k = 0; k < WO_EAGER_ROW; ++k) {
        if (k >= deg || on[k]) continue;
        const double d = nd_or_eps(dn[k]);
        const double slope = fabs((double)er - (double)en[k]) / d;
        double f = 1 - slope;
        if (!(f > 0)) f = (f != f) ? f : 0;
        F.e[nbs[k]] = (float)((double)en[k] - deepening * 0.4 * f);
    }
    if (up >= 2) er = (float)((double)er - gConvergenceBonus * pow(fl, 0.4));
    F.e[r] = er;
    F.doneAt[r] = round;
    return true;
}

// Everything a carve turn reads that does not change during the glacial step's rounds, in one 224-byte record per active
// task (built once per step next to the dependency list): a round then is record -> {dependencies' tags, heights} -> stores.
struct alignas(16) CarveRec { int32_t r, depCnt, deg; float fl; int32_t up, pad_[3]; double deepening, bonus; int32_t nbs[WO_EAGER_ROW]; float dist[WO_EAGER_ROW]; int32_t deps[WO_CARVE_DEPS]; };
static_assert(sizeof(CarveRec) == 240, "CarveRec layout");
// k_carve_granules (kernels_impl.h): the tag each granule a task reads must carry before its turn — tag[k]: neighbour k (-1: past the
// row's end or ocean, not read), tag[WO_EAGER_ROW]: the cell itself; myTag: what the task's own stores carry
struct alignas(16) CarveExpect { int32_t tag[WO_EAGER_ROW + 1]; int32_t myTag; int32_t pad_[2]; };
static_assert(sizeof(CarveExpect) == 64, "CarveExpect layout");
// deepening / bonus: the two pow() terms of the turn (js/terrain-post.js:510,522) — functions of the task's ice flow alone, so they
// are worked out here, once, instead of on the critical path of a round (a round lasts as long as its slowest task, and a
// double-precision pow is a few hundred dependent instructions)
// withDeps: 2 = dependency list in descending rank (k_carve_flow), 1 = as listed (the rounds), 0 = none (k_carve_granules does not read it; should
// that launch leave tasks to the rounds, the records are made again with their lists)
WO_HD inline void carve_record_cell(const Fields& F, int32_t r, int32_t slot, CarveRec* recs, double gCarveRate, double gConvergenceBonus, double glacialStrength, int withDeps = 2) {
    CarveRec R;
    const int32_t jb = F.off[r], deg = F.off[r + 1] - jb;
    R.r = r; R.depCnt = withDeps ? F.carveDepCnt[slot] : 0; R.deg = deg; R.fl = F.iceFlow[r]; R.up = F.iceUp[r]; R.pad_[0] = R.pad_[1] = R.pad_[2] = 0;
    { const double fl = R.fl; R.deepening = gCarveRate * pow(fl, 0.6) * glacialStrength; R.bonus = gConvergenceBonus * pow(fl, 0.4); }
    for (int k = 0; k < WO_EAGER_ROW; ++k) { R.nbs[k] = k < deg ? F.adj[jb + k] : r; R.dist[k] = k < deg ? F.dist[jb + k] : 1.0f; }
    for (int k = 0; k < WO_CARVE_DEPS; ++k) R.deps[k] = (R.depCnt >= 0 && k < R.depCnt) ? F.carveDeps[(size_t)slot * WO_CARVE_DEPS + k] : r;
    // highest rank first: the dependency closest to the task in rank tends to finish last, and the one-launch carve (k_carve_flow)
    // watches the first open entry instead of polling all of them (the order of the list means nothing to the rounds)
    for (int a = 1; withDeps == 2 && a < R.depCnt; ++a) {
        const int32_t c = R.deps[a], rc = F.arank[c];
        int b = a - 1;
        while (b >= 0 && F.arank[R.deps[b]] < rc) { R.deps[b + 1] = R.deps[b]; --b; }
        R.deps[b + 1] = c;
    }
    recs[slot] = R;
}
// the turn from its record (same arithmetic as carve_task / carve_task_eager); records with depCnt < 0 or a row longer than
// WO_EAGER_ROW take carve_task_eager
WO_HD inline bool carve_task_rec(const Fields& F, const CarveRec& R, int32_t slot, int32_t round, double gCarveRate, double gConvergenceBonus, double glacialStrength) {
    if (R.depCnt < 0 || R.deg > WO_EAGER_ROW) return carve_task_eager(F, R.r, slot, round, gCarveRate, gConvergenceBonus, glacialStrength);
    int32_t da[WO_CARVE_DEPS]; float en[WO_EAGER_ROW]; uint8_t on[WO_EAGER_ROW];
#pragma unroll
    for (int k = 0; k < WO_CARVE_DEPS; ++k) da[k] = k < R.depCnt ? F.doneAt[R.deps[k]] : -1;
    const float e0 = F.e[R.r];
#pragma unroll
    for (int k = 0; k < WO_EAGER_ROW; ++k) { on[k] = F.ocean[R.nbs[k]]; en[k] = F.e[R.nbs[k]]; }
    bool ready = true;
#pragma unroll
    for (int k = 0; k < WO_CARVE_DEPS; ++k) if (k < R.depCnt && !(da[k] < round)) ready = false;
    if (!ready) return false;
    const double deepening = R.deepening;
    float er = (float)((double)e0 - deepening);
#pragma unroll
    for (int k = 0; k < WO_EAGER_ROW; ++k) {
        if (k >= R.deg || on[k]) continue;
        const double d = nd_or_eps(R.dist[k]);
        const double slope = fabs((double)er - (double)en[k]) / d;
        double f = 1 - slope;
        if (!(f > 0)) f = (f != f) ? f : 0;
        F.e[R.nbs[k]] = (float)((double)en[k] - deepening * 0.4 * f);
    }
    if (R.up >= 2) er = (float)((double)er - R.bonus);
    F.e[R.r] = er;
    F.doneAt[R.r] = round;
    return true;
}


// moraine (js/terrain-post.js:529-537) gathered per target, then fjord (540-551) and clamp (554-556)
WO_HD inline void moraine_fjord_cell(const Fields& F, int32_t t, double gDepositAmount, double gFjordCarve) {
    if (F.ocean[t]) return;
    float e = F.e[t];
    int32_t last = -1;
    for (;;) {
        int32_t pick = -1, pr = 0x7fffffff;
        for (int32_t j = F.off[t]; j < F.off[t + 1]; ++j) {
            const int32_t n = F.adj[j];
            if (F.iceTarget[n] != t) continue;
            if (!((double)F.iceFlow[n] > 0.1)) continue;
            if (!((double)F.glac[t] < (double)F.glac[n] * 0.3)) continue;
            const int32_t rk = F.rank[n];
            if (rk > last && rk < pr) { pr = rk; pick = n; }
        }
        if (pick < 0) break;
        e = (float)((double)e + gDepositAmount * pow((double)F.iceFlow[pick], 0.3));
        last = pr;
    }
    if ((double)F.glac[t] > 0.2 && (double)F.iceFlow[t] > 0.5 && F.coast[t]) {
        e = (float)((double)e - gFjordCarve * pow((double)F.iceFlow[t], 0.5));
        if (e < 0) e = 0;
    }
    if (e < 0) e = 0;
    F.e[t] = e;
}

}  // namespace wo
