// Basin-local implicit solve (reference pass: js/terrain-post.js:614-641; dataflow form of a turn: erode_ops.h).
//
// A turn of the solve touches the cell r, its receiver t (deposit) and reads t's receiver t2; its <=3 predecessor events
// are turns of r's own donors / siblings / t's siblings (erode_ops.h: latest_event_before).  Every one of those cells is
// joined to r by land-to-land receiver edges, so two cells in different connected components of the receiver graph
// (edges r -> drainTarget[r], land targets only, direction ignored) never exchange a value: a DRAINAGE COMPONENT can be
// solved start to finish by one workgroup without waiting for anybody else.  Measured on the bench planet (10 M cells,
// research/basin_schedule.*): 128 572 components right after the first flood (largest 21 176 cells, 52 % of the land in
// components of <= 1 024 cells); after 20 iterations 133 006 components, largest 6 023 cells, 81 % in components of
// <= 1 024 — pits re-form between the two floods and cut the forest into small pieces.
//
// So, every pass:
//   1. k_basin_init / k_basin_keys: the root of every land cell's component by pointer jumping on the receiver array
//      (in place, asynchronous: whatever a thread reads is an ancestor), 2-cycles of mutually draining cells cut at the
//      cell with the smaller Morton slot.  key = root's Morton slot >> shift (a GROUP = the components whose roots fall
//      in the same 2^shift Morton slots: a union of components is closed under the dependencies just the same, and
//      16 key bits mean two radix passes instead of three).
//   2. a stable radix sort of the land list, taken in PROCESSING order (descending rank), by key: group-major, each
//      group in processing order — a topological order of the group's dependency DAG.  The position in that list is
//      the task's store index for this pass (Fields::slotOf).
//   3. k_solve_setup as before (records at the store index), then ONE launch of k_solve_basin: workgroup k owns the
//      groups that start in [k*T, (k+1)*T) and walks them in windows of WO_PATCH slots; inside a window the chains run
//      through LDS granules exactly like a visit of k_solve_patch, across windows through the workgroup's own earlier
//      writes.  No polling cap, no settle step, no external granules.
// The schedule is only a schedule: tasks are single-assignment, so any dependency-respecting order gives the same bits,
// and a task whose predecessor is NOT where the layout promised (a cycle longer than two cells that the jumping gave
// up on) simply stays pending and is finished by k_solve_patch launches over the same store order (planet.hip).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <string>

#include "device.h"

namespace wo {

namespace {

constexpr int WO_BASIN_CHASE_CAP = 1 << 16;         // pointer-jumping steps of one thread before it gives up (never reached: chains and rings are shorter)

// J[s] = Morton slot of the receiver of the cell at Morton slot s, or s itself for a root: no land receiver, or the
// lower-slot cell of a pair draining into each other
__global__ __launch_bounds__(WO_BLOCK) void k_basin_init(Fields F, const int32_t* __restrict__ slotCell, const int32_t* __restrict__ mslot, int32_t* __restrict__ J, int32_t L) {
    for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < L; s += gridDim.x * blockDim.x) {
        const int32_t c = slotCell[s];
        const TargetRank trc = F.tr[c];
        int32_t j = s;
        if (trc.target >= 0) {
            const TargetRank trt = F.tr[trc.target];
            if (trt.rank >= 0) {                                   // ocean cells carry rank -1
                const int32_t st = mslot[trc.target];
                if (!(trt.target == c && st > s)) j = st;
            }
        }
        J[s] = j;
    }
}

// Root of the component of the cell at Morton slot s.  Every value ever stored in J[x] is an ancestor of x (or, on a ring
// of cells draining into each other — possible on flats, where "least ascent" is zero — another cell of the ring), so
// concurrent chases only help each other.  A ring is found by Brent's checkpoints and cut at its smallest slot, which every
// thread that meets the ring computes alike.
__device__ inline int32_t basin_root(int32_t* J, int32_t s) {
    auto ld = [&](int32_t x) { return __atomic_load_n(&J[x], __ATOMIC_RELAXED); };
    int32_t j = ld(s), tort = j;
    int power = 1, lam = 0;
    for (int it = 0; it < WO_BASIN_CHASE_CAP; ++it) {
        const int32_t jj = ld(j);
        if (jj == j) return j;
        if (jj == tort) {
            int32_t m = jj, y = ld(jj);
            for (int k = 0; k < WO_BASIN_CHASE_CAP && y != jj; ++k) {
                const int32_t yy = ld(y);
                if (yy == y) return y;                              // somebody cut the ring meanwhile
                m = y < m ? y : m;
                y = yy;
            }
            __atomic_store_n(&J[m], m, __ATOMIC_RELAXED);
            return m;
        }
        if (++lam == power) { tort = jj; power <<= 1; lam = 0; }
        j = jj;
    }
    return j;
}

// thread s: the cell at Morton slot s (neighbouring slots are neighbouring cells, so the chases of a wave share their lines);
// leaves the group key of the cell where the next kernel finds it with one gather
__global__ __launch_bounds__(WO_BLOCK) void k_basin_jump(const int32_t* __restrict__ slotCell, int32_t* J, int32_t L, int32_t shift, uint32_t* __restrict__ keyOfCell) {
    for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < L; s += gridDim.x * blockDim.x) {
        const int32_t j = basin_root(J, s);
        __atomic_store_n(&J[s], j, __ATOMIC_RELAXED);
        keyOfCell[slotCell[s]] = (uint32_t)j >> shift;
    }
}
// thread i: the i-th cell in processing order (largest rank first)
__global__ __launch_bounds__(WO_BLOCK) void k_basin_keys(const int32_t* __restrict__ land, const uint32_t* __restrict__ keyOfCell, int32_t L,
                                                          uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) {
        const int32_t c = land[L - 1 - i];
        keys[i] = keyOfCell[c];
        vals[i] = c;
    }
}

// slotOf[cell] = position in the group-major list; rangeStart[k] = first position >= k*T that starts a group (0x7f7f7f7f: none)
__global__ __launch_bounds__(WO_BLOCK) void k_basin_slots(const int32_t* __restrict__ order, const uint32_t* __restrict__ keys, int32_t* __restrict__ slotOf, int32_t L,
                                                           int32_t* rangeStart, int32_t rangeT) {
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < L; q += gridDim.x * blockDim.x) {
        slotOf[order[q]] = q;
        if (q == 0 || keys[q] != keys[q - 1]) atomicMin(&rangeStart[q / rangeT], q);
    }
}
constexpr int32_t WO_RANGE_NONE = 0x7f7f7f7f;

// One window = W consecutive store slots of the workgroup's range, one thread per slot.  Every task of a window is runnable
// (its predecessors are in the window or in an earlier window of the same workgroup), so the question is only in which
// order the threads take them.  One task per thread with every thread polling its predecessors (the first form of this
// kernel, 420 us per launch at 10 M cells) spends its time in divergence: the tasks of a wave sit at ~20 different depths of
// the window's dependency DAG, so the wave runs the expensive part of a turn (three f64 divisions) ~20 times with a few
// lanes each, and eight such waves share a SIMD.  So a window is run in two steps:
//   1. levels.  level(task) = 1 + max level of its in-window predecessors, by polling 4-byte words in LDS (integer work
//      only); the tasks are counting-sorted by level (one LDS atomic per wave and level) and thread i takes the i-th task
//      of that order: its record goes through LDS, external predecessor values already filled in.
//   2. turns, level by level with a workgroup barrier in between.  The tasks of a level are neighbours in the thread
//      order, so a level costs one execution of the turn on the waves that hold it (the shallow levels hold hundreds of
//      tasks, the deep ones a handful) and nobody polls.  The results go to global memory after the loop (a store inside
//      it would make every barrier wait for the write to land).
constexpr int32_t WO_LEV_BLOCKED = -1;
struct alignas(8) BasinRec { double factor; float er, et, et2, cellDistT; int16_t w0, w1, w2, t; int16_t lev; uint16_t flags; int32_t pad; };
static_assert(sizeof(BasinRec) == 40, "40-byte hand-over record");

template <int W>
__global__ __launch_bounds__(W, 8) void k_solve_basin(Fields F, int32_t L, const int32_t* __restrict__ rangeStart, int32_t nRanges, int32_t launchTag,
                                                                         int32_t* patchPending, int32_t* totalPending, unsigned long long* dbg) {
    __shared__ unsigned long long s_out[2 * W];
    __shared__ BasinRec s_rec[W];
    __shared__ int32_t s_lev[W];
    __shared__ int32_t s_cnt[W + 2];                   // tasks per level, then first thread of each level
    __shared__ int32_t s_wsum[W / 64];
    __shared__ int32_t s_maxLev, s_blocked;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // dbg (diagnostic, may be null): shader clocks of thread 0 per phase, summed over the workgroups: [1] records + levels,
    // [2] counting sort + hand-over, [4] turns, [5] store; [6] windows, [7] levels summed, [8] runnable tasks
    long long tc = dbg ? clock64() : 0;
    auto lap = [&](int k) { if (dbg && tid == 0) { const long long now = clock64(); atomicAdd(&dbg[k], (unsigned long long)(now - tc)); tc = now; } };
    const int32_t S = rangeStart[blockIdx.x];
    if (S == WO_RANGE_NONE) return;                                 // no group starts in this stretch
    const long long tStart = dbg ? clock64() : 0;
    const unsigned long long wStart = dbg ? wall_clock64() : 0;
    int32_t nWin = 0;
    int32_t E = L;
    for (int32_t j = blockIdx.x + 1; j < nRanges; ++j) { const int32_t v = rangeStart[j]; if (v != WO_RANGE_NONE) { E = v; break; } }
    const Granule* G = reinterpret_cast<const Granule*>(F.out);
    volatile unsigned long long* vs = s_out;
    volatile int32_t* vlev = s_lev;
    auto pack = [](Granule g) { return (unsigned long long)__float_as_uint(g.v) | ((unsigned long long)(uint32_t)g.tag << 32); };
    for (int32_t base = S; base < E; base += W) {
        const int32_t wHi = E < base + W ? E : base + W;
        const bool mine = base + tid < wHi;
        // granule -> its word in s_out, -1 when the producer is not in this window
        auto word_of = [&](int32_t g) { const int32_t sq = g >> 1; return (g >= 0 && sq >= base && sq < wHi) ? g - 2 * base : -1; };
        // ---- 1. record, external predecessors, level
        SolveTask T;
        double er = 0, et = 0, et2 = 0;
        int32_t myLev = 0, w0 = -1, w1 = -1, w2 = -1;
        if (mine) {
            T = F.task[base + tid];
            er = T.e0r; et = T.e0t; et2 = T.e0t2;
            w0 = word_of(T.predSelf); w1 = word_of(T.predT); w2 = word_of(T.predT2);
            // a predecessor outside the window: an earlier window of this workgroup (there by now, whatever its tag) or — only
            // when the layout is off — somebody else's task, which counts when an earlier launch produced it
            bool blocked = false;
            auto ext = [&](int32_t g, int32_t w, double& v) {
                if (g < 0 || w >= 0) return;
                const Granule gq = G[g];
                const int32_t sq = g >> 1;
                const bool own = sq >= S && sq < base;
                if (gq.tag == 0 || (!own && gq.tag >= launchTag)) { blocked = true; return; }
                v = gq.v;
            };
            ext(T.predSelf, w0, er); ext(T.predT, w1, et); ext(T.predT2, w2, et2);
            myLev = blocked ? WO_LEV_BLOCKED : ((w0 < 0 && w1 < 0 && w2 < 0) ? 1 : 0);          // 0: not known yet
        }
        s_lev[tid] = myLev;
        s_cnt[tid] = 0;
        if (tid == 0) { s_cnt[W] = 0; s_cnt[W + 1] = 0; s_maxLev = 0; s_blocked = 0; }
        __syncthreads();
        while (__any(mine && myLev == 0)) {
            if (mine && myLev == 0) {
                // a level word is 0 until known; an absent predecessor counts as known, level 0
                const int32_t a = w0 >= 0 ? vlev[w0 >> 1] : 0x40000000, b = w1 >= 0 ? vlev[w1 >> 1] : 0x40000000, c = w2 >= 0 ? vlev[w2 >> 1] : 0x40000000;
                if (a < 0 || b < 0 || c < 0) myLev = WO_LEV_BLOCKED;
                else if (a > 0 && b > 0 && c > 0) {
                    const int32_t x = a & 0x3fffffff, y = b & 0x3fffffff, z = c & 0x3fffffff;
                    const int32_t m = x > y ? x : y;
                    myLev = 1 + (m > z ? m : z);
                }
                if (myLev != 0) vlev[tid] = myLev;
            }
        }
        lap(1);
        // ---- counting sort by level: one LDS atomic per wave and level (half of a window sits on level 1: one atomic per task
        // was 25 k clocks of serialised adds on a few words)
        const bool runnable = mine && myLev > 0;
        int32_t inLevel = 0;                                         // my index among the wave's tasks of my level
        {
            unsigned long long todo = __ballot(runnable);
            while (todo) {
                const int32_t l = __shfl(myLev, (int)__builtin_ctzll(todo));
                const unsigned long long m = __ballot(runnable && myLev == l);
                if (runnable && myLev == l) inLevel = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == (int)__builtin_ctzll(m)) atomicAdd(&s_cnt[l], __popcll(m));
                todo &= ~m;
            }
            const unsigned long long bm = __ballot(mine && myLev < 0);
            if (bm && lane == 0) atomicAdd(&s_blocked, __popcll(bm));
            int32_t mx = runnable ? myLev : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int32_t v = __shfl_xor(mx, o); mx = v > mx ? v : mx; }
            if (lane == 0 && mx) atomicMax(&s_maxLev, mx);
        }
        __syncthreads();
        const int32_t maxLev = s_maxLev;
        {   // exclusive scan of s_cnt[1 .. W] (thread t owns level t + 1)
            const int32_t c = s_cnt[tid + 1];
            int32_t incl = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int32_t v = __shfl_up(incl, o); if (lane >= o) incl += v; }
            if (lane == 63) s_wsum[wave] = incl;
            __syncthreads();
            if (tid < W / 64) {
                const int32_t w = s_wsum[tid];
                int32_t wi = w;
#pragma unroll
                for (int o = 1; o < W / 64; o <<= 1) { const int32_t v = __shfl_up(wi, o); if (tid >= o) wi += v; }
                s_wsum[tid] = wi - w;
            }
            __syncthreads();
            s_cnt[tid + 1] = s_wsum[wave] + incl - c;               // first thread of level tid + 1
        }
        vs[2 * tid] = 0; vs[2 * tid + 1] = 0;
        __syncthreads();
        {   // positions: the wave's tasks of a level take consecutive threads; s_cnt[l] ends up as the first thread of level l + 1
            unsigned long long todo = __ballot(runnable);
            int32_t pos = -1;
            while (todo) {
                const int src = (int)__builtin_ctzll(todo);
                const int32_t l = __shfl(myLev, src);
                const unsigned long long m = __ballot(runnable && myLev == l);
                int32_t b0 = 0;
                if (lane == src) b0 = atomicAdd(&s_cnt[l], __popcll(m));
                b0 = __shfl(b0, src);
                if (runnable && myLev == l) pos = b0 + inLevel;
                todo &= ~m;
            }
            if (runnable) {
                BasinRec R;
                R.factor = T.factor; R.er = (float)er; R.et = (float)et; R.et2 = (float)et2; R.cellDistT = T.cellDistT;
                R.w0 = (int16_t)w0; R.w1 = (int16_t)w1; R.w2 = (int16_t)w2; R.t = (int16_t)tid; R.lev = (int16_t)myLev; R.flags = (uint16_t)T.flags; R.pad = 0;
                s_rec[pos] = R;
            }
        }
        __syncthreads();
        const int32_t total = maxLev > 0 ? s_cnt[maxLev] : 0;
        lap(2);
        if (dbg && tid == 0) { atomicAdd(&dbg[6], 1ull); atomicAdd(&dbg[7], (unsigned long long)maxLev); atomicAdd(&dbg[8], (unsigned long long)total); }
        // ---- 2. turns
        SolvePrepared pre;
        int32_t t = 0, lev = 0;
        if (tid < total) {
            const BasinRec R = s_rec[tid];
            T.factor = R.factor; T.cellDistT = R.cellDistT; T.flags = R.flags;
            er = R.er; et = R.et; et2 = R.et2; w0 = R.w0; w1 = R.w1; w2 = R.w2; t = R.t; lev = R.lev;
            pre = solve_prepare(T, F.solveK, F.solveM, F.solveDt);
        }
        SolveOut o; o.self.v = 0; o.self.tag = 0; o.dep.v = 0; o.dep.tag = 0;
        for (int32_t k = 1; k <= maxLev; ++k) {
            if (lev == k) {
                if (w0 >= 0) er = __uint_as_float((uint32_t)vs[w0]);
                if (w1 >= 0) et = __uint_as_float((uint32_t)vs[w1]);
                if (w2 >= 0) et2 = __uint_as_float((uint32_t)vs[w2]);
                o = solve_apply(T, pre, er, et, et2, launchTag);
                vs[2 * t] = pack(o.self); vs[2 * t + 1] = pack(o.dep);
            }
            __syncthreads();
        }
        lap(4);
        if (tid < total) F.out[base + t] = o;
        if (mine && myLev < 0) { atomicAdd(&patchPending[(base + tid) / WO_PATCH], 1); }
        if (tid == 0 && s_blocked) atomicAdd(totalPending, s_blocked);
        __threadfence_block();
        __syncthreads();
        lap(5);
        ++nWin;
    }
    // [9] longest workgroup (clocks), [10] most windows of a workgroup, [11] workgroup clocks summed, [12] workgroups that ran,
    // [13] / [14] first start / last end on the 100 MHz wall clock, [15] windows of the longest workgroup (approximate: last writer)
    if (dbg && tid == 0) {
        const unsigned long long d = (unsigned long long)(clock64() - tStart);
        const unsigned long long old = atomicMax(&dbg[9], d);
        if (d > old) dbg[15] = (unsigned long long)nWin;
        atomicMax(&dbg[10], (unsigned long long)nWin); atomicAdd(&dbg[11], d); atomicAdd(&dbg[12], 1ull);
        atomicMin(&dbg[13], wStart); atomicMax(&dbg[14], wall_clock64());
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Streaming form: ONE WAVE per range, no barrier and no polling.  The store order of a group is its processing order, a
// topological order of its dependency DAG, so a wave can simply walk its range front to back in chunks of 64 tasks: every
// predecessor of a chunk's task is either
//   * further back than the ring (global memory: written by this wave long ago),
//   * in one of the last WO_RING - 64 slots (the wave's LDS ring of {value, tag} granules), or
//   * in the chunk itself, a lower lane: the chunk takes as many passes as its longest in-chunk chain (2-3), each pass running
//     the lanes whose in-chunk predecessors are done (a ballot tells), values through the ring.
// The windowed form above keeps a 16-wave workgroup on a CU for ~90 k clocks per 1 024 tasks, nearly all of it waiting at
// barriers or polling (2 workgroups per CU); here a CU carries 32 independent waves that never wait for each other, and the
// next chunk's records are in flight while the current one computes.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WO_RING = 256;                       // tasks whose granules a wave keeps in LDS (power of two, >= 256)

__global__ __launch_bounds__(64, 4) void k_solve_stream(Fields F, int32_t L, const int32_t* __restrict__ rangeStart, int32_t nRanges, int32_t launchTag,
                                                         int32_t* patchPending, int32_t* totalPending) {
    __shared__ unsigned long long s_ring[2 * WO_RING];
    const int lane = threadIdx.x;
    const int32_t S = rangeStart[blockIdx.x];
    if (S == WO_RANGE_NONE) return;                                 // no group starts in this stretch
    int32_t E = L;
    for (int32_t j = blockIdx.x + 1; j < nRanges; ++j) { const int32_t v = rangeStart[j]; if (v != WO_RANGE_NONE) { E = v; break; } }
    const unsigned long long* G = reinterpret_cast<const unsigned long long*>(F.out);      // granule = {value, tag} in one 8-byte word
    unsigned long long* ring = s_ring;           // one wave: its LDS accesses execute in program order, no volatile (which would drain the loads in flight)
    const unsigned long long BLOCKED = 0xffffffff00000000ull;       // tag -1
    auto pack = [](Granule g) { return (unsigned long long)__float_as_uint(g.v) | ((unsigned long long)(uint32_t)g.tag << 32); };
    // Software pipeline, all loads unconditional (clamped indices) so that they stay in flight across the chunk's work:
    // records two chunks ahead; the predecessors of the NEXT chunk that lie further back than its ring window (or outside the
    // range) one chunk ahead — those turns ended at least a whole chunk ago.
    auto record = [&](int32_t q) { return F.task[q < L ? q : L - 1]; };
    auto far_index = [&](int32_t g, int32_t cbase) { const int32_t sq = g >> 1; return (g >= 0 && (sq < S || sq < cbase - (WO_RING - 64))) ? g : 0; };
    SolveTask T1 = record(S + lane), T2 = record(S + 64 + lane);
    unsigned long long a0 = G[far_index(T1.predSelf, S)], a1 = G[far_index(T1.predT, S)], a2 = G[far_index(T1.predT2, S)];
    SolveOut oPrev; oPrev.self.v = 0; oPrev.self.tag = 0; oPrev.dep.v = 0; oPrev.dep.tag = 0;
    bool storePrev = false;
    for (int32_t base = S; base < E; base += 64) {
        const bool mine = base + lane < E;
        // the previous chunk's results go out now, a chunk late: the wait for this chunk's loads at the top of the loop would
        // otherwise also wait for a store issued a moment ago (whoever reads them from memory is at least two chunks behind)
        if (storePrev) F.out[base - 64 + lane] = oPrev;
        const SolveTask T = T1;
        T1 = T2;
        T2 = record(base + 128 + lane);
        const unsigned long long n0 = G[far_index(T1.predSelf, base + 64)], n1 = G[far_index(T1.predT, base + 64)], n2 = G[far_index(T1.predT2, base + 64)];
        double er = T.e0r, et = T.e0t, et2 = T.e0t2;
        unsigned long long predMask = 0;                                 // lanes of this chunk I wait for
        bool blocked = false;
        int32_t r0 = -1, r1 = -1, r2 = -1;                               // ring words of the predecessors that come through the ring
        if (mine) {
            auto classify = [&](int32_t g, unsigned long long far, double& v, int32_t& rw) {
                if (g < 0) return;
                const int32_t sq = g >> 1;
                if (sq >= base) {
                    if (sq >= base + 64 || sq - base >= lane) { blocked = true; return; }      // not in processing order: the layout is off
                    predMask |= 1ull << (sq - base);
                    rw = g & (2 * WO_RING - 1);
                } else if (sq >= S && sq >= base - (WO_RING - 64)) {
                    rw = g & (2 * WO_RING - 1);
                } else {
                    const int32_t tag = (int32_t)(far >> 32);
                    const bool own = sq >= S;                                                    // written by this wave, long ago
                    if (tag <= 0 || (!own && tag >= launchTag)) { blocked = true; return; }
                    v = __uint_as_float((uint32_t)far);
                }
            };
            classify(T.predSelf, a0, er, r0); classify(T.predT, a1, et, r1); classify(T.predT2, a2, et2, r2);
        }
        const SolvePrepared pre = solve_prepare(T, F.solveK, F.solveM, F.solveDt);
        bool done = !mine;
        SolveOut o; o.self.v = 0; o.self.tag = 0; o.dep.v = 0; o.dep.tag = 0;
        const int32_t myWord = (2 * (base + lane)) & (2 * WO_RING - 1);
        if (mine && blocked) { ring[myWord] = BLOCKED; ring[myWord + 1] = BLOCKED; done = true; }
        for (;;) {
            const unsigned long long dm = __ballot(done);
            if (dm == ~0ull) break;
            const bool go = !done && (predMask & ~dm) == 0;
            if (!__any(go)) {                                            // cannot happen (lower lanes only); give the rest up rather than spin
                if (!done) { blocked = true; ring[myWord] = BLOCKED; ring[myWord + 1] = BLOCKED; done = true; }
                continue;
            }
            if (go) {
                bool bad = false;
                auto rd = [&](int32_t rw, double& v) {
                    if (rw < 0) return;
                    const unsigned long long w = ring[rw];
                    if ((int32_t)(w >> 32) <= 0) { bad = true; return; }
                    v = __uint_as_float((uint32_t)w);
                };
                rd(r0, er); rd(r1, et); rd(r2, et2);
                if (bad) { blocked = true; ring[myWord] = BLOCKED; ring[myWord + 1] = BLOCKED; }
                else {
                    o = solve_apply(T, pre, er, et, et2, launchTag);
                    ring[myWord] = pack(o.self); ring[myWord + 1] = pack(o.dep);
                }
                done = true;
            }
        }
        oPrev = o; storePrev = mine && !blocked;
        if (mine && blocked) { atomicAdd(&patchPending[(base + lane) / WO_PATCH], 1); atomicAdd(totalPending, 1); }
        a0 = n0; a1 = n1; a2 = n2;
    }
    if (storePrev) F.out[S + ((E - S - 1) / 64) * 64 + lane] = oPrev;
}

// ---------------------------------------------------------------------------------------------------------------------
// Cooperative streaming form: NW waves per range.  A launch of k_solve_stream lasts as long as its longest range — one wave
// walking the biggest component alone, 100-330 chunks at ~2.4 us — while the chip is empty.  Here a workgroup of NW waves
// walks the range in super-chunks of 64 x NW tasks: every lane takes one task and polls the granules of its predecessors
// that lie in the workgroup's LDS ring (the current super-chunk included) until they carry the slot it waits for; the
// lowest open slot of a super-chunk always finds its predecessors done, so the polling ends.  One barrier per super-chunk
// keeps the waves together (ring entries are reused RING slots later).  Loads are pipelined as in k_solve_stream.
// ---------------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(64 * NW, 6) void k_solve_coop(Fields F, int32_t L, const int32_t* __restrict__ rangeStart, int32_t nRanges, int32_t launchTag,
                                                            int32_t* patchPending, int32_t* totalPending) {
    constexpr int SC = 64 * NW;                        // tasks per super-chunk
    constexpr int RING = 1024;                         // tasks whose granules the workgroup keeps in LDS
    __shared__ unsigned long long s_ring[2 * RING];
    const int tid = threadIdx.x;
    const int32_t S = rangeStart[blockIdx.x];
    if (S == WO_RANGE_NONE) return;
    int32_t E = L;
    for (int32_t j = blockIdx.x + 1; j < nRanges; ++j) { const int32_t v = rangeStart[j]; if (v != WO_RANGE_NONE) { E = v; break; } }
    const unsigned long long* G = reinterpret_cast<const unsigned long long*>(F.out);
    // LDS keeps what the last workgroup left: last pass's launch wrote the very tags this one waits for.  Clear the ring first.
    for (int i = tid; i < 2 * RING; i += SC) s_ring[i] = 0;
    __syncthreads();
    // ring word = {value, tag}: tag = slot + 1 once the granule is there, -(slot + 1) when its task is blocked; anything else: not yet
    auto ring_put = [&](int32_t word, float v, int32_t tag) {
        __hip_atomic_store(&s_ring[word], (unsigned long long)__float_as_uint(v) | ((unsigned long long)(uint32_t)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto record = [&](int32_t q) { return F.task[q < L ? q : L - 1]; };
    auto far_index = [&](int32_t g, int32_t cbase) { const int32_t sq = g >> 1; return (g >= 0 && (sq < S || sq < cbase - (RING - SC))) ? g : 0; };
    SolveTask T1 = record(S + tid), T2 = record(S + SC + tid);
    unsigned long long a0 = G[far_index(T1.predSelf, S)], a1 = G[far_index(T1.predT, S)], a2 = G[far_index(T1.predT2, S)];
    SolveOut oPrev; oPrev.self.v = 0; oPrev.self.tag = 0; oPrev.dep.v = 0; oPrev.dep.tag = 0;
    bool storePrev = false;
    for (int32_t base = S; base < E; base += SC) {
        const int32_t q = base + tid;
        const bool mine = q < E;
        if (storePrev) F.out[q - SC] = oPrev;                          // a super-chunk late (see k_solve_stream)
        const SolveTask T = T1;
        T1 = T2;
        T2 = record(base + 2 * SC + tid);
        const unsigned long long n0 = G[far_index(T1.predSelf, base + SC)], n1 = G[far_index(T1.predT, base + SC)], n2 = G[far_index(T1.predT2, base + SC)];
        double er = T.e0r, et = T.e0t, et2 = T.e0t2;
        bool blocked = false;
        int32_t r0 = -1, r1 = -1, r2 = -1, x0 = 0, x1 = 0, x2 = 0;     // ring word and expected tag of the predecessors that come through the ring
        if (mine) {
            auto classify = [&](int32_t g, unsigned long long far, double& v, int32_t& rw, int32_t& expect) {
                if (g < 0) return;
                const int32_t sq = g >> 1;
                if (sq >= q) { blocked = true; return; }                                         // not in processing order: the layout is off
                if (sq >= S && sq >= base - (RING - SC)) { rw = g & (2 * RING - 1); expect = sq + 1; return; }
                const int32_t tag = (int32_t)(far >> 32);
                const bool own = sq >= S;                                                        // written by this workgroup, long ago
                if (tag <= 0 || (!own && tag >= launchTag)) { blocked = true; return; }
                v = __uint_as_float((uint32_t)far);
            };
            classify(T.predSelf, a0, er, r0, x0); classify(T.predT, a1, et, r1, x1); classify(T.predT2, a2, et2, r2, x2);
        }
        const SolvePrepared pre = solve_prepare(T, F.solveK, F.solveM, F.solveDt);
        const int32_t myWord = (2 * q) & (2 * RING - 1);
        bool open = mine;
        SolveOut o; o.self.v = 0; o.self.tag = 0; o.dep.v = 0; o.dep.tag = 0;
        if (mine && blocked) { ring_put(myWord, 0.0f, -(q + 1)); ring_put(myWord + 1, 0.0f, -(q + 1)); open = false; }
        while (__any(open)) {
            if (open) {
                auto poll = [&](int32_t& rw, int32_t expect, double& v) {
                    if (rw < 0) return;
                    const unsigned long long w = __hip_atomic_load(&s_ring[rw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const int32_t tag = (int32_t)(w >> 32);
                    if (tag == expect) { v = __uint_as_float((uint32_t)w); rw = -1; }
                    else if (tag == -expect) { blocked = true; }
                };
                poll(r0, x0, er); poll(r1, x1, et); poll(r2, x2, et2);
                if (blocked) { ring_put(myWord, 0.0f, -(q + 1)); ring_put(myWord + 1, 0.0f, -(q + 1)); open = false; }
                else if (r0 < 0 && r1 < 0 && r2 < 0) {
                    o = solve_apply(T, pre, er, et, et2, launchTag);
                    ring_put(myWord, o.self.v, q + 1); ring_put(myWord + 1, o.dep.v, q + 1);
                    open = false;
                }
            }
        }
        oPrev = o; storePrev = mine && !blocked;
        if (mine && blocked) { atomicAdd(&patchPending[q / WO_PATCH], 1); atomicAdd(totalPending, 1); }
        a0 = n0; a1 = n1; a2 = n2;
        __syncthreads();
    }
    if (storePrev) F.out[S + ((E - S - 1) / SC) * SC + tid] = oPrev;
}

}  // namespace

// slots per workgroup range (a range = the groups that start in one stretch of this many slots)
static int basin_range() {
    static const int r = getenv("WO_BASIN_RANGE") ? std::max(64, (atoi(getenv("WO_BASIN_RANGE")) / 64) * 64) : 256;
    return r;
}

// Group-major store order for this pass (d_basinSlot) and the sorted group keys (d_keys[1]).  Call after the receivers
// pass (F.tr) and before k_solve_setup; everything is enqueued on the planet's stream, no host sync.
void basin_layout(wo_planet* p) {
    const int32_t N = p->N, L = p->L;
    hipStream_t s = cur_stream(p);
    if (!p->d_basinJ) {
        WO_HIP(hipMalloc((void**)&p->d_basinJ, (size_t)N * 4));
        WO_HIP(hipMalloc((void**)&p->d_basinSlot, (size_t)N * 4));
        WO_HIP(hipMalloc((void**)&p->d_basinKey, (size_t)N * 4));
        WO_HIP(hipMalloc((void**)&p->d_basinVals[0], (size_t)N * 4)); WO_HIP(hipMalloc((void**)&p->d_basinVals[1], (size_t)N * 4));   // own buffers: the layout runs beside the flow accumulation, whose rounds use the planet's lists
        WO_HIP(hipMalloc((void**)&p->d_basinRange, ((size_t)N / 64 + 4) * 4));
        WO_HIP(hipMemsetAsync(p->d_basinSlot, 0xff, (size_t)N * 4, s));
    }
    const Fields F = p->fields();
    int bitsL = 1;
    while (((int64_t)1 << bitsL) < (int64_t)L) ++bitsL;
    static const int keyBits = getenv("WO_BASIN_KEY_BITS") ? std::max(8, std::min(30, atoi(getenv("WO_BASIN_KEY_BITS")))) : 16;
    const int shift = bitsL > keyBits ? bitsL - keyBits : 0;
    const int grid = blocks_for(L, 1 << 16);
    launch(p, FAM_BASIN, k_basin_init, grid, WO_BLOCK, F, (const int32_t*)p->d_patchOrder, (const int32_t*)p->d_slotOf, p->d_basinJ, L);
    launch(p, FAM_BASIN, k_basin_jump, grid, WO_BLOCK, (const int32_t*)p->d_patchOrder, p->d_basinJ, L, (int32_t)shift, p->d_basinKey);
    launch(p, FAM_BASIN, k_basin_keys, grid, WO_BLOCK, (const int32_t*)p->d_land[p->landCur], (const uint32_t*)p->d_basinKey, L, p->d_keys[0], p->d_basinVals[0]);
    {
        hipEvent_t a = nullptr, b = nullptr;
        if (p->profiling) { a = profile_event(p); b = profile_event(p); WO_HIP(hipEventRecord(a, s)); }
        size_t bytes = p->sortTempBytes;
        WO_HIP(hipcub::DeviceRadixSort::SortPairs(p->d_sortTemp, bytes, (const uint32_t*)p->d_keys[0], p->d_keys[1],
                                                 (const int32_t*)p->d_basinVals[0], p->d_basinVals[1], L, 0, bitsL - shift, s));
        if (p->profiling) { WO_HIP(hipEventRecord(b, s)); p->pending.push_back({FAM_BASIN_SORT, a, b}); }
    }
    const int rangeT = basin_range();
    const int nRanges = (int)(((int64_t)L + rangeT - 1) / rangeT);
    WO_HIP(hipMemsetAsync(p->d_basinRange, 0x7f, (size_t)(nRanges + 1) * 4, s));
    launch(p, FAM_BASIN, k_basin_slots, blocks_for(L, 4096), WO_BLOCK, (const int32_t*)p->d_basinVals[1], (const uint32_t*)p->d_keys[1], p->d_basinSlot, L, p->d_basinRange, (int32_t)rangeT);
}

// the one launch of the pass; F.slotOf must be d_basinSlot, patchPending zeroed
void basin_solve_launch(wo_planet* p, const Fields& F, int32_t launchTag, int32_t* totalPending) {
    const int rangeT = basin_range();
    const int nRanges = (int)(((int64_t)p->L + rangeT - 1) / rangeT);
    // WO_BASIN_STATS=<n>: per-phase clocks of the n-th basin launch of the planet -> stderr (diagnostic)
    static const int statsAt = getenv("WO_BASIN_STATS") ? atoi(getenv("WO_BASIN_STATS")) : -1;
    static const int window = getenv("WO_BASIN_WINDOW") ? atoi(getenv("WO_BASIN_WINDOW")) : 512;
    unsigned long long* dbg = nullptr;
    if (statsAt >= 0 && p->basinLaunches == statsAt) { WO_HIP(hipMalloc((void**)&dbg, 16 * 8)); WO_HIP(hipMemsetAsync(dbg, 0, 16 * 8, p->ctx->stream)); WO_HIP(hipMemsetAsync(dbg + 13, 0xff, 8, p->ctx->stream)); }
    ++p->basinLaunches;
    static const bool stream = !(getenv("WO_BASIN_KERNEL") && std::string(getenv("WO_BASIN_KERNEL")) == "window");
    static const int coopWaves = getenv("WO_BASIN_WAVES") ? atoi(getenv("WO_BASIN_WAVES")) : 4;
    if (stream && coopWaves >= 4) launch(p, FAM_SOLVE_BASIN, k_solve_coop<4>, nRanges, 256, F, p->L, (const int32_t*)p->d_basinRange, (int32_t)nRanges, launchTag, p->d_patchPending, totalPending);
    else if (stream && coopWaves >= 2) launch(p, FAM_SOLVE_BASIN, k_solve_coop<2>, nRanges, 128, F, p->L, (const int32_t*)p->d_basinRange, (int32_t)nRanges, launchTag, p->d_patchPending, totalPending);
    else if (stream) launch(p, FAM_SOLVE_BASIN, k_solve_stream, nRanges, 64, F, p->L, (const int32_t*)p->d_basinRange, (int32_t)nRanges, launchTag, p->d_patchPending, totalPending);
    else if (window >= 1024) launch(p, FAM_SOLVE_BASIN, k_solve_basin<1024>, nRanges, 1024, F, p->L, (const int32_t*)p->d_basinRange, (int32_t)nRanges, launchTag, p->d_patchPending, totalPending, dbg);
    else if (window >= 512) launch(p, FAM_SOLVE_BASIN, k_solve_basin<512>, nRanges, 512, F, p->L, (const int32_t*)p->d_basinRange, (int32_t)nRanges, launchTag, p->d_patchPending, totalPending, dbg);
    else launch(p, FAM_SOLVE_BASIN, k_solve_basin<256>, nRanges, 256, F, p->L, (const int32_t*)p->d_basinRange, (int32_t)nRanges, launchTag, p->d_patchPending, totalPending, dbg);
    if (dbg) {
        unsigned long long h[16];
        WO_HIP(hipStreamSynchronize(p->ctx->stream));
        WO_HIP(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        WO_HIP(hipFree(dbg));
        const double nw = std::max<double>(1, (double)h[6]);
        fprintf(stderr, "basin stats (launch %d): window %d, workgroups %d, windows %llu, levels per window %.1f, runnable tasks %llu of %d; clocks per window: records+levels %.0f sort+handover %.0f turns %.0f store %.0f (total %.0f)\n",
                statsAt, window, nRanges, h[6], (double)h[7] / nw, h[8], p->L, h[1] / nw, h[2] / nw, h[4] / nw, h[5] / nw, (double)(h[1] + h[2] + h[4] + h[5]) / nw);
        fprintf(stderr, "basin stats (launch %d): workgroups that ran %llu, clocks summed %.3g, longest workgroup %llu clocks with %llu windows, most windows %llu, kernel span %.1f us (100 MHz wall clock)\n",
                statsAt, h[12], (double)h[11], h[9], h[15], h[10], (double)(h[14] - h[13]) / 100.0);
    }
}

void basin_free(wo_planet* p) {
    if (p->d_basinJ) (void)hipFree(p->d_basinJ);
    if (p->d_basinSlot) (void)hipFree(p->d_basinSlot);
    if (p->d_basinKey) (void)hipFree(p->d_basinKey);
    for (auto& v : p->d_basinVals) { if (v) (void)hipFree(v); v = nullptr; }
    if (p->d_basinRange) (void)hipFree(p->d_basinRange);
    p->d_basinJ = nullptr; p->d_basinSlot = nullptr; p->d_basinKey = nullptr; p->d_basinRange = nullptr;
}

}  // namespace wo
