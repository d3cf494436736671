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
//   1. k_receivers_flow_init (start state) / k_basin_jump / k_basin_keys: the root of every land cell's component by pointer jumping on the receiver array
//      (in place, asynchronous: whatever a thread reads is an ancestor), 2-cycles of mutually draining cells cut at the
//      cell with the smaller Morton slot.  key = root's Morton slot >> shift (a GROUP = the components whose roots fall
//      in the same 2^shift Morton slots: a union of components is closed under the dependencies just the same, and
//      16 key bits mean two radix passes instead of three).
//   2. a stable radix sort of the land list, taken in PROCESSING order (descending rank), by key: group-major, each
//      group in processing order — a topological order of the group's dependency DAG.  The position in that list is
//      the task's store index for this pass (Fields::slotOf).
//   3. k_solve_setup as before (records at the store index), then ONE launch: workgroup k owns the groups that start in
//      [k*T, (k+1)*T) and walks them front to back (k_solve_coop / k_solve_stream below).  No polling cap, no settle step,
//      no external granules.  (Round 3 also built and measured a windowed form — 1 024-slot windows, tasks counting-sorted by
//      dependency level in LDS, one barrier per level: 440-515 us per launch against 335 here, the workgroup holds its CU
//      for ~40 barrier-separated levels doing almost nothing; dropped, DESIGN.md section 5 and profiles/r03c..r03g.)
// The schedule is only a schedule: tasks are single-assignment, so any dependency-respecting order gives the same bits,
// and a task whose predecessor is NOT where the layout promised (a cycle longer than two cells that the jumping gave
// up on) simply stays pending and is finished by k_solve_patch launches over the same store order (planet.hip).
#include <hip/hip_runtime.h>

#include <string>

#include "device.h"

namespace wo {

namespace {

constexpr int32_t WO_RANGE_NONE = 0x7f7f7f7f;
#ifndef WO_BASIN_RANGE_SLOTS
#define WO_BASIN_RANGE_SLOTS 256
#endif
constexpr int WO_BASIN_RANGE = WO_BASIN_RANGE_SLOTS;                 // slots per workgroup range (a range = the groups that start in one stretch of this many slots)
constexpr int WO_BASIN_KEY_BITS = 16;               // bits of a group key (two radix passes)
constexpr int WO_BASIN_CHASE_CAP = 1 << 16;         // pointer-jumping steps of one thread before it gives up (never reached: chains and rings are shorter)

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
        keyOfCell[slotCell ? slotCell[s] : s] = (uint32_t)j >> shift;      // slotCell == nullptr: slot == cell id
    }
}
// thread i: the i-th cell in processing order (largest rank first)
// scramble (test hook, WO_BASIN_SCRAMBLE=1): every third cell is sent to the neighbouring group, so groups are no longer closed under
// the dependencies and the launch leaves tasks pending — the k_solve_patch launches that finish them must give the same bits
__global__ __launch_bounds__(WO_BLOCK) void k_basin_keys(const int32_t* __restrict__ land, const uint32_t* __restrict__ keyOfCell, int32_t L,
                                                          uint32_t* __restrict__ keys, int32_t* __restrict__ vals, int32_t scramble, int32_t* rangeStart, int32_t nRangeWords) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nRangeWords; i += gridDim.x * blockDim.x) rangeStart[i] = WO_RANGE_NONE;     // k_basin_slots takes minima into it (was a memset launch)
    if (blockIdx.x == 0 && threadIdx.x == 0) rangeStart[nRangeWords] = 0;                                                                     // number of long ranges (k_basin_long)
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) {
        const int32_t c = land[L - 1 - i];
        keys[i] = (scramble && c % 3 == 0) ? (keyOfCell[c] ^ 1u) : keyOfCell[c];
        vals[i] = c;
    }
}

// slotOf[cell] = position in the group-major list; rangeStart[k] = first position >= k*T that starts a group (0x7f7f7f7f: none)
__global__ __launch_bounds__(WO_BLOCK) void k_basin_slots(const int32_t* __restrict__ order, const uint32_t* __restrict__ keys, int32_t* __restrict__ slotOf, int32_t L,
                                                           int32_t* rangeStart, int32_t rangeT) {
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < L; q += gridDim.x * blockDim.x) {
        if (slotOf) slotOf[order[q]] = q;           // (the in-tree sort's last pass has written it)
        if (q == 0 || keys[q] != keys[q - 1]) atomicMin(&rangeStart[q / rangeT], q);
    }
}

// (Round 4: the list used to hold 256 ranges of >= 1 536 slots, and a pass of the benched planet has 240-330 of those: whichever 256 the atomic
// happened to admit were listed, and some of the longest ranges — 8-10 k slots, the ones that set the length of the launch — started late with the
// unlisted.  With 2 048 places for ranges of >= 2 048 slots every long range is listed: solve launches 45.4 -> 38.2 ms per step.)
// Long ranges first.  A launch of the solve lasts as long as its longest range (one workgroup walking the biggest drainage
// component), and workgroups are handed out in block order: a long range whose block comes late starts late, with the chip
// already emptying.  The ranges of at least WO_LONG_RANGE slots are listed here (after all range starts are known), the first
// WO_LONG_MAX blocks of the solve launch take them, and the block that would have met such a range in its turn skips it.
// big[0] = count, big[1 + i] = range index; flag[k] = 1: range k is on the list.
#ifndef WO_LONG_RANGE_SLOTS
#define WO_LONG_RANGE_SLOTS 2048
#endif
#ifndef WO_LONG_LIST
#define WO_LONG_LIST 2048
#endif
constexpr int32_t WO_LONG_RANGE = WO_LONG_RANGE_SLOTS, WO_LONG_MAX = WO_LONG_LIST;
__device__ inline int32_t range_end(const int32_t* __restrict__ rangeStart, int32_t k, int32_t nRanges, int32_t L) {
    for (int32_t j = k + 1; j < nRanges; ++j) { const int32_t v = rangeStart[j]; if (v != WO_RANGE_NONE) return v; }
    return L;
}
__global__ __launch_bounds__(WO_BLOCK) void k_basin_long(const int32_t* __restrict__ rangeStart, int32_t nRanges, int32_t L, int32_t* big, uint8_t* __restrict__ flag) {
    for (int32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nRanges; k += gridDim.x * blockDim.x) {
        const int32_t S = rangeStart[k];
        uint8_t f = 0;
        if (S != WO_RANGE_NONE && range_end(rangeStart, k, nRanges, L) - S >= WO_LONG_RANGE) {
            const int32_t at = atomicAdd(&big[0], 1);
            if (at < WO_LONG_MAX) { big[1 + at] = k; f = 1; }
        }
        flag[k] = f;
    }
}
// the range a block of the solve launch walks: blocks 0 .. WO_LONG_MAX-1 take the listed long ranges, block WO_LONG_MAX + k the
// range k unless it is listed.  false: nothing to do.
__device__ inline bool block_range(const int32_t* __restrict__ rangeStart, int32_t nRanges, int32_t L, const int32_t* __restrict__ big, const uint8_t* __restrict__ flag, int32_t& S, int32_t& E) {
    int32_t k;
    if (big) {
        if ((int32_t)blockIdx.x < WO_LONG_MAX) { const int32_t n = big[0] < WO_LONG_MAX ? big[0] : WO_LONG_MAX; if ((int32_t)blockIdx.x >= n) return false; k = big[1 + blockIdx.x]; }
        else { k = (int32_t)blockIdx.x - WO_LONG_MAX; if (flag[k]) return false; }
    } else k = (int32_t)blockIdx.x;
    S = rangeStart[k];
    if (S == WO_RANGE_NONE) return false;
    E = range_end(rangeStart, k, nRanges, L);
    return true;
}

// (Round 3's cooperative form — NW waves per range walking super-chunks of 64 x NW tasks with ONE barrier per super-chunk, k_solve_coop — was the
// default until round 4 and a cross-check route until round 6: 350 us per launch against 228 for what follows; removed.)
// ---------------------------------------------------------------------------------------------------------------------
// The walk of a range by a workgroup of NW waves, super-chunk by super-chunk of 64 x NW tasks, WITHOUT a barrier.  Phase clocks of the barrier form on a 1 600-slot range (profiles/r04c_*): 48 % of
// a wave's time is the wait at the super-chunk barrier (the waves' in-chunk chains differ in length and the launch pays the
// SUM over the super-chunks of the longest chain in each), 35 % the polling loop at ~1 500 clocks per pass (three LDS reads,
// each waited for on its own, then the turn's three divisions).  Here
//   * a wave moves on to its 64 tasks of the next super-chunk as soon as its own are done: it may run LAG super-chunks ahead
//     of the slowest wave of the workgroup (progress counters in LDS instead of the barrier).  The ring keeps 8 super-chunks:
//     a slot of super-chunk k is overwritten by k + 8, whose wave starts only when every wave has finished k + 8 - 1 - LAG
//     >= k + LOOKBACK, i.e. after the last task that reads super-chunk k through the ring; predecessors further back than
//     LOOKBACK = LAG + 2 super-chunks come from global memory, where their owner stored them on starting its next
//     super-chunk — at the latest before the reader issued the load (it starts super-chunk c - 1, where the loads for c go
//     out, only when everybody has finished c - 2 - LAG);
//   * a polling pass issues its three LDS reads together (clamped word: the task's own) and waits once.
// Same tasks, same inputs: bit-identical results (single-assignment dataflow).
// ---------------------------------------------------------------------------------------------------------------------
// The two divisions of a turn whose divisor is a per-task constant (1 + factor, cellDist[t]) without the divisor's part of the
// division on the dependency chain.  hipcc expands x / y into v_div_scale of both operands, v_rcp, two Newton steps on the reciprocal
// (4 fma), then q0 = x * r, e = fma(-y, q0, x), q = fma(e, r, q0) (v_div_fmas) and v_div_fixup; when neither operand needs scaling the
// scales are identities and the fix-up passes q through, so everything before q0 depends on y alone: recip_refined() is that part,
// run when the record arrives (off the chain), div_tail() the three dependent operations that remain — the same instructions on the
// same values, hence the same bits (checked on 8.6e9 operand pairs in round 2, profiles/microbench/fastdiv.hip, and by the CRC of the
// benched field).  "No scaling" holds for every operand a turn can see when factor_ok(): the dividends are sums / differences of
// f32 heights (0 or >= 2^-149) times a factor in [2^-300, 2^300], the divisors 1 + factor and an f32 distance.  A task that is not
// ok takes the plain divisions (the whole wave, for that pass).  Round 3 measured a form with the range tests ON the chain: slower.
__device__ inline double recip_refined(double y) {
    const double r0 = __builtin_amdgcn_rcp(y);
    const double f0 = __builtin_fma(-y, r0, 1.0);
    const double r1 = __builtin_fma(r0, f0, r0);
    const double f1 = __builtin_fma(-y, r1, 1.0);
    return __builtin_fma(r1, f1, r1);
}
__device__ inline double div_tail(double x, double y, double r) {
    const double q0 = x * r;
    const double e = __builtin_fma(-y, q0, x);
    return __builtin_fma(e, r, q0);
}
__device__ inline bool factor_ok(double f, float cellDistT, bool hasT2) {
    const bool fOk = f == 0.0 || (f >= 0x1p-300 && f <= 0x1p300);                    // (NaN / inf fail both)
    const bool dOk = !hasT2 || (cellDistT >= 0x1p-126f && cellDistT <= 0x1p100f);   // a normal f32
    return fOk && dOk;
}
// solve_apply_flat with the two constant divisors' reciprocals in hand
__device__ inline SolveOut solve_apply_recip(const SolveTask& T, double factor, double y1, double r1, double y2, double r2, double er, double et, double et2, int32_t tag) {
    const bool hasT = (T.flags & 4u) != 0, tOcean = (T.flags & 1u) != 0, hasT2 = (T.flags & 8u) != 0;
    const double hr = et > 0 ? et : 0;
    double hn = div_tail(er + factor * hr, y1, r1);
    hn = hn < hr ? hr : hn;
    hn = hn < 0 ? 0 : hn;
    const double eroded = er - hn;
    const double sl = div_tail(fabs(et - et2), y2, r2);
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

// Waves per range: TWO (measured at 10 M cells, solve launches per step: 1 wave 66.7 ms, 2 waves 45.5, 4 waves 48.2, 8 waves 77.7; lag 1 / 2 / 6
// super-chunks with two waves: 45.6 / 45.5 / 52.5).  A wave spends ~2 000 clocks in a pass that holds a turn (WO_BASIN_STATS: 80 % of the
// slowest range's time is inside turns, 75 % of its passes hold one) and a hand-off to another wave costs about as much again, so fewer,
// fuller waves win until one wave has to take every level alone.
template <int NW, bool STATS, int LAG_ = 2>
__global__ __launch_bounds__(64 * NW, (NW <= 4 ? 4 : 2)) void k_solve_flowing(Fields F, int32_t L, const int32_t* __restrict__ rangeStart, int32_t nRanges, int32_t launchTag,
                                                               int32_t* patchPending, int32_t* totalPending, const int32_t* __restrict__ big, const uint8_t* __restrict__ longFlag,
                                                               unsigned long long* stats) {
    constexpr int SC = 64 * NW;                        // tasks per super-chunk
    constexpr int LAG = LAG_, LOOKBACK = LAG + 2;
    constexpr int RCHUNKS = (LOOKBACK + LAG + 1 <= 8) ? 8 : 16;
    constexpr int RING = RCHUNKS * SC;                 // >= (LOOKBACK + LAG + 1) super-chunks, power of two
    static_assert(LOOKBACK + LAG + 1 <= RCHUNKS, "ring too small for the lag");
    __shared__ unsigned long long s_ring[2 * RING];
    __shared__ int32_t s_prog[NW];                     // super-chunks each wave has finished
    __shared__ int32_t s_level[STATS ? RING : 1];      // WO_BASIN_STATS (diagnostic): depth of every task in the dependency DAG
    __shared__ int32_t s_maxLevel;
    __shared__ unsigned long long s_passes[STATS ? NW : 1], s_readyPasses[STATS ? NW : 1], s_clk[4];
    unsigned long long myPasses = 0, myReady = 0, cRead = 0, cTurn = 0, cRest = 0, cWait = 0;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const long long cStart = STATS ? clock64() : 0;
    if (STATS && tid == 0) s_maxLevel = 0;
    int32_t S, E;
    if (!block_range(rangeStart, nRanges, L, big, longFlag, S, E)) return;
    const unsigned long long* G = reinterpret_cast<const unsigned long long*>(F.out);
    for (int i = tid; i < 2 * RING; i += SC) s_ring[i] = 0;        // last pass's launch left the very tags this one waits for
    if (tid < NW) s_prog[tid] = 0;
    __syncthreads();
    auto ring_put = [&](int32_t word, float v, int32_t tag) {
        __hip_atomic_store(&s_ring[word], (unsigned long long)__float_as_uint(v) | ((unsigned long long)(uint32_t)tag << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto ring_get = [&](int32_t word) { return __hip_atomic_load(&s_ring[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto record = [&](int32_t q) { return F.task[q < L ? q : L - 1]; };
    auto far_index = [&](int32_t g, int32_t cbase) { const int32_t sq = g >> 1; return (g >= 0 && (sq < S || sq < cbase - LOOKBACK * SC)) ? g : 0; };
    SolveTask T1 = record(S + tid), T2 = record(S + SC + tid);
    unsigned long long a0 = G[far_index(T1.predSelf, S)], a1 = G[far_index(T1.predT, S)], a2 = G[far_index(T1.predT2, S)];
    SolveOut oPrev; oPrev.self.v = 0; oPrev.self.tag = 0; oPrev.dep.v = 0; oPrev.dep.tag = 0;
    bool storePrev = false;
    uint32_t fPrev = 0; int32_t rPrev = 0, tPrev = 0;
    auto store_prev = [&](int32_t slot) {
        F.out[slot] = oPrev;
        if (F.solveFinals) {
            if (fPrev & 16u) { F.e2[rPrev] = oPrev.self.v; F.me[rPrev] = oPrev.self.v; }
            if (fPrev & 32u) { F.e2[tPrev] = oPrev.dep.v; F.me[tPrev] = oPrev.dep.v; }
        }
    };
    int32_t ci = 0;                                    // index of the super-chunk
    for (int32_t base = S; base < E; base += SC, ++ci) {
        const int32_t q = base + tid;
        const bool mine = q < E;
        // not more than LAG super-chunks ahead of the slowest wave (see above)
        const long long cw0 = STATS ? clock64() : 0;
        if (ci > LAG) {
            for (;;) {
                int32_t mn = 0x7fffffff;
#pragma unroll
                for (int w = 0; w < NW; ++w) { const int32_t v = __hip_atomic_load(&s_prog[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); mn = v < mn ? v : mn; }
                if (mn >= ci - LAG) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      // the far loads below read what the owners stored before their release of s_prog
        }
        const long long cw1 = STATS ? clock64() : 0;
        if (storePrev) store_prev(q - SC);                             // a super-chunk late: the wait for this super-chunk's loads at the top of the loop would otherwise also wait for a store issued a moment ago
        const SolveTask T = T1;
        T1 = T2;
        T2 = record(base + 2 * SC + tid);
        const unsigned long long n0 = G[far_index(T1.predSelf, base + SC)], n1 = G[far_index(T1.predT, base + SC)], n2 = G[far_index(T1.predT2, base + SC)];
        double er = T.e0r, et = T.e0t, et2 = T.e0t2;
        bool blocked = false;
        const int32_t myWord = (2 * q) & (2 * RING - 1);
        int32_t r0 = -1, r1 = -1, r2 = -1, x0 = 0, x1 = 0, x2 = 0;     // ring word and expected tag of the predecessors that come through the ring
        if (mine) {
            auto classify = [&](int32_t g, unsigned long long far, double& v, int32_t& rw, int32_t& expect) {
                if (g < 0) return;
                const int32_t sq = g >> 1;
                if (sq >= q) { blocked = true; return; }                                         // not in processing order: the layout is off
                if (sq >= S && sq >= base - LOOKBACK * SC) { rw = g & (2 * RING - 1); expect = sq + 1; return; }
                const int32_t tag = (int32_t)(far >> 32);
                const bool own = sq >= S;                                                        // written by this workgroup, long ago
                // valid: this workgroup's own output of THIS launch (launchTag: 1 in a checked pass, whose outputs were cleared; in an unchecked
                // pass a number no earlier pass used — nothing is cleared and whatever an earlier pass left there carries another tag).  A
                // predecessor outside the range is never valid in this launch: the layout promised there is none.
                if (!own || tag != launchTag) { blocked = true; return; }
                v = __uint_as_float((uint32_t)far);
            };
            classify(T.predSelf, a0, er, r0, x0); classify(T.predT, a1, et, r1, x1); classify(T.predT2, a2, et2, r2, x2);
        }
        // The polling loop as straight-line code: a single wave issues one instruction every few clocks, and the loop the
        // compiler made of the nested conditions above was ~250 instructions per pass (~2 000 clocks per level of the DAG,
        // WO_BASIN_STATS, profiles/r04e_*).  Here a pass reads the three ring words (a predecessor already in hand reads the
        // task's own word), tests the tags with integer compares, and — only when some lane is ready — runs the turn for the
        // whole wave without a branch (solve_apply_flat: the same operations on the same values, selected at the end).  Tags
        // do not change once written (the ring slot is reused LAG super-chunks later at the earliest), so nothing is latched.
        const int32_t w0 = r0 >= 0 ? r0 : myWord, w1 = r1 >= 0 ? r1 : myWord, w2 = r2 >= 0 ? r2 : myWord;
        const bool n0b = r0 >= 0, n1b = r1 >= 0, n2b = r2 >= 0;
        const float fr = (float)er, ft = (float)et, ft2 = (float)et2;      // inputs that are already in hand (exact: they came from floats)
        const SolvePrepared pre = solve_prepare(T, F.solveK, F.solveM, F.solveDt);
        // (off the chain: the record has just arrived, the predecessors have not)
        const double y1 = 1 + pre.factor, y2 = (double)T.cellDistT;
        const double rc1 = recip_refined(y1), rc2 = recip_refined(y2);
        const bool recipOk = !mine || factor_ok(pre.factor, T.cellDistT, (T.flags & 8u) != 0);
        bool open = mine;
        SolveOut o; o.self.v = 0; o.self.tag = 0; o.dep.v = 0; o.dep.tag = 0;
        if (mine && blocked) { ring_put(myWord, 0.0f, -(q + 1)); ring_put(myWord + 1, 0.0f, -(q + 1)); open = false; }
        const bool waveRecipOk = !__any(!recipOk);
        if (STATS) { const long long cw2 = clock64() + (long long)(__float_as_int((float)er) & 0); cRest += (unsigned long long)(cw2 - cw1); cWait += (unsigned long long)(cw1 - cw0); }
        while (__any(open)) {
            const long long c0 = STATS ? clock64() : 0;
            const unsigned long long g0 = ring_get(w0), g1 = ring_get(w1), g2 = ring_get(w2);
            const int32_t t0 = (int32_t)(g0 >> 32), t1 = (int32_t)(g1 >> 32), t2 = (int32_t)(g2 >> 32);
            const long long c1 = STATS ? (long long)(clock64() + (t0 & 0)) : 0;       // (after the reads have arrived)
            const bool ok = (!n0b | (t0 == x0)) & (!n1b | (t1 == x1)) & (!n2b | (t2 == x2));
            const bool bad = (n0b & (t0 == -x0)) | (n1b & (t1 == -x1)) | (n2b & (t2 == -x2));
            const bool ready = open & ok;
            if (STATS) { ++myPasses; if (__any(ready)) ++myReady; }
            if (__any(ready)) {
                const double ver = n0b ? (double)__uint_as_float((uint32_t)g0) : (double)fr;
                const double vet = n1b ? (double)__uint_as_float((uint32_t)g1) : (double)ft;
                const double vet2 = n2b ? (double)__uint_as_float((uint32_t)g2) : (double)ft2;
                const SolveOut oo = waveRecipOk ? solve_apply_recip(T, pre.factor, y1, rc1, y2, rc2, ver, vet, vet2, launchTag) : solve_apply_flat(T, pre, ver, vet, vet2, launchTag);
                if (ready) {
                    o = oo;
                    if (STATS && LAG_ != 3) {        // predecessors beyond the ring window count as depth 0 (they finished long ago); (LAG_ == 3: diagnostic build without the depth bookkeeping)
                        int32_t lv = 0;
                        if (n0b) lv = max(lv, s_level[w0 >> 1]);
                        if (n1b) lv = max(lv, s_level[w1 >> 1]);
                        if (n2b) lv = max(lv, s_level[w2 >> 1]);
                        s_level[myWord >> 1] = lv + 1;
                        atomicMax(&s_maxLevel, lv + 1);
                    }
                    ring_put(myWord, oo.self.v, q + 1); ring_put(myWord + 1, oo.dep.v, q + 1);
                    open = false;
                }
            }
            const long long c2 = STATS ? clock64() : 0;
            if (STATS) { cRead += (unsigned long long)(c1 - c0); cTurn += (unsigned long long)(c2 - c1); }
            if (__any(open & bad)) {                                     // a predecessor is blocked (never on real layouts: WO_BASIN_SCRAMBLE)
                if (open & bad) { blocked = true; ring_put(myWord, 0.0f, -(q + 1)); ring_put(myWord + 1, 0.0f, -(q + 1)); open = false; }
            }
        }
        oPrev = o; storePrev = mine && !blocked; fPrev = T.flags; rPrev = T.pad_[0]; tPrev = T.pad_[1];
        if (mine && blocked) { atomicAdd(&patchPending[q / WO_PATCH], 1); atomicAdd(totalPending, 1); }
        a0 = n0; a1 = n1; a2 = n2;
        if (lane == 0) __hip_atomic_store(&s_prog[wave], ci + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);     // after this wave's ring words (a wave's LDS operations execute in order)
    }
    if (storePrev) store_prev(S + ((E - S - 1) / SC) * SC + tid);
    if (STATS) {        // the range that took longest: {clocks, slots, DAG depth seen through the ring}, and its waves' polling passes
        if (lane == 0) { s_passes[wave] = myPasses; s_readyPasses[wave] = myReady; }
        if (tid == 0) { s_clk[0] = cWait; s_clk[1] = cTurn; s_clk[2] = myPasses; s_clk[3] = cRest; }
        __syncthreads();
        if (tid == 0) {
            const unsigned long long c = (unsigned long long)(clock64() - cStart);
            const unsigned long long packed = (c << 24) | (unsigned long long)(s_maxLevel & 0xffffff);
            const unsigned long long old = atomicMax(&stats[0], packed);
            if (packed > old) {       // (racy by design: diagnostic)
                stats[1] = (unsigned long long)(E - S);
                unsigned long long mp = 0, mr = 0;
                for (int w = 0; w < NW; ++w) { mp = s_passes[w] > mp ? s_passes[w] : mp; mr = s_readyPasses[w] > mr ? s_readyPasses[w] : mr; }
                stats[2] = mp; stats[3] = mr; stats[4] = s_clk[0]; stats[5] = s_clk[1]; stats[6] = s_clk[2]; stats[7] = s_clk[3];
            }
        }
    }
}

}  // namespace

// Group-major store order for this pass (d_basinSlot) and the sorted group keys (d_keys[1]).  Call after the receivers
// pass (F.tr) and before k_solve_setup; everything is enqueued on the planet's stream, no host sync.
void basin_alloc(wo_planet* p) {
    const int32_t N = p->N;
    hipStream_t s = cur_stream(p);
    if (!p->d_basinJ) {
        WO_HIP(hipMalloc((void**)&p->d_basinJ, (size_t)N * 4));
        WO_HIP(hipMalloc((void**)&p->d_basinSlot, (size_t)N * 4));
        WO_HIP(hipMalloc((void**)&p->d_basinKey, (size_t)N * 4));
        WO_HIP(hipMalloc((void**)&p->d_basinVals[0], (size_t)N * 4)); WO_HIP(hipMalloc((void**)&p->d_basinVals[1], (size_t)N * 4));   // own buffers: the layout runs beside the flow accumulation, whose rounds use the planet's lists
        WO_HIP(hipMalloc((void**)&p->d_basinRange, ((size_t)N / 64 + 8 + WO_LONG_MAX) * 4));        // range starts, then {count, long ranges}
        WO_HIP(hipMalloc((void**)&p->d_basinLong, (size_t)N / 64 + 8));
        WO_HIP(hipMemsetAsync(p->d_basinSlot, 0xff, (size_t)N * 4, s));
    }
}

// J already holds the start state (k_receivers_flow_init, Fields::basinJ; pairs of cells draining into each other are rings of
// two there, which the search cuts like any ring: at the smaller slot).  slotIdentity: a land
// cell's Morton slot is its id (land-first mirror).
void basin_layout(wo_planet* p, bool slotIdentity) {
    const int32_t L = p->L;
    basin_alloc(p);
    int bitsL = 1;
    while (((int64_t)1 << bitsL) < (int64_t)L) ++bitsL;
    const int shift = bitsL > WO_BASIN_KEY_BITS ? bitsL - WO_BASIN_KEY_BITS : 0;
    const int grid = blocks_for(L, 1 << 16);
    launch(p, FAM_BASIN, k_basin_jump, grid, WO_BLOCK, slotIdentity ? (const int32_t*)nullptr : (const int32_t*)p->d_patchOrder, p->d_basinJ, L, (int32_t)shift, p->d_basinKey);
    const int rangeT = WO_BASIN_RANGE;
    const int nRanges = (int)(((int64_t)L + rangeT - 1) / rangeT);
    launch(p, FAM_BASIN, k_basin_keys, grid, WO_BLOCK, (const int32_t*)p->d_land[p->landCur], (const uint32_t*)p->d_basinKey, L, p->d_keys[0], p->d_basinVals[0],
           (int32_t)(p->opt.basinScramble ? 1 : 0), p->d_basinRange, (int32_t)(nRanges + 1));
    // the in-tree sort (radix.hip); its last pass also writes slotOf[cell] = position
    uint32_t* const kb[2] = {p->d_keys[0], p->d_keys[1]};
    int32_t* const vb[2] = {p->d_basinVals[0], p->d_basinVals[1]};
    const int sorted = radix_sort_pairs(p, FAM_BASIN_SORT, kb, vb, L, 0, 16 /* always two digits: the sort's two group-total buffers swap roles every pass and only stay consistent over an even number of passes; key bits above bitsL - shift are zero */, p->d_basinSlot, radix_scratch(p, 1), p->N, p->rsFlip[1]);
    launch(p, FAM_BASIN, k_basin_slots, blocks_for(L, 4096), WO_BLOCK, (const int32_t*)p->d_basinVals[sorted], (const uint32_t*)p->d_keys[sorted], (int32_t*)nullptr, L, p->d_basinRange, (int32_t)rangeT);
    launch(p, FAM_BASIN, k_basin_long, blocks_for(nRanges), WO_BLOCK, (const int32_t*)p->d_basinRange, (int32_t)nRanges, L, p->d_basinRange + nRanges + 1, p->d_basinLong);
}

// the one launch of the pass; F.slotOf must be d_basinSlot, patchPending zeroed
void basin_solve_launch(wo_planet* p, const Fields& F, int32_t launchTag, int32_t* totalPending) {
    const int nRanges = (int)(((int64_t)p->L + WO_BASIN_RANGE - 1) / WO_BASIN_RANGE);
    ++p->basinLaunches;
    const int32_t* big = (const int32_t*)(p->d_basinRange + nRanges + 1);
    const uint8_t* flag = (const uint8_t*)p->d_basinLong;
    const int grid = nRanges + WO_LONG_MAX;
    launch(p, FAM_SOLVE_BASIN, k_solve_flowing<2, false>, grid, 128, F, p->L, (const int32_t*)p->d_basinRange, (int32_t)nRanges, launchTag, p->d_patchPending, totalPending, big, flag, (unsigned long long*)nullptr);
    // (k_solve_flowing<2, true>: the same kernel with phase clocks and the depth of the slowest range's dependency DAG written to its last argument — the
    // diagnostic build behind DESIGN.md's "clocks per level" figures, profiles/r04c_*; not instantiated in the product)
}

void basin_free(wo_planet* p) {
    if (p->d_basinJ) (void)hipFree(p->d_basinJ);
    if (p->d_basinSlot) (void)hipFree(p->d_basinSlot);
    if (p->d_basinKey) (void)hipFree(p->d_basinKey);
    for (auto& v : p->d_basinVals) { if (v) (void)hipFree(v); v = nullptr; }
    if (p->d_basinRange) (void)hipFree(p->d_basinRange);
    if (p->d_basinLong) (void)hipFree(p->d_basinLong);
    p->d_basinLong = nullptr;
    p->d_basinJ = nullptr; p->d_basinSlot = nullptr; p->d_basinKey = nullptr; p->d_basinRange = nullptr;
}

}  // namespace wo
