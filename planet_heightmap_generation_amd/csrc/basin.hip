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

#include "device.h"

namespace wo {

namespace {

constexpr int WO_BASIN_THREADS = WO_PATCH;          // one task per thread, one window per WO_PATCH store slots
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

// thread i: the i-th cell in processing order (largest rank first)
__global__ __launch_bounds__(WO_BLOCK) void k_basin_keys(const int32_t* __restrict__ land, const int32_t* __restrict__ mslot, int32_t* J, int32_t L, int32_t shift,
                                                          uint32_t* __restrict__ keys, int32_t* __restrict__ vals) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x) {
        const int32_t c = land[L - 1 - i];
        const int32_t s = mslot[c];
        const int32_t j = basin_root(J, s);
        __atomic_store_n(&J[s], j, __ATOMIC_RELAXED);
        keys[i] = (uint32_t)j >> shift;
        vals[i] = c;
    }
}

__global__ __launch_bounds__(WO_BLOCK) void k_basin_slots(const int32_t* __restrict__ order, int32_t* __restrict__ slotOf, int32_t L) {
    for (int32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < L; q += gridDim.x * blockDim.x) slotOf[order[q]] = q;
}

// first position q >= from that starts a group (keys[q] != keys[q-1], or q == 0), L if there is none
__device__ inline int32_t first_head(const uint32_t* __restrict__ keys, int32_t L, int64_t from, int32_t* s_min) {
    const int tid = threadIdx.x;
    for (int64_t base = from;; base += WO_BASIN_THREADS) {
        if (tid == 0) *s_min = 0x7fffffff;
        __syncthreads();
        const int64_t q = base + tid;
        const bool head = q >= L || q == 0 || keys[q] != keys[q - 1];
        const unsigned long long m = __ballot(head);
        if (m && (tid & 63) == 0) { const int64_t qq = q + __builtin_ctzll(m); atomicMin(s_min, (int32_t)(qq >= L ? L : qq)); }      // the wave's lowest head
        __syncthreads();
        const int32_t r = *s_min;
        __syncthreads();
        if (r != 0x7fffffff) return r;
    }
}

__global__ __launch_bounds__(WO_BASIN_THREADS) void k_solve_basin(Fields F, int32_t L, const uint32_t* __restrict__ keys, int32_t rangeT, int32_t launchTag,
                                                                   int32_t* patchPending, int32_t* totalPending) {
    __shared__ unsigned long long s_out[2 * WO_PATCH];
    __shared__ int32_t s_min;
    const int tid = threadIdx.x;
    const int64_t from = (int64_t)blockIdx.x * rangeT;
    if (from >= L) return;
    const int32_t S = first_head(keys, L, from, &s_min);
    if (S >= L || (int64_t)S >= from + rangeT) return;               // the groups that start in this stretch: none
    const int32_t E = first_head(keys, L, from + rangeT, &s_min);
    const Granule* G = reinterpret_cast<const Granule*>(F.out);
    volatile unsigned long long* vs = s_out;
    const unsigned long long BLOCKED = 0xffffffff00000000ull;       // tag -1
    auto pack = [](Granule g) { return (unsigned long long)__float_as_uint(g.v) | ((unsigned long long)(uint32_t)g.tag << 32); };
    for (int32_t p = S / WO_PATCH; p <= (E - 1) / WO_PATCH; ++p) {
        const int32_t base = p * WO_PATCH;
        const int32_t q = base + tid;
        const int32_t wLo = S > base ? S : base, wHi = E < base + WO_PATCH ? E : base + WO_PATCH;
        const bool mine = q >= wLo && q < wHi;
        vs[2 * tid] = 0; vs[2 * tid + 1] = 0;
        __syncthreads();
        auto is_local = [&](int32_t g) { const int32_t sq = g >> 1; return sq >= wLo && sq < wHi; };
        SolveTask T;
        SolvePrepared pre;
        double er = 0, et = 0, et2 = 0;
        bool unresolved = mine, blocked = false;
        if (mine) {
            T = F.task[q];
            er = T.e0r; et = T.e0t; et2 = T.e0t2;
            // a predecessor outside the window: an earlier window of this workgroup (there by now, whatever its tag) or — only
            // when the layout is off — somebody else's task, which counts when an earlier launch produced it
            auto ext = [&](int32_t g, double& v) {
                if (g < 0 || is_local(g)) return;
                const Granule gq = G[g];
                const int32_t sq = g >> 1;
                const bool own = sq >= S && sq < wLo;
                if (gq.tag == 0 || (!own && gq.tag >= launchTag)) { blocked = true; return; }
                v = gq.v;
            };
            ext(T.predSelf, er); ext(T.predT, et); ext(T.predT2, et2);
            if (blocked) { unresolved = false; vs[2 * tid] = BLOCKED; vs[2 * tid + 1] = BLOCKED; }
            else pre = solve_prepare(T, F.solveK, F.solveM, F.solveDt);
        }
        for (int spin = 0; __any(unresolved); ++spin) {
            if (spin) __builtin_amdgcn_s_sleep(1);
            if (!unresolved) continue;
            int32_t open = 0, fail = 0;
            double a = er, b = et, c = et2;
            auto rd = [&](int32_t g, double& v) {
                if (g < 0 || !is_local(g)) return;
                const unsigned long long w = vs[g - 2 * base];
                const int32_t tg = (int32_t)(w >> 32);
                if (tg < 0) fail = 1;
                else if (tg == 0) open = 1;
                else v = __uint_as_float((uint32_t)w);
            };
            rd(T.predSelf, a); rd(T.predT, b); rd(T.predT2, c);
            if (fail) { vs[2 * tid] = BLOCKED; vs[2 * tid + 1] = BLOCKED; unresolved = false; blocked = true; }
            else if (!open) {
                const SolveOut o = solve_apply(T, pre, a, b, c, launchTag);
                vs[2 * tid] = pack(o.self); vs[2 * tid + 1] = pack(o.dep);
                F.out[q] = o;
                unresolved = false;
            }
        }
        const unsigned long long bm = __ballot(blocked);
        if (bm && (tid & 63) == 0) { const int32_t n = __popcll(bm); atomicAdd(&patchPending[p], n); atomicAdd(totalPending, n); }
        __threadfence_block();
        __syncthreads();
    }
}

}  // namespace

// Group-major store order for this pass (d_basinSlot) and the sorted group keys (d_keys[1]).  Call after the receivers
// pass (F.tr) and before k_solve_setup; everything is enqueued on the planet's stream, no host sync.
void basin_layout(wo_planet* p) {
    const int32_t N = p->N, L = p->L;
    hipStream_t s = p->ctx->stream;
    if (!p->d_basinJ) {
        WO_HIP(hipMalloc((void**)&p->d_basinJ, (size_t)N * 4));
        WO_HIP(hipMalloc((void**)&p->d_basinSlot, (size_t)N * 4));
        WO_HIP(hipMemsetAsync(p->d_basinSlot, 0xff, (size_t)N * 4, s));
    }
    const Fields F = p->fields();
    int bitsL = 1;
    while (((int64_t)1 << bitsL) < (int64_t)L) ++bitsL;
    static const int keyBits = getenv("WO_BASIN_KEY_BITS") ? std::max(8, std::min(30, atoi(getenv("WO_BASIN_KEY_BITS")))) : 16;
    const int shift = bitsL > keyBits ? bitsL - keyBits : 0;
    const int grid = blocks_for(L, 1 << 16);
    launch(p, FAM_BASIN, k_basin_init, grid, WO_BLOCK, F, (const int32_t*)p->d_patchOrder, (const int32_t*)p->d_slotOf, p->d_basinJ, L);
    launch(p, FAM_BASIN, k_basin_keys, grid, WO_BLOCK, (const int32_t*)p->d_land[p->landCur], (const int32_t*)p->d_slotOf, p->d_basinJ, L, (int32_t)shift,
           p->d_keys[0], p->d_listA);
    {
        hipEvent_t a = nullptr, b = nullptr;
        if (p->profiling) { a = profile_event(p); b = profile_event(p); WO_HIP(hipEventRecord(a, s)); }
        size_t bytes = p->sortTempBytes;
        WO_HIP(hipcub::DeviceRadixSort::SortPairs(p->d_sortTemp, bytes, (const uint32_t*)p->d_keys[0], p->d_keys[1],
                                                 (const int32_t*)p->d_listA, p->d_listB, L, 0, bitsL - shift, s));
        if (p->profiling) { WO_HIP(hipEventRecord(b, s)); p->pending.push_back({FAM_BASIN_SORT, a, b}); }
    }
    launch(p, FAM_BASIN, k_basin_slots, blocks_for(L, 4096), WO_BLOCK, (const int32_t*)p->d_listB, p->d_basinSlot, L);
}

// the one launch of the pass; F.slotOf must be d_basinSlot, patchPending zeroed
void basin_solve_launch(wo_planet* p, const Fields& F, int32_t launchTag, int32_t* totalPending) {
    static const int rangeT = getenv("WO_BASIN_RANGE") ? std::max(WO_PATCH, (atoi(getenv("WO_BASIN_RANGE")) / WO_PATCH) * WO_PATCH) : WO_PATCH;
    const int grid = (int)(((int64_t)p->L + rangeT - 1) / rangeT);
    launch(p, FAM_SOLVE_BASIN, k_solve_basin, grid, WO_BASIN_THREADS, F, p->L, (const uint32_t*)p->d_keys[1], (int32_t)rangeT, launchTag, p->d_patchPending, totalPending);
}

void basin_free(wo_planet* p) {
    if (p->d_basinJ) (void)hipFree(p->d_basinJ);
    if (p->d_basinSlot) (void)hipFree(p->d_basinSlot);
    p->d_basinJ = nullptr; p->d_basinSlot = nullptr;
}

}  // namespace wo
