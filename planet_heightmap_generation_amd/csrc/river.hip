// River-aligned patches for the implicit solve (reference pass: js/terrain-post.js:614-641; the solve's dataflow form:
// erode_ops.h, kernel k_solve_patch in kernels_impl.h).
//
// The patch solve advances a dependency chain at LDS speed inside a patch and pays a kernel launch for every patch
// boundary the chain crosses.  The solve's dependencies follow the drainage forest two levels deep (a task waits for the
// turn of its receiver, for the deposits of its lower siblings and for events on the receiver's receiver).  With
// spatial (Morton) patches a 10 M-cell planet needs ~41-49 launches per pass (research/solve_schedule.*); ordering the
// patch list along the forest needs 8-13:
//
//     after a cell, ALL its donors are listed (sibling chains stay together), then the donors' subtrees, heaviest
//     first — a river's main stem with its confluences is one contiguous run, tributaries follow as runs of their own.
//
// position(child i of c) = A(c) + i,   A(child i) = A(c) + #children(c) + sum over heavier siblings of (size - 1),
// A(root) = 1 + (sizes of the roots before it): a top-down path sum, evaluated by pointer jumping.  Subtree sizes are
// what the flow accumulation left in accA (sizes in the forest of forward edges); a pit hangs under the neighbour it
// drains up to (its size is not propagated further: the layout regions then overlap a little), a pit and a neighbour
// draining into each other are cut at the pit.  The positions are therefore KEYS, and the patch list is the land list
// SORTED by key: always a permutation, whatever the sizes.  The order is only a schedule — the solve is a
// single-assignment dataflow and gives the same bits under any patch order (checked by the parity tests and the CRC of
// the benched field) — so a stale or imperfect order costs launches, never results.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "device.h"

namespace wo {

namespace {

struct RiverBuf { uint32_t* A[2]; int32_t* J[2]; int32_t* idx; uint32_t* rootSize; uint32_t* rootBase; int32_t* flag; };

// parent in the layout forest: the receiver, unless the cell is an outlet (no land receiver) or the pit of a 2-cycle
__device__ inline int32_t river_parent(const Fields& F, int32_t c, TargetRank trc, TargetRank& trp) {
    const int32_t t = trc.target;
    trp.target = -1; trp.rank = -1;
    if (t < 0) return -1;
    trp = F.tr[t];
    if (trp.rank < 0) return -1;                                          // ocean cells carry rank -1
    if (trp.target == c && trc.rank > trp.rank) return -1;                // c and t drain into each other: cut at the lower one
    return t;
}

__global__ __launch_bounds__(WO_BLOCK) void k_river_offsets(Fields F, RiverBuf B) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < F.L; i += gridDim.x * blockDim.x) {
        const int32_t c = F.landIdx[i];
        const TargetRank trc = F.tr[c];
        TargetRank trp;
        const int32_t p = river_parent(F, c, trc, trp);
        if (p < 0) { B.rootSize[i] = F.accA[c]; B.idx[c] = -1; B.J[0][c] = -1; B.A[0][c] = 0; continue; }
        B.rootSize[i] = 0;
        const uint32_t mine = F.accA[c];
        int32_t b, nbs[WO_ROW];
        const int deg = load_row(F, p, b, nbs);
        uint32_t k = 0, before = 0; int32_t at = 0; bool passed = false;
        if (deg <= WO_ROW) {
            TargetRank q[WO_ROW]; uint32_t sz[WO_ROW];
#pragma unroll
            for (int j = 0; j < WO_ROW; ++j) { q[j] = F.tr[nbs[j]]; sz[j] = F.accA[nbs[j]]; }
#pragma unroll
            for (int j = 0; j < WO_ROW; ++j) {
                if (j >= deg) continue;
                const int32_t s = nbs[j];
                if (q[j].target != p || (trp.target == s && q[j].rank > trp.rank)) continue;      // not a donor of p in the layout forest
                ++k;
                if (s == c) { passed = true; continue; }
                if (sz[j] > mine || (sz[j] == mine && !passed)) { ++at; before += sz[j] - 1; }
            }
        } else {
            for (int32_t j = b; j < b + deg; ++j) {
                const int32_t s = F.adj[j];
                const TargetRank qs = F.tr[s];
                if (qs.target != p || (trp.target == s && qs.rank > trp.rank)) continue;
                ++k;
                if (s == c) { passed = true; continue; }
                const uint32_t ss = F.accA[s];
                if (ss > mine || (ss == mine && !passed)) { ++at; before += ss - 1; }
            }
        }
        B.A[0][c] = k + before; B.idx[c] = at; B.J[0][c] = p;
    }
}

__global__ __launch_bounds__(WO_BLOCK) void k_river_roots(Fields F, RiverBuf B, uint32_t* keys) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < F.L; i += gridDim.x * blockDim.x) {
        const int32_t c = F.landIdx[i];
        if (B.idx[c] < 0) { const uint32_t base = B.rootBase[i]; keys[i] = base; B.A[0][c] = base + 1; }
    }
}

// one pointer-jumping round of the path sums: A[c] += A[J[c]], J[c] = J[J[c]]  (ping-pong buffers: every read is of the previous round)
__global__ __launch_bounds__(WO_BLOCK) void k_river_jump(Fields F, RiverBuf B, int from, int32_t raiseFlag) {
    const uint32_t* Ai = B.A[from]; const int32_t* Ji = B.J[from];
    uint32_t* Ao = B.A[from ^ 1]; int32_t* Jo = B.J[from ^ 1];
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < F.L; i += gridDim.x * blockDim.x) {
        const int32_t c = F.landIdx[i];
        const int32_t j = Ji[c];
        uint32_t a = Ai[c]; int32_t j2 = -1;
        if (j >= 0) { a += Ai[j]; j2 = Ji[j]; }
        Ao[c] = a; Jo[c] = j2;
        if (raiseFlag && j2 >= 0) *B.flag = 1;
    }
}

__global__ __launch_bounds__(WO_BLOCK) void k_river_keys(Fields F, RiverBuf B, int fin, uint32_t* keys) {
    for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < F.L; i += gridDim.x * blockDim.x) {
        const int32_t c = F.landIdx[i];
        const int32_t at = B.idx[c];
        if (at >= 0) keys[i] = B.A[fin][F.tr[c].target] + (uint32_t)at;
    }
}

__global__ __launch_bounds__(WO_BLOCK) void k_river_slots(const int32_t* __restrict__ order, int32_t* __restrict__ slotOf, int32_t L) {
    for (int32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < L; s += gridDim.x * blockDim.x) slotOf[order[s]] = s;
}

}  // namespace

// Rebuilds d_patchOrder / d_slotOf from the drainage forest of the current iteration.  Call after the flow accumulation
// (F.tr and F.accA final) and before k_solve_setup.  All work is enqueued on the planet's stream; no host sync.
void river_patch_slots(wo_planet* p) {
    const int32_t N = p->N, L = p->L;
    hipStream_t s = p->ctx->stream;
    if (!p->d_riverA[0]) {
        for (int k = 0; k < 2; ++k) { WO_HIP(hipMalloc((void**)&p->d_riverA[k], (size_t)N * 4)); WO_HIP(hipMalloc((void**)&p->d_riverJ[k], (size_t)N * 4)); }
        WO_HIP(hipMalloc((void**)&p->d_riverIdx, (size_t)N * 4));
        WO_HIP(hipMalloc((void**)&p->d_riverRootSize, (size_t)N * 4));
        WO_HIP(hipMalloc((void**)&p->d_riverRootBase, (size_t)N * 4));
        WO_HIP(hipMalloc((void**)&p->d_riverFlag, 64));
    }
    const Fields F = p->fields();
    RiverBuf B;
    B.A[0] = p->d_riverA[0]; B.A[1] = p->d_riverA[1]; B.J[0] = p->d_riverJ[0]; B.J[1] = p->d_riverJ[1];
    B.idx = p->d_riverIdx; B.rootSize = p->d_riverRootSize; B.rootBase = p->d_riverRootBase; B.flag = p->d_riverFlag;
    const int grid = blocks_for(L, 1 << 16);
    launch(p, FAM_RIVER, k_river_offsets, grid, WO_BLOCK, F, B);
    {
        size_t bytes = 0;
        WO_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint32_t*)B.rootSize, B.rootBase, L, s));
        if (bytes > p->sortTempBytes) throw HipError{"river_patch_slots: scan temp storage exceeds the sort's"};
        hipEvent_t a = nullptr, b = nullptr;
        if (p->profiling) { a = profile_event(p); b = profile_event(p); WO_HIP(hipEventRecord(a, s)); }
        bytes = p->sortTempBytes;
        WO_HIP(hipcub::DeviceScan::ExclusiveSum(p->d_sortTemp, bytes, (const uint32_t*)B.rootSize, B.rootBase, L, s));
        if (p->profiling) { WO_HIP(hipEventRecord(b, s)); p->pending.push_back({FAM_RIVER, a, b}); }
    }
    launch(p, FAM_RIVER, k_river_roots, grid, WO_BLOCK, F, B, p->d_keys[0]);
    // path sums: 2^rounds must exceed the depth of the layout forest (the solve's own dependency depth is a few hundred
    // levels at 10 M cells); deeper cells just get a partial sum, i.e. a worse place in the order
    const int rounds = 12;
    int cur = 0;
    for (int r = 0; r < rounds; ++r, cur ^= 1) launch(p, FAM_RIVER, k_river_jump, grid, WO_BLOCK, F, B, cur, (int32_t)0);
    launch(p, FAM_RIVER, k_river_keys, grid, WO_BLOCK, F, B, cur, p->d_keys[0]);
    {
        hipEvent_t a = nullptr, b = nullptr;
        if (p->profiling) { a = profile_event(p); b = profile_event(p); WO_HIP(hipEventRecord(a, s)); }
        size_t bytes = p->sortTempBytes;
        WO_HIP(hipcub::DeviceRadixSort::SortPairs(p->d_sortTemp, bytes, (const uint32_t*)p->d_keys[0], p->d_keys[1],
                                                 (const int32_t*)p->d_landIdx, p->d_patchOrder, L, 0, 32, s));
        if (p->profiling) { WO_HIP(hipEventRecord(b, s)); p->pending.push_back({FAM_RIVER, a, b}); }
    }
    launch(p, FAM_RIVER, k_river_slots, blocks_for(L, 4096), WO_BLOCK, (const int32_t*)p->d_patchOrder, p->d_slotOf, L);
}

void river_free(wo_planet* p) {
    for (int k = 0; k < 2; ++k) { if (p->d_riverA[k]) (void)hipFree(p->d_riverA[k]); if (p->d_riverJ[k]) (void)hipFree(p->d_riverJ[k]); p->d_riverA[k] = nullptr; p->d_riverJ[k] = nullptr; }
    if (p->d_riverIdx) (void)hipFree(p->d_riverIdx);
    if (p->d_riverRootSize) (void)hipFree(p->d_riverRootSize);
    if (p->d_riverRootBase) (void)hipFree(p->d_riverRootBase);
    if (p->d_riverFlag) (void)hipFree(p->d_riverFlag);
    p->d_riverIdx = nullptr; p->d_riverRootSize = nullptr; p->d_riverRootBase = nullptr; p->d_riverFlag = nullptr;
}

}  // namespace wo
