// gfx950 kernels of the device flood (pass 1 of priorityFloodCarve as a label-correcting fixed point; bodies and the
// method in flood_ops.h).  A round is two launches: k_fl_eval (every cell whose neighbourhood changed re-derives its
// label from the previous round's labels into P[]) and k_fl_apply (the changed cells commit, their neighbours go on the
// next round's list; its last workgroup advances the control block: list flip, epoch change, termination).  The host
// queues rounds in batches and reads the control block back once per batch; lists are appended with one atomic per
// wavefront (ballot + popcount).  Included by planet.hip only (after kernels_impl.h).
#pragma once
#include <hip/hip_runtime.h>

#include "device.h"
#include "flood_ops.h"

namespace wo {

struct FlCtrl {
    int32_t dirtyN[2]; int32_t cur;        // round lists (ping-pong) and which one is read next
    int32_t changedN;
    int32_t pendN[2]; int32_t pcur;        // cells queued for an infeasible move; pcur = list new entries go to
    int32_t release, relIdx;               // the next evaluation pass runs the queued cells of list relIdx with the move allowed
    int32_t epoch;
    int32_t done;
    int32_t rounds, epochs;
    int32_t overflow;                      // a label stack exceeded FL_LD
    int32_t ticket;
    int32_t notFixed, ties;                // k_fl_verify
    int32_t pad_;
    long long evals, changes;
};

struct FlLists { int32_t* dirty[2]; int32_t* pend[2]; int32_t* changed; };

__device__ inline void wave_append(bool flag, int32_t value, int32_t* list, int32_t* counter) {
    const unsigned long long mask = __ballot(flag);
    if (mask == 0) return;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    int32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (int32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (flag) list[base + __popcll(mask & ((1ull << lane) - 1ull))] = value;
}

__global__ __launch_bounds__(WO_BLOCK) void k_fl_init(FloodDev D, float* eL, const float* e, FlCtrl* C) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { FlCtrl z{}; z.epoch = 1; *C = z; }
    WO_GRID_STRIDE(i, D.L) {
        const float v = e[D.cell[i]];
        eL[i] = v;
        FlHead h; h.par = FL_NONE; h.S = v; h.K = 0; h.dep = 0; h.top = 0; h.spare = 0;
        if (D.seedIdx[i] >= 0) {                                             // :118-128: key of a seed is f32(e + noise)
            h.par = FL_SEED; h.K = (float)((double)v + D.nz[i]); h.dep = 1; h.top = fl_pack(h.K, D.cell[i]);
            D.Astk[(size_t)i * FL_LD] = h.top;
        }
        D.A[i] = h;
        D.fdEpoch[i] = 0; D.inDirty[i] = 0; D.isPending[i] = 0;
    }
}
// first list: the unlabelled neighbours of the seeds (launched after k_fl_init on the same stream)
__global__ __launch_bounds__(WO_BLOCK) void k_fl_seed_dirty(FloodDev D, const int32_t* seeds, int32_t nSeeds, FlLists Ls, FlCtrl* C) {
    WO_BLOCK_STRIDE(s, valid, nSeeds) {
        const int32_t c = valid ? seeds[s] : 0;
        const int32_t b = valid ? D.off[c] : 0, deg = valid ? D.off[c + 1] - b : 0;
        for (int k = 0; __any(k < deg); ++k) {
            bool add = false; int32_t y = -1;
            if (k < deg) { y = D.adj[b + k]; add = D.seedIdx[y] < 0 && atomicExch(&D.inDirty[y], 1) == 0; }
            wave_append(add, y, Ls.dirty[0], &C->dirtyN[0]);
        }
    }
}

__global__ __launch_bounds__(WO_BLOCK) void k_fl_eval(FloodDev D, FlLists Ls, FlCtrl* C) {
    if (C->done) return;
    const bool rel = C->release != 0;
    const int32_t* list = rel ? Ls.pend[C->relIdx] : Ls.dirty[C->cur];
    const int32_t n = rel ? C->pendN[C->relIdx] : C->dirtyN[C->cur];
    const int32_t epoch = C->epoch;
    int32_t* pendOut = Ls.pend[C->pcur]; int32_t* pendCnt = &C->pendN[C->pcur];
    WO_BLOCK_STRIDE(i, valid, n) {
        bool changed = false, pendNew = false, over = false; int32_t x = -1;
        if (valid) {
            x = list[i];
            D.inDirty[x] = 0;
            changed = flood_eval_cell(D, x, epoch, rel, &pendNew, &over);
            if (over) C->overflow = 1;
        }
        wave_append(changed, x, Ls.changed, &C->changedN);
        wave_append(pendNew, x, pendOut, pendCnt);
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) C->evals += n;
}

__global__ __launch_bounds__(WO_BLOCK) void k_fl_apply(FloodDev D, FlLists Ls, FlCtrl* C) {
    if (C->done) return;
    const int32_t n = C->changedN, epoch = C->epoch;
    int32_t* out = Ls.dirty[C->cur ^ 1]; int32_t* outCnt = &C->dirtyN[C->cur ^ 1];
    WO_BLOCK_STRIDE(i, valid, n) {
        int32_t b = 0, deg = 0;
        if (valid) {
            const int32_t x = Ls.changed[i];
            flood_apply_cell(D, x, epoch);
            b = D.off[x]; deg = D.off[x + 1] - b;
        }
        for (int k = 0; __any(k < deg); ++k) {
            bool add = false; int32_t y = -1;
            if (k < deg) { y = D.adj[b + k]; add = D.seedIdx[y] < 0 && atomicExch(&D.inDirty[y], 1) == 0; }
            wave_append(add, y, out, outCnt);
        }
    }
    // the last workgroup to finish closes the round
    __shared__ int s_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(&C->ticket, 1) == (int32_t)gridDim.x - 1);
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    const int32_t next = atomicAdd(outCnt, 0);
    C->changes += n;
    C->dirtyN[C->cur] = 0; C->cur ^= 1; C->changedN = 0; C->rounds += 1; C->ticket = 0;
    if (C->release) { C->pendN[C->relIdx] = 0; C->release = 0; }
    if (next == 0) {
        if (C->pendN[C->pcur] > 0) { C->epoch += 1; C->epochs += 1; C->release = 1; C->relIdx = C->pcur; C->pcur ^= 1; }
        else C->done = 1;
    }
}

__global__ __launch_bounds__(WO_BLOCK) void k_fl_verify(FloodDev D, FlCtrl* C) {
    WO_GRID_STRIDE(i, D.L) {
        bool nf = false, tie = false;
        flood_verify_cell(D, i, &nf, &tie);
        if (nf) atomicAdd(&C->notFixed, 1);
        if (tie) atomicAdd(&C->ties, 1);
    }
}
// tree id of every labelled cell: pointer jumping along the parents to the seed of the tree
__global__ __launch_bounds__(WO_BLOCK) void k_fl_jump_init(FloodDev D, int32_t* jump) {
    WO_GRID_STRIDE(i, D.L) { const int32_t p = D.A[i].par; jump[i] = p >= 0 ? p : i; }
}
__global__ __launch_bounds__(WO_BLOCK) void k_fl_jump(int32_t* jump, int32_t L) {
    WO_GRID_STRIDE(i, L) { const int32_t j = jump[i]; const int32_t jj = jump[j]; if (jj != j) jump[i] = jj; }
}
__global__ __launch_bounds__(WO_BLOCK) void k_fl_export(FloodDev D, const int32_t* jump, int32_t* par, float* surface, int32_t* root) {
    WO_GRID_STRIDE(i, D.L) {
        const FlHead h = D.A[i];
        par[i] = h.par; surface[i] = h.S;
        root[i] = (h.par == FL_NONE) ? -1 : D.seedIdx[jump[i]];
    }
}

}  // namespace wo
