// Device formulation of pass 1 of priorityFloodCarve (reference: js/terrain-post.js:107-147): the noise-keyed
// best-first flood as a label-correcting fixed point.  Bodies only (one call = one GPU thread); the kernels are in
// flood_kernels.h, the round driver in planet.hip, and the test-only emulator (tests/emu) drives the same bodies on
// the CPU.
//
// What the reference computes.  A binary heap pops the visited cell with the smallest key K = f32(surface + noise);
// a popped cell c claims its unvisited neighbours nb: drainTo[nb] = c, surface[nb] = max(e[nb], f32(surface[c] + EPS)),
// key from that surface.  The outputs (drainTo, surface) depend on the pop ORDER only through "which neighbour of a
// cell pops first".
//
// The order without the heap.  A child may get a smaller key than its parent and then pops right after it, before
// every other heap entry; so the pop time of a cell is ordered by its LABEL: the stack of suffix maxima of the keys
// along its drain path from the seed (non-increasing keys; the entry of the cell itself last), compared
// lexicographically, a prefix first.  (The first entry is the running maximum tau of the path: cells pop in order of
// tau, cells that share it in order of the next entry, ...)  Hence
//       parent(x) = the neighbour with the smallest label,    surface/key(x) from parent(x),
//       label(x)  = label(parent) with the entries smaller than K(x) popped, then (K(x), x) appended,
// and this system has exactly one fixed point (induction along the pop order): the reference's result.  Equal keys of
// two different cells are the one thing a label cannot order (the reference's heap orders them by its array
// mechanics, a function of its whole history); they are ordered by cell id here and flood_verify_cell counts the
// decisions that needed that (the driver falls back to the serial host walk when exactness is requested).
//
// Reaching the fixed point.  Synchronous rounds over the cells whose neighbourhood changed (Jacobi): a cell
// re-derives its label from the previous round's labels.  Labels are stored BY VALUE (a comparison never chases
// pointers through a structure that is being rewritten).  Because keys depend on the parent's surface a label can
// grow when a cell moves to an earlier-popping parent, so stale labels derived from a cell's old label may still
// circulate and plain Bellman-Ford loops forever (count to infinity; measured).  Loop freedom is DUAL's feasibility
// condition (Garcia-Luna-Aceves 1993): a cell may move to a NEW parent only if that parent's label is smaller than
// the smallest label the cell itself has held since the last reset (its feasible label FD) — every label derived from
// the cell is larger than that.  A cell that wants an infeasible move keeps its parent (re-deriving from it is always
// allowed) and is queued; when a round changes nothing, every label is consistent with its parent's, all FDs are
// reset to the current labels (epoch + 1, lazily) and the queued cells move.  Measured (research/): 3 epochs / 200
// rounds at 10^6 cells, ~12 epochs / ~4000 rounds at 10^7, 6-23 evaluations per land cell.
#pragma once
#include <cstdint>

#include "noise.h"

namespace wo {

constexpr int FL_LD = 32;                 // label stack capacity (deepest seen: 18 at 10^7 cells); overflow -> host walk
constexpr int32_t FL_NONE = -1;           // par: not labelled (unvisited so far / unreachable)
constexpr int32_t FL_SEED = -2;           // par: drains to the open ocean (js/terrain-post.js:118-128)

struct alignas(32) FlHead {
    int32_t par;                          // land index of the parent, FL_NONE, FL_SEED
    float S;                              // surface
    float K;                              // key
    int32_t dep;                          // stack entries (0 when not labelled)
    unsigned long long top;               // stack[0] (copy: most comparisons end here)
    unsigned long long spare;
};

struct FloodDev {
    int32_t L;                            // land cells (compact Morton index space of flood_host.cc)
    const int32_t* off;                   // [L+1] land-only CSR
    const int32_t* adj;
    const int32_t* cell;                  // [L] original region id
    const double* nz;                     // [L] cellNoise(original id) (js/terrain-post.js:100-105)
    const int32_t* seedIdx;               // [L] position in the seed list, -1 for other cells
    const float* e;                       // [L] elevation at call time
    FlHead* A; unsigned long long* Astk;  // current labels
    FlHead* P; unsigned long long* Pstk;  // proposals of the round
    FlHead* F; unsigned long long* Fstk;  // feasible labels (valid when fdEpoch == epoch)
    int32_t* fdEpoch;
    int32_t* inDirty;                     // 1 while the cell sits on the next round's list (set with atomicExch by the kernels)
    uint8_t* isPending;                   // queued for an infeasible move (touched only by the cell's own evaluation)
};

// monotone float -> uint32 (total order of finite floats, -0 == +0 mapped apart but keys are never -0: sums with noise >= 0)
WO_HD inline uint32_t fl_ord(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    return (v.u & 0x80000000u) ? ~v.u : (v.u | 0x80000000u);
}
WO_HD inline unsigned long long fl_pack(float key, int32_t cellId) { return ((unsigned long long)fl_ord(key) << 32) | (uint32_t)cellId; }

// label a < label b ?  (*tie is set when the decision came from the cell ids of two EQUAL keys)
WO_HD inline bool fl_less(const FlHead& ha, const unsigned long long* sa, const FlHead& hb, const unsigned long long* sb, bool* tie) {
    if (ha.top != hb.top) {
        if (tie && (ha.top >> 32) == (hb.top >> 32)) *tie = true;
        return ha.top < hb.top;
    }
    const int n = ha.dep < hb.dep ? ha.dep : hb.dep;
    for (int i = 1; i < n; ++i) {
        const unsigned long long a = sa[i], b = sb[i];
        if (a != b) {
            if (tie && (a >> 32) == (b >> 32)) *tie = true;
            return a < b;
        }
    }
    return ha.dep < hb.dep;               // the prefix pops first
}

// label of x as a child of `src` -> P[x]; returns true when it equals A[x] (nothing to do)
WO_HD inline bool fl_derive(const FloodDev& D, int32_t x, int32_t src, bool* overflow) {
    FlHead h; h.spare = 0;
    unsigned long long* ps = D.Pstk + (size_t)x * FL_LD;
    const FlHead cur = D.A[x];
    const unsigned long long* cs = D.Astk + (size_t)x * FL_LD;
    if (src < 0) {
        h.par = FL_NONE; h.S = D.e[x]; h.K = 0; h.dep = 0; h.top = 0;
        D.P[x] = h;
        return cur.par == FL_NONE;
    }
    const FlHead B = D.A[src];
    const unsigned long long* bs = D.Astk + (size_t)src * FL_LD;
    const double lim = (double)B.S + 1e-7;                                   // :139-142
    h.par = src;
    h.S = ((double)D.e[x] < lim) ? (float)lim : D.e[x];
    h.K = (float)((double)h.S + D.nz[x]);
    const uint32_t kb = fl_ord(h.K);
    int d = B.dep;
    while (d > 0 && (uint32_t)(bs[d - 1] >> 32) < kb) --d;                  // entries with a strictly smaller key are popped
    if (d >= FL_LD) { *overflow = true; d = FL_LD - 1; }
    bool same = cur.par == h.par && cur.S == h.S && cur.K == h.K && cur.dep == d + 1;
    for (int i = 0; i < d; ++i) { const unsigned long long v = bs[i]; ps[i] = v; if (same && cs[i] != v) same = false; }
    const unsigned long long own = fl_pack(h.K, D.cell[x]);
    ps[d] = own;
    h.dep = d + 1;
    h.top = ps[0];
    D.P[x] = h;
    return same;
}

// One evaluation of cell x against the labels of the previous round.  Returns true when P[x] differs from A[x].
// *pendingNew: x wants an infeasible move and was not queued yet (the caller appends it to the pending list).
WO_HD inline bool flood_eval_cell(const FloodDev& D, int32_t x, int32_t epoch, bool force, bool* pendingNew, bool* overflow) {
    if (force) D.isPending[x] = 0;
    const FlHead hx = D.A[x];
    int32_t best = -1; FlHead hb{};
    for (int32_t j = D.off[x]; j < D.off[x + 1]; ++j) {
        const int32_t y = D.adj[j];
        const FlHead hy = D.A[y];
        if (hy.par == FL_NONE || hy.par == x) continue;                      // not labelled / x's own child
        if (best < 0 || fl_less(hy, D.Astk + (size_t)y * FL_LD, hb, D.Astk + (size_t)best * FL_LD, nullptr)) { best = y; hb = hy; }
    }
    int32_t src = best;
    if (best >= 0 && best != hx.par && !force) {
        const FlHead* fh; const unsigned long long* fs;
        if (D.fdEpoch[x] == epoch) { fh = &D.F[x]; fs = D.Fstk + (size_t)x * FL_LD; }
        else { fh = &D.A[x]; fs = D.Astk + (size_t)x * FL_LD; }            // FD = the label held when the epoch began
        if (fh->par != FL_NONE && !fl_less(hb, D.Astk + (size_t)best * FL_LD, *fh, fs, nullptr)) {
            // not feasible: stay with the current parent (if it still carries a label) and queue
            src = (hx.par >= 0 && D.A[hx.par].par != FL_NONE) ? hx.par : -1;
            if (!D.isPending[x]) { D.isPending[x] = 1; *pendingNew = true; }
        }
    }
    return !fl_derive(D, x, src, overflow);
}

// Commit P[x] (x is on the changed list of the round)
WO_HD inline void flood_apply_cell(const FloodDev& D, int32_t x, int32_t epoch) {
    const FlHead nw = D.P[x];
    unsigned long long* as = D.Astk + (size_t)x * FL_LD;
    unsigned long long* fs = D.Fstk + (size_t)x * FL_LD;
    const unsigned long long* ps = D.Pstk + (size_t)x * FL_LD;
    if (D.fdEpoch[x] != epoch) {                                             // first change of the epoch: FD = label at its start
        const FlHead old = D.A[x];
        D.F[x] = old;
        for (int i = 0; i < old.dep; ++i) fs[i] = as[i];
        D.fdEpoch[x] = epoch;
    }
    if (nw.par != FL_NONE) {
        const FlHead f = D.F[x];
        if (f.par == FL_NONE || fl_less(nw, ps, f, fs, nullptr)) { D.F[x] = nw; for (int i = 0; i < nw.dep; ++i) fs[i] = ps[i]; }
    }
    D.A[x] = nw;
    for (int i = 0; i < nw.dep; ++i) as[i] = ps[i];
}

// After convergence: is x's parent the smallest-label neighbour (fixed point), and did that need the id order of two
// equal keys (a decision the reference's heap takes by its array mechanics)?
WO_HD inline void flood_verify_cell(const FloodDev& D, int32_t x, bool* notFixed, bool* tieDecided) {
    const FlHead hx = D.A[x];
    if (hx.par == FL_SEED) return;
    int32_t best = -1; FlHead hb{};
    for (int32_t j = D.off[x]; j < D.off[x + 1]; ++j) {
        const int32_t y = D.adj[j];
        const FlHead hy = D.A[y];
        if (hy.par == FL_NONE || hy.par == x) continue;
        if (best < 0 || fl_less(hy, D.Astk + (size_t)y * FL_LD, hb, D.Astk + (size_t)best * FL_LD, nullptr)) { best = y; hb = hy; }
    }
    if ((best < 0) != (hx.par == FL_NONE) || (best >= 0 && best != hx.par)) { *notFixed = true; return; }
    if (best < 0) return;
    for (int32_t j = D.off[x]; j < D.off[x + 1]; ++j) {
        const int32_t y = D.adj[j];
        if (y == best) continue;
        const FlHead hy = D.A[y];
        if (hy.par == FL_NONE) continue;
        // children of x count too: whether y pops before or after x's parent decides who claims x
        bool tie = false;
        (void)fl_less(hb, D.Astk + (size_t)best * FL_LD, hy, D.Astk + (size_t)y * FL_LD, &tie);
        if (tie) *tieDecided = true;
    }
}

}  // namespace wo
