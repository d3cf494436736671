// Research tool (not product, not oracle): structure of the reference's noise-keyed priority flood
// (js/terrain-post.js:59-147) — what a parallel order-equivalent formulation would have to cope with.
// Runs the serial walk with the reference heap, then measures units (bursts), tau windows and in-window chain depths.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

static double cell_noise(int32_t r) {
    const double p = (double)r * 2654435761.0;
    uint32_t h = (uint32_t)(uint64_t)p;
    const int32_t x = (int32_t)((h >> 16) ^ h);
    const double q = (double)x * 73244475.0;
    h = (uint32_t)(int64_t)q;
    h = (h >> 16) ^ h;
    return ((double)h / 4294967295.0) * 0.01;
}

struct Heap {
    std::vector<int32_t> d; const float* key;
    void push(int32_t c) {
        size_t i = d.size(); d.push_back(c);
        while (i > 0) { size_t p = (i - 1) >> 1; if (key[d[i]] >= key[d[p]]) break; std::swap(d[i], d[p]); i = p; }
    }
    int32_t pop() {
        int32_t top = d[0]; int32_t last = d.back(); d.pop_back();
        const size_t n = d.size();
        if (n > 0) {
            d[0] = last; size_t i = 0;
            for (;;) {
                size_t s = i, l = 2 * i + 1, r = 2 * i + 2;
                if (l < n && key[d[l]] < key[d[s]]) s = l;
                if (r < n && key[d[r]] < key[d[s]]) s = r;
                if (s == i) break;
                std::swap(d[i], d[s]); i = s;
            }
        }
        return top;
    }
};

extern "C" int flood_structure(int32_t N, const int32_t* off, const int32_t* adj, const float* e, const uint8_t* ocean,
                               int32_t nDelta, const double* deltas, double* out /* 64 + 8*nDelta doubles */,
                               int32_t* outT, int32_t* outParent, float* outS, float* outK) {
    const double EPS = 1e-7;
    std::vector<int32_t> label(N, -1), stack;
    std::vector<int64_t> compSize;
    for (int32_t r = 0; r < N; ++r) {
        if (!ocean[r] || label[r] >= 0) continue;
        const int32_t lab = (int32_t)compSize.size(); int64_t sz = 0;
        stack.push_back(r); label[r] = lab;
        while (!stack.empty()) { int32_t c = stack.back(); stack.pop_back(); ++sz; for (int32_t i = off[c]; i < off[c + 1]; ++i) { int32_t nb = adj[i]; if (ocean[nb] && label[nb] < 0) { label[nb] = lab; stack.push_back(nb); } } }
        compSize.push_back(sz);
    }
    int32_t mainLab = 0; for (size_t i = 1; i < compSize.size(); ++i) if (compSize[i] > compSize[mainLab]) mainLab = (int32_t)i;
    std::vector<float> S(e, e + N), K(N);
    std::vector<int32_t> parent(N, -1), T(N, -1);
    std::vector<uint8_t> vis(N, 0);
    for (int32_t r = 0; r < N; ++r) K[r] = (float)((double)e[r] + cell_noise(r));
    Heap H; H.key = K.data();
    int64_t L = 0, nSeeds = 0;
    for (int32_t r = 0; r < N; ++r) {
        if (ocean[r]) { vis[r] = 1; continue; }
        ++L;
        for (int32_t i = off[r]; i < off[r + 1]; ++i) if (ocean[adj[i]] && label[adj[i]] == mainLab) { vis[r] = 1; parent[r] = -2; H.push(r); ++nSeeds; break; }
    }
    int32_t t = 0; size_t maxHeap = 0;
    std::vector<int32_t> order; order.reserve(L);
    int64_t consecutiveEqual = 0; float lastKey = -1e30f;
    while (!H.d.empty()) {
        maxHeap = std::max(maxHeap, H.d.size());
        const int32_t c = H.pop();
        if (K[c] == lastKey) ++consecutiveEqual;
        lastKey = K[c];
        T[c] = t++; order.push_back(c);
        const double lim = (double)S[c] + EPS;
        for (int32_t i = off[c]; i < off[c + 1]; ++i) {
            const int32_t nb = adj[i];
            if (vis[nb]) continue;
            vis[nb] = 1; parent[nb] = c;
            if ((double)e[nb] < lim) { S[nb] = (float)lim; K[nb] = (float)((double)S[nb] + cell_noise(nb)); }
            H.push(nb);
        }
    }
    const int64_t popped = (int64_t)order.size();
    // tau = path max of K; unit root = nearest ancestor-or-self whose K equals tau chain start (K >= tau(parent))
    std::vector<float> tau(N, 0);
    std::vector<int32_t> depth(N, 0), unitRoot(N, -1), unitSize(N, 0), hopInUnit(N, 0);
    int32_t maxDepth = 0; int64_t nUnits = 0, flooded = 0;
    int32_t maxHopInUnit = 0;
    for (int32_t c : order) {
        const int32_t p = parent[c];
        if ((double)S[c] > (double)e[c]) ++flooded;
        if (p < 0) { tau[c] = K[c]; depth[c] = 1; unitRoot[c] = c; ++nUnits; hopInUnit[c] = 0; }
        else {
            depth[c] = depth[p] + 1;
            if (K[c] >= tau[p]) { tau[c] = K[c]; unitRoot[c] = c; ++nUnits; hopInUnit[c] = 0; }
            else { tau[c] = tau[p]; unitRoot[c] = unitRoot[p]; hopInUnit[c] = hopInUnit[p] + 1; }
        }
        unitSize[unitRoot[c]]++;
        maxDepth = std::max(maxDepth, depth[c]);
        maxHopInUnit = std::max(maxHopInUnit, hopInUnit[c]);
    }
    // verify: T order == sorted by (tau, then within unit by T) i.e. tau non-decreasing along pop order
    int64_t tauViol = 0; { float prev = -1e30f; for (int32_t c : order) { if (tau[c] < prev) ++tauViol; prev = std::max(prev, tau[c]); } }
    int32_t maxUnit = 0; int64_t u8 = 0, u64 = 0, u1k = 0, u16k = 0, cellsInBig = 0;
    for (int32_t c : order) if (unitRoot[c] == c) { const int32_t s = unitSize[c]; maxUnit = std::max(maxUnit, s); if (s > 8) ++u8; if (s > 64) { ++u64; cellsInBig += s; } if (s > 1024) ++u1k; if (s > 16384) ++u16k; }
    // interacting ties: equal keys among cells within two hops that were in the heap at overlapping times
    int64_t tiePairs2hop = 0;
    for (int32_t c : order) {
        for (int32_t i = off[c]; i < off[c + 1]; ++i) {
            const int32_t x = adj[i]; if (ocean[x]) continue;
            if (x > c && K[x] == K[c]) ++tiePairs2hop;
            for (int32_t j = off[x]; j < off[x + 1]; ++j) { const int32_t y = adj[j]; if (y > c && !ocean[y] && y != x && K[y] == K[c]) ++tiePairs2hop; }
        }
    }
    out[0] = (double)L; out[1] = (double)nSeeds; out[2] = (double)popped; out[3] = (double)maxHeap; out[4] = (double)maxDepth;
    out[5] = (double)nUnits; out[6] = (double)maxUnit; out[7] = (double)u8; out[8] = (double)u64; out[9] = (double)u1k; out[10] = (double)u16k;
    out[11] = (double)cellsInBig; out[12] = (double)tauViol; out[13] = (double)consecutiveEqual; out[14] = (double)tiePairs2hop; out[15] = (double)flooded;
    out[16] = (double)maxHopInUnit;
    float tmin = 1e30f, tmax = -1e30f; for (int32_t c : order) { tmin = std::min(tmin, tau[c]); tmax = std::max(tmax, tau[c]); }
    out[17] = tmin; out[18] = tmax;
    // windows by tau
    for (int32_t d = 0; d < nDelta; ++d) {
        const double D = deltas[d];
        const int64_t nW = (int64_t)std::floor(((double)tmax - (double)tmin) / D) + 1;
        std::vector<int32_t> wmaxChain(nW, 0), wPops(nW, 0);
        std::vector<int32_t> chain(N, 0);
        for (int32_t c : order) {
            const int64_t w = (int64_t)std::floor(((double)tau[c] - (double)tmin) / D);
            const int32_t p = parent[c];
            int32_t h = 1;
            if (p >= 0) { const int64_t wp = (int64_t)std::floor(((double)tau[p] - (double)tmin) / D); if (wp == w) h = chain[p] + 1; }
            chain[c] = h; wPops[w]++; wmaxChain[w] = std::max(wmaxChain[w], h);
        }
        int64_t sumChain = 0, nonEmpty = 0; int32_t maxChain = 0, maxPops = 0;
        for (int64_t w = 0; w < nW; ++w) { if (wPops[w]) { ++nonEmpty; sumChain += wmaxChain[w] + 1; } maxChain = std::max(maxChain, wmaxChain[w]); maxPops = std::max(maxPops, wPops[w]); }
        double* o = out + 64 + 8 * d;
        o[0] = D; o[1] = (double)nW; o[2] = (double)nonEmpty; o[3] = (double)sumChain; o[4] = (double)maxChain; o[5] = (double)maxPops;
    }
    if (outT) std::memcpy(outT, T.data(), sizeof(int32_t) * N);
    if (outParent) std::memcpy(outParent, parent.data(), sizeof(int32_t) * N);
    if (outS) std::memcpy(outS, S.data(), sizeof(float) * N);
    if (outK) std::memcpy(outK, K.data(), sizeof(float) * N);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Prototype of the label-correcting (Bellman-Ford style) formulation of pass 1.
// A cell's pop time is ordered by its LABEL: the non-increasing stack of (key, cell) along its drain path from the
// seed (suffix maxima); parent(x) = the neighbour with the smallest label; the unique fixed point of
//   parent(x) = argmin label(y),  S(x), K(x) from parent,  label(x) = extend(label(parent), K(x))
// is the serial result (ties between equal keys of different cells aside).  Jacobi rounds from "nothing labelled".
// Labels are stored by value (depth <= LD): a comparison never chases pointers of a changing structure.
// ---------------------------------------------------------------------------------------------------------
constexpr int LD = 24;
struct Lab { int32_t par; float S, K; int32_t dep; float k[LD]; int32_t c[LD]; };
static int64_t g_walks = 0, g_walkSteps = 0, g_walkReject = 0, g_maxWalk = 0;
static int64_t g_ties = 0, g_deep = 0;
static int labcmp(const Lab& a, const Lab& b) {        // -1: a pops first
    const int n = a.dep < b.dep ? a.dep : b.dep;
    for (int i = 0; i < n; ++i) {
        if (a.k[i] != b.k[i]) { if (i > 0) ++g_deep; return a.k[i] < b.k[i] ? -1 : 1; }
        if (a.c[i] == b.c[i]) continue;
        ++g_ties; return a.c[i] < b.c[i] ? -1 : 1;
    }
    if (a.dep == b.dep) return 0;
    return a.dep < b.dep ? -1 : 1;
}
static bool labsame(const Lab& a, const Lab& b) {
    if (a.par != b.par || a.S != b.S || a.K != b.K || a.dep != b.dep) return false;
    for (int i = 0; i < a.dep; ++i) if (a.c[i] != b.c[i] || a.k[i] != b.k[i]) return false;
    return true;
}

static void derive(Lab& P, const Lab& B, int32_t best, int32_t x, const float* e, int64_t& overflow) {
    const double EPS = 1e-7;
    const double lim = (double)B.S + EPS;
    P.par = best;
    P.S = ((double)e[x] < lim) ? (float)lim : e[x];
    P.K = (float)((double)P.S + cell_noise(x));
    int d = B.dep;
    while (d > 0 && B.k[d - 1] < P.K) --d;          // pop the strictly smaller keys
    for (int i = 0; i < d; ++i) { P.k[i] = B.k[i]; P.c[i] = B.c[i]; }
    if (d >= LD) { ++overflow; d = LD - 1; }
    P.k[d] = P.K; P.c[d] = x; P.dep = d + 1;
}
extern "C" int flood_bf(int32_t N, const int32_t* off, const int32_t* adj, const float* e, const uint8_t* ocean,
                        const int32_t* refParent, const float* refS, const float* refK, double* out, double DELTA, int32_t STALL, int32_t verbose) {
    std::vector<int32_t> label(N, -1), stack; std::vector<int64_t> compSize;
    for (int32_t r = 0; r < N; ++r) {
        if (!ocean[r] || label[r] >= 0) continue;
        const int32_t lab = (int32_t)compSize.size(); int64_t sz = 0;
        stack.push_back(r); label[r] = lab;
        while (!stack.empty()) { int32_t c = stack.back(); stack.pop_back(); ++sz; for (int32_t i = off[c]; i < off[c + 1]; ++i) { int32_t nb = adj[i]; if (ocean[nb] && label[nb] < 0) { label[nb] = lab; stack.push_back(nb); } } }
        compSize.push_back(sz);
    }
    int32_t mainLab = 0; for (size_t i = 1; i < compSize.size(); ++i) if (compSize[i] > compSize[mainLab]) mainLab = (int32_t)i;
    std::vector<Lab> A(N), FD(N);
    std::vector<uint8_t> hasFD(N, 0);
    for (int32_t r = 0; r < N; ++r) { A[r].par = -1; A[r].dep = 0; A[r].S = e[r]; A[r].K = 0; }
    std::vector<uint8_t> isSeed(N, 0), inDirty(N, 0), isPending(N, 0), forceSwitch(N, 0);
    std::vector<int32_t> dirty, next, pending;
    for (int32_t r = 0; r < N; ++r) {
        if (ocean[r]) continue;
        for (int32_t i = off[r]; i < off[r + 1]; ++i) if (ocean[adj[i]] && label[adj[i]] == mainLab) { isSeed[r] = 1; break; }
        if (isSeed[r]) { Lab& a = A[r]; a.par = -2; a.K = (float)((double)e[r] + cell_noise(r)); a.dep = 1; a.k[0] = a.K; a.c[0] = r; }
    }
    for (int32_t r = 0; r < N; ++r) if (isSeed[r]) for (int32_t i = off[r]; i < off[r + 1]; ++i) { const int32_t nb = adj[i]; if (!ocean[nb] && !isSeed[nb] && !inDirty[nb]) { inDirty[nb] = 1; dirty.push_back(nb); } }
    std::vector<std::pair<int32_t, Lab>> props;
    int64_t rounds = 0, evals = 0, changes = 0, maxDirty = 0, overflow = 0, resets = 0, pendingTotal = 0; int32_t maxDep = 0;
    const bool throttle = DELTA > 0 && DELTA < 5;
    std::vector<int32_t> waiting, waiting2; std::vector<uint8_t> inWait(N, 0);
    double Chor = throttle ? -1e30 : 1e30, lastMin = 1e30;
    if (throttle) { for (int32_t r = 0; r < N; ++r) if (isSeed[r]) { waiting.push_back(r); inWait[r] = 1; } dirty.clear(); std::fill(inDirty.begin(), inDirty.end(), 0); }
    for (;;) {
        if (throttle) {
            // horizon: lowest tau still moving (changed last round / pending / waiting) + DELTA; never decreases
            double m = lastMin;
            for (int32_t w : waiting) if (A[w].par != -1) m = std::min(m, (double)A[w].k[0]);
            if (m < 1e29) Chor = std::max(Chor, m + DELTA);
            waiting2.clear();
            for (int32_t w : waiting) {
                if (A[w].par == -1) { inWait[w] = 0; continue; }
                if ((double)A[w].k[0] < Chor) { inWait[w] = 0; for (int32_t i = off[w]; i < off[w + 1]; ++i) { const int32_t nb = adj[i]; if (!ocean[nb] && !isSeed[nb] && !inDirty[nb]) { inDirty[nb] = 1; dirty.push_back(nb); } } }
                else waiting2.push_back(w);
            }
            waiting.swap(waiting2);
        }
        if (dirty.empty() && pending.empty() && !waiting.empty()) { lastMin = 1e30; ++rounds; if (rounds > 60000) break; continue; }
        if (dirty.empty()) {
            if (pending.empty()) break;
            // global quiescence: every label is consistent with its parent's; pending switches are safe now
            ++resets; pendingTotal += (int64_t)pending.size();
            if (verbose) fprintf(stderr, "reset %lld at round %lld: %zu pending\n", (long long)resets, (long long)rounds, pending.size());
            for (int32_t r = 0; r < N; ++r) if (!ocean[r] && A[r].par != -1) { FD[r] = A[r]; hasFD[r] = 1; }
            for (int32_t x : pending) { isPending[x] = 0; forceSwitch[x] = 1; if (!inDirty[x]) { inDirty[x] = 1; dirty.push_back(x); } }
            pending.clear();
        }
        ++rounds; maxDirty = std::max<int64_t>(maxDirty, (int64_t)dirty.size());
        props.clear();
        for (int32_t x : dirty) {
            ++evals;
            int32_t best = -1;
            for (int32_t i = off[x]; i < off[x + 1]; ++i) {
                const int32_t y = adj[i];
                if (ocean[y] || A[y].par == -1) continue;
                if (A[y].par == x) continue;            // x's child cannot be its parent
                if (!((double)A[y].k[0] < Chor)) continue;   // beyond the horizon: not yet a parent
                if (best < 0 || labcmp(A[y], A[best]) < 0) best = y;
            }
            Lab P; P.par = -1; P.dep = 0; P.S = e[x]; P.K = 0;
            bool pend = false;
            if (best >= 0) {
                const int32_t cur = A[x].par;
                if (STALL != 0 && best != cur && hasFD[x] && !forceSwitch[x] && !(labcmp(A[best], FD[x]) < 0)) {
                    // not feasible: stay with the current parent (re-derived) if it is still labelled, else drop the label
                    pend = true;
                    if (cur >= 0 && A[cur].par != -1) derive(P, A[cur], cur, x, e, overflow);
                } else derive(P, A[best], best, x, e, overflow);
            }
            if (pend && !isPending[x]) { isPending[x] = 1; pending.push_back(x); }
            if (!labsame(P, A[x])) props.push_back({x, P});
        }
        for (int32_t x : dirty) { inDirty[x] = 0; forceSwitch[x] = 0; }
        next.clear();
        if (false) {
            fprintf(stderr, "== round %lld: %zu props\n", (long long)rounds, props.size());
            int shown = 0;
            for (auto& pr : props) { if (shown++ > 40) break; const Lab& o = A[pr.first]; const Lab& n = pr.second;
                fprintf(stderr, "  x %d: par %d->%d S %.9g->%.9g K %.9g->%.9g dep %d->%d top (%.9g,%d)->(%.9g,%d) pend %d\n", pr.first, o.par, n.par, o.S, n.S, o.K, n.K, o.dep, n.dep,
                        o.dep ? o.k[0] : 0.f, o.dep ? o.c[0] : -1, n.dep ? n.k[0] : 0.f, n.dep ? n.c[0] : -1, (int)isPending[pr.first]); }
        }
        lastMin = 1e30;
        for (auto& pr : props) {
            ++changes;
            const int32_t x = pr.first;
            if (A[x].par != -1) lastMin = std::min(lastMin, (double)A[x].k[0]);
            if (pr.second.par != -1) lastMin = std::min(lastMin, (double)pr.second.k[0]);
            if (throttle && pr.second.par != -1 && !((double)pr.second.k[0] < Chor) && !inWait[x]) { inWait[x] = 1; waiting.push_back(x); }
            A[x] = pr.second;
            if (pr.second.par >= 0 && (!hasFD[x] || labcmp(pr.second, FD[x]) < 0)) { FD[x] = pr.second; hasFD[x] = 1; }
            maxDep = std::max(maxDep, pr.second.dep);
            for (int32_t i = off[x]; i < off[x + 1]; ++i) { const int32_t nb = adj[i]; if (!ocean[nb] && !isSeed[nb] && !inDirty[nb]) { inDirty[nb] = 1; next.push_back(nb); } }
        }
        dirty.swap(next);
        if (verbose && rounds % 100 == 0) fprintf(stderr, "round %lld dirty %zu pending %zu changes %lld\n", (long long)rounds, dirty.size(), pending.size(), (long long)changes);
        if (rounds > 60000) break;
    }
    out[15] = (double)resets; out[16] = (double)pendingTotal;
    int64_t mismatchParent = 0, mismatchS = 0, mismatchK = 0;
    for (int32_t r = 0; r < N; ++r) {
        if (ocean[r]) continue;
        if (A[r].par != refParent[r]) ++mismatchParent;
        if (A[r].par != -1 && A[r].S != refS[r]) ++mismatchS;
        if (A[r].par != -1 && A[r].K != refK[r]) ++mismatchK;
    }
    out[0] = (double)rounds; out[1] = (double)evals; out[2] = (double)changes; out[3] = (double)maxDirty; out[4] = (double)mismatchParent;
    out[5] = (double)mismatchS; out[6] = (double)mismatchK; out[7] = (double)g_ties; out[8] = (double)g_deep; out[9] = (double)maxDep; out[10] = (double)overflow; out[11] = (double)g_walks; out[12] = (double)g_walkSteps; out[13] = (double)g_walkReject; out[14] = (double)g_maxWalk;
    g_walks = g_walkSteps = g_walkReject = g_maxWalk = 0; g_ties = g_deep = 0;
    return 0;
}
