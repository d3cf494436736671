// Research tool (not product, not oracle): the drainage components of the implicit solve and what a component-local
// schedule (one workgroup per component, windows of W consecutive tasks in processing order, in-window chains through LDS)
// would cost.  Dependencies as in csrc/erode_ops.h (latest_event_before): a task's <=3 predecessor events.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <vector>

extern "C" {

static int32_t latest_before(const int32_t* off, const int32_t* adj, const int32_t* target, const int32_t* rank, const uint8_t* ocean, int32_t x, int32_t r) {
    const int32_t rr = rank[r];
    int32_t best = -1, bestRank = -1;
    if (x != r && target[x] >= 0 && rank[x] > rr) { best = x; bestRank = rank[x]; }
    for (int32_t j = off[x]; j < off[x + 1]; ++j) {
        const int32_t n = adj[j];
        if (n == r || ocean[n] || target[n] != x) continue;
        if (rank[n] > rr && (best < 0 || rank[n] < bestRank)) { best = n; bestRank = rank[n]; }
    }
    return best;
}

static int32_t find(std::vector<int32_t>& p, int32_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }

// out: [0] components, [1] largest, [2] cells in components <= W, [3] max DAG depth, [4] estimated us of the slowest workgroup,
// [5] total windows, [6] sum over windows of in-window depth, [7] windows of the largest component, [8] its summed in-window depth
// sizes (optional, L entries): component size per component (descending)
void bs_analyse(int32_t N, const int32_t* off, const int32_t* adj, const int32_t* target, const int32_t* rank, const uint8_t* ocean,
                int32_t W, double usWindow, double usLevel, double* out, int32_t* sizesOut, int32_t nSizes, int32_t* order, int32_t* compStartOut) {
    std::vector<int32_t> land;
    for (int32_t r = 0; r < N; ++r) if (!ocean[r]) land.push_back(r);
    const int32_t L = (int32_t)land.size();
    std::vector<int32_t> par(N);
    std::iota(par.begin(), par.end(), 0);
    for (int32_t r : land) { const int32_t t = target[r]; if (t >= 0 && !ocean[t]) { const int32_t a = find(par, r), b = find(par, t); if (a != b) par[std::max(a, b)] = std::min(a, b); } }
    std::vector<int32_t> byProc(land);                       // processing order: larger rank first
    std::sort(byProc.begin(), byProc.end(), [&](int32_t a, int32_t b) { return rank[a] > rank[b]; });
    std::vector<int32_t> csize(N, 0);
    for (int32_t r : land) ++csize[find(par, r)];
    std::vector<int32_t> roots;
    for (int32_t r : land) if (par[r] == r) roots.push_back(r);
    std::sort(roots.begin(), roots.end(), [&](int32_t a, int32_t b) { return csize[a] != csize[b] ? csize[a] > csize[b] : a < b; });
    std::vector<int32_t> cstart(N, 0);
    { int32_t run = 0; for (int32_t c : roots) { cstart[c] = run; run += csize[c]; } }
    std::vector<int32_t> pos(N, 0), slot(N, -1), ord(L);
    for (int32_t r : byProc) { const int32_t c = find(par, r); const int32_t s = cstart[c] + pos[c]++; slot[r] = s; ord[s] = r; }
    if (order) for (int32_t i = 0; i < L; ++i) order[i] = ord[i];
    // levels (global DAG) and in-window levels
    std::vector<int32_t> level(N, 0), wlevel(N, 0);
    int32_t maxLevel = 0;
    std::vector<int32_t> winDepth;                           // per window (global window index along the layout, windows restart at each big component)
    // windows: a component larger than W is cut into windows of W from its start; small components are packed into windows without straddling
    std::vector<int32_t> winOf(L, 0);
    int32_t nWin = 0;
    {
        int32_t fill = 0;                                     // slots used in the current packing window
        for (int32_t c : roots) {
            const int32_t sz = csize[c], s0 = cstart[c];
            if (sz > W) {
                if (fill) { ++nWin; fill = 0; }
                for (int32_t k = 0; k < sz; ++k) winOf[s0 + k] = nWin + k / W;
                nWin += (sz + W - 1) / W;
            } else {
                if (fill + sz > W) { ++nWin; fill = 0; }
                for (int32_t k = 0; k < sz; ++k) winOf[s0 + k] = nWin;
                fill += sz;
            }
        }
        if (fill) ++nWin;
    }
    winDepth.assign(nWin, 0);
    int64_t badOrder = 0;
    for (int32_t i = 0; i < L; ++i) {
        // process in layout order is not a global topological order across components, but inside a component it is; levels only look inside
    }
    for (int32_t r : byProc) {
        const int32_t t = target[r];
        int32_t preds[3] = {latest_before(off, adj, target, rank, ocean, r, r), -1, -1};
        if (t >= 0 && !ocean[t]) {
            preds[1] = latest_before(off, adj, target, rank, ocean, t, r);
            const int32_t t2 = target[t];
            if (t2 >= 0 && !ocean[t2]) preds[2] = latest_before(off, adj, target, rank, ocean, t2, r);
        }
        int32_t lv = 1, wl = 1;
        const int32_t w = winOf[slot[r]];
        for (int k = 0; k < 3; ++k) {
            const int32_t p = preds[k];
            if (p < 0) continue;
            if (find(par, p) != find(par, r)) ++badOrder;
            lv = std::max(lv, level[p] + 1);
            if (winOf[slot[p]] == w) wl = std::max(wl, wlevel[p] + 1);
        }
        level[r] = lv; wlevel[r] = wl;
        maxLevel = std::max(maxLevel, lv);
        winDepth[w] = std::max(winDepth[w], wl);
    }
    int64_t sumDepth = 0;
    for (int32_t d : winDepth) sumDepth += d;
    // the largest component's windows
    int32_t bigWin = 0; int64_t bigDepth = 0;
    if (!roots.empty()) {
        const int32_t c = roots[0];
        const int32_t w0 = winOf[cstart[c]], w1 = winOf[cstart[c] + csize[c] - 1];
        bigWin = w1 - w0 + 1;
        for (int32_t w = w0; w <= w1; ++w) bigDepth += winDepth[w];
    }
    int64_t small = 0;
    for (int32_t c : roots) if (csize[c] <= W) small += csize[c];
    out[0] = (double)roots.size(); out[1] = roots.empty() ? 0 : csize[roots[0]]; out[2] = (double)small; out[3] = maxLevel;
    out[4] = bigWin * usWindow + bigDepth * usLevel; out[5] = nWin; out[6] = (double)sumDepth; out[7] = bigWin; out[8] = (double)bigDepth; out[9] = (double)badOrder;
    for (int32_t i = 0; i < nSizes && i < (int32_t)roots.size(); ++i) sizesOut[i] = csize[roots[i]];
    (void)compStartOut;
}
}
