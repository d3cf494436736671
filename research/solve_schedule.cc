// Research tool (not product, not oracle): how many launches would the patch solve (k_solve_patch) need for a given
// patch order?  Builds the drainage forest and the solve's dependency edges (js/terrain-post.js:566-641 as restated
// in csrc/erode_ops.h: latest_event_before) on the host and evaluates
//     launch(r) = max over predecessors p of launch(p) + (patch(p) != patch(r)),   launch >= 1,
// which is what the kernel's rule "external granules must come from an earlier launch" amounts to when a visit runs
// every in-patch chain to its end.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <vector>

extern "C" {

// e: elevations after the flood; land list in ascending id; out: target[N], rank[N] (index in the descending stable sort)
void ss_receivers(int32_t N, const int32_t* off, const int32_t* adj, const float* e, const uint8_t* ocean, int32_t* target, int32_t* rank) {
    std::vector<int32_t> land;
    for (int32_t r = 0; r < N; ++r) if (!ocean[r]) land.push_back(r);
    std::stable_sort(land.begin(), land.end(), [&](int32_t a, int32_t b) { return e[a] > e[b]; });
    for (int32_t r = 0; r < N; ++r) rank[r] = -1;
    for (size_t i = 0; i < land.size(); ++i) rank[land[i]] = (int32_t)i;
    for (int32_t r = 0; r < N; ++r) {
        target[r] = -1;
        if (ocean[r]) continue;
        const double h = e[r];
        int32_t best = -1; double bestDrop = -INFINITY;
        for (int32_t j = off[r]; j < off[r + 1]; ++j) { const double d = h - (double)e[adj[j]]; if (d > bestDrop) { bestDrop = d; best = adj[j]; } }
        if (bestDrop <= 0) { double mn = INFINITY; for (int32_t j = off[r]; j < off[r + 1]; ++j) { const double a = (double)e[adj[j]] - h; if (a < mn) { mn = a; best = adj[j]; } } }
        target[r] = best;
    }
}

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

// slotOf[N]: position of each land cell in the patch order; patch = slot / patchCells.
// out[0] = launches needed, out[1] = DAG depth (levels), out[2] = cross-patch edges, out[3] = edges,
// hist[k] (k < nHist) = tasks that complete in launch k+1
void ss_launches(int32_t N, const int32_t* off, const int32_t* adj, const int32_t* target, const int32_t* rank, const uint8_t* ocean,
                 const int32_t* slotOf, int32_t patchCells, double* out, int64_t* hist, int32_t nHist) {
    std::vector<int32_t> byRank;
    for (int32_t r = 0; r < N; ++r) if (!ocean[r]) byRank.push_back(r);
    std::sort(byRank.begin(), byRank.end(), [&](int32_t a, int32_t b) { return rank[a] > rank[b]; });   // processing order: largest rank first
    std::vector<int32_t> launch(N, 0), level(N, 0);
    int64_t cross = 0, edges = 0; int32_t maxLaunch = 0, maxLevel = 0;
    for (int32_t r : byRank) {
        const int32_t t = target[r];
        int32_t preds[3] = {latest_before(off, adj, target, rank, ocean, r, r), -1, -1};
        if (t >= 0 && !ocean[t]) {
            preds[1] = latest_before(off, adj, target, rank, ocean, t, r);
            const int32_t t2 = target[t];
            if (t2 >= 0 && !ocean[t2]) preds[2] = latest_before(off, adj, target, rank, ocean, t2, r);
        }
        int32_t la = 1, lv = 1;
        const int32_t pr = slotOf[r] / patchCells;
        for (int k = 0; k < 3; ++k) {
            const int32_t p = preds[k];
            if (p < 0) continue;
            ++edges;
            const bool x = slotOf[p] / patchCells != pr;
            cross += x;
            la = std::max(la, launch[p] + (x ? 1 : 0));
            lv = std::max(lv, level[p] + 1);
        }
        launch[r] = la; level[r] = lv;
        maxLaunch = std::max(maxLaunch, la); maxLevel = std::max(maxLevel, lv);
        if (la - 1 < nHist) ++hist[la - 1];
    }
    out[0] = maxLaunch; out[1] = maxLevel; out[2] = (double)cross; out[3] = (double)edges;
}

// the river order of csrc/planet.hip (river_patch_order), variant selects experiments
void ss_river_order(int32_t N, const int32_t* target, const uint8_t* ocean, int32_t variant, int32_t* order, double* info, const int32_t* rank) {
    std::vector<int32_t> landAsc;
    for (int32_t r = 0; r < N; ++r) if (!ocean[r]) landAsc.push_back(r);
    const int32_t L = (int32_t)landAsc.size();
    std::vector<int32_t> cnt((size_t)N + 1, 0), child((size_t)L), sz((size_t)N, 1), bfs;
    // variant & 4: the forest of EARLY edges only (donor ranked before its receiver, the edges the flow accumulation forwards along): acyclic by construction
    auto parent = [&](int32_t r) { const int32_t t = target[r]; if (t < 0 || ocean[t]) return -1; if ((variant & 4) && !(rank[r] < rank[t])) return -1; return t; };
    for (int32_t r : landAsc) { const int32_t t = parent(r); if (t >= 0) ++cnt[t + 1]; }
    for (int32_t r = 0; r < N; ++r) cnt[r + 1] += cnt[r];
    { std::vector<int32_t> pos(cnt.begin(), cnt.end() - 1); for (int32_t r : landAsc) { const int32_t t = parent(r); if (t >= 0) child[pos[t]++] = r; } }
    std::vector<uint8_t> seen((size_t)N, 0), isRoot((size_t)N, 0);
    std::vector<int32_t> rootsAll, mark((size_t)N, -1);
    for (int32_t r : landAsc) if (parent(r) < 0) { bfs.push_back(r); seen[r] = 1; isRoot[r] = 1; rootsAll.push_back(r); }
    size_t head = 0;
    auto expand = [&]() { for (; head < bfs.size(); ++head) { const int32_t c = bfs[head]; for (int32_t j = cnt[c]; j < cnt[c + 1]; ++j) { const int32_t d = child[j]; if (!seen[d]) { seen[d] = 1; bfs.push_back(d); } } } };
    expand();
    int32_t closed = 0, pseudo = 0;
    // closed basins: every cell drains into a cycle (pits drain to their least-steep-ascent neighbour); one cell of the cycle stands in as the outlet
    for (int32_t u : landAsc) {
        if (seen[u]) continue;
        int32_t v = u;
        while (mark[v] != u) { mark[v] = u; v = parent(v); }
        seen[v] = 1; isRoot[v] = 1; rootsAll.push_back(v); bfs.push_back(v); ++pseudo;
        const size_t before = bfs.size();
        expand();
        closed += (int32_t)(bfs.size() - before) + 1;
    }
    for (size_t h = bfs.size(); h-- > 0;) { const int32_t c = bfs[h]; if (!isRoot[c]) sz[parent(c)] += sz[c]; }
    for (size_t h = 0; h < bfs.size(); ++h) { const int32_t c = bfs[h]; std::stable_sort(child.begin() + cnt[c], child.begin() + cnt[c + 1], [&](int32_t a, int32_t b) { return sz[a] > sz[b]; }); }
    const size_t nRoots = rootsAll.size();
    int32_t n = 0;
    std::vector<int32_t> stack;
    std::vector<int32_t> roots(rootsAll);
    if (variant & 1) std::stable_sort(roots.begin(), roots.end(), [&](int32_t a, int32_t b) { return sz[a] > sz[b]; });
    for (int32_t root : roots) {
        if (variant & 2) {           // plain preorder, heavy first
            stack.push_back(root);
            while (!stack.empty()) { const int32_t c = stack.back(); stack.pop_back(); order[n++] = c; for (int32_t j = cnt[c + 1]; j-- > cnt[c];) if (!isRoot[child[j]]) stack.push_back(child[j]); }
            continue;
        }
        order[n++] = root;
        stack.push_back(root);
        while (!stack.empty()) {
            const int32_t c = stack.back(); stack.pop_back();
            for (int32_t j = cnt[c]; j < cnt[c + 1]; ++j) if (!isRoot[child[j]]) order[n++] = child[j];
            for (int32_t j = cnt[c + 1]; j-- > cnt[c];) if (!isRoot[child[j]] && cnt[child[j] + 1] > cnt[child[j]]) stack.push_back(child[j]);
        }
    }
    const int32_t unreached = closed; (void)pseudo;
    int32_t maxSz = 0; for (int32_t r : roots) maxSz = std::max(maxSz, sz[r]);
    info[0] = (double)nRoots; info[1] = unreached; info[2] = maxSz; info[3] = n;
}

// The order as the device builds it (csrc: k_river_*): early-forest subtree sizes (what the flow accumulation leaves in
// accA), pits attached to the neighbour they drain up to (2-cycles cut at the pit), keys = layout positions computed
// with those (not quite consistent) sizes, then a sort by key.  Always a permutation; overlaps only cost locality.
void ss_river_keys(int32_t N, const int32_t* off, const int32_t* adj, const int32_t* target, const int32_t* rank, const uint8_t* ocean, int32_t rounds, int32_t* order, double* info) {
    std::vector<int32_t> land;
    for (int32_t r = 0; r < N; ++r) if (!ocean[r]) land.push_back(r);
    const int32_t L = (int32_t)land.size();
    auto early = [&](int32_t r) { const int32_t t = target[r]; return t >= 0 && !ocean[t] && rank[r] < rank[t]; };
    // early-forest subtree sizes (descending elevation = ascending rank order: donors before receivers)
    std::vector<int32_t> byRank(land);
    std::sort(byRank.begin(), byRank.end(), [&](int32_t a, int32_t b) { return rank[a] < rank[b]; });
    std::vector<uint32_t> acc(N, 0);
    for (int32_t r : land) acc[r] = 1;
    for (int32_t r : byRank) if (early(r)) acc[target[r]] += acc[r];
    auto par = [&](int32_t c) {
        const int32_t t = target[c];
        if (t < 0 || ocean[t]) return -1;
        if (target[t] == c && rank[c] > rank[t]) return -1;         // 2-cycle: cut at the pit
        return t;
    };
    std::vector<int64_t> A(N, 0), A2(N, 0); std::vector<int32_t> J(N, -1), J2(N, -1), idx(N, 0);
    int64_t base = 0; int32_t nroots = 0;
    for (int32_t c : land) {
        const int32_t p = par(c);
        if (p < 0) { A[c] = base + 1; idx[c] = -1; J[c] = -1; base += acc[c]; ++nroots; A2[c] = base - acc[c]; continue; }   // A2 temporarily: the root's own key
        int32_t k = 0, i = 0; int64_t before = 0; bool passed = false;
        for (int32_t j = off[p]; j < off[p + 1]; ++j) {
            const int32_t s = adj[j];
            if (ocean[s] || par(s) != p) continue;
            ++k;
            if (s == c) { passed = true; continue; }
            if (acc[s] > acc[c] || (acc[s] == acc[c] && !passed)) { ++i; before += acc[s] - 1; }
        }
        A[c] = k + before; idx[c] = i; J[c] = p;
    }
    std::vector<int64_t> rootKey(N, 0);
    for (int32_t c : land) if (idx[c] < 0) rootKey[c] = A2[c];
    int32_t active = 0;
    for (int r = 0; r < rounds; ++r) {
        active = 0;
        for (int32_t c : land) { const int32_t j = J[c]; if (j >= 0) { A2[c] = A[c] + A[j]; J2[c] = J[j]; ++active; } else { A2[c] = A[c]; J2[c] = -1; } }
        A.swap(A2); J.swap(J2);
    }
    std::vector<std::pair<int64_t, int32_t>> kv(L);
    for (int32_t i = 0; i < L; ++i) { const int32_t c = land[i]; const int32_t p = par(c); kv[i] = {p < 0 ? rootKey[c] : A[p] + idx[c], c}; }
    std::stable_sort(kv.begin(), kv.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    for (int32_t i = 0; i < L; ++i) order[i] = kv[i].second;
    info[0] = nroots; info[1] = active; info[2] = (double)base; info[3] = (double)kv.back().first;
}
}
