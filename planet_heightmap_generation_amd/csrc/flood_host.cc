// priorityFloodCarve — host-resident stage of the erosion stack (reference: js/terrain-post.js:59-215).
//
// Why this one stage runs on the host in this round: the noise-perturbed best-first flood pops cells in an
// order that is defined by a binary heap over float keys (ties included), pass 2 carves along drain paths
// sequentially in ascending cell order reading already-carved heights, and SURVEY §6.4 measured that
// relaxing either changes the result by ~1e-2 RMS.  A serial walk on one GPU lane would take seconds per
// call, so the two calls per erodeComposite run here on one host core between device phases (the field
// makes one D2H + H2D round trip per call).  DESIGN.md lists an order-equivalent device flood as the next step.
//
// This is product code (it is the designed path, it is not a fallback and it does not touch oracle/).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <atomic>
#include <numeric>
#include <thread>

#include "host_util.h"
#include "wo_internal.h"

namespace wo {

namespace {

// js/terrain-post.js:100-105 — Number (double) products reduced mod 2^32 afterwards
inline double cell_noise(int32_t r) {
    const double p = (double)r * 2654435761.0;
    uint32_t h = (uint32_t)(uint64_t)p;
    const int32_t x = (int32_t)((h >> 16) ^ h);
    const double q = (double)x * 73244475.0;
    h = (uint32_t)(int64_t)q;
    h = (h >> 16) ^ h;
    return ((double)h / 4294967295.0) * 0.01;
}

// Binary min-heap of (key, cell) pairs with the reference's exact sift rules (js/terrain-post.js:18-46):
// sift-up stops on >=, sift-down prefers the left child unless the right is strictly smaller.  The reference
// keys the heap through an external Float32Array; every cell is pushed exactly once and its key never changes
// afterwards, so carrying the key next to the cell id compares the same values while keeping the sifts inside
// one small contiguous array (the external-array form costs a cache miss per comparison at 10^7 cells).
using HeapItem = FloodHeapItem;
struct KeyHeap {
    std::vector<HeapItem>& d;
    size_t n = 0;
    explicit KeyHeap(std::vector<HeapItem>& storage) : d(storage) {}
    void push(int32_t c, float kc) {
        size_t i = n++;
        if (d.size() < n + 2) d.resize(d.size() * 2 + 1024);
        HeapItem* h = d.data();
        while (i > 0) {
            const size_t parent = (i - 1) >> 1;
            if (kc >= h[parent].key) break;
            h[i] = h[parent];
            i = parent;
        }
        h[i] = HeapItem{kc, c};
    }
    // Same comparisons as js/terrain-post.js:36-42 (left vs current, then right vs the smaller of the two),
    // evaluated with selects instead of branches; slots n and n+1 hold +inf sentinels so absent children lose.
    int32_t pop() {
        HeapItem* h = d.data();
        const int32_t top = h[0].cell;
        const HeapItem last = h[--n];
        h[n].key = INFINITY; h[n + 1].key = INFINITY;
        if (n > 0) {
            size_t i = 0;
            const float kc = last.key;
            for (;;) {
                const size_t l = 2 * i + 1;
                if (l >= n) break;
                const float kl = h[l].key, kr = h[l + 1].key;
                const bool a = kl < kc;
                const float mk = a ? kl : kc;
                size_t s = a ? l : i;
                s = (kr < mk) ? l + 1 : s;
                if (s == i) break;
                h[i] = h[s];
                i = s;
            }
            h[i] = last;
        }
        return top;
    }
};

inline uint32_t asc_bits(float f) {
    if (f == 0.0f) f = 0.0f;
    uint32_t u; std::memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

}  // namespace

void FloodScratch::ensure(int32_t N) {
    if ((int32_t)surface.size() >= N) return;
    surface.resize(N); drainTo.resize(N); visited.resize(N); path.resize(N); order.resize(N); order2.resize(N);
    bits.resize(N); bits2.resize(N);
    staticValid = false;
}

// Everything that depends only on (mesh, r_isOcean): the open-ocean mask (largest ocean component, first
// wins ties, js/terrain-post.js:66-94) and the seed list (land cells whose first open-ocean neighbour in
// adjacency order exists, ascending r, :118-128).  Both flood calls of an erodeComposite share it.
static void build_static(int32_t N, const int32_t* off, const int32_t* adj, const uint8_t* ocean, FloodScratch& S) {
    std::vector<int32_t> label(N, -1), stack(N);
    std::vector<int32_t> sizes;
    for (int32_t r = 0; r < N; ++r) {
        if (!ocean[r] || label[r] >= 0) continue;
        const int32_t lab = (int32_t)sizes.size();
        int32_t sp = 0, size = 0;
        stack[sp++] = r; label[r] = lab;
        while (sp > 0) {
            const int32_t cur = stack[--sp];
            ++size;
            for (int32_t i = off[cur]; i < off[cur + 1]; ++i) {
                const int32_t nb = adj[i];
                if (ocean[nb] && label[nb] < 0) { label[nb] = lab; stack[sp++] = nb; }
            }
        }
        sizes.push_back(size);
    }
    int32_t mainLab = 0;
    for (size_t i = 1; i < sizes.size(); ++i) if (sizes[i] > sizes[mainLab]) mainLab = (int32_t)i;
    S.seedCell.clear(); S.seedTarget.clear();
    for (int32_t r = 0; r < N; ++r) {
        if (ocean[r]) continue;
        for (int32_t i = off[r]; i < off[r + 1]; ++i) {
            const int32_t nb = adj[i];
            if (ocean[nb] && label[nb] == mainLab) { S.seedCell.push_back(r); S.seedTarget.push_back(nb); break; }
        }
    }
    S.staticValid = true;
}

void priority_flood_carve_host(int32_t N, const int32_t* off, const int32_t* adj, float* e,
                               const uint8_t* ocean, double carveStrength, FloodScratch& S) {
    const double EPS = 1e-7;
    const bool timing = std::getenv("WO_FLOOD_TIMING") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[flood] %-10s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
        tp = now;
    };
    S.ensure(N);
    if (!S.staticValid) build_static(N, off, adj, ocean, S);
    lap("static");

    // --- surface / drainTo / visited (:107-113); keys are formed when a cell is pushed.  The flood's
    // per-neighbour state (elevation, visited flag, drain target) is packed in one 8-byte record so that
    // visiting a neighbour costs one cache line, not three: drain == UNVISITED / OCEAN / NO_TARGET(-1) / cell id.
    float* surface = S.surface.data();
    int32_t* drainTo = S.drainTo.data();
    constexpr int32_t UNVISITED = -2, OCEAN = -3;
    if ((int32_t)S.state.size() < N) S.state.resize(N);
    FloodCell* st = S.state.data();
    if ((int32_t)S.root.size() < N) S.root.resize(N);
    int32_t* root = S.root.data();          // drainage tree id (= ordinal of the seed the cell finally drains through), -1 for none
    parallel_ranges(N, [&](int64_t b, int64_t en, int) {
        for (int64_t r = b; r < en; ++r) { st[r].e = e[r]; st[r].drain = ocean[r] ? OCEAN : UNVISITED; root[r] = -1; }
    });
    std::memcpy(surface, e, sizeof(float) * (size_t)N);
    if (S.heapStore.size() < 4096) S.heapStore.resize(4096);
    KeyHeap heap(S.heapStore);
    for (size_t i = 0; i < S.seedCell.size(); ++i) {        // :118-128, ascending r
        const int32_t r = S.seedCell[i];
        st[r].drain = S.seedTarget[i];
        root[r] = (int32_t)i;
        heap.push(r, (float)((double)e[r] + cell_noise(r)));
    }
    lap("init+seeds");
    // --- pass 1 (:131-147)
    const FloodHeapItem* hp = S.heapStore.data();
    while (heap.n > 0) {
        const int32_t c = heap.pop();
        hp = S.heapStore.data();
        // The cells that pop next sit in the first heap levels: start pulling their rows in now (the flood visits
        // cells in key order, i.e. scattered over the globe, so every pop would otherwise start with cold misses).
        {
            const size_t lim2 = heap.n < 7 ? heap.n : 7;
            for (size_t q = 0; q < lim2; ++q) {
                const int32_t cc = hp[q].cell;
                __builtin_prefetch(&off[cc]); __builtin_prefetch(&st[cc]); __builtin_prefetch(&surface[cc]); __builtin_prefetch(&root[cc]);
            }
            if (heap.n > 0) {          // the very next pop: its row was requested while it sat deeper; now request its neighbours' state
                const int32_t c0 = hp[0].cell;
                const int32_t o0 = off[c0], o1 = off[c0 + 1];
                for (int32_t j = o0; j < o1; ++j) __builtin_prefetch(&st[adj[j]]);
            }
            for (size_t q = 1; q < lim2 && q < 3; ++q) __builtin_prefetch(&adj[off[hp[q].cell]]);
        }
        const double lim = (double)surface[c] + EPS;
        const int32_t iEnd = off[c + 1];
        const int32_t rootC = root[c];
        for (int32_t i = off[c]; i < iEnd; ++i) {
            const int32_t nb = adj[i];
            FloodCell& sn = st[nb];
            if (sn.drain != UNVISITED) continue;
            sn.drain = c;
            root[nb] = rootC;
            float k;
            if ((double)sn.e < lim) {
                surface[nb] = (float)lim;
                k = (float)((double)surface[nb] + cell_noise(nb));
            } else {
                k = (float)((double)sn.e + cell_noise(nb));
            }
            heap.push(nb, k);
        }
    }
    // unreachable land (enclosed by inland seas) keeps drainTo = -1 (:108)
    parallel_ranges(N, [&](int64_t b, int64_t en, int) {
        for (int64_t r = b; r < en; ++r) { const int32_t d = st[r].drain; drainTo[r] = d >= 0 ? d : -1; }
    });
    lap("pass1");
    // --- pass 2 (:152-196) and pass 3 (:200-214).  Both are sequential in the reference, but every cell a turn
    // reads or writes (the drain path of r, the carve window on it, r itself; in pass 3 the cell and its
    // drain target) lies inside r's drainage tree, and trees share no land cell.  So the trees are processed
    // concurrently on the host's cores while each tree keeps the reference's order (ascending r in pass 2,
    // ascending (surface, r) in pass 3): identical results, no relaxation.
    const int32_t nTrees = (int32_t)S.seedCell.size();
    std::vector<int32_t> cnt2(nTrees + 1, 0);
    // all cells of each tree, ascending r inside.  (Not just the cells with an initial deficit: a carve lowers
    // other cells of the path below their flood surface, and the reference tests `deficit > EPS` against the
    // current height when it reaches them, :154-155.)
    int32_t* list2 = S.order.data();
    {
        for (int32_t r = 0; r < N; ++r) if (root[r] >= 0) cnt2[root[r] + 1]++;
        for (int32_t t = 0; t < nTrees; ++t) cnt2[t + 1] += cnt2[t];
        std::vector<int32_t> fill(cnt2.begin(), cnt2.end() - 1);
        for (int32_t r = 0; r < N; ++r) if (root[r] >= 0) list2[fill[root[r]]++] = r;
    }
    // Trees are numbered by their seed's cell id, i.e. along the Fibonacci spiral: consecutive tree ids are
    // spatial neighbours.  Workers take contiguous chunks of tree ids so that each core works inside its own
    // band of the elevation array (random hand-out makes cores fight over shared cache lines and erases the gain).
    auto for_trees = [&](const std::vector<int32_t>& cnt, auto body) {
        // measured on the 2-socket EPYC GPU box: 4-8 workers give ~3x on these pointer-chasing passes; 16+ cores
        // contend in the memory system (cross-CCD/NUMA coherence on the written field) and per-tree time rises 7x
        const int nt = std::max(1, std::min<int>(std::min(host_threads(), 6), nTrees));
        const int64_t total = cnt[nTrees];
        const int64_t perChunk = std::max<int64_t>(2048, total / (nt * 16));
        std::vector<int32_t> chunkStart;            // chunk boundaries with ~equal cell counts
        chunkStart.push_back(0);
        for (int32_t t = 0, last = 0; t < nTrees; ++t)
            if (cnt[t + 1] - cnt[last] >= perChunk) { chunkStart.push_back(t + 1); last = t + 1; }
        if (chunkStart.back() != nTrees) chunkStart.push_back(nTrees);
        const size_t nChunks = chunkStart.size() - 1;
        std::atomic<size_t> next{0};
        std::atomic<int64_t> busyUs{0}, maxTreeUs{0};
        auto worker = [&]() {
            int64_t myBusy = 0, myMax = 0;
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= nChunks) break;
                for (int32_t t = chunkStart[c]; t < chunkStart[c + 1]; ++t) if (cnt[t + 1] > cnt[t]) {
                    if (timing) {
                        auto t0 = std::chrono::steady_clock::now();
                        body(t);
                        const int64_t us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                        myBusy += us; if (us > myMax) myMax = us;
                    } else body(t);
                }
            }
            if (timing) { busyUs += myBusy; int64_t prev = maxTreeUs.load(); while (myMax > prev && !maxTreeUs.compare_exchange_weak(prev, myMax)) {} }
        };
        if (nt == 1) { worker(); return; }
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i) th.emplace_back(worker);
        for (auto& t : th) t.join();
        if (timing) std::fprintf(stderr, "[flood]   %d threads, %zu chunks: summed busy %.1f ms, slowest single tree %.1f ms\n", nt, nChunks, busyUs.load() / 1e3, maxTreeUs.load() / 1e3);
    };
    std::atomic<int64_t> nDef{0}, totLen{0}, maxTreeLen{0};
    lap("group2");
    for_trees(cnt2, [&](int32_t tree) {
        std::vector<int32_t> path;
        int64_t myLen = 0;
        for (int32_t q = cnt2[tree]; q < cnt2[tree + 1]; ++q) {
            const int32_t r = list2[q];
            // the deficit is taken against the CURRENT (already carved) height, as in the reference (:154)
            const double deficit = (double)surface[r] - (double)e[r];
            if (deficit <= EPS) continue;
            if (timing) ++nDef;
            path.clear();
            int32_t peakIdx = -1;
            double peakElev = -INFINITY;
            for (int32_t cur = r; cur >= 0 && !ocean[cur]; cur = drainTo[cur]) {
                path.push_back(cur);
                if ((double)e[cur] > peakElev) { peakElev = e[cur]; peakIdx = (int32_t)path.size() - 1; }
            }
            const int32_t len = (int32_t)path.size();
            myLen += len;
            if (peakIdx < 0) continue;
            const double carveAmount = deficit * carveStrength;
            const double rc = std::ceil((double)len * 0.3);
            const int32_t radius = rc > 3.0 ? (int32_t)rc : 3;
            const int32_t k0 = peakIdx - radius > 0 ? peakIdx - radius : 0;
            const int32_t k1 = peakIdx + radius < len - 1 ? peakIdx + radius : len - 1;
            double kernelSum = 0;
            for (int32_t k = k0; k <= k1; ++k) kernelSum += 1 - std::fabs((double)(k - peakIdx)) / (radius + 1);
            if (kernelSum > 0) {
                for (int32_t k = k0; k <= k1; ++k) {
                    const double w = (1 - std::fabs((double)(k - peakIdx)) / (radius + 1)) / kernelSum;
                    float v = (float)((double)e[path[k]] - carveAmount * w);
                    if (v < 0) v = 0;
                    e[path[k]] = v;
                }
            }
            e[r] = (float)((double)e[r] + deficit * (1 - carveStrength));
        }
        if (timing) { totLen += myLen; int64_t prev = maxTreeLen.load(); while (myLen > prev && !maxTreeLen.compare_exchange_weak(prev, myLen)) {} }
    });
    if (timing) std::fprintf(stderr, "[flood] pass2: %lld deficit cells in %d trees, total path length %lld\n", (long long)nDef.load(), nTrees, (long long)totLen.load());
    if (timing) std::fprintf(stderr, "[flood] pass2: largest tree walks %lld path steps\n", (long long)maxTreeLen.load());
    lap("pass2");
    // --- pass 3: land cells by ascending surface (stable => ties ascending r), then grouped by tree
    int32_t nLand = 0;
    int32_t* order = S.order.data();
    int32_t* order2 = S.order2.data();
    uint32_t* b0 = S.bits.data();
    uint32_t* b1 = S.bits2.data();
    for (int32_t r = 0; r < N; ++r) if (root[r] >= 0) { order[nLand] = r; b0[nLand] = asc_bits(surface[r]); ++nLand; }
    for (int pass = 0; pass < 3; ++pass) {          // 11 + 11 + 10 bit LSD radix, stable
        const int sh = pass * 11;
        const uint32_t mask = pass == 2 ? 1023u : 2047u;
        uint32_t cnt[2049];
        std::memset(cnt, 0, sizeof(cnt));
        for (int32_t i = 0; i < nLand; ++i) cnt[((b0[i] >> sh) & mask) + 1]++;
        for (int i = 0; i < 2048; ++i) cnt[i + 1] += cnt[i];
        for (int32_t i = 0; i < nLand; ++i) {
            const uint32_t d = cnt[(b0[i] >> sh) & mask]++;
            b1[d] = b0[i]; order2[d] = order[i];
        }
        std::swap(b0, b1); std::swap(order, order2);
    }
    std::vector<int32_t> cnt3(nTrees + 1, 0);
    for (int32_t i = 0; i < nLand; ++i) cnt3[root[order[i]] + 1]++;
    for (int32_t t = 0; t < nTrees; ++t) cnt3[t + 1] += cnt3[t];
    {
        std::vector<int32_t> fill(cnt3.begin(), cnt3.end() - 1);
        for (int32_t i = 0; i < nLand; ++i) { const int32_t c = order[i]; order2[fill[root[c]]++] = c; }   // stable
    }
    for_trees(cnt3, [&](int32_t tree) {
        for (int32_t q = cnt3[tree]; q < cnt3[tree + 1]; ++q) {
            const int32_t c = order2[q], t = drainTo[c];
            if (t < 0) continue;
            const double te = ocean[t] ? 0.0 : (double)e[t];
            if ((double)e[c] <= te) e[c] = (float)(te + EPS);
        }
    });
    lap("pass3");
}

}  // namespace wo
