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
    for (int32_t r = 0; r < N; ++r) { st[r].e = e[r]; st[r].drain = ocean[r] ? OCEAN : UNVISITED; }
    std::memcpy(surface, e, sizeof(float) * (size_t)N);
    if (S.heapStore.size() < 4096) S.heapStore.resize(4096);
    KeyHeap heap(S.heapStore);
    for (size_t i = 0; i < S.seedCell.size(); ++i) {        // :118-128, ascending r
        const int32_t r = S.seedCell[i];
        st[r].drain = S.seedTarget[i];
        heap.push(r, (float)((double)e[r] + cell_noise(r)));
    }
    lap("init+seeds");
    // --- pass 1 (:131-147)
    while (heap.n > 0) {
        const int32_t c = heap.pop();
        const double lim = (double)surface[c] + EPS;
        const int32_t iEnd = off[c + 1];
        for (int32_t i = off[c]; i < iEnd; ++i) {
            const int32_t nb = adj[i];
            FloodCell& sn = st[nb];
            if (sn.drain != UNVISITED) continue;
            sn.drain = c;
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
    for (int32_t r = 0; r < N; ++r) { const int32_t d = st[r].drain; drainTo[r] = d >= 0 ? d : -1; }
    lap("pass1");
    // --- pass 2 (:152-196): ascending r, sequential
    int32_t* path = S.path.data();
    int64_t nDef = 0, totLen = 0;
    for (int32_t r = 0; r < N; ++r) {
        if (ocean[r]) continue;
        const double deficit = (double)surface[r] - (double)e[r];
        if (deficit <= EPS) continue;
        int32_t len = 0, peakIdx = -1;
        double peakElev = -INFINITY;
        for (int32_t cur = r; cur >= 0 && !ocean[cur]; cur = drainTo[cur]) {
            path[len++] = cur;
            if ((double)e[cur] > peakElev) { peakElev = e[cur]; peakIdx = len - 1; }
        }
        if (timing) { ++nDef; totLen += len; }
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
    if (timing) std::fprintf(stderr, "[flood] pass2: %lld deficit cells, total path length %lld\n", (long long)nDef, (long long)totLen);
    lap("pass2");
    // --- pass 3 (:200-214): land cells by ascending surface (stable), enforce descent along drainTo
    int32_t nLand = 0;
    int32_t* order = S.order.data();
    int32_t* order2 = S.order2.data();
    uint32_t* b0 = S.bits.data();
    uint32_t* b1 = S.bits2.data();
    for (int32_t r = 0; r < N; ++r) if (!ocean[r]) { order[nLand] = r; b0[nLand] = asc_bits(surface[r]); ++nLand; }
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
    for (int32_t i = 0; i < nLand; ++i) {
        const int32_t c = order[i], t = drainTo[c];
        if (t < 0) continue;
        const double te = ocean[t] ? 0.0 : (double)e[t];
        if ((double)e[c] <= te) e[c] = (float)(te + EPS);
    }
    lap("pass3");
}

}  // namespace wo
