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

// Binary min-heap over cell ids keyed by an external float array, with the reference's exact sift rules
// (js/terrain-post.js:18-46): sift-up stops on >=, sift-down prefers the left child unless the right is
// strictly smaller.
struct KeyHeap {
    std::vector<int32_t>& d;
    const float* key;
    size_t n = 0;
    KeyHeap(std::vector<int32_t>& storage, const float* k) : d(storage), key(k) {}
    void push(int32_t c) {
        size_t i = n++;
        d[i] = c;
        const float kc = key[c];
        while (i > 0) {
            const size_t parent = (i - 1) >> 1;
            if (kc >= key[d[parent]]) break;
            d[i] = d[parent]; d[parent] = c;
            i = parent;
        }
    }
    int32_t pop() {
        const int32_t top = d[0];
        const int32_t last = d[--n];
        if (n > 0) {
            size_t i = 0;
            d[0] = last;
            for (;;) {
                size_t s = i;
                const size_t l = 2 * i + 1, r = l + 1;
                if (l < n && key[d[l]] < key[d[s]]) s = l;
                if (r < n && key[d[r]] < key[d[s]]) s = r;
                if (s == i) break;
                const int32_t t = d[i]; d[i] = d[s]; d[s] = t;
                i = s;
            }
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
    if ((int32_t)label.size() >= N) return;
    label.resize(N); stack.resize(N); open.resize(N); surface.resize(N); drainTo.resize(N);
    visited.resize(N); key.resize(N); path.resize(N); order.resize(N); order2.resize(N);
    bits.resize(N); bits2.resize(N);
}

void priority_flood_carve_host(int32_t N, const int32_t* off, const int32_t* adj, float* e,
                               const uint8_t* ocean, double carveStrength, FloodScratch& S) {
    const double EPS = 1e-7;
    S.ensure(N);
    int32_t* label = S.label.data();
    // --- largest ocean component = open ocean (:66-94)
    std::fill(label, label + N, -1);
    std::vector<int32_t> sizes;
    for (int32_t r = 0; r < N; ++r) {
        if (!ocean[r] || label[r] >= 0) continue;
        const int32_t lab = (int32_t)sizes.size();
        int32_t sp = 0, size = 0;
        S.stack[sp++] = r; label[r] = lab;
        while (sp > 0) {
            const int32_t cur = S.stack[--sp];
            ++size;
            for (int32_t i = off[cur]; i < off[cur + 1]; ++i) {
                const int32_t nb = adj[i];
                if (ocean[nb] && label[nb] < 0) { label[nb] = lab; S.stack[sp++] = nb; }
            }
        }
        sizes.push_back(size);
    }
    int32_t mainLab = 0;
    for (size_t i = 1; i < sizes.size(); ++i) if (sizes[i] > sizes[mainLab]) mainLab = (int32_t)i;
    for (int32_t r = 0; r < N; ++r) S.open[r] = (ocean[r] && label[r] == mainLab) ? 1 : 0;

    // --- keys, seeds (:107-128)
    float* surface = S.surface.data();
    int32_t* drainTo = S.drainTo.data();
    uint8_t* visited = S.visited.data();
    float* key = S.key.data();
    for (int32_t r = 0; r < N; ++r) {
        surface[r] = e[r]; drainTo[r] = -1; visited[r] = 0;
        key[r] = (float)((double)e[r] + cell_noise(r));
    }
    KeyHeap heap(S.stack, key);
    for (int32_t r = 0; r < N; ++r) {
        if (ocean[r]) { visited[r] = 1; continue; }
        for (int32_t i = off[r]; i < off[r + 1]; ++i) {
            if (S.open[adj[i]]) { visited[r] = 1; drainTo[r] = adj[i]; heap.push(r); break; }
        }
    }
    // --- pass 1 (:131-147)
    while (heap.n > 0) {
        const int32_t c = heap.pop();
        const double lim = (double)surface[c] + EPS;
        for (int32_t i = off[c]; i < off[c + 1]; ++i) {
            const int32_t nb = adj[i];
            if (visited[nb]) continue;
            visited[nb] = 1;
            drainTo[nb] = c;
            if ((double)e[nb] < lim) {
                surface[nb] = (float)lim;
                key[nb] = (float)((double)surface[nb] + cell_noise(nb));
            }
            heap.push(nb);
        }
    }
    // --- pass 2 (:152-196): ascending r, sequential
    int32_t* path = S.path.data();
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
}

}  // namespace wo
