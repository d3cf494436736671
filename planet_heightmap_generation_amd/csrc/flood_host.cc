// priorityFloodCarve — host-resident stage of the erosion stack (reference: js/terrain-post.js:59-215).
//
// Why this one stage runs on the host in this round: the noise-perturbed best-first flood pops cells in an
// order that is defined by a binary heap over float keys (ties included), pass 2 carves along drain paths
// sequentially in ascending cell order reading already-carved heights, and SURVEY §6.4 measured that
// relaxing either changes the result by ~1e-2 RMS.  A serial walk on one GPU lane would take seconds per
// call, so the two calls per erodeComposite run here between device phases (the field makes one D2H + H2D
// round trip per call).  DESIGN.md lists an order-equivalent device flood as the next step.
// This is product code (it is the designed path, not a fallback, and it does not touch oracle/).
//
// Layout (all results identical to the reference; only *where* data sits changes):
//  * land cells are renumbered compactly in Morton order of their positions, with their own CSR (ocean
//    neighbours dropped: the reference skips them as "visited" without side effects).  The flood pops cells in
//    key order, i.e. scattered over the globe; with 10^7 cells every pop used to touch ~6 cold cache lines in
//    40-240 MB arrays.  In the compact Morton layout a cell, its row and its neighbours' state share a few lines
//    in arrays 4x smaller.  Orders the reference defines by cell id (seed order, pass-2 order, pass-3 tie
//    order, the noise hash) still use the original ids.
//  * pass 1 is serial (heap order); the rows of the next few heap entries are prefetched.
//  * passes 2 and 3 are separable per drainage tree (every path, carve window and drain target lies inside
//    one tree), so trees run on a few host cores while each tree keeps the reference's order.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <stdexcept>
#include <vector>

#include "host_util.h"
#include "wo_internal.h"

namespace wo {

namespace {

inline double cell_noise(int32_t r) { return flood_cell_noise_of(r); }      // (wo_internal.h: shared with the device)

// Binary min-heap of (key, cell) pairs with the reference's exact sift rules (js/terrain-post.js:18-46):
// sift-up stops on >=, sift-down compares left with the moving item, then right with the smaller of the two.
// The reference keys the heap through an external Float32Array; every cell is pushed exactly once and its key
// never changes afterwards, so carrying the key next to the cell compares the same values.
using HeapItem = FloodHeapItem;
struct KeyHeap {
    hvec<HeapItem>& d;
    size_t n = 0;
    explicit KeyHeap(hvec<HeapItem>& storage) : d(storage) {}
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
    // The reference's sift-down (:33-45) takes the left child unless the right one is strictly smaller and stops when
    // that child is not smaller than the moving item; choosing the child first keeps the item's key off the
    // dependency chain between levels (load pair -> compare -> index), which is what bounds a pop.
    int32_t pop() {      // slots n, n+1 hold +inf so an absent right child loses
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
                // the item sinks ~log2(n) levels and below the cached top every level is a miss on the dependency chain: the 16
                // possible positions four levels down are one contiguous 128-byte stretch, asked for now
                { const size_t f = 16 * i + 15; if (f < n) { __builtin_prefetch(&h[f]); __builtin_prefetch(&h[f + 8]); __builtin_prefetch(&h[f + 15]); } }
                const size_t s = l + (h[l + 1].key < h[l].key ? 1 : 0);
                if (!(h[s].key < kc)) break;
                h[i] = h[s];
                i = s;
            }
            h[i] = last;
        }
        return top;
    }
};

// The landmass walks never rely on how their queue orders EQUAL keys (that is what the tie groups are for), so they need the
// reference's binary heap only where the single heap itself is replayed.  They use a 4-ary heap: half the levels, the four
// children of a node are one 32-byte stretch.  Same interface as KeyHeap.
struct KeyHeap4 {
    hvec<HeapItem>& d;
    size_t n = 0;
    explicit KeyHeap4(hvec<HeapItem>& storage) : d(storage) {}
    const HeapItem* front() const { return d.data(); }        // [0] is the next pop; the first few entries are the likely next ones (prefetch hints)
    size_t front_count() const { return n; }
    void prime() {}
    void push(int32_t c, float kc) {
        size_t i = n++;
        if (d.size() < n + 8) d.resize(d.size() * 2 + 1024);
        HeapItem* h = d.data();
        while (i > 0) {
            const size_t parent = (i - 1) >> 2;
            if (kc >= h[parent].key) break;
            h[i] = h[parent];
            i = parent;
        }
        h[i] = HeapItem{kc, c};
    }
    int32_t pop() {      // slots n .. n+3 hold +inf so absent children lose
        HeapItem* h = d.data();
        const int32_t top = h[0].cell;
        const HeapItem last = h[--n];
        h[n].key = INFINITY; h[n + 1].key = INFINITY; h[n + 2].key = INFINITY; h[n + 3].key = INFINITY;
        if (n > 0) {
            size_t i = 0;
            const float kc = last.key;
            for (;;) {
                const size_t c0 = 4 * i + 1;
                if (c0 >= n) break;
                { const size_t f = 16 * i + 5; if (f < n) { __builtin_prefetch(&h[f]); __builtin_prefetch(&h[f + 8]); } }     // the 16 grandchildren
                size_t s = c0; float ks = h[c0].key;
                if (h[c0 + 1].key < ks) { s = c0 + 1; ks = h[c0 + 1].key; }
                if (h[c0 + 2].key < ks) { s = c0 + 2; ks = h[c0 + 2].key; }
                if (h[c0 + 3].key < ks) { s = c0 + 3; ks = h[c0 + 3].key; }
                if (!(ks < kc)) break;
                h[i] = h[s];
                i = s;
            }
            h[i] = last;
        }
        return top;
    }
};

// A queue for walks whose heap grows large.  The second flood of a step meets eroded terrain: 86 % of the cells of the bench planet's
// largest landmass are pushed with a RAISED key (level + EPS + noise, noise < 0.01) and wait for the level to pass their noise, so the
// heap holds 100 000 entries on average (170 000 at most; 11 600 in the first flood) — nine levels of a 4-ary heap over 800 KB.  All
// these keys sit within ~0.01 of the level.  So: buckets of width 2^-15 on a ring of 1024 (a window of 0.031 above the current bucket),
// unsorted; only the CURRENT bucket is a heap (a few hundred entries), which also takes every key below it; keys beyond the window go
// to an ordinary heap and come back when the ring reaches them.  A push is an append, a pop works on a small heap.  Exact: the
// bucket index is a monotone function of the key, the current heap orders by the key itself, and equal keys always share a bucket
// (the tie test of the walk looks at the next key of the current heap: once that heap is empty every remaining key is strictly larger).
struct RingQueue {
    static constexpr int NB = 1024;
    static constexpr int64_t FAR_BELOW = -(int64_t(1) << 40);
    KeyHeap4 cur, over;
    size_t n = 0;
    int64_t b0 = FAR_BELOW;                              // until prime(): every push goes to `over`
    struct Ring { std::vector<HeapItem> b[NB]; uint64_t occ[NB / 64]; hvec<HeapItem> overStore; };
    Ring& R;
    static Ring& ring() { static thread_local Ring* r = new Ring(); return *r; }
    explicit RingQueue(hvec<HeapItem>& storage) : cur(storage), over(ring().overStore), R(ring()) {
        if (R.overStore.size() < 1024) R.overStore.resize(1024);
        std::memset(R.occ, 0, sizeof(R.occ));
        for (auto& v : R.b) v.clear();
    }
    static int64_t bucket_of(float k) { return (int64_t)(k * 32768.0f); }      // exact scaling; truncation is monotone
    const HeapItem* front() const { return cur.d.data(); }
    size_t front_count() const { return cur.n; }
    void push(int32_t c, float k) {
        ++n;
        const int64_t bi = bucket_of(k);
        if (bi <= b0) cur.push(c, k);
        else if (bi - b0 < NB) { const int s = (int)(bi & (NB - 1)); R.b[s].push_back(HeapItem{k, c}); R.occ[s >> 6] |= 1ull << (s & 63); }
        else over.push(c, k);
        if (primed && cur.n == 0) advance();              // the queue was empty (its last entry just popped): the invariant "n > 0 => the current heap holds the front" again
    }
    void advance() {                                       // cur is empty, n > 0: the next occupied bucket becomes the current one
        int64_t br = INT64_MAX;
        {
            const int start = (int)((b0 + 1) & (NB - 1));
            int w = start >> 6; uint64_t m = R.occ[w] & (~0ull << (start & 63));
            for (int step = 0; step <= NB / 64; ++step) {
                if (m) { const int s = (w << 6) + __builtin_ctzll(m); br = b0 + 1 + ((s - start) & (NB - 1)); break; }
                w = (w + 1) & (NB / 64 - 1);
                m = R.occ[w];
                if (step == NB / 64 - 1) m &= ~(~0ull << (start & 63));       // back at the first word: the bits before `start`
            }
        }
        const int64_t bo = over.n ? bucket_of(over.d.data()[0].key) : INT64_MAX;
        const int64_t b = br < bo ? br : bo;
        b0 = b;
        if (br == b) {
            const int s = (int)(b & (NB - 1));
            for (const HeapItem& it : R.b[s]) cur.push(it.cell, it.key);
            R.b[s].clear(); R.occ[s >> 6] &= ~(1ull << (s & 63));
        }
        while (over.n && bucket_of(over.d.data()[0].key) <= b) { const float k = over.d.data()[0].key; const int32_t c = over.pop(); cur.push(c, k); }
    }
    bool primed = false;
    void prime() { primed = true; if (cur.n == 0 && n > 0) advance(); }
    int32_t pop() {
        const int32_t c = cur.pop();
        --n;
        if (cur.n == 0 && n > 0) advance();
        return c;
    }
};

// Test support (tests/emu): the ring against the 4-ary heap on a random operation sequence; returns the number of pops whose key differs.
inline int64_t queues_differ(int64_t ops, uint64_t seed) {
    hvec<HeapItem> sa(1024), sb(1024);
    KeyHeap4 a(sa); RingQueue b(sb);
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 7;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    auto unit = [&]() { return (double)(rnd() >> 11) / 9007199254740992.0; };
    int64_t bad = 0; int32_t id = 0; float level = 0.01f;
    for (int q = 0; q < 64; ++q) { const float k = (float)(unit() * 0.3); a.push(id, k); b.push(id, k); ++id; }      // "seeds": before prime()
    b.prime();
    bool draining = false;
    for (int64_t op = 0; op < ops; ++op) {
        if (a.n == 0) draining = false;
        if (!draining && rnd() % 4096 == 0) draining = true;                      // now and then: pop until empty
        const bool doPop = a.n > 0 && (draining || rnd() % 100 < 50);
        if (doPop) {
            const float ka = a.front()[0].key, kb = b.front()[0].key;
            a.pop(); b.pop();
            if (std::memcmp(&ka, &kb, 4) != 0) ++bad;
            if (ka > level) level = ka;
        } else {
            const uint64_t kind = rnd() % 100;
            float k = kind < 80 ? (float)(level + unit() * 0.012) : kind < 95 ? (float)(level - unit() * 0.01) : (float)(level + unit() * 0.5);
            if (k < 0) k = 0;
            if (rnd() % 64 == 0) k = level;                                       // equal keys
            a.push(id, k); b.push(id, k); ++id;
        }
        if (a.n != b.n) return -1;
    }
    while (a.n > 0) { const float ka = a.front()[0].key, kb = b.front()[0].key; a.pop(); b.pop(); if (std::memcmp(&ka, &kb, 4) != 0) ++bad; }
    return bad + (b.n != 0);
}

inline uint32_t asc_bits(float f) {
    if (f == 0.0f) f = 0.0f;
    uint32_t u; std::memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

inline uint32_t spread3(uint32_t v) {           // 10 bits -> every third bit
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000FFu;
    v = (v | (v << 8)) & 0x0300F00Fu;
    v = (v | (v << 4)) & 0x030C30C3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

template <class KV, class VV>
void radix_sort_u32(KV& keys, VV& vals) {     // stable LSD, 11/11/10 bits
    const size_t n = keys.size();
    KV k2(n); VV v2(n);
    for (int pass = 0; pass < 3; ++pass) {
        const int sh = pass * 11; const uint32_t mask = pass == 2 ? 1023u : 2047u;
        uint32_t cnt[2049]; std::memset(cnt, 0, sizeof(cnt));
        for (size_t i = 0; i < n; ++i) cnt[((keys[i] >> sh) & mask) + 1]++;
        for (int i = 0; i < 2048; ++i) cnt[i + 1] += cnt[i];
        for (size_t i = 0; i < n; ++i) { const uint32_t d = cnt[(keys[i] >> sh) & mask]++; k2[d] = keys[i]; v2[d] = vals[i]; }
        keys.swap(k2); vals.swap(v2);
    }
}

constexpr int32_t UNVISITED = -2, TO_OCEAN = -3, NO_TARGET = -1;

}  // namespace
int64_t flood_queues_differ(int64_t ops, uint64_t seed) { return queues_differ(ops, seed); }

// Everything that depends only on (mesh, positions, r_isOcean): Morton order, compact land numbering and CSR, the
// open-ocean component (largest, first wins ties, js/terrain-post.js:66-94) and the seed list (land cells whose
// first open-ocean neighbour in adjacency order exists, ascending r, :118-128).  Shared by both flood calls of
// an erodeComposite and kept across calls while the ocean mask is unchanged.
void flood_build_static(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, const uint8_t* ocean, FloodScratch& S, const int32_t* mortonAll) {
    const bool timing = std::getenv("WO_FLOOD_TIMING") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[flood static] %-12s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
        tp = now;
    };
    // Ocean components (:66-94).  The reference labels them by a stack walk in ascending r and keeps the largest, the
    // first label winning ties — i.e. among the largest components the one holding the smallest cell id.  Only that
    // choice is observable, so the components come from a concurrent union-find whose roots are the smallest id of
    // each component (link the larger root under the smaller with a CAS; path halving on the way up).
    hvec<int32_t> parentStore((size_t)N);               // (not zero-filled: mesh_components writes every entry)
    int32_t* parent = parentStore.data();
    mesh_components(N, off, adj, [&](int32_t r) { return ocean[r] != 0; }, [](int32_t, int32_t) { return true; }, parent);
    // component sizes per thread as (root, count) runs, merged afterwards
    std::vector<std::vector<std::pair<int32_t, int64_t>>> runs(host_threads() + 1);
    parallel_ranges(N, [&](int64_t b, int64_t e, int t) {
        auto& out = runs[t];
        int32_t last = -1; int64_t cnt = 0;
        for (int64_t r = b; r < e; ++r) {
            if (!ocean[r]) continue;
            const int32_t root = parent[r];
            if (root == last) { ++cnt; continue; }
            if (cnt) out.push_back({last, cnt});
            last = root; cnt = 1;
        }
        if (cnt) out.push_back({last, cnt});
    });
    int32_t mainLab = -1;
    {
        std::vector<std::pair<int32_t, int64_t>> all;
        for (auto& v : runs) all.insert(all.end(), v.begin(), v.end());
        std::sort(all.begin(), all.end());
        int64_t best = 0;
        for (size_t i = 0; i < all.size();) {
            size_t j = i; int64_t sz = 0;
            while (j < all.size() && all[j].first == all[i].first) sz += all[j++].second;
            if (sz > best) { best = sz; mainLab = all[i].first; }      // ascending roots: the first of the largest wins
            i = j;
        }
    }
    const int32_t* label = parent;
    lap("ocean labels");
    // land cells in Morton order of their positions (identity order when no positions are given)
    hvec<int32_t> landCells;
    if (xyz && mortonAll) {
        // the caller has ALL cells in that order already (morton_order_cells: same keys, same stable sort, so the land cells appear in it in the order the sort
        // below would give them): a new mask on a known mesh is a filter, not a sort (32 -> 3 ms of a new terrain's set-up at 10 M cells)
        std::vector<int64_t> cnt(host_threads() + 2, 0);
        parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t c = 0; for (int64_t i = b; i < e; ++i) c += ocean[mortonAll[i]] ? 0 : 1; cnt[t + 1] = c; });
        for (size_t t = 1; t < cnt.size(); ++t) cnt[t] += cnt[t - 1];
        landCells.resize(cnt.back());
        parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t o = cnt[t]; for (int64_t i = b; i < e; ++i) { const int32_t r = mortonAll[i]; if (!ocean[r]) landCells[o++] = r; } });
    } else {
        std::vector<int64_t> cnt(host_threads() + 2, 0);
        parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t c = 0; for (int64_t r = b; r < e; ++r) c += ocean[r] ? 0 : 1; cnt[t + 1] = c; });
        for (size_t t = 1; t < cnt.size(); ++t) cnt[t] += cnt[t - 1];
        landCells.resize(cnt.back());
        parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t o = cnt[t]; for (int64_t r = b; r < e; ++r) if (!ocean[r]) landCells[o++] = (int32_t)r; });
    }
    const int32_t L = (int32_t)landCells.size();
    if (xyz && L > 1 && !mortonAll) {
        std::vector<uint32_t> keys(L);
        parallel_ranges(L, [&](int64_t b, int64_t e, int) {
            for (int64_t i = b; i < e; ++i) {
                const int32_t r = landCells[i];
                auto q = [](float v) { int32_t t = (int32_t)((v + 1.0f) * 511.5f); return (uint32_t)(t < 0 ? 0 : (t > 1023 ? 1023 : t)); };
                keys[i] = spread3(q(xyz[3 * r])) | (spread3(q(xyz[3 * r + 1])) << 1) | (spread3(q(xyz[3 * r + 2])) << 2);
            }
        });
        radix_sort_u32(keys, landCells);
    }
    lap("morton sort");
    S.L = L;
    S.landCell.swap(landCells);
    S.landIndex.resize(N);
    parallel_ranges(N, [&](int64_t b, int64_t e, int) { for (int64_t r = b; r < e; ++r) S.landIndex[r] = -1; });
    parallel_ranges(L, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; ++i) S.landIndex[S.landCell[i]] = (int32_t)i; });
    {   // land indices in ascending original id (the order passes 2 and 3 are defined in)
        std::vector<int64_t> cnt(host_threads() + 2, 0);
        parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t c = 0; for (int64_t r = b; r < e; ++r) c += ocean[r] ? 0 : 1; cnt[t + 1] = c; });
        for (size_t t = 1; t < cnt.size(); ++t) cnt[t] += cnt[t - 1];
        S.landByR.resize(L);
        parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t o = cnt[t]; for (int64_t r = b; r < e; ++r) if (!ocean[r]) S.landByR[o++] = S.landIndex[r]; });
    }
    S.offL.assign(L + 1, 0);
    parallel_ranges(L, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            const int32_t r = S.landCell[i];
            int32_t c = 0;
            for (int32_t j = off[r]; j < off[r + 1]; ++j) if (!ocean[adj[j]]) ++c;
            S.offL[i + 1] = c;
        }
    });
    inclusive_scan_parallel(S.offL.data() + 1, L);
    S.adjL.resize(S.offL[L]);
    parallel_ranges(L, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            const int32_t r = S.landCell[i];
            int32_t o = S.offL[i];
            for (int32_t j = off[r]; j < off[r + 1]; ++j) { const int32_t nb = adj[j]; if (!ocean[nb]) S.adjL[o++] = S.landIndex[nb]; }
        }
    });
    lap("compact csr");
    S.seedCell.clear();                     // land index of each seed, in ascending original id
    {
        std::vector<std::vector<int32_t>> part(host_threads() + 1);
        parallel_ranges(N, [&](int64_t b, int64_t e, int t) {
            for (int64_t r = b; r < e; ++r) {
                if (ocean[r]) continue;
                for (int32_t j = off[r]; j < off[r + 1]; ++j) {
                    const int32_t nb = adj[j];
                    if (ocean[nb] && label[nb] == mainLab) { part[t].push_back(S.landIndex[r]); break; }
                }
            }
        });
        for (auto& v : part) S.seedCell.insert(S.seedCell.end(), v.begin(), v.end());      // ranges ascend with the thread index
    }
    S.surface.resize(L); S.state.resize(L); S.root.resize(L); S.eL.resize(L); S.localIdx.resize(L);
    S.order.resize(L); S.order2.resize(L); S.bits.resize(L); S.bits2.resize(L); S.list2.resize(L);
    lap("seeds+alloc");
    {   // Landmasses: connected components of the compact land graph.  The flood never crosses water (ocean cells are
        // "visited" from the start, :119), so every landmass floods from its own seeds and pass 1 can give each its own
        // heap (flood_pass1_landmasses).  Seeds are grouped by landmass (ascending original id inside, the order the
        // reference pushes them in); landmasses are taken largest first.
        hvec<int32_t> comp((size_t)L);
        mesh_components(L, S.offL.data(), S.adjL.data(), [](int32_t) { return true; }, [](int32_t, int32_t) { return true; }, comp.data());
        // cells per component root: runs of equal roots (Morton order keeps a landmass's cells together) added with one atomic each
        hvec<int32_t> size((size_t)L);
        parallel_ranges(L, [&](int64_t b, int64_t e, int) { std::memset(size.data() + b, 0, sizeof(int32_t) * (size_t)(e - b)); });
        parallel_ranges(L, [&](int64_t b, int64_t e, int) {
            int32_t last = -1, run = 0;
            for (int64_t i = b; i < e; ++i) {
                const int32_t c = comp[i];
                if (c == last) { ++run; continue; }
                if (run) __atomic_fetch_add(&size[last], run, __ATOMIC_RELAXED);
                last = c; run = 1;
            }
            if (run) __atomic_fetch_add(&size[last], run, __ATOMIC_RELAXED);
        });
        const int32_t nS = (int32_t)S.seedCell.size();
        std::vector<int32_t> ord(nS);
        for (int32_t k = 0; k < nS; ++k) ord[k] = k;
        std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
            const int32_t ca = comp[S.seedCell[a]], cb = comp[S.seedCell[b]];
            return size[ca] != size[cb] ? size[ca] > size[cb] : ca < cb;
        });
        S.compSeeds.assign(ord.begin(), ord.end());
        S.compSeedStart.clear(); S.compSize.clear();
        for (int32_t k = 0; k < nS; ++k)
            if (k == 0 || comp[S.seedCell[ord[k]]] != comp[S.seedCell[ord[k - 1]]]) { S.compSeedStart.push_back(k); S.compSize.push_back(size[comp[S.seedCell[ord[k]]]]); }
        S.compSeedStart.push_back(nS);
        S.stamp.resize(L);
        // per seeded landmass: its cells in ascending original id (the order pass 2 is defined in), and each seed's position
        // among its landmass's seeds (= local tree number)
        const int32_t nComp = (int32_t)S.compSize.size();
        hvec<int32_t>& compIndex = size;                     // by component root: the landmass's number, -1 for an unseeded one (the sizes are in compSize now)
        parallel_ranges(L, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; ++i) compIndex[i] = -1; });
        S.seedLocal.assign(nS, 0);
        for (int32_t k = 0; k < nComp; ++k) {
            compIndex[comp[S.seedCell[ord[S.compSeedStart[k]]]]] = k;
            for (int32_t q = S.compSeedStart[k]; q < S.compSeedStart[k + 1]; ++q) S.seedLocal[ord[q]] = q - S.compSeedStart[k];
        }
        S.compCellStart.assign(nComp + 1, 0);
        for (int32_t k = 0; k < nComp; ++k) S.compCellStart[k + 1] = S.compCellStart[k] + S.compSize[k];
        S.compCells.resize(S.compCellStart[nComp]);
        // the cells of every landmass in ascending original id: per range of that order the count per landmass, then every range writes behind the ranges before it
        {
            const int nt = host_threads() + 2;
            std::vector<int32_t> cnt((size_t)nt * (size_t)std::max(nComp, 1), 0);
            int usedRanges = 0;
            parallel_ranges(L, [&](int64_t b, int64_t e, int t) {
                int32_t* c = cnt.data() + (size_t)t * (size_t)nComp;
                for (int64_t q = b; q < e; ++q) { const int32_t k = compIndex[comp[S.landByR[q]]]; if (k >= 0) ++c[k]; }
                int cur = usedRanges; while (cur < t + 1 && !__atomic_compare_exchange_n(&usedRanges, &cur, t + 1, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
            });
            for (int32_t k = 0; k < nComp; ++k) {                // exclusive scan over the ranges, per landmass
                int32_t at = S.compCellStart[k];
                for (int t = 0; t < usedRanges; ++t) { int32_t& c = cnt[(size_t)t * (size_t)nComp + k]; const int32_t n = c; c = at; at += n; }
            }
            parallel_ranges(L, [&](int64_t b, int64_t e, int t) {
                int32_t* c = cnt.data() + (size_t)t * (size_t)nComp;
                for (int64_t q = b; q < e; ++q) { const int32_t i = S.landByR[q]; const int32_t k = compIndex[comp[i]]; if (k >= 0) S.compCells[c[k]++] = i; }
            });
        }
    }
    lap("landmasses");
    S.staticValid = true;
    S.staticN = N;
    ++S.staticVersion;
}

// All cells in Morton order of their positions, ties in ascending id: the same keys and the same stable sort as the land list
// above, so the land cells appear in it in exactly the order of `landCell` (planet.hip: patch-major mirror for erodeComposite).
void morton_order_cells(int32_t N, const float* xyz, hvec<int32_t>& cells) {
    cells.resize(N);
    std::vector<uint32_t> keys(N);
    parallel_ranges(N, [&](int64_t b, int64_t e, int) {
        for (int64_t r = b; r < e; ++r) {
            auto q = [](float v) { int32_t t = (int32_t)((v + 1.0f) * 511.5f); return (uint32_t)(t < 0 ? 0 : (t > 1023 ? 1023 : t)); };
            cells[r] = (int32_t)r;
            keys[r] = spread3(q(xyz[3 * r])) | (spread3(q(xyz[3 * r + 1])) << 1) | (spread3(q(xyz[3 * r + 2])) << 2);
        }
    });
    radix_sort_u32(keys, cells);
}

// cellNoise of every land cell, in the compact (Morton) land order — uploaded once per land mask for the device flood
void flood_cell_noise(const FloodScratch& S, double* out) {
    parallel_ranges(S.L, [&](int64_t b, int64_t en, int) { for (int64_t i = b; i < en; ++i) out[i] = cell_noise(S.landCell[i]); });
}

namespace {
struct FloodTimer {
    bool on;
    explicit FloodTimer(const FloodScratch& S) : on(S.hooks.timing) {}
    std::chrono::steady_clock::time_point tp = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[flood] %-10s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
        tp = now;
    }
};
}  // namespace

void FloodHooks::read() {
    // test hooks (host_util.h: WO_TEST_HOOKS), then the two documented switches
    ringMin = (int32_t)test_hook_int("flood_ring_min", 4096);
    chainsMin = (int32_t)test_hook_int("flood_chains_min", 2048);
    forceDirty = (int32_t)test_hook_int("flood_force_dirty", -1);
    std::string rs;
    hasReplayStop = test_hook("flood_replay_stop", &rs); replayStop = hasReplayStop ? (float)std::atof(rs.c_str()) : 0.0f;
    replayPrefix = test_hook_int("flood_prefix", 1) != 0;
    forcePrefixPermille = (int32_t)std::min<long long>(1000, std::max<long long>(0, test_hook_int("flood_force_prefix", 0)));
    timing = std::getenv("WO_FLOOD_TIMING") != nullptr;
    const char* pn = std::getenv("WO_FLOOD_PIN");
    pin = pn && std::atoi(pn) != 0;
}

// land elevations into the compact arrays + the start state of pass 1 (:107-113); every flood call starts here: the hooks are read
void flood_gather(const float* e, FloodScratch& S) {
    S.hooks.read();
    const int32_t L = S.L;
    const bool landOrder = S.landOrder;             // e holds the land heights only, in land-index order (FloodScratch::landOrder)
    const int32_t* landCell = S.landCell.data();
    float* eL = S.eL.data();
    FloodCell* st = S.state.data();
    if ((int32_t)S.seen.size() < L) S.seen.resize(L);
    uint8_t* seen = S.seen.data();
    // (the tie stamps and the path marks of the landmass pipeline are cleared in the same sweep: two dispatches of the worker pool less per call)
    if ((int32_t)S.stamp.size() < L) S.stamp.resize(L);
    if ((int32_t)S.onPath.size() < L) S.onPath.resize(L);
    int32_t* stamp = S.stamp.data(); uint8_t* onPath = S.onPath.data();
    parallel_ranges(L, [&](int64_t b, int64_t en, int) {
        for (int64_t i = b; i < en; ++i) { const float v = landOrder ? e[i] : e[landCell[i]]; eL[i] = v; st[i].surface = v; st[i].e = v; st[i].drain = UNVISITED; st[i].root = -1; }
        std::memset(seen + b, 0, (size_t)(en - b));
        std::memset(onPath + b, 0, (size_t)(en - b));
        std::memset(stamp + b, 0, sizeof(int32_t) * (size_t)(en - b));
    });
}

// pass 1 (:118-147): the serial heap walk — the reference's order including its heap's tie mechanics
void flood_pass1_host(FloodScratch& S) {
    const double EPS = 1e-7;
    FloodTimer T(S);
    const int32_t* landCell = S.landCell.data();
    const int32_t* offL = S.offL.data();
    const int32_t* adjL = S.adjL.data();
    float* eL = S.eL.data();
    FloodCell* st = S.state.data();
    if (S.heapStore.size() < 4096) S.heapStore.resize(4096);
    KeyHeap heap(S.heapStore);
    for (size_t s = 0; s < S.seedCell.size(); ++s) {        // :118-128, ascending r
        const int32_t i = S.seedCell[s];
        st[i].drain = TO_OCEAN;
        st[i].root = (int32_t)s;
        heap.push(i, (float)((double)eL[i] + cell_noise(landCell[i])));
    }
    T.lap("init+seeds");
    // --- pass 1 (:131-147)
    while (heap.n > 0) {
        const int32_t c = heap.pop();
        {   // the cells that pop next sit in the first heap levels: start pulling their rows in now
            const FloodHeapItem* hp = S.heapStore.data();
            const size_t lim2 = heap.n < 7 ? heap.n : 7;
            for (size_t q = 0; q < lim2; ++q) {
                const int32_t cc = hp[q].cell;
                __builtin_prefetch(&offL[cc]); __builtin_prefetch(&st[cc]);
            }
            if (heap.n > 0) {
                const int32_t c0 = hp[0].cell;
                for (int32_t j = offL[c0]; j < offL[c0 + 1]; ++j) __builtin_prefetch(&st[adjL[j]]);
            }
            for (size_t q = 1; q < lim2 && q < 3; ++q) __builtin_prefetch(&adjL[offL[hp[q].cell]]);
        }
        const double lim = (double)st[c].surface + EPS;
        const int32_t iEnd = offL[c + 1];
        const int32_t rootC = st[c].root;
        for (int32_t i = offL[c]; i < iEnd; ++i) {
            const int32_t nb = adjL[i];
            FloodCell& sn = st[nb];
            if (sn.drain != UNVISITED) continue;
            sn.drain = c;
            st[nb].root = rootC;
            float k;
            if ((double)sn.e < lim) {
                st[nb].surface = (float)lim;
                k = (float)((double)st[nb].surface + cell_noise(landCell[nb]));
            } else {
                k = (float)((double)sn.e + cell_noise(landCell[nb]));
            }
            heap.push(nb, k);
        }
    }
    T.lap("pass1");
}

// ---------------------------------------------------------------------------------------------------------------
// Pass 1, one heap per landmass.  With distinct keys a binary heap pops the smallest key whatever its array looks
// like, and landmasses share neither cells nor claims, so the pops of the reference's single heap, restricted to one
// landmass, are the pops of a heap that holds only that landmass.  What a separate heap cannot reproduce is the
// single heap's choice between EQUAL keys (it depends on the array's whole history, other landmasses included), so
// that choice is never relied upon here; instead the walk tracks where it could matter:
//
//  * a TIE GROUP opens when a cell pops with key K while another entry with key K is in the heap (then, and only
//    then, the new top has key K); it lasts until a key > K pops.  Its members are the tied cells (popped in an
//    order we must not trust) and, after each, the cascade of descendants with keys < K: together a FAMILY.
//    Cascades may hold tie groups of their own (a stack of open groups).
//  * the families of a group evolve independently of their order unless a cell of one family meets a cell that
//    another family of the same group claimed.  Every claim made inside a group is stamped with the family's id
//    (ids grow monotonically, so "claimed by an earlier family of this group" is firstFam <= stamp < fam), and
//    every already-visited neighbour a popping cell sees is checked: a hit is a CONTESTED cell.
//  * no contested cell  ->  drainTo / surface are the single heap's, bit for bit, whatever it did with its ties.
//  * a contested cell whose surface (hence key) is the same under either claimant and whose key is above the
//    group's level pops after the group either way: only its drainTo is open.  It is reported in rep.alt and
//    passes 2/3 decide whether the elevations depend on it (flood_pass23_host).
//  * anything else (a claimant would change a surface, or the cell cascades inside the group)  ->  return false,
//    the caller runs the serial walk (flood_pass1_host), the reference's order by construction.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct TieGroup { float level; int32_t firstFam, fam; };
struct Contest { int32_t cell, other; float level; };

// The pops of a walk up to the first tie group that holds a contested cell (of either kind), in order.  `prefix` = the pops before the
// OUTERMOST group open at that moment opened: up to there every equal-key decision was without consequence, so any min-first walk of the
// landmass — the single heap's included — has popped exactly these cells, each claiming exactly the children it claimed here, when its
// smallest key first reaches that group's level (between top-level groups the smallest key is unique; an uncontested top-level group ends
// in the same state whatever the order inside it).  The replay of the single heap treats them like cells of a decided landmass.
struct PopLog { std::vector<int32_t> cells; int32_t n = 0, outerOpen = 0, prefix = -1; };
struct WalkStats { int64_t pops = 0, descending = 0, raised = 0, heapSum = 0, heapMax = 0; };     // WO_FLOOD_TIMING: the largest landmass's walk
template <class Heap, bool STATS = false>
void walk_landmass_h(FloodScratch& S, const int32_t* seeds, int32_t nSeeds, hvec<FloodHeapItem>& store,
                   std::vector<Contest>& contests, int64_t& nGroups, int64_t& nNested, WalkStats* ws = nullptr,
                   const FloodHeapItem* resume = nullptr, size_t nResume = 0, PopLog* log = nullptr) {
    const double EPS = 1e-7;
    const int32_t* landCell = S.landCell.data();
    const int32_t* offL = S.offL.data();
    const int32_t* adjL = S.adjL.data();
    const float* eL = S.eL.data();
    FloodCell* st = S.state.data();
    int32_t* stamp = S.stamp.data();
    // "claimed?" is asked six times per pop and answered from one byte per cell (the landmass's share of `seen` stays in the core's
    // L2; the 16-byte records it used to be read from are 4 lines per pop that the frontier, which jumps all over the landmass, had
    // to fetch from further away); heights come from the compact eL (this landmass's passes 2/3, the only writers of its eL, run
    // after its walk).  The records are only written here (claims) and read for the popped cell (prefetched while it nears the top).
    uint8_t* seen = S.seen.data();
    if (store.size() < 1024) store.resize(1024);
    Heap heap(store);             // any exact priority queue will do here (see KeyHeap4)
    if (resume) {                                           // a walk that the replay of the single heap began: its frontier, as that heap held it
        for (size_t q = 0; q < nResume; ++q) heap.push(resume[q].cell, resume[q].key);
    } else for (int32_t q = 0; q < nSeeds; ++q) {           // :118-128, ascending r
        const int32_t s = seeds[q], i = S.seedCell[s];
        st[i].drain = TO_OCEAN;
        st[i].root = s;
        seen[i] = 1;
        heap.push(i, (float)((double)eL[i] + cell_noise(landCell[i])));
    }
    heap.prime();
    int32_t famCounter = 0;
    TieGroup groups[64]; int nOpen = 0;
    float highest = -INFINITY;
    int32_t* logCells = nullptr;
    if (log) { log->n = 0; log->outerOpen = 0; log->prefix = -1; logCells = log->cells.data(); }
    while (heap.n > 0) {
        const float kc = heap.front()[0].key;
        if (STATS) { ++ws->pops; ws->heapSum += (int64_t)heap.n; if ((int64_t)heap.n > ws->heapMax) ws->heapMax = (int64_t)heap.n; if (kc < highest) ++ws->descending; else highest = kc; }
        const int32_t c = heap.pop();
        const FloodHeapItem* hp = heap.front();
        const size_t nFront = heap.front_count();
        const bool tieTop = nFront > 0 && hp[0].key == kc;
        if (logCells) logCells[log->n++] = c;
        if (tieTop || nOpen) {
            while (nOpen && kc > groups[nOpen - 1].level) --nOpen;
            if (nOpen && kc == groups[nOpen - 1].level) groups[nOpen - 1].fam = ++famCounter;       // the next tied cell: a new family
            else if (tieTop) {
                if (nOpen == 64) { contests.push_back(Contest{c, -1, kc}); if (logCells) { log->prefix = log->outerOpen; logCells = nullptr; } }     // cannot happen in practice: reported as unresolved
                else { if (nOpen) ++nNested; else if (logCells) log->outerOpen = log->n - 1; ++nGroups; ++famCounter; groups[nOpen++] = TieGroup{kc, famCounter, famCounter}; }
            }
        }
        {
            const size_t lim2 = nFront < 7 ? nFront : 7;
            for (size_t q = 0; q < lim2; ++q) {
                const int32_t cc = hp[q].cell;
                __builtin_prefetch(&offL[cc]); __builtin_prefetch(&st[cc]);
            }
            for (size_t q = 0; q < lim2 && q < 3; ++q) __builtin_prefetch(&adjL[offL[hp[q].cell]]);
        }
        const int32_t curFam = nOpen ? groups[nOpen - 1].fam : 0;
        const double lim = (double)st[c].surface + EPS;
        const int32_t iEnd = offL[c + 1];
        const int32_t rootC = st[c].root;
        for (int32_t i = offL[c]; i < iEnd; ++i) {
            const int32_t nb = adjL[i];
            if (seen[nb]) {
                if (curFam) {
                    const int32_t sv = stamp[nb];
                    if (sv) for (int g = 0; g < nOpen; ++g)
                        if (sv >= groups[g].firstFam && sv < groups[g].fam) {
                            contests.push_back(Contest{nb, c, groups[g].level});
                            if (logCells) { log->prefix = log->outerOpen; logCells = nullptr; }
                            break;
                        }
                }
                continue;
            }
            seen[nb] = 1;
            FloodCell& sn = st[nb];
            sn.drain = c;
            sn.root = rootC;
            if (curFam) stamp[nb] = curFam;
            const double en = (double)eL[nb];
            float k;
            if (en < lim) {
                sn.surface = (float)lim;
                k = (float)((double)sn.surface + cell_noise(landCell[nb]));
                if (STATS) ++ws->raised;
            } else {
                k = (float)(en + cell_noise(landCell[nb]));
            }
            heap.push(nb, k);
        }
    }
    if (logCells) log->prefix = log->n;                     // no contested cell at all
}
// landmasses of at least WO_FLOOD_RING_MIN cells (default 4096; FloodHooks: the tests run both queues) walk on the ring queue
inline bool walk_on_ring(const FloodScratch& S, int32_t nCells) { return nCells >= S.hooks.ringMin; }
void walk_landmass_with_stats(FloodScratch& S, const int32_t* seeds, int32_t nSeeds, int32_t nCells, hvec<FloodHeapItem>& store,
                              std::vector<Contest>& contests, int64_t& nGroups, int64_t& nNested, WalkStats& ws, PopLog* log = nullptr) {
    if (log && (int32_t)log->cells.size() < nCells) log->cells.resize(nCells);
    if (walk_on_ring(S, nCells)) walk_landmass_h<RingQueue, true>(S, seeds, nSeeds, store, contests, nGroups, nNested, &ws, nullptr, 0, log);
    else walk_landmass_h<KeyHeap4, true>(S, seeds, nSeeds, store, contests, nGroups, nNested, &ws, nullptr, 0, log);
}
void walk_landmass_resume(FloodScratch& S, const std::vector<FloodHeapItem>& frontier, int32_t nCells, hvec<FloodHeapItem>& store,
                          std::vector<Contest>& contests, int64_t& nGroups, int64_t& nNested) {
    if (walk_on_ring(S, nCells)) walk_landmass_h<RingQueue>(S, nullptr, 0, store, contests, nGroups, nNested, nullptr, frontier.data(), frontier.size());
    else walk_landmass_h<KeyHeap4>(S, nullptr, 0, store, contests, nGroups, nNested, nullptr, frontier.data(), frontier.size());
}
void walk_landmass(FloodScratch& S, const int32_t* seeds, int32_t nSeeds, int32_t nCells, hvec<FloodHeapItem>& store,
                   std::vector<Contest>& contests, int64_t& nGroups, int64_t& nNested, PopLog* log = nullptr) {
    // Where a walk's time goes (round 4, the 402 k-cell landmass of the bench planet; research/flood_walk_bench.py replays the walk's op log
    // on the queue alone and the recorded pop order on the expansion alone).  FIRST flood of a step (fresh terrain): heap 11 600 entries on
    // average, 88 % of the pops in ascending order, 86 % of the pushes carry the cell's own key (height + noise); queue 20-24 ms and
    // expansion 15-20 ms in the build container; 27-31 ms for the walk on the GPU box with either queue.  SECOND flood (after 150 erosion
    // iterations): 86 % of the pushes carry a RAISED key (level + EPS + noise) and wait for the level to pass their noise — 100 000 entries on
    // average, 170 000 at most, 44 % of the pops below the level already reached: 44.6-46.7 ms on the 4-ary heap, 34.1-36.9 ms on the ring
    // (GPU box, bucket widths 2^-15 ... 2^-18 alike; no difference in the build container, whose cores have twice the L2).  The binary heap
    // was slower than the 4-ary one (35 against 31-32 ms).  Built, measured and removed in round 4: the ranks of the cells' OWN keys sorted on the
    // device before the stage (4 radix passes + 3 copies: 0.8 ms), the frontier of those keys a bitmap over the ranks and only the raised keys in
    // the ring — exact (device keys and ranks == the host's arithmetic on every cell, fields == oracle), 8.7 against 20 ms on the queue-only replay
    // with ranks per landmass, but with ranks over the whole planet (a landmass's bits are sparse in it) the first flood's walk went 27.6-28.8 ->
    // 25-26 ms and the second flood's, whose keys are mostly raised, 34 -> 41-50 ms: a loss per step.
    if (log && (int32_t)log->cells.size() < nCells) log->cells.resize(nCells);
    if (walk_on_ring(S, nCells)) walk_landmass_h<RingQueue>(S, seeds, nSeeds, store, contests, nGroups, nNested, nullptr, nullptr, 0, log);
    else walk_landmass_h<KeyHeap4>(S, seeds, nSeeds, store, contests, nGroups, nNested, nullptr, nullptr, 0, log);
}
int flood_workers(int64_t items) {
    static const int capThreads = [] { const char* e = std::getenv("WO_FLOOD_THREADS"); const int v = e ? std::atoi(e) : 0; return v >= 1 ? v : 24; }();
    return (int)std::max<int64_t>(1, std::min<int64_t>(std::min(host_threads(), capThreads), items));
}
}  // namespace

bool flood_pass1_landmasses(FloodScratch& S, FloodTieReport& rep) {
    const double EPS = 1e-7;
    FloodTimer T(S);
    const int32_t nComp = (int32_t)S.compSize.size();
    rep = FloodTieReport{};
    rep.landmasses = nComp;
    if (nComp == 0) return true;
    parallel_ranges(S.L, [&](int64_t b, int64_t en, int) { std::memset(S.stamp.data() + b, 0, sizeof(int32_t) * (size_t)(en - b)); });
    const int nt = flood_workers(nComp);
    rep.workers = nt;
    if ((int)S.workerHeaps.size() < nt) S.workerHeaps.resize(nt);
    std::vector<std::vector<Contest>> contests(nt);
    std::vector<int64_t> ng(nt, 0), nn(nt, 0);
    std::atomic<int32_t> next{0};
    auto worker = [&](int w) {
        for (;;) {
            const int32_t k = next.fetch_add(1);
            if (k >= nComp) break;
            walk_landmass(S, S.compSeeds.data() + S.compSeedStart[k], S.compSeedStart[k + 1] - S.compSeedStart[k], S.compSize[k], S.workerHeaps[w], contests[w], ng[w], nn[w]);
        }
    };
    if (nt == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int w = 0; w < nt; ++w) th.emplace_back(worker, w);
        for (auto& t : th) t.join();
    }
    const FloodCell* st = S.state.data();
    for (int w = 0; w < nt; ++w) {
        rep.groups += ng[w]; rep.nested += nn[w];
        for (const Contest& ct : contests[w]) {
            ++rep.contested;
            if (ct.other < 0) { ++rep.unresolved; continue; }
            const FloodCell& x = st[ct.cell];
            const float kx = (float)((double)x.surface + cell_noise(S.landCell[ct.cell]));
            const double limO = (double)st[ct.other].surface + EPS;
            const float altSurface = ((double)x.e < limO) ? (float)limO : x.e;
            const bool sameSurface = std::memcmp(&altSurface, &x.surface, 4) == 0;
            // pass 3 visits cells by ascending surface: both possible targets must be final when the cell's turn comes
            const bool ordered = x.drain >= 0 && st[x.drain].surface < x.surface && st[ct.other].surface < x.surface;
            if (kx > ct.level && sameSurface && ordered) { ++rep.openParents; rep.alt.push_back({ct.cell, ct.other}); }
            else ++rep.unresolved;
        }
    }
    T.lap("pass1 par");
    if (T.on) std::fprintf(stderr, "[flood] landmasses %d, workers %d, tie groups %lld (nested %lld), contested %lld, open parents %lld, unresolved %lld\n",
                           rep.landmasses, rep.workers, (long long)rep.groups, (long long)rep.nested, (long long)rep.contested, (long long)rep.openParents, (long long)rep.unresolved);
    return rep.unresolved == 0;
}

// results of the device pass 1 (flood_kernels.h) into the host state of passes 2 and 3: parent in land-index space
// (FL_NONE = -1 unreached, FL_SEED = -2 drains to the open ocean), surface, tree id (position of the tree's seed)
void flood_import_pass1(const int32_t* par, const float* surface, const int32_t* root, FloodScratch& S) {
    FloodCell* st = S.state.data();
    parallel_ranges(S.L, [&](int64_t b, int64_t en, int) {
        for (int64_t i = b; i < en; ++i) {
            const int32_t p = par[i];
            st[i].drain = (p == -2) ? TO_OCEAN : (p == -1 ? UNVISITED : p);
            st[i].surface = surface[i];
            st[i].root = (p == -1) ? -1 : root[i];
        }
    });
}

// Passes 2 and 3 of one drainage tree (:152-214).  cells: the tree's land indices in ascending ORIGINAL id (the order pass 2
// visits them in); on return sorted by (surface, id), the order of pass 3.  surface may be null (then st[].surface).
namespace {
struct TreeCtx { const FloodCell* st; float* eL; const float* surface; double carveStrength; uint8_t* onPath; int32_t* localIdx = nullptr; int32_t chainsMin = 0; };

// Pass 2 of a BIG tree (:152-196), same turns in the same order on the same values, laid out for the memory system.  A turn of the
// reference walks the whole drain path of its cell (find the peak), then rewrites a window of 0.6 x its length around the peak: in
// the 41 k-cell tree of the bench planet's largest landmass that is 20 k turns x 360 cells = 7.2 M dependent pointer steps
// through 16-byte records plus as many gathered read-modify-writes, 15-20 ms on one thread AFTER the landmass's walk — the tail
// of the flood stage.  Here the tree is first cut into chains (heavy-path decomposition: every cell continues its chain into the
// child with the largest subtree, so a path to the sea crosses at most log2(n) chains) and the heights are copied into one
// array in chain order, upstream end first: the drain path of any cell is then a handful of contiguous stretches, the peak
// search is a linear scan, the carve a linear read-modify-write with the weights (which depend on the distance to the peak only)
// taken from a small array computed once per turn instead of twice.  The arithmetic is the reference's, expression for expression
// (kernel sum accumulated in path order, weight = term / sum, float store, clamp at 0).  Returns false (nothing changed) if the
// tree is not a single rooted tree of claimed cells — the caller then takes the plain form.
// trees of at least this many cells take the chain form (WO_FLOOD_CHAINS_MIN, read per call: the tests run both forms; 0 = never)
struct ChainScratch {
    std::vector<int32_t> par, kids, kidStart, order, sz, heavy, posOf, headPos, jump, cellAt, depth;
    std::vector<float> E, S;
    std::vector<double> term;
};
bool tree_pass2_chains(const TreeCtx& X, const int32_t* cells, int32_t n, int64_t& lenSum, int64_t& nDeficit) {
    const double EPS = 1e-7;
    static thread_local ChainScratch C;
    const FloodCell* st = X.st;
    float* eL = X.eL;
    int32_t* loc = X.localIdx;
    const double carveStrength = X.carveStrength;
    auto surf = [&](int32_t i) { return X.surface ? X.surface[i] : st[i].surface; };
    for (int32_t q = 0; q < n; ++q) loc[cells[q]] = q;
    C.par.resize(n); C.kidStart.assign((size_t)n + 1, 0);
    int32_t rootQ = -1, nRoots = 0;
    for (int32_t q = 0; q < n; ++q) {
        const int32_t d = st[cells[q]].drain;
        if (d >= 0) { const int32_t pq = loc[d]; if (pq < 0 || pq >= n || cells[pq] != d) return false; C.par[q] = pq; ++C.kidStart[pq + 1]; }
        else { C.par[q] = -1; rootQ = q; ++nRoots; }
    }
    if (nRoots != 1) return false;
    for (int32_t q = 0; q < n; ++q) C.kidStart[q + 1] += C.kidStart[q];
    C.kids.resize(n); C.order.resize(n);
    {   // children lists, then breadth-first order from the seed cell (parents before children)
        std::vector<int32_t>& fill = C.sz; fill.assign(C.kidStart.begin(), C.kidStart.end() - 1);
        for (int32_t q = 0; q < n; ++q) if (C.par[q] >= 0) C.kids[fill[C.par[q]]++] = q;
        int32_t head = 0, tail = 0;
        C.order[tail++] = rootQ;
        while (head < tail) { const int32_t v = C.order[head++]; for (int32_t j = C.kidStart[v]; j < C.kidStart[v + 1]; ++j) C.order[tail++] = C.kids[j]; }
        if (tail != n) return false;                        // a cycle or a second component: not a tree
    }
    C.sz.assign(n, 1); C.heavy.assign(n, -1);
    for (int32_t i = n - 1; i > 0; --i) { const int32_t v = C.order[i]; C.sz[C.par[v]] += C.sz[v]; }
    for (int32_t i = 1; i < n; ++i) { const int32_t v = C.order[i], p = C.par[v]; if (C.heavy[p] < 0 || C.sz[v] > C.sz[C.heavy[p]]) C.heavy[p] = v; }
    C.posOf.resize(n); C.headPos.resize(n); C.jump.assign(n, -1); C.cellAt.resize(n); C.E.resize(n); C.S.resize(n); C.depth.resize(n);
    C.depth[rootQ] = 0;
    for (int32_t i = 1; i < n; ++i) { const int32_t v = C.order[i]; C.depth[v] = C.depth[C.par[v]] + 1; }
    {   // chain by chain in breadth-first order of the chain heads: positions grow towards the sea inside a chain
        int32_t base = 0;
        for (int32_t i = 0; i < n; ++i) {
            const int32_t v = C.order[i];
            if (C.par[v] >= 0 && C.heavy[C.par[v]] == v) continue;         // continues its parent's chain
            int32_t m = 0;
            for (int32_t u = v; u >= 0; u = C.heavy[u]) ++m;
            const int32_t top = base + m - 1;
            int32_t at = top;
            for (int32_t u = v; u >= 0; u = C.heavy[u], --at) { C.posOf[u] = at; C.headPos[at] = top; C.cellAt[at] = cells[u]; C.E[at] = eL[cells[u]]; C.S[at] = surf(cells[u]); }
            C.jump[top] = C.par[v] >= 0 ? C.posOf[C.par[v]] : -1;        // the head's receiver: its chain was laid out earlier
            base += m;
        }
    }
    float* E = C.E.data();
    const float* S = C.S.data();
    const int32_t* headPos = C.headPos.data();
    const int32_t* jump = C.jump.data();
    int32_t segA[64], segB[64], segK[64];
    for (int32_t q = 0; q < n; ++q) {
        const int32_t r = cells[q], pr = C.posOf[q];
        const double deficit = (double)surf(r) - (double)E[pr];      // against the CURRENT height (:154)
        if (deficit <= EPS) continue;
        ++nDeficit;
        // The path's length is the cell's depth in the tree; its peak (the FIRST maximum of the current heights, :160-165) is found without walking
        // all of it: a height never exceeds its cell's flood surface (pass 2 lowers heights, and lifts the deficit cell by less than its deficit),
        // and the surfaces do not increase towards the sea (pass 1: a claimed cell's surface is at least its claimant's), so once the surface of
        // the next cell is no higher than the maximum in hand nothing further down can beat it.  The segments are then only followed as far as the
        // carve window reaches.  (A tracked landmass marks every cell of the path: no shortcut there.)
        const int32_t len = C.depth[q] + 1;
        int nseg = 0; int32_t seen = 0, peakIdx = -1;
        float peakElev = -INFINITY;
        bool scanning = true;
        const bool whole = X.onPath != nullptr;
        int32_t need = len;                                  // path cells the segment list has to cover (shrinks to the window's end when the scan stops)
        for (int32_t a = pr; a >= 0 && seen < need;) {
            const int32_t b = headPos[a];
            segA[nseg] = a; segB[nseg] = b; segK[nseg] = seen; ++nseg;
            if (scanning) {
                int32_t p = a;
                for (; p <= b; ++p) {
                    if (!whole && S[p] <= peakElev) { scanning = false; break; }
                    if (E[p] > peakElev) { peakElev = E[p]; peakIdx = seen + (p - a); }
                }
                if (!scanning) {
                    const double rcs = std::ceil((double)len * 0.3);
                    const int32_t rad = rcs > 3.0 ? (int32_t)rcs : 3;
                    need = peakIdx + rad + 1 < len ? peakIdx + rad + 1 : len;
                }
            }
            seen += b - a + 1;
            a = jump[b];
        }
        if (X.onPath) for (int s = 0; s < nseg; ++s) for (int32_t p = segA[s]; p <= segB[s]; ++p) X.onPath[C.cellAt[p]] = 1;
        lenSum += len;
        if (peakIdx < 0) continue;
        const double carveAmount = deficit * carveStrength;
        const double rc = std::ceil((double)len * 0.3);
        const int32_t radius = rc > 3.0 ? (int32_t)rc : 3;
        const int32_t k0 = peakIdx - radius > 0 ? peakIdx - radius : 0;
        const int32_t k1 = peakIdx + radius < len - 1 ? peakIdx + radius : len - 1;
        const int32_t m = k1 - k0 + 1;
        if ((int32_t)C.term.size() < m) C.term.resize((size_t)m + 256);
        double* term = C.term.data();
        for (int32_t i = 0; i < m; ++i) term[i] = 1 - std::fabs((double)(k0 + i - peakIdx)) / (radius + 1);
        double kernelSum = 0;
        for (int32_t i = 0; i < m; ++i) kernelSum += term[i];
        if (kernelSum > 0) {
            for (int s = 0; s < nseg; ++s) {
                const int32_t ks = segK[s], ke = ks + (segB[s] - segA[s]);
                const int32_t lo = k0 > ks ? k0 : ks, hi = k1 < ke ? k1 : ke;
                float* Es = E + (segA[s] - ks);
                const double* ts = term - k0;
                for (int32_t k = lo; k <= hi; ++k) {
                    const double w = ts[k] / kernelSum;
                    float v = (float)((double)Es[k] - carveAmount * w);
                    if (v < 0) v = 0;
                    Es[k] = v;
                }
            }
        }
        E[pr] = (float)((double)E[pr] + deficit * (1 - carveStrength));
    }
    for (int32_t q = 0; q < n; ++q) eL[cells[q]] = E[C.posOf[q]];
    return true;
}

void tree_pass23(const TreeCtx& X, int32_t* cells, int32_t n, std::vector<int32_t>& path, int64_t& lenSum, int64_t& nDeficit) {
    const double EPS = 1e-7;
    const FloodCell* st = X.st;
    float* eL = X.eL;
    const double carveStrength = X.carveStrength;
    uint8_t* onPath = X.onPath;
    auto surf = [&](int32_t i) { return X.surface ? X.surface[i] : st[i].surface; };
    const bool chained = X.localIdx && X.chainsMin > 0 && n >= X.chainsMin && tree_pass2_chains(X, cells, n, lenSum, nDeficit);

    for (int32_t q = 0; q < n && !chained; ++q) {
        const int32_t r = cells[q];
        const double deficit = (double)surf(r) - (double)eL[r];      // against the CURRENT height (:154)
        if (deficit <= EPS) continue;
        ++nDeficit;
        path.clear();
        int32_t peakIdx = -1;
        double peakElev = -INFINITY;
        for (int32_t cur = r; cur >= 0; cur = st[cur].drain) {
            path.push_back(cur);
            if ((double)eL[cur] > peakElev) { peakElev = eL[cur]; peakIdx = (int32_t)path.size() - 1; }
        }
        if (onPath) for (int32_t c : path) onPath[c] = 1;
        const int32_t len = (int32_t)path.size();
        lenSum += len;
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
                float v = (float)((double)eL[path[k]] - carveAmount * w);
                if (v < 0) v = 0;
                eL[path[k]] = v;
            }
        }
        eL[r] = (float)((double)eL[r] + deficit * (1 - carveStrength));
    }
    // --- pass 3 for this tree (:199-214): its cells by ascending surface, ties by ascending original id (the
    // reference's stable sort of an ascending-id list), then the ordered fix-up.  The tree's cells are hot in cache.
    if (n > 1) {
        if (n <= 2048) {
            std::stable_sort(cells, cells + n, [&](int32_t a, int32_t b) { return asc_bits(surf(a)) < asc_bits(surf(b)); });
        } else {                                        // big trees: stable LSD radix on the key bits
            std::vector<uint32_t> k0(n), k1(n); std::vector<int32_t> c1(n);
            for (int32_t i = 0; i < n; ++i) k0[i] = asc_bits(surf(cells[i]));
            uint32_t* ka = k0.data(); uint32_t* kb = k1.data(); int32_t* ca = cells; int32_t* cb = c1.data();
            for (int pass = 0; pass < 3; ++pass) {
                const int sh = pass * 11; const uint32_t mask = pass == 2 ? 1023u : 2047u;
                uint32_t cnt[2049]; std::memset(cnt, 0, sizeof(cnt));
                for (int32_t i = 0; i < n; ++i) cnt[((ka[i] >> sh) & mask) + 1]++;
                for (int i = 0; i < 2048; ++i) cnt[i + 1] += cnt[i];
                for (int32_t i = 0; i < n; ++i) { const uint32_t d = cnt[(ka[i] >> sh) & mask]++; kb[d] = ka[i]; cb[d] = ca[i]; }
                std::swap(ka, kb); std::swap(ca, cb);
            }
            if (ca != cells) std::memcpy(cells, ca, sizeof(int32_t) * (size_t)n);     // 3 passes: result sits in c1
        }
    }
    for (int32_t q = 0; q < n; ++q) {
        const int32_t c = cells[q], t = st[c].drain;
        if (t == NO_TARGET || t == UNVISITED) continue;
        const double te = (t == TO_OCEAN) ? 0.0 : (double)eL[t];
        if ((double)eL[c] <= te) eL[c] = (float)(te + EPS);
    }
}
}  // namespace

// passes 2 and 3 (:152-214) on the state pass 1 left, then the land elevations back into e
bool flood_pass23_host(float* e, double carveStrength, FloodScratch& S, const std::vector<std::pair<int32_t, int32_t>>* openAlt) {
    FloodTimer T(S);
    // Cells whose parent pass 1 left open (equal keys, same surface under either parent).  The elevations do not depend
    // on the choice when (i) no carve path runs through the cell — a path exists only below a deficit cell, so then the
    // cell keeps its height through pass 2 and both parents see the same carves — and (ii) pass 3 leaves the cell alone
    // under either parent (it stands above both parents' final heights).  Checked after pass 3; otherwise: false.
    const bool track = openAlt && !openAlt->empty();
    uint8_t* onPath = nullptr;
    if (track) {
        S.onPath.resize(S.L);
        onPath = S.onPath.data();
        parallel_ranges(S.L, [&](int64_t b, int64_t en, int) { std::memset(onPath + b, 0, (size_t)(en - b)); });
    }
    const bool timing = T.on;
    const int32_t L = S.L;
    const int32_t* landCell = S.landCell.data();
    float* eL = S.eL.data();
    FloodCell* st = S.state.data();
    // passes 2 and 3 stream over surface / tree id: give them compact arrays again
    float* surface = S.surface.data();
    int32_t* root = S.root.data();
    parallel_ranges(L, [&](int64_t b, int64_t en, int) { for (int64_t i = b; i < en; ++i) { surface[i] = st[i].surface; root[i] = st[i].root; } });
    // --- pass 2 (:152-196) and pass 3 (:200-214).  Both are sequential in the reference, but every cell a turn
    // reads or writes (the drain path of r, the carve window on it, r itself; in pass 3 the cell and its
    // drain target) lies inside r's drainage tree, and trees share no land cell.  So trees are processed
    // concurrently while each tree keeps the reference's order (ascending r in pass 2, ascending (surface, r)
    // in pass 3): identical results, no relaxation.
    const int32_t nTrees = (int32_t)S.seedCell.size();
    std::vector<int32_t> cnt2(nTrees + 1, 0);
    // all cells of each tree, ascending ORIGINAL id inside.  (Not just the cells with an initial deficit: a carve
    // lowers other cells of the path below their flood surface, and the reference tests `deficit > EPS` against
    // the current height when it reaches them, :154-155.)
    int32_t* list2 = S.list2.data();
    {   // stable counting sort of the ascending-id land list by tree, in parallel: per-worker counts per tree
        // ([tree][worker] so the prefix runs sequentially), then every worker scatters its own contiguous share
        const int32_t* byR = S.landByR.data();
        const int T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)host_threads(), 16, (int64_t)L / 65536 + 1}));
        std::vector<int32_t> cw((size_t)nTrees * T, 0);
        auto share = [&](int t, int64_t& b, int64_t& e) { b = (int64_t)L * t / T; e = (int64_t)L * (t + 1) / T; };
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
            int64_t b, e; share(t, b, e);
            for (int64_t k = b; k < e; ++k) { const int32_t r = root[byR[k]]; if (r >= 0) ++cw[(size_t)r * T + t]; }
        });
        for (auto& x : th) x.join();
        int32_t run = 0;
        for (int32_t tr = 0; tr < nTrees; ++tr) {
            cnt2[tr] = run;
            for (int t = 0; t < T; ++t) { const int32_t c = cw[(size_t)tr * T + t]; cw[(size_t)tr * T + t] = run; run += c; }
        }
        cnt2[nTrees] = run;
        th.clear();
        for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
            int64_t b, e; share(t, b, e);
            for (int64_t k = b; k < e; ++k) { const int32_t i = byR[k]; const int32_t r = root[i]; if (r >= 0) list2[cw[(size_t)r * T + t]++] = i; }
        });
        for (auto& x : th) x.join();
    }
    // Trees are numbered by their seed's original id, i.e. along the Fibonacci spiral: consecutive ids are spatial
    // neighbours, so workers take contiguous chunks.  Measured on the 2-socket EPYC GPU box with the packed records and
    // huge pages: 6 workers 199 ms, 12 -> 136, 24 -> 102, 48 -> 91 ms for the two calls of a step; 24 is the default
    // (WO_FLOOD_THREADS overrides), which leaves room for several ranks / planets per socket.
    auto for_trees = [&](const std::vector<int32_t>& cnt, auto body) {
        static const int capThreads = [] { const char* e = std::getenv("WO_FLOOD_THREADS"); const int v = e ? std::atoi(e) : 0; return v >= 1 ? v : 24; }();
        const int nt = std::max(1, std::min<int>(std::min(host_threads(), capThreads), nTrees));
        const int64_t total = cnt[nTrees];
        const int64_t perChunk = std::max<int64_t>(2048, total / (nt * 16));
        std::vector<int32_t> chunkStart;
        chunkStart.push_back(0);
        for (int32_t t = 0, last = 0; t < nTrees; ++t)
            if (cnt[t + 1] - cnt[last] >= perChunk) { chunkStart.push_back(t + 1); last = t + 1; }
        if (chunkStart.back() != nTrees) chunkStart.push_back(nTrees);
        const size_t nChunks = chunkStart.size() - 1;
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                const size_t c = next.fetch_add(1);
                if (c >= nChunks) break;
                for (int32_t t = chunkStart[c]; t < chunkStart[c + 1]; ++t) if (cnt[t + 1] > cnt[t]) body(t);
            }
        };
        if (nt == 1) { worker(); return; }
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i) th.emplace_back(worker);
        for (auto& t : th) t.join();
    };
    T.lap("group2");
    std::atomic<int64_t> nDeficit{0}, totLen{0};
    const TreeCtx ctx{st, eL, surface, carveStrength, onPath, S.localIdx.data(), S.hooks.chainsMin};
    for_trees(cnt2, [&](int32_t tree) {
        static thread_local std::vector<int32_t> path;
        int64_t myLen = 0, myDef = 0;
        tree_pass23(ctx, list2 + cnt2[tree], cnt2[tree + 1] - cnt2[tree], path, myLen, myDef);
        if (timing) { totLen += myLen; nDeficit += myDef; }
    });
    if (timing) std::fprintf(stderr, "[flood] pass2: %lld deficit cells in %d trees, total path length %lld\n", (long long)nDeficit.load(), nTrees, (long long)totLen.load());
    T.lap("pass2+3");
    if (track) {
        for (const auto& oa : *openAlt) {
            const int32_t x = oa.first, p0 = st[x].drain, p1 = oa.second;
            const bool untouched = !onPath[x] && std::memcmp(&eL[x], &st[x].e, 4) == 0;
            const double h = (double)st[x].e;
            if (!(untouched && p0 >= 0 && h > (double)eL[p0] && h > (double)eL[p1])) return false;
        }
    }
    if (S.landOrder) parallel_ranges(L, [&](int64_t b, int64_t en, int) { std::memcpy(e + b, eL + b, sizeof(float) * (size_t)(en - b)); });
    else parallel_ranges(L, [&](int64_t b, int64_t en, int) { for (int64_t i = b; i < en; ++i) e[landCell[i]] = eL[i]; });
    T.lap("writeback");
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Equal keys that DO matter: the replay.  When two tied cells (or their cascades) reach the same cell and the result depends
// on who is first, only the reference's single heap knows — its choice between equal keys is a function of the array's whole
// history, every landmass included.  What the separate heaps did establish is everything else: in a landmass without such a
// cell ("clean") every cell's claimant, surface and key are the single heap's, whatever it does with ties.  So the single
// heap is run again over the whole planet, but as a bare heap: a clean cell pushes the children it is already known to
// claim, with their known keys, in adjacency order (a contiguous list, no graph walk, no state), and only the landmasses
// with an open decision ("dirty") are walked for real inside it.  The heap sees exactly the pushes and pops of the
// reference's, in the same order, so its array — and with it every choice between equal keys — is the reference's.
// ---------------------------------------------------------------------------------------------------------------
namespace {
// stopLevel: the replay ends as soon as the heap's smallest key exceeds it (+inf: runs to the end); then `frontier[k]` receives the
// entries the heap still holds for dirty landmass k, in array order, and true is returned.  See flood_landmass_pipeline.
// prefix[k]: the cells of dirty landmass k that its own walk popped before its first tie group with a contested cell (PopLog) — they, and
// the claims they made, are kept and they push like clean cells (dirty == 2); only the rest of the landmass is walked for real (dirty == 1).
bool replay_dirty_landmasses(FloodScratch& S, const std::vector<uint8_t>& dirtyComp, const float* e, float stopLevel,
                             std::vector<std::vector<FloodHeapItem>>* frontier, const std::vector<std::vector<int32_t>>* prefix) {
    const double EPS = 1e-7;
    FloodTimer T(S);
    const int32_t L = S.L;
    const int32_t* landCell = S.landCell.data();
    const int32_t* offL = S.offL.data();
    const int32_t* adjL = S.adjL.data();
    float* eL = S.eL.data();
    FloodCell* st = S.state.data();
    S.replayDirty.resize(L);
    uint8_t* dirty = S.replayDirty.data();
    parallel_ranges(L, [&](int64_t b, int64_t en, int) { std::memset(dirty + b, 0, (size_t)(en - b)); });
    const int32_t nComp = (int32_t)S.compSize.size();
    int64_t prefixLeft = 0;
    for (int32_t k = 0; k < nComp; ++k) {
        if (!dirtyComp[k]) continue;
        const int32_t* cells = S.compCells.data() + S.compCellStart[k];
        const int32_t n = S.compCellStart[k + 1] - S.compCellStart[k];
        parallel_ranges(n, [&](int64_t b, int64_t en, int) { for (int64_t q = b; q < en; ++q) dirty[cells[q]] = 1; });
        if (prefix) {
            const std::vector<int32_t>& pre = (*prefix)[k];
            parallel_ranges((int64_t)pre.size(), [&](int64_t b, int64_t en, int) { for (int64_t q = b; q < en; ++q) dirty[pre[q]] = 2; });
            prefixLeft += (int64_t)pre.size();
        }
        parallel_ranges(n, [&](int64_t b, int64_t en, int) {
            for (int64_t q = b; q < en; ++q) {                       // back to the start state of pass 1 (flood_gather) ...
                const int32_t i = cells[q];
                S.localIdx[i] = k;                                   // (scratch of the carve pass, free until round 2: which landmass a frontier entry belongs to)
                const float v = st[i].e;                             // the height at the start of the call (flood_gather's copy: e may BE eL, which round 1 has carved)
                eL[i] = v;
                const int32_t par = st[i].drain;                     // ... except what the prefix popped or claimed (a claimant is a neighbour: same landmass)
                if (dirty[i] == 2 || (par >= 0 && dirty[par] == 2)) continue;
                st[i].surface = v; st[i].drain = UNVISITED; st[i].root = -1;
            }
        });
    }
    // children of the clean cells, in the order their claimant pushed them (adjacency order), with their keys
    S.childStart.resize((size_t)L + 1);
    int32_t* cs = S.childStart.data();
    parallel_ranges(L, [&](int64_t b, int64_t en, int) {
        for (int64_t i = b; i < en; ++i) {
            int32_t c = 0;
            if (dirty[i] != 1 && st[i].drain != UNVISITED) for (int32_t j = offL[i]; j < offL[i + 1]; ++j) c += st[adjL[j]].drain == (int32_t)i ? 1 : 0;
            cs[i + 1] = c;
        }
    });
    {   // exclusive scan, in parallel: per-range sums, then offsets
        const int nt = host_threads();
        std::vector<int64_t> part(nt + 2, 0);
        parallel_ranges(L, [&](int64_t b, int64_t en, int t) { int64_t a = 0; for (int64_t i = b; i < en; ++i) a += cs[i + 1]; part[t + 1] = a; });
        for (int t = 1; t <= nt + 1; ++t) part[t] += part[t - 1];
        cs[0] = 0;
        parallel_ranges(L, [&](int64_t b, int64_t en, int t) { int64_t run = part[t]; for (int64_t i = b; i < en; ++i) { run += cs[i + 1]; cs[i + 1] = (int32_t)run; } });
    }
    S.childItem.resize((size_t)cs[L] + 1);
    FloodHeapItem* ci = S.childItem.data();
    // An entry of this heap carries "walked for real" in the top bit of its cell (the heap compares keys only): the loop below never has
    // to look the popped cell up to know which of the two kinds it is, nor to decide what to pull in for the cells about to pop.
    constexpr int32_t REAL = INT32_MIN, CELL = INT32_MAX;
    parallel_ranges(L, [&](int64_t b, int64_t en, int) {
        for (int64_t i = b; i < en; ++i) {
            if (cs[i + 1] == cs[i]) continue;
            int32_t o = cs[i];
            for (int32_t j = offL[i]; j < offL[i + 1]; ++j) {
                const int32_t nb = adjL[j];
                if (st[nb].drain == (int32_t)i) ci[o++] = FloodHeapItem{(float)((double)st[nb].surface + cell_noise(landCell[nb])), dirty[nb] == 1 ? (nb | REAL) : nb};
            }
        }
    });
    T.lap("replay prep");
    if (S.heapStore.size() < 4096) S.heapStore.resize(4096);
    KeyHeap heap(S.heapStore);
    for (size_t s = 0; s < S.seedCell.size(); ++s) {        // :118-128, ascending r: every seed of the planet
        const int32_t i = S.seedCell[s];
        if (dirty[i] == 1) { st[i].drain = TO_OCEAN; st[i].root = (int32_t)s; }
        heap.push(dirty[i] == 1 ? (i | REAL) : i, (float)((double)st[i].e + cell_noise(landCell[i])));    // a seed's surface is its height at the start of the call (state copy: eL of a clean landmass is carved by now)
    }
    int64_t realPops = 0, pops = 0;
    bool stopped = false;
    while (heap.n > 0) {
        if (S.heapStore.data()[0].key > stopLevel && prefixLeft == 0) { stopped = true; break; }     // (a prefix cell left in the heap — test hook levels only — has children that are claimed and not yet pushed)
        const int32_t popped = heap.pop();
        const int32_t c = popped & CELL;
        ++pops;
        {   // what the next pops will touch (the first heap levels hold them): child lists of the bare cells; rows, records and — for the
            // very next one — the neighbours' records of the cells walked for real, as in flood_pass1_host
            const FloodHeapItem* hp = S.heapStore.data();
            const size_t lim2 = heap.n < 7 ? heap.n : 7;
            for (size_t q = 0; q < lim2; ++q) {
                const int32_t raw = hp[q].cell, cc = raw & CELL;
                if (raw < 0) { __builtin_prefetch(&offL[cc]); __builtin_prefetch(&st[cc]); }
                else __builtin_prefetch(&cs[cc]);
            }
            if (heap.n > 0) {
                const int32_t raw = hp[0].cell, c0 = raw & CELL;
                if (raw < 0) for (int32_t j = offL[c0]; j < offL[c0 + 1]; ++j) __builtin_prefetch(&st[adjL[j]]);
                else __builtin_prefetch(&ci[cs[c0]]);
            }
            for (size_t q = 1; q < lim2 && q < 3; ++q) if (hp[q].cell < 0) __builtin_prefetch(&adjL[offL[hp[q].cell & CELL]]);
        }
        if (popped >= 0) {
            for (int32_t j = cs[c]; j < cs[c + 1]; ++j) heap.push(ci[j].cell, ci[j].key);
            if (prefixLeft) prefixLeft -= dirty[c] >> 1;
            continue;
        }
        ++realPops;
        const double lim = (double)st[c].surface + EPS;
        const int32_t rootC = st[c].root;
        for (int32_t i = offL[c]; i < offL[c + 1]; ++i) {
            const int32_t nb = adjL[i];
            FloodCell& sn = st[nb];
            if (sn.drain != UNVISITED) continue;
            sn.drain = c;
            sn.root = rootC;
            float k;
            if ((double)sn.e < lim) { sn.surface = (float)lim; k = (float)((double)sn.surface + cell_noise(landCell[nb])); }
            else k = (float)((double)sn.e + cell_noise(landCell[nb]));
            heap.push(nb | REAL, k);
        }
    }
    if (stopped && frontier) {
        const FloodHeapItem* hp = S.heapStore.data();
        for (size_t q = 0; q < heap.n; ++q) {
            const int32_t cc = hp[q].cell & CELL;
            if (dirty[cc]) (*frontier)[S.localIdx[cc]].push_back(FloodHeapItem{hp[q].key, cc});
        }
    }
    T.lap("replay");
    if (T.on) std::fprintf(stderr, "[flood] replay: %lld cells walked for real of %d; %s after %lld pops (level %.9g), %zu entries left in the heap\n", (long long)realPops, L,
                           stopped ? "stopped" : "ran to the end", (long long)pops, (double)stopLevel, heap.n);
    return stopped;
}
}  // namespace

// Pass 1 + passes 2/3, pipelined per landmass.  A landmass's trees, carve paths and fix-ups stay inside it, so its passes
// 2/3 need only its own pass 1.  Workers take landmasses largest first; after the walk of one they resolve its contested
// cells, group its cells by tree and run the trees — a big landmass hands its trees out in chunks so that every worker
// that has run out of landmasses helps.  While the largest landmass (14 % of the land on the bench planet) is still in
// its walk, the carving of all the others is already done.  Landmasses whose walk met an equal-key decision that matters
// are left out of the first round, re-walked inside the replay of the single heap (above) and carved in a second round.
// On return e holds the reference's result; FloodScratch is consumed (the caller gathers again before another call).
bool flood_landmass_pipeline(float* e, double carveStrength, FloodScratch& S, FloodTieReport& rep, int64_t& pathRedo, bool replayAllowed) {
    const double EPS = 1e-7;
    FloodTimer T(S);
    const int32_t nComp = (int32_t)S.compSize.size();
    const int32_t L = S.L;
    rep = FloodTieReport{};
    rep.landmasses = nComp;
    // (stamp and onPath start at zero: flood_gather, which every caller runs first)
    const int nt = std::min(flood_workers(std::max(nComp, 1)), std::max(1, L / 16384));    // small planets: a thread costs more than it saves
    rep.workers = nt;
    if ((int)S.workerHeaps.size() < nt) S.workerHeaps.resize(nt);
    FloodCell* st = S.state.data();
    float* eL = S.eL.data();
    int32_t* list2 = S.list2.data();
    constexpr int32_t BIG = 32768, CHUNK = 4096;
    const int32_t chainsMin = S.hooks.chainsMin;
    struct BigJob {
        int32_t k = -1; std::vector<int32_t> cnt, chunkStart, chunkOrder; bool track = false;
        std::atomic<int> ready{0}; std::atomic<size_t> nextChunk{0}, doneChunks{0};
    };
    struct Local { std::vector<Contest> contests; std::vector<std::pair<int32_t, int32_t>> alt; std::vector<int32_t> altComp; int64_t groups = 0, nested = 0, contested = 0, unresolved = 0; float maxLevel = -INFINITY; bool noLevel = false; };
    std::vector<Local> loc(nt);
    std::vector<uint8_t> dirty(std::max(nComp, 1), 0);
    const int forceDirty = S.hooks.forceDirty;     // test hook: treat this landmass (by rank in size) as undecided
    // the pops of a landmass with a contested cell up to its first such tie group (PopLog): bare pushes in the replay, like a decided landmass
    std::vector<std::vector<int32_t>> prefix(S.hooks.replayPrefix ? std::max(nComp, 1) : 0);
    // One round over a list of landmasses.  walked: pass 1 of these landmasses is already there (the replay's).
    const auto tRound0 = std::chrono::steady_clock::now();
    auto run_round = [&](const std::vector<int32_t>& list, bool walked) {
        const int32_t nList = (int32_t)list.size();
        int32_t nBig = 0;
        while (nBig < nList && S.compSize[list[nBig]] >= BIG) ++nBig;             // lists are in descending size
        std::vector<BigJob> big(nBig);
        std::atomic<int32_t> next{0}, bigLeft{nBig};
        auto run_chunks = [&](BigJob& J, std::vector<int32_t>& path) {
            const TreeCtx ctx{st, eL, nullptr, carveStrength, J.track ? S.onPath.data() : nullptr, S.localIdx.data(), chainsMin};
            const size_t nChunks = J.chunkStart.size() - 1;
            const int32_t base = S.compCellStart[J.k];
            for (;;) {
                const size_t cq = J.nextChunk.fetch_add(1);
                if (cq >= nChunks) break;
                const size_t c = (size_t)J.chunkOrder[cq];
                int64_t a = 0, b = 0;
                for (int32_t t = J.chunkStart[c]; t < J.chunkStart[c + 1]; ++t)
                    if (J.cnt[t + 1] > J.cnt[t]) {
                        const int32_t nT = J.cnt[t + 1] - J.cnt[t];
                        const bool timed = T.on && nT >= 16384;
                        const auto tt0 = timed ? std::chrono::steady_clock::now() : tRound0;
                        const int64_t a0 = a, b0 = b;
                        tree_pass23(ctx, list2 + base + J.cnt[t], nT, path, a, b);
                        if (timed) {
                            const auto now = std::chrono::steady_clock::now();
                            std::fprintf(stderr, "[flood] landmass %d: tree of %d cells, passes 2+3 %.2f ms (%lld deficit cells, %lld path steps), done at %.1f ms\n", J.k, nT,
                                         std::chrono::duration<double, std::milli>(now - tt0).count(), (long long)(b - b0), (long long)(a - a0), std::chrono::duration<double, std::milli>(now - tRound0).count());
                        }
                    }
                if (J.doneChunks.fetch_add(1) + 1 == nChunks) bigLeft.fetch_sub(1);
            }
        };
        // The walk of the largest landmass is the critical path of the call, one thread for 27-40 ms, and its working set (~20 MB) lives in the
        // last-level cache of the core it runs on.  That thread stays on the CPU it is on and every other worker of the round keeps off the
        // CPUs that share its L3 (an EPYC CCD) until it is done (WO_FLOOD_PIN=0: nobody's affinity is touched).
        const CpuGroups& cpus = CpuGroups::get();
        const bool pinning = S.hooks.pin && !walked && nBig > 0 && !cpus.groups.empty() && nt > 1;
        std::atomic<int> reservedGroup{-1};
        auto worker = [&](int w) {
            Local& me = loc[w];
            std::vector<int32_t> path, cnt;
            static thread_local PopLog popLog;
            PopLog* const log = prefix.empty() ? nullptr : &popLog;
            AffinityScope affinity;
            bool placed = false;
            for (;;) {
                const int32_t q = next.fetch_add(1);
                if (q >= nList) break;
                const int32_t k = list[q];
                bool track = false, undecided = false;
                if (pinning && !placed) {
                    if (q == 0) {
                        const int cpu = sched_getcpu();
                        const int g = cpu >= 0 && cpu < (int)cpus.groupOf.size() ? cpus.groupOf[cpu] : -1;
                        if (g >= 0 && affinity.only(cpu)) reservedGroup.store(g, std::memory_order_release);
                        else reservedGroup.store(-2, std::memory_order_release);
                        if (T.on) std::fprintf(stderr, "[flood] the largest walk stays on cpu %d; the other workers keep off its L3 (group %d of %zu)\n", cpu, g, cpus.groups.size());
                        placed = true;
                    } else {
                        const int g = reservedGroup.load(std::memory_order_acquire);
                        if (g >= 0) { (void)affinity.all_but_group(cpus, g); placed = true; }
                        else if (g == -2) placed = true;
                    }
                }
                if (!walked) {
                    // --- pass 1 of landmass k
                    me.contests.clear();
                    const auto tw0 = std::chrono::steady_clock::now();
                    WalkStats ws;
                    if (T.on && q == 0) walk_landmass_with_stats(S, S.compSeeds.data() + S.compSeedStart[k], S.compSeedStart[k + 1] - S.compSeedStart[k], S.compSize[k], S.workerHeaps[w], me.contests, me.groups, me.nested, ws, log);
                    else walk_landmass(S, S.compSeeds.data() + S.compSeedStart[k], S.compSeedStart[k + 1] - S.compSeedStart[k], S.compSize[k], S.workerHeaps[w], me.contests, me.groups, me.nested, log);
                    if (log && (!me.contests.empty() || q == forceDirty)) {
                        // (the test hook's landmass has no contested cell: every cut of its pops is as good as any; WO_FLOOD_FORCE_PREFIX permille of them)
                        const int32_t len = q == forceDirty && me.contests.empty() ? (int32_t)((int64_t)log->n * S.hooks.forcePrefixPermille / 1000) : log->prefix;
                        prefix[k].assign(log->cells.begin(), log->cells.begin() + len);
                        if (T.on) std::fprintf(stderr, "[flood] landmass %d (%d cells): %d pops before its first tie group with a contested cell\n", k, S.compSize[k], len);
                    }
                    // (measured and dropped in round 3: a bucket queue — 2^16 buckets of width 2^-14 behind a two-level bitmap — instead of the
                    // binary heap for the walks, which do not depend on the order of equal keys: 83-130 ms against 45-60 ms for this landmass in
                    // the build container; the heap of one landmass stays in cache, the buckets' vectors do not)
                    if (T.on && q == 0) std::fprintf(stderr, "[flood] walk of the largest landmass (%d cells): %.1f ms; heap mean %lld max %lld entries, %lld pops below the level reached, %lld raised keys, %d seeds\n", S.compSize[k], std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count(),
                                                     (long long)(ws.heapSum / std::max<int64_t>(ws.pops, 1)), (long long)ws.heapMax, (long long)ws.descending, (long long)ws.raised, S.compSeedStart[k + 1] - S.compSeedStart[k]);
                    const size_t alt0 = me.alt.size();
                    for (const Contest& ct : me.contests) {
                        ++me.contested;
                        bool open = false;
                        if (ct.other < 0) me.noLevel = true; else if (ct.level > me.maxLevel) me.maxLevel = ct.level;
                        if (ct.other >= 0) {
                            const FloodCell& x = st[ct.cell];
                            const float kx = (float)((double)x.surface + cell_noise(S.landCell[ct.cell]));
                            const double limO = (double)st[ct.other].surface + EPS;
                            const float altSurface = ((double)x.e < limO) ? (float)limO : x.e;
                            const bool sameSurface = std::memcmp(&altSurface, &x.surface, 4) == 0;
                            const bool ordered = x.drain >= 0 && st[x.drain].surface < x.surface && st[ct.other].surface < x.surface;
                            open = kx > ct.level && sameSurface && ordered;
                            if (!open && T.on) std::fprintf(stderr, "[flood] undecided contested cell %d (other %d): %s%s%s level %.9g key %.9g surface %.9g / %.9g e %.9g\n", S.landCell[ct.cell], S.landCell[ct.other],
                                                            kx > ct.level ? "" : "cascades-inside-the-group ", sameSurface ? "" : "surface-differs ", ordered ? "" : "pass3-order ", (double)ct.level, (double)kx,
                                                            (double)x.surface, (double)altSurface, (double)x.e);
                        }
                        if (open) { me.alt.push_back({ct.cell, ct.other}); me.altComp.push_back(k); track = true; }
                        else { ++me.unresolved; undecided = true; }
                    }
                    if (q == forceDirty) undecided = true;
                    if (undecided) { me.alt.resize(alt0); me.altComp.resize(alt0); dirty[k] = 1; }
                }
                // --- its cells grouped by tree (stable: ascending original id inside a tree); an undecided landmass waits for the replay
                const bool isBig = q < nBig;
                const int32_t base = S.compCellStart[k], n = S.compCellStart[k + 1] - base, nTrees = S.compSeedStart[k + 1] - S.compSeedStart[k];
                std::vector<int32_t>& c = isBig ? big[q].cnt : cnt;
                if (undecided) {
                    if (isBig) { BigJob& J = big[q]; J.k = k; J.chunkStart.assign(1, 0); J.ready.store(1, std::memory_order_release); bigLeft.fetch_sub(1); }
                    continue;
                }
                const int32_t* cells = S.compCells.data() + base;
                c.assign((size_t)nTrees + 1, 0);
                for (int32_t qq = 0; qq < n; ++qq) ++c[S.seedLocal[st[cells[qq]].root] + 1];
                for (int32_t t = 0; t < nTrees; ++t) c[t + 1] += c[t];
                {
                    static thread_local std::vector<int32_t> pos;
                    pos.assign(c.begin(), c.end() - 1);
                    for (int32_t qq = 0; qq < n; ++qq) { const int32_t i = cells[qq]; list2[base + pos[S.seedLocal[st[i].root]]++] = i; }
                }
                if (T.on && q == 0) std::fprintf(stderr, "[flood] largest landmass: walk + contests + tree lists done at %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tRound0).count());
                if (isBig) {                                  // hand the trees out in chunks
                    BigJob& J = big[q];
                    J.k = k; J.track = track;
                    J.chunkStart.clear(); J.chunkStart.push_back(0);
                    for (int32_t t = 0, last = 0; t < nTrees; ++t) if (c[t + 1] - c[last] >= CHUNK) { J.chunkStart.push_back(t + 1); last = t + 1; }
                    if (J.chunkStart.back() != nTrees) J.chunkStart.push_back(nTrees);
                    // biggest chunks first: the largest tree (41 k of this landmass's 402 k cells on the bench planet) is the longest single job of the tail
                    J.chunkOrder.resize(J.chunkStart.size() - 1);
                    for (size_t x = 0; x < J.chunkOrder.size(); ++x) J.chunkOrder[x] = (int32_t)x;
                    std::stable_sort(J.chunkOrder.begin(), J.chunkOrder.end(), [&](int32_t x, int32_t y) { return c[J.chunkStart[x + 1]] - c[J.chunkStart[x]] > c[J.chunkStart[y + 1]] - c[J.chunkStart[y]]; });
                    J.ready.store(1, std::memory_order_release);
                    run_chunks(J, path);
                } else {
                    const TreeCtx ctx{st, eL, nullptr, carveStrength, track ? S.onPath.data() : nullptr, S.localIdx.data(), chainsMin};
                    int64_t a = 0, b = 0;
                    for (int32_t t = 0; t < nTrees; ++t) if (c[t + 1] > c[t]) tree_pass23(ctx, list2 + base + c[t], c[t + 1] - c[t], path, a, b);
                }
            }
            // --- no landmass left to start: help with the big ones until all of them are through
            while (bigLeft.load() > 0) {
                bool did = false;
                for (int32_t q = 0; q < nBig; ++q) {
                    BigJob& J = big[q];
                    if (J.ready.load(std::memory_order_acquire) && J.nextChunk.load() < J.chunkStart.size() - 1) { run_chunks(J, path); did = true; }
                }
                if (!did) std::this_thread::sleep_for(std::chrono::microseconds(25));      // not a busy wait: the walk of the largest landmass is the critical path and may share a core
            }
        };
        const int use = std::min(nt, std::max(1, nList));
        if (T.on) std::fprintf(stderr, "[flood] round setup done at %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tRound0).count());
        if (use == 1 && nBig == 0) worker(0);
        else {
            const int th_n = nBig > 0 ? nt : use;                  // chunks of a big landmass are worth every worker
            // on the persistent host workers (host_util.h: HostPool; threads of its own when the pool is busy with another planet)
            parallel_ranges((int64_t)th_n, [&](int64_t b, int64_t en, int) { for (int64_t w = b; w < en; ++w) worker((int)w); }, 1);
        }
        if (T.on) std::fprintf(stderr, "[flood] round joined at %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tRound0).count());
    };
    std::vector<int32_t> all(nComp);
    for (int32_t k = 0; k < nComp; ++k) all[k] = k;
    run_round(all, false);
    T.lap("pipeline");
    // open parents (see flood_pass23_host): the elevations must not depend on the choice; a landmass where they do is undecided after all
    for (const Local& l : loc)
        for (size_t a = 0; a < l.alt.size(); ++a) {
            const int32_t x = l.alt[a].first, p0 = st[x].drain, p1 = l.alt[a].second;
            const bool untouched = !S.onPath[x] && std::memcmp(&eL[x], &st[x].e, 4) == 0;
            const double h = (double)st[x].e;
            if (!(untouched && p0 >= 0 && h > (double)eL[p0] && h > (double)eL[p1])) { if (!dirty[l.altComp[a]]) { dirty[l.altComp[a]] = 1; ++pathRedo; } }
        }
    for (const Local& l : loc) { rep.groups += l.groups; rep.nested += l.nested; rep.contested += l.contested; rep.unresolved += l.unresolved; rep.openParents += (int64_t)l.alt.size(); }
    std::vector<int32_t> redo;
    for (int32_t k = 0; k < nComp; ++k) if (dirty[k]) redo.push_back(k);
    if (!redo.empty()) {
        // A replay is going to run.  A landmass that is otherwise decided but holds an open parent x (claimants p0, p1 with equal
        // keys) would push x when the walk's claimant p0 pops; the reference may push it when p1 pops — same result for the
        // landmass, but a different heap array from then on, i.e. possibly a different choice between equal keys elsewhere.
        // Such landmasses are walked for real inside the replay as well (they are rare and the replay's cost is the bare heap's).
        bool added = false;
        for (const Local& l : loc) for (int32_t k : l.altComp) if (!dirty[k]) { dirty[k] = 1; added = true; }
        if (added) { redo.clear(); for (int32_t k = 0; k < nComp; ++k) if (dirty[k]) redo.push_back(k); }
    }
    rep.replayed = (int32_t)redo.size();
    if (!redo.empty() && !replayAllowed) return false;
    if (T.on) std::fprintf(stderr, "[flood] landmasses %d, workers %d, tie groups %lld (nested %lld), contested %lld, open parents %lld, undecided %lld -> %d landmasses through the replay\n",
                           rep.landmasses, rep.workers, (long long)rep.groups, (long long)rep.nested, (long long)rep.contested, (long long)rep.openParents, (long long)rep.unresolved, rep.replayed);
    if (!redo.empty()) {
        // The replay has to reproduce the reference's heap only up to the last equal-key decision that matters: every contested cell of
        // the first round (open parents included: they change the array) belongs to a tie group, and once the heap's smallest key has
        // passed the highest of those groups' levels no decision is left that the landmass walks could not vouch for themselves.  From
        // there each undecided landmass goes on alone, on a queue of its own seeded with the entries the single heap held for it —
        // concurrently, with the tie bookkeeping of a first-round walk; should that meet a contested cell after all (the history below
        // the level may differ from the first round's), the replay is run again, to the end.
        float stopLevel = -INFINITY;
        bool toTheEnd = forceDirty >= 0;                     // (the test hook has no level: its landmass is replayed in full)
        for (const Local& l : loc) { if (l.noLevel) toTheEnd = true; if (l.maxLevel > stopLevel) stopLevel = l.maxLevel; }
        if (toTheEnd) stopLevel = INFINITY;
        if (S.hooks.hasReplayStop) stopLevel = S.hooks.replayStop;      // test hook
        std::vector<std::vector<FloodHeapItem>> frontier(nComp);
        if (replay_dirty_landmasses(S, dirty, e, stopLevel, &frontier, prefix.empty() ? nullptr : &prefix)) {
            std::atomic<int> contestedAgain{0};
            uint8_t* seen = S.seen.data();
            int32_t* stamp = S.stamp.data();
            parallel_ranges((int64_t)redo.size(), [&](int64_t b, int64_t en, int) {
                for (int64_t q = b; q < en; ++q) {
                    const int32_t k = redo[q];
                    const int32_t* cells = S.compCells.data() + S.compCellStart[k];
                    const int32_t n = S.compCellStart[k + 1] - S.compCellStart[k];
                    for (int32_t x = 0; x < n; ++x) { const int32_t i = cells[x]; seen[i] = st[i].drain != UNVISITED; stamp[i] = 0; }
                    static thread_local hvec<FloodHeapItem> store;
                    std::vector<Contest> contests; int64_t g = 0, nn = 0;
                    walk_landmass_resume(S, frontier[k], n, store, contests, g, nn);
                    if (!contests.empty()) contestedAgain.fetch_add(1);
                }
            }, 1);
            T.lap("resumed");
            if (T.on) std::fprintf(stderr, "[flood] %zu landmasses resumed on their own queues above level %.9g%s\n", redo.size(), (double)stopLevel, contestedAgain.load() ? "; a contested cell turned up: full replay" : "");
            if (contestedAgain.load()) replay_dirty_landmasses(S, dirty, e, INFINITY, nullptr, prefix.empty() ? nullptr : &prefix);
        }
        run_round(redo, true);
        T.lap("round 2");
    }
    const int32_t* landCell = S.landCell.data();
    if (S.landOrder) parallel_ranges(L, [&](int64_t b, int64_t en, int) { std::memcpy(e + b, eL + b, sizeof(float) * (size_t)(en - b)); });
    else parallel_ranges(L, [&](int64_t b, int64_t en, int) { for (int64_t i = b; i < en; ++i) e[landCell[i]] = eL[i]; });
    T.lap("writeback");
    return true;
}

// gather + pass 1 + passes 2/3 on the host.  Pass 1 runs one heap per landmass on the tree workers; landmasses where that
// cannot vouch for the single heap's result (FloodTieReport) are decided by the replay of the single heap
// (replay_dirty_landmasses).  WO_FLOOD_HOST=serial: the plain serial walk; =two-phase: all walks, then all trees, serial
// walk when undecided (round 2's form, kept for comparison).
void flood_host_passes(float* e, double carveStrength, FloodScratch& S, FloodHostStats* stats) {
    static const bool serialOnly = [] { const char* v = std::getenv("WO_FLOOD_HOST"); return v && std::string(v) == "serial"; }();
    using clock = std::chrono::steady_clock;
    auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    FloodHostStats local;
    FloodHostStats& st = stats ? *stats : local;
    ++st.calls;
    flood_gather(e, S);
    bool done = false;
    if (!serialOnly) {
        static const bool twoPhase = [] { const char* v = std::getenv("WO_FLOOD_HOST"); return v && std::string(v) == "two-phase"; }();
        FloodTieReport rep;
        auto t0 = clock::now();
        if (twoPhase) {                                     // all walks first, then all trees (kept for comparison)
            const bool exact = flood_pass1_landmasses(S, rep);
            auto t1 = clock::now();
            st.pass1Ms += ms(t0, t1);
            if (exact) {
                done = flood_pass23_host(e, carveStrength, S, &rep.alt);
                st.pass23Ms += ms(t1, clock::now());
                if (!done) ++st.pathRedo;
            }
        } else {
            done = flood_landmass_pipeline(e, carveStrength, S, rep, st.pathRedo);
            st.pass1Ms += ms(t0, clock::now());             // pass 1 and passes 2/3 overlap: one figure
            if (rep.replayed) { ++st.replays; st.replayedLandmasses += rep.replayed; }
        }
        st.tieGroups += rep.groups; st.contested += rep.contested; st.openParents += rep.openParents; st.unresolved += rep.unresolved;
        if (!done) flood_gather(e, S);
    }
    if (!done) {
        auto t0 = clock::now();
        flood_pass1_host(S);
        auto t1 = clock::now();
        flood_pass23_host(e, carveStrength, S);
        st.pass1Ms += ms(t0, t1); st.pass23Ms += ms(t1, clock::now());
        ++st.serialPass1;
    }
}

int flood_host_passes_exchange(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, float* e, double carveStrength,
                               FloodScratch& S, FloodHostStats* stats, FloodExchange& X) {
    using clock = std::chrono::steady_clock;
    auto ms = [](clock::time_point a, clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    FloodHostStats local;
    FloodHostStats& st = stats ? *stats : local;
    ++st.calls; ++X.calls;
    // the heights at this call, before anything is carved: what this share contributes to the pool
    X.snapshot.resize((size_t)N);
    float* snap = X.snapshot.data();
    parallel_ranges(N, [&](int64_t b, int64_t en, int) { std::memcpy(snap + b, e + b, sizeof(float) * (size_t)(en - b)); });
    int32_t flag = 0;
    if (S.L > 0) {
        flood_gather(e, S);
        FloodTieReport rep;
        auto t0 = clock::now();
        const bool done = flood_landmass_pipeline(e, carveStrength, S, rep, st.pathRedo, false);
        st.pass1Ms += ms(t0, clock::now());
        st.tieGroups += rep.groups; st.contested += rep.contested; st.openParents += rep.openParents; st.unresolved += rep.unresolved;
        flag = done ? 0 : 1;
    }
    // Who floods the whole planet when a call is undecided: ONE rank — the undecided one that owns the land cell with the smallest id (the
    // ranks' cells are disjoint, so "INT32_MAX - that id" under the max picks it and every rank can tell whether it is the one).  It hands
    // the planet's land heights back (phase 2: it sends; phase 3: the others receive) and the other undecided ranks keep their own cells.
    const bool mine = flag != 0;
    if (S.L > 0 && (X.posVersion != S.staticVersion || X.ownPos.size() != (size_t)S.L)) {
        X.ownPos.resize((size_t)S.L);
        const uint8_t* toc = X.trueOcean.data();
        const int32_t* li = S.landIndex.data();
        int32_t cnt = 0, lowest = -1, placed = 0;
        for (int32_t r = 0; r < N; ++r) if (!toc[r]) { if (li[r] >= 0) { X.ownPos[(size_t)li[r]] = cnt; ++placed; if (lowest < 0) lowest = r; } ++cnt; }
        // every land cell of the resident (this rank's) mask must be land in the planet's true mask: a cell that is not has no place among the planet's
        // land heights (its ownPos entry would stay unset) and the election below has no bid for a rank without one
        if (placed != S.L) throw std::runtime_error("flood exchange: the resident ocean mask has land cells that the planet's true mask (wo_planet_set_flood_exchange) calls ocean");
        X.landTotal = cnt; X.minOwnCell = lowest; X.posVersion = S.staticVersion;
    }
    const int32_t bid = mine ? INT32_MAX - X.minOwnCell : 0;
    flag = bid;
    if (int rc = X.fn(X.user, 0, &flag, 1)) return rc;
    if (!flag) return 0;
    ++X.gathers;
    if (int rc = X.fn(X.user, 1, snap, N)) return rc;
    if (X.landTotal < 0) { int64_t cnt = 0; for (int32_t r = 0; r < N; ++r) cnt += X.trueOcean[r] ? 0 : 1; X.landTotal = cnt; }      // (a rank without land)
    X.landPack.resize((size_t)std::max<int64_t>(X.landTotal, 1));
    float* pack = X.landPack.data();
    if (mine && flag == bid) {
        // undecided, and the one to do it: the whole planet, as the unpartitioned run floods it
        ++X.globalFloods;
        FloodScratch& G = X.global;
        if (!G.staticValid || G.staticN != N) flood_build_static(N, off, adj, xyz, X.trueOcean.data(), G);
        flood_gather(snap, G);
        FloodTieReport rep;
        auto t0 = clock::now();
        flood_landmass_pipeline(snap, carveStrength, G, rep, st.pathRedo, true);
        st.pass1Ms += ms(t0, clock::now());
        if (rep.replayed) { ++st.replays; st.replayedLandmasses += rep.replayed; }
        const int32_t* gCell = G.landCell.data();
        const int32_t* byR = G.landByR.data();
        parallel_ranges(G.L, [&](int64_t b, int64_t en, int) { for (int64_t q = b; q < en; ++q) pack[q] = snap[gCell[byR[q]]]; });
        if (int rc = X.fn(X.user, 2, pack, X.landTotal)) return rc;
    } else {
        if (int rc = X.fn(X.user, 3, pack, X.landTotal)) return rc;
        if (mine) ++X.received;
    }
    if (mine) {
        const int32_t* landCell = S.landCell.data();
        const int32_t* pos = X.ownPos.data();
        parallel_ranges(S.L, [&](int64_t b, int64_t en, int) { for (int64_t i = b; i < en; ++i) e[landCell[i]] = pack[pos[i]]; });
    }
    return 0;
}

void priority_flood_carve_host(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, float* e,
                               const uint8_t* ocean, double carveStrength, FloodScratch& S) {
    if (!S.staticValid || S.staticN != N) flood_build_static(N, off, adj, xyz, ocean, S);
    if (S.L == 0) return;
    flood_host_passes(e, carveStrength, S, nullptr);
}

}  // namespace wo
