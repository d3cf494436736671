// Host-side helpers shared by the native (non-kernel) parts of libworogen.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <thread>
#include <vector>
#include <sys/mman.h>

namespace wo {

inline int host_threads() {
    if (const char* ev = std::getenv("WO_HOST_THREADS")) { int v = std::atoi(ev); if (v >= 1) return v > 256 ? 256 : v; }
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    if (n > 64) n = 64;
    return (int)n;
}

// Allocator for the large host tables that are walked in data-dependent order (the flood's compact land arrays, its
// heap): 2 MB-aligned blocks advised for transparent huge pages before first touch.  With 4 KB pages the ~150 MB the
// flood touches per call is ~40 000 pages against a few thousand TLB entries, and every pop pays page walks.
template <class T>
struct HugeAlloc {
    using value_type = T;
    HugeAlloc() = default;
    template <class U> HugeAlloc(const HugeAlloc<U>&) {}
    T* allocate(size_t n) {
        const size_t bytes = n * sizeof(T);
        if (bytes < (size_t)(4u << 20)) return static_cast<T*>(::operator new(bytes));
        const size_t huge = (size_t)2 << 20, rounded = (bytes + huge - 1) / huge * huge;
        void* p = std::aligned_alloc(huge, rounded);
        if (!p) throw std::bad_alloc();
        madvise(p, rounded, MADV_HUGEPAGE);                 // advisory: failure just means 4 KB pages
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t n) {
        if (n * sizeof(T) < (size_t)(4u << 20)) ::operator delete(p); else std::free(p);
    }
    template <class U> bool operator==(const HugeAlloc<U>&) const { return true; }
    template <class U> bool operator!=(const HugeAlloc<U>&) const { return false; }
};
template <class T> using hvec = std::vector<T, HugeAlloc<T>>;

// Static-chunked parallel loop over [0, n). fn(begin, end, tid).
template <class F>
inline void parallel_ranges(int64_t n, F fn, int64_t min_chunk = 4096) {
    int nt = host_threads();
    if (n < min_chunk * 2 || nt == 1) { fn((int64_t)0, n, 0); return; }
    int64_t chunks = std::min<int64_t>(nt, (n + min_chunk - 1) / min_chunk);
    std::vector<std::thread> th;
    th.reserve(chunks);
    for (int64_t c = 0; c < chunks; ++c) {
        int64_t b = n * c / chunks, e = n * (c + 1) / chunks;
        th.emplace_back([=, &fn]() { fn(b, e, (int)c); });
    }
    for (auto& t : th) t.join();
}

// Connected components of the cells r with member(r), joined along mesh edges (r, nb) with joined(r, nb): a concurrent
// union-find (link the larger root under the smaller with a CAS, path halving on the way up).  On return parent[r] is
// the smallest cell id of r's component (parent[r] == r for non-members).
template <class Member, class Joined>
inline void mesh_components(int32_t N, const int32_t* off, const int32_t* adj, Member member, Joined joined, int32_t* parent) {
    auto par = [&](int32_t x) { return reinterpret_cast<std::atomic<int32_t>*>(parent + x); };
    auto find = [&](int32_t x) {
        for (;;) {
            int32_t px = par(x)->load(std::memory_order_relaxed);
            if (px == x) return x;
            const int32_t gp = par(px)->load(std::memory_order_relaxed);
            if (gp != px) par(x)->compare_exchange_weak(px, gp, std::memory_order_relaxed);    // path halving; losing the race is harmless
            x = gp;
        }
    };
    parallel_ranges(N, [&](int64_t b, int64_t e, int) { for (int64_t r = b; r < e; ++r) parent[r] = (int32_t)r; });
    parallel_ranges(N, [&](int64_t b, int64_t e, int) {
        for (int64_t r = b; r < e; ++r) {
            if (!member((int32_t)r)) continue;
            for (int32_t i = off[r]; i < off[r + 1]; ++i) {
                const int32_t nb = adj[i];
                if (nb > r || !member(nb) || !joined((int32_t)r, nb)) continue;
                int32_t a = (int32_t)r, c = nb;
                for (;;) {
                    a = find(a); c = find(c);
                    if (a == c) break;
                    if (a < c) std::swap(a, c);             // a > c: hang a under c
                    int32_t expect = a;
                    if (par(a)->compare_exchange_strong(expect, c, std::memory_order_relaxed)) break;
                }
            }
        }
    });
    parallel_ranges(N, [&](int64_t b, int64_t e, int) { for (int64_t r = b; r < e; ++r) if (member((int32_t)r)) parent[r] = find((int32_t)r); });
}

// Park-Miller LCG exactly as the reference seeds and steps it (js/rng.js:3-6).
struct ParkMiller {
    uint64_t s;                                           // state is an exact integer < 2^31: integer arithmetic gives the
    explicit ParkMiller(double seed) {                    // same sequence as the reference's double arithmetic, faster
        double v = std::abs(std::floor(seed * 9301.0 + 49297.0));
        s = (uint64_t)(std::fmod(v, 2147483646.0) + 1.0);
    }
    inline double next() {
        s = (s * 16807u) % 2147483647u;
        return (double)(s - 1) / 2147483646.0;
    }
};

}  // namespace wo
