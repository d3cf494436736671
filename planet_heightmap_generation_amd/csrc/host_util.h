// Host-side helpers shared by the native (non-kernel) parts of libworogen.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <new>
#include <pthread.h>
#include <sched.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <thread>
#include <string>
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

// Persistent host workers for parallel_ranges().  A flood call makes half a dozen parallel sweeps (clears, gather, write-back) and the
// composite's setup a few more; creating up to 64 threads for each cost 1-2 ms per sweep on the 256-thread host of the GPU box.
// One job at a time: a caller that finds the pool busy (several planets in flight, each driven by its own host thread), or is itself
// a pool worker, falls back to the plain form (threads of its own / inline).  The pool is never destroyed (its threads sleep on a
// condition variable between jobs) and is rebuilt in a forked child.
class HostPool {
public:
    static HostPool& get() { static HostPool* p = new HostPool(); return *p; }
    // runs job(c) for c in [0, chunks); false: pool not available, nothing was run
    bool run(int64_t chunks, const std::function<void(int64_t)>& job) {
        if (tl_in_worker()) return false;
        std::unique_lock<std::mutex> own(jobLock_, std::try_to_lock);
        if (!own.owns_lock()) return false;
        ensure_threads();
        if (workers_ == 0) return false;
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &job; chunks_ = chunks; next_.store(0); remaining_.store(chunks); error_ = nullptr; ++gen_;
        }
        cv_.notify_all();
        {   // the caller takes chunks too; a job that itself calls parallel_ranges must find the pool unavailable on this thread as
            // well (it owns jobLock_: locking it again from the same thread would be undefined behaviour)
            struct InJob { bool& f; bool was; explicit InJob(bool& x) : f(x), was(x) { f = true; } ~InJob() { f = was; } } guard(tl_in_worker());
            work(job, chunks);
        }
        std::unique_lock<std::mutex> g(m_);
        cvDone_.wait(g, [&] { return remaining_.load() == 0 && active_ == 0; });
        job_ = nullptr;
        if (error_) std::rethrow_exception(error_);
        return true;
    }
private:
    HostPool() { pthread_atfork(nullptr, nullptr, [] { HostPool::get().after_fork(); }); }
    static bool& tl_in_worker() { static thread_local bool v = false; return v; }
    void after_fork() { workers_ = 0; new (&m_) std::mutex(); new (&jobLock_) std::mutex(); new (&cv_) std::condition_variable(); new (&cvDone_) std::condition_variable(); active_ = 0; job_ = nullptr; }
    void ensure_threads() {
        if (workers_ > 0) return;
        const int n = std::max(0, host_threads() - 1);
        for (int i = 0; i < n; ++i) std::thread([this] { tl_in_worker() = true; loop(); }).detach();
        workers_ = n;
    }
    void work(const std::function<void(int64_t)>& job, int64_t chunks) {
        for (;;) {
            const int64_t c = next_.fetch_add(1);
            if (c >= chunks) break;
            try { job(c); } catch (...) { std::lock_guard<std::mutex> g(m_); if (!error_) error_ = std::current_exception(); }
            if (remaining_.fetch_sub(1) == 1) { std::lock_guard<std::mutex> g(m_); cvDone_.notify_all(); }
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int64_t)>* job; int64_t chunks;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                if (!job_ || next_.load() >= chunks_) continue;            // woke late: the job is already handed out
                job = job_; chunks = chunks_; ++active_;
            }
            work(*job, chunks);
            { std::lock_guard<std::mutex> g(m_); --active_; if (active_ == 0) cvDone_.notify_all(); }
        }
    }
    std::mutex m_, jobLock_;
    std::condition_variable cv_, cvDone_;
    const std::function<void(int64_t)>* job_ = nullptr;
    int64_t chunks_ = 0;
    std::atomic<int64_t> next_{0}, remaining_{0};
    int active_ = 0, workers_ = 0;
    uint64_t gen_ = 0;
    std::exception_ptr error_;
};

// The test suite's hooks: ONE environment variable, WO_TEST_HOOKS="key=value,key=value,..." (a key without a value reads "1"), never set in production and
// read where the product reads its options (once per API call / flood call), never inside a pass.  Returns whether the key is there.
inline bool test_hook(const char* key, std::string* value = nullptr) {
    const char* v = std::getenv("WO_TEST_HOOKS");
    if (!v || !*v) return false;
    const std::string s(v), k(key);
    for (size_t pos = 0; pos <= s.size();) {
        size_t e = s.find(',', pos);
        if (e == std::string::npos) e = s.size();
        const std::string item = s.substr(pos, e - pos);
        const size_t eq = item.find('=');
        if (item.substr(0, eq) == k) { if (value) *value = eq == std::string::npos ? std::string("1") : item.substr(eq + 1); return true; }
        pos = e + 1;
    }
    return false;
}
inline long long test_hook_int(const char* key, long long dflt) { std::string v; return test_hook(key, &v) ? std::atoll(v.c_str()) : dflt; }

// a[i] := a[0] + ... + a[i], i < n, on the host's workers (three sweeps: range sums, their serial scan, ranges again; the ranges of two
// parallel_ranges calls over the same n are the same).  Declared here, defined after parallel_ranges.
template <class T> inline void inclusive_scan_parallel(T* a, int64_t n);

// Static-chunked parallel loop over [0, n). fn(begin, end, tid).
template <class F>
inline void parallel_ranges(int64_t n, F fn, int64_t min_chunk = 4096) {
    int nt = host_threads();
    if (n < min_chunk * 2 || nt == 1) { fn((int64_t)0, n, 0); return; }
    int64_t chunks = std::min<int64_t>(nt, (n + min_chunk - 1) / min_chunk);
    constexpr bool pooled = true;
    if (pooled) {
        const std::function<void(int64_t)> job = [&](int64_t c) { fn(n * c / chunks, n * (c + 1) / chunks, (int)c); };
        if (HostPool::get().run(chunks, job)) return;
    }
    std::vector<std::thread> th;
    th.reserve(chunks);
    for (int64_t c = 0; c < chunks; ++c) {
        int64_t b = n * c / chunks, e = n * (c + 1) / chunks;
        th.emplace_back([=, &fn]() { fn(b, e, (int)c); });
    }
    for (auto& t : th) t.join();
}
template <class T> inline void inclusive_scan_parallel(T* a, int64_t n) {
    std::vector<T> sum(host_threads() + 2, T(0));
    parallel_ranges(n, [&](int64_t b, int64_t e, int t) { T s = 0; for (int64_t i = b; i < e; ++i) s += a[i]; sum[t + 1] = s; });
    for (size_t t = 1; t < sum.size(); ++t) sum[t] += sum[t - 1];
    parallel_ranges(n, [&](int64_t b, int64_t e, int t) { T s = sum[t]; for (int64_t i = b; i < e; ++i) { s += a[i]; a[i] = s; } });
}

// ---- where the host threads run (Linux) ----
// The CPUs this process may use, grouped by the last-level cache they share (one CCD of an EPYC: 8 cores, 32 MB).  Read once from sysfs;
// a machine where that fails, or whose mask holds a single group, has no groups and nothing is ever pinned.
struct CpuGroups {
    std::vector<std::vector<int>> groups;          // logical CPUs per L3
    std::vector<int> groupOf;                      // by logical CPU number, -1: not ours
    static const CpuGroups& get() {
        static const CpuGroups* g = [] {
            CpuGroups* G = new CpuGroups();
            cpu_set_t mask;
            CPU_ZERO(&mask);
            if (sched_getaffinity(0, sizeof(mask), &mask) != 0) return G;
            std::vector<int> l3ids;
            G->groupOf.assign(CPU_SETSIZE, -1);
            for (int c = 0; c < CPU_SETSIZE; ++c) {
                if (!CPU_ISSET(c, &mask)) continue;
                char path[128];
                std::snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/id", c);
                int id = -1;
                if (FILE* f = std::fopen(path, "r")) { if (std::fscanf(f, "%d", &id) != 1) id = -1; std::fclose(f); }
                if (id < 0) { G->groups.clear(); std::fill(G->groupOf.begin(), G->groupOf.end(), -1); return G; }
                size_t k = 0;
                while (k < l3ids.size() && l3ids[k] != id) ++k;
                if (k == l3ids.size()) { l3ids.push_back(id); G->groups.emplace_back(); }
                G->groups[k].push_back(c);
                G->groupOf[c] = (int)k;
            }
            if (G->groups.size() < 2) { G->groups.clear(); std::fill(G->groupOf.begin(), G->groupOf.end(), -1); }
            return G;
        }();
        return *g;
    }
};
// The calling thread's affinity for the lifetime of the object (restored on destruction; a failed call changes nothing).
struct AffinityScope {
    cpu_set_t before; bool changed = false;
    AffinityScope() { CPU_ZERO(&before); }
    bool only(int cpu) { cpu_set_t m; CPU_ZERO(&m); CPU_SET(cpu, &m); return set(m); }
    bool all_but_group(const CpuGroups& G, int group) {
        cpu_set_t m; CPU_ZERO(&m);
        int n = 0;
        for (size_t g = 0; g < G.groups.size(); ++g) if ((int)g != group) for (int c : G.groups[g]) { CPU_SET(c, &m); ++n; }
        return n > 0 && set(m);
    }
    ~AffinityScope() { if (changed) (void)pthread_setaffinity_np(pthread_self(), sizeof(before), &before); }
private:
    bool set(const cpu_set_t& m) {
        if (!changed && pthread_getaffinity_np(pthread_self(), sizeof(before), &before) != 0) return false;
        if (pthread_setaffinity_np(pthread_self(), sizeof(m), &m) != 0) return false;
        changed = true;
        return true;
    }
};

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
    parallel_ranges(N, [&](int64_t b, int64_t e, int) { for (int64_t r = b; r < e; ++r) if (member((int32_t)r)) par((int32_t)r)->store(find((int32_t)r), std::memory_order_relaxed); });      // (other threads' find() may pass through r meanwhile: its root is a valid parent at any time)
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
