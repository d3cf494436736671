// smoothAndReconnectPlates — native host stage (reference: js/plates.js:241-348).
//
// The majority-vote passes rewrite r_plate in place in ascending cell order (a cell's vote reads neighbours that
// were already rewritten in the same pass), the orphan sweep marks cells as it goes and the final fill is a FIFO
// walk: all three are order-defined, so they stay serial here.  Only the component search is order-free (which
// cells form a component does not depend on the visiting order; the reference keeps, per plate, the largest
// component and among equals the one found first = the one holding the smallest cell id), so it runs as a
// concurrent union-find.  Plate ids are integers: results are identical to the reference's, not approximately so.
#include <algorithm>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_util.h"
#include "wo_internal.h"

namespace wo {

void smooth_reconnect_plates_host(int32_t N, const int32_t* off, const int32_t* adj, int32_t* r_plate, int32_t numSeeds,
                                  const int32_t* plateSeeds, int32_t numPasses) {
    // :252-255 seed protection (a seed id protects cell `id` only when that cell still carries plate `id`)
    std::vector<uint8_t> isSeed(N, 0);
    for (int32_t i = 0; i < numSeeds; ++i) {
        const int32_t pid = plateSeeds[i];
        if (pid >= 0 && pid < N && r_plate[pid] == pid) isSeed[pid] = 1;
    }
    // :265-287 majority vote, in place, ascending r.  Distinct plates are collected in neighbour order; the first
    // plate reaching the highest count wins (strict > while scanning).  A cell whose neighbours all carry its own plate
    // keeps it (the vote returns that plate), so only cells with a differing neighbour need the vote: the boundary
    // cells at the start of the pass (parallel scan, ascending) plus, as the sweep moves on, higher-numbered neighbours
    // of cells it has just changed (a small min-heap merged into the sweep).  Same visiting order for every cell that
    // can change, a few per cent of the cells visited.
    auto vote = [&](int32_t r, double threshold) -> bool {
        int32_t cand[64]; int32_t cnt[64];
        const int32_t b = off[r], e = off[r + 1], deg = e - b;
        int32_t nd = 0;
        bool overflow = false;
        for (int32_t j = b; j < e; ++j) {
            const int32_t p = r_plate[adj[j]];
            int32_t k = 0;
            while (k < nd && cand[k] != p) ++k;
            if (k < nd) { ++cnt[k]; continue; }
            if (nd == 64) { overflow = true; break; }
            cand[nd] = p; cnt[nd] = 1; ++nd;
        }
        int32_t bestPlate = r_plate[r], bestCount = 0;
        if (!overflow) {
            for (int32_t k = 0; k < nd; ++k) if (cnt[k] > bestCount) { bestCount = cnt[k]; bestPlate = cand[k]; }
        } else {                                    // more than 64 distinct neighbour plates: general path
            std::vector<std::pair<int32_t, int32_t>> v;
            for (int32_t j = b; j < e; ++j) {
                const int32_t p = r_plate[adj[j]];
                size_t k = 0;
                while (k < v.size() && v[k].first != p) ++k;
                if (k < v.size()) ++v[k].second; else v.push_back({p, 1});
            }
            for (auto& pr : v) if (pr.second > bestCount) { bestCount = pr.second; bestPlate = pr.first; }
        }
        if ((double)bestCount > deg * threshold && !isSeed[r] && r_plate[r] != bestPlate) { r_plate[r] = bestPlate; return true; }
        return false;
    };
    std::vector<uint8_t> queued(N, 0);
    for (int32_t pass = 0; pass < numPasses; ++pass) {
        const double threshold = pass == 0 ? 0.4 : 0.5;
        std::vector<std::vector<int32_t>> part(host_threads() + 1);
        parallel_ranges(N, [&](int64_t b, int64_t e, int t) {
            for (int64_t r = b; r < e; ++r) {
                const int32_t own = r_plate[r];
                for (int32_t j = off[r]; j < off[r + 1]; ++j) if (r_plate[adj[j]] != own) { part[t].push_back((int32_t)r); break; }
            }
        });
        std::vector<int32_t> sweep;
        for (auto& v : part) sweep.insert(sweep.end(), v.begin(), v.end());           // ascending: ranges ascend with t
        for (int32_t r : sweep) queued[r] = 1;
        std::vector<int32_t> heap;                                                     // cells that became boundary during this pass
        auto cmp = [](int32_t a, int32_t b) { return a > b; };
        size_t si = 0;
        for (;;) {
            int32_t r;
            if (!heap.empty() && (si >= sweep.size() || heap.front() < sweep[si])) { std::pop_heap(heap.begin(), heap.end(), cmp); r = heap.back(); heap.pop_back(); }
            else if (si < sweep.size()) r = sweep[si++];
            else break;
            queued[r] = 0;
            if (vote(r, threshold)) {
                for (int32_t j = off[r]; j < off[r + 1]; ++j) {
                    const int32_t n = adj[j];
                    if (n > r && !queued[n]) { queued[n] = 1; heap.push_back(n); std::push_heap(heap.begin(), heap.end(), cmp); }
                }
            }
        }
    }
    // :292-320 components of equal plate id; per plate the largest, the first found winning ties
    std::vector<int32_t> root(N);
    mesh_components(N, off, adj, [](int32_t) { return true; }, [&](int32_t r, int32_t nb) { return r_plate[r] == r_plate[nb]; }, root.data());
    std::vector<int32_t> size(N, 0);
    for (int32_t r = 0; r < N; ++r) ++size[root[r]];
    std::unordered_map<int32_t, int32_t> mainOf;        // plate id -> root of its kept component
    for (int32_t r = 0; r < N; ++r) {                   // roots in ascending id = discovery order of the reference's scan
        if (root[r] != r) continue;
        auto it = mainOf.find(r_plate[r]);
        if (it == mainOf.end()) mainOf.emplace(r_plate[r], r);
        else if (size[r] > size[it->second]) it->second = r;
    }
    std::vector<uint8_t> inMain(N, 0);
    std::vector<int32_t> orphans;
    {
        int32_t lastPlate = 0, lastMain = -1; bool have = false;
        for (int32_t r = 0; r < N; ++r) {
            if (!have || r_plate[r] != lastPlate) { lastPlate = r_plate[r]; lastMain = mainOf[lastPlate]; have = true; }
            if (root[r] == lastMain) inMain[r] = 1; else orphans.push_back(r);
        }
    }
    // :324-335 orphans touching the kept part take the plate of their first kept neighbour (marks apply at once)
    std::vector<int32_t> queue;
    for (int32_t r : orphans) {
        for (int32_t j = off[r]; j < off[r + 1]; ++j)
            if (inMain[adj[j]]) { r_plate[r] = r_plate[adj[j]]; inMain[r] = 1; queue.push_back(r); break; }
    }
    // :336-346 FIFO fill of what is left
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const int32_t r = queue[qi];
        for (int32_t j = off[r]; j < off[r + 1]; ++j) {
            const int32_t nb = adj[j];
            if (!inMain[nb]) { r_plate[nb] = r_plate[r]; inMain[nb] = 1; queue.push_back(nb); }
        }
    }
}

}  // namespace wo
