// Test infrastructure (never linked into the product): the attribute-carrying BFS fields of assignElevation as the reference computes them — plain FIFO
// queues in the reference's own order (js/elevation.js:464-631, 1059-1086).  The product computes these fields on the device (csrc/elevation_bfs.h:
// level-synchronous BFS with FIFO-order reconstruction); the emulator of those kernel bodies (emu_bfs_fields) is checked against this restatement, and the
// emulated elevation pass can run on either (tests/test_elevation_emulated.py).
#pragma once
#include <cmath>
#include <functional>
#include <thread>
#include <vector>

#include "../../planet_heightmap_generation_amd/csrc/elevation_host.h"
#include "../../planet_heightmap_generation_amd/csrc/host_util.h"

namespace wo {

// FIFO walks know their next entries exactly: pull the rows / distances of the entries a few steps ahead
inline void fifo_prefetch(const ElevMesh& M, const std::vector<int32_t>& queue, size_t qi, const float* dist) {
    const size_t n = queue.size();
    if (qi + 12 < n) { const int32_t a = queue[qi + 12]; __builtin_prefetch(&M.off[a]); __builtin_prefetch(&dist[a]); }
    if (qi + 6 < n) { const int32_t a = queue[qi + 6]; __builtin_prefetch(&M.adj[M.off[a]]); }
    if (qi + 3 < n) { const int32_t a = queue[qi + 3]; for (int32_t ni = M.off[a]; ni < M.off[a + 1]; ++ni) __builtin_prefetch(&dist[M.adj[ni]]); }
}

// bounded FIFO BFS used by rift / ridge / fracture / back-arc / island-arc fields
// (js/elevation.js:511-631, 1059-1086): `pass(nr, r)` is the extra admission rule; carry copies an attribute
inline void bounded_bfs(const ElevMesh& M, std::vector<int32_t>& queue, float* dist, double maxDist,
                 const std::function<bool(int32_t, int32_t)>& pass, float* carry) {
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        fifo_prefetch(M, queue, qi, dist);
        const int32_t r = queue[qi];
        const double nd = (double)dist[r] + 1;
        if (nd > maxDist) continue;
        for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni) {
            const int32_t nr = M.adj[ni];
            if (nd < (double)dist[nr] && pass(nr, r)) {
                dist[nr] = (float)nd;
                if (carry) carry[nr] = carry[r];
                queue.push_back(nr);
            }
        }
    }
}


inline void fifo_bfs_fields(const ElevMesh& M, const ElevInputs& I, ElevHostState& H, const ElevParams& Q, int32_t maxCD, double maxStress) {
    const int32_t N = M.N;
    // ---- BFS fields (independent of each other) ----
    H.dBdry.assign(N, (float)(maxCD + 1)); H.coastStressMax.assign(N, 0.f); H.coastSubductMax.assign(N, 0.f); H.coastConvergent.assign(N, 0);
    H.riftDist.assign(N, INFINITY); H.ridgeDist.assign(N, INFINITY); H.fractureDist.assign(N, INFINITY);
    H.backArcDist.assign(N, INFINITY); H.backArcStress.assign(N, 0.f);
    H.arcDist.assign(N, (float)(Q.maxArcDist + 1)); H.arcStress.assign(N, 0.f);
    auto coast_bfs = [&]() {                                   // :464-509
        std::vector<int32_t> q;
        {   // boundary cells in ascending r: scanned in parallel, concatenated in range order
            std::vector<std::vector<int32_t>> part(host_threads() + 1);
            parallel_ranges(N, [&](int64_t b, int64_t e, int t) {
                for (int64_t r = b; r < e; ++r) {
                    const uint8_t rOc = H.isOcean[r];
                    for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni) if (H.isOcean[M.adj[ni]] != rOc) { part[t].push_back((int32_t)r); break; }
                }
            });
            q.reserve((size_t)N + 16);
            for (auto& v : part) q.insert(q.end(), v.begin(), v.end());
        }
        for (int32_t r : q) {
            H.dBdry[r] = 0;
            H.coastStressMax[r] = (float)std::min(1.0, (double)H.stress[r] / maxStress);
            H.coastSubductMax[r] = H.subduct[r];
            H.coastConvergent[r] = H.btype[r] == 1 ? 1 : 0;
        }
        for (size_t qi = 0; qi < q.size(); ++qi) {
            fifo_prefetch(M, q, qi, H.dBdry.data());
            const int32_t r = q[qi];
            const double nd = (double)H.dBdry[r] + 1;
            if (nd > maxCD) continue;
            for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni) {
                const int32_t nr = M.adj[ni];
                if (nd < (double)H.dBdry[nr]) {
                    H.dBdry[nr] = (float)nd;
                    H.coastStressMax[nr] = H.coastStressMax[r]; H.coastSubductMax[nr] = H.coastSubductMax[r]; H.coastConvergent[nr] = H.coastConvergent[r];
                    q.push_back(nr);
                } else if (nd == (double)H.dBdry[nr] && H.coastStressMax[r] > H.coastStressMax[nr]) {
                    H.coastStressMax[nr] = H.coastStressMax[r]; H.coastSubductMax[nr] = H.coastSubductMax[r]; H.coastConvergent[nr] = H.coastConvergent[r];
                }
            }
        }
    };
    auto rift_bfs = [&]() {                                    // :511-538
        std::vector<int32_t> q;
        for (int32_t r = 0; r < N; ++r) if (H.btype[r] == 2 && !H.hasOcean[r]) { q.push_back(r); H.riftDist[r] = 0; }
        bounded_bfs(M, q, H.riftDist.data(), Q.riftHalfWidth, [&](int32_t nr, int32_t r) { return I.plate[nr] == I.plate[r] && !H.isOcean[nr]; }, nullptr);
    };
    auto ridge_bfs = [&]() {                                   // :542-568
        std::vector<int32_t> q;
        for (int32_t r = 0; r < N; ++r) if (H.btype[r] == 2 && H.bothOcean[r]) { q.push_back(r); H.ridgeDist[r] = 0; }
        bounded_bfs(M, q, H.ridgeDist.data(), Q.ridgeHalfWidth, [&](int32_t nr, int32_t) { return H.isOcean[nr] != 0; }, nullptr);
    };
    auto fracture_bfs = [&]() {                                // :570-596
        std::vector<int32_t> q;
        for (int32_t r = 0; r < N; ++r) if (H.btype[r] == 3 && H.bothOcean[r]) { q.push_back(r); H.fractureDist[r] = 0; }
        bounded_bfs(M, q, H.fractureDist.data(), Q.fractureHalfWidth, [&](int32_t nr, int32_t) { return H.isOcean[nr] != 0; }, nullptr);
    };
    auto backarc_bfs = [&]() {                                 // :598-631
        std::vector<int32_t> q;
        for (int32_t r = 0; r < N; ++r)
            if (H.btype[r] == 1 && H.hasOcean[r] && (double)H.subduct[r] < 0.50) {
                q.push_back(r); H.backArcDist[r] = 0; H.backArcStress[r] = (float)std::min(1.0, (double)H.stress[r] / maxStress);
            }
        bounded_bfs(M, q, H.backArcDist.data(), Q.baEnd, [&](int32_t nr, int32_t r) { return I.plate[nr] == I.plate[r]; }, H.backArcStress.data());
    };
    auto arc_bfs = [&]() {                                     // :1059-1086
        std::vector<int32_t> q;
        for (int32_t r = 0; r < N; ++r)
            if (H.btype[r] == 1 && H.bothOcean[r] && (double)H.subduct[r] < 0.45) {
                q.push_back(r); H.arcDist[r] = 0; H.arcStress[r] = (float)std::min(1.0, (double)H.stress[r] / maxStress);
            }
        bounded_bfs(M, q, H.arcDist.data(), Q.maxArcDist, [&](int32_t nr, int32_t r) { return I.plate[nr] == I.plate[r] && H.isOcean[nr] != 0; }, H.arcStress.data());
    };
    {
        std::thread a(coast_bfs), b(rift_bfs), c(ridge_bfs), d(fracture_bfs), e(backarc_bfs);
        arc_bfs();
        a.join(); b.join(); c.join(); d.join(); e.join();
    }
}

}  // namespace wo
