// Serial, order-defined parts of assignElevation (reference: js/elevation.js) — native host code.
//
// These loops are defined by traversal order (Set insertion order, frontier order with in-place updates,
// an LCG-driven random pick per step, FIFO queues carrying "first / strongest parent" attributes), so they
// are restated serially; independent fields are computed on separate host threads.  The per-cell work
// (collisions, the uplift loop, coastal roughening, arcs, hotspots) is in elevation_ops.h / the HIP kernels.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "elevation_host.h"
#include "host_util.h"

namespace wo {

namespace {

// ordered set of cell ids (JS Set: insertion order, no duplicates)
struct OrderedSet {
    std::vector<int32_t> items;
    std::vector<uint8_t> in;
    explicit OrderedSet(int32_t N) : in(N, 0) {}
    bool has(int32_t r) const { return in[r] != 0; }
    void add(int32_t r) { if (!in[r]) { in[r] = 1; items.push_back(r); } }
};

// propagateStress (js/elevation.js:127-159).  The relaxation is in place and order-defined inside a pass (a frontier
// entry may have been raised earlier in the same pass; entries may repeat), but it never crosses a plate boundary
// (`r_plate[nb] === plate`): the sub-sequences of different plates touch disjoint cells and commute.  So the start
// frontier is split by plate, stably, and every plate runs all its passes on its own — same values, plates in
// parallel.  Within a plate the reference's order is kept entry for entry.
void propagate_stress(const ElevMesh& M, float* stress, float* subduct, const int32_t* plate, const uint8_t* plateIsOcean,
                      double decayFactor, double subductDecayFactor, int32_t numPasses) {
    // start frontier (:132-135), grouped by plate id in first-appearance order, ascending r inside a group
    std::vector<int32_t> all;
    for (int32_t r = 0; r < M.N; ++r) if ((double)stress[r] > 0.01) all.push_back(r);
    std::vector<int32_t> groupOf;                     // plate id -> group, grown on demand
    std::vector<std::vector<int32_t>> groups;
    for (int32_t r : all) {
        const int32_t pl = plate[r];
        if (plateIsOcean[pl]) continue;               // :141 ocean plates never propagate (and are never pushed by others)
        if ((size_t)pl >= groupOf.size()) groupOf.resize((size_t)pl + 1, -1);
        if (groupOf[pl] < 0) { groupOf[pl] = (int32_t)groups.size(); groups.emplace_back(); }
        groups[groupOf[pl]].push_back(r);
    }
    auto run_plate = [&](std::vector<int32_t>& frontier) {
        std::vector<int32_t> next;
        const int32_t pl = plate[frontier[0]];
        for (int32_t pass = 0; pass < numPasses && !frontier.empty(); ++pass) {
            next.clear();
            const size_t nf = frontier.size();
            for (size_t fi = 0; fi < nf; ++fi) {
                // the frontier of a pass is fixed: pull the rows of the entries a few steps ahead
                if (fi + 12 < nf) { const int32_t a = frontier[fi + 12]; __builtin_prefetch(&M.off[a]); __builtin_prefetch(&stress[a]); __builtin_prefetch(&subduct[a]); }
                if (fi + 6 < nf) { const int32_t a = frontier[fi + 6]; __builtin_prefetch(&M.adj[M.off[a]]); }
                if (fi + 3 < nf) { const int32_t a = frontier[fi + 3]; for (int32_t ni = M.off[a]; ni < M.off[a + 1]; ++ni) { __builtin_prefetch(&stress[M.adj[ni]]); __builtin_prefetch(&plate[M.adj[ni]]); } }
                const int32_t r = frontier[fi];
                const float sfF = subduct[r];
                const double sf = sfF;
                const double effDecay = sf > 0.5 ? subductDecayFactor : decayFactor;
                const double propagated = (double)stress[r] * effDecay;
                if (propagated < 0.005) continue;
                for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni) {
                    const int32_t nb = M.adj[ni];
                    if (plate[nb] == pl && propagated > (double)stress[nb]) {
                        stress[nb] = (float)propagated;
                        subduct[nb] = sfF;
                        next.push_back(nb);
                    }
                }
            }
            frontier.swap(next);
        }
    };
    // largest groups first so the pool drains evenly
    std::vector<int32_t> order(groups.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int32_t)i;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return groups[a].size() > groups[b].size(); });
    std::atomic<size_t> nextGroup{0};
    auto worker = [&]() { for (;;) { const size_t k = nextGroup.fetch_add(1); if (k >= order.size()) break; run_plate(groups[order[k]]); } };
    const int nt = std::max(1, std::min<int>({host_threads(), 16, (int)groups.size()}));
    if (nt == 1) { worker(); return; }
    std::vector<std::thread> th;
    for (int i = 0; i < nt; ++i) th.emplace_back(worker);
    for (auto& t : th) t.join();
}

// assignDistanceField (js/elevation.js:164-189): random-pick frontier growth — step qi draws u, takes the queue entry
// at qi + floor(u * (length - qi)), moves entry qi into its place and expands it.  The order is defined by the RNG,
// so the walk is serial; what can be hidden is its memory latency (every step lands on a random queue slot, a random
// cell and its neighbours: ~5 cold lines at 10^7 cells).  The draws do not depend on the walk, so they are produced
// a few steps ahead and the slots they will *probably* select (the queue length is only known to within a few
// entries) are prefetched in stages: queue line -> cell row and distance -> neighbours' distances.
void distance_field(const ElevMesh& M, const std::vector<int32_t>& seeds, const uint8_t* isStop, double rngSeed, float* dist) {
    ParkMiller rng(rngSeed);
    for (int32_t r = 0; r < M.N; ++r) dist[r] = INFINITY;
    hvec<int32_t> queueStore((size_t)M.N + seeds.size() + 16);   // a seed entry plus at most one entry per cell
    int32_t* queue = queueStore.data();
    size_t len = 0;
    for (int32_t r : seeds) { queue[len++] = r; dist[r] = 0; }    // seeds are distinct cells (Set / filtered lists)
    constexpr int LOOK = 16;                                       // draws kept ahead (ring)
    double u[LOOK];
    for (int k = 0; k < LOOK; ++k) u[k] = rng.next();
    auto guess = [&](size_t step, size_t lenNow, size_t qiNow) -> size_t {        // probable slot of a later step
        const size_t lenThen = lenNow + (step - qiNow);                            // steady frontier: one push per pop
        if (step >= lenThen) return lenNow - 1;
        size_t p = step + (size_t)std::floor(u[step % LOOK] * (double)(lenThen - step));
        return p < lenNow ? p : lenNow - 1;
    };
    for (size_t qi = 0; qi < len; ++qi) {
        {   // staged prefetch for steps qi+12, qi+8, qi+4, qi+2
            __builtin_prefetch(&queue[guess(qi + 12, len, qi)]);
            const int32_t c8 = queue[guess(qi + 8, len, qi)];
            __builtin_prefetch(&M.off[c8]); __builtin_prefetch(&dist[c8]);
            const int32_t c4 = queue[guess(qi + 4, len, qi)];
            __builtin_prefetch(&M.adj[M.off[c4]]);
            const int32_t c2 = queue[guess(qi + 2, len, qi)];
            for (int32_t ni = M.off[c2]; ni < M.off[c2 + 1]; ++ni) { __builtin_prefetch(&dist[M.adj[ni]]); if (isStop) __builtin_prefetch(&isStop[M.adj[ni]]); }
        }
        const size_t pos = qi + (size_t)std::floor(u[qi % LOOK] * (double)(len - qi));
        u[qi % LOOK] = rng.next();                                 // the draw for step qi + LOOK
        const int32_t cur = queue[pos];
        queue[pos] = queue[qi];
        const float dn = (float)((double)dist[cur] + 1);
        for (int32_t ni = M.off[cur]; ni < M.off[cur + 1]; ++ni) {
            const int32_t nb = M.adj[ni];
            if (dist[nb] == INFINITY && !(isStop && isStop[nb])) { dist[nb] = dn; queue[len++] = nb; }
        }
    }
}


}  // namespace

void blend_collision_layers(int32_t N, const CollisionHost& S, const CollisionHost* P, ElevHostState& H) {
    H.stress.resize(N); H.subduct.resize(N); H.btype.resize(N); H.bothOcean.resize(N); H.hasOcean.resize(N);
    if (!P) {
        std::memcpy(H.stress.data(), S.stress.data(), N * 4); std::memcpy(H.subduct.data(), S.subduct.data(), N * 4);
        std::memcpy(H.btype.data(), S.btype.data(), N); std::memcpy(H.bothOcean.data(), S.bothOcean.data(), N);
        std::memcpy(H.hasOcean.data(), S.hasOcean.data(), N);
        return;
    }
    const double SMALL_W = 0.05, SUPER_W = 0.95;
    float maxSuper = 0;
    for (int32_t r = 0; r < N; ++r) if (P->stress[r] > maxSuper) maxSuper = P->stress[r];
    const double invMax = (double)maxSuper > 1e-6 ? 1 / (double)maxSuper : 0;
    for (int32_t r = 0; r < N; ++r) {
        const double sS = S.stress[r], sP = P->stress[r];
        const double proximity = std::min(1.0, sP * invMax * 3);
        const double effectiveSmallW = SMALL_W * (SMALL_W + (1 - SMALL_W) * proximity);
        H.stress[r] = (float)(effectiveSmallW * sS + SUPER_W * sP);
        const double wS = SMALL_W * sS, wP = SUPER_W * sP, total = wS + wP;
        if (total > 1e-6) H.subduct[r] = (float)((wS * (double)S.subduct[r] + wP * (double)P->subduct[r]) / total);
        else H.subduct[r] = (float)(SMALL_W * (double)S.subduct[r] + SUPER_W * (double)P->subduct[r]);
        H.btype[r] = wS > wP ? S.btype[r] : P->btype[r];
        H.bothOcean[r] = S.bothOcean[r] | P->bothOcean[r];
        H.hasOcean[r] = S.hasOcean[r] | P->hasOcean[r];
    }
}

// Everything between the collision kernels and the uplift kernel (js/elevation.js:249-631, 1059-1086, 1116-1261).
void elevation_host_stage(const ElevMesh& M, const ElevInputs& I, const CollisionHost& S, const CollisionHost* P,
                          ElevHostState& H, ElevParams& Q, std::vector<Dome>& domes,
                          const std::function<void(const ElevParams&, int32_t, double)>& bfsOnDevice) {
    const int32_t N = M.N;
    const bool hasSuper = P != nullptr;
    const double SMALL_W = 0.05, SUPER_W = 0.95;
    const bool timing = std::getenv("WO_ELEV_TIMING") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[elevation host] %-16s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tp).count());
        tp = now;
    };

    // ---- sets (:257-271) ----
    OrderedSet mountain(N), coastline(N), ocean(N);
    auto add_coded = [&](const CollisionHost& C, int code, OrderedSet& dst) { for (int32_t r = 0; r < N; ++r) if (C.setCode[r] == code) dst.add(r); };
    if (!hasSuper) {
        add_coded(S, 1, mountain); add_coded(S, 2, coastline); add_coded(S, 3, ocean);
    } else {
        add_coded(*P, 1, mountain); add_coded(S, 1, mountain);
        add_coded(*P, 3, ocean); add_coded(S, 3, ocean);
        for (int32_t r = 0; r < N; ++r) if (P->setCode[r] == 2 && !mountain.has(r)) coastline.add(r);
        for (int32_t r = 0; r < N; ++r) if (S.setCode[r] == 2 && !mountain.has(r) && !coastline.has(r)) coastline.add(r);
    }
    blend_collision_layers(N, S, P, H);

    lap("sets+blend");
    // ---- plate representatives (:368-382) ----
    {
        std::vector<int32_t> rep(I.plates.numIds, -1);
        for (int32_t r = 0; r < N; ++r) {
            const int32_t pid = I.plate[r];
            if (rep[pid] < 0 && !mountain.has(r) && !coastline.has(r) && !ocean.has(r)) rep[pid] = r;
        }
        for (int32_t i = 0; i < I.numPlateSeeds; ++i) {
            const int32_t pid = I.plateSeeds[i];
            if (pid >= 0 && pid < I.plates.numIds && rep[pid] >= 0) (I.plates.isOcean[pid] ? ocean : coastline).add(rep[pid]);
        }
    }
    // ---- isOcean by plate, coast seeds (:396-425) ----
    H.isOcean.assign(N, 0);
    for (int32_t r = 0; r < N; ++r) H.isOcean[r] = I.plates.isOcean[I.plate[r]] ? 1 : 0;
    OrderedSet coastSeeds(N);
    std::vector<int32_t> landCoastSeeds;
    for (int32_t r = 0; r < N; ++r) {
        if (H.isOcean[r]) continue;
        for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni)
            if (H.isOcean[M.adj[ni]]) { coastSeeds.add(M.adj[ni]); landCoastSeeds.push_back(r); break; }
    }
    // ---- five distance fields (:392-426): serial by definition (every step draws from the LCG), independent of each other ->
    // one host thread each, started once the stress has propagated (the mountain seeds and the coastline stops depend on it, and
    // those two walks are the longest: 1.5-1.6 s at 10 M cells; starting the other three earlier was measured and only slowed the
    // stress propagation and the late walks: 2.22 -> 2.31 s for assignElevation).
    H.distMountain.resize(N); H.distOcean.resize(N); H.distCoastline.resize(N); H.distCoast.resize(N); H.distCoastLand.resize(N);
    std::vector<std::thread> walks; walks.reserve(5);
    struct JoinAll { std::vector<std::thread>& t; ~JoinAll() { for (auto& x : t) if (x.joinable()) x.join(); } } joinWalks{walks};
    auto walk = [&](const char* name, std::function<void()> f) {
        walks.emplace_back([name, f, timing]() {
            const auto t0 = std::chrono::steady_clock::now();
            f();
            if (timing) std::fprintf(stderr, "[elevation host]   walk %-12s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        });
    };

    lap("reps+seeds");
    // ---- stress propagation (:329-362) ----
    const double scaleFactor = std::sqrt((double)N / 10000);
    const double baseDecay = 0.5 + I.spread * 0.04;
    const double decayFactor = std::pow(baseDecay, 1 / scaleFactor);
    const double subductDecayFactor = std::pow(baseDecay * 0.45, 1 / scaleFactor);
    const int32_t numPasses = (int32_t)std::max(1.0, std::floor(I.spread * 3 * scaleFactor + 0.5));
    if (!hasSuper) {
        propagate_stress(M, H.stress.data(), H.subduct.data(), I.plate, I.plates.isOcean, decayFactor, subductDecayFactor, numPasses);
    } else {
        std::vector<float> sStress(S.stress), sSub(S.subduct), pStress(P->stress), pSub(P->subduct);
        std::thread t1([&]() { propagate_stress(M, sStress.data(), sSub.data(), I.plate, I.plates.isOcean, decayFactor, subductDecayFactor, numPasses); });
        propagate_stress(M, pStress.data(), pSub.data(), I.superPlate, I.superPlates.isOcean, decayFactor, subductDecayFactor, numPasses);
        t1.join();
        for (int32_t r = 0; r < N; ++r) {
            H.stress[r] = (float)(SMALL_W * (double)sStress[r] + SUPER_W * (double)pStress[r]);
            const double wS = SMALL_W * (double)sStress[r], wP = SUPER_W * (double)pStress[r], total = wS + wP;
            if (total > 1e-6) H.subduct[r] = (float)((wS * (double)sSub[r] + wP * (double)pSub[r]) / total);
        }
    }

    lap("stress");
    std::vector<int32_t> stressMountain;
    for (int32_t r : mountain.items) if ((double)H.subduct[r] < 0.55) stressMountain.push_back(r);
    H.mountain = mountain.items; H.coastline = coastline.items; H.ocean = ocean.items;

    std::vector<uint8_t> stopAll(N, 0);
    for (int32_t r : stressMountain) stopAll[r] = 1;
    for (int32_t r : coastline.items) stopAll[r] = 1;
    for (int32_t r : ocean.items) stopAll[r] = 1;

    walk("coast", [&]() { distance_field(M, coastSeeds.items, nullptr, I.seed + 4, H.distCoast.data()); });
    walk("ocean", [&]() { distance_field(M, ocean.items, coastline.in.data(), I.seed + 2, H.distOcean.data()); });
    walk("land coast", [&]() { distance_field(M, landCoastSeeds, H.isOcean.data(), I.seed + 5, H.distCoastLand.data()); });
    walk("mountain", [&]() { distance_field(M, stressMountain, ocean.in.data(), I.seed + 1, H.distMountain.data()); });
    walk("coastline", [&]() { distance_field(M, coastline.items, stopAll.data(), I.seed + 3, H.distCoastline.data()); });

    // ---- scalars (:431-460) ----
    auto rnd = [](double x) { return std::floor(x + 0.5); };
    Q.N = N; Q.scaleFactor = scaleFactor; Q.noiseMag = I.noiseMag;
    Q.interiorBand = (int32_t)std::max(4.0, rnd(16 * scaleFactor));
    Q.tectonicReach = (int32_t)std::max(6.0, rnd(20 * scaleFactor));
    {
        double maxStress = 0;
        std::vector<float> vals;
        for (int32_t r = 0; r < N; ++r) {
            if ((double)H.stress[r] > 0.01) vals.push_back(H.stress[r]);
            if ((double)H.stress[r] > maxStress) maxStress = H.stress[r];
        }
        if (!vals.empty()) {
            const size_t idx = std::min(vals.size() - 1, (size_t)std::floor((double)vals.size() * 0.97));
            std::nth_element(vals.begin(), vals.begin() + (std::ptrdiff_t)idx, vals.end());     // the value a full sort leaves at idx
            maxStress = vals[idx];
        }
        if (maxStress < 0.01) maxStress = 1;
        Q.maxStress = maxStress;
    }
    Q.warpOctaves = N > 200000 ? 2 : 3;
    Q.plateauStart = (int32_t)std::max(2.0, rnd(3 * scaleFactor));
    const int32_t maxCD = (int32_t)std::max(8.0, rnd(8 * scaleFactor));
    Q.riftHalfWidth = (int32_t)std::max(2.0, rnd(4 * scaleFactor));
    Q.ridgeHalfWidth = (int32_t)std::max(2.0, rnd(4 * scaleFactor));
    Q.fractureHalfWidth = (int32_t)std::max(2.0, rnd(3 * scaleFactor));
    Q.baStart = (int32_t)std::max(1.0, rnd(2 * scaleFactor));
    Q.baPeak = (int32_t)std::max(2.0, rnd(3 * scaleFactor));
    Q.baEnd = (int32_t)std::max(3.0, rnd(5 * scaleFactor));
    Q.coastRoughenDist = (int32_t)std::max(8.0, rnd(8 * scaleFactor));
    Q.islandDist = (int32_t)std::max(4.0, rnd(4 * scaleFactor));
    Q.maxArcDist = (int32_t)std::max(5.0, rnd(5 * scaleFactor));
    const double maxStress = Q.maxStress;

    lap("scalars+pctl");
    // the FIFO BFS fields do not depend on the distance fields: the product starts them on the device now and they run
    // while the serial RNG-ordered walks occupy the host threads
    bfsOnDevice(Q, maxCD, maxStress);
    for (auto& x : walks) x.join();

    lap("distance fields");
    // (the attribute-carrying BFS fields — coast boundary, rift, ridge, fracture, back-arc, island arc: js/elevation.js:464-631, 1059-1086 — are the device's:
    // csrc/elevation_bfs.h; a host form of them was a cross-check route until round 6)

    lap("bfs fields");
    // ---- hotspot dome list (:1116-1261) ----
    domes.clear();
    {
        const double DOME_SIGMA = 0.006, DOME_STRENGTH = 0.60, CHAIN_DECAY = 0.75, CHAIN_SPACING = 0.06;
        ParkMiller hsRng(I.seed + 999), hsRandInt(I.seed + 1001);
        auto frame = [](Dome& dm, double px, double py, double pz, double dx, double dy, double dz) {
            const double dd = dx * px + dy * py + dz * pz;
            double ux = dx - dd * px, uy = dy - dd * py, uz = dz - dd * pz;
            double uLen = std::sqrt(ux * ux + uy * uy + uz * uz);
            if (uLen == 0 || uLen != uLen) uLen = 1;
            ux /= uLen; uy /= uLen; uz /= uLen;
            dm.ux = ux; dm.uy = uy; dm.uz = uz;
            dm.vx = py * uz - pz * uy; dm.vy = pz * ux - px * uz; dm.vz = px * uy - py * ux;
        };
        for (int h = 0; h < 5; ++h) {
            const double hStrength = DOME_STRENGTH * (0.4 + hsRng.next() * 1.2);
            const double hSigma = DOME_SIGMA * (0.4 + hsRng.next() * 1.2);
            const double hDecay = CHAIN_DECAY + (hsRng.next() - 0.5) * 0.35;
            const int32_t hLength = (int32_t)std::max(3.0, 6 + rnd((hsRng.next() - 0.5) * 10));
            const int32_t centerR = (int32_t)std::floor(hsRandInt.next() * (double)N);
            const double hx = M.xyz[3 * centerR], hy = M.xyz[3 * centerR + 1], hz = M.xyz[3 * centerR + 2];
            const int32_t plate = I.plate[centerR];
            if (!(plate >= 0 && plate < I.plates.numIds && I.plates.hasVec[plate])) continue;
            const double* pole = I.plates.pole + 3 * plate; const double om = I.plates.omega[plate];
            double drift[3] = {om * (pole[1] * hz - pole[2] * hy), om * (pole[2] * hx - pole[0] * hz), om * (pole[0] * hy - pole[1] * hx)};
            const double driftLen = std::sqrt(drift[0] * drift[0] + drift[1] * drift[1] + drift[2] * drift[2]);
            if (driftLen < 1e-6) continue;
            drift[0] /= driftLen; drift[1] /= driftLen; drift[2] /= driftLen;
            const double oceanBoost = I.plates.isOcean[plate] ? 1.8 : 1.0;
            const double baseRiftAngle = noise3d(I.hsNoise3.P, I.hsNoise3.M, hx * 10, hy * 10, hz * 10) * EL_PI;
            auto rifts = [&](Dome& dm, int32_t ci, int32_t cl) {
                dm.numRifts = 0;
                if (ci == 0) { dm.riftAngles[0] = baseRiftAngle; dm.riftAngles[1] = baseRiftAngle + EL_PI * 0.6; dm.riftAngles[2] = baseRiftAngle - EL_PI * 0.6; dm.numRifts = 3; }
                else if (ci == 1) { dm.riftAngles[0] = baseRiftAngle; dm.riftAngles[1] = baseRiftAngle + EL_PI; dm.numRifts = 2; }
                else if (ci <= (int32_t)std::floor(cl * 0.4)) { dm.riftAngles[0] = baseRiftAngle; dm.numRifts = 1; }
            };
            Dome d0{};
            d0.x = hx; d0.y = hy; d0.z = hz; d0.strength = hStrength * oceanBoost; d0.baseStrength = hStrength; d0.sigma = hSigma;
            d0.chainIndex = 0; d0.chainLength = hLength;
            frame(d0, hx, hy, hz, drift[0], drift[1], drift[2]);
            rifts(d0, 0, hLength);
            domes.push_back(d0);
            double perpX = drift[1] * hz - drift[2] * hy, perpY = drift[2] * hx - drift[0] * hz, perpZ = drift[0] * hy - drift[1] * hx;
            double perpLen = std::sqrt(perpX * perpX + perpY * perpY + perpZ * perpZ);
            if (perpLen == 0 || perpLen != perpLen) perpLen = 1;
            perpX /= perpLen; perpY /= perpLen; perpZ /= perpLen;
            double cx = hx, cy = hy, cz = hz, str = hStrength * oceanBoost, baseStr = hStrength;
            for (int32_t c = 0; c < hLength; ++c) {
                const int32_t ci = c + 1;
                const double decayJitter = hDecay * (0.7 + hsRng.next() * 0.6);
                str *= decayJitter; baseStr *= decayJitter;
                const double stepSpacing = CHAIN_SPACING * (0.3 + hsRng.next() * 1.4);
                const double ageBroadening = 1.0 + ci * 0.06;
                const double stepSigma = hSigma * (0.5 + hsRng.next() * 1.0) * ageBroadening;
                const double wobble = (hsRng.next() - 0.5) * 0.8;
                const double ddx = -drift[0] + perpX * wobble, ddy = -drift[1] + perpY * wobble, ddz = -drift[2] + perpZ * wobble;
                const double dot = ddx * cx + ddy * cy + ddz * cz;
                double tx = ddx - dot * cx, ty = ddy - dot * cy, tz = ddz - dot * cz;
                const double tLen = std::sqrt(tx * tx + ty * ty + tz * tz);
                if (tLen < 1e-6) break;
                tx /= tLen; ty /= tLen; tz /= tLen;
                const double cosA = std::cos(stepSpacing), sinA = std::sin(stepSpacing);
                cx = cx * cosA + tx * sinA; cy = cy * cosA + ty * sinA; cz = cz * cosA + tz * sinA;
                const double nL = std::sqrt(cx * cx + cy * cy + cz * cz);
                cx /= nL; cy /= nL; cz /= nL;
                Dome dc{};
                dc.x = cx; dc.y = cy; dc.z = cz; dc.strength = str; dc.baseStrength = baseStr; dc.sigma = stepSigma;
                dc.chainIndex = ci; dc.chainLength = hLength;
                frame(dc, cx, cy, cz, drift[0], drift[1], drift[2]);
                rifts(dc, ci, hLength);
                domes.push_back(dc);
            }
        }
        for (Dome& dm : domes) {
            dm.cosThreshPeak = std::cos(dm.sigma * 5.5);
            dm.invS2 = -0.5 / (dm.sigma * dm.sigma);
            const double swSigma = dm.sigma * 2;
            dm.swellStrength = dm.baseStrength * 0.10;
            dm.cosThreshSwell = std::cos(swSigma * 3);
            dm.invS2Swell = -0.5 / (swSigma * swSigma);
            dm.driftStretch = 1.0 / 1.4;
            dm.hasCaldera = (dm.chainIndex <= 1 && dm.strength > 0.15) ? 1 : 0;
            const double calderaSigma = dm.sigma * 0.25;
            dm.calderaDepth = dm.strength * 0.20;
            dm.invS2Caldera = -0.5 / (calderaSigma * calderaSigma);
            dm.ageFactor = dm.chainLength > 0 ? (double)dm.chainIndex / dm.chainLength : 0;
        }
    }
    Q.numDomes = (int32_t)domes.size();
}

}  // namespace wo
