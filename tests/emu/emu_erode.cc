// TEST-ONLY emulator: drives the kernel bodies of planet_heightmap_generation_amd/csrc/erode_ops.h one
// "thread" at a time on the CPU, with the same round structure the HIP host code uses (a task may only
// consume results of earlier rounds).  It exists because the build container has no GPU: it lets the
// parallel re-formulations be checked bit-for-bit against the oracle before they are run on gfx950.
// It is never linked into libworogen.so and is not reachable from the product API.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "../../planet_heightmap_generation_amd/csrc/erode_ops.h"
#include "../../planet_heightmap_generation_amd/csrc/flood_ops.h"
#include "../../planet_heightmap_generation_amd/csrc/plates_ops.h"
#include "../../planet_heightmap_generation_amd/csrc/wo_internal.h"

using namespace wo;

namespace {

struct Emu {
    Fields F{};
    std::vector<uint8_t> coast, iceUp;
    std::vector<float> e2, cellDist, flow, glac, iceFlow;
    std::vector<SolveTask> task; std::vector<SolveOut> out;
    std::vector<int32_t> land, landIdx, rank, target, jumpA, doneAt, iceTarget, arank, blocker, blk;
    std::vector<TargetRank> tr;
    std::vector<EventList> ev;
    std::vector<float> me;
    std::vector<uint32_t> accA;
    std::vector<double> totalExcess;
    int64_t solveRounds = 0, flowRounds = 0, iceRounds = 0, carveRounds = 0, maxSolveRounds = 0;
};

void stable_sort_desc(Emu& E) {
    const int32_t L = E.F.L;
    std::vector<uint32_t> k0(L), k1(L);
    std::vector<int32_t> v1(L);
    for (int32_t i = 0; i < L; ++i) k0[i] = desc_key(E.F.e[E.land[i]]);
    for (int pass = 0; pass < 4; ++pass) {
        uint32_t cnt[257] = {0};
        const int sh = pass * 8;
        for (int32_t i = 0; i < L; ++i) cnt[((k0[i] >> sh) & 255) + 1]++;
        for (int i = 0; i < 256; ++i) cnt[i + 1] += cnt[i];
        for (int32_t i = 0; i < L; ++i) { uint32_t d = cnt[(k0[i] >> sh) & 255]++; k1[d] = k0[i]; v1[d] = E.land[i]; }
        k0.swap(k1); E.land.swap(v1);
    }
    E.F.land = E.land.data();
    for (int32_t i = 0; i < L; ++i) E.rank[E.land[i]] = i;
}

// generic synchronous rounds over a task list
template <class Task>
int64_t run_rounds(std::vector<int32_t> list, Task task) {
    int64_t round = 0;
    std::vector<int32_t> next;
    while (!list.empty()) {
        ++round;
        next.clear();
        for (int32_t r : list) if (!task(r, (int32_t)round)) next.push_back(r);
        if (next.size() == list.size()) return -round;   // no progress: dependency cycle (bug)
        list.swap(next);
    }
    return round;
}

// solve rounds with level prediction, mirroring run_solve_rounds() in planet.hip: round k examines the
// leftovers plus the tasks whose level in the previous iteration was k
static int LOOKAHEAD = 2;
int64_t run_solve_rounds_pred(Emu& E, std::vector<int32_t>& level, double K, double m, double dt) {
    const Fields& F = E.F;
    int32_t maxPred = 1;
    for (int32_t r : E.landIdx) { if (level[r] < 1) level[r] = 1; if (level[r] > maxPred) maxPred = level[r]; }
    std::vector<std::vector<int32_t>> bucket(maxPred + 2);
    for (int32_t r : E.landIdx) bucket[level[r]].push_back(r);
    std::vector<int32_t> left, next;
    int64_t round = 0;
    for (;;) {
        ++round;
        next.clear();
        for (int32_t r : left) if (!solve_task(F, r, (int32_t)round, K, m, dt)) next.push_back(r);
        // tasks enter LOOKAHEAD rounds before their predicted level (cheap re-examination, fewer cascaded delays)
        const int64_t lo = (round == 1) ? 1 : round + LOOKAHEAD, hi = round + LOOKAHEAD;
        for (int64_t l = lo; l <= hi && l <= maxPred; ++l)
            for (int32_t r : bucket[l]) if (!solve_task(F, r, (int32_t)round, K, m, dt)) next.push_back(r);
        left.swap(next);
        if (round + LOOKAHEAD >= maxPred && left.empty()) break;
        if (round > 8 * (int64_t)F.N) return -round;
    }
    for (int32_t r : E.landIdx) level[r] = F.out[store_index(F, r)].self.tag;
    return round;
}

// patch-local solve schedule, mirroring k_solve_patch / run_solve_patches: records live at the patch slot, external
// predecessors must come from an earlier launch, patch-local ones from any earlier point of the same visit
int64_t run_solve_patches_emu(Emu& E, int32_t L, double K, double m, double dt) {
    const Fields& F = E.F;
    const int32_t np = (L + WO_PATCH - 1) / WO_PATCH;
    const Granule* G = reinterpret_cast<const Granule*>(F.out);
    int64_t launches = 0;
    for (int32_t tag = 1;; ++tag) {
        ++launches;
        int64_t pending = 0;
        for (int32_t p = 0; p < np; ++p) {
            const int32_t s0 = p * WO_PATCH, s1 = std::min(L, s0 + WO_PATCH);
            auto local = [&](int32_t g) { return g >= 2 * s0 && g < 2 * (s0 + WO_PATCH); };
            std::vector<uint8_t> ext(s1 - s0, 0);
            std::vector<double> er(s1 - s0), et(s1 - s0), et2(s1 - s0);
            for (int32_t s = s0; s < s1; ++s) {
                if (F.out[s].self.tag != 0) continue;
                const SolveTask& T = F.task[s];
                bool ok = true; double a = T.e0r, b = T.e0t, d = T.e0t2;
                auto extp = [&](int32_t g, double& v) { if (g >= 0 && !local(g)) { const Granule q = G[g]; if (q.tag == 0 || q.tag >= tag) ok = false; else v = q.v; } };
                extp(T.predSelf, a); extp(T.predT, b); extp(T.predT2, d);
                ext[s - s0] = ok; er[s - s0] = a; et[s - s0] = b; et2[s - s0] = d;
            }
            for (;;) {                      // sweeps: results of a sweep become visible to the next one
                std::vector<std::pair<int32_t, SolveOut>> produced;
                for (int32_t s = s0; s < s1; ++s) {
                    if (F.out[s].self.tag != 0 || !ext[s - s0]) continue;
                    const SolveTask& T = F.task[s];
                    bool ok = true; double a = er[s - s0], b = et[s - s0], d = et2[s - s0];
                    auto loc = [&](int32_t g, double& v) { if (g >= 0 && local(g)) { const Granule q = G[g]; if (q.tag == 0) ok = false; else v = q.v; } };
                    loc(T.predSelf, a); loc(T.predT, b); loc(T.predT2, d);
                    if (ok) produced.push_back({s, solve_compute(T, a, b, d, tag, K, m, dt)});
                }
                if (produced.empty()) break;
                for (auto& pr : produced) F.out[pr.first] = pr.second;
            }
            for (int32_t s = s0; s < s1; ++s) if (F.out[s].self.tag == 0) ++pending;
        }
        if (pending == 0) break;
        if (launches > 8 * (int64_t)F.N) return -launches;
    }
    return launches;
}

}  // namespace

extern "C" int emu_erode_composite(int32_t N, const int32_t* off, const int32_t* adj, float* e, const float* xyz,
                                   const uint8_t* ocean, int32_t hIters, double K, double m, double dt,
                                   int32_t tIters, double talus, double kThermal, int32_t gIters, double gStrength,
                                   const float* dist, double* stats /* 8 */) {
    if (gIters < 0) gIters = 0;
    int32_t total = hIters > tIters ? hIters : tIters;
    if (gIters > total) total = gIters;
    if (total <= 0) return 0;
    Emu E;
    Fields& F = E.F;
    F.N = N; F.off = off; F.adj = adj; F.dist = dist; F.xyz = xyz; F.ocean = ocean; F.e = e;
    E.coast.resize(N); E.e2.resize(N); E.rank.assign(N, -1); E.target.resize(N); E.cellDist.resize(N); E.flow.resize(N);
    E.accA.resize(N); E.jumpA.resize(N); E.task.resize(N); E.out.resize(N); E.doneAt.resize(N); E.totalExcess.resize(N);
    E.glac.assign(N, 0.f); E.iceTarget.resize(N); E.iceFlow.resize(N); E.iceUp.resize(N); E.arank.resize(N);
    F.e2 = E.e2.data(); F.rank = E.rank.data(); F.target = E.target.data(); F.cellDist = E.cellDist.data();
    F.flow = E.flow.data(); F.accA = E.accA.data(); F.jumpA = E.jumpA.data(); F.task = E.task.data(); F.out = E.out.data();
    F.doneAt = E.doneAt.data(); F.totalExcess = E.totalExcess.data(); F.glac = E.glac.data();
    F.iceTarget = E.iceTarget.data(); F.iceFlow = E.iceFlow.data(); F.iceUp = E.iceUp.data(); F.arank = E.arank.data(); E.blocker.assign(N, -1); F.blocker = E.blocker.data(); E.blk.assign(N, -1); F.blk = E.blk.data(); E.tr.assign(N, TargetRank{-1, -1}); F.tr = E.tr.data(); E.ev.resize(N); F.ev = std::getenv("WO_NO_EVENT_LISTS") ? nullptr : E.ev.data(); E.me.assign(N, 0.f); F.me = E.me.data();
    F.coast = E.coast.data();
    for (int32_t r = 0; r < N; ++r) E.coast[r] = coast_flag(F, r);
    for (int32_t r = 0; r < N; ++r) if (!ocean[r]) E.land.push_back(r);
    E.landIdx = E.land;
    F.L = (int32_t)E.land.size(); F.land = E.land.data();
    if (F.L == 0) return 0;
    for (int32_t i = 0; i < F.L; ++i) E.rank[E.land[i]] = i;
    std::vector<int32_t> level(N, 1);
    // odd iteration counts exercise the patch-local schedule (patches = chunks of the ascending-id land list), even ones the level rounds
    const bool usePatches = (hIters & 1) != 0;
    std::vector<int32_t> slotOf(N, -1);
    for (int32_t i = 0; i < F.L; ++i) slotOf[E.landIdx[i]] = i;
    F.slotOf = usePatches ? slotOf.data() : nullptr;
    FloodScratch fs;
    if (hIters > 0) priority_flood_carve_host(N, off, adj, xyz, e, ocean, 0.5, fs);
    const bool glacial = gIters > 0 && gStrength > 0;
    if (glacial) for (int32_t r = 0; r < N; ++r) E.glac[r] = glac_index_cell(F, r, gStrength);
    const double gScale = gIters > 0 ? 1.0 / gIters : 0;
    const double gCarve = 0.02 * gScale, gConv = 0.01 * gScale, gDep = 0.005 * gScale, gFjord = 0.015 * gScale;
    const int32_t midIter = (int32_t)std::floor(total * 0.75 + 0.5);
    bool midDone = false;
    int rc = 0;
    for (int32_t iter = 0; iter < total; ++iter) {
        if (!midDone && iter >= midIter) { midDone = true; priority_flood_carve_host(N, off, adj, xyz, e, ocean, 0.85, fs); }
        const bool gNow = iter < gIters && glacial, hNow = iter < hIters;
        if (gNow || hNow) stable_sort_desc(E);
        if (gNow) {
            for (int32_t r = 0; r < N; ++r) ice_receiver_cell(F, r);
            for (int32_t r = 0; r < N; ++r) if (ocean[r]) { F.iceFlow[r] = 0; F.iceUp[r] = 0; }
            int64_t n1 = run_rounds(E.landIdx, [&](int32_t t, int32_t k) { return ice_accumulate_task(F, t, k); });
            if (n1 < 0) rc = 10;
            E.iceRounds += n1;
            std::vector<int32_t> act;
            for (int32_t r = 0; r < N; ++r) { carve_setup_cell(F, r); }
            for (int32_t r : E.landIdx) if (F.arank[r] != WO_NOT_DONE) act.push_back(r);
            int64_t n2 = run_rounds(act, [&](int32_t r, int32_t k) { return carve_task(F, r, k, gCarve, gConv, gStrength); });
            if (n2 < 0) rc = 11;
            E.carveRounds += n2;
            for (int32_t r = 0; r < N; ++r) moraine_fjord_cell(F, r, gDep, gFjord);
        }
        if (hNow) {
            if (gNow) stable_sort_desc(E);
            for (int32_t r = 0; r < N; ++r) receiver_cell(F, r);
            // flow: pointer doubling over the forward forest
            std::vector<int32_t> act, nj;
            std::vector<uint32_t> snap;
            for (int32_t r = 0; r < N; ++r) { F.accA[r] = ocean[r] ? 0u : 1u; F.jumpA[r] = ocean[r] ? -1 : flow_forward_target(F, r); }
            for (int32_t r : E.landIdx) if (F.jumpA[r] >= 0) act.push_back(r);
            while (!act.empty()) {
                ++E.flowRounds;
                snap.resize(act.size()); nj.resize(act.size());
                for (size_t i = 0; i < act.size(); ++i) { const int32_t d = act[i]; snap[i] = F.accA[d]; nj[i] = F.jumpA[F.jumpA[d]]; }
                std::vector<int32_t> nxt;
                for (size_t i = 0; i < act.size(); ++i) {
                    const int32_t d = act[i];
                    F.accA[F.jumpA[d]] += snap[i];
                    F.jumpA[d] = nj[i];
                    if (nj[i] >= 0) nxt.push_back(d);
                }
                act.swap(nxt);
            }
            for (int32_t c = 0; c < N; ++c) {
                if (ocean[c]) { F.flow[c] = 0; continue; }
                flow_final_cell(F, c);                      // flow + the event list of c (k_flow_final)
            }
            F.solveK = K; F.solveM = m; F.solveDt = dt;          // folded into the task records by solve_setup
            for (int32_t r = 0; r < N; ++r) solve_setup_cell(F, r);
            int64_t n3 = usePatches ? run_solve_patches_emu(E, F.L, K, m, dt) : run_solve_rounds_pred(E, level, K, m, dt);
            if (n3 < 0) rc = 12;
            E.solveRounds += n3; if (n3 > E.maxSolveRounds) E.maxSolveRounds = n3;
            for (int32_t r = 0; r < N; ++r) F.e2[r] = solve_final_cell(F, r);
            std::memcpy(e, F.e2, sizeof(float) * (size_t)N);
        }
        if (iter < tIters) {
            for (int32_t r = 0; r < N; ++r) F.me[r] = masked_elev_cell(F, r);
            for (int32_t r = 0; r < N; ++r) thermal_excess_cell(F, r, talus);
            { double inShare[WO_MAX_DEG], outShare[WO_MAX_DEG]; int32_t inRank[WO_MAX_DEG];
              // alternate the two own-turn forms (stored / recomputed shares) so both stay covered
              for (int32_t r = 0; r < N; ++r) F.e2[r] = thermal_apply_cell(F, r, talus, kThermal, inShare, inRank, 1, (r & 1) ? outShare : nullptr); }
            std::memcpy(e, F.e2, sizeof(float) * (size_t)N);
        }
    }
    if (glacial) {
        for (int32_t r = 0; r < N; ++r) F.e2[r] = glacial_blend_cell(F, e, r);
        std::memcpy(e, F.e2, sizeof(float) * (size_t)N);
    }
    if (stats) {
        stats[0] = (double)F.L; stats[1] = (double)E.solveRounds; stats[2] = (double)E.maxSolveRounds;
        stats[3] = (double)E.flowRounds; stats[4] = (double)E.iceRounds; stats[5] = (double)E.carveRounds;
    }
    return rc;
}

extern "C" void emu_jacobi(int32_t kind, int32_t N, const int32_t* off, const int32_t* adj, float* e,
                           const uint8_t* ocean, int32_t iterations, double strength) {
    Fields F{};
    F.N = N; F.off = off; F.adj = adj; F.ocean = ocean; F.e = e;
    std::vector<uint8_t> coast(N);
    F.coast = coast.data();
    for (int32_t r = 0; r < N; ++r) coast[r] = coast_flag(F, r);
    std::vector<float> tmp(N), original(e, e + N);
    for (int32_t it = 0; it < iterations; ++it) {
        for (int32_t r = 0; r < N; ++r)
            tmp[r] = kind == 0 ? smooth_cell(F, e, r, strength) : kind == 1 ? sharpen_cell(F, e, original.data(), r, strength)
                                                                           : creep_cell(F, e, r, strength);
        std::memcpy(e, tmp.data(), sizeof(float) * (size_t)N);
    }
}

extern "C" void emu_warp(int32_t N, const int32_t* off, const int32_t* adj, float* e, const float* xyz, double seed,
                         double strength, const float* hot) {
    if (strength <= 0) return;
    Fields F{};
    F.N = N; F.off = off; F.adj = adj; F.xyz = xyz; F.e = e;
    uint8_t P[512], M[512];
    noise_tables(seed + 9999, P, M);
    std::vector<float> out(N);
    const double maxAmp = 0.12 * strength, bias = 0.25 + 0.5 * strength;
    for (int32_t r = 0; r < N; ++r) {
        const int32_t src = warp_source_cell(F, P, M, r, maxAmp);
        out[r] = warp_blend(e[r], e[src], bias, hot != nullptr, hot ? hot[r] : 0.f);
    }
    std::memcpy(e, out.data(), sizeof(float) * (size_t)N);
}

// priorityFloodCarve through the host stage's two pass-1 routes.  mode 0: the serial heap walk; mode 1: one heap per
// landmass with the tie-group checks (falls back to the serial walk when it cannot vouch for the result).
// stats: [calls, serialPass1, tieGroups, contested, openParents, unresolved, pathRedo, pass1Ms, pass23Ms] (mode + 10: two more slots, [replays, replayedLandmasses];
// mode + 100: the heights go in and out in land order)
extern "C" void emu_flood_host(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, float* e, const uint8_t* ocean, double cs,
                               int32_t mode, int32_t repeats, double* stats) {
    FloodScratch S;
    flood_build_static(N, off, adj, xyz, ocean, S);
    if (S.L == 0) return;
    FloodHostStats hs;
    std::vector<float> e0(e, e + N);
    for (int32_t k = 0; k < (repeats < 1 ? 1 : repeats); ++k) {
        std::memcpy(e, e0.data(), sizeof(float) * (size_t)N);
        if (mode >= 100) {                                  // the heights handed over in land order (FloodScratch::landOrder), as the mirrored planet does
            std::vector<float> land(S.L);
            for (int32_t i = 0; i < S.L; ++i) land[i] = e[S.landCell[i]];
            S.landOrder = true;
            flood_host_passes(land.data(), cs, S, &hs);
            S.landOrder = false;
            for (int32_t i = 0; i < S.L; ++i) e[S.landCell[i]] = land[i];
        }
        else if (mode % 10 == 0) { ++hs.calls; ++hs.serialPass1; flood_gather(e, S); flood_pass1_host(S); flood_pass23_host(e, cs, S); }
        else flood_host_passes(e, cs, S, &hs);
    }
    if (stats) {
        // (the first nine are what older callers size their buffer for; WO_EMU_FLOOD_STATS11 callers pass eleven slots)
        const double v[11] = {(double)hs.calls, (double)hs.serialPass1, (double)hs.tieGroups, (double)hs.contested, (double)hs.openParents,
                              (double)hs.unresolved, (double)hs.pathRedo, hs.pass1Ms, hs.pass23Ms, (double)hs.replays, (double)hs.replayedLandmasses};
        for (int i = 0; i < (mode >= 10 ? 11 : 9); ++i) stats[i] = v[i];
    }
}

// The flood call of a planet split into landmass shares (decomposed.py): share k sees the other shares' landmasses as ocean and
// floods with flood_host_passes_exchange; the shares run as threads and exchange through a barrier, as ranks would through a
// collective.  useExchange == 0: every share floods on its own with flood_host_passes (the pre-round-4 behaviour, for contrast).
// e: in the initial field, out the merged result.  stats: [exchange gathers, whole-planet floods, replays] summed over the shares.
#include <condition_variable>
#include <mutex>
#include <thread>
namespace {
struct ShareBarrier {
    std::mutex m; std::condition_variable cv; int n, waiting = 0; long gen = 0;
    explicit ShareBarrier(int n_) : n(n_) {}
    void wait() { std::unique_lock<std::mutex> l(m); const long g = gen; if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); } else cv.wait(l, [&] { return gen != g; }); }
};
struct ShareGroup { ShareBarrier bar; std::vector<int32_t> flags; std::vector<float*> fields; const float* sent = nullptr; const int32_t* owner; int32_t N; explicit ShareGroup(int n) : bar(n), flags(n, 0), fields(n, nullptr) {} };
struct ShareCtx { ShareGroup* g; int rank; };
int share_exchange(void* user, int32_t phase, void* buf, int64_t n) {
    ShareCtx* c = (ShareCtx*)user; ShareGroup& G = *c->g;
    if (phase == 0) {
        G.flags[c->rank] = *(int32_t*)buf;
        G.bar.wait();
        int32_t m = 0; for (int32_t f : G.flags) m = std::max(m, f);
        G.bar.wait();
        *(int32_t*)buf = m;
        return 0;
    }
    float* field = (float*)buf;
    if (phase >= 2) {                                  // the flooding share's land heights: 2 sends, 3 receives
        if (phase == 2) G.sent = field;
        G.bar.wait();
        if (phase == 3) std::memcpy(field, G.sent, sizeof(float) * (size_t)n);
        G.bar.wait();
        return 0;
    }
    G.fields[c->rank] = field;
    G.bar.wait();
    for (int64_t r = 0; r < n; ++r) { const int32_t o = G.owner[r]; if (o >= 0 && o != c->rank) field[r] = G.fields[o][r]; }
    G.bar.wait();
    return 0;
}
}  // namespace
extern "C" void emu_flood_shares(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, float* e, const uint8_t* trueOcean, const int32_t* owner,
                                 int32_t nShares, double cs, int32_t useExchange, double* stats) {
    ShareGroup G(nShares); G.owner = owner; G.N = N;
    std::vector<std::vector<float>> field(nShares, std::vector<float>(e, e + N));
    std::vector<FloodHostStats> hs(nShares);
    std::vector<FloodExchange> X(nShares);
    std::vector<ShareCtx> ctx(nShares);
    std::vector<std::thread> th;
    for (int k = 0; k < nShares; ++k) th.emplace_back([&, k]() {
        std::vector<uint8_t> mask(N);
        for (int32_t r = 0; r < N; ++r) mask[r] = (trueOcean[r] || owner[r] != k) ? 1 : 0;
        FloodScratch S;
        flood_build_static(N, off, adj, xyz, mask.data(), S);
        if (!useExchange) { if (S.L > 0) flood_host_passes(field[k].data(), cs, S, &hs[k]); return; }
        ctx[k] = ShareCtx{&G, k};
        X[k].on = true; X[k].fn = share_exchange; X[k].user = &ctx[k]; X[k].trueOcean.assign(trueOcean, trueOcean + N);
        flood_host_passes_exchange(N, off, adj, xyz, field[k].data(), cs, S, &hs[k], X[k]);
    });
    for (auto& t : th) t.join();
    double gathers = 0, globals = 0, replays = 0, received = 0;
    for (int k = 0; k < nShares; ++k) { gathers += (double)X[k].gathers; globals += (double)X[k].globalFloods; replays += (double)hs[k].replays; received += (double)X[k].received; }
    for (int32_t r = 0; r < N; ++r) if (owner[r] >= 0) e[r] = field[owner[r]][r];
    if (stats) { stats[0] = gathers; stats[1] = globals; stats[2] = replays; stats[3] = received; }
}

extern "C" void emu_flood(int32_t N, const int32_t* off, const int32_t* adj, float* e, const uint8_t* ocean, double cs) {
    FloodScratch fs;
    priority_flood_carve_host(N, off, adj, nullptr, e, ocean, cs, fs);
}


// Device flood (flood_ops.h): the same bodies and the same round / epoch control as k_fl_eval / k_fl_apply
// (flood_kernels.h), one "thread" at a time.  Jacobi: every evaluation of a round sees the labels of the previous
// round.  stats: [rounds, epochs, evaluations, changes, equal-key decisions, not-fixed cells, overflow, max stack depth]
extern "C" int emu_flood_device(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, float* e, const uint8_t* ocean, double cs,
                                int32_t acceptIdOrder, double* stats) {
    FloodScratch S;
    flood_build_static(N, off, adj, xyz, ocean, S);
    const int32_t L = S.L;
    if (L == 0) return 0;
    std::vector<double> nz(L); flood_cell_noise(S, nz.data());
    std::vector<int32_t> seedIdx(L, -1);
    for (size_t k = 0; k < S.seedCell.size(); ++k) seedIdx[S.seedCell[k]] = (int32_t)k;
    std::vector<float> eL(L);
    std::vector<FlHead> A(L), P(L), Fd(L);
    std::vector<unsigned long long> As((size_t)L * FL_LD), Ps((size_t)L * FL_LD), Fs((size_t)L * FL_LD);
    std::vector<int32_t> fdEpoch(L, 0), inDirty(L, 0); std::vector<uint8_t> isPending(L, 0);
    FloodDev D{};
    D.L = L; D.off = S.offL.data(); D.adj = S.adjL.data(); D.cell = S.landCell.data(); D.nz = nz.data(); D.seedIdx = seedIdx.data(); D.e = eL.data();
    D.A = A.data(); D.Astk = As.data(); D.P = P.data(); D.Pstk = Ps.data(); D.F = Fd.data(); D.Fstk = Fs.data(); D.fdEpoch = fdEpoch.data();
    D.inDirty = inDirty.data(); D.isPending = isPending.data();
    for (int32_t i = 0; i < L; ++i) {                                          // k_fl_init
        const float v = e[D.cell[i]]; eL[i] = v;
        FlHead h; h.par = FL_NONE; h.S = v; h.K = 0; h.dep = 0; h.top = 0; h.spare = 0;
        if (seedIdx[i] >= 0) { h.par = FL_SEED; h.K = (float)((double)v + nz[i]); h.dep = 1; h.top = fl_pack(h.K, D.cell[i]); As[(size_t)i * FL_LD] = h.top; }
        A[i] = h;
    }
    std::vector<int32_t> dirty, next, changed, pend[2];
    for (int32_t c : S.seedCell)                                               // k_fl_seed_dirty
        for (int32_t j = D.off[c]; j < D.off[c + 1]; ++j) { const int32_t y = D.adj[j]; if (seedIdx[y] < 0 && !inDirty[y]) { inDirty[y] = 1; dirty.push_back(y); } }
    int64_t rounds = 0, epochs = 0, evals = 0, changes = 0; int32_t epoch = 1, pcur = 0, relIdx = 0; bool release = false, over = false, done = false;
    int32_t maxDep = 0;
    while (!done) {
        const std::vector<int32_t>& list = release ? pend[relIdx] : dirty;
        changed.clear();
        for (int32_t x : list) {                                               // k_fl_eval
            inDirty[x] = 0;
            bool pendNew = false, o = false;
            const bool ch = flood_eval_cell(D, x, epoch, release, &pendNew, &o);
            if (o) over = true;
            if (ch) changed.push_back(x);
            if (pendNew) pend[pcur].push_back(x);
        }
        evals += (int64_t)list.size();
        next.clear();
        for (int32_t x : changed) {                                            // k_fl_apply
            flood_apply_cell(D, x, epoch);
            if (A[x].dep > maxDep) maxDep = A[x].dep;
            for (int32_t j = D.off[x]; j < D.off[x + 1]; ++j) { const int32_t y = D.adj[j]; if (seedIdx[y] < 0 && !inDirty[y]) { inDirty[y] = 1; next.push_back(y); } }
        }
        changes += (int64_t)changed.size();
        dirty.swap(next); ++rounds;
        if (release) { pend[relIdx].clear(); release = false; }
        if (dirty.empty()) {
            if (!pend[pcur].empty()) { ++epoch; ++epochs; release = true; relIdx = pcur; pcur ^= 1; }
            else done = true;
        }
        if (over || rounds > 4000000) break;
    }
    int64_t notFixed = 0, ties = 0;
    for (int32_t i = 0; i < L; ++i) { bool nf = false, tie = false; flood_verify_cell(D, i, &nf, &tie); notFixed += nf; ties += tie; }
    if (stats) { stats[0] = (double)rounds; stats[1] = (double)epochs; stats[2] = (double)evals; stats[3] = (double)changes; stats[4] = (double)ties;
                 stats[5] = (double)notFixed; stats[6] = over ? 1 : 0; stats[7] = (double)maxDep; }
    const bool ok = done && !over && notFixed == 0 && (ties == 0 || acceptIdOrder);
    flood_gather(e, S);
    if (ok) {
        std::vector<int32_t> par(L), root(L, -1); std::vector<float> surf(L);
        for (int32_t i = 0; i < L; ++i) { par[i] = A[i].par; surf[i] = A[i].S; }
        for (int32_t i = 0; i < L; ++i) {                                      // k_fl_jump / k_fl_export
            if (par[i] == FL_NONE) continue;
            int32_t c = i; while (par[c] >= 0) c = par[c];
            root[i] = seedIdx[c];
        }
        flood_import_pass1(par.data(), surf.data(), root.data(), S);
    } else flood_pass1_host(S);
    flood_pass23_host(e, cs, S);
    return ok ? 0 : 1;
}

// plate projection: the kernel bodies of csrc/plates_ops.h, bucket grid included, one "thread" per cell
extern "C" void emu_project_plates(int32_t N, const float* xyz, int32_t NC, const int32_t* cOff, const int32_t* cAdj, const float* cxyz,
                                   const int32_t* cPlate, double seed, int32_t numPlates, int32_t* out) {
    CoarsePlates C{};
    C.NC = NC; C.off = cOff; C.adj = cAdj; C.xyz = cxyz; C.plate = cPlate; C.grid = nullptr; C.gridZ = 64; C.gridLon = 128;
    std::vector<int32_t> grid((size_t)C.gridZ * C.gridLon);
    for (int32_t b = 0; b < C.gridZ * C.gridLon; ++b) grid[b] = plate_grid_cell(C, b);
    C.grid = grid.data();
    uint8_t P[512], M[512];
    noise_tables(seed + 999, P, M);
    const double coarseEdgeRad = 3.141592653589793 / std::sqrt((double)NC);
    double lowPlateT = 0;
    if (numPlates >= 0) lowPlateT = std::max(0.0, std::min(1.0, (80 - numPlates) / 60.0));
    const double perturbAmp = coarseEdgeRad * (1.5 + 1.0 * lowPlateT);
    for (int32_t r = 0; r < N; ++r) out[r] = plate_project_cell(C, P, M, xyz, r, perturbAmp);
}

extern "C" void emu_set_lookahead(int v) { LOOKAHEAD = v; }

// ---- assignElevation: the same host stage as the product plus the per-cell bodies driven on the CPU ----
#include "../../planet_heightmap_generation_amd/csrc/elevation_host.h"
#include "../../planet_heightmap_generation_amd/csrc/elevation_bfs.h"
#include "emu_fifo_bfs.h"

// The device formulation of the FIFO BFS fields (elevation_bfs.h), kernel by kernel with the atomics executed one
// "thread" at a time: level-synchronous claims for the fields without attributes; push / count / scan / assign per level
// for the fields that carry the attributes of the first (coast: the strongest) parent in queue order.
static int g_bfsDevice = 0;
extern "C" void emu_set_bfs_device(int v) { g_bfsDevice = v; }
namespace {
void emu_bfs_fields(const ElevMesh& M, const ElevInputs& I, ElevHostState& H, const ElevParams& Q, int32_t maxCD, double maxStress) {
    const int32_t N = M.N;
    BfsCtx B{N, M.off, M.adj, H.isOcean.data(), I.plate};
    std::vector<int32_t> cur, nxt, pushPos(N), cnt, base;
    std::vector<unsigned long long> attrKey(N);
    auto run = [&](int32_t mode, std::vector<int32_t> seeds, float* dist, float init, float* a0, float* a1, uint8_t* a2, int32_t maxDist) {
        const bool carry = a0 != nullptr;
        for (int32_t r = 0; r < N; ++r) { dist[r] = init; if (a0) a0[r] = 0; if (a1) a1[r] = 0; if (a2) a2[r] = 0; pushPos[r] = 0x7fffffff; attrKey[r] = 0; }    // k_bfs_init
        for (int32_t r : seeds) {                                                                                              // k_bfs_seed
            dist[r] = 0.0f;
            if (mode == BFS_COAST || mode == BFS_BACKARC || mode == BFS_ARC) { const double v = (double)H.stress[r] / maxStress; a0[r] = (float)(v < 1.0 ? v : 1.0); }
            if (mode == BFS_COAST) { a1[r] = H.subduct[r]; a2[r] = H.btype[r] == 1 ? 1 : 0; }
        }
        cur = seeds;
        for (int32_t level = 1; level <= maxDist; ++level) {
            const float nd = (float)level;
            nxt.clear();
            if (!carry) {                                                                                                      // k_bfs_plain
                const uint32_t ndBits = bfs_f32_bits(nd);
                for (size_t i = cur.size(); i-- > 0;) {              // any order: take the reverse of the queue on purpose
                    const int32_t r = cur[i];
                    for (int32_t j = M.off[r]; j < M.off[r + 1]; ++j) {
                        const int32_t nr = M.adj[j];
                        if (nd < dist[nr] && bfs_admit(B, mode, nr, r)) {
                            const uint32_t old = bfs_f32_bits(dist[nr]);
                            if (old > ndBits) { dist[nr] = nd; nxt.push_back(nr); }
                        }
                    }
                }
            } else {
                const int32_t n = (int32_t)cur.size();
                for (int32_t i = n - 1; i >= 0; --i) {                                                                         // k_bfs_push (reverse order: atomics commute)
                    const int32_t r = cur[i];
                    for (int32_t j = M.off[r]; j < M.off[r + 1]; ++j) {
                        const int32_t nr = M.adj[j];
                        if (!(nd <= dist[nr])) continue;
                        if (!bfs_admit(B, mode, nr, r)) continue;
                        if (nd < dist[nr] && i < pushPos[nr]) pushPos[nr] = i;
                        if (mode == BFS_COAST) { const unsigned long long k = bfs_attr_key(a0[r], i); if (k > attrKey[nr]) attrKey[nr] = k; }
                    }
                }
                cnt.assign(n, 0); base.assign(n, 0);
                for (int32_t i = 0; i < n; ++i) {                                                                              // k_bfs_count
                    const int32_t r = cur[i];
                    for (int32_t j = M.off[r]; j < M.off[r + 1]; ++j) { const int32_t nr = M.adj[j]; if (pushPos[nr] == i && nd < dist[nr]) ++cnt[i]; }
                }
                int32_t run = 0; for (int32_t i = 0; i < n; ++i) { base[i] = run; run += cnt[i]; }                             // k_bfs_scan
                nxt.assign(run, -1);
                for (int32_t i = n - 1; i >= 0; --i) {                                                                         // k_bfs_assign
                    const int32_t r = cur[i];
                    int32_t k = base[i];
                    for (int32_t j = M.off[r]; j < M.off[r + 1]; ++j) {
                        const int32_t nr = M.adj[j];
                        if (!(pushPos[nr] == i && nd < dist[nr])) continue;
                        nxt[k++] = nr;
                        int32_t src = r;
                        if (mode == BFS_COAST) src = cur[bfs_attr_pos(attrKey[nr])];
                        if (a0) a0[nr] = a0[src];
                        if (a1) a1[nr] = a1[src];
                        if (a2) a2[nr] = a2[src];
                        dist[nr] = nd;
                    }
                }
            }
            cur.swap(nxt);
        }
    };
    auto seeds_of = [&](auto pred) { std::vector<int32_t> q; for (int32_t r = 0; r < N; ++r) if (pred(r)) q.push_back(r); return q; };
    H.dBdry.resize(N); H.coastStressMax.resize(N); H.coastSubductMax.resize(N); H.coastConvergent.resize(N);
    H.riftDist.resize(N); H.ridgeDist.resize(N); H.fractureDist.resize(N); H.backArcDist.resize(N); H.backArcStress.resize(N); H.arcDist.resize(N); H.arcStress.resize(N);
    const uint8_t* oc = H.isOcean.data();
    run(BFS_COAST, seeds_of([&](int32_t r) { for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni) if (oc[M.adj[ni]] != oc[r]) return true; return false; }),
        H.dBdry.data(), (float)(maxCD + 1), H.coastStressMax.data(), H.coastSubductMax.data(), H.coastConvergent.data(), maxCD);
    run(BFS_RIFT, seeds_of([&](int32_t r) { return H.btype[r] == 2 && !H.hasOcean[r]; }), H.riftDist.data(), INFINITY, nullptr, nullptr, nullptr, Q.riftHalfWidth);
    run(BFS_RIDGE, seeds_of([&](int32_t r) { return H.btype[r] == 2 && H.bothOcean[r]; }), H.ridgeDist.data(), INFINITY, nullptr, nullptr, nullptr, Q.ridgeHalfWidth);
    run(BFS_FRACTURE, seeds_of([&](int32_t r) { return H.btype[r] == 3 && H.bothOcean[r]; }), H.fractureDist.data(), INFINITY, nullptr, nullptr, nullptr, Q.fractureHalfWidth);
    run(BFS_BACKARC, seeds_of([&](int32_t r) { return H.btype[r] == 1 && H.hasOcean[r] && (double)H.subduct[r] < 0.50; }), H.backArcDist.data(), INFINITY,
        H.backArcStress.data(), nullptr, nullptr, Q.baEnd);
    run(BFS_ARC, seeds_of([&](int32_t r) { return H.btype[r] == 1 && H.bothOcean[r] && (double)H.subduct[r] < 0.45; }), H.arcDist.data(), (float)(Q.maxArcDist + 1),
        H.arcStress.data(), nullptr, nullptr, Q.maxArcDist);
}
}  // namespace

extern "C" int emu_assign_elevation(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, const int32_t* r_plate,
                                    int32_t numIds, const uint8_t* hasVec, const double* pole, const double* omega, const uint8_t* isOcean,
                                    const double* density, const int32_t* plateSeeds, int32_t nSeeds, const int32_t* r_super, int32_t sNumIds,
                                    const uint8_t* sHasVec, const double* sPole, const double* sOmega, const uint8_t* sIsOcean,
                                    const double* sDensity, const uint8_t* perm, const uint8_t* pm12, double noiseMag, double seed, double spread,
                                    float* out_elev, float* out_stress, float* out_dl, int32_t* mountain, int32_t* coastline, int32_t* ocean,
                                    int32_t* counts) {
    PlateTable T{numIds, hasVec, pole, omega, isOcean, density}, TS{sNumIds, sHasVec, sPole, sOmega, sIsOcean, sDensity};
    const bool hasSuper = r_super != nullptr;
    enum { NT = 9 };
    std::vector<uint8_t> tabs(NT * 1024), hs3(1024);
    std::memcpy(tabs.data(), perm, 512); std::memcpy(tabs.data() + 512, pm12, 512);
    const double offs[NT] = {0, 419, 557, 77, 133, 211, 307, 501, 502};
    for (int k = 1; k < NT; ++k) noise_tables(seed + offs[k], tabs.data() + k * 1024, tabs.data() + k * 1024 + 512);
    noise_tables(seed + 503, hs3.data(), hs3.data() + 512);
    auto tab = [&](int k) { return NoiseTab{tabs.data() + k * 1024, tabs.data() + k * 1024 + 512}; };
    CollisionHost hS, hP;
    auto collide = [&](const int32_t* plate, const PlateTable& tt, CollisionHost& h) {
        h.resize(N);
        CollisionOut O{h.stress.data(), h.subduct.data(), h.btype.data(), h.bothOcean.data(), h.hasOcean.data(), h.setCode.data()};
        for (int32_t r = 0; r < N; ++r) collision_cell(r, N, off, adj, xyz, plate, tt, tab(0), O);
    };
    collide(r_plate, T, hS);
    if (hasSuper) collide(r_super, TS, hP);
    ElevMesh M{N, off, adj, xyz};
    ElevInputs I{};
    I.plate = r_plate; I.plates = T; I.plateSeeds = plateSeeds; I.numPlateSeeds = nSeeds; I.superPlate = r_super; I.superPlates = TS;
    I.seed = seed; I.spread = spread; I.noiseMag = noiseMag; I.hsNoise3 = NoiseTab{hs3.data(), hs3.data() + 512};
    ElevHostState H; ElevParams Q{}; std::vector<Dome> domes;
    if (g_bfsDevice)
        elevation_host_stage(M, I, hS, hasSuper ? &hP : nullptr, H, Q, domes,
                             [&](const ElevParams& Qs, int32_t maxCD, double maxStress) { emu_bfs_fields(M, I, H, Qs, maxCD, maxStress); });
    else elevation_host_stage(M, I, hS, hasSuper ? &hP : nullptr, H, Q, domes,
                              [&](const ElevParams& Qs, int32_t maxCD, double maxStress) { fifo_bfs_fields(M, I, H, Qs, maxCD, maxStress); });
    ElevFields F{};
    F.xyz = xyz; F.plate = r_plate; F.isOcean = H.isOcean.data(); F.stress = H.stress.data(); F.subduct = H.subduct.data(); F.btype = H.btype.data();
    F.distMountain = H.distMountain.data(); F.distOcean = H.distOcean.data(); F.distCoastline = H.distCoastline.data(); F.distCoast = H.distCoast.data();
    F.distCoastLand = H.distCoastLand.data(); F.dBdry = H.dBdry.data(); F.coastStressMax = H.coastStressMax.data(); F.coastSubductMax = H.coastSubductMax.data();
    F.coastConvergent = H.coastConvergent.data(); F.riftDist = H.riftDist.data(); F.ridgeDist = H.ridgeDist.data(); F.fractureDist = H.fractureDist.data();
    F.backArcDist = H.backArcDist.data(); F.backArcStress = H.backArcStress.data(); F.arcDist = H.arcDist.data(); F.arcStress = H.arcStress.data();
    F.elev = out_elev; F.dl = out_dl;
    if (out_dl) std::memset(out_dl, 0, sizeof(float) * (size_t)DL_COUNT * N);
    for (int32_t r = 0; r < N; ++r) {
        float e = elevation_main_cell(F, Q, T, r, tab(0), tab(1), tab(2));
        e = coastal_cell(F, Q, r, e, tab(0), tab(3), tab(4), tab(5));
        e = arc_cell(F, Q, r, e, tab(6));
        e = hotspot_cell(F, Q, r, e, domes.data(), tab(7), tab(8));
        out_elev[r] = compress_cell(e);
    }
    std::memcpy(out_stress, H.stress.data(), sizeof(float) * (size_t)N);
    std::memcpy(mountain, H.mountain.data(), H.mountain.size() * 4); std::memcpy(coastline, H.coastline.data(), H.coastline.size() * 4);
    std::memcpy(ocean, H.ocean.data(), H.ocean.size() * 4);
    counts[0] = (int32_t)H.mountain.size(); counts[1] = (int32_t)H.coastline.size(); counts[2] = (int32_t)H.ocean.size();
    return 0;
}

extern "C" double emu_pair_intensity(int32_t a, int32_t b) { return pair_intensity(a, b); }

// The two forms of a solve turn (erode_ops.h): solve_apply, as the serial loop writes it, and solve_apply_flat, every expression
// evaluated and selected (what a wave of k_solve_flowing runs for whichever lanes are ready).  n random tasks over every flag
// combination, heights and factors of the magnitudes the solve sees plus the awkward ones (0, equal heights, huge / tiny factors,
// missing t2 with cellDistT 0); returns the number of tasks whose outputs differ in any bit.
extern "C" int64_t emu_solve_turn_forms_differ(int64_t n, uint64_t seed) {
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto unit = [&]() { return (double)(rnd() >> 11) / 9007199254740992.0; };
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        SolveTask T{};
        const uint32_t combo = (uint32_t)(rnd() % 7);
        // flags: bit0 target is ocean, bit1 t2 is ocean, bit2 has a target, bit3 has a t2
        static const uint32_t kFlags[7] = {0u, 4u, 4u | 1u, 4u | 8u, 4u | 8u | 2u, 4u | 8u, 4u};
        T.flags = kFlags[combo];
        auto height = [&]() { const uint64_t k = rnd() % 10; return k == 0 ? 0.0f : k == 1 ? 1e-7f : k == 2 ? 1.5f : (float)(unit() * 1.3); };
        float er = height(), et = height(), et2 = height();
        if (rnd() % 8 == 0) et = er;                      // flats
        if (rnd() % 8 == 0) et2 = et;
        if (rnd() % 16 == 0) et = -et;                    // (heights below zero reach the clamp)
        const uint64_t fk = rnd() % 12;
        T.factor = fk == 0 ? 0.0 : fk == 1 ? 1e-300 : fk == 2 ? 1e30 : fk == 3 ? 1e-12 : unit() * std::pow(10.0, (double)(int)(rnd() % 7) - 3.0);
        T.cellDistT = (T.flags & 8u) ? (float)(1e-4 + unit() * 2e-3) : 0.0f;
        if ((T.flags & 8u) && rnd() % 16 == 0) T.cellDistT = 1e-6f;
        T.e0r = er; T.e0t = et; T.e0t2 = et2;
        const SolvePrepared P = solve_prepare(T, 3e-4, 0.5, 1.0);
        const SolveOut a = solve_apply(T, P, (double)er, (double)et, (double)et2, 7);
        const SolveOut b = solve_apply_flat(T, P, (double)er, (double)et, (double)et2, 7);
        if (std::memcmp(&a, &b, sizeof(a)) != 0) ++bad;
    }
    return bad;
}

extern "C" int64_t emu_flood_queues_differ(int64_t ops, uint64_t seed) { return flood_queues_differ(ops, seed); }
