// Device context, planet handle, pass orchestration and the device half of the C ABI (include/worogen.h).
//
// Orchestration follows the reference's call structure (js/terrain-post.js:369-707 for erodeComposite,
// js/planet-worker.js:40-102 for the caller) but every per-cell loop is a kernel launch on the planet's
// stream and every order-defined loop is executed as synchronous dependency rounds (erode_ops.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>

#include "../../include/worogen.h"
#include "device.h"
#include "elevation_host.h"
#include "elevation_kernels.h"
#include "kernels_impl.h"
#include "flood_kernels.h"
#include "elevation_bfs.h"
#include "noise.h"

namespace wo {

const char* const kFamilyNames[FAM_COUNT] = {
    "coast_flags", "smooth_elevation", "sharpen_ridges", "soil_creep", "warp_terrain", "noise_eval", "synthetic_terrain",
    "ocean_from_elevation", "sort_keys", "sort_radix", "rank_scatter", "receivers", "flow_climb",
    "flow_final", "solve_setup", "solve_round", "solve_final", "thermal_excess", "thermal_apply",
    "glac_index", "ice_receivers", "ice_round", "carve_setup", "carve_round", "moraine_fjord", "glacial_blend", "solve_patch", "elev_collisions", "elev_uplift_fused", "plate_grid", "plate_project", "smooth_field", "flood_eval", "flood_apply", "flood_misc", "climate_sweeps", "basin_layout", "basin_sort", "solve_basin", "flow_tiles", "misc", "event_pair_empty", "event_pair_noop_kernel", "event_pair_two_noop_kernels"};

hipEvent_t profile_event(wo_planet* p) {
    if (!p->eventPool.empty()) { hipEvent_t e = p->eventPool.back(); p->eventPool.pop_back(); return e; }
    hipEvent_t e; WO_HIP(hipEventCreate(&e)); return e;
}
void profile_resolve(wo_planet* p) {
    if (p->pending.empty()) return;
    WO_HIP(hipStreamSynchronize(p->ctx->stream));
    if (p->side) WO_HIP(hipStreamSynchronize(p->side));
    for (auto& pe : p->pending) {
        float ms = 0; WO_HIP(hipEventElapsedTime(&ms, pe.a, pe.b));
        p->famMs[pe.fam] += ms; p->famLaunches[pe.fam] += 1;
        p->eventPool.push_back(pe.a); p->eventPool.push_back(pe.b);
    }
    p->pending.clear();
}

template <class T> static T* dalloc(size_t n) { void* q = nullptr; WO_HIP(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T))); return (T*)q; }
template <class T> static void dfree(T*& q) { if (q) { (void)hipFree(q); q = nullptr; } }

constexpr int WO_PATCH_TOTAL_SLOTS = 4096;         // pending-total slots of the patch solve (one per launch, reused modulo)


static void ensure_scratch(wo_planet* p) {
    if (p->scratch) return;
    const size_t N = (size_t)p->N;
    p->d_landIdx = dalloc<int32_t>(N); p->d_land[0] = dalloc<int32_t>(N); p->d_land[1] = dalloc<int32_t>(N);
    p->d_keys[0] = dalloc<uint32_t>(N); p->d_keys[1] = dalloc<uint32_t>(N);
    p->d_rank = dalloc<int32_t>(N);
    p->d_cellDist = dalloc<float>(N); p->d_flow = dalloc<float>(N); p->d_task = dalloc<SolveTask>(N); p->d_out = dalloc<SolveOut>(N); WO_HIP(hipMemset(p->d_out, 0, N * sizeof(SolveOut)));   /* tags of the unchecked basin passes count up from here: no stale tag may look like a coming one */ p->d_flowCnt = dalloc<int32_t>(N); WO_HIP(hipMemset(p->d_flowCnt, 0, (size_t)N * 4)); p->d_tr = dalloc<TargetRank>(N); p->d_ev = dalloc<EventList>(N); p->d_me = dalloc<float>(N); p->d_carveSlot = dalloc<int32_t>(N);
    p->d_accCnt = dalloc<unsigned long long>(N); p->d_jump = dalloc<int32_t>(N); p->d_nj = dalloc<int32_t>(N);
    p->d_doneAt = dalloc<int32_t>(N);
    p->d_totalExcess = dalloc<double>(N);
    p->d_glac = dalloc<float>(N); p->d_iceFlow = dalloc<float>(N); p->d_iceTarget = dalloc<int32_t>(N); p->d_arank = dalloc<int32_t>(N);
    p->d_iceUp = dalloc<uint8_t>(N);
    p->d_listA = dalloc<int32_t>(N); p->d_listB = dalloc<int32_t>(N); p->d_counters = dalloc<int32_t>(8);
    p->d_patchOrder = dalloc<int32_t>(N); p->d_slotOf = dalloc<int32_t>(N); p->d_patchPending = dalloc<int32_t>(N / WO_PATCH + 2); p->d_patchTotals = dalloc<int32_t>(WO_PATCH_TOTAL_SLOTS); p->d_patchBlk = dalloc<int32_t>(N);
    WO_HIP(hipHostMalloc((void**)&p->h_patchTotals, (size_t)WO_PATCH_TOTAL_SLOTS * sizeof(int32_t)));          // read-back buffer of run_solve_patches
    p->sortTempBytes = sort_temp_bytes(p->N);
    WO_HIP(hipMalloc(&p->d_sortTemp, std::max<size_t>(p->sortTempBytes, 16)));
    WO_HIP(hipMemsetAsync(p->d_glac, 0, N * sizeof(float), p->ctx->stream));
    p->scratch = true;
}

}  // namespace wo

wo::Fields wo_planet::fields() const {
    wo::Fields F{};
    F.N = N; F.xcdTile = wo::xcd_tile(N); F.tileLds = opt.tileLds ? 1 : 0; F.off = d_off; F.adj = d_adj; F.dist = d_dist; F.xyz = d_xyz; F.ocean = d_ocean; F.coast = d_coast;
    F.e = d_e; F.e2 = d_e2; F.L = L; F.land = d_land[landCur]; F.landIdx = landIdentity ? nullptr : d_landIdx; F.xcdTileL = wo::xcd_tile(L > 0 ? L : 1); F.rank = d_rank; F.target = nullptr; F.tr = d_tr; F.ev = d_ev; F.me = d_me; F.carveSlot = d_carveSlot; F.carveDeps = nullptr; F.carveDepCnt = d_carveDepCnt; F.carveDepPos = d_carveDepPos; F.cellDist = d_cellDist;
    F.flow = d_flow; F.accA = nullptr; F.accB = nullptr; F.accCnt = d_accCnt; F.jumpA = d_jump; F.jumpB = nullptr;
    F.task = d_task; F.out = d_out; F.slotOf = (patchVersion >= 0) ? d_slotOf : nullptr; F.blk = d_patchBlk; F.doneAt = d_doneAt;
    F.totalExcess = d_totalExcess; F.glac = d_glac; F.iceTarget = d_iceTarget; F.iceFlow = d_iceFlow; F.iceUp = d_iceUp; F.arank = d_arank; F.blocker = d_nj;
    return F;
}

namespace wo {

static inline void swap_elev(wo_planet* p) { std::swap(p->d_e, p->d_e2); }
// The device ocean mask changed (or may have): the host copy is refreshed lazily.  The flood's cached
// open-ocean / seed data is only rebuilt when the mask really differs (a "reapply" with the same mask keeps it).
static inline void ocean_changed(wo_planet* p) { p->h_ocean_valid = false; }
__global__ __launch_bounds__(wo::WO_BLOCK) void k_mask_differs(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int32_t N, int32_t* flag) {
    bool diff = false;
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < N; r += gridDim.x * blockDim.x) diff |= a[r] != b[r];
    if (diff) atomicOr(flag, 1);
}
static int32_t read_count(wo_planet* p, const int32_t* d_ptr);
// The host's copy of the ocean mask, brought up to date.  A mask written on the device (ocean_from_elevation, restore_state, synthetic_terrain) is
// first compared ON the device with the mask the host copy describes (a copy of it kept there): a step that re-derives the same mask — every step of
// the bench — then costs one 10 us kernel and one word instead of a 10 MB download and a 10 MB memcmp.
static void refresh_host_ocean(wo_planet* p) {
    if (p->h_ocean_valid) return;
    hipStream_t s = p->ctx->stream;
    if (p->d_oceanKnown && p->oceanKnownValid && p->h_ocean.size() == (size_t)p->N) {
        WO_HIP(hipMemsetAsync(p->d_maskDiff, 0, sizeof(int32_t), s));
        hipLaunchKernelGGL(k_mask_differs, dim3(wo::blocks_for(p->N, 1 << 15)), dim3(wo::WO_BLOCK), 0, s, (const uint8_t*)p->d_ocean, (const uint8_t*)p->d_oceanKnown, p->N, p->d_maskDiff);
        if (read_count(p, p->d_maskDiff) == 0) { p->h_ocean_valid = true; return; }
    }
    hvec<uint8_t> tmp(p->N);
    WO_HIP(hipMemcpyAsync(tmp.data(), p->d_ocean, p->N, hipMemcpyDeviceToHost, s));
    WO_HIP(hipStreamSynchronize(s));
    if (tmp.size() != p->h_ocean.size() || std::memcmp(tmp.data(), p->h_ocean.data(), tmp.size()) != 0) {
        p->h_ocean.swap(tmp);
        p->flood.staticValid = false; ++p->oceanVersion;
    }
    p->h_ocean_valid = true;
    if (!p->d_oceanKnown) { WO_HIP(hipMalloc((void**)&p->d_oceanKnown, (size_t)p->N)); WO_HIP(hipMalloc((void**)&p->d_maskDiff, sizeof(int32_t))); }
    WO_HIP(hipMemcpyAsync(p->d_oceanKnown, p->d_ocean, (size_t)p->N, hipMemcpyDeviceToDevice, s));
    p->oceanKnownValid = true;
}

// One device integer for the host, through a host-mapped word the host polls: ~10 us from the producing kernel's end to the next
// launch instead of ~30 (copy kernel, stream synchronisation, wake-up).  Falls back to the copy when the word is not there or the
// poll outlasts 2 s (a faulted stream must surface as an error, not as a hang).
static int32_t publish_and_wait(wo_planet* p, const int32_t* d_ptr) {
    const bool poll = true;
    hipStream_t s = p->ctx->stream;
    if (poll && !p->h_word) {
        if (hipHostMalloc((void**)&p->h_word, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&p->d_word, p->h_word, 0) != hipSuccess) { p->h_word = nullptr; p->d_word = nullptr; (void)hipGetLastError(); }
        else *p->h_word = 0;
    }
    if (poll && p->h_word) {
        const uint32_t serial = ++p->wordSerial;
        hipLaunchKernelGGL(k_publish_count, dim3(1), dim3(1), 0, s, d_ptr, p->d_word, serial);
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t spins = 0;; ++spins) {
            const unsigned long long w = __atomic_load_n(p->h_word, __ATOMIC_ACQUIRE);
            if ((uint32_t)(w >> 32) == serial) return (int32_t)(uint32_t)w;
            if ((spins & 0xfff) == 0xfff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
            __builtin_ia32_pause();
        }
        WO_HIP(hipStreamSynchronize(s));                           // surfaces a device fault; a merely slow kernel ends up here too
        const unsigned long long w = __atomic_load_n(p->h_word, __ATOMIC_ACQUIRE);
        if ((uint32_t)(w >> 32) == serial) return (int32_t)(uint32_t)w;
    }
    WO_HIP(hipMemcpyAsync(p->h_count, d_ptr, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    WO_HIP(hipStreamSynchronize(s));
    return p->h_count[0];
}

int32_t read_count(wo_planet* p, const int32_t* d_ptr) { return publish_and_wait(p, d_ptr); }

// Patch-local solve driver: launches k_solve_patch until no task is pending.  Returns the number of launches.
// basin: the store order is the group-major one of basin_layout() and the first launch of the pass is k_solve_basin, which
// normally leaves nothing pending; whatever it does leave (layout off: see basin.hip) is finished by k_solve_patch launches.
static int64_t run_solve_patches(wo_planet* p, const Fields& F, double K, double m, double dt, bool basin, bool countersCleared = false, bool deferCheck = false, int32_t passTag = 1) {
    hipStream_t s = p->ctx->stream;
    const int np = p->numPatches;
    if (basin && deferCheck && countersCleared) {
        // the one launch of the basin solve, and no look at what it left: tasks left pending are counted into a word that is not
        // cleared during the call and looked at where the host synchronises anyway (erode_composite: RedoWithChecks)
        basin_solve_launch(p, F, passTag, p->d_pendingEver);
        p->lastPatchLaunches = 1;
        return 1;
    }
    if (!countersCleared) launch(p, FAM_MISC, k_fill_i32, blocks_for(np, 64), WO_BLOCK, p->d_patchPending, basin ? 0 : 1, (int32_t)np);
    // one pending-total slot per launch, cleared once per pass (a memset per launch was 13.6 k fill kernels per step)
    int32_t* tot = p->d_patchTotals;
    if (!countersCleared) WO_HIP(hipMemsetAsync(tot, 0, (size_t)WO_PATCH_TOTAL_SLOTS * sizeof(int32_t), s));
    int64_t launches = 0;
    // polling passes per visit (kernels_impl.h); launches from lateFrom on (few patches still open: no queue of visits behind a
    // long one) may poll longer
    constexpr int spinCap = WO_PATCH_SPIN_CAP, lateFrom = WO_PATCH_LATE_FROM, lateCap = WO_PATCH_LATE_SPIN_CAP;
    int64_t need = 0;
    for (int32_t tag = 1;; ) {
        // The pending totals are read back (one stream sync) after every burst.  The number of launches a pass needs barely
        // changes from one erosion iteration to the next, so the first burst is the previous need plus one; then small bursts.
        // Launches after the one that emptied the list find nothing pending and return at once.
        const int32_t first = tag;
        const int burst = (tag == 1) ? std::max<int>(1, (int)std::min<int64_t>(p->lastPatchLaunches, WO_PATCH_TOTAL_SLOTS / 2)) : 3;
        for (int b = 0; b < burst; ++b, ++tag) {
            // A burst never spans a wrap of the slot ring: the wrap's memset would clear the slots of the burst's earlier
            // tags before they are read back (they would read as 0 = "nothing pending").  The wrap starts the next burst.
            if (tag % WO_PATCH_TOTAL_SLOTS == 0) {
                if (b > 0) break;
                WO_HIP(hipMemsetAsync(tot, 0, (size_t)WO_PATCH_TOTAL_SLOTS * sizeof(int32_t), s));   // wrapped: every earlier slot has been read back
            }
            if (basin && tag == 1) basin_solve_launch(p, F, tag, tot + 1);
            else {
            // the first k_solve_patch launch after a lean setup: its blocker hints are made from the records now (only tasks the basin
            // launch left pending get one; normally there is no such launch at all)
            if (basin && tag == 2 && F.solveLean) launch(p, FAM_MISC, k_solve_blk_init, blocks_for(p->L, 4096), WO_BLOCK, F, p->L);
            launch(p, FAM_SOLVE_PATCH, k_solve_patch, np, WO_PATCH_THREADS, F, p->L, tag, p->d_patchPending, tot + (tag % WO_PATCH_TOTAL_SLOTS), K, m, dt,
                   (int32_t*)nullptr, (int32_t)(tag >= lateFrom ? lateCap : spinCap));
            }
            ++launches;
        }
        if (basin && first == 1 && tag == 2) {                     // the usual case: the one launch of the basin solve, one total to look at
            if (publish_and_wait(p, tot + 1) == 0) need = 1;
        } else {
            WO_HIP(hipMemcpyAsync(p->h_patchTotals, tot, (size_t)WO_PATCH_TOTAL_SLOTS * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            WO_HIP(hipStreamSynchronize(s));
            for (int32_t t = first; t < tag && !need; ++t) if (p->h_patchTotals[t % WO_PATCH_TOTAL_SLOTS] == 0) need = t;
        }
        if (need) break;
        if (launches > 4 * (int64_t)p->N + 1024) throw HipError{"patch solve does not converge"};
    }
    p->lastPatchLaunches = basin ? need : need + 1;      // basin: the first burst is exactly what the last pass needed (one launch)
    return need;
}

// Stage timings (wo_last_stage_timing, the reference's _postTiming).  A pair of event records costs ~10 us of stream time, and the
// composite loop has five stages per iteration (a 1.6 ms iteration: 8 ms of a 460 ms step went into timing it), so inside the loop
// only every 8th iteration is bracketed (`on`) and a stage's sum is scaled by occurrences / bracketed occurrences; stages outside
// the loop (setup, floods) are always bracketed.  WO_STAGE_TIMING=all, or per-launch profiling, brackets every iteration.
struct StageClock {
    wo_planet* p; std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> ev;
    std::map<std::string, std::pair<int64_t, int64_t>> seen;      // stage -> {occurrences, bracketed}
    bool on = true, open = false;
    explicit StageClock(wo_planet* pl) : p(pl) {}
    void begin(const char* name) {
        auto& c = seen[name]; ++c.first;
        open = on;
        if (!on) return;
        ++c.second;
        hipEvent_t a = profile_event(p), b = profile_event(p); WO_HIP(hipEventRecord(a, p->ctx->stream)); ev.push_back({name, {a, b}});
    }
    void end() { if (open) WO_HIP(hipEventRecord(ev.back().second.second, p->ctx->stream)); open = false; }
    void count_only(const char* name) { ++seen[name].first; }       // an occurrence that is not bracketed (an iteration replayed from the graph)
    void finish() {
        // the brackets are handed to the planet as they are; wo_last_stage_timing turns them into milliseconds when somebody asks
        for (auto& b : p->stageBrackets) { p->eventPool.push_back(b.a); p->eventPool.push_back(b.b); }      // (an earlier call's, never asked for)
        p->stageBrackets.clear(); p->stageSeen.clear();
        std::vector<std::string> order;
        for (auto& e : ev) {
            bool known = false;
            for (auto& n : order) if (n == e.first) { known = true; break; }
            if (!known) order.push_back(e.first);
            p->stageBrackets.push_back({e.first, e.second.first, e.second.second});
        }
        for (auto& n : order) p->stageSeen.push_back({n, seen[n]});
        ev.clear();
        p->stageTiming.clear();
        p->stagePending = true;
    }
    // a call that ends by exception (RedoWithChecks, a failed exchange) never reaches finish(): its events go back to the pool
    ~StageClock() { for (auto& e : ev) { p->eventPool.push_back(e.second.first); p->eventPool.push_back(e.second.second); } }
};

// ---------------------------------------------------------------------------------------------------
// priorityFloodCarve (js/terrain-post.js:59-215).  Pass 1 (the noise-keyed best-first flood) runs on the device as a
// label-correcting fixed point (flood_ops.h / flood_kernels.h); passes 2 and 3 (sequential carve along the drain
// paths, ordered fix-up) run per drainage tree on host threads (flood_host.cc) from the downloaded drainTo / surface.
// Which pass 1 runs is read from the environment at every call: WO_FLOOD=device selects the device formulation; the
// default is the host heap walk (the reference's own order, and at 10^7 cells still the faster of the two: see
// DESIGN.md).  The device result is the reference's whenever no decision hinged on two EQUAL keys (the heap orders
// those by its array mechanics); the decisions that did are counted and, unless WO_FLOOD_TIES=id accepts the cell-id
// order, pass 1 is redone on the host.
// ---------------------------------------------------------------------------------------------------
struct FloodRun { double deviceMs = 0; int64_t rounds = 0, epochs = 0, evals = 0, ties = 0; bool usedDevice = false, fellBack = false; FloodHostStats host; };

static void flood_gpu_free(wo_flood_gpu& G) {
    dfree(G.off); dfree(G.adj); dfree(G.cell); dfree(G.seedIdx); dfree(G.seeds); dfree(G.nz); dfree(G.e);
    dfree(G.A); dfree(G.P); dfree(G.F); dfree(G.Astk); dfree(G.Pstk); dfree(G.Fstk); dfree(G.fdEpoch); dfree(G.inDirty); dfree(G.isPending);
    for (auto& l : G.lists) dfree(l);
    if (G.ctrl) { (void)hipFree(G.ctrl); G.ctrl = nullptr; }
    if (G.h_ctrl) { (void)hipHostFree(G.h_ctrl); G.h_ctrl = nullptr; }
    dfree(G.jump); dfree(G.outPar); dfree(G.outRoot); dfree(G.outSurf);
    if (G.h_par) { (void)hipHostFree(G.h_par); G.h_par = nullptr; }
    if (G.h_root) { (void)hipHostFree(G.h_root); G.h_root = nullptr; }
    if (G.h_surf) { (void)hipHostFree(G.h_surf); G.h_surf = nullptr; }
    G.version = -1; G.cap = 0; G.L = 0;
}

static void flood_gpu_upload_static(wo_planet* p) {
    wo_flood_gpu& G = p->fgpu;
    const FloodScratch& S = p->flood;
    hipStream_t s = p->ctx->stream;
    const int32_t L = S.L;
    if (L > G.cap) {
        flood_gpu_free(G);
        G.cap = L + L / 16 + 1024;
        const size_t C = (size_t)G.cap;
        G.off = dalloc<int32_t>(C + 1); G.cell = dalloc<int32_t>(C); G.seedIdx = dalloc<int32_t>(C); G.seeds = dalloc<int32_t>(C); G.nz = dalloc<double>(C);
        G.e = dalloc<float>(C);
        G.A = dalloc<FlHead>(C); G.P = dalloc<FlHead>(C); G.F = dalloc<FlHead>(C);
        G.Astk = dalloc<unsigned long long>(C * FL_LD); G.Pstk = dalloc<unsigned long long>(C * FL_LD); G.Fstk = dalloc<unsigned long long>(C * FL_LD);
        G.fdEpoch = dalloc<int32_t>(C); G.inDirty = dalloc<int32_t>(C); G.isPending = dalloc<uint8_t>(C);
        for (auto& l : G.lists) l = dalloc<int32_t>(C);
        WO_HIP(hipMalloc(&G.ctrl, sizeof(FlCtrl)));
        WO_HIP(hipHostMalloc(&G.h_ctrl, sizeof(FlCtrl)));
        G.jump = dalloc<int32_t>(C); G.outPar = dalloc<int32_t>(C); G.outRoot = dalloc<int32_t>(C); G.outSurf = dalloc<float>(C);
        WO_HIP(hipHostMalloc((void**)&G.h_par, C * 4)); WO_HIP(hipHostMalloc((void**)&G.h_root, C * 4)); WO_HIP(hipHostMalloc((void**)&G.h_surf, C * 4));
    }
    dfree(G.adj);
    G.adj = dalloc<int32_t>(S.adjL.size());
    std::vector<double> nz((size_t)L);
    flood_cell_noise(S, nz.data());
    std::vector<int32_t> seedIdx((size_t)L, -1);
    for (size_t k = 0; k < S.seedCell.size(); ++k) seedIdx[S.seedCell[k]] = (int32_t)k;
    WO_HIP(hipMemcpyAsync(G.off, S.offL.data(), (size_t)(L + 1) * 4, hipMemcpyHostToDevice, s));
    WO_HIP(hipMemcpyAsync(G.adj, S.adjL.data(), S.adjL.size() * 4, hipMemcpyHostToDevice, s));
    WO_HIP(hipMemcpyAsync(G.cell, S.landCell.data(), (size_t)L * 4, hipMemcpyHostToDevice, s));
    WO_HIP(hipMemcpyAsync(G.seedIdx, seedIdx.data(), (size_t)L * 4, hipMemcpyHostToDevice, s));
    WO_HIP(hipMemcpyAsync(G.seeds, S.seedCell.data(), S.seedCell.size() * 4, hipMemcpyHostToDevice, s));
    WO_HIP(hipMemcpyAsync(G.nz, nz.data(), (size_t)L * 8, hipMemcpyHostToDevice, s));
    WO_HIP(hipStreamSynchronize(s));          // the staging vectors go out of scope
    G.L = L; G.nSeeds = (int32_t)S.seedCell.size(); G.version = S.staticVersion;
}

// Returns true when the device result (h_par / h_surf / h_root, land-index space) is valid and may be imported.
static bool flood_device_pass1(wo_planet* p, FloodRun& R) {
    wo_flood_gpu& G = p->fgpu;
    hipStream_t s = p->ctx->stream;
    if (G.version != p->flood.staticVersion) flood_gpu_upload_static(p);
    const int32_t L = G.L;
    if (L == 0) return false;
    FloodDev D{};
    D.L = L; D.off = G.off; D.adj = G.adj; D.cell = G.cell; D.nz = G.nz; D.seedIdx = G.seedIdx; D.e = G.e;
    D.A = G.A; D.Astk = G.Astk; D.P = G.P; D.Pstk = G.Pstk; D.F = G.F; D.Fstk = G.Fstk; D.fdEpoch = G.fdEpoch; D.inDirty = G.inDirty; D.isPending = G.isPending;
    FlLists Ls{{G.lists[0], G.lists[1]}, {G.lists[2], G.lists[3]}, G.lists[4]};
    FlCtrl* C = (FlCtrl*)G.ctrl; FlCtrl* hC = (FlCtrl*)G.h_ctrl;
    hipEvent_t e0 = profile_event(p), e1 = profile_event(p);
    WO_HIP(hipEventRecord(e0, s));
    launch(p, FAM_FLOOD_MISC, k_fl_init, blocks_for(L, 4096), WO_BLOCK, D, G.e, (const float*)p->d_e, C);
    launch(p, FAM_FLOOD_MISC, k_fl_seed_dirty, blocks_for(G.nSeeds, 1024), WO_BLOCK, D, (const int32_t*)G.seeds, G.nSeeds, Ls, C);
    const int grid = blocks_for(L / 4 + 1, 1024);
    constexpr int64_t maxRounds = 200000;
    bool done = false, over = false;
    int batch = 32;
    for (int64_t launched = 0; launched < maxRounds && !done;) {
        for (int b = 0; b < batch; ++b) {
            launch(p, FAM_FLOOD_EVAL, k_fl_eval, grid, WO_BLOCK, D, Ls, C);
            launch(p, FAM_FLOOD_APPLY, k_fl_apply, grid, WO_BLOCK, D, Ls, C);
        }
        launched += batch;
        WO_HIP(hipMemcpyAsync(hC, C, sizeof(FlCtrl), hipMemcpyDeviceToHost, s));
        WO_HIP(hipStreamSynchronize(s));
        done = hC->done != 0; over = hC->overflow != 0;
        if (over) break;
        if (batch < 256) batch *= 2;
    }
    bool ok = done && !over;
    if (ok) {
        launch(p, FAM_FLOOD_MISC, k_fl_verify, blocks_for(L, 4096), WO_BLOCK, D, C);
        launch(p, FAM_FLOOD_MISC, k_fl_jump_init, blocks_for(L, 4096), WO_BLOCK, D, G.jump);
        for (int k = 0; k < 16; ++k) launch(p, FAM_FLOOD_MISC, k_fl_jump, blocks_for(L, 4096), WO_BLOCK, G.jump, L);    // 2^16 hops
        launch(p, FAM_FLOOD_MISC, k_fl_export, blocks_for(L, 4096), WO_BLOCK, D, (const int32_t*)G.jump, G.outPar, G.outSurf, G.outRoot);
        WO_HIP(hipMemcpyAsync(G.h_par, G.outPar, (size_t)L * 4, hipMemcpyDeviceToHost, s));
        WO_HIP(hipMemcpyAsync(G.h_surf, G.outSurf, (size_t)L * 4, hipMemcpyDeviceToHost, s));
        WO_HIP(hipMemcpyAsync(G.h_root, G.outRoot, (size_t)L * 4, hipMemcpyDeviceToHost, s));
        WO_HIP(hipMemcpyAsync(hC, C, sizeof(FlCtrl), hipMemcpyDeviceToHost, s));
    }
    WO_HIP(hipEventRecord(e1, s));
    WO_HIP(hipStreamSynchronize(s));
    float ms = 0; WO_HIP(hipEventElapsedTime(&ms, e0, e1));
    p->eventPool.push_back(e0); p->eventPool.push_back(e1);
    R.deviceMs += ms; R.rounds += hC->rounds; R.epochs += hC->epochs; R.evals += hC->evals; R.usedDevice = true;
    if (ok && hC->notFixed != 0) ok = false;                // cannot happen at termination; refuse the result if it does
    if (ok) {
        R.ties += hC->ties;
        if (hC->ties > 0) ok = false;
    }
    if (!ok) R.fellBack = true;
    return ok;
}

// every cell of the planet in Morton order, if the mirror has worked it out (mask-independent, once per planet): the flood's tables filter it
static const int32_t* morton_if_known(const wo_planet* p) {
    return (!p->h_xyz.empty() && p->mirror.h_morton.size() == (size_t)p->N) ? p->mirror.h_morton.data() : nullptr;
}
static void flood_stage(wo_planet* p, double carveStrength, FloodRun& R) {
    hipStream_t s = p->ctx->stream;
    const size_t bytes = (size_t)p->N * sizeof(float);
    const bool timing = p->opt.floodTiming;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[flood stage] %-12s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t0).count());
        t0 = now;
    };
    if (timing) { WO_HIP(hipStreamSynchronize(s)); lap("drain gpu"); }
    refresh_host_ocean(p);
    lap("ocean mask");
    FloodScratch& S = p->flood;
    if (!S.staticValid || S.staticN != p->N)
        flood_build_static(p->N, p->h_off.data(), p->h_adj.data(), p->h_xyz.empty() ? nullptr : p->h_xyz.data(), p->h_ocean.data(), S, morton_if_known(p));
    lap("static");
    if (S.L == 0 && !p->floodX.on) return;       // (a share without land still takes part in the exchange)
    const bool hostOnly = p->floodX.on || !p->opt.floodDevice;
    WO_HIP(hipMemcpyAsync(p->h_pinned, p->d_e, bytes, hipMemcpyDeviceToHost, s));
    const bool useDevice = !hostOnly && flood_device_pass1(p, R);       // synchronises the stream
    if (hostOnly) WO_HIP(hipStreamSynchronize(s));
    lap(hostOnly ? "D2H" : "device pass1");
    if (useDevice) {
        flood_gather(p->h_pinned, S);
        flood_import_pass1(p->fgpu.h_par, p->fgpu.h_surf, p->fgpu.h_root, S);
        flood_pass23_host(p->h_pinned, carveStrength, S);
    } else if (p->floodX.on) {
        if (p->floodX.trueOcean.size() != (size_t)p->N) throw HipError{"flood exchange: the true ocean mask does not fit the planet"};
        const int rc = flood_host_passes_exchange(p->N, p->h_off.data(), p->h_adj.data(), p->h_xyz.empty() ? nullptr : p->h_xyz.data(), p->h_pinned, carveStrength, S, &R.host, p->floodX);
        if (rc) throw HipError{"flood exchange: the host's exchange function failed (status " + std::to_string(rc) + ")"};
    } else {
        flood_host_passes(p->h_pinned, carveStrength, S, &R.host);
    }
    lap(useDevice ? "import + host pass2+3" : "host passes");
    WO_HIP(hipMemcpyAsync(p->d_e, p->h_pinned, bytes, hipMemcpyHostToDevice, s));
    if (timing) { WO_HIP(hipStreamSynchronize(s)); lap("H2D"); }
}

// The flood stage of a planet inside its land-first mirror.  The mirror numbers the land cells 0 .. L-1 in Morton order of their positions,
// which is the order of the host flood's own land list (both come from morton_order_cells on the same mask; checked once per (mirror,
// flood tables) pair by comparing the two lists): the first L floats of the mirrored field ARE the flood's land heights.  So the stage
// copies 4 L bytes each way instead of 4 N (11 instead of 40 MB at the benched size), the host passes index them directly
// (FloodScratch::landOrder) and the field never leaves the mirror (no scatter into the planet's order before, no gather after).
static bool flood_land_is_mirror_prefix(wo_planet* p) {
    auto& M = p->mirror;
    const FloodScratch& S = p->flood;
    if (p->floodPrefixMirror != M.version || p->floodPrefixStatic != S.staticVersion) {
        p->floodPrefixMirror = M.version; p->floodPrefixStatic = S.staticVersion;
        p->floodPrefixOk = S.L == p->L && M.h_perm.size() == (size_t)p->N && (size_t)S.L <= M.h_perm.size() &&
                           std::memcmp(M.h_perm.data(), S.landCell.data(), sizeof(int32_t) * (size_t)S.L) == 0;
    }
    return p->floodPrefixOk;
}
static void flood_stage_land(wo_planet* p, double carveStrength, FloodRun& R) {
    hipStream_t s = p->ctx->stream;
    FloodScratch& S = p->flood;
    const size_t bytes = (size_t)S.L * sizeof(float);
    const bool timing = p->opt.floodTiming;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[flood stage] %-12s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t0).count());
        t0 = now;
    };
    if (timing) { WO_HIP(hipStreamSynchronize(s)); lap("drain gpu"); }
    // The land heights travel through the planet's own hipHostMalloc'ed buffer (allocated once per planet, N floats).  Round 5 page-locked the flood's
    // own array instead (hipHostRegister of a THP-advised heap block, 0.5 ms of a 312 ms step): user-pointer registrations of transparent-huge-page memory
    // are invalidated whenever the kernel collapses or splits those pages, and a copy in flight at that moment faults — the SIGABRT under
    // hipStreamSynchronize of the round-5 GPU suite.  Nothing in this library registers memory it did not get from hipHostMalloc.
    float* host = p->h_pinned;
    WO_HIP(hipMemcpyAsync(host, p->d_e, bytes, hipMemcpyDeviceToHost, s));
    WO_HIP(hipStreamSynchronize(s));
    lap("D2H (land)");
    S.landOrder = true;
    try { flood_host_passes(host, carveStrength, S, &R.host); } catch (...) { S.landOrder = false; throw; }
    S.landOrder = false;
    lap("host passes");
    WO_HIP(hipMemcpyAsync(p->d_e, host, bytes, hipMemcpyHostToDevice, s));
    if (timing) { WO_HIP(hipStreamSynchronize(s)); lap("H2D (land)"); }
}

static void coast_flags(wo_planet* p) {
    launch(p, FAM_COAST, k_coast, xcd_grid(p->N), WO_BLOCK, p->fields(), p->d_coast);
}

// ---------------------------------------------------------------------------------------------------
// Patch-major mirror of the mesh for erodeComposite.  The reference numbers the cells along the Fibonacci spiral, where the
// neighbours of consecutive cells lie in ~8 separate clusters 5*sqrt(N) ids apart and a 128-byte line of per-cell state
// holds 32 cells of which 9 are land: the neighbour gathers of every erosion pass cost an L2 transaction each.
// Inside erodeComposite nothing depends on the NAME of a cell except (i) the initial order of landCells (ascending id,
// js/terrain-post.js:384-390; later orders are stable sorts of it) and (ii) the flood's cellNoise(r) — rows keep the
// reference's adjacency order, ranks are positions in landCells.  So the whole call runs on a renamed copy of the graph
// (cells in Morton order of their positions: a wave's 64 cells and their neighbours share a few lines), with the initial
// land order mapped through the renaming and the field moved back to the planet's own order around each flood and at the end.
// Results are bit-identical (same operations on the same values in the same order); measured at 10 M cells the step went
// 1.10 -> 0.85 s.  WO_LAYOUT=index switches the mirror off.
// ---------------------------------------------------------------------------------------------------
static bool mirror_wanted(const wo_planet* p) {
    return !p->opt.layoutIndex && !p->h_xyz.empty() && p->N > 1;
}
// mask (may be null): the ocean mask of the call the mirror is entered for.  With a mask the renaming is LAND FIRST: the land
// cells in Morton order take the ids 0 .. L-1, the ocean cells follow in Morton order.  Every per-cell array of the erosion
// passes is then dense over the land (the passes touch land cells only: 28 % of a synthetic planet, so in plain Morton order a
// 64-byte line of per-cell state held ~4.5 useful entries of 16 and every 4-8-byte store of a pass was a partial line), the land
// list of the index-order passes is the identity, and a land cell's land neighbours sit 3.6x closer in memory.  The renaming is
// still only a renaming (rows keep their order, the initial land order is mapped through it): bit-identical results.  The mirror
// is rebuilt when a call arrives with a different mask (~50 ms at 10 M cells; the bench's steps all have the same one).
static void mirror_build(wo_planet* p, const uint8_t* mask = nullptr) {
    auto& M = p->mirror;
    const int32_t N = p->N; const size_t E = (size_t)p->E;
    // (the planet's own host mask carries a version: the 10 MB comparison — 0.4 ms of host time with the device idle, at the top of every call — is only made for other masks)
    const bool ownMask = mask && mask == p->h_ocean.data();
    if (M.built && ownMask && M.h_mask.size() == (size_t)N && p->mirrorMaskVersion == p->oceanVersion) return;
    if (M.built && (!mask || (M.h_mask.size() == (size_t)N && std::memcmp(M.h_mask.data(), mask, (size_t)N) == 0))) { if (ownMask) p->mirrorMaskVersion = p->oceanVersion; return; }
    hipStream_t s = p->ctx->stream;
    auto tLap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {          // WO_FLOOD_TIMING: what a new terrain's tables cost (bench.py: new_terrain_step_ms)
        if (!p->opt.floodTiming) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mirror] %-18s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - tLap).count());
        tLap = now;
    };
    if (M.h_morton.empty()) { morton_order_cells(N, p->h_xyz.data(), M.h_morton); lap("morton order"); }
    M.h_perm.resize(N);
    if (mask) {
        // land cells first, each class in Morton order: per range of the Morton list the land count, then both classes written behind the ranges before
        std::vector<int64_t> cnt(host_threads() + 2, 0);
        const int32_t* mo = M.h_morton.data();
        parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t c = 0; for (int64_t i = b; i < e; ++i) c += mask[mo[i]] ? 0 : 1; cnt[t + 1] = c; });
        for (size_t t = 1; t < cnt.size(); ++t) cnt[t] += cnt[t - 1];
        const int64_t nl = cnt.back();
        int32_t* perm = M.h_perm.data();
        parallel_ranges(N, [&](int64_t b, int64_t e, int t) {
            int64_t a = cnt[t], o = nl + (b - cnt[t]);          // land before this range: cnt[t]; ocean before it: b - cnt[t]
            for (int64_t i = b; i < e; ++i) { const int32_t r = mo[i]; if (mask[r]) perm[o++] = r; else perm[a++] = r; }
        });
        M.h_mask.assign(mask, mask + N);
        p->mirrorMaskVersion = ownMask ? p->oceanVersion : -1;
    } else {
        std::memcpy(M.h_perm.data(), M.h_morton.data(), (size_t)N * 4);
        M.h_mask.clear();
    }
    lap("land-first perm");
    hvec<int32_t> moff((size_t)N + 1);
    moff[0] = 0;
    {
        const int32_t* perm = M.h_perm.data(); const int32_t* ho = p->h_off.data(); int32_t* mo = moff.data();
        parallel_ranges(N, [&](int64_t b, int64_t e, int) { for (int64_t i = b; i < e; ++i) { const int32_t r = perm[i]; mo[i + 1] = ho[r + 1] - ho[r]; } });
        inclusive_scan_parallel(mo + 1, (int64_t)N);
    }
    lap("row offsets");
    if (!M.perm) {
        M.perm = dalloc<int32_t>(N); M.inv = dalloc<int32_t>(N); M.off = dalloc<int32_t>((size_t)N + 1); M.adj = dalloc<int32_t>(E + WO_ROW);
        M.dist = dalloc<float>(E + WO_ROW);
        WO_HIP(hipMemsetAsync(M.adj + E, 0, WO_ROW * sizeof(int32_t), s)); WO_HIP(hipMemsetAsync(M.dist + E, 0, WO_ROW * sizeof(float), s)); M.xyz = dalloc<float>(3 * (size_t)N); M.e = dalloc<float>(N); M.e2 = dalloc<float>(N);
        M.ocean = dalloc<uint8_t>(N); M.coast = dalloc<uint8_t>(N);
    }
    // the mirror's rows are rebuilt from the planet's own arrays: a scope must not be active (its pointers would be the mirror's)
    const int32_t* o_off = M.active ? M.o_off : p->d_off; const int32_t* o_adj = M.active ? M.o_adj : p->d_adj;
    const float* o_dist = M.active ? M.o_dist : p->d_dist; const float* o_xyz = M.active ? M.o_xyz : p->d_xyz;
    WO_HIP(hipMemcpyAsync(M.perm, M.h_perm.data(), (size_t)N * 4, hipMemcpyHostToDevice, s));
    WO_HIP(hipMemcpyAsync(M.off, moff.data(), ((size_t)N + 1) * 4, hipMemcpyHostToDevice, s));
    launch(p, FAM_MISC, k_mirror_invert, blocks_for(N, 4096), WO_BLOCK, (const int32_t*)M.perm, M.inv, N);
    launch(p, FAM_MISC, k_mirror_rows, blocks_for(N, 4096), WO_BLOCK, o_off, o_adj, o_dist, o_xyz, (const int32_t*)M.perm, (const int32_t*)M.inv, (const int32_t*)M.off, M.adj, M.dist, M.xyz, N);
    WO_HIP(hipStreamSynchronize(s));                    // moff / h_perm uploads done
    lap("upload + rows");
    M.built = true;
    ++M.version;
}
static void mirror_free(wo_planet* p) {
    auto& M = p->mirror;
    dfree(M.perm); dfree(M.inv); dfree(M.off); dfree(M.adj); dfree(M.dist); dfree(M.xyz); dfree(M.e); dfree(M.e2); dfree(M.ocean); dfree(M.coast); dfree(M.hot);
    M.built = false;
}
struct MirrorScope {
    wo_planet* p; bool on = false; float *cur = nullptr, *cur2 = nullptr;
    explicit MirrorScope(wo_planet* pl) : p(pl) {}
    MirrorScope(const MirrorScope&) = delete;
    void point_at_mirror(float* e, float* e2) {
        auto& M = p->mirror;
        p->d_off = M.off; p->d_adj = M.adj; p->d_dist = M.dist; p->d_xyz = M.xyz; p->d_e = e; p->d_e2 = e2; p->d_ocean = M.ocean; p->d_coast = M.coast;
    }
    void point_at_planet() {
        auto& M = p->mirror;
        p->d_off = M.o_off; p->d_adj = M.o_adj; p->d_dist = M.o_dist; p->d_xyz = M.o_xyz; p->d_e = M.o_e; p->d_e2 = M.o_e2; p->d_ocean = M.o_ocean; p->d_coast = M.o_coast;
    }
    // field and ocean mask into the mirror, the planet's pointers onto it
    void enter(const uint8_t* mask = nullptr) {
        if (!mirror_wanted(p) || p->mirror.active) return;
        mirror_build(p, mask);
        auto& M = p->mirror;
        const int32_t N = p->N;
        launch(p, FAM_MISC, k_mirror_gather_f32, blocks_for(N, 4096), WO_BLOCK, (const float*)p->d_e, (const int32_t*)M.perm, M.e, N);
        launch(p, FAM_MISC, k_mirror_gather_u8, blocks_for(N, 4096), WO_BLOCK, (const uint8_t*)p->d_ocean, (const int32_t*)M.perm, M.ocean, N);
        M.o_off = p->d_off; M.o_adj = p->d_adj; M.o_dist = p->d_dist; M.o_xyz = p->d_xyz; M.o_e = p->d_e; M.o_e2 = p->d_e2; M.o_ocean = p->d_ocean; M.o_coast = p->d_coast;
        point_at_mirror(M.e, M.e2);
        M.active = on = true;
    }
    // the current field back in the planet's own buffer and order, the planet's pointers on its own buffers (flood stage)
    void suspend() {
        if (!on) return;
        auto& M = p->mirror;
        launch(p, FAM_MISC, k_mirror_scatter_f32, blocks_for(p->N, 4096), WO_BLOCK, (const float*)p->d_e, (const int32_t*)M.perm, M.o_e, p->N);
        cur = p->d_e; cur2 = p->d_e2;
        point_at_planet();
    }
    void resume() {
        if (!on) return;
        auto& M = p->mirror;
        launch(p, FAM_MISC, k_mirror_gather_f32, blocks_for(p->N, 4096), WO_BLOCK, (const float*)M.o_e, (const int32_t*)M.perm, cur, p->N);
        point_at_mirror(cur, cur2);
    }
    void finish() {
        if (!on) return;
        suspend();
        p->mirror.active = on = false;
    }
    ~MirrorScope() { if (on) { point_at_planet(); p->mirror.active = false; } }      // error path: pointers only, the planet's field is what it was at the last suspend
};

// The planet's side stream (the basin layout beside the flow accumulation).
static void ensure_side_stream(wo_planet* p) {
    if (p->side) return;
    // the layout's chain of short launches is the longer of the two: at equal priority its workgroups queue behind the thousands of
    // the flow kernels' (a 22 us scatter pass took 108 us beside k_flow_final), so the side stream gets the highest priority
    int prLeast = 0, prGreatest = 0;
    WO_HIP(hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest));
    WO_HIP(hipStreamCreateWithPriority(&p->side, hipStreamNonBlocking, prGreatest));
    WO_HIP(hipEventCreateWithFlags(&p->evFork, hipEventDisableTiming)); WO_HIP(hipEventCreateWithFlags(&p->evJoin, hipEventDisableTiming));
}

// erodeComposite on the resident field (js/terrain-post.js:369-707)
// Thrown by erode_composite when a basin-solve launch whose result was not checked on the spot turns out to have left tasks pending:
// the caller restores the field and runs the call again with the check after every pass (erode_composite_checked).
struct RedoWithChecks {};

static void erode_composite(wo_planet* p, int32_t hIters, double K, double m, double dt, int32_t tIters, double talus,
                            double kThermal, int32_t gIters, double gStrength, bool checkEveryPass = true) {
    if (gIters < 0) gIters = 0;
    if (gStrength != gStrength) gStrength = 0;
    const int32_t total = std::max(hIters, std::max(tIters, gIters));
    for (auto& b : p->stageBrackets) { p->eventPool.push_back(b.a); p->eventPool.push_back(b.b); }
    p->stageBrackets.clear(); p->stageSeen.clear(); p->stagePending = false;
    p->stageTiming.clear(); p->erodeStats.clear();
    p->floodX.calls = 0; p->floodX.gathers = 0; p->floodX.globalFloods = 0; p->floodX.received = 0;       // per call (the stats of a step, not of the planet's life)
    if (total <= 0) return;
    ensure_scratch(p);
    hipStream_t s = p->ctx->stream;
    if (hIters > 0) WO_HIP(hipMemsetAsync(p->d_flowCnt, 0, (size_t)p->N * sizeof(int32_t), s));   // k_flow_final keeps it zero between iterations; a call that was cut short may not have
    if (hIters > 0 && p->d_ftInflow) { WO_HIP(hipMemsetAsync(p->d_ftInflow, 0, (size_t)p->N * 4, s)); WO_HIP(hipMemsetAsync(p->d_ftExtCnt, 0, (size_t)p->N * 4, s)); }   // (k_flow_tiles<true> keeps them zero)
    const int32_t N = p->N;
    const int gridN = xcd_grid(N);
    StageClock clk(p);
    int64_t carveFlowLeft = 0, maxSolve = 0, iceRounds = 0, carveRounds = 0, sorts = 0, patchLaunches = 0;
    double floodHostMs = 0;

    clk.begin("setup");
    refresh_host_ocean(p);          // the planet's own mask, before the pointers move
    MirrorScope mir(p);
    mir.enter(p->h_ocean.data());          // land first (mirror_build)
    p->landIdentity = mir.on && p->mirror.h_mask.size() == (size_t)N;      // land cells are the ids 0 .. L-1: the index-order passes skip the land list
    coast_flags(p);
    // landCells in ascending r (js/terrain-post.js:384-390): host-side compaction of the ocean mask
    {
        int32_t* hl = reinterpret_cast<int32_t*>(p->h_pinned);
        int32_t L = 0;
        // both lists only depend on the mask (and on whether the call runs on the mirror): a call with the mask of the previous one takes them as they are
        const bool listsKept = p->d_landInit && p->landListsOcean == p->oceanVersion && p->landListsMirror == mir.on && p->landListsL >= 0;
        if (listsKept) L = p->landListsL;
        else {
            const uint8_t* oc = p->h_ocean.data();
            std::vector<int64_t> cnt(host_threads() + 2, 0);
            parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t c = 0; for (int64_t r = b; r < e; ++r) c += oc[r] ? 0 : 1; cnt[t + 1] = c; });
            for (size_t t = 1; t < cnt.size(); ++t) cnt[t] += cnt[t - 1];
            parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t o = cnt[t]; for (int64_t r = b; r < e; ++r) if (!oc[r]) hl[o++] = (int32_t)r; });
            L = (int32_t)cnt.back();
        }
        p->L = L;
        if (L == 0) {
            // a share without land (landmass decomposition of a small planet) still answers the other shares' flood exchanges
            if (p->floodX.on) {
                const int32_t midIter0 = (int32_t)std::floor(total * 0.75 + 0.5);
                const int calls = (hIters > 0 ? 1 : 0) + (midIter0 < total ? 1 : 0);
                WO_HIP(hipMemcpyAsync(p->h_pinned, mir.on ? p->mirror.o_e : p->d_e, (size_t)N * sizeof(float), hipMemcpyDeviceToHost, s));
                WO_HIP(hipStreamSynchronize(s));
                if (!p->flood.staticValid || p->flood.staticN != N)
                    flood_build_static(N, p->h_off.data(), p->h_adj.data(), p->h_xyz.empty() ? nullptr : p->h_xyz.data(), p->h_ocean.data(), p->flood, morton_if_known(p));
                for (int k = 0; k < calls; ++k) {
                    const int rc = flood_host_passes_exchange(N, p->h_off.data(), p->h_adj.data(), p->h_xyz.empty() ? nullptr : p->h_xyz.data(), p->h_pinned, 0.5, p->flood, nullptr, p->floodX);
                    if (rc) throw HipError{"flood exchange: the host's exchange function failed (status " + std::to_string(rc) + ")"};
                }
            }
            mir.finish();
            clk.end(); clk.finish(); return;
        }
        if (listsKept) {
            WO_HIP(hipMemcpyAsync(p->d_land[0], p->d_landInit, (size_t)L * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        } else if (mir.on) {
            // initial landCells: the same cells in the same (ascending-r) order, under their mirror names
            WO_HIP(hipMemcpyAsync(p->d_listA, hl, (size_t)L * sizeof(int32_t), hipMemcpyHostToDevice, s));
            launch(p, FAM_MISC, k_mirror_map_i32, blocks_for(L, 4096), WO_BLOCK, (const int32_t*)p->d_listA, (const int32_t*)p->mirror.inv, p->d_land[0], L);
            WO_HIP(hipStreamSynchronize(s));
            // the list the index-order passes iterate: land cells in ascending mirror id
            const int32_t* perm = p->mirror.h_perm.data();
            const uint8_t* oc = p->h_ocean.data();
            std::vector<int64_t> cnt(host_threads() + 2, 0);
            parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t c = 0; for (int64_t i = b; i < e; ++i) c += oc[perm[i]] ? 0 : 1; cnt[t + 1] = c; });
            for (size_t t = 1; t < cnt.size(); ++t) cnt[t] += cnt[t - 1];
            parallel_ranges(N, [&](int64_t b, int64_t e, int t) { int64_t o = cnt[t]; for (int64_t i = b; i < e; ++i) if (!oc[perm[i]]) hl[o++] = (int32_t)i; });
            WO_HIP(hipMemcpyAsync(p->d_landIdx, hl, (size_t)L * sizeof(int32_t), hipMemcpyHostToDevice, s));
        } else {
            WO_HIP(hipMemcpyAsync(p->d_landIdx, hl, (size_t)L * sizeof(int32_t), hipMemcpyHostToDevice, s));
            WO_HIP(hipMemcpyAsync(p->d_land[0], p->d_landIdx, (size_t)L * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        }
        if (!listsKept) {
            if (!p->d_landInit) p->d_landInit = dalloc<int32_t>((size_t)N);
            WO_HIP(hipMemcpyAsync(p->d_landInit, p->d_land[0], (size_t)L * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
            p->landListsOcean = p->oceanVersion; p->landListsMirror = mir.on; p->landListsL = L;
        }
        WO_HIP(hipStreamSynchronize(s));       // h_pinned is reused by the flood stage
        p->landCur = 0;
        launch(p, FAM_MISC, k_init_rank, gridN, WO_BLOCK, p->d_rank, N);
        rank_from_land(p);      // thermal-only runs never sort: landCells stays in ascending-r order
        // spatial patches for the patch-local solve: land cells in Morton order (shared with the host flood's layout)
        if (hIters > 0) {
            if (!p->flood.staticValid || p->flood.staticN != N)
                flood_build_static(N, p->h_off.data(), p->h_adj.data(), p->h_xyz.data(), p->h_ocean.data(), p->flood, morton_if_known(p));
            if (p->patchVersion != p->flood.staticVersion || p->patchMirror != mir.on) {
                p->patchMirror = mir.on;
                if (mir.on) WO_HIP(hipMemcpyAsync(p->d_patchOrder, p->d_landIdx, (size_t)L * sizeof(int32_t), hipMemcpyDeviceToDevice, s));   // ascending mirror id IS Morton order
                else WO_HIP(hipMemcpyAsync(p->d_patchOrder, p->flood.landCell.data(), (size_t)L * sizeof(int32_t), hipMemcpyHostToDevice, s));
                launch(p, FAM_MISC, k_fill_i32, gridN, WO_BLOCK, p->d_slotOf, -1, N);
                launch(p, FAM_MISC, k_slot_scatter, blocks_for(L, 4096), WO_BLOCK, (const int32_t*)p->d_patchOrder, p->d_slotOf, L);
                WO_HIP(hipStreamSynchronize(s));
                p->patchVersion = p->flood.staticVersion;
                p->lastPatchLaunches = 1;
                p->numPatches = (L + WO_PATCH - 1) / WO_PATCH;
            }
        } else {
            p->patchVersion = -1;
        }
    }
    const int32_t L = p->L;
    const int gridL = xcd_grid(L);              // index-order passes over the ascending land list
    if (hIters > 0 || tIters > 0) {
        // ocean cells keep these values through the whole call: both elevation buffers hold them, and the per-ocean-cell
        // constants of the land passes are written once
        WO_HIP(hipMemcpyAsync(p->d_e2, p->d_e, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, s));
        launch(p, FAM_MISC, k_erode_ocean_init, blocks_for(N, 4096), WO_BLOCK, p->fields());
    }
    clk.end();

    FloodRun floodRun;
    auto leftovers_so_far = [&]() {
        if (checkEveryPass || !p->d_pendingEver) return;
        if (read_count(p, p->d_pendingEver) != 0) {
            if (p->opt.floodTiming) {          // (diagnostic runs only) which of the two reasons: a splitter-sort bucket that did not fit, or a basin launch with leftovers
                int32_t h[8] = {0};
                WO_HIP(hipMemcpy(h, p->d_pendingEver, sizeof(h), hipMemcpyDeviceToHost));
                std::fprintf(stderr, "[erode] call runs again with checks: %d basin-solve tasks were left pending\n", h[0]);
            }
            throw RedoWithChecks{};
        }
    };
    auto flood = [&](double cs) {
        leftovers_so_far();                    // the host is about to read the field
        clk.begin("priority_flood");
        auto t0 = std::chrono::steady_clock::now();
        // inside the land-first mirror, with the mask and the flood's tables in place: the land heights straight from the mirrored field
        const bool landOnly = mir.on && p->landIdentity && !p->floodX.on && !p->opt.floodDevice && p->h_ocean_valid && p->flood.staticValid &&
                              p->flood.staticN == p->N && p->flood.L > 0 && flood_land_is_mirror_prefix(p);
        if (landOnly) flood_stage_land(p, cs, floodRun);
        else {
            mir.suspend();
            flood_stage(p, cs, floodRun);
            mir.resume();
        }
        floodHostMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        clk.end();
    };
    if (hIters > 0) flood(0.5);

    const bool glacial = gIters > 0 && gStrength > 0;
    if (glacial) launch(p, FAM_GLAC_INDEX, k_glac_index, gridN, WO_BLOCK, p->fields(), gStrength);
    const double gScale = gIters > 0 ? 1.0 / gIters : 0;
    const double gCarve = 0.02 * gScale, gConv = 0.01 * gScale, gDep = 0.005 * gScale, gFjord = 0.015 * gScale;
    const int32_t midIter = (int32_t)std::floor(total * 0.75 + 0.5);
    bool midDone = false;
    // (round 2's river-aligned patch lists, WO_RIVER_PATCHES, were measured and dropped: profiles/r02c_river_patch_experiment.txt)
    int64_t basinPasses = 0, basinLeftoverPasses = 0, carveActive = 0;

    const bool stageAll = p->opt.stageTimingAll;
    // (one composite iteration as a hipGraph — captured once per call, replayed 167 times — was built and measured in round 4: 499.5 ms per step against 384.1 with
    // plain launches, profiles/r04j_*: a graph launch costs more than the ~25 stream launches it replaces; removed in round 6)
    bool sortAfterFlood = false;
    for (int32_t iter = 0; iter < total; ++iter) {
        clk.on = true;
        if (!midDone && iter >= midIter) { midDone = true; flood(0.85); sortAfterFlood = true; }
        clk.on = stageAll || p->profiling || total <= 16 || iter % 8 == 0;
        const bool gNow = iter < gIters && glacial, hNow = iter < hIters;
        // WO_RELAXED_SORT_EVERY=K (relaxed mode, NOT the reference's semantics: SURVEY 7.3): landCells is re-sorted only every K-th
        // iteration; in between the passes run with a stale visiting order (still a consistent order: every pass compares ranks
        // pairwise, so the dataflow is well defined, it is just not the reference's).  Measured, never reported as parity.
        const bool relaxedFull = p->opt.relaxedFull;           // RELAXED MODE, not parity (kernels_impl.h): one sort per flood, affine solve, Jacobi carve
        const bool sortNow = relaxedFull ? (iter == 0 || sortAfterFlood) : (p->opt.relaxedSortEvery <= 1 || iter % p->opt.relaxedSortEvery == 0);
        if ((gNow || hNow) && sortNow) sortAfterFlood = false;
        if ((gNow || hNow) && sortNow) { clk.begin("sort"); sort_land_by_elevation(p); ++sorts; clk.end(); }

        if (gNow) {
            clk.begin("glacial");
            Fields F = p->fields();
            launch(p, FAM_ICE_RECV, k_ice_receivers, gridN, WO_BLOCK, F);
            // ice accumulation: one launch in which the last donor to arrive runs its receiver's task (k_ice_climb)
            launch(p, FAM_ICE_ROUND, k_ice_climb, gridL, WO_BLOCK, F, F.blocker); ++iceRounds;
            if (relaxedFull) {          // RELAXED: the carve from a snapshot of the heights, one sweep
                launch(p, FAM_CARVE_ROUND, k_carve_jacobi, gridL, WO_BLOCK, F, (const float*)p->d_e, p->d_e2, gCarve, gConv, gStrength);
                swap_elev(p);
                launch(p, FAM_MORAINE, k_moraine_fjord, gridN, WO_BLOCK, p->fields(), gDep, gFjord);
                clk.end();
            } else {
            launch(p, FAM_CARVE_SETUP, k_carve_setup_cells, gridN, WO_BLOCK, F);
            select_active_by_rank(p, F.arank, p->d_listB, p->d_counters + 3);     // the active tasks in landCells order
            int32_t activeTasks = 0;
            {   // dependency lists of the active tasks (once per glacial step)
                const int32_t active = read_count(p, p->d_counters + 3);
                activeTasks = active;
                carveActive += active;
                if ((int64_t)active > p->carveCap) {
                    dfree(p->d_carveDeps); dfree(p->d_carveDepCnt); dfree(p->d_carveDepPos); dfree(p->d_carveRecs); dfree(p->d_carveSlotDone); dfree(p->d_carveExpect);
                    p->carveCap = (int64_t)active + active / 4 + 1024;
                    p->d_carveDeps = dalloc<int32_t>((size_t)p->carveCap * WO_CARVE_DEPS);
                    p->d_carveDepCnt = dalloc<int32_t>((size_t)p->carveCap); p->d_carveDepPos = dalloc<int32_t>((size_t)p->carveCap);
                    p->d_carveRecs = dalloc<CarveRec>((size_t)p->carveCap); p->d_carveSlotDone = dalloc<int32_t>((size_t)p->carveCap); p->d_carveExpect = dalloc<CarveExpect>((size_t)p->carveCap);
                }
                F.carveDeps = p->d_carveDeps; F.carveDepCnt = p->d_carveDepCnt; F.carveDepPos = p->d_carveDepPos;
                // (the dependency lists — a two-hop walk per task — are for the rounds: the granule launch waits on the heights themselves and the lists are only
                // made if it leaves tasks to the rounds)
                if (active > 0)
                    launch(p, FAM_CARVE_SETUP, k_carve_records, blocks_for(active), WO_BLOCK, F, (const int32_t*)p->d_listB, (const int32_t*)(p->d_counters + 3), p->d_carveRecs, p->d_carveSlotDone, gCarve, gConv, gStrength,
                           (int32_t)0, (int32_t)1);
            }
            {
                int32_t* c = p->d_counters;
                const int32_t count = activeTasks;
                int64_t k = 1;
                // The one launch (k_carve_granules), then — for whatever it leaves — rounds over the static activation list (k_carve_round_static): every
                // launch covers all active tasks, a finished one leaves after one load, an open one issues its loads at once; the number of finished tasks
                // is read back after a burst.  (Round 2 also tried all rounds in ONE cooperative launch with a grid barrier: slower,
                // profiles/r02f_persistent_rounds_grid.txt; the done-word form of the one launch, k_carve_flow, and the rounds-only route were removed in round 6.)
                if (count > 0) {
                    const int32_t active = count;
                    int32_t* done = c + 4;
                    WO_HIP(hipMemsetAsync(done, 0, sizeof(int32_t), s));
                    const int grid = blocks_for(active);
                    // the depth of the carve DAG falls from one glacial iteration to the next (the ice smooths its bed), so the count of
                    // finished tasks is read back every 32 rounds (a read-back costs about as much as three empty rounds)
                    constexpr int burst = 32;
                    bool allDone = false;
                    {
                        // every task in one launch; the grid is what is certainly resident at once: the occupancy query's blocks per CU less one
                        // (the query is known to answer one too many near register-file edges)
                        static int flowBlocks = 0;
                        if (!flowBlocks) {
                            int perCu = 0, dev = 0; hipDeviceProp_t prop;
                            WO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, k_carve_granules, WO_BLOCK, 0));
                            WO_HIP(hipGetDevice(&dev)); WO_HIP(hipGetDeviceProperties(&prop, dev));
                            flowBlocks = std::max(1, std::min(perCu, 8) - 1) * prop.multiProcessorCount;
                        }
                        // hook carve_blocks=<n>: at most n workgroups, so that every thread takes many tasks in turn
                        const int blocksNow = p->opt.carveBlocks > 0 ? std::max(1, std::min(flowBlocks, p->opt.carveBlocks)) : flowBlocks;
                        const long long flowBudget = p->opt.carveBudgetMs * 100000ll;   // 100 MHz ticks
                        if (!p->d_carveG) p->d_carveG = dalloc<unsigned long long>((size_t)N);
                        launch(p, FAM_CARVE_SETUP, k_carve_expect, grid, WO_BLOCK, F, (const CarveRec*)p->d_carveRecs, (const int32_t*)(c + 3), p->d_carveExpect);
                        launch(p, FAM_CARVE_SETUP, k_carve_pack, blocks_for(N, 4096), WO_BLOCK, (const float*)F.e, p->d_carveG, N);
                        launch(p, FAM_CARVE_ROUND, k_carve_granules, std::min(grid, blocksNow), WO_BLOCK, F, (const CarveRec*)p->d_carveRecs, (const CarveExpect*)p->d_carveExpect, p->d_carveG,
                               p->d_carveSlotDone, (const int32_t*)(c + 3), done, flowBudget);
                        launch(p, FAM_CARVE_SETUP, k_carve_unpack, blocks_for(N, 4096), WO_BLOCK, (const unsigned long long*)p->d_carveG, F.e, N);
                        ++k;
                        const int32_t fin = read_count(p, done);
                        allDone = fin >= active;
                        if (!allDone) {          // the rounds want the dependency lists after all
                            ++carveFlowLeft;
                            launch(p, FAM_CARVE_SETUP, k_carve_deps, grid, WO_BLOCK, F, (const int32_t*)p->d_listB, (const int32_t*)(c + 3), p->d_carveSlot);
                            launch(p, FAM_CARVE_SETUP, k_carve_records, grid, WO_BLOCK, F, (const int32_t*)p->d_listB, (const int32_t*)(c + 3), p->d_carveRecs, p->d_carveSlotDone, gCarve, gConv, gStrength, (int32_t)1, (int32_t)0);
                        }
                    }
                    for (; !allDone;) {
                        for (int b = 0; b < burst; ++b, ++k)
                            launch(p, FAM_CARVE_ROUND, k_carve_round_static, grid, WO_BLOCK, F, (const CarveRec*)p->d_carveRecs, p->d_carveSlotDone, (const int32_t*)(c + 3), (int32_t)k, gCarve, gConv, gStrength, done);
                        if (read_count(p, done) >= active) break;
                        if (k > 4 * (int64_t)p->N + 1024) throw HipError{"carve rounds do not converge"};
                    }
                }
                carveRounds += k - 1;
            }
            launch(p, FAM_MORAINE, k_moraine_fjord, gridN, WO_BLOCK, F, gDep, gFjord);
            clk.end();
            }
        }

        if (hNow) {
            if (gNow && sortNow) { clk.begin("sort"); sort_land_by_elevation(p); ++sorts; clk.end(); }
            Fields F = p->fields();
            F.solveK = K; F.solveM = m; F.solveDt = dt;
            const bool basin = !relaxedFull;           // exact mode: the basin-local solve (basin.hip); relaxed mode: the affine recurrence below
            const bool slotIdentity = mir.on && p->mirror.h_mask.size() == (size_t)N;          // land-first mirror: a land cell's Morton slot is its id
            // the receivers pass also leaves the start state of the layout's component search
            if (basin) { basin_alloc(p); F.basinJ = p->d_basinJ; F.basinMslot = slotIdentity ? nullptr : p->d_slotOf; }
            // two-level accumulation (k_flow_tiles): needs the land cells to be the ids 0 .. L-1 in Morton order (land-first mirror); on the planet's
            // own cell order (WO_LAYOUT=index) the one-launch climb over all cells (k_flow_climb)
            const bool flowTiles = p->landIdentity;
            FlowTiles FT{};
            if (flowTiles) {
                if (!p->d_ftLr) {
                    p->d_ftLr = dalloc<int32_t>((size_t)N); p->d_ftParent = dalloc<int32_t>((size_t)N); p->d_ftExtCnt = dalloc<int32_t>((size_t)N); p->d_ftInflow = dalloc<uint32_t>((size_t)N); p->d_ftRootAcc = dalloc<unsigned long long>((size_t)N);
                    WO_HIP(hipMemsetAsync(p->d_ftInflow, 0, (size_t)N * 4, s)); WO_HIP(hipMemsetAsync(p->d_ftExtCnt, 0, (size_t)N * 4, s));
                }
                FT.lr = p->d_ftLr; FT.parent = p->d_ftParent; FT.rootAcc = p->d_ftRootAcc; FT.inflow = p->d_ftInflow; FT.extCnt = p->d_ftExtCnt;
            }
            int32_t* const donorCnt = flowTiles ? (int32_t*)nullptr : p->d_flowCnt;
            bool tilesFirstDone = false, linksDone = false;
            // unchecked basin pass: its one launch tags what it produces with a number no earlier pass of this planet used
            const bool passTagged = basin && !checkEveryPass;
            clk.begin("receivers");
            launch(p, FAM_RECEIVERS, k_receivers_flow_init, gridL, WO_BLOCK, F, donorCnt);        // + flow start state and donor counts
            clk.end();
            F.basinJ = nullptr;
            // basin-local solve (basin.hip): this pass's store order groups every drainage component with everything it depends on.
            // The layout needs the receivers only and touches none of the flow accumulation's arrays: it runs on the planet's side stream
            // beside the flow accumulation and the solve's setup waits for both.  (A third stream for the solve's event lists was measured in
            // round 3 and a side stream for the elevation sort in round 6 — profiles/r03bc_*, r06d_*: no faster, a launch of these sizes already
            // occupies the chip's workgroup slots and two side by side take turns; both removed.)
            if (basin) {
                ensure_side_stream(p);
                // two-level flow accumulation: its first kernel also shortens the layout's start state inside every tile (k_flow_tiles<false>:
                // J[c] <- an ancestor at most a tile away), so the layout's component search starts after it, from chains of tiles instead of cells
                const bool tilesFeedLayout = flowTiles && slotIdentity;
                if (tilesFeedLayout) {
                    FT.basinJ = p->d_basinJ;
                    launch(p, FAM_FLOW_TILES, k_flow_tiles<false>, (int)(((int64_t)L + FT_CELLS - 1) / FT_CELLS), FT_THREADS, F, FT);
                    tilesFirstDone = true;
                    // (the root links too before the fork: beside the layout's first kernel — high-priority stream — they took 33 us instead of 15; flow stage 45.9 -> 43.8 ms per step, profiles/r05h_*)
                    launch(p, FAM_FLOW_TILES, k_flow_root_links, blocks_for(L, 4096), WO_BLOCK, F, FT);
                    linksDone = true;
                }
                WO_HIP(hipEventRecord(p->evFork, s));
                WO_HIP(hipStreamWaitEvent(p->side, p->evFork, 0));
                p->onSide = true;
                try { basin_layout(p, slotIdentity); } catch (...) { p->onSide = false; throw; }
                p->onSide = false;
                WO_HIP(hipEventRecord(p->evJoin, p->side));
            }
            clk.begin("flow");
            // Flow accumulation = subtree sizes of the forward forest (integers: any order of the additions is exact): in two levels under the mirror
            // (kernels_impl.h: k_flow_tiles), else one launch in which every leaf hands its total to its receiver and the thread that completes a
            // receiver carries on with it (k_flow_climb).  Every cell with a forward receiver is retired either way; k_flow_final reads the packed totals.
            // (10 M cells, flow stage per step: round 2's rake rounds + pointer doubling 108 ms, the climb 45, two levels 29: profiles/r02r_*, r05g_*.)
            if (flowTiles) {
                const int tiles = (int)(((int64_t)L + FT_CELLS - 1) / FT_CELLS);
                if (!tilesFirstDone) launch(p, FAM_FLOW_TILES, k_flow_tiles<false>, tiles, FT_THREADS, F, FT);
                if (!linksDone) launch(p, FAM_FLOW_TILES, k_flow_root_links, blocks_for(L, 4096), WO_BLOCK, F, FT);
                launch(p, FAM_FLOW_TILES, k_flow_root_climb, blocks_for(L, 4096), WO_BLOCK, F, FT);
                launch(p, FAM_FLOW_TILES, k_flow_tiles<true>, tiles, FT_THREADS, F, FT);
            } else
                launch(p, FAM_FLOW_SNAP, k_flow_climb, gridL, WO_BLOCK, F, (const int32_t*)p->d_flowCnt, (int32_t)0x7fffffff);
            {
                Fields Ff = F;
                if (relaxedFull) Ff.ev = nullptr;                // (no event lists: the relaxed solve has no order)
                // the solve's outputs are cleared (tags 0) only for a pass whose result is checked on the spot (k_solve_patch / k_solve_final read
                // the tags as launch numbers); the unchecked pass stamps them with a tag of its own instead (passTag below): 16 B per land cell less to write
                launch(p, FAM_FLOW_FINAL, k_flow_final, gridL, WO_BLOCK, Ff, donorCnt, (basin && !passTagged) ? p->d_out : (SolveOut*)nullptr);
            }
            clk.end();
            clk.begin("solve");
            if (relaxedFull) {
                // RELAXED: h' = a + b h'(receiver) composed by pointer jumping (12 doublings cover chains of 4 096 cells), then heights + deposits in one sweep
                if (!p->d_affine[0]) { p->d_affine[0] = dalloc<Affine>((size_t)N); p->d_affine[1] = dalloc<Affine>((size_t)N); }
                launch(p, FAM_SOLVE_SETUP, k_affine_init, gridL, WO_BLOCK, F, p->d_affine[0]);
                int cur = 0;
                for (int q = 0; q < 12; ++q, cur ^= 1) launch(p, FAM_SOLVE_ROUND, k_affine_jump, gridL, WO_BLOCK, F, (const Affine*)p->d_affine[cur], p->d_affine[cur ^ 1]);
                launch(p, FAM_SOLVE_FINAL, k_affine_apply, gridL, WO_BLOCK, F, (const Affine*)p->d_affine[cur], p->d_e2);
                swap_elev(p);
                clk.end();
            } else {
                WO_HIP(hipStreamWaitEvent(s, p->evJoin, 0));
                F.slotOf = p->d_basinSlot;
                F.solveLean = 1;
                // (the outputs' tags were cleared by k_flow_final: one coalesced sweep instead of one scattered 16-byte write per task)
                // the solve launch writes the final heights itself (SolveTask finality flags) when its result is not looked at pass by pass and
                // no k_solve_final then
                const bool solveFinals = !checkEveryPass;
                F.solveFinals = solveFinals ? 1 : 0;
                // (the setup launch also clears the counters of the solve launch: run_solve_patches' countersCleared)
                launch(p, FAM_SOLVE_SETUP, k_solve_setup_batched<true>, gridL, WO_BLOCK, F, p->d_patchPending, (int32_t)p->numPatches, p->d_patchTotals, (int32_t)WO_PATCH_TOTAL_SLOTS);
                int32_t passTag = 1;
                if (passTagged) {
                    if (p->solvePassSerial >= 0x3ff00000) { WO_HIP(hipMemsetAsync(p->d_out, 0, (size_t)N * sizeof(SolveOut), s)); p->solvePassSerial = 0; }
                    passTag = (1 << 20) + (int32_t)(++p->solvePassSerial);
                }
                const int64_t r = run_solve_patches(p, F, K, m, dt, true, true, !checkEveryPass, passTag);
                ++basinPasses; if (r > 1) ++basinLeftoverPasses;
                patchLaunches += r; maxSolve = std::max(maxSolve, r);
                if (!solveFinals) launch(p, FAM_SOLVE_FINAL, k_solve_final, gridL, WO_BLOCK, F, p->d_e2, (iter < tIters) ? p->d_me : (float*)nullptr);
                swap_elev(p);
                clk.end();
            }
        }

        if (iter < tIters) {
            clk.begin("thermal");
            Fields F = p->fields();
            if (!hNow) launch(p, FAM_THERMAL_EXCESS, k_masked_elev, gridL, WO_BLOCK, F);      // else written by k_solve_final
            launch(p, FAM_THERMAL_EXCESS, k_thermal_excess, gridL, WO_BLOCK, F, talus);
            if (p->maxDeg <= 12)
                launch(p, FAM_THERMAL_APPLY, k_thermal_apply_reg<12>, gridL, WO_BLOCK, F, p->d_e2, talus, kThermal);
            else if (p->maxDeg <= 16)
                launch(p, FAM_THERMAL_APPLY, k_thermal_apply_reg<16>, gridL, WO_BLOCK, F, p->d_e2, talus, kThermal);
            else
                launch_shmem(p, FAM_THERMAL_APPLY, k_thermal_apply, gridL, WO_BLOCK, (size_t)p->maxDeg * WO_BLOCK * 12, F, p->d_e2, talus, kThermal,
                             (int32_t)p->maxDeg);
            swap_elev(p);
            clk.end();
        }
    }
    clk.on = true;
    leftovers_so_far();
    if (glacial) {
        clk.begin("glacial_blend");
        launch(p, FAM_GLAC_BLEND, k_glacial_blend, gridN, WO_BLOCK, p->fields(), (const float*)p->d_e, p->d_e2);
        swap_elev(p);
        clk.end();
    }
    const bool mirrored = mir.on;
    if (mir.on) { clk.begin("setup"); mir.finish(); clk.end(); }
    clk.finish();
    p->erodeStats = {{"land_cells", (double)L}, {"mirror_layout", mirrored ? 1.0 : 0.0}, {"iterations", (double)total}, {"sorts", (double)sorts},
                     {"solve_launches_max_per_pass", (double)maxSolve}, {"solve_patch_launches_total", (double)patchLaunches},
                     {"solve_basin_passes", (double)basinPasses}, {"solve_basin_passes_with_leftovers", (double)basinLeftoverPasses},
                     {"flow_two_level", (p->d_ftLr && p->landIdentity && hIters > 0) ? 1.0 : 0.0}, {"ice_rounds_total", (double)iceRounds},
                     {"carve_rounds_total", (double)carveRounds}, {"carve_active_total", (double)carveActive}, {"carve_flow_launches_with_leftovers", (double)carveFlowLeft}, {"solve_check_every_pass", checkEveryPass ? 1.0 : 0.0}, {"calls_run_again_with_checks", (double)p->redoCalls}, {"flood_stage_ms", floodHostMs},
                     {"flood_device_pass1_ms", floodRun.deviceMs}, {"flood_device_rounds", (double)floodRun.rounds}, {"flood_device_epochs", (double)floodRun.epochs},
                     {"flood_device_evaluations", (double)floodRun.evals}, {"flood_equal_key_decisions", (double)floodRun.ties},
                     {"flood_pass1_on_host", (floodRun.usedDevice && !floodRun.fellBack) ? 0.0 : 1.0},
                     {"flood_host_calls", (double)floodRun.host.calls}, {"flood_host_serial_pass1", (double)floodRun.host.serialPass1},
                     {"flood_host_tie_groups", (double)floodRun.host.tieGroups}, {"flood_host_contested", (double)floodRun.host.contested},
                     {"flood_host_open_parents", (double)floodRun.host.openParents}, {"flood_host_unresolved", (double)floodRun.host.unresolved},
                     {"flood_host_path_redo", (double)floodRun.host.pathRedo}, {"flood_host_replays", (double)floodRun.host.replays}, {"flood_host_replayed_landmasses", (double)floodRun.host.replayedLandmasses}, {"flood_host_pass1_ms", floodRun.host.pass1Ms},
                     {"flood_host_pass23_ms", floodRun.host.pass23Ms},
                     {"relaxed_sort_every", (double)p->opt.relaxedSortEvery}, {"relaxed_full", p->opt.relaxedFull ? 1.0 : 0.0}, {"flood_exchange_calls", (double)p->floodX.calls}, {"flood_exchange_gathers", (double)p->floodX.gathers}, {"flood_exchange_whole_planet_floods", (double)p->floodX.globalFloods}, {"flood_exchange_received", (double)p->floodX.received}};
}

static void jacobi(wo_planet* p, int kind, int32_t iterations, double strength) {
    if (iterations <= 0) return;
    MirrorScope mir(p);                     // Jacobi passes read rows and neighbour values only: any naming of the cells gives the same field
    if (iterations >= 2) mir.enter();       // (moving the field in and out costs about one pass)
    coast_flags(p);
    const int gridN = xcd_grid(p->N);
    hipStream_t s = p->ctx->stream;
    if (kind == 1) WO_HIP(hipMemcpyAsync(p->d_orig, p->d_e, (size_t)p->N * sizeof(float), hipMemcpyDeviceToDevice, s));
    for (int32_t it = 0; it < iterations; ++it) {
        Fields F = p->fields();
        if (kind == 0) launch(p, FAM_SMOOTH, k_smooth, gridN, WO_BLOCK, F, (const float*)p->d_e, p->d_e2, strength);
        else if (kind == 1) launch(p, FAM_SHARPEN, k_sharpen, gridN, WO_BLOCK, F, (const float*)p->d_e, (const float*)p->d_orig, p->d_e2, strength);
        else launch(p, FAM_CREEP, k_creep, gridN, WO_BLOCK, F, (const float*)p->d_e, p->d_e2, strength);
        swap_elev(p);
    }
    mir.finish();
}

static void upload_tables(wo_planet* p, double seed) {
    uint8_t t[1024];
    noise_tables(seed, t, t + 512);
    WO_HIP(hipMemcpyAsync(p->d_tables, t, 1024, hipMemcpyHostToDevice, p->ctx->stream));
    WO_HIP(hipStreamSynchronize(p->ctx->stream));     // t is a stack buffer
}

static void warp(wo_planet* p, double seed, double strength, bool useHot) {
    if (!(strength > 0)) return;          // js/terrain-post.js:234
    upload_tables(p, seed + 9999);
    const double maxAmp = 0.12 * strength, bias = 0.25 + 0.5 * strength;
    // the greedy walk reads rows and positions of cells along its path, nothing that depends on a cell's name: it runs on the
    // patch-major mirror as well (10 M cells: 15 -> 5 ms)
    MirrorScope mir(p);
    mir.enter();
    const float* hot = useHot ? (const float*)p->d_hot : (const float*)nullptr;
    if (mir.on && useHot) {
        auto& M = p->mirror;
        if (!M.hot) M.hot = dalloc<float>(p->N);
        launch(p, FAM_MISC, k_mirror_gather_f32, blocks_for(p->N, 4096), WO_BLOCK, (const float*)p->d_hot, (const int32_t*)M.perm, M.hot, p->N);
        hot = M.hot;
    }
    launch(p, FAM_WARP, k_warp, xcd_grid(p->N), WO_BLOCK, p->fields(), (const uint8_t*)p->d_tables, (const float*)p->d_e, p->d_e2,
           maxAmp, bias, hot);
    swap_elev(p);
    mir.finish();
}

}  // namespace wo

namespace wo {
// device copies of per-region host arrays, freed on scope exit (also when a HIP call throws)
struct DevBufs {
    std::vector<void*> bufs;
    ~DevBufs() { for (void* b : bufs) (void)hipFree(b); }
    template <class T> T* up(const T* host, size_t n, hipStream_t s) {
        if (!host) return nullptr;
        void* d = nullptr; WO_HIP(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T))); bufs.push_back(d);
        WO_HIP(hipMemcpyAsync(d, host, n * sizeof(T), hipMemcpyHostToDevice, s));
        return (T*)d;
    }
    template <class T> T* alloc(size_t n) { void* d = nullptr; WO_HIP(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T))); bufs.push_back(d); return (T*)d; }
};
ClimateMesh climate_mesh(const wo_planet* p) { ClimateMesh M; M.N = p->N; M.off = p->d_off; M.adj = p->d_adj; M.xyz = p->d_xyz; return M; }
}  // namespace wo


// =====================================================================================================
// C ABI
// =====================================================================================================
using namespace wo;

#define WO_TRY try {
#define WO_CATCH(fn)                                                                   \
    } catch (const HipError& e) { set_error(std::string(fn) + ": " + e.msg); return 2; } \
      catch (const std::exception& e) { set_error(std::string(fn) + ": " + e.what()); return 3; }

Options Options::from_env() {
    auto str = [](const char* n) { const char* v = std::getenv(n); return std::string(v ? v : ""); };
    auto on = [](const char* n) { const char* v = std::getenv(n); return v && std::atoi(v) != 0; };
    auto set = [](const char* n) { return std::getenv(n) != nullptr; };
    Options o;
    o.layoutIndex = str("WO_LAYOUT") == "index";
    o.tileLds = on("WO_TILE_LDS");
    o.floodDevice = str("WO_FLOOD") == "device";
    o.floodTiming = set("WO_FLOOD_TIMING");
    o.stageTimingAll = str("WO_STAGE_TIMING") == "all";
    if (set("WO_RELAXED_SORT_EVERY")) o.relaxedSortEvery = std::max(1, std::atoi(std::getenv("WO_RELAXED_SORT_EVERY")));
    o.relaxedFull = str("WO_RELAXED") == "full";
    o.basinScramble = test_hook_int("basin_scramble", 0) != 0;
    o.carveBudgetMs = test_hook_int("carve_budget_ms", 200);
    o.carveBlocks = (int)std::max<long long>(0, test_hook_int("carve_blocks", 0));
    return o;
}

static bool check_planet(wo_planet* p, const char* fn) {
    if (!p) { set_error(std::string(fn) + ": null planet handle"); return false; }
    p->opt = Options::from_env();
    hipError_t e = hipSetDevice(p->ctx->device);
    if (e != hipSuccess) { set_error(std::string(fn) + ": hipSetDevice failed: " + hipGetErrorString(e)); return false; }
    return true;
}

// The host used to look at the basin solve's pending count after every pass (a read-back and ~30 us of idle GPU per iteration, and
// the host never got ahead of the device).  No real layout has ever left a task pending, so the look is deferred: the launches add
// into one word per call, the word is read where the host synchronises anyway (before the second flood, at the end), and if it is
// not zero the field is restored from a copy taken at entry and the call runs again with the check after every pass — the form
// that finishes pending tasks with k_solve_patch launches.  WO_SOLVE_CHECK=pass: always that form.
static void erode_composite_checked(wo_planet* p, int32_t hIters, double K, double m, double dt, int32_t tIters, double talus,
                                    double kThermal, int32_t gIters, double gStrength) {
    if (hIters <= 0) { erode_composite(p, hIters, K, m, dt, tIters, talus, kThermal, gIters, gStrength, true); return; }
    // With a flood exchange set (shares of one planet) every flood call of this rank is a pair of collectives with its peers: a rank that
    // ran the call a second time would repeat them alone (its peers are past them and no longer hold that call's heights).  Such a call is
    // therefore checked pass by pass from the start — pending tasks are finished where they arise and nothing is ever run again.
    if (p->floodX.on) { erode_composite(p, hIters, K, m, dt, tIters, talus, kThermal, gIters, gStrength, true); return; }
    hipStream_t s = p->ctx->stream;
    if (!p->d_redoE) p->d_redoE = dalloc<float>((size_t)p->N);
    if (!p->d_pendingEver) p->d_pendingEver = dalloc<int32_t>(16);
    WO_HIP(hipMemcpyAsync(p->d_redoE, p->d_e, (size_t)p->N * sizeof(float), hipMemcpyDeviceToDevice, s));
    WO_HIP(hipMemsetAsync(p->d_pendingEver, 0, 16 * sizeof(int32_t), s));
    try {
        erode_composite(p, hIters, K, m, dt, tIters, talus, kThermal, gIters, gStrength, false);
    } catch (const RedoWithChecks&) {
        WO_HIP(hipStreamSynchronize(s));
        if (p->side) WO_HIP(hipStreamSynchronize(p->side));
            WO_HIP(hipMemcpyAsync(p->d_e, p->d_redoE, (size_t)p->N * sizeof(float), hipMemcpyDeviceToDevice, s));
        ++p->redoCalls;
        erode_composite(p, hIters, K, m, dt, tIters, talus, kThermal, gIters, gStrength, true);
    }
}

extern "C" {

int wo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

wo_ctx* wo_ctx_create(int32_t device) {
    try {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
            set_error("wo_ctx_create: no usable HIP device (libworogen has no CPU fallback)");
            return nullptr;
        }
        if (device < 0 || device >= n) { set_error("wo_ctx_create: device index out of range"); return nullptr; }
        auto* c = new wo_ctx();
        c->device = device;
        WO_HIP(hipSetDevice(device));
        WO_HIP(hipGetDeviceProperties(&c->prop, device));
        WO_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        return c;
    } catch (const HipError& e) { set_error(std::string("wo_ctx_create: ") + e.msg); return nullptr; }
}

void wo_ctx_destroy(wo_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

wo_planet* wo_planet_create(wo_ctx* ctx, int32_t numRegions, const int32_t* adjOffset, const int32_t* adjList,
                            const float* r_xyz, const float* neighborDist) {
    if (!ctx || numRegions < 1 || !adjOffset || !adjList || !r_xyz) { set_error("wo_planet_create: bad arguments"); return nullptr; }
    wo_planet* p = nullptr;
    try {
        WO_HIP(hipSetDevice(ctx->device));
        const int32_t N = numRegions, E = adjOffset[N];
        int32_t maxDeg = 0;
        if (adjOffset[0] != 0) { set_error("wo_planet_create: adjOffset[0] != 0"); return nullptr; }
        for (int32_t r = 0; r < N; ++r) {
            const int32_t d = adjOffset[r + 1] - adjOffset[r];
            if (d < 0) { set_error("wo_planet_create: adjOffset is not monotone"); return nullptr; }
            maxDeg = std::max(maxDeg, d);
        }
        if (maxDeg > WO_MAX_DEG) { set_error("wo_planet_create: vertex degree " + std::to_string(maxDeg) + " exceeds the supported maximum " + std::to_string(WO_MAX_DEG)); return nullptr; }
        for (int32_t i = 0; i < E; ++i) if (adjList[i] < 0 || adjList[i] >= N) { set_error("wo_planet_create: adjList entry out of range"); return nullptr; }
        {   // simple graph: the order-exact parallel passes rely on a cell's neighbours being distinct cells other than itself
            std::atomic<int> bad{0};
            parallel_ranges(N, [&](int64_t b, int64_t e, int) {
                for (int64_t r = b; r < e && !bad.load(std::memory_order_relaxed); ++r)
                    for (int32_t i = adjOffset[r]; i < adjOffset[r + 1]; ++i) {
                        if (adjList[i] == r) { bad = 1; break; }
                        for (int32_t j = adjOffset[r]; j < i; ++j) if (adjList[j] == adjList[i]) { bad = 2; break; }
                    }
            });
            if (bad) { set_error(bad == 1 ? "wo_planet_create: a cell lists itself as a neighbour" : "wo_planet_create: a cell lists the same neighbour twice"); return nullptr; }
        }
        p = new wo_planet();
        p->ctx = ctx; p->N = N; p->E = E; p->maxDeg = maxDeg;
        p->h_off.assign(adjOffset, adjOffset + N + 1);
        p->h_adj.assign(adjList, adjList + E);
        p->h_xyz.assign(r_xyz, r_xyz + 3 * (size_t)N);
        hipStream_t s = ctx->stream;
        p->d_off = dalloc<int32_t>(N + 1); p->d_adj = dalloc<int32_t>((size_t)E + WO_ROW); p->d_dist = dalloc<float>((size_t)E + WO_ROW); p->d_xyz = dalloc<float>(3 * (size_t)N);   // + WO_ROW: load_row reads whole 16-byte pieces
        WO_HIP(hipMemsetAsync(p->d_adj + E, 0, WO_ROW * sizeof(int32_t), s)); WO_HIP(hipMemsetAsync(p->d_dist + E, 0, WO_ROW * sizeof(float), s));
        p->d_e = dalloc<float>(N); p->d_e2 = dalloc<float>(N); p->d_hot = dalloc<float>(N); p->d_orig = dalloc<float>(N);
        p->d_ocean = dalloc<uint8_t>(N); p->d_coast = dalloc<uint8_t>(N); p->d_tables = dalloc<uint8_t>(1024);
        WO_HIP(hipHostMalloc((void**)&p->h_pinned, std::max<size_t>((size_t)N * sizeof(float), 64)));
        WO_HIP(hipHostMalloc((void**)&p->h_count, 64));
        WO_HIP(hipMemcpyAsync(p->d_off, adjOffset, (size_t)(N + 1) * 4, hipMemcpyHostToDevice, s));
        WO_HIP(hipMemcpyAsync(p->d_adj, adjList, (size_t)E * 4, hipMemcpyHostToDevice, s));
        WO_HIP(hipMemcpyAsync(p->d_xyz, r_xyz, (size_t)N * 12, hipMemcpyHostToDevice, s));
        if (neighborDist) {
            WO_HIP(hipMemcpyAsync(p->d_dist, neighborDist, (size_t)E * 4, hipMemcpyHostToDevice, s));
        } else {
            std::vector<float> nd(E);
            neighbor_dist(N, adjOffset, adjList, r_xyz, nd.data());
            WO_HIP(hipMemcpyAsync(p->d_dist, nd.data(), (size_t)E * 4, hipMemcpyHostToDevice, s));
            WO_HIP(hipStreamSynchronize(s));
        }
        WO_HIP(hipMemsetAsync(p->d_e, 0, (size_t)N * 4, s));
        WO_HIP(hipMemsetAsync(p->d_ocean, 0, (size_t)N, s));
        WO_HIP(hipEventCreate(&p->evStart)); WO_HIP(hipEventCreate(&p->evStop));
        WO_HIP(hipStreamSynchronize(s));
        return p;
    } catch (const HipError& e) {
        set_error(std::string("wo_planet_create: ") + e.msg);
        if (p) wo_planet_destroy(p);
        return nullptr;
    }
}

void wo_planet_destroy(wo_planet* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    dfree(p->d_off); dfree(p->d_adj); dfree(p->d_dist); dfree(p->d_xyz); dfree(p->d_e); dfree(p->d_e2); dfree(p->d_hot); dfree(p->d_orig);
    flood_gpu_free(p->fgpu);
    if (p->floodLink && p->floodLinkFree) p->floodLinkFree(p->floodLink);
    p->floodLink = nullptr;
    basin_free(p);
    if (p->side) { (void)hipStreamSynchronize(p->side); (void)hipStreamDestroy(p->side); p->side = nullptr; }
    if (p->evFork) { (void)hipEventDestroy(p->evFork); p->evFork = nullptr; }
    if (p->evJoin) { (void)hipEventDestroy(p->evJoin); p->evJoin = nullptr; }
    mirror_free(p);
    dfree(p->d_ocean); dfree(p->d_coast); dfree(p->d_tables); dfree(p->d_savedE); dfree(p->d_savedOcean);
    if (p->d_oceanKnown) { (void)hipFree(p->d_oceanKnown); p->d_oceanKnown = nullptr; } if (p->d_maskDiff) { (void)hipFree(p->d_maskDiff); p->d_maskDiff = nullptr; }
    dfree(p->d_landInit); dfree(p->d_landIdx); dfree(p->d_land[0]); dfree(p->d_land[1]); dfree(p->d_keys[0]); dfree(p->d_keys[1]); dfree(p->d_rank);
    dfree(p->d_cellDist); dfree(p->d_flow); dfree(p->d_task); dfree(p->d_out); dfree(p->d_flowCnt); dfree(p->d_tr); dfree(p->d_ev); dfree(p->d_me); dfree(p->d_haloSend); dfree(p->d_haloRecv); dfree(p->d_haloBuf); if (p->h_haloBuf) { (void)hipHostFree(p->h_haloBuf); p->h_haloBuf = nullptr; } dfree(p->d_carveSlot); dfree(p->d_redoE); dfree(p->d_pendingEver); for (auto& r : p->d_rs) { if (r) (void)hipFree(r); r = nullptr; } dfree(p->d_carveG); dfree(p->d_carveExpect); dfree(p->d_carveRecs); dfree(p->d_carveSlotDone); dfree(p->d_carveDeps); dfree(p->d_carveDepCnt); dfree(p->d_carveDepPos); dfree(p->d_ftLr); dfree(p->d_ftParent); dfree(p->d_affine[0]); dfree(p->d_affine[1]); dfree(p->d_ftExtCnt); dfree(p->d_ftInflow); dfree(p->d_ftRootAcc); dfree(p->d_accCnt); dfree(p->d_jump); dfree(p->d_nj);
    dfree(p->d_doneAt); dfree(p->d_totalExcess);
    dfree(p->d_glac); dfree(p->d_iceFlow); dfree(p->d_iceTarget); dfree(p->d_arank); dfree(p->d_iceUp);
    dfree(p->d_patchOrder); dfree(p->d_slotOf); dfree(p->d_patchPending); dfree(p->d_patchTotals); dfree(p->d_patchBlk);
    dfree(p->d_listA); dfree(p->d_listB); dfree(p->d_counters);
    if (p->h_patchTotals) (void)hipHostFree(p->h_patchTotals);
    if (p->d_sortTemp) (void)hipFree(p->d_sortTemp);
    if (p->h_pinned) (void)hipHostFree(p->h_pinned);
    if (p->h_count) (void)hipHostFree(p->h_count);
    if (p->h_word) { (void)hipHostFree(p->h_word); p->h_word = nullptr; }
    for (auto& pe : p->pending) { (void)hipEventDestroy(pe.a); (void)hipEventDestroy(pe.b); }
    for (auto& b : p->stageBrackets) { (void)hipEventDestroy(b.a); (void)hipEventDestroy(b.b); }
    for (auto e : p->eventPool) (void)hipEventDestroy(e);
    if (p->evStart) (void)hipEventDestroy(p->evStart);
    if (p->evStop) (void)hipEventDestroy(p->evStop);
    delete p;
}

int wo_planet_upload(wo_planet* p, const float* r_elevation, const uint8_t* r_isOcean) {
    if (!check_planet(p, "wo_planet_upload")) return 1;
    WO_TRY
    hipStream_t s = p->ctx->stream;
    if (r_elevation) WO_HIP(hipMemcpyAsync(p->d_e, r_elevation, (size_t)p->N * 4, hipMemcpyHostToDevice, s));
    if (r_isOcean) {
        WO_HIP(hipMemcpyAsync(p->d_ocean, r_isOcean, (size_t)p->N, hipMemcpyHostToDevice, s));
        if (p->h_ocean.size() != (size_t)p->N || std::memcmp(p->h_ocean.data(), r_isOcean, (size_t)p->N) != 0) {
            p->h_ocean.assign(r_isOcean, r_isOcean + p->N);
            p->flood.staticValid = false; ++p->oceanVersion;
            p->oceanKnownValid = false;                    // (the device-side copy of the known mask is brought up to date by the next download)
        }
        p->h_ocean_valid = true;
    }
    WO_HIP(hipStreamSynchronize(s));
    return 0;
    WO_CATCH("wo_planet_upload")
}

// ---- halo lists of a band-decomposed Jacobi pass (planet = one band + its one-ring halo, see banded.py) ----
int wo_planet_set_halo(wo_planet* p, const int32_t* sendIdx, int32_t nSend, const int32_t* recvIdx, int32_t nRecv) {
    if (!check_planet(p, "wo_planet_set_halo")) return 1;
    if (nSend < 0 || nRecv < 0 || (nSend && !sendIdx) || (nRecv && !recvIdx)) { set_error("wo_planet_set_halo: bad arguments"); return 1; }
    for (int32_t i = 0; i < nSend; ++i) if (sendIdx[i] < 0 || sendIdx[i] >= p->N) { set_error("wo_planet_set_halo: send index out of range"); return 1; }
    for (int32_t i = 0; i < nRecv; ++i) if (recvIdx[i] < 0 || recvIdx[i] >= p->N) { set_error("wo_planet_set_halo: receive index out of range"); return 1; }
    WO_TRY
    hipStream_t s = p->ctx->stream;
    dfree(p->d_haloSend); dfree(p->d_haloRecv); dfree(p->d_haloBuf);
    if (p->h_haloBuf) { (void)hipHostFree(p->h_haloBuf); p->h_haloBuf = nullptr; }
    p->nHaloSend = nSend; p->nHaloRecv = nRecv;
    const size_t m = (size_t)std::max(nSend, nRecv);
    p->d_haloSend = dalloc<int32_t>(nSend); p->d_haloRecv = dalloc<int32_t>(nRecv); p->d_haloBuf = dalloc<float>(m);
    WO_HIP(hipHostMalloc((void**)&p->h_haloBuf, std::max<size_t>(m * sizeof(float), 64)));
    if (nSend) WO_HIP(hipMemcpyAsync(p->d_haloSend, sendIdx, (size_t)nSend * 4, hipMemcpyHostToDevice, s));
    if (nRecv) WO_HIP(hipMemcpyAsync(p->d_haloRecv, recvIdx, (size_t)nRecv * 4, hipMemcpyHostToDevice, s));
    WO_HIP(hipStreamSynchronize(s));
    return 0;
    WO_CATCH("wo_planet_set_halo")
}
// ---- landmass decomposition: the shares' flood exchange (flood_host.cc: flood_host_passes_exchange) ----
int wo_planet_set_flood_exchange(wo_planet* p, const uint8_t* trueOcean, wo_flood_exchange_fn fn, void* user) {
    if (!check_planet(p, "wo_planet_set_flood_exchange")) return 1;
    try {
        if (p->floodLink && p->floodLinkFree) { p->floodLinkFree(p->floodLink); }
        p->floodLink = nullptr; p->floodLinkFree = nullptr;
        FloodExchange& X = p->floodX;
        if (!fn) { X.on = false; X.fn = nullptr; X.user = nullptr; X.trueOcean.clear(); X.global = FloodScratch{}; return 0; }
        if (!trueOcean) { set_error("wo_planet_set_flood_exchange: the planet's true ocean mask is required"); return 1; }
        const bool same = X.trueOcean.size() == (size_t)p->N && std::memcmp(X.trueOcean.data(), trueOcean, (size_t)p->N) == 0;
        if (!same) { X.trueOcean.assign(trueOcean, trueOcean + p->N); X.global.staticValid = false; }
        // protocol handshake (phase -1): the callback must know THIS protocol (phases 0-3; round 4's had 0-1 only, and a callback written for it that
        // treats every phase != 0 as the all-gather would gather over the wrong buffer on phases 2 / 3 and report success)
        int32_t hello = WO_FLOOD_EXCHANGE_PROTOCOL;
        const int hrc = fn(user, -1, &hello, 1);
        if (hrc != 0 || hello != -WO_FLOOD_EXCHANGE_PROTOCOL) {
            X.on = false; X.fn = nullptr; X.user = nullptr;
            set_error("wo_planet_set_flood_exchange: the callback did not acknowledge exchange protocol " + std::to_string(WO_FLOOD_EXCHANGE_PROTOCOL) +
                      " (phase -1: negate buf[0] and return 0; return non-zero for any phase it does not implement)");
            return 1;
        }
        X.fn = fn; X.user = user; X.on = true;
        X.calls = X.gathers = X.globalFloods = X.received = 0; X.posVersion = -1; X.landTotal = -1;
        return 0;
    WO_CATCH("wo_planet_set_flood_exchange")
}

int wo_planet_pack_halo(wo_planet* p, float* hostOut, void* deviceOut) {
    if (!check_planet(p, "wo_planet_pack_halo")) return 1;
    if ((hostOut == nullptr) == (deviceOut == nullptr)) { set_error("wo_planet_pack_halo: pass exactly one of hostOut / deviceOut"); return 1; }
    WO_TRY
    hipStream_t s = p->ctx->stream;
    const int32_t n = p->nHaloSend;
    if (n > 0) {
        float* dst = deviceOut ? (float*)deviceOut : p->d_haloBuf;
        launch(p, FAM_MISC, k_halo_pack, blocks_for(n), WO_BLOCK, (const float*)p->d_e, (const int32_t*)p->d_haloSend, n, dst);
        if (hostOut) {
            WO_HIP(hipMemcpyAsync(p->h_haloBuf, p->d_haloBuf, (size_t)n * 4, hipMemcpyDeviceToHost, s));
            WO_HIP(hipStreamSynchronize(s));
            std::memcpy(hostOut, p->h_haloBuf, (size_t)n * 4);
            return 0;
        }
    }
    WO_HIP(hipStreamSynchronize(s));           // deviceOut is handed to another stream (RCCL) next
    return 0;
    WO_CATCH("wo_planet_pack_halo")
}
int wo_planet_unpack_halo(wo_planet* p, const float* hostIn, const void* deviceIn) {
    if (!check_planet(p, "wo_planet_unpack_halo")) return 1;
    if ((hostIn == nullptr) == (deviceIn == nullptr)) { set_error("wo_planet_unpack_halo: pass exactly one of hostIn / deviceIn"); return 1; }
    WO_TRY
    hipStream_t s = p->ctx->stream;
    const int32_t n = p->nHaloRecv;
    if (n > 0) {
        const float* src = (const float*)deviceIn;
        if (hostIn) {
            std::memcpy(p->h_haloBuf, hostIn, (size_t)n * 4);
            WO_HIP(hipMemcpyAsync(p->d_haloBuf, p->h_haloBuf, (size_t)n * 4, hipMemcpyHostToDevice, s));
            src = p->d_haloBuf;
        }
        launch(p, FAM_MISC, k_halo_unpack, blocks_for(n), WO_BLOCK, p->d_e, (const int32_t*)p->d_haloRecv, n, src);
        WO_HIP(hipStreamSynchronize(s));       // the caller may reuse its buffer; the next pass reads d_e on this stream anyway
    }
    return 0;
    WO_CATCH("wo_planet_unpack_halo")
}

int wo_planet_download(wo_planet* p, float* r_elevation) {
    if (!check_planet(p, "wo_planet_download") || !r_elevation) { if (p) set_error("wo_planet_download: null pointer"); return 1; }
    WO_TRY
    WO_HIP(hipMemcpyAsync(r_elevation, p->d_e, (size_t)p->N * 4, hipMemcpyDeviceToHost, p->ctx->stream));
    WO_HIP(hipStreamSynchronize(p->ctx->stream));
    return 0;
    WO_CATCH("wo_planet_download")
}

int wo_planet_ocean_from_elevation(wo_planet* p) {
    if (!check_planet(p, "wo_planet_ocean_from_elevation")) return 1;
    WO_TRY
    launch(p, FAM_OCEAN, k_ocean_from_elev, blocks_for(p->N), WO_BLOCK, (const float*)p->d_e, p->d_ocean, p->N);
    ocean_changed(p);
    return 0;
    WO_CATCH("wo_planet_ocean_from_elevation")
}

int wo_planet_download_ocean(wo_planet* p, uint8_t* r_isOcean) {
    if (!check_planet(p, "wo_planet_download_ocean") || !r_isOcean) { if (p) set_error("wo_planet_download_ocean: null pointer"); return 1; }
    WO_TRY
    WO_HIP(hipMemcpyAsync(r_isOcean, p->d_ocean, (size_t)p->N, hipMemcpyDeviceToHost, p->ctx->stream));
    WO_HIP(hipStreamSynchronize(p->ctx->stream));
    return 0;
    WO_CATCH("wo_planet_download_ocean")
}

int wo_planet_sync(wo_planet* p) {
    if (!check_planet(p, "wo_planet_sync")) return 1;
    WO_TRY
    WO_HIP(hipStreamSynchronize(p->ctx->stream));
    return 0;
    WO_CATCH("wo_planet_sync")
}

int wo_planet_save_state(wo_planet* p) {
    if (!check_planet(p, "wo_planet_save_state")) return 1;
    WO_TRY
    if (!p->d_savedE) { p->d_savedE = dalloc<float>(p->N); p->d_savedOcean = dalloc<uint8_t>(p->N); }
    WO_HIP(hipMemcpyAsync(p->d_savedE, p->d_e, (size_t)p->N * 4, hipMemcpyDeviceToDevice, p->ctx->stream));
    WO_HIP(hipMemcpyAsync(p->d_savedOcean, p->d_ocean, (size_t)p->N, hipMemcpyDeviceToDevice, p->ctx->stream));
    p->saved = true;
    return 0;
    WO_CATCH("wo_planet_save_state")
}

int wo_planet_restore_state(wo_planet* p) {
    if (!check_planet(p, "wo_planet_restore_state")) return 1;
    if (!p->saved) { set_error("wo_planet_restore_state: nothing saved"); return 1; }
    WO_TRY
    WO_HIP(hipMemcpyAsync(p->d_e, p->d_savedE, (size_t)p->N * 4, hipMemcpyDeviceToDevice, p->ctx->stream));
    WO_HIP(hipMemcpyAsync(p->d_ocean, p->d_savedOcean, (size_t)p->N, hipMemcpyDeviceToDevice, p->ctx->stream));
    ocean_changed(p);
    return 0;
    WO_CATCH("wo_planet_restore_state")
}

int wo_planet_upload_hotspot(wo_planet* p, const float* r_hotspot) {
    if (!check_planet(p, "wo_planet_upload_hotspot") || !r_hotspot) { if (p) set_error("wo_planet_upload_hotspot: null pointer"); return 1; }
    WO_TRY
    WO_HIP(hipMemcpyAsync(p->d_hot, r_hotspot, (size_t)p->N * 4, hipMemcpyHostToDevice, p->ctx->stream));
    WO_HIP(hipStreamSynchronize(p->ctx->stream));
    p->hot_valid = true;
    return 0;
    WO_CATCH("wo_planet_upload_hotspot")
}

int wo_warp_terrain_resident(wo_planet* p, double seed, double strength, int32_t useHotspot) {
    if (!check_planet(p, "wo_warp_terrain_resident")) return 1;
    if (useHotspot && !p->hot_valid) { set_error("wo_warp_terrain_resident: no hotspot field uploaded"); return 1; }
    WO_TRY
    warp(p, seed, strength, useHotspot != 0);
    return 0;
    WO_CATCH("wo_warp_terrain_resident")
}

int wo_smooth_elevation_resident(wo_planet* p, int32_t iterations, double strength) {
    if (!check_planet(p, "wo_smooth_elevation_resident")) return 1;
    WO_TRY jacobi(p, 0, iterations, strength); return 0; WO_CATCH("wo_smooth_elevation_resident")
}
int wo_sharpen_ridges_resident(wo_planet* p, int32_t iterations, double strength) {
    if (!check_planet(p, "wo_sharpen_ridges_resident")) return 1;
    WO_TRY jacobi(p, 1, iterations, strength); return 0; WO_CATCH("wo_sharpen_ridges_resident")
}
int wo_soil_creep_resident(wo_planet* p, int32_t iterations, double strength) {
    if (!check_planet(p, "wo_soil_creep_resident")) return 1;
    WO_TRY jacobi(p, 2, iterations, strength); return 0; WO_CATCH("wo_soil_creep_resident")
}

int wo_erode_composite_resident(wo_planet* p, int32_t hIters, double K, double m, double dt, int32_t tIters, double talusSlope,
                                double kThermal, int32_t gIters, double glacialStrength) {
    if (!check_planet(p, "wo_erode_composite_resident")) return 1;
    WO_TRY
    erode_composite_checked(p, hIters, K, m, dt, tIters, talusSlope, kThermal, gIters, glacialStrength);
    return 0;
    WO_CATCH("wo_erode_composite_resident")
}

int wo_planet_synthetic_terrain(wo_planet* p, double seed) {
    if (!check_planet(p, "wo_planet_synthetic_terrain")) return 1;
    WO_TRY
    upload_tables(p, seed);
    launch(p, FAM_SYNTH, k_synthetic, blocks_for(p->N), WO_BLOCK, (const uint8_t*)p->d_tables, (const float*)p->d_xyz, p->d_e, p->d_ocean, p->N);
    ocean_changed(p);
    return 0;
    WO_CATCH("wo_planet_synthetic_terrain")
}

// ---- JS call surface: host arrays in, mutated in place ----
static int with_host_field(wo_planet* p, const char* fn, float* e, const uint8_t* oc, bool needOcean, void (*body)(wo_planet*, void*), void* arg) {
    if (!check_planet(p, fn)) return 1;
    if (!e || (needOcean && !oc)) { set_error(std::string(fn) + ": null pointer"); return 1; }
    try {
        int rc = wo_planet_upload(p, e, oc);
        if (rc) return rc;
        body(p, arg);
        return wo_planet_download(p, e);
    } catch (const HipError& ex) { set_error(std::string(fn) + ": " + ex.msg); return 2; }
      catch (const std::exception& ex) { set_error(std::string(fn) + ": " + ex.what()); return 3; }
}

struct JacArgs { int kind; int32_t it; double s; };
struct WarpArgs { double seed, strength; bool hot; };
struct ErodeArgs { int32_t h; double K, m, dt; int32_t t; double talus, kT; int32_t g; double gs; };

int wo_warp_terrain(wo_planet* p, float* r_elevation, double seed, double strength, const float* r_hotspot) {
    if (p && r_hotspot) { int rc = wo_planet_upload_hotspot(p, r_hotspot); if (rc) return rc; }
    WarpArgs a{seed, strength, r_hotspot != nullptr};
    return with_host_field(p, "wo_warp_terrain", r_elevation, nullptr, false,
                           [](wo_planet* q, void* v) { auto* w = (WarpArgs*)v; warp(q, w->seed, w->strength, w->hot); }, &a);
}
int wo_smooth_elevation(wo_planet* p, float* e, const uint8_t* oc, int32_t it, double s) {
    JacArgs a{0, it, s};
    return with_host_field(p, "wo_smooth_elevation", e, oc, true, [](wo_planet* q, void* v) { auto* j = (JacArgs*)v; jacobi(q, j->kind, j->it, j->s); }, &a);
}
int wo_sharpen_ridges(wo_planet* p, float* e, const uint8_t* oc, int32_t it, double s) {
    JacArgs a{1, it, s};
    return with_host_field(p, "wo_sharpen_ridges", e, oc, true, [](wo_planet* q, void* v) { auto* j = (JacArgs*)v; jacobi(q, j->kind, j->it, j->s); }, &a);
}
int wo_soil_creep(wo_planet* p, float* e, const uint8_t* oc, int32_t it, double s) {
    JacArgs a{2, it, s};
    return with_host_field(p, "wo_soil_creep", e, oc, true, [](wo_planet* q, void* v) { auto* j = (JacArgs*)v; jacobi(q, j->kind, j->it, j->s); }, &a);
}
int wo_erode_composite(wo_planet* p, float* e, const uint8_t* oc, int32_t hIters, double K, double m, double dt, int32_t tIters,
                       double talusSlope, double kThermal, int32_t gIters, double glacialStrength) {
    ErodeArgs a{hIters, K, m, dt, tIters, talusSlope, kThermal, gIters, glacialStrength};
    return with_host_field(p, "wo_erode_composite", e, oc, true,
                           [](wo_planet* q, void* v) { auto* x = (ErodeArgs*)v; erode_composite_checked(q, x->h, x->K, x->m, x->dt, x->t, x->talus, x->kT, x->g, x->gs); }, &a);
}

// smoothField (js/climate-util.js:5-25) on a caller-owned field; the planet's resident elevation is not touched
int wo_smooth_field(wo_planet* p, float* field, int32_t passes) {
    if (!check_planet(p, "wo_smooth_field")) return 1;
    if (!field) { set_error("wo_smooth_field: null field"); return 1; }
    if (passes <= 0) return 0;
    float *a = nullptr, *b = nullptr;
    try {
        hipStream_t s = p->ctx->stream;
        const size_t bytes = (size_t)p->N * sizeof(float);
        a = dalloc<float>(p->N); b = dalloc<float>(p->N);
        WO_HIP(hipMemcpyAsync(a, field, bytes, hipMemcpyHostToDevice, s));
        Fields F = p->fields();
        for (int32_t pass = 0; pass < passes; ++pass) {
            launch(p, FAM_SMOOTH_FIELD, k_smooth_field, xcd_grid(p->N), WO_BLOCK, F, (const float*)a, b);
            std::swap(a, b);
        }
        WO_HIP(hipMemcpyAsync(field, a, bytes, hipMemcpyDeviceToHost, s));
        WO_HIP(hipStreamSynchronize(s));
        dfree(a); dfree(b);
        return 0;
    } catch (const HipError& e) { dfree(a); dfree(b); set_error(std::string("wo_smooth_field: ") + e.msg); return 2; }
      catch (const std::exception& e) { dfree(a); dfree(b); set_error(std::string("wo_smooth_field: ") + e.what()); return 3; }
}

// ---- climate sweeps on caller-owned fields (js/temperature.js:19-66, js/precipitation.js:18-52, :59-195) ----
int wo_diffuse_ocean_warmth(wo_planet* p, const float* r_oceanWarmth, const uint8_t* r_isLand, const float* r_plateContinentality,
                            int32_t passes, float* out) {
    if (!check_planet(p, "wo_diffuse_ocean_warmth")) return 1;
    if (!r_isLand || !out) { set_error("wo_diffuse_ocean_warmth: null pointer"); return 1; }
    WO_TRY
    hipStream_t s = p->ctx->stream; const size_t N = (size_t)p->N;
    DevBufs B;
    const float* w = B.up(r_oceanWarmth, N, s); const uint8_t* land = B.up(r_isLand, N, s); const float* cont = B.up(r_plateContinentality, N, s);
    float* a = B.alloc<float>(N); float* b = B.alloc<float>(N);
    launch(p, FAM_CLIMATE, k_warmth_seed, blocks_for(p->N, 4096), WO_BLOCK, w, land, a, p->N);
    const Fields F = p->fields(); const ClimateMesh M = climate_mesh(p);
    for (int32_t pass = 0; pass < passes; ++pass) { launch(p, FAM_CLIMATE, k_warmth_diffuse, xcd_grid(p->N), WO_BLOCK, F, M, (const float*)a, cont, b); std::swap(a, b); }
    WO_HIP(hipMemcpyAsync(out, a, N * sizeof(float), hipMemcpyDeviceToHost, s));
    WO_HIP(hipStreamSynchronize(s));
    return 0;
    WO_CATCH("wo_diffuse_ocean_warmth")
}

int wo_wind_convergence(wo_planet* p, const float* r_wind3dX, const float* r_wind3dY, const float* r_wind3dZ, float* out) {
    if (!check_planet(p, "wo_wind_convergence")) return 1;
    if (!r_wind3dX || !r_wind3dY || !r_wind3dZ || !out) { set_error("wo_wind_convergence: null pointer"); return 1; }
    WO_TRY
    hipStream_t s = p->ctx->stream; const size_t N = (size_t)p->N;
    DevBufs B;
    const float* wx = B.up(r_wind3dX, N, s); const float* wy = B.up(r_wind3dY, N, s); const float* wz = B.up(r_wind3dZ, N, s);
    float* o = B.alloc<float>(N);
    launch(p, FAM_CLIMATE, k_wind_convergence, xcd_grid(p->N), WO_BLOCK, p->fields(), climate_mesh(p), wx, wy, wz, o);
    WO_HIP(hipMemcpyAsync(out, o, N * sizeof(float), hipMemcpyDeviceToHost, s));
    WO_HIP(hipStreamSynchronize(s));
    return 0;
    WO_CATCH("wo_wind_convergence")
}

int wo_advect_moisture(wo_planet* p, const float* r_heightKm, const uint8_t* r_isLand, const float* r_windE, const float* r_windN,
                       const float* r_wind3dX, const float* r_wind3dY, const float* r_wind3dZ, const float* r_oceanWarmth,
                       const int32_t* r_coastDistLand, int32_t maxHops, float* out) {
    if (!check_planet(p, "wo_advect_moisture")) return 1;
    if (!r_heightKm || !r_isLand || !r_windE || !r_windN || !r_wind3dX || !r_wind3dY || !r_wind3dZ || !r_coastDistLand || !out) { set_error("wo_advect_moisture: null pointer"); return 1; }
    if (maxHops < 1) { set_error("wo_advect_moisture: maxHops must be >= 1"); return 1; }
    WO_TRY
    hipStream_t s = p->ctx->stream; const size_t N = (size_t)p->N;
    DevBufs B;
    const float* hk = B.up(r_heightKm, N, s); const uint8_t* land = B.up(r_isLand, N, s); const float* we = B.up(r_windE, N, s); const float* wn = B.up(r_windN, N, s);
    const float* wx = B.up(r_wind3dX, N, s); const float* wy = B.up(r_wind3dY, N, s); const float* wz = B.up(r_wind3dZ, N, s);
    const float* w = B.up(r_oceanWarmth, N, s); const int32_t* cd = B.up(r_coastDistLand, N, s);
    float* a = B.alloc<float>(N); float* b = B.alloc<float>(N);
    const Fields F = p->fields(); const ClimateMesh M = climate_mesh(p);
    const double depletionBase = 1 - std::pow(0.78, 1.0 / maxHops);                  // js/precipitation.js:123
    launch(p, FAM_CLIMATE, k_moisture_seed, xcd_grid(p->N), WO_BLOCK, F, M, land, wx, wy, wz, w, cd, a);
    for (int32_t it = 0; it < maxHops; ++it) {
        launch(p, FAM_CLIMATE, k_moisture_advect, xcd_grid(p->N), WO_BLOCK, F, M, (const float*)a, hk, land, we, wn, wx, wy, wz, maxHops, depletionBase, b);
        std::swap(a, b);
    }
    WO_HIP(hipMemcpyAsync(out, a, N * sizeof(float), hipMemcpyDeviceToHost, s));
    WO_HIP(hipStreamSynchronize(s));
    return 0;
    WO_CATCH("wo_advect_moisture")
}

int32_t wo_planet_num_regions(const wo_planet* p) { return p ? p->N : 0; }

// projectCoarsePlates (js/coarse-plates.js:51-117) on the planet's resident r_xyz
int wo_project_coarse_plates(wo_planet* p, int32_t coarseRegions, const int32_t* coarseAdjOffset, const int32_t* coarseAdjList,
                             const float* coarse_xyz, const int32_t* coarse_r_plate, double seed, int32_t numPlates, int32_t* r_plate) {
    if (!check_planet(p, "wo_project_coarse_plates")) return 1;
    if (coarseRegions < 1 || !coarseAdjOffset || !coarseAdjList || !coarse_xyz || !coarse_r_plate || !r_plate) {
        set_error("wo_project_coarse_plates: bad arguments"); return 1;
    }
    {   // the kernel walks this CSR on the device: validate it the way wo_planet_create validates the planet's
        if (coarseAdjOffset[0] != 0) { set_error("wo_project_coarse_plates: coarseAdjOffset[0] != 0"); return 1; }
        for (int32_t r = 0; r < coarseRegions; ++r)
            if (coarseAdjOffset[r + 1] < coarseAdjOffset[r]) { set_error("wo_project_coarse_plates: coarseAdjOffset is not monotone"); return 1; }
        const int32_t Ec = coarseAdjOffset[coarseRegions];
        for (int32_t i = 0; i < Ec; ++i)
            if (coarseAdjList[i] < 0 || coarseAdjList[i] >= coarseRegions) { set_error("wo_project_coarse_plates: coarseAdjList entry out of range"); return 1; }
    }
    int32_t *d_off = nullptr, *d_adj = nullptr, *d_plate = nullptr, *d_grid = nullptr, *d_out = nullptr; float* d_cxyz = nullptr;
    WO_TRY
    hipStream_t s = p->ctx->stream;
    const int32_t NC = coarseRegions, E = coarseAdjOffset[NC];
    CoarsePlates C;
    C.NC = NC; C.gridZ = 64; C.gridLon = 128;
    d_off = dalloc<int32_t>(NC + 1); d_adj = dalloc<int32_t>(E); d_plate = dalloc<int32_t>(NC); d_cxyz = dalloc<float>(3 * (size_t)NC);
    d_grid = dalloc<int32_t>((size_t)C.gridZ * C.gridLon); d_out = dalloc<int32_t>(p->N);
    WO_HIP(hipMemcpyAsync(d_off, coarseAdjOffset, (size_t)(NC + 1) * 4, hipMemcpyHostToDevice, s));
    WO_HIP(hipMemcpyAsync(d_adj, coarseAdjList, (size_t)E * 4, hipMemcpyHostToDevice, s));
    WO_HIP(hipMemcpyAsync(d_plate, coarse_r_plate, (size_t)NC * 4, hipMemcpyHostToDevice, s));
    WO_HIP(hipMemcpyAsync(d_cxyz, coarse_xyz, (size_t)NC * 12, hipMemcpyHostToDevice, s));
    C.off = d_off; C.adj = d_adj; C.xyz = d_cxyz; C.plate = d_plate; C.grid = nullptr;
    launch(p, FAM_PLATE_GRID, k_plate_grid, blocks_for((int64_t)C.gridZ * C.gridLon), WO_BLOCK, C, d_grid);
    C.grid = d_grid;
    upload_tables(p, seed + 999);                                                       // :57
    const double coarseEdgeRad = 3.141592653589793 / std::sqrt((double)NC);              // :58
    double lowPlateT = 0;                                                                // :59 (numPlates < 0: null)
    if (numPlates >= 0) lowPlateT = std::max(0.0, std::min(1.0, (80 - numPlates) / 60.0));
    const double perturbAmp = coarseEdgeRad * (1.5 + 1.0 * lowPlateT);                   // :60
    launch(p, FAM_PLATE_PROJECT, k_plate_project, blocks_for(p->N), WO_BLOCK, C, (const uint8_t*)p->d_tables, (const float*)p->d_xyz, p->N, perturbAmp, d_out);
    WO_HIP(hipMemcpyAsync(r_plate, d_out, (size_t)p->N * 4, hipMemcpyDeviceToHost, s));
    WO_HIP(hipStreamSynchronize(s));
    dfree(d_off); dfree(d_adj); dfree(d_plate); dfree(d_cxyz); dfree(d_grid); dfree(d_out);
    return 0;
    } catch (const HipError& e) {
        dfree(d_off); dfree(d_adj); dfree(d_plate); dfree(d_cxyz); dfree(d_grid); dfree(d_out);
        set_error(std::string("wo_project_coarse_plates: ") + e.msg); return 2;
    } catch (const std::exception& e) {
        dfree(d_off); dfree(d_adj); dfree(d_plate); dfree(d_cxyz); dfree(d_grid); dfree(d_out);
        set_error(std::string("wo_project_coarse_plates: ") + e.what()); return 3;
    }
}

int wo_noise_eval(wo_ctx* ctx, double seed, int32_t kind, int32_t octaves, double p0, double p1, double p2, int64_t n,
                  const double* xyz, double* out) {
    if (!ctx || !xyz || !out || n < 0 || kind < 0 || kind > 2) { set_error("wo_noise_eval: bad arguments"); return 1; }
    if (n == 0) return 0;
    double *d_in = nullptr, *d_out = nullptr; uint8_t* d_t = nullptr;
    try {
        WO_HIP(hipSetDevice(ctx->device));
        uint8_t t[1024];
        noise_tables(seed, t, t + 512);
        d_in = dalloc<double>(3 * (size_t)n); d_out = dalloc<double>((size_t)n); d_t = dalloc<uint8_t>(1024);
        hipStream_t s = ctx->stream;
        WO_HIP(hipMemcpyAsync(d_t, t, 1024, hipMemcpyHostToDevice, s));
        WO_HIP(hipMemcpyAsync(d_in, xyz, 3 * (size_t)n * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_noise_eval, dim3(blocks_for(n, 4096)), dim3(WO_BLOCK), 0, s, (const uint8_t*)d_t, kind, octaves, p0, p1, p2, n,
                           (const double*)d_in, d_out);
        WO_HIP(hipMemcpyAsync(out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, s));
        WO_HIP(hipStreamSynchronize(s));
        dfree(d_in); dfree(d_out); dfree(d_t);
        return 0;
    } catch (const HipError& e) { dfree(d_in); dfree(d_out); dfree(d_t); set_error(std::string("wo_noise_eval: ") + e.msg); return 2; }
      catch (const std::exception& e) { dfree(d_in); dfree(d_out); dfree(d_t); set_error(std::string("wo_noise_eval: ") + e.what()); return 3; }
}

// ---- measurement ----
int wo_timer_start(wo_planet* p) {
    if (!check_planet(p, "wo_timer_start")) return 1;
    WO_TRY WO_HIP(hipEventRecord(p->evStart, p->ctx->stream)); return 0; WO_CATCH("wo_timer_start")
}
int wo_timer_stop_ms(wo_planet* p, double* ms) {
    if (!check_planet(p, "wo_timer_stop_ms") || !ms) return 1;
    WO_TRY
    WO_HIP(hipEventRecord(p->evStop, p->ctx->stream));
    WO_HIP(hipEventSynchronize(p->evStop));
    float f = 0; WO_HIP(hipEventElapsedTime(&f, p->evStart, p->evStop));
    *ms = f;
    return 0;
    WO_CATCH("wo_timer_stop_ms")
}
// What an event pair measures around nothing, around a kernel that does nothing, and around two of them back to back: every family's time is a sum
// of such pairs, one per launch, so for launches of a few microseconds (the radix sort's: 5-25 us) the pair itself is a visible share of the figure.
// Reported as three families of their own; 2 x (one kernel) - (two kernels) is what a pair adds to the kernel it brackets (bench.py takes it off, and says so).
__global__ void k_profile_noop() {}
static void profile_calibrate(wo_planet* p) {
    hipStream_t s = p->ctx->stream;
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(k_profile_noop, dim3(1), dim3(64), 0, s);      // (code object loaded, queues warm)
    WO_HIP(hipStreamSynchronize(s));
    auto pair = [&](int fam, int kernels) {
        hipEvent_t a = profile_event(p), b = profile_event(p);
        WO_HIP(hipEventRecord(a, s));
        for (int q = 0; q < kernels; ++q) hipLaunchKernelGGL(k_profile_noop, dim3(1), dim3(64), 0, s);
        WO_HIP(hipEventRecord(b, s));
        p->pending.push_back({fam, a, b});
    };
    for (int k = 0; k < 64; ++k) { pair(FAM_EVENT_PAIR, 0); pair(FAM_EVENT_PAIR_NOOP, 1); pair(FAM_EVENT_PAIR_NOOP2, 2); }
    profile_resolve(p);
}
int wo_profile_enable(wo_planet* p, int32_t on) {
    if (!check_planet(p, "wo_profile_enable")) return 1;
    WO_TRY if (!on) profile_resolve(p); p->profiling = on != 0; if (on) profile_calibrate(p); return 0; WO_CATCH("wo_profile_enable")
}
int wo_profile_reset(wo_planet* p) {
    if (!check_planet(p, "wo_profile_reset")) return 1;
    WO_TRY
    profile_resolve(p);
    for (int i = 0; i < FAM_COUNT; ++i) { p->famMs[i] = 0; p->famLaunches[i] = 0; }
    return 0;
    WO_CATCH("wo_profile_reset")
}
int wo_profile_report(wo_planet* p, int32_t cap, const char** names, double* total_ms, int64_t* launches, int32_t* count) {
    if (!check_planet(p, "wo_profile_report") || !count) return 1;
    WO_TRY
    profile_resolve(p);
    int32_t n = 0;
    for (int i = 0; i < FAM_COUNT && n < cap; ++i) {
        if (p->famLaunches[i] == 0) continue;
        if (names) names[n] = kFamilyNames[i];
        if (total_ms) total_ms[n] = p->famMs[i];
        if (launches) launches[n] = p->famLaunches[i];
        ++n;
    }
    *count = n;
    return 0;
    WO_CATCH("wo_profile_report")
}
static void stage_timing_resolve(wo_planet* p) {
    if (!p->stagePending) return;
    p->stagePending = false;
    WO_HIP(hipStreamSynchronize(p->ctx->stream));
    std::map<std::string, double> acc;
    for (auto& b : p->stageBrackets) {
        float ms = 0; WO_HIP(hipEventElapsedTime(&ms, b.a, b.b));
        acc[b.name] += ms;
        p->eventPool.push_back(b.a); p->eventPool.push_back(b.b);
    }
    p->stageBrackets.clear();
    p->stageTiming.clear();
    for (auto& n : p->stageSeen) { const auto& c = n.second; p->stageTiming.push_back({n.first, acc[n.first] * (c.second > 0 ? (double)c.first / (double)c.second : 1.0)}); }
    p->stageSeen.clear();
}
int wo_last_stage_timing(wo_planet* p, int32_t cap, const char** stages, double* ms, int32_t* count) {
    if (!p || !count) return 1;
    try { stage_timing_resolve(p); } catch (const wo::HipError& e) { wo::set_error(std::string("wo_last_stage_timing: ") + e.msg); return 1; }
    int32_t n = 0;
    for (auto& st : p->stageTiming) { if (n >= cap) break; if (stages) stages[n] = st.first.c_str(); if (ms) ms[n] = st.second; ++n; }
    *count = n;
    return 0;
}
int wo_last_erode_stats(wo_planet* p, int32_t cap, const char** names, double* values, int32_t* count) {
    if (!p || !count) return 1;
    int32_t n = 0;
    for (auto& st : p->erodeStats) { if (n >= cap) break; if (names) names[n] = st.first.c_str(); if (values) values[n] = st.second; ++n; }
    *count = n;
    return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// assignElevation (js/elevation.js:216-1391): collisions on the device, order-defined graph work on the
// host (elevation_host.cc), the fused per-cell uplift pass on the device.
// ---------------------------------------------------------------------------------------------------
namespace wo {

struct DevPlateTable {
    PlateTable t{}; uint8_t* hasVec = nullptr; double* pole = nullptr; double* omega = nullptr; uint8_t* isOcean = nullptr; double* density = nullptr;
    void upload(const wo_plate_table& h, hipStream_t s) {
        const size_t n = (size_t)h.numIds;
        hasVec = dalloc<uint8_t>(n); pole = dalloc<double>(3 * n); omega = dalloc<double>(n); isOcean = dalloc<uint8_t>(n); density = dalloc<double>(n);
        WO_HIP(hipMemcpyAsync(hasVec, h.hasVec, n, hipMemcpyHostToDevice, s));
        WO_HIP(hipMemcpyAsync(pole, h.pole, 3 * n * 8, hipMemcpyHostToDevice, s));
        WO_HIP(hipMemcpyAsync(omega, h.omega, n * 8, hipMemcpyHostToDevice, s));
        WO_HIP(hipMemcpyAsync(isOcean, h.isOcean, n, hipMemcpyHostToDevice, s));
        WO_HIP(hipMemcpyAsync(density, h.density, n * 8, hipMemcpyHostToDevice, s));
        t.numIds = h.numIds; t.hasVec = hasVec; t.pole = pole; t.omega = omega; t.isOcean = isOcean; t.density = density;
    }
    void release() { dfree(hasVec); dfree(pole); dfree(omega); dfree(isOcean); dfree(density); }
};

struct DevCollision {
    CollisionOut o{};
    void alloc(size_t N) { o.stress = dalloc<float>(N); o.subduct = dalloc<float>(N); o.btype = dalloc<int8_t>(N); o.bothOcean = dalloc<uint8_t>(N); o.hasOcean = dalloc<uint8_t>(N); o.setCode = dalloc<uint8_t>(N); }
    void download(CollisionHost& h, size_t N, hipStream_t s) {
        h.resize((int32_t)N);
        WO_HIP(hipMemcpyAsync(h.stress.data(), o.stress, N * 4, hipMemcpyDeviceToHost, s)); WO_HIP(hipMemcpyAsync(h.subduct.data(), o.subduct, N * 4, hipMemcpyDeviceToHost, s));
        WO_HIP(hipMemcpyAsync(h.btype.data(), o.btype, N, hipMemcpyDeviceToHost, s)); WO_HIP(hipMemcpyAsync(h.bothOcean.data(), o.bothOcean, N, hipMemcpyDeviceToHost, s));
        WO_HIP(hipMemcpyAsync(h.hasOcean.data(), o.hasOcean, N, hipMemcpyDeviceToHost, s)); WO_HIP(hipMemcpyAsync(h.setCode.data(), o.setCode, N, hipMemcpyDeviceToHost, s));
    }
    void release() { dfree(o.stress); dfree(o.subduct); dfree(o.btype); dfree(o.bothOcean); dfree(o.hasOcean); dfree(o.setCode); }
};

static PlateTable host_table(const wo_plate_table& h) { PlateTable t; t.numIds = h.numIds; t.hasVec = h.hasVec; t.pole = h.pole; t.omega = h.omega; t.isOcean = h.isOcean; t.density = h.density; return t; }

template <class T, class A> static T* upload_vec(const std::vector<T, A>& v, hipStream_t s) {
    T* d = dalloc<T>(v.size());
    WO_HIP(hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
    return d;
}

static void assign_elevation(wo_planet* p, const int32_t* r_plate, const wo_plate_table* plates, const int32_t* plateSeeds, int32_t numPlateSeeds,
                             const int32_t* r_superPlate, const wo_plate_table* superPlates, const uint8_t* perm, const uint8_t* pm12,
                             double noiseMag, double seed, double spread, float* r_elevation, float* r_stress, float* debugLayers,
                             int32_t* mountain_r, int32_t* coastline_r, int32_t* ocean_r, int32_t* setCounts) {
    hipStream_t s = p->ctx->stream;
    const int32_t N = p->N;
    const int gridN = blocks_for(N);
    const bool hasSuper = r_superPlate != nullptr && superPlates != nullptr;
    for (int32_t r = 0; r < N; ++r) if (r_plate[r] < 0 || r_plate[r] >= plates->numIds) throw HipError{"r_plate entry outside the plate table"};
    if (hasSuper) for (int32_t r = 0; r < N; ++r) if (r_superPlate[r] < 0 || r_superPlate[r] >= superPlates->numIds) throw HipError{"r_superPlate entry outside the super-plate table"};
    std::vector<std::pair<std::string, double>> timing;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* stage) { WO_HIP(hipStreamSynchronize(s)); auto now = std::chrono::steady_clock::now(); timing.push_back({stage, std::chrono::duration<double, std::milli>(now - t0).count()}); t0 = now; };

    // noise instances: the caller's `noise` plus the nine the reference seeds itself (js/elevation.js:539,636,980-982,1056,1127-1129)
    std::vector<uint8_t> tabs(EL_TAB_COUNT * 1024), hs3(1024);
    std::memcpy(tabs.data(), perm, 512); std::memcpy(tabs.data() + 512, pm12, 512);
    const double offs[EL_TAB_COUNT] = {0, 419, 557, 77, 133, 211, 307, 501, 502};
    for (int k = 1; k < EL_TAB_COUNT; ++k) noise_tables(seed + offs[k], tabs.data() + k * 1024, tabs.data() + k * 1024 + 512);
    noise_tables(seed + 503, hs3.data(), hs3.data() + 512);
    uint8_t* d_tabs = upload_vec(tabs, s);
    int32_t* d_plate = dalloc<int32_t>(N);
    WO_HIP(hipMemcpyAsync(d_plate, r_plate, (size_t)N * 4, hipMemcpyHostToDevice, s));
    DevPlateTable dT, dTS; dT.upload(*plates, s);
    DevCollision cS, cP; cS.alloc(N);
    CollisionHost hS, hP;
    launch(p, FAM_ELEV_COLLISION, k_collision, gridN, WO_BLOCK, N, (const int32_t*)p->d_off, (const int32_t*)p->d_adj, (const float*)p->d_xyz,
           (const int32_t*)d_plate, dT.t, (const uint8_t*)d_tabs, cS.o);
    cS.download(hS, N, s);
    int32_t* d_super = nullptr;
    if (hasSuper) {
        d_super = dalloc<int32_t>(N);
        WO_HIP(hipMemcpyAsync(d_super, r_superPlate, (size_t)N * 4, hipMemcpyHostToDevice, s));
        dTS.upload(*superPlates, s); cP.alloc(N);
        launch(p, FAM_ELEV_COLLISION, k_collision, gridN, WO_BLOCK, N, (const int32_t*)p->d_off, (const int32_t*)p->d_adj, (const float*)p->d_xyz,
               (const int32_t*)d_super, dTS.t, (const uint8_t*)d_tabs, cP.o);
        cP.download(hP, N, s);
    }
    lap(hasSuper ? "Collisions (dual)" : "Collisions");

    // host stage
    if (p->h_xyz.empty()) throw HipError{"planet has no host copy of r_xyz"};
    ElevMesh M{N, p->h_off.data(), p->h_adj.data(), p->h_xyz.data()};
    ElevInputs I{};
    I.plate = r_plate; I.plates = host_table(*plates); I.plateSeeds = plateSeeds; I.numPlateSeeds = numPlateSeeds;
    I.superPlate = hasSuper ? r_superPlate : nullptr; if (hasSuper) I.superPlates = host_table(*superPlates);
    I.seed = seed; I.spread = spread; I.noiseMag = noiseMag; I.hsNoise3 = NoiseTab{hs3.data(), hs3.data() + 512};
    ElevHostState H; ElevParams Q{}; std::vector<Dome> domes;
    // The FIFO BFS fields (js/elevation.js:464-631, 1059-1086) run on the device (elevation_bfs.h), started from inside the
    // host stage as soon as their inputs exist and overlapped with the serial RNG-ordered distance fields on the host.
    ElevFields F{};
    F.xyz = p->d_xyz; F.plate = d_plate;
    std::vector<void*> tmp;
    auto up = [&](auto& v) { auto* d = upload_vec(v, s); tmp.push_back((void*)d); return d; };
    auto dev = [&](auto* proto, size_t n) { using T = std::remove_pointer_t<decltype(proto)>; T* d = dalloc<T>(n); tmp.push_back((void*)d); return d; };
    float *b_dBdry = nullptr, *b_csm = nullptr, *b_cssm = nullptr, *b_rift = nullptr, *b_ridge = nullptr, *b_frac = nullptr, *b_ba = nullptr, *b_bas = nullptr, *b_arc = nullptr, *b_arcs = nullptr;
    uint8_t* b_conv = nullptr;
    std::vector<std::vector<int32_t>> bfsSeeds;             // host seed lists stay alive until the stream has consumed them
    bfsSeeds.reserve(8);
    auto bfs_on_device = [&](const ElevParams& Qs, int32_t maxCD, double maxStress) {
        const size_t n = (size_t)N;
        F.isOcean = up(H.isOcean); F.stress = up(H.stress); F.subduct = up(H.subduct); F.btype = up(H.btype);
        const uint8_t* d_both = up(H.bothOcean); const uint8_t* d_has = up(H.hasOcean); (void)d_both; (void)d_has;
        b_dBdry = dev((float*)nullptr, n); b_csm = dev((float*)nullptr, n); b_cssm = dev((float*)nullptr, n); b_conv = dev((uint8_t*)nullptr, n);
        b_rift = dev((float*)nullptr, n); b_ridge = dev((float*)nullptr, n); b_frac = dev((float*)nullptr, n);
        b_ba = dev((float*)nullptr, n); b_bas = dev((float*)nullptr, n); b_arc = dev((float*)nullptr, n); b_arcs = dev((float*)nullptr, n);
        int32_t* listA = dev((int32_t*)nullptr, n); int32_t* listB = dev((int32_t*)nullptr, n);
        int32_t* cnt = dev((int32_t*)nullptr, n); int32_t* base = dev((int32_t*)nullptr, n); int32_t* pushPos = dev((int32_t*)nullptr, n);
        unsigned long long* attrKey = dev((unsigned long long*)nullptr, n);
        int32_t* counters = dev((int32_t*)nullptr, 4);
        BfsCtx B{N, p->d_off, p->d_adj, F.isOcean, d_plate};
        // seed lists in ascending id (the reference's scan order), built on the host from the arrays the host stage holds
        auto seeds_of = [&](auto pred) {
            std::vector<std::vector<int32_t>> part(host_threads() + 1);
            parallel_ranges(N, [&](int64_t b, int64_t e, int t) { for (int64_t r = b; r < e; ++r) if (pred((int32_t)r)) part[t].push_back((int32_t)r); });
            std::vector<int32_t> q; for (auto& v : part) q.insert(q.end(), v.begin(), v.end());
            return q;
        };
        const int grid = 512;
        auto run_field = [&](int32_t mode, std::vector<int32_t>&& seedList, float* dist, float init, float* a0, float* a1, uint8_t* a2, int32_t maxDist) {
            bfsSeeds.push_back(std::move(seedList));
            const std::vector<int32_t>& seeds = bfsSeeds.back();
            const bool carry = a0 != nullptr;
            launch(p, FAM_ELEV_COLLISION, k_bfs_init, blocks_for(N, 4096), WO_BLOCK, dist, init, a0, a1, a2, carry ? pushPos : (int32_t*)nullptr,
                   mode == BFS_COAST ? attrKey : (unsigned long long*)nullptr, N);
            const int32_t ns = (int32_t)seeds.size();
            if (ns == 0) return;
            WO_HIP(hipMemcpyAsync(listA, seeds.data(), (size_t)ns * 4, hipMemcpyHostToDevice, s));       // no host wait: the device keeps running the previous field
            hipLaunchKernelGGL(k_set_counters, dim3(1), dim3(1), 0, s, counters, ns, 0, 0, 0);
            launch(p, FAM_ELEV_COLLISION, k_bfs_seed, blocks_for(ns, 1024), WO_BLOCK, mode, (const int32_t*)listA, ns, dist, a0, a1, a2,
                   (const float*)F.stress, (const float*)F.subduct, (const int8_t*)F.btype, maxStress);
            int32_t* cur = listA; int32_t* nxt = listB;
            for (int32_t level = 1; level <= maxDist; ++level) {
                int32_t* cc = counters + ((level - 1) % 3); int32_t* nc = counters + (level % 3); int32_t* zc = counters + ((level + 1) % 3);
                if (!carry) {
                    launch(p, FAM_ELEV_COLLISION, k_bfs_plain, grid, WO_BLOCK, B, mode, dist, (const int32_t*)cur, (const int32_t*)cc, nxt, nc, zc, level);
                } else {
                    launch(p, FAM_ELEV_COLLISION, k_bfs_push, grid, WO_BLOCK, B, mode, (const float*)dist, (const int32_t*)cur, (const int32_t*)cc, pushPos,
                           mode == BFS_COAST ? attrKey : (unsigned long long*)nullptr, (const float*)a0, level);
                    launch(p, FAM_ELEV_COLLISION, k_bfs_count, grid, WO_BLOCK, B, (const float*)dist, (const int32_t*)cur, (const int32_t*)cc, (const int32_t*)pushPos, cnt, level);
                    launch(p, FAM_ELEV_COLLISION, k_bfs_scan, 1, 1024, (const int32_t*)cnt, (const int32_t*)cc, base, nc);
                    launch(p, FAM_ELEV_COLLISION, k_bfs_assign, grid, WO_BLOCK, B, mode, dist, (const int32_t*)cur, (const int32_t*)cc, (const int32_t*)pushPos,
                           (const unsigned long long*)(mode == BFS_COAST ? attrKey : nullptr), (const int32_t*)base, nxt, a0, a1, a2, level);
                }
                std::swap(cur, nxt);
            }
        };
        const uint8_t* oc = H.isOcean.data(); const int8_t* bt = H.btype.data(); const uint8_t* both = H.bothOcean.data(); const uint8_t* has = H.hasOcean.data();
        const float* sub = H.subduct.data();
        run_field(BFS_COAST, seeds_of([&](int32_t r) { const uint8_t o = oc[r]; for (int32_t ni = M.off[r]; ni < M.off[r + 1]; ++ni) if (oc[M.adj[ni]] != o) return true; return false; }),
                  b_dBdry, (float)(maxCD + 1), b_csm, b_cssm, b_conv, maxCD);
        run_field(BFS_RIFT, seeds_of([&](int32_t r) { return bt[r] == 2 && !has[r]; }), b_rift, INFINITY, nullptr, nullptr, nullptr, Qs.riftHalfWidth);
        run_field(BFS_RIDGE, seeds_of([&](int32_t r) { return bt[r] == 2 && both[r]; }), b_ridge, INFINITY, nullptr, nullptr, nullptr, Qs.ridgeHalfWidth);
        run_field(BFS_FRACTURE, seeds_of([&](int32_t r) { return bt[r] == 3 && both[r]; }), b_frac, INFINITY, nullptr, nullptr, nullptr, Qs.fractureHalfWidth);
        run_field(BFS_BACKARC, seeds_of([&](int32_t r) { return bt[r] == 1 && has[r] && (double)sub[r] < 0.50; }), b_ba, INFINITY, b_bas, nullptr, nullptr, Qs.baEnd);
        run_field(BFS_ARC, seeds_of([&](int32_t r) { return bt[r] == 1 && both[r] && (double)sub[r] < 0.45; }), b_arc, (float)(Qs.maxArcDist + 1), b_arcs, nullptr, nullptr, Qs.maxArcDist);
    };
    elevation_host_stage(M, I, hS, hasSuper ? &hP : nullptr, H, Q, domes, bfs_on_device);
    lap("Stress, sets, distance fields (host) || BFS fields (device)");

    // per-cell pass
    F.dBdry = b_dBdry; F.coastStressMax = b_csm; F.coastSubductMax = b_cssm; F.coastConvergent = b_conv; F.riftDist = b_rift; F.ridgeDist = b_ridge;
    F.fractureDist = b_frac; F.backArcDist = b_ba; F.backArcStress = b_bas; F.arcDist = b_arc; F.arcStress = b_arcs;
    F.distMountain = up(H.distMountain); F.distOcean = up(H.distOcean); F.distCoastline = up(H.distCoastline); F.distCoast = up(H.distCoast);
    F.distCoastLand = up(H.distCoastLand);
    F.elev = p->d_e;
    float* d_dl = nullptr;
    if (debugLayers) { d_dl = dalloc<float>((size_t)DL_COUNT * N); WO_HIP(hipMemsetAsync(d_dl, 0, (size_t)DL_COUNT * N * 4, s)); }
    F.dl = d_dl;
    if (domes.empty()) domes.push_back(Dome{});
    Dome* d_domes = upload_vec(domes, s);
    launch(p, FAM_ELEV_MAIN, k_elevation, gridN, WO_BLOCK, F, Q, dT.t, (const uint8_t*)d_tabs, (const Dome*)d_domes);
    if (r_elevation) WO_HIP(hipMemcpyAsync(r_elevation, p->d_e, (size_t)N * 4, hipMemcpyDeviceToHost, s));
    if (debugLayers) WO_HIP(hipMemcpyAsync(debugLayers, d_dl, (size_t)DL_COUNT * N * 4, hipMemcpyDeviceToHost, s));
    lap("Elevation loop + coastal + arcs + hotspots + compression (device)");
    if (r_stress) std::memcpy(r_stress, H.stress.data(), (size_t)N * 4);
    if (mountain_r) std::memcpy(mountain_r, H.mountain.data(), H.mountain.size() * 4);
    if (coastline_r) std::memcpy(coastline_r, H.coastline.data(), H.coastline.size() * 4);
    if (ocean_r) std::memcpy(ocean_r, H.ocean.data(), H.ocean.size() * 4);
    if (setCounts) { setCounts[0] = (int32_t)H.mountain.size(); setCounts[1] = (int32_t)H.coastline.size(); setCounts[2] = (int32_t)H.ocean.size(); }
    for (void* d : tmp) (void)hipFree(d);
    (void)hipFree(d_tabs); (void)hipFree(d_plate); (void)hipFree(d_domes);
    if (d_super) (void)hipFree(d_super);
    if (d_dl) (void)hipFree(d_dl);
    dT.release(); dTS.release(); cS.release(); cP.release();
    for (auto& b : p->stageBrackets) { p->eventPool.push_back(b.a); p->eventPool.push_back(b.b); }
    p->stageBrackets.clear(); p->stageSeen.clear(); p->stagePending = false;
    p->stageTiming = timing;
}

}  // namespace wo

extern "C" int wo_assign_elevation(wo_planet* p, const int32_t* r_plate, const wo_plate_table* plates, const int32_t* plateSeeds,
                                   int32_t numPlateSeeds, const int32_t* r_superPlate, const wo_plate_table* superPlates,
                                   const uint8_t* noisePerm512, const uint8_t* noisePm12_512, double noiseMag, double seed, double spread,
                                   float* r_elevation, float* r_stress, float* debugLayers, int32_t* mountain_r, int32_t* coastline_r,
                                   int32_t* ocean_r, int32_t* setCounts) {
    if (!check_planet(p, "wo_assign_elevation")) return 1;
    if (!r_plate || !plates || !plateSeeds || !noisePerm512 || !noisePm12_512) { set_error("wo_assign_elevation: null pointer"); return 1; }
    WO_TRY
    assign_elevation(p, r_plate, plates, plateSeeds, numPlateSeeds, r_superPlate, superPlates, noisePerm512, noisePm12_512, noiseMag, seed, spread,
                     r_elevation, r_stress, debugLayers, mountain_r, coastline_r, ocean_r, setCounts);
    return 0;
    WO_CATCH("wo_assign_elevation")
}
