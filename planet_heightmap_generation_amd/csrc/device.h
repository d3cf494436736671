// Device-side state of libworogen: context (device + stream), planet (resident mesh + fields + scratch),
// launch / profiling helpers.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "host_util.h"

#include "erode_ops.h"
#include "flood_ops.h"
#include "wo_internal.h"

namespace wo {

struct HipError { std::string msg; };

#define WO_HIP(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            throw ::wo::HipError{std::string(#call) + " -> " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + \
                                 std::to_string(__LINE__) + ")"};                                      \
    } while (0)

constexpr int WO_BLOCK = 256;
inline int blocks_for(int64_t n, int maxBlocks = 1 << 20) {
    int64_t b = (n + WO_BLOCK - 1) / WO_BLOCK;
    if (b < 1) b = 1;
    if (b > maxBlocks) b = maxBlocks;
    return (int)b;
}

// grid for WO_XCD_CELLS kernels: one cell per thread, block count rounded up to a multiple of 8
inline int xcd_tile(int64_t n) {          // blocks per tile: a power of two, ~1/64 of the grid, at most 256 (65 k cells)
    const int64_t b = (n + WO_BLOCK - 1) / WO_BLOCK;
    int t = 1;
    while (t < 256 && (int64_t)t * 128 <= b) t <<= 1;
    return t;
}
inline int xcd_grid(int64_t n) {          // block count padded to whole rounds of 8 tiles
    const int64_t b = (n + WO_BLOCK - 1) / WO_BLOCK, per = 8 * (int64_t)xcd_tile(n);
    return (int)(((b + per - 1) / per) * per);
}

// Kernel families for HIP-event profiling (wo_profile_*) — one entry per distinct kernel of the path.
enum Family : int {
    FAM_COAST = 0, FAM_SMOOTH, FAM_SHARPEN, FAM_CREEP, FAM_WARP, FAM_NOISE, FAM_SYNTH, FAM_OCEAN,
    FAM_SORT_KEYS, FAM_SORT_RADIX, FAM_RANK, FAM_RECEIVERS, FAM_FLOW_SNAP, FAM_FLOW_FINAL,
    FAM_SOLVE_SETUP, FAM_SOLVE_ROUND, FAM_SOLVE_FINAL, FAM_THERMAL_EXCESS, FAM_THERMAL_APPLY,
    FAM_GLAC_INDEX, FAM_ICE_RECV, FAM_ICE_ROUND, FAM_CARVE_SETUP, FAM_CARVE_ROUND, FAM_MORAINE, FAM_GLAC_BLEND,
    FAM_SOLVE_PATCH, FAM_ELEV_COLLISION, FAM_ELEV_MAIN, FAM_PLATE_GRID, FAM_PLATE_PROJECT, FAM_SMOOTH_FIELD,
    FAM_FLOOD_EVAL, FAM_FLOOD_APPLY, FAM_FLOOD_MISC, FAM_CLIMATE, FAM_BASIN, FAM_BASIN_SORT, FAM_SOLVE_BASIN, FAM_FLOW_TILES, FAM_MISC, FAM_EVENT_PAIR, FAM_EVENT_PAIR_NOOP, FAM_EVENT_PAIR_NOOP2, FAM_COUNT
};
extern const char* const kFamilyNames[FAM_COUNT];

}  // namespace wo

struct wo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop{};
};

// device state of the flood's pass 1 (flood_kernels.h); static part per land mask, the rest per call
struct wo_flood_gpu {
    int64_t version = -1; int32_t L = 0, cap = 0, nSeeds = 0;
    int32_t *off = nullptr, *adj = nullptr, *cell = nullptr, *seedIdx = nullptr, *seeds = nullptr; double* nz = nullptr;
    float* e = nullptr;
    wo::FlHead *A = nullptr, *P = nullptr, *F = nullptr;
    unsigned long long *Astk = nullptr, *Pstk = nullptr, *Fstk = nullptr;
    int32_t *fdEpoch = nullptr, *inDirty = nullptr; uint8_t* isPending = nullptr;
    int32_t* lists[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};      // dirty x2, pending x2, changed
    void* ctrl = nullptr; void* h_ctrl = nullptr;
    int32_t *jump = nullptr, *outPar = nullptr, *outRoot = nullptr; float* outSurf = nullptr;
    int32_t *h_par = nullptr, *h_root = nullptr; float* h_surf = nullptr;   // pinned
};

namespace wo {
// Environment switches, read once per API call (check_planet -> Options::from_env), never inside an iteration: cross-check routes the tests drive
// (every one must give the default route's bits), the labelled relaxed mode, diagnostics.  The full list with the host-side ones (WO_HOST_THREADS,
// WO_FLOOD_THREADS, WO_FLOOD_HOST, WO_FLOOD_PIN, WO_ELEV_TIMING) is in include/worogen.h; the test suite's hooks come in ONE variable (host_util.h: WO_TEST_HOOKS).
struct Options {
    bool layoutIndex = false;          // WO_LAYOUT=index          no patch-major mirror: the call runs on the planet's own cell order (flow accumulation by k_flow_climb)
    bool tileLds = false;              // WO_TILE_LDS=1            neighbour window of a workgroup's tile staged in LDS (receivers, thermal)
    bool floodDevice = false;          // WO_FLOOD=device          pass 1 of the flood as the device label-correcting fixed point
    bool floodTiming = false;          // WO_FLOOD_TIMING          stage laps -> stderr
    bool stageTimingAll = false;       // WO_STAGE_TIMING=all      bracket every iteration
    int  relaxedSortEvery = 1;         // WO_RELAXED_SORT_EVERY=K  RELAXED MODE (not parity): re-sort landCells every K-th iteration only
    bool relaxedFull = false;          // WO_RELAXED=full          RELAXED MODE (not parity): one sort per flood, affine pointer-jumping solve with deferred deposition, Jacobi carve (kernels_impl.h)
    // test hooks (WO_TEST_HOOKS)
    bool basinScramble = false;        // basin_scramble=1         a deliberately wrong basin layout (leftovers for the patch finisher)
    long long carveBudgetMs = 200;     // carve_budget_ms=<n>      spin budget of the one-launch carve (0: it gives up at once and the rounds finish)
    int  carveBlocks = 0;              // carve_blocks=<n>         at most this many workgroups for the one-launch carve
    static Options from_env();
};
}  // namespace wo

struct wo_planet {
    wo_ctx* ctx = nullptr;
    wo::Options opt;
    int32_t N = 0, E = 0, maxDeg = 0;
    // host copies kept for the host-resident flood stage
    wo::hvec<int32_t> h_off, h_adj;          // host mirrors walked in data-dependent order: huge-page advised (host_util.h)
    wo::hvec<float> h_xyz;
    wo::hvec<uint8_t> h_ocean;
    bool h_ocean_valid = false;
    float* h_pinned = nullptr;          // N floats, pinned
    int32_t* h_count = nullptr;         // pinned scalar(s) for round-count read-back
    float* d_redoE = nullptr; int32_t* d_pendingEver = nullptr; int64_t redoCalls = 0;   // erode_composite_checked: the field at entry, tasks any basin launch of the call left pending, calls that had to run again
    unsigned long long *h_word = nullptr, *d_word = nullptr; uint32_t wordSerial = 0;   // host-mapped {serial, value} word the host polls (planet.hip: publish_and_wait)
    wo::FloodScratch flood;
    wo::FloodExchange floodX;           // landmass decomposition: the shares pool their heights when a flood call needs the whole planet's heap (wo_planet_set_flood_exchange)
    void* floodLink = nullptr; void (*floodLinkFree)(void*) = nullptr;   // comm.hip: state of the RCCL form of that exchange
    wo_flood_gpu fgpu;

    // resident mesh
    int32_t *d_off = nullptr, *d_adj = nullptr;
    float *d_dist = nullptr, *d_xyz = nullptr;
    // resident fields
    float *d_e = nullptr, *d_e2 = nullptr, *d_hot = nullptr, *d_orig = nullptr;
    uint8_t *d_ocean = nullptr, *d_coast = nullptr;
    int64_t mirrorMaskVersion = -1;      // oceanVersion of the mask the mirror was built for (mirror_build), -1: another mask
    uint8_t* d_oceanKnown = nullptr; int32_t* d_maskDiff = nullptr; bool oceanKnownValid = false;   // the mask h_ocean describes, on the device (refresh_host_ocean)
    bool hot_valid = false;
    float* d_savedE = nullptr; uint8_t* d_savedOcean = nullptr; bool saved = false;
    uint8_t* d_tables = nullptr;        // perm[512] + pm12[512]
    // erode scratch (allocated on first erodeComposite)
    bool scratch = false;
    bool landIdentity = false;          // erode_composite under the land-first mirror: landIdx[i] == i
    int64_t floodPrefixMirror = -1, floodPrefixStatic = -1;      // (mirror version, flood static version) for which the mirror's first L ids were checked to be the flood's land order
    bool floodPrefixOk = false;
    int32_t *d_landIdx = nullptr, *d_land[2] = {nullptr, nullptr}, *d_rank = nullptr;
    // the initial land list (ascending r, js/terrain-post.js:384-390) and the index-order list are functions of the ocean mask alone: kept while it stays
    int32_t* d_landInit = nullptr; int64_t oceanVersion = 0, landListsOcean = -1; bool landListsMirror = false; int32_t landListsL = -1;
    uint32_t* d_keys[2] = {nullptr, nullptr};
    float *d_cellDist = nullptr, *d_flow = nullptr;
    wo::SolveTask* d_task = nullptr; wo::SolveOut* d_out = nullptr; int32_t *d_haloSend = nullptr, *d_haloRecv = nullptr; float *d_haloBuf = nullptr, *h_haloBuf = nullptr; int32_t nHaloSend = 0, nHaloRecv = 0;   // banded Jacobi passes
    int32_t* d_flowCnt = nullptr; wo::TargetRank* d_tr = nullptr; wo::EventList* d_ev = nullptr; float* d_me = nullptr;
    int32_t *d_carveSlot = nullptr, *d_carveDeps = nullptr, *d_carveDepCnt = nullptr, *d_carveDepPos = nullptr; uint32_t* d_rs[2] = {nullptr, nullptr}; int rsFlip[2] = {0, 0};   /* radix.hip scratch: elevation sort, basin sort */ wo::CarveRec* d_carveRecs = nullptr; wo::CarveExpect* d_carveExpect = nullptr; unsigned long long* d_carveG = nullptr;   /* k_carve_granules: expected tags per task, height granules per cell */ int32_t* d_carveSlotDone = nullptr; int64_t carveCap = 0;   // carve dependency lists
    unsigned long long* d_accCnt = nullptr;
    int32_t *d_ftLr = nullptr, *d_ftParent = nullptr, *d_ftExtCnt = nullptr; uint32_t* d_ftInflow = nullptr; unsigned long long* d_ftRootAcc = nullptr;      // two-level flow accumulation (kernels_impl.h: FlowTiles)
    int32_t *d_jump = nullptr, *d_nj = nullptr;
    int32_t* d_doneAt = nullptr;
    double* d_totalExcess = nullptr;
    float *d_glac = nullptr, *d_iceFlow = nullptr;
    int32_t *d_iceTarget = nullptr, *d_arank = nullptr;
    uint8_t* d_iceUp = nullptr;
    int32_t *d_patchOrder = nullptr, *d_slotOf = nullptr, *d_patchPending = nullptr, *d_patchTotals = nullptr, *d_patchBlk = nullptr; int64_t patchVersion = -1; bool patchMirror = false; int32_t numPatches = 0; int64_t lastPatchLaunches = 1; int64_t solveCalls = 0;
    hipStream_t side = nullptr; hipEvent_t evFork = nullptr, evJoin = nullptr; bool onSide = false;
    uint32_t* d_basinKey = nullptr; int32_t* d_basinVals[2] = {nullptr, nullptr}; int32_t *d_basinJ = nullptr, *d_basinSlot = nullptr, *d_basinRange = nullptr; uint8_t* d_basinLong = nullptr; int64_t basinLaunches = 0; wo::Affine* d_affine[2] = {nullptr, nullptr};   /* relaxed mode: the affine recurrence, ping-pong */ int64_t solvePassSerial = 0;   /* unchecked basin passes so far (their output tag: planet.hip, passTag) */   // basin.hip: component roots (Morton slot space), group-major store order of the pass
    int32_t* h_patchTotals = nullptr;      // pinned: the pending totals of a burst of k_solve_patch launches (run_solve_patches)
    int32_t *d_listA = nullptr, *d_listB = nullptr, *d_counters = nullptr;   // round lists + 4 counters
    void* d_sortTemp = nullptr; size_t sortTempBytes = 0;
    int landCur = 0;                    // which of d_land[] holds the current order
    int32_t L = 0;

    // Patch-major mirror of the mesh for erodeComposite (planet.hip, MirrorScope): the same graph with the cells renamed in
    // Morton order of their positions, rows in the reference's order.  While a scope is active the pointers above (mesh, d_e,
    // d_e2, d_ocean, d_coast) point at the mirror and the o_* members hold the planet's own buffers.
    struct Mirror {
        bool built = false, active = false;
        int32_t *perm = nullptr, *inv = nullptr, *off = nullptr, *adj = nullptr;       // perm: mirror id -> cell id
        float *dist = nullptr, *xyz = nullptr, *e = nullptr, *e2 = nullptr, *hot = nullptr;
        uint8_t *ocean = nullptr, *coast = nullptr;
        wo::hvec<int32_t> h_perm, h_morton;           // h_morton: all cells in Morton order (mask-independent)
        wo::hvec<uint8_t> h_mask;                     // the ocean mask the renaming was built for (land first), empty: plain Morton order
        int64_t version = 0;
        int32_t *o_off = nullptr, *o_adj = nullptr; float *o_dist = nullptr, *o_xyz = nullptr, *o_e = nullptr, *o_e2 = nullptr;
        uint8_t *o_ocean = nullptr, *o_coast = nullptr;
    } mirror;

    // measurement
    hipEvent_t evStart = nullptr, evStop = nullptr;
    bool profiling = false;
    struct Pending { int fam; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> eventPool;
    double famMs[wo::FAM_COUNT] = {0};
    int64_t famLaunches[wo::FAM_COUNT] = {0};
    std::vector<std::pair<std::string, double>> stageTiming;
    // the stage brackets of the last erodeComposite call, not yet turned into milliseconds (wo_last_stage_timing does that: the queries are a
    // millisecond of host work that the call itself no longer waits for)
    struct StageBracket { std::string name; hipEvent_t a, b; };
    std::vector<StageBracket> stageBrackets; std::vector<std::pair<std::string, std::pair<int64_t, int64_t>>> stageSeen; bool stagePending = false;
    std::vector<std::pair<std::string, double>> erodeStats;

    wo::Fields fields() const;
};

namespace wo {

// profiling-aware launch: when p->profiling every launch is bracketed by events on the planet's stream
hipEvent_t profile_event(wo_planet* p);
void profile_resolve(wo_planet* p);

// the stream launches of this planet currently go to (its context's stream, or the planet's side stream while onSide is set)
inline hipStream_t cur_stream(const wo_planet* p) { return (p->onSide && p->side) ? p->side : p->ctx->stream; }

template <class... KArgs, class... Args>
inline void launch(wo_planet* p, int fam, void (*kernel)(KArgs...), int grid, int block, Args... args) {
    hipStream_t s = cur_stream(p);
    if (p->profiling) {
        hipEvent_t a = profile_event(p), b = profile_event(p);
        WO_HIP(hipEventRecord(a, s));
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, s, args...);
        WO_HIP(hipGetLastError());                 // a launch that cannot be configured would otherwise be a silent no-op
        WO_HIP(hipEventRecord(b, s));
        p->pending.push_back({fam, a, b});
        if (p->pending.size() >= 4096) profile_resolve(p);
    } else {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, s, args...);
        WO_HIP(hipGetLastError());
    }
}

template <class... KArgs, class... Args>
inline void launch_shmem(wo_planet* p, int fam, void (*kernel)(KArgs...), int grid, int block, size_t shmem, Args... args) {
    hipStream_t s = p->ctx->stream;
    hipEvent_t a = nullptr, b = nullptr;
    if (p->profiling) { a = profile_event(p); b = profile_event(p); WO_HIP(hipEventRecord(a, s)); }
    if (shmem > (size_t)(64 << 10)) WO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), shmem, s, args...);
    WO_HIP(hipGetLastError());
    if (p->profiling) { WO_HIP(hipEventRecord(b, s)); p->pending.push_back({fam, a, b}); if (p->pending.size() >= 4096) profile_resolve(p); }
}

// sort.hip: stable descending sort of the land list by current elevation + rank scatter
void sort_land_by_elevation(wo_planet* p);
size_t sort_temp_bytes(int32_t n);
void rank_from_land(wo_planet* p);
// radix.hip: the in-tree stable radix sort (scratch: radix_scratch_words(nMax) u32, zero before the first use; flip: call parity kept by the caller)
size_t radix_scratch_words(int32_t nMax);
int radix_sort_pairs(wo_planet* p, int family, uint32_t* const keys[2], int32_t* const vals[2], int32_t n, int beginBit, int endBit, int32_t* posOut,
                     uint32_t* scratch, int32_t nMax, int& flip);
uint32_t* radix_scratch(wo_planet* p, int which);      // 0: elevation sort, 1: basin sort (allocated and cleared on first use)
void select_active_by_rank(wo_planet* p, const int32_t* arank, int32_t* out, int32_t* outCount);   // carve tasks in landCells order
// basin.hip: group-major store order of the solve (d_basinSlot, sorted group keys in d_keys[1]) and the one-launch solve over it
void basin_alloc(wo_planet* p);
void basin_layout(wo_planet* p, bool slotIdentity);
void basin_solve_launch(wo_planet* p, const Fields& F, int32_t launchTag, int32_t* totalPending);
void basin_free(wo_planet* p);

}  // namespace wo
