// Internal declarations shared between the translation units of libworogen (not part of the C ABI).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "host_util.h"

namespace wo {

// mesh_builder.cc
void fib_sphere_points(int N, double jitter, double seed, float* xyz);
int sphere_delaunay(int V, const float* xyz, int* triangles, int* halfedges, std::string& err);
int mesh_csr(int V, int numSides, const int* triangles, const int* halfedges,
             int* adjOffset, int* adjList, int* adjTri, std::string& err);
void neighbor_dist(int V, const int* adjOffset, const int* adjList, const float* xyz, float* out);

// flood_host.cc
#if defined(__HIPCC__)
#define WO_FLOOD_HD __host__ __device__
#else
#define WO_FLOOD_HD
#endif
// cellNoise of priorityFloodCarve, js/terrain-post.js:100-105 — Number (double) products reduced mod 2^32 afterwards.  Host and
// device run the same IEEE double operations: the same bits.
WO_FLOOD_HD inline double flood_cell_noise_of(int32_t r) {
    const double p = (double)r * 2654435761.0;
    uint32_t h = (uint32_t)(uint64_t)p;
    const int32_t x = (int32_t)((h >> 16) ^ h);
    const double q = (double)x * 73244475.0;
    h = (uint32_t)(int64_t)q;
    h = (h >> 16) ^ h;
    return ((double)h / 4294967295.0) * 0.01;
}
struct FloodHeapItem { float key; int32_t cell; };
struct FloodCell { float e; int32_t drain; float surface; int32_t root; };   // one 16-byte record per land cell: a pop touches one line for all four
// Test hooks (WO_TEST_HOOKS, host_util.h) and diagnostics of the host flood, read ONCE per flood call (flood_gather) — never inside a walk.
struct FloodHooks {
    int32_t ringMin = 4096;        // hook flood_ring_min     landmasses of at least this many cells walk on the ring of key buckets
    int32_t chainsMin = 2048;      // hook flood_chains_min   drainage trees of at least this many cells carve on the chain-ordered copy
    int32_t forceDirty = -1;       // hook flood_force_dirty  treat this landmass (by rank in size) as undecided
    bool hasReplayStop = false; float replayStop = 0.0f;     // hook flood_replay_stop  the level at which the replay of the single heap stops
    bool replayPrefix = true;      // hook flood_prefix=0     the replay walks undecided landmasses from their seeds (not from their first contested tie group)
    int32_t forcePrefixPermille = 0;   // hook flood_force_prefix  how much of the forced landmass's pops counts as its decided prefix
    bool pin = false;              // WO_FLOOD_PIN=1        the walk of the largest landmass keeps its CPU and the other flood workers keep off its L3 (opt-in: the library touches no thread affinity unasked)
    bool timing = false;           // WO_FLOOD_TIMING       laps -> stderr
    void read();
};
struct FloodScratch {
    FloodHooks hooks;
    // static per (mesh, positions, ocean mask)
    bool staticValid = false; int32_t staticN = -1; int32_t L = 0; int64_t staticVersion = 0;
    hvec<int32_t> landCell, landIndex, offL, adjL, seedCell, landByR;   // landByR: land indices in ascending original id
    // per call (land-index space)
    bool landOrder = false;                  // the caller's height array holds the land cells only, element i = land index i (landCell order), instead of every cell by id
    hvec<float> surface, eL;                 // surface / root: compact copies of the pass-1 results for passes 2 and 3
    hvec<FloodCell> state;
    hvec<int32_t> root, order, order2, list2;
    hvec<uint32_t> bits, bits2;
    hvec<FloodHeapItem> heapStore;
    // landmasses of the compact land graph (static per mask): pass 1 walks them concurrently (flood_pass1_landmasses)
    hvec<int32_t> compSeeds;                 // seed positions (indices into seedCell) grouped by landmass, ascending inside
    std::vector<int32_t> compSeedStart;      // compSeeds range per seeded landmass, landmasses in descending size
    std::vector<int32_t> compSize;
    std::vector<int32_t> compCellStart;      // per seeded landmass: its cells in compCells (ascending original id inside)
    hvec<int32_t> compCells, seedLocal;      // seedLocal[s]: position of seed s among its landmass's seeds
    hvec<int32_t> stamp;                     // tie-family id of the claim that reached a cell (0 = outside any tie group)
    hvec<int32_t> localIdx;                  // pass 2 of a big tree: a cell's position in its tree's cell list (tree_pass2_chains)
    hvec<uint8_t> seen;                      // pass 1 of the landmass walks: the cell has been claimed (one byte per land cell)
    hvec<uint8_t> onPath;                    // pass 2: the cell lay on a carve path (only tracked when pass 1 left open parents)
    std::vector<hvec<FloodHeapItem>> workerHeaps;
    // replay of the single heap (flood_host.cc: replay_dirty_landmasses)
    hvec<uint8_t> replayDirty;               // the cell belongs to a landmass that is walked again inside the replay
    hvec<int32_t> childStart;                // per clean cell: its claimed children in childItem, in the order it pushed them
    hvec<FloodHeapItem> childItem;
};
// What the landmass-parallel pass 1 reports.  A cell is "contested" when two cells with EQUAL keys that sat in the heap
// together (and their sub-key cascades) both reached it: which one claims it is decided by the array history of the
// reference's single heap, which separate heaps cannot know.
struct FloodTieReport {
    int64_t groups = 0, nested = 0, contested = 0, openParents = 0, unresolved = 0;
    int32_t landmasses = 0, workers = 0, replayed = 0;   // replayed: landmasses decided by the replay of the single heap
    std::vector<std::pair<int32_t, int32_t>> alt;      // (cell, alternative parent) where only drainTo is undecided
};
// (re)builds the mask-dependent tables (Morton-ordered land list `landCell`, compact CSR, seeds)
// mortonAll (nullable): every cell in the order of morton_order_cells(N, xyz) — the land list is then a filter of it instead of a sort
void flood_build_static(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, const uint8_t* ocean, FloodScratch& S, const int32_t* mortonAll = nullptr);
// every cell in Morton order of its position (ties: ascending id); restricted to the land cells this is FloodScratch::landCell
void morton_order_cells(int32_t N, const float* xyz, hvec<int32_t>& cells);
// xyz (3*N floats) orders the compact land arrays spatially; may be nullptr (index order)
void priority_flood_carve_host(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, float* e,
                               const uint8_t* ocean, double carveStrength, FloodScratch& S);
// the pieces of the call above, used by the device flood (pass 1 on the GPU, passes 2/3 per drainage tree on the host)
int64_t flood_queues_differ(int64_t ops, uint64_t seed);            // test support: RingQueue against the 4-ary heap on a random operation sequence
void flood_cell_noise(const FloodScratch& S, double* out);            // cellNoise per land cell, compact order
void flood_gather(const float* e, FloodScratch& S);                   // land elevations -> compact arrays, pass-1 start state
void flood_pass1_host(FloodScratch& S);                               // serial heap walk (reference order incl. heap tie mechanics)
// One heap per landmass, landmasses concurrently; equal-key decisions that could differ from the single heap's are
// detected (FloodTieReport).  Returns false when one of them changes surfaces/keys: the caller redoes pass 1 serially.
bool flood_pass1_landmasses(FloodScratch& S, FloodTieReport& rep);
// The same per landmass, with passes 2/3 of a landmass starting as soon as its own pass 1 is over (the largest landmass
// bounds pass 1; the others' carving runs beside it).  Landmasses where an equal-key decision matters are walked again
// inside a replay of the reference's single heap (bare heap operations for everything else) and carved afterwards, so the
// call always ends with the reference's result in e.  FloodScratch's per-call arrays are consumed: flood_gather() before
// the next use.
// replayAllowed == false: a call that would need the replay returns false instead and leaves e untouched (the caller has a
// better heap to replay: flood_host_passes_exchange)
bool flood_landmass_pipeline(float* e, double carveStrength, FloodScratch& S, FloodTieReport& rep, int64_t& pathRedo, bool replayAllowed = true);
// pass 1 (landmass-parallel when exact, else the serial walk) + passes 2/3; stats: see flood_host.cc
struct FloodHostStats { int64_t calls = 0, serialPass1 = 0, tieGroups = 0, contested = 0, openParents = 0, unresolved = 0, pathRedo = 0, replays = 0, replayedLandmasses = 0; double pass1Ms = 0, pass23Ms = 0; };
void flood_host_passes(float* e, double carveStrength, FloodScratch& S, FloodHostStats* stats);
// The flood of ONE SHARE of a planet (landmass decomposition, decomposed.py: the other shares' landmasses are ocean to S).  A share's
// landmasses flood exactly as in the whole planet as long as no equal-key decision matters; when one does, only the reference's
// single heap over the WHOLE planet knows the answer (js/terrain-post.js:131-147), and that heap holds the other shares' cells at
// their CURRENT heights.  So the shares agree (phase 0: any share undecided?) and, if so, pool the heights their land cells have
// at this flood call (phase 1); an undecided share then floods the whole planet on the true mask — landmass pipeline + replay,
// exactly what the unpartitioned run does — and keeps its own cells of the result.
//   fn(user, 0, int32_t flag[1], 1)          flag := max over the shares                                  (collective)
//   fn(user, 1, float field[N], N)           in: own land cells hold their heights; out: every land cell (collective)
// Both return 0 on success.  trueOcean: the planet's real mask (N bytes); off / adj / xyz: the mesh (borrowed).
using FloodExchangeFn = int (*)(void* user, int32_t phase, void* buf, int64_t n);
struct FloodExchange {
    bool on = false;
    FloodExchangeFn fn = nullptr; void* user = nullptr;
    hvec<uint8_t> trueOcean;
    hvec<float> snapshot;
    FloodScratch global;                     // tables of the true mask, built when first needed (on a rank that floods the whole planet)
    // the planet's land cells in ascending id (true mask) are what the flooding rank hands back: positions of this rank's cells in that list
    hvec<int32_t> ownPos; hvec<float> landPack; int64_t landTotal = -1, posVersion = -1; int32_t minOwnCell = -1;
    int64_t calls = 0, gathers = 0, globalFloods = 0, received = 0;
};
// returns 0, or the non-zero status of fn
int flood_host_passes_exchange(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, float* e, double carveStrength,
                               FloodScratch& S, FloodHostStats* stats, FloodExchange& X);
void flood_import_pass1(const int32_t* par, const float* surface, const int32_t* root, FloodScratch& S);
// openAlt: cells whose parent pass 1 could not decide between equal keys (same surface either way); returns false when
// the carve / fix-up result depends on that choice (nothing is written back to e then)
bool flood_pass23_host(float* e, double carveStrength, FloodScratch& S, const std::vector<std::pair<int32_t, int32_t>>* openAlt = nullptr);

// plates_host.cc — js/plates.js:241-348 (r_plate rewritten in place; plateSeeds in the Set's iteration order)
void smooth_reconnect_plates_host(int32_t N, const int32_t* off, const int32_t* adj, int32_t* r_plate, int32_t numSeeds,
                                  const int32_t* plateSeeds, int32_t numPasses);

// error slot used by every extern "C" entry point (thread-local)
void set_error(const std::string& msg);

}  // namespace wo
