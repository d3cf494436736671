/*
 * worogen.h — C ABI of libworogen, the MI355X-native implementation of World Orogen's per-cell
 * terrain pipeline (reference: raguilar011095/planet_heightmap_generation, js/terrain-post.js,
 * js/simplex-noise.js, js/rng.js, js/sphere-mesh.js).
 *
 * This is the drop-in boundary: plain pointers and sizes, no torch / HIP types.  The N-API shim
 * (planet_heightmap_generation_amd/napi/worogen_napi.cc) and the Python ctypes loader
 * (planet_heightmap_generation_amd/capi.py) bind exactly these symbols.  Every entry point cites the
 * reference interface it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - every function returning int returns 0 on success, non-zero on failure; the message is then
 *     available from wo_last_error() (thread-local, valid until the next failing call on the thread).
 *     The JS/Python hosts turn a non-zero status into a thrown Error/exception, which is how the
 *     reference reports failures (exceptions caught by js/planet-worker.js:336-338).
 *   - host pointers are borrowed for the duration of the call only.
 *   - "numRegions" is the reference's mesh.numRegions (requested N + 1: the pole cell is appended,
 *     js/sphere-mesh.js:179-184).  CSR arrays are mesh.adjOffset (numRegions+1) / mesh.adjList.
 *   - device entry points fail (never fall back to the CPU) when no HIP device is usable.
 *   - threading: a wo_ctx (device + stream) and the planets created on it belong to one host thread at a time, like
 *     the reference's single worker.  Different contexts may be driven from different threads concurrently (the
 *     library keeps no shared mutable state); several planets in flight on one GPU this way raise throughput ~2x
 *     (DESIGN.md section 7).
 *
 * Environment (every variable the library reads; read once per API call or flood call, never inside a pass)
 *   sizing       WO_HOST_THREADS=<n>        workers of the host-side parallel sweeps (default: the process's CPUs, <= 64)
 *                WO_FLOOD_THREADS=<n>       workers of the priority flood's per-landmass walks (default 24)
 *                WO_FLOOD_PIN=1             the walk of the largest landmass keeps its CPU and the other flood workers keep off its L3
 *                                           (the library touches no thread affinity unless asked)
 *   routes       WO_LAYOUT=index            erodeComposite on the planet's own cell order instead of the land-first Morton mirror
 *   (same bits)  WO_TILE_LDS=1              receivers / thermal passes stage their tile's neighbour window in LDS (measured no faster)
 *                WO_FLOOD=device            pass 1 of the flood as the device label-correcting fixed point (exact, ~16x slower than the host walk)
 *                WO_FLOOD_HOST=two-phase    host flood: pass 1 of all landmasses, then passes 2 / 3 (default: pipelined per landmass)
 *                WO_FLOOD_HOST=serial       host flood: the reference's single heap, one thread (WO_FLOOD_HOST is read once per process)
 *   relaxed      WO_RELAXED=full            NOT the reference's semantics: one sort per flood, affine solve, Jacobi carve
 *   (labelled)   WO_RELAXED_SORT_EVERY=<k>  NOT the reference's semantics: landCells re-sorted every k-th iteration only
 *   diagnostics  WO_FLOOD_TIMING=1          laps of the flood stage and of a new terrain's set-up -> stderr
 *                WO_STAGE_TIMING=all        wo_last_stage_timing brackets every iteration instead of every 8th
 *                WO_ELEV_TIMING=1           laps of assignElevation's host stage -> stderr
 *   tests only   WO_TEST_HOOKS=k=v,k=v      the test suite's hooks (csrc/host_util.h: forced undecided landmasses, replay cuts, a wrong basin
 *                                           layout, a carve launch that gives up, ...); never set in production
 * (The Python loader additionally honours WO_LIBWOROGEN=<path to the .so>; bench.py has WO_BENCH_* switches of its own.)
 */
#ifndef WOROGEN_H
#define WOROGEN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WO_ABI_VERSION 1

typedef struct wo_ctx wo_ctx;       /* one HIP device + stream + scratch pools                     */
typedef struct wo_planet wo_planet; /* device-resident mesh + fields; mirrors the worker's retained
                                       state W (js/planet-worker.js:22,277-292)                     */

/* ---------------------------------------------------------------- status ---------------------- */
int         wo_abi_version(void);
const char* wo_last_error(void);
/* number of usable HIP devices (0 when none / no driver); never fails */
int         wo_device_count(void);

/* ------------------------------------------------ host-side input producers (no GPU needed) --- */
/* generateFibonacciSphere + pole append: js/sphere-mesh.js:9-37,179-181.  r_xyz has 3*(N+1) floats;
 * `seed` is what the caller passes to makeRng (js/rng.js:3). */
int wo_fib_sphere_points(int32_t N, double jitter, double seed, float* r_xyz);
/* Spherical Delaunay of numRegions unit vectors == the reference's stereographic Delaunator run +
 * addPoleToMesh (js/sphere-mesh.js:41-90,174-186).  triangles / halfedges have 3*(2*numRegions-4)
 * entries, counter-clockwise seen from outside, same contract as Delaunator's arrays. */
int wo_sphere_delaunay(int32_t numRegions, const float* r_xyz, int32_t* triangles, int32_t* halfedges);
/* SphereMesh constructor's CSR (js/sphere-mesh.js:94-146).  adjTriList may be NULL. */
int wo_mesh_csr(int32_t numRegions, int32_t numSides, const int32_t* triangles, const int32_t* halfedges,
                int32_t* adjOffset, int32_t* adjList, int32_t* adjTriList);
/* computeNeighborDist: js/sphere-mesh.js:191-203 */
int wo_neighbor_dist(int32_t numRegions, const int32_t* adjOffset, const int32_t* adjList,
                     const float* r_xyz, float* neighborDist);
/* computeTriangleElevations: js/planet-worker.js:29-37 */
int wo_triangle_elevations(int32_t numTriangles, const int32_t* triangles, const float* r_elevation,
                           float* t_elevation);

/* ------------------------------------------------ SimplexNoise (js/simplex-noise.js:5-54) ----- */
/* constructor: perm[512] and permMod12[512] for makeRng(seed) (js/simplex-noise.js:8-14) */
int wo_noise_tables(double seed, uint8_t* perm512, uint8_t* pm12_512);
/* one point on the host, for callers that evaluate the noise a point at a time (js/wind.js:394, js/coarse-plates.js:57):
 * perm512 / pm12_512 are the instance's tables; kind / octaves / p0..p2 as for wo_noise_eval.  Same arithmetic as the
 * device passes (csrc/noise.h). */
int wo_noise_point(const uint8_t* perm512, const uint8_t* pm12_512, int32_t kind, int32_t octaves, double p0, double p1, double p2,
                   double x, double y, double z, double* out);
/* batch evaluation on the device.  kind: 0 noise3D, 1 fbm(octaves, persistence),
 * 2 ridgedFbm(octaves, lacunarity=p0, gain=p1, offset=p2).  xyz: n interleaved double triples. */
#define WO_NOISE_3D 0
#define WO_NOISE_FBM 1
#define WO_NOISE_RIDGED 2
int wo_noise_eval(wo_ctx* ctx, double seed, int32_t kind, int32_t octaves, double p0, double p1, double p2,
                  int64_t n, const double* xyz, double* out);

/* ------------------------------------------------ context / planet handle --------------------- */
wo_ctx* wo_ctx_create(int32_t device);              /* NULL on failure (see wo_last_error)        */
void    wo_ctx_destroy(wo_ctx* ctx);
/* Uploads the mesh once (the worker keeps mesh / r_xyz / neighborDist in W between commands,
 * js/planet-worker.js:277-292).  neighborDist may be NULL: it is then computed on the device. */
wo_planet* wo_planet_create(wo_ctx* ctx, int32_t numRegions, const int32_t* adjOffset,
                            const int32_t* adjList, const float* r_xyz, const float* neighborDist);
void       wo_planet_destroy(wo_planet* p);

/* ------------------------------------------------ terrain-post, JS call surface --------------- */
/* mesh.numRegions the planet was created with (0 for a NULL planet) */
int32_t wo_planet_num_regions(const wo_planet* p);

/* Each mutates r_elevation (host, numRegions floats) in place and returns nothing else, exactly like
 * the five exports of js/terrain-post.js.  r_isOcean is numRegions bytes, read-only. */
/* warpTerrain(mesh, r_elevation, r_xyz, seed, strength, r_hotspot?)   js/terrain-post.js:233 */
int wo_warp_terrain(wo_planet* p, float* r_elevation, double seed, double strength, const float* r_hotspot);
/* smoothElevation(mesh, r_elevation, r_isOcean, iterations, strength) js/terrain-post.js:317 */
int wo_smooth_elevation(wo_planet* p, float* r_elevation, const uint8_t* r_isOcean, int32_t iterations, double strength);
/* erodeComposite(mesh, r_elevation, r_xyz, r_isOcean, hIters, K, m, dt, tIters, talusSlope,
 *                kThermal, gIters, glacialStrength, neighborDist)      js/terrain-post.js:369 */
int wo_erode_composite(wo_planet* p, float* r_elevation, const uint8_t* r_isOcean,
                       int32_t hIters, double K, double m, double dt,
                       int32_t tIters, double talusSlope, double kThermal,
                       int32_t gIters, double glacialStrength);
/* sharpenRidges(mesh, r_elevation, r_isOcean, iterations, strength)   js/terrain-post.js:713 */
int wo_sharpen_ridges(wo_planet* p, float* r_elevation, const uint8_t* r_isOcean, int32_t iterations, double strength);
/* applySoilCreep(mesh, r_elevation, r_isOcean, iterations, strength)  js/terrain-post.js:758 */
int wo_soil_creep(wo_planet* p, float* r_elevation, const uint8_t* r_isOcean, int32_t iterations, double strength);

/* ------------------------------------------------ assignElevation (js/elevation.js:216-1391) -- */
/* Plate tables are dense by plate id (the reference keys plateVec / plateDensity / plateIsOcean by plate id;
 * ids are coarse-mesh region indices for plates, 0..n-1 for super plates). */
typedef struct wo_plate_table {
    int32_t        numIds;   /* table length = max plate id + 1                              */
    const uint8_t* hasVec;   /* [numIds] plateVec[id] is defined                              */
    const double*  pole;     /* [3*numIds] plateVec[id].pole                                  */
    const double*  omega;    /* [numIds]   plateVec[id].omega                                 */
    const uint8_t* isOcean;  /* [numIds]   plateIsOcean.has(id)                               */
    const double*  density;  /* [numIds]   plateDensity[id]                                   */
} wo_plate_table;
/* assignElevation(mesh, r_xyz, plateIsOcean, r_plate, plateVec, plateSeeds, noise, noiseMag, seed, spread,
 *                 plateDensity, superPlateData)  ->  { r_elevation, mountain_r, coastline_r, ocean_r, r_stress,
 *                 debugLayers }                                                 js/elevation.js:216,1386-1390
 * plateSeeds is the Set in iteration order; r_superPlate / superPlates are NULL when superPlateData is null;
 * noisePerm512 / noisePm12_512 are the `noise` instance's tables; debugLayers (NULL or 12*numRegions floats,
 * layer-major) in the order base, tectonic, noise, interior, coastal, ocean, hotspot, tecActivity, margins,
 * backArc, foldRidge, orogenicPower; the three Sets come back in insertion order (arrays of numRegions ints,
 * sizes in setCounts[3]).  The resident r_elevation of the planet is set to the result as well. */
int wo_assign_elevation(wo_planet* p, const int32_t* r_plate, const wo_plate_table* plates,
                        const int32_t* plateSeeds, int32_t numPlateSeeds,
                        const int32_t* r_superPlate, const wo_plate_table* superPlates,
                        const uint8_t* noisePerm512, const uint8_t* noisePm12_512,
                        double noiseMag, double seed, double spread,
                        float* r_elevation, float* r_stress, float* debugLayers,
                        int32_t* mountain_r, int32_t* coastline_r, int32_t* ocean_r, int32_t* setCounts);

/* ------------------------------------------------ plate projection ----------- */
/* projectCoarsePlates(mesh, r_xyz, coarseMesh, coarse_xyz, coarse_r_plate, seed, numPlates) -> r_plate
 *                                                                       js/coarse-plates.js:51-117
 * Runs on the planet's resident r_xyz; the coarse mesh (coarseMesh.adjOffset / adjList, coarse_xyz, coarse_r_plate,
 * as generateCoarsePlates returns them, js/coarse-plates.js:19-42) is borrowed for the call.  numPlates < 0 stands
 * for `numPlates == null`.  r_plate: numRegions ints, bit-exact plate ids. */
int wo_project_coarse_plates(wo_planet* p, int32_t coarseRegions, const int32_t* coarseAdjOffset, const int32_t* coarseAdjList,
                             const float* coarse_xyz, const int32_t* coarse_r_plate, double seed, int32_t numPlates, int32_t* r_plate);
/* smoothAndReconnectPlates(mesh, r_plate, plateSeeds, numPasses)         js/plates.js:241-348
 * Host stage (order-defined in-place passes); needs no GPU.  plateSeeds in the Set's iteration order; r_plate is
 * rewritten in place. */
int wo_smooth_reconnect_plates(int32_t numRegions, const int32_t* adjOffset, const int32_t* adjList, int32_t* r_plate,
                               const int32_t* plateSeeds, int32_t numPlateSeeds, int32_t numPasses);

/* diffuseOceanWarmth(mesh, r_oceanWarmth, r_isLand, r_plateContinentality, passes)   js/temperature.js:19-66
 * Seeds the ocean cells with their warmth, then `passes` Jacobi sweeps of (self + neighbours) / (1 + degree); cells
 * with plate continentality >= 0.95 keep their value.  r_oceanWarmth / r_plateContinentality may be NULL (the reference
 * accepts null for both).  `out` (numRegions floats) receives the Float32Array the reference returns. */
int wo_diffuse_ocean_warmth(wo_planet* planet, const float* r_oceanWarmth, const uint8_t* r_isLand, const float* r_plateContinentality,
                            int32_t passes, float* out);
/* computeWindConvergence(mesh, r_xyz, r_wind3dX, r_wind3dY, r_wind3dZ)               js/precipitation.js:18-52
 * r_xyz is the planet's resident position array. */
int wo_wind_convergence(wo_planet* planet, const float* r_wind3dX, const float* r_wind3dY, const float* r_wind3dZ, float* out);
/* advectMoisture(mesh, r_xyz, r_heightKm, r_isLand, r_windE, r_windN, r_wind3dX, r_wind3dY, r_wind3dZ, r_oceanWarmth,
 *                r_coastDistLand, maxHops, avgEdgeKm)                                js/precipitation.js:59-195
 * Start moisture per cell, then maxHops upwind-gather sweeps (ping-pong buffers); avgEdgeKm is not read by the
 * reference's body and is not part of this entry point.  r_oceanWarmth may be NULL.
 * Exactness: the depletion base 1 - pow(0.78, 1/maxHops) (js/precipitation.js:113) is evaluated with the host libm's pow,
 * which agrees with V8's Math.pow bit for bit for maxHops 1..26 — this covers the reference's own clamp of maxHops to
 * 8..20 (js/precipitation.js:210) — and differs by 1 ulp at maxHops = 27, 98, 169 (checked for 1..200).  Bit-exact results
 * are guaranteed for the reference's range only. */
int wo_advect_moisture(wo_planet* planet, const float* r_heightKm, const uint8_t* r_isLand, const float* r_windE, const float* r_windN,
                       const float* r_wind3dX, const float* r_wind3dY, const float* r_wind3dZ, const float* r_oceanWarmth,
                       const int32_t* r_coastDistLand, int32_t maxHops, float* out);

/* ------------------------------------------------ multi-GPU exchange over RCCL --- */
/* The reference is one single-threaded worker (js/planet-worker.js:944-954): it has no exchange to replace.  A multi-GPU
 * host runs one process (or worker thread) per GPU; these entry points keep the data path on the devices — RCCL over xGMI on
 * the planet's own stream.  Rank 0 obtains the 128-byte id, the HOST distributes it by whatever channel it has (a message to
 * its workers, torch.distributed, a file), every rank creates its communicator with it (collective call).
 *   wo_planet_exchange_allgather  every rank contributes the values of its send list (wo_planet_set_halo) and receives all
 *                                 the others' (its receive list = the others' contributions in rank order): the merge of the
 *                                 landmass decomposition.  counts[j] = length of rank j's send list.  ncclAllGather.
 *   wo_planet_exchange_neighbors  the first nToPrev entries of the send list go to rank-1, the rest to rank+1; the first
 *                                 nFromPrev entries of the receive list come from rank-1, the rest from rank+1: the one-ring
 *                                 halo of a band decomposition.  ncclSend / ncclRecv in one group.
 * Both return when the received values are in the resident field. */
#define WO_COMM_ID_BYTES 128
typedef struct wo_comm wo_comm;
int wo_comm_unique_id(uint8_t* id);                                                    /* WO_COMM_ID_BYTES bytes out */
int wo_comm_create(wo_ctx* ctx, const uint8_t* id, int32_t nranks, int32_t rank, wo_comm** out);
int wo_comm_destroy(wo_comm* comm);
int wo_comm_rank(const wo_comm* comm);
int wo_comm_size(const wo_comm* comm);
int wo_planet_exchange_allgather(wo_planet* planet, wo_comm* comm, const int32_t* counts);
int wo_planet_exchange_neighbors(wo_planet* planet, wo_comm* comm, int32_t nToPrev, int32_t nFromPrev);

/* The flood exchange of the landmass decomposition.  priorityFloodCarve pops ONE heap over the whole planet
 * (js/terrain-post.js:131-147); how that heap orders EQUAL keys depends on everything in it, the other ranks' landmasses at
 * their current heights included.  A rank's flood proves for each of its landmasses that no equal-key decision matters, or
 * reports it undecided (csrc/flood_host.cc).  With an exchange set, every flood call of erodeComposite then (phase 0) agrees
 * with the other ranks whether any rank is undecided and, if so, (phase 1) pools the heights of all land cells at that call;
 * ONE undecided rank (the flag is a bid: INT32_MAX - its lowest land cell id, 0 when decided; the maximum names it) floods the
 * whole planet on the true mask exactly as the unpartitioned run does and (phases 2 / 3) hands the land heights back; the
 * undecided ranks keep their own cells of them.  Never needed at 10 M cells; at 40 M cells in every call.
 * All ranks make the same sequence of calls (0; then 1 and one of 2 / 3 when the maximum is not 0).
 *   fn(user, 0, int32_t flag[1], 1)        flag := max over the ranks
 *   fn(user, 1, float field[numRegions], numRegions)   in: the rank's own land cells hold their heights; out: every land cell does
 *   fn(user, 2, float land[n], n)          this rank flooded: it SENDS land (the planet's land cells in ascending id) to every rank
 *   fn(user, 3, float land[n], n)          every other rank: land := what the one rank in phase 2 sent
 *   fn(user, -1, int32_t proto[1], 1)      handshake, once, inside wo_planet_set_flood_exchange: proto[0] arrives as
 *                                          WO_FLOOD_EXCHANGE_PROTOCOL; a callback that implements exactly these phases sets
 *                                          proto[0] = -proto[0] and returns 0 (anything else: set_flood_exchange fails)
 * fn returns 0 on success; a non-zero status fails the erodeComposite call.  fn MUST return non-zero for a phase it does
 * not implement (never guess from "phase != 0").  r_isOcean_true: the planet's real mask (the
 * resident mask is the rank's: other ranks' landmasses are ocean).  fn == NULL switches the exchange off.
 * wo_planet_set_flood_exchange_comm: the same over RCCL — counts[j] land cells of rank j, their region ids concatenated in
 * rank order in cellsByRank (ncclAllReduce of the flag, ncclAllGather of the heights, ncclBroadcast of the flooded heights).
 * The comm must outlive the exchange: switch it off (fn == NULL) or destroy the planet before wo_comm_destroy. */
#define WO_FLOOD_EXCHANGE_PROTOCOL 2
typedef int (*wo_flood_exchange_fn)(void* user, int32_t phase, void* buf, int64_t n);
int wo_planet_set_flood_exchange(wo_planet* planet, const uint8_t* r_isOcean_true, wo_flood_exchange_fn fn, void* user);
int wo_planet_set_flood_exchange_comm(wo_planet* planet, const uint8_t* r_isOcean_true, wo_comm* comm, const int32_t* counts,
                                      const int32_t* cellsByRank);

/* ------------------------------------------------ landmass decomposition --------- */
/* Connected components of the land cells (cells with r_isOcean == 0, joined along mesh edges).  label[r] = smallest
 * region id of r's landmass, -1 for ocean cells.  Every order-defined pass of erodeComposite (js/terrain-post.js:369-707:
 * flood, sort, receivers, flow, implicit solve + deposition, thermal, glacial) and applySoilCreep (:758-794) only ever
 * couples a land cell to land cells of its own landmass (rivers and talus do not cross water), so landmasses are the
 * exact domain decomposition of the stack: a rank erodes the planet with every other rank's landmasses masked as ocean
 * and the land elevations are merged once at the end (planet_heightmap_generation_amd/decomposed.py).  Host stage
 * (concurrent union-find); needs no GPU. */
int wo_land_components(int32_t numRegions, const int32_t* adjOffset, const int32_t* adjList, const uint8_t* r_isOcean,
                       int32_t* label);

/* ------------------------------------------------ climate-util --------------- */
/* smoothField(mesh, field, passes)                                       js/climate-util.js:5-25
 * `passes` Jacobi sweeps of (self + neighbours) / (1 + degree) on a caller-owned Float32Array (numRegions floats,
 * rewritten in place); the planet's resident elevation is untouched. */
int wo_smooth_field(wo_planet* p, float* field, int32_t passes);

/* ------------------------------------------------ device-resident variants -------------------- */
/* The "reapply" pattern (js/planet-worker.js:341-440): fields stay in HBM, only scalars arrive.
 * wo_planet_upload sets the resident r_elevation (and r_isOcean when not NULL); the *_resident
 * functions are the same passes as above without the H2D/D2H copies; all are asynchronous on the
 * planet's stream until wo_planet_sync / wo_planet_download. */
int wo_planet_upload(wo_planet* p, const float* r_elevation, const uint8_t* r_isOcean);
int wo_planet_download(wo_planet* p, float* r_elevation);
/* Band-decomposed Jacobi passes (smoothElevation / applySoilCreep on one index band per GPU, banded.py): the planet is a
 * band plus its one-ring halo.  set_halo stores the local indices whose values go to / come from the neighbouring
 * bands; after each *_resident iteration pack gathers the resident r_elevation at the send indices (into hostOut, or
 * into the device buffer deviceOut that is handed to RCCL) and unpack scatters the received values to the receive
 * indices.  Exactly one of the host / device pointers is non-NULL. */
int wo_planet_set_halo(wo_planet* p, const int32_t* sendIdx, int32_t nSend, const int32_t* recvIdx, int32_t nRecv);
int wo_planet_pack_halo(wo_planet* p, float* hostOut, void* deviceOut);
int wo_planet_unpack_halo(wo_planet* p, const float* hostIn, const void* deviceIn);
/* r_isOcean[r] = r_elevation[r] <= 0 on the resident field (js/planet-worker.js:51-54) */
int wo_planet_ocean_from_elevation(wo_planet* p);
int wo_planet_download_ocean(wo_planet* p, uint8_t* r_isOcean);
int wo_planet_sync(wo_planet* p);
/* Keep / bring back a device-side copy of (r_elevation, r_isOcean): the worker's W.prePostElev that every
 * "reapply" starts from (js/planet-worker.js:353); D2D copies on the planet's stream. */
int wo_planet_save_state(wo_planet* p);
int wo_planet_restore_state(wo_planet* p);
int wo_warp_terrain_resident(wo_planet* p, double seed, double strength, int32_t useHotspot);
int wo_planet_upload_hotspot(wo_planet* p, const float* r_hotspot);
int wo_smooth_elevation_resident(wo_planet* p, int32_t iterations, double strength);
int wo_erode_composite_resident(wo_planet* p, int32_t hIters, double K, double m, double dt,
                                int32_t tIters, double talusSlope, double kThermal,
                                int32_t gIters, double glacialStrength);
int wo_sharpen_ridges_resident(wo_planet* p, int32_t iterations, double strength);
int wo_soil_creep_resident(wo_planet* p, int32_t iterations, double strength);
/* Synthetic bench terrain: e = 0.9*fbm(1.5p,5) - 0.12 + 0.25*ridged(3p,4)*max(0,fbm(1.5p,5))
 * with SimplexNoise(seed), written to the resident elevation; isOcean = e <= 0. */
int wo_planet_synthetic_terrain(wo_planet* p, double seed);

/* ------------------------------------------------ measurement ------------------------------- */
/* HIP-event stopwatch on the planet's stream (the stream every kernel of the path is launched on). */
int wo_timer_start(wo_planet* p);
int wo_timer_stop_ms(wo_planet* p, double* ms);   /* records, synchronises the stop event, returns elapsed */
/* Per-kernel-family HIP-event profiling: when enabled every launch is bracketed by events on the
 * planet's stream (slower; used by bench.py's roofline pass only). */
int wo_profile_enable(wo_planet* p, int32_t on);
int wo_profile_reset(wo_planet* p);
/* Fills up to cap entries; returns the number of families through *count. names[i] points to static
 * storage.  total_ms / launches are per family since the last reset. */
int wo_profile_report(wo_planet* p, int32_t cap, const char** names, double* total_ms, int64_t* launches,
                      int32_t* count);
/* Stage timings of the last erodeComposite call, same {stage, ms} shape the reference ships in
 * _postTiming (js/planet-worker.js:42-93).  Stages inside the iteration loop are bracketed by events on every
 * 8th iteration only and scaled to all of them (the brackets themselves cost stream time); setup and the
 * floods are timed exactly.  WO_STAGE_TIMING=all brackets every iteration. */
int wo_last_stage_timing(wo_planet* p, int32_t cap, const char** stages, double* ms, int32_t* count);
/* Counters of the last erodeComposite call (land cells, dependency rounds, ...), for DESIGN/bench. */
int wo_last_erode_stats(wo_planet* p, int32_t cap, const char** names, double* values, int32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* WOROGEN_H */
