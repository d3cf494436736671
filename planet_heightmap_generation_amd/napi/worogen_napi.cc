// N-API shim: the thin binding between the JavaScript host (planet_heightmap_generation_amd/js/*.js) and
// the C ABI of libworogen (include/worogen.h).  It only unwraps typed arrays into pointers, forwards the
// call and turns a non-zero status into a thrown JS Error carrying wo_last_error() — which is how the
// reference's worker expects failures to surface (try/catch around each handler, js/planet-worker.js:336-338).
// No computation happens here.  Built with plain g++ against /usr/include/node (N-API v8, Node >= 12).
#include <node_api.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/worogen.h"

namespace {

#define NAPI_OK(call)                                                          \
    do { if ((call) != napi_ok) { napi_throw_error(env, nullptr, "N-API call failed: " #call); return nullptr; } } while (0)

napi_value throw_wo(napi_env env, const char* what) {
    std::string msg = std::string(what) + ": " + wo_last_error();
    napi_throw_error(env, nullptr, msg.c_str());
    return nullptr;
}

struct Args {
    napi_env env; size_t argc = 16; napi_value argv[16];
    bool ok = true;
    Args(napi_env e, napi_callback_info info) : env(e) { ok = napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) == napi_ok; }
    bool has(size_t i) const {
        if (i >= argc) return false;
        napi_valuetype t; napi_typeof(env, argv[i], &t);
        return t != napi_undefined && t != napi_null;
    }
    double num(size_t i) { double v = 0; if (i < argc) napi_get_value_double(env, argv[i], &v); return v; }
    int32_t i32(size_t i) { return (int32_t)num(i); }
    // typed array of the expected element type; returns nullptr (and throws) on mismatch
    void* ta(size_t i, napi_typedarray_type want, size_t* len) {
        bool is = false;
        if (i >= argc || napi_is_typedarray(env, argv[i], &is) != napi_ok || !is) { napi_throw_type_error(env, nullptr, "expected a typed array"); ok = false; return nullptr; }
        napi_typedarray_type t; size_t n; void* data; napi_value ab; size_t off;
        napi_get_typedarray_info(env, argv[i], &t, &n, &data, &ab, &off);
        if (t != want) { napi_throw_type_error(env, nullptr, "typed array has the wrong element type"); ok = false; return nullptr; }
        if (len) *len = n;
        return data;
    }
    void* ext(size_t i) { void* p = nullptr; if (i < argc) napi_get_value_external(env, argv[i], &p); return p; }
};

napi_value make_ta(napi_env env, napi_typedarray_type t, size_t n, size_t elem, void** data) {
    napi_value ab, ta;
    if (napi_create_arraybuffer(env, n * elem, data, &ab) != napi_ok) return nullptr;
    if (napi_create_typedarray(env, t, n, ab, 0, &ta) != napi_ok) return nullptr;
    return ta;
}

void set_prop(napi_env env, napi_value obj, const char* k, napi_value v) { napi_set_named_property(env, obj, k, v); }

// ---- host-side producers -------------------------------------------------------------------------
napi_value FibSpherePoints(napi_env env, napi_callback_info info) {            // (N, jitter, seed) -> Float32Array(3*(N+1))
    Args a(env, info);
    const int32_t N = a.i32(0);
    void* d; napi_value out = make_ta(env, napi_float32_array, 3 * (size_t)(N + 1), 4, &d);
    if (!out) return nullptr;
    if (wo_fib_sphere_points(N, a.num(1), a.num(2), (float*)d)) return throw_wo(env, "fibSpherePoints");
    return out;
}

napi_value SphereDelaunay(napi_env env, napi_callback_info info) {             // (xyz) -> {triangles, halfedges}
    Args a(env, info);
    size_t n; float* xyz = (float*)a.ta(0, napi_float32_array, &n); if (!a.ok) return nullptr;
    const int32_t V = (int32_t)(n / 3);
    const size_t ns = 3 * (size_t)(2 * V - 4);
    void *t, *h; napi_value ta = make_ta(env, napi_int32_array, ns, 4, &t), ha = make_ta(env, napi_int32_array, ns, 4, &h);
    if (wo_sphere_delaunay(V, xyz, (int32_t*)t, (int32_t*)h)) return throw_wo(env, "sphereDelaunay");
    napi_value o; napi_create_object(env, &o); set_prop(env, o, "triangles", ta); set_prop(env, o, "halfedges", ha);
    return o;
}

napi_value MeshCsr(napi_env env, napi_callback_info info) {                    // (numRegions, triangles, halfedges)
    Args a(env, info);
    const int32_t V = a.i32(0);
    size_t ns, nh; int32_t* tri = (int32_t*)a.ta(1, napi_int32_array, &ns); int32_t* he = (int32_t*)a.ta(2, napi_int32_array, &nh);
    if (!a.ok) return nullptr;
    if (ns != nh || V < 0) { napi_throw_range_error(env, nullptr, "meshCsr: triangles and halfedges must have the same length"); return nullptr; }
    void *o1, *o2, *o3;
    napi_value off = make_ta(env, napi_int32_array, (size_t)V + 1, 4, &o1);
    std::vector<int32_t> adj(ns), adjt(ns);
    if (wo_mesh_csr(V, (int32_t)ns, tri, he, (int32_t*)o1, adj.data(), adjt.data())) return throw_wo(env, "meshCsr");
    const size_t E = (size_t)((int32_t*)o1)[V];
    napi_value al = make_ta(env, napi_int32_array, E, 4, &o2), at = make_ta(env, napi_int32_array, E, 4, &o3);
    std::memcpy(o2, adj.data(), E * 4); std::memcpy(o3, adjt.data(), E * 4);
    napi_value o; napi_create_object(env, &o);
    set_prop(env, o, "adjOffset", off); set_prop(env, o, "adjList", al); set_prop(env, o, "adjTriList", at);
    return o;
}

napi_value NeighborDist(napi_env env, napi_callback_info info) {               // (adjOffset, adjList, xyz)
    Args a(env, info);
    size_t no, na, nx; int32_t* off = (int32_t*)a.ta(0, napi_int32_array, &no); int32_t* adj = (int32_t*)a.ta(1, napi_int32_array, &na);
    float* xyz = (float*)a.ta(2, napi_float32_array, &nx); if (!a.ok) return nullptr;
    void* d; napi_value out = make_ta(env, napi_float32_array, na, 4, &d);
    if (wo_neighbor_dist((int32_t)no - 1, off, adj, xyz, (float*)d)) return throw_wo(env, "neighborDist");
    return out;
}

napi_value TriangleElevations(napi_env env, napi_callback_info info) {         // (triangles, r_elevation)
    Args a(env, info);
    size_t ns, ne; int32_t* tri = (int32_t*)a.ta(0, napi_int32_array, &ns); float* e = (float*)a.ta(1, napi_float32_array, &ne);
    if (!a.ok) return nullptr;
    void* d; napi_value out = make_ta(env, napi_float32_array, ns / 3, 4, &d);
    if (wo_triangle_elevations((int32_t)(ns / 3), tri, e, (float*)d)) return throw_wo(env, "triangleElevations");
    return out;
}

napi_value NoiseTables(napi_env env, napi_callback_info info) {                // (seed) -> {perm, pm12}
    Args a(env, info);
    void *p, *m; napi_value pa = make_ta(env, napi_uint8_array, 512, 1, &p), ma = make_ta(env, napi_uint8_array, 512, 1, &m);
    if (wo_noise_tables(a.num(0), (uint8_t*)p, (uint8_t*)m)) return throw_wo(env, "noiseTables");
    napi_value o; napi_create_object(env, &o); set_prop(env, o, "perm", pa); set_prop(env, o, "pm12", ma);
    return o;
}

// ---- handles ---------------------------------------------------------------------------------------
void FinalizeCtx(napi_env, void* data, void*) { wo_ctx_destroy((wo_ctx*)data); }
// wo_planet_destroy uses the planet's context (device, stream), and the order in which the garbage collector finalizes
// two externals is unspecified: the planet holds a strong reference to its context value and drops it only after the
// planet itself is gone.
// The external's data is a box, so that planetDestroy() can free the device memory NOW (a planet at 40 M cells holds several GB,
// and the external's tiny JS footprint gives the collector no reason to run) and leave a dead handle behind: the finalizer then
// only frees the box, and every entry point sees a null planet ("expected a live planet handle").
struct PlanetBox { wo_planet* p; };
void FinalizePlanet(napi_env env, void* data, void* hint) {
    PlanetBox* b = (PlanetBox*)data;
    if (b) { if (b->p) wo_planet_destroy(b->p); delete b; }
    if (hint) napi_delete_reference(env, (napi_ref)hint);
}
static wo_planet* planet_at(Args& a, size_t i) { PlanetBox* b = (PlanetBox*)a.ext(i); return b ? b->p : nullptr; }

napi_value DeviceCount(napi_env env, napi_callback_info) { napi_value v; napi_create_int32(env, wo_device_count(), &v); return v; }

napi_value CtxCreate(napi_env env, napi_callback_info info) {
    Args a(env, info);
    wo_ctx* c = wo_ctx_create(a.i32(0));
    if (!c) return throw_wo(env, "ctxCreate");
    napi_value v; NAPI_OK(napi_create_external(env, c, FinalizeCtx, nullptr, &v));
    return v;
}

napi_value PlanetCreate(napi_env env, napi_callback_info info) {               // (ctx, numRegions, adjOffset, adjList, xyz, neighborDist|null)
    Args a(env, info);
    wo_ctx* c = (wo_ctx*)a.ext(0);
    const int32_t V = a.i32(1);
    size_t no, na, nx, nd = 0;
    int32_t* off = (int32_t*)a.ta(2, napi_int32_array, &no); int32_t* adj = (int32_t*)a.ta(3, napi_int32_array, &na);
    float* xyz = (float*)a.ta(4, napi_float32_array, &nx); if (!a.ok) return nullptr;
    float* dist = a.has(5) ? (float*)a.ta(5, napi_float32_array, &nd) : nullptr; if (!a.ok) return nullptr;
    if (no != (size_t)V + 1 || nx != 3 * (size_t)V || (dist && nd != na)) { napi_throw_range_error(env, nullptr, "planetCreate: array sizes do not match numRegions"); return nullptr; }
    if (!c) { napi_throw_type_error(env, nullptr, "planetCreate: expected a context handle"); return nullptr; }
    wo_planet* p = wo_planet_create(c, V, off, adj, xyz, dist);
    if (!p) return throw_wo(env, "planetCreate");
    napi_ref ctxRef = nullptr;
    if (napi_create_reference(env, a.argv[0], 1, &ctxRef) != napi_ok) { wo_planet_destroy(p); napi_throw_error(env, nullptr, "planetCreate: cannot reference the context"); return nullptr; }
    napi_value v;
    PlanetBox* box = new PlanetBox{p};
    if (napi_create_external(env, box, FinalizePlanet, ctxRef, &v) != napi_ok) { wo_planet_destroy(p); delete box; napi_delete_reference(env, ctxRef); napi_throw_error(env, nullptr, "planetCreate: cannot wrap the planet"); return nullptr; }
    return v;
}
napi_value PlanetDestroy(napi_env env, napi_callback_info info) {               // (planet): frees the device memory now; the handle stays, dead
    Args a(env, info);
    PlanetBox* b = (PlanetBox*)a.ext(0);
    if (b && b->p) { wo_planet_destroy(b->p); b->p = nullptr; }
    napi_value u; napi_get_undefined(env, &u); return u;
}

// A planet argument must be a live handle, and every per-region array handed over with it must have exactly
// mesh.numRegions entries: the C ABI copies numRegions elements in and out of the pointer it is given (the reference
// would merely index `undefined`; here a short array would be an out-of-bounds read and write in the V8 heap).
bool planet_ok(napi_env env, wo_planet* p) {
    if (p) return true;
    napi_throw_type_error(env, nullptr, "expected a planet handle");
    return false;
}
bool regions_ok(napi_env env, wo_planet* p, size_t len, const char* what) {
    if ((int64_t)len == (int64_t)wo_planet_num_regions(p)) return true;
    std::string msg = std::string(what) + " length must equal mesh.numRegions";
    napi_throw_range_error(env, nullptr, msg.c_str());
    return false;
}

// ---- terrain-post, JS call surface (arrays mutated in place, return undefined) --------------------------
#define PLANET_AND_ELEV()                                                                        \
    Args a(env, info); wo_planet* p = planet_at(a, 0);                                      \
    if (!planet_ok(env, p)) return nullptr;                                                      \
    size_t ne; float* e = (float*)a.ta(1, napi_float32_array, &ne); if (!a.ok) return nullptr;   \
    if (!regions_ok(env, p, ne, "r_elevation")) return nullptr;

napi_value WarpTerrain(napi_env env, napi_callback_info info) {                // (planet, elev, seed, strength, hotspot|null)
    PLANET_AND_ELEV();
    size_t nh = 0; float* hot = a.has(4) ? (float*)a.ta(4, napi_float32_array, &nh) : nullptr; if (!a.ok) return nullptr;
    if (hot && !regions_ok(env, p, nh, "r_hotspot")) return nullptr;
    if (wo_warp_terrain(p, e, a.num(2), a.num(3), hot)) return throw_wo(env, "warpTerrain");
    return nullptr;
}
napi_value SmoothElevation(napi_env env, napi_callback_info info) {            // (planet, elev, isOcean, iterations, strength)
    PLANET_AND_ELEV();
    size_t no; uint8_t* oc = (uint8_t*)a.ta(2, napi_uint8_array, &no); if (!a.ok) return nullptr;
    if (no != ne) { napi_throw_range_error(env, nullptr, "r_isOcean length mismatch"); return nullptr; }
    if (wo_smooth_elevation(p, e, oc, a.i32(3), a.num(4))) return throw_wo(env, "smoothElevation");
    return nullptr;
}
napi_value SharpenRidges(napi_env env, napi_callback_info info) {
    PLANET_AND_ELEV();
    size_t no; uint8_t* oc = (uint8_t*)a.ta(2, napi_uint8_array, &no); if (!a.ok) return nullptr;
    if (no != ne) { napi_throw_range_error(env, nullptr, "r_isOcean length mismatch"); return nullptr; }
    if (wo_sharpen_ridges(p, e, oc, a.i32(3), a.num(4))) return throw_wo(env, "sharpenRidges");
    return nullptr;
}
napi_value ApplySoilCreep(napi_env env, napi_callback_info info) {
    PLANET_AND_ELEV();
    size_t no; uint8_t* oc = (uint8_t*)a.ta(2, napi_uint8_array, &no); if (!a.ok) return nullptr;
    if (no != ne) { napi_throw_range_error(env, nullptr, "r_isOcean length mismatch"); return nullptr; }
    if (wo_soil_creep(p, e, oc, a.i32(3), a.num(4))) return throw_wo(env, "applySoilCreep");
    return nullptr;
}
napi_value ErodeComposite(napi_env env, napi_callback_info info) {             // (planet, elev, isOcean, h,K,m,dt, t,talus,kT, g,gs)
    PLANET_AND_ELEV();
    size_t no; uint8_t* oc = (uint8_t*)a.ta(2, napi_uint8_array, &no); if (!a.ok) return nullptr;
    if (no != ne) { napi_throw_range_error(env, nullptr, "r_isOcean length mismatch"); return nullptr; }
    if (wo_erode_composite(p, e, oc, a.i32(3), a.num(4), a.num(5), a.num(6), a.i32(7), a.num(8), a.num(9), a.i32(10), a.num(11)))
        return throw_wo(env, "erodeComposite");
    return nullptr;
}

// ---- resident variants ("reapply": the field stays in HBM) ---------------------------------------------
napi_value PlanetUpload(napi_env env, napi_callback_info info) {               // (planet, elev|null, isOcean|null)
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    size_t n1 = 0, n2 = 0;
    float* e = a.has(1) ? (float*)a.ta(1, napi_float32_array, &n1) : nullptr; if (!a.ok) return nullptr;
    uint8_t* oc = a.has(2) ? (uint8_t*)a.ta(2, napi_uint8_array, &n2) : nullptr; if (!a.ok) return nullptr;
    if ((e && !regions_ok(env, p, n1, "r_elevation")) || (oc && !regions_ok(env, p, n2, "r_isOcean"))) return nullptr;
    if (wo_planet_upload(p, e, oc)) return throw_wo(env, "planetUpload");
    return nullptr;
}
napi_value PlanetDownload(napi_env env, napi_callback_info info) {             // (planet, elevOut)
    PLANET_AND_ELEV();
    if (wo_planet_download(p, e)) return throw_wo(env, "planetDownload");
    return nullptr;
}
napi_value PlanetDownloadOcean(napi_env env, napi_callback_info info) {        // (planet, isOceanOut)
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    size_t n; uint8_t* oc = (uint8_t*)a.ta(1, napi_uint8_array, &n); if (!a.ok) return nullptr;
    if (!regions_ok(env, p, n, "r_isOcean")) return nullptr;
    if (wo_planet_download_ocean(p, oc)) return throw_wo(env, "planetDownloadOcean");
    return nullptr;
}
napi_value PlanetUploadHotspot(napi_env env, napi_callback_info info) {
    PLANET_AND_ELEV();
    if (wo_planet_upload_hotspot(p, e)) return throw_wo(env, "planetUploadHotspot");
    return nullptr;
}
#define SIMPLE_PLANET_CALL(NAME, EXPR)                                                     \
    napi_value NAME(napi_env env, napi_callback_info info) {                               \
        Args a(env, info); wo_planet* p = planet_at(a, 0);                            \
        if (!planet_ok(env, p)) return nullptr;                                            \
        if (EXPR) return throw_wo(env, #NAME);                                             \
        return nullptr;                                                                    \
    }
SIMPLE_PLANET_CALL(PlanetOceanFromElevation, wo_planet_ocean_from_elevation(p))
SIMPLE_PLANET_CALL(PlanetSync, wo_planet_sync(p))
SIMPLE_PLANET_CALL(PlanetSaveState, wo_planet_save_state(p))
SIMPLE_PLANET_CALL(PlanetRestoreState, wo_planet_restore_state(p))
SIMPLE_PLANET_CALL(PlanetSyntheticTerrain, wo_planet_synthetic_terrain(p, a.num(1)))
SIMPLE_PLANET_CALL(WarpTerrainResident, wo_warp_terrain_resident(p, a.num(1), a.num(2), a.i32(3)))
SIMPLE_PLANET_CALL(SmoothElevationResident, wo_smooth_elevation_resident(p, a.i32(1), a.num(2)))
SIMPLE_PLANET_CALL(SharpenRidgesResident, wo_sharpen_ridges_resident(p, a.i32(1), a.num(2)))
SIMPLE_PLANET_CALL(ApplySoilCreepResident, wo_soil_creep_resident(p, a.i32(1), a.num(2)))
SIMPLE_PLANET_CALL(ErodeCompositeResident, wo_erode_composite_resident(p, a.i32(1), a.num(2), a.num(3), a.num(4), a.i32(5), a.num(6), a.num(7), a.i32(8), a.num(9)))
SIMPLE_PLANET_CALL(TimerStart, wo_timer_start(p))

napi_value TimerStopMs(napi_env env, napi_callback_info info) {
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    double ms = 0;
    if (wo_timer_stop_ms(p, &ms)) return throw_wo(env, "timerStopMs");
    napi_value v; napi_create_double(env, ms, &v); return v;
}

// [{stage, ms}] of the last erodeComposite — same shape as the reference's _postTiming entries
napi_value LastStageTiming(napi_env env, napi_callback_info info) {
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    const char* names[64]; double ms[64]; int32_t n = 0;
    if (wo_last_stage_timing(p, 64, names, ms, &n)) return throw_wo(env, "lastStageTiming");
    napi_value arr; napi_create_array_with_length(env, n, &arr);
    for (int32_t i = 0; i < n; ++i) {
        napi_value o, s, v; napi_create_object(env, &o);
        napi_create_string_utf8(env, names[i], NAPI_AUTO_LENGTH, &s); napi_create_double(env, ms[i], &v);
        set_prop(env, o, "stage", s); set_prop(env, o, "ms", v);
        napi_set_element(env, arr, i, o);
    }
    return arr;
}

// lastErodeStats(planet) -> {name: number}: counters of the last erodeComposite (rounds, launches, the host flood's route)
napi_value LastErodeStats(napi_env env, napi_callback_info info) {
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    const char* names[64]; double vals[64]; int32_t n = 0;
    if (wo_last_erode_stats(p, 64, names, vals, &n)) return throw_wo(env, "lastErodeStats");
    napi_value o; napi_create_object(env, &o);
    for (int32_t i = 0; i < n; ++i) { napi_value v; napi_create_double(env, vals[i], &v); napi_set_named_property(env, o, names[i], v); }
    return o;
}

// noisePoint(perm, pm12, kind, octaves, p0, p1, p2, x, y, z) -> number (host)
napi_value NoisePoint(napi_env env, napi_callback_info info) {
    Args a(env, info);
    size_t np_, nm;
    uint8_t* P = (uint8_t*)a.ta(0, napi_uint8_array, &np_); if (!a.ok) return nullptr;
    uint8_t* M = (uint8_t*)a.ta(1, napi_uint8_array, &nm); if (!a.ok) return nullptr;
    if (np_ != 512 || nm != 512) { napi_throw_range_error(env, nullptr, "noise tables must have 512 entries"); return nullptr; }
    double out = 0;
    if (wo_noise_point(P, M, a.i32(2), a.i32(3), a.num(4), a.num(5), a.num(6), a.num(7), a.num(8), a.num(9), &out)) return throw_wo(env, "noisePoint");
    napi_value v; napi_create_double(env, out, &v); return v;
}
napi_value NoiseEval(napi_env env, napi_callback_info info) {                  // (ctx, seed, kind, octaves, p0, p1, p2, xyz Float64Array) -> Float64Array
    Args a(env, info); wo_ctx* c = (wo_ctx*)a.ext(0);
    size_t n; double* xyz = (double*)a.ta(7, napi_float64_array, &n); if (!a.ok) return nullptr;
    void* d; napi_value out = make_ta(env, napi_float64_array, n / 3, 8, &d);
    if (wo_noise_eval(c, a.num(1), a.i32(2), a.i32(3), a.num(4), a.num(5), a.num(6), (int64_t)(n / 3), xyz, (double*)d)) return throw_wo(env, "noiseEval");
    return out;
}

// assignElevation(planet, r_plate, plates, plateSeeds, r_superPlate|null, superPlates|null, perm, pm12, noiseMag, seed, spread, wantDebug)
// plates = { numIds, hasVec:Uint8Array, pole:Float64Array, omega:Float64Array, isOcean:Uint8Array, density:Float64Array }
bool read_table(Args& a, napi_value obj, wo_plate_table* t) {
    napi_env env = a.env;
    auto get = [&](const char* k) { napi_value v; napi_get_named_property(env, obj, k, &v); return v; };
    double n = 0; napi_get_value_double(env, get("numIds"), &n);
    t->numIds = (int32_t)n;
    auto arr = [&](const char* k, napi_typedarray_type want, size_t need) -> void* {
        napi_value v = get(k); bool is = false; napi_is_typedarray(env, v, &is);
        if (!is) { napi_throw_type_error(env, nullptr, "plate table: expected typed arrays"); a.ok = false; return nullptr; }
        napi_typedarray_type ty; size_t len; void* data; napi_value ab; size_t off;
        napi_get_typedarray_info(env, v, &ty, &len, &data, &ab, &off);
        if (ty != want || len != need) { napi_throw_range_error(env, nullptr, "plate table: wrong array type or length"); a.ok = false; return nullptr; }
        return data;
    };
    const size_t m = (size_t)t->numIds;
    t->hasVec = (const uint8_t*)arr("hasVec", napi_uint8_array, m); if (!a.ok) return false;
    t->pole = (const double*)arr("pole", napi_float64_array, 3 * m); if (!a.ok) return false;
    t->omega = (const double*)arr("omega", napi_float64_array, m); if (!a.ok) return false;
    t->isOcean = (const uint8_t*)arr("isOcean", napi_uint8_array, m); if (!a.ok) return false;
    t->density = (const double*)arr("density", napi_float64_array, m); if (!a.ok) return false;
    return true;
}

napi_value AssignElevation(napi_env env, napi_callback_info info) {
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    size_t n, ns, nsup = 0, np_, nm;
    int32_t* r_plate = (int32_t*)a.ta(1, napi_int32_array, &n); if (!a.ok) return nullptr;
    if (!regions_ok(env, p, n, "r_plate")) return nullptr;
    wo_plate_table T{}, TS{};
    if (!read_table(a, a.argv[2], &T)) return nullptr;
    int32_t* seeds = (int32_t*)a.ta(3, napi_int32_array, &ns); if (!a.ok) return nullptr;
    int32_t* r_super = a.has(4) ? (int32_t*)a.ta(4, napi_int32_array, &nsup) : nullptr; if (!a.ok) return nullptr;
    const bool hasSuper = r_super != nullptr && a.has(5);
    if (hasSuper && !read_table(a, a.argv[5], &TS)) return nullptr;
    if (hasSuper && nsup != n) { napi_throw_range_error(env, nullptr, "r_superPlate length mismatch"); return nullptr; }
    uint8_t* perm = (uint8_t*)a.ta(6, napi_uint8_array, &np_); uint8_t* pm12 = (uint8_t*)a.ta(7, napi_uint8_array, &nm); if (!a.ok) return nullptr;
    if (np_ != 512 || nm != 512) { napi_throw_range_error(env, nullptr, "noise tables must have 512 entries"); return nullptr; }
    bool wantDebug = true; if (a.argc > 11) napi_get_value_bool(env, a.argv[11], &wantDebug);
    void *e, *st, *dl = nullptr;
    napi_value ea = make_ta(env, napi_float32_array, n, 4, &e), sa = make_ta(env, napi_float32_array, n, 4, &st), da = nullptr;
    if (wantDebug) da = make_ta(env, napi_float32_array, 12 * n, 4, &dl);
    std::vector<int32_t> mo(n), co(n), oc(n); int32_t cnt[3] = {0, 0, 0};
    if (wo_assign_elevation(p, r_plate, &T, seeds, (int32_t)ns, hasSuper ? r_super : nullptr, hasSuper ? &TS : nullptr, perm, pm12,
                            a.num(8), a.num(9), a.num(10), (float*)e, (float*)st, (float*)dl, mo.data(), co.data(), oc.data(), cnt))
        return throw_wo(env, "assignElevation");
    napi_value o; napi_create_object(env, &o);
    set_prop(env, o, "r_elevation", ea); set_prop(env, o, "r_stress", sa);
    if (da) set_prop(env, o, "debugLayers", da);
    const char* names[3] = {"mountain", "coastline", "ocean"}; std::vector<int32_t>* v[3] = {&mo, &co, &oc};
    for (int k = 0; k < 3; ++k) { void* d; napi_value t = make_ta(env, napi_int32_array, (size_t)cnt[k], 4, &d); std::memcpy(d, v[k]->data(), (size_t)cnt[k] * 4); set_prop(env, o, names[k], t); }
    return o;
}

// smoothField(planet, field Float32Array (in place), passes)
napi_value SmoothField(napi_env env, napi_callback_info info) {
    PLANET_AND_ELEV();
    if (wo_smooth_field(p, e, a.i32(2))) return throw_wo(env, "smoothField");
    return nullptr;
}
// optional Float32Array / Uint8Array / Int32Array of numRegions entries (null / undefined -> nullptr)
static void* opt_regions(Args& a, size_t i, napi_typedarray_type t, wo_planet* p, const char* what, bool required) {
    if (!a.has(i)) { if (required) { napi_throw_type_error(a.env, nullptr, what); a.ok = false; } return nullptr; }
    size_t n; void* d = a.ta(i, t, &n); if (!a.ok) return nullptr;
    if (!regions_ok(a.env, p, n, what)) { a.ok = false; return nullptr; }
    return d;
}
// diffuseOceanWarmth(planet, r_oceanWarmth|null, r_isLand, r_plateContinentality|null, passes) -> Float32Array
napi_value DiffuseOceanWarmth(napi_env env, napi_callback_info info) {
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    float* w = (float*)opt_regions(a, 1, napi_float32_array, p, "r_oceanWarmth", false); if (!a.ok) return nullptr;
    uint8_t* land = (uint8_t*)opt_regions(a, 2, napi_uint8_array, p, "r_isLand", true); if (!a.ok) return nullptr;
    float* c = (float*)opt_regions(a, 3, napi_float32_array, p, "r_plateContinentality", false); if (!a.ok) return nullptr;
    void* d; napi_value out = make_ta(env, napi_float32_array, (size_t)wo_planet_num_regions(p), 4, &d);
    if (wo_diffuse_ocean_warmth(p, w, land, c, a.i32(4), (float*)d)) return throw_wo(env, "diffuseOceanWarmth");
    return out;
}
// computeWindConvergence(planet, r_wind3dX, r_wind3dY, r_wind3dZ) -> Float32Array
napi_value WindConvergence(napi_env env, napi_callback_info info) {
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    float* x = (float*)opt_regions(a, 1, napi_float32_array, p, "r_wind3dX", true); if (!a.ok) return nullptr;
    float* y = (float*)opt_regions(a, 2, napi_float32_array, p, "r_wind3dY", true); if (!a.ok) return nullptr;
    float* z = (float*)opt_regions(a, 3, napi_float32_array, p, "r_wind3dZ", true); if (!a.ok) return nullptr;
    void* d; napi_value out = make_ta(env, napi_float32_array, (size_t)wo_planet_num_regions(p), 4, &d);
    if (wo_wind_convergence(p, x, y, z, (float*)d)) return throw_wo(env, "computeWindConvergence");
    return out;
}
// advectMoisture(planet, r_heightKm, r_isLand, r_windE, r_windN, r_wind3dX, r_wind3dY, r_wind3dZ, r_oceanWarmth|null, r_coastDistLand, maxHops) -> Float32Array
napi_value AdvectMoisture(napi_env env, napi_callback_info info) {
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    float* hk = (float*)opt_regions(a, 1, napi_float32_array, p, "r_heightKm", true); if (!a.ok) return nullptr;
    uint8_t* land = (uint8_t*)opt_regions(a, 2, napi_uint8_array, p, "r_isLand", true); if (!a.ok) return nullptr;
    float* we = (float*)opt_regions(a, 3, napi_float32_array, p, "r_windE", true); if (!a.ok) return nullptr;
    float* wn = (float*)opt_regions(a, 4, napi_float32_array, p, "r_windN", true); if (!a.ok) return nullptr;
    float* x = (float*)opt_regions(a, 5, napi_float32_array, p, "r_wind3dX", true); if (!a.ok) return nullptr;
    float* y = (float*)opt_regions(a, 6, napi_float32_array, p, "r_wind3dY", true); if (!a.ok) return nullptr;
    float* z = (float*)opt_regions(a, 7, napi_float32_array, p, "r_wind3dZ", true); if (!a.ok) return nullptr;
    float* w = (float*)opt_regions(a, 8, napi_float32_array, p, "r_oceanWarmth", false); if (!a.ok) return nullptr;
    int32_t* cd = (int32_t*)opt_regions(a, 9, napi_int32_array, p, "r_coastDistLand", true); if (!a.ok) return nullptr;
    void* d; napi_value out = make_ta(env, napi_float32_array, (size_t)wo_planet_num_regions(p), 4, &d);
    if (wo_advect_moisture(p, hk, land, we, wn, x, y, z, w, cd, a.i32(10), (float*)d)) return throw_wo(env, "advectMoisture");
    return out;
}
// landComponents(numRegions, adjOffset, adjList, r_isOcean) -> Int32Array (label = smallest id of the landmass, -1 for ocean)
napi_value LandComponents(napi_env env, napi_callback_info info) {
    Args a(env, info);
    size_t no, na, nc;
    int32_t* off = (int32_t*)a.ta(1, napi_int32_array, &no); if (!a.ok) return nullptr;
    int32_t* adj = (int32_t*)a.ta(2, napi_int32_array, &na); if (!a.ok) return nullptr;
    uint8_t* oc = (uint8_t*)a.ta(3, napi_uint8_array, &nc); if (!a.ok) return nullptr;
    const int32_t n = a.i32(0);
    if (n < 1 || (size_t)n + 1 != no || (size_t)n != nc) { napi_throw_range_error(env, nullptr, "mesh / r_isOcean length mismatch"); return nullptr; }
    void* d; napi_value out = make_ta(env, napi_int32_array, (size_t)n, 4, &d);
    if (wo_land_components(n, off, adj, oc, (int32_t*)d)) return throw_wo(env, "landComponents");
    return out;
}
// projectCoarsePlates(planet, coarseAdjOffset, coarseAdjList, coarse_xyz, coarse_r_plate, seed, numPlates|null) -> Int32Array
napi_value ProjectCoarsePlates(napi_env env, napi_callback_info info) {
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    size_t no, na, nx, np_;
    int32_t* off = (int32_t*)a.ta(1, napi_int32_array, &no); if (!a.ok) return nullptr;
    int32_t* adj = (int32_t*)a.ta(2, napi_int32_array, &na); if (!a.ok) return nullptr;
    float* xyz = (float*)a.ta(3, napi_float32_array, &nx); if (!a.ok) return nullptr;
    int32_t* pl = (int32_t*)a.ta(4, napi_int32_array, &np_); if (!a.ok) return nullptr;
    if (no < 2 || nx != 3 * (no - 1) || np_ != no - 1) { napi_throw_range_error(env, nullptr, "coarse mesh arrays do not match"); return nullptr; }
    const int32_t n = wo_planet_num_regions(p);
    if (n < 1) return throw_wo(env, "projectCoarsePlates");
    void* d; napi_value out = make_ta(env, napi_int32_array, (size_t)n, 4, &d);
    if (wo_project_coarse_plates(p, (int32_t)(no - 1), off, adj, xyz, pl, a.num(5), a.has(6) ? a.i32(6) : -1, (int32_t*)d)) return throw_wo(env, "projectCoarsePlates");
    return out;
}
// smoothAndReconnectPlates(numRegions, adjOffset, adjList, r_plate (in place), plateSeeds Int32Array, numPasses)
napi_value SmoothAndReconnectPlates(napi_env env, napi_callback_info info) {
    Args a(env, info);
    size_t no, na, nr, ns;
    int32_t* off = (int32_t*)a.ta(1, napi_int32_array, &no); if (!a.ok) return nullptr;
    int32_t* adj = (int32_t*)a.ta(2, napi_int32_array, &na); if (!a.ok) return nullptr;
    int32_t* rp = (int32_t*)a.ta(3, napi_int32_array, &nr); if (!a.ok) return nullptr;
    int32_t* seeds = (int32_t*)a.ta(4, napi_int32_array, &ns); if (!a.ok) return nullptr;
    const int32_t n = a.i32(0);
    if ((size_t)n + 1 != no || (size_t)n != nr) { napi_throw_range_error(env, nullptr, "mesh / r_plate length mismatch"); return nullptr; }
    if (wo_smooth_reconnect_plates(n, off, adj, rp, seeds, (int32_t)ns, a.i32(5))) return throw_wo(env, "smoothAndReconnectPlates");
    return nullptr;
}

// ---- multi-GPU exchange over RCCL (include/worogen.h: wo_comm_*; one worker thread per GPU holds one communicator) -------
void FinalizeComm(napi_env, void* data, void*) { wo_comm_destroy((wo_comm*)data); }
napi_value CommUniqueId(napi_env env, napi_callback_info) {                   // () -> Uint8Array(128): rank 0 makes it, the host posts it to every worker
    void* d; napi_value out = make_ta(env, napi_uint8_array, WO_COMM_ID_BYTES, 1, &d);
    if (wo_comm_unique_id((uint8_t*)d)) return throw_wo(env, "commUniqueId");
    return out;
}
napi_value CommCreate(napi_env env, napi_callback_info info) {                 // (ctx, id Uint8Array(128), nranks, rank) -> comm handle (collective)
    Args a(env, info);
    wo_ctx* c = (wo_ctx*)a.ext(0);
    size_t n; uint8_t* id = (uint8_t*)a.ta(1, napi_uint8_array, &n); if (!a.ok) return nullptr;
    if (!c || n != WO_COMM_ID_BYTES) { napi_throw_type_error(env, nullptr, "commCreate: expected (ctx, Uint8Array(128), nranks, rank)"); return nullptr; }
    wo_comm* cm = nullptr;
    if (wo_comm_create(c, id, a.i32(2), a.i32(3), &cm)) return throw_wo(env, "commCreate");
    napi_value v; NAPI_OK(napi_create_external(env, cm, FinalizeComm, nullptr, &v));
    return v;
}
napi_value PlanetSetHalo(napi_env env, napi_callback_info info) {              // (planet, sendIdx Int32Array, recvIdx Int32Array)
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    size_t ns, nr; int32_t* s = (int32_t*)a.ta(1, napi_int32_array, &ns); if (!a.ok) return nullptr;
    int32_t* r = (int32_t*)a.ta(2, napi_int32_array, &nr); if (!a.ok) return nullptr;
    if (wo_planet_set_halo(p, s, (int32_t)ns, r, (int32_t)nr)) return throw_wo(env, "planetSetHalo");
    return nullptr;
}
napi_value PlanetExchangeAllgather(napi_env env, napi_callback_info info) {    // (planet, comm, counts Int32Array(nranks))
    Args a(env, info); wo_planet* p = planet_at(a, 0); wo_comm* cm = (wo_comm*)a.ext(1);
    if (!planet_ok(env, p)) return nullptr;
    size_t n; int32_t* counts = (int32_t*)a.ta(2, napi_int32_array, &n); if (!a.ok) return nullptr;
    if (!cm || (int)n != wo_comm_size(cm)) { napi_throw_range_error(env, nullptr, "planetExchangeAllgather: one count per rank"); return nullptr; }
    if (wo_planet_exchange_allgather(p, cm, counts)) return throw_wo(env, "planetExchangeAllgather");
    return nullptr;
}
napi_value PlanetExchangeNeighbors(napi_env env, napi_callback_info info) {    // (planet, comm, nToPrev, nFromPrev)
    Args a(env, info); wo_planet* p = planet_at(a, 0); wo_comm* cm = (wo_comm*)a.ext(1);
    if (!planet_ok(env, p)) return nullptr;
    if (!cm) { napi_throw_type_error(env, nullptr, "planetExchangeNeighbors: expected a communicator handle"); return nullptr; }
    if (wo_planet_exchange_neighbors(p, cm, a.i32(2), a.i32(3))) return throw_wo(env, "planetExchangeNeighbors");
    return nullptr;
}

napi_value PlanetSetFloodExchange(napi_env env, napi_callback_info info) {    // (planet, trueOcean Uint8Array | null, comm, counts Int32Array(nranks), cellsByRank Int32Array)
    Args a(env, info); wo_planet* p = planet_at(a, 0);
    if (!planet_ok(env, p)) return nullptr;
    napi_valuetype t; napi_typeof(env, a.argv[1], &t);
    if (t == napi_null || t == napi_undefined) {               // exchange off
        if (wo_planet_set_flood_exchange(p, nullptr, nullptr, nullptr)) return throw_wo(env, "planetSetFloodExchange");
        return nullptr;
    }
    size_t nOc, nCounts, nCells;
    uint8_t* oc = (uint8_t*)a.ta(1, napi_uint8_array, &nOc); if (!a.ok) return nullptr;
    wo_comm* cm = (wo_comm*)a.ext(2);
    int32_t* counts = (int32_t*)a.ta(3, napi_int32_array, &nCounts); if (!a.ok) return nullptr;
    int32_t* cells = (int32_t*)a.ta(4, napi_int32_array, &nCells); if (!a.ok) return nullptr;
    if ((int32_t)nOc != wo_planet_num_regions(p)) { napi_throw_range_error(env, nullptr, "planetSetFloodExchange: r_isOcean length must equal mesh.numRegions"); return nullptr; }
    if (!cm || (int)nCounts != wo_comm_size(cm)) { napi_throw_range_error(env, nullptr, "planetSetFloodExchange: one count per rank"); return nullptr; }
    int64_t total = 0; for (size_t j = 0; j < nCounts; ++j) total += counts[j];
    if (total != (int64_t)nCells) { napi_throw_range_error(env, nullptr, "planetSetFloodExchange: cellsByRank must hold every rank's cells, in rank order"); return nullptr; }
    if (wo_planet_set_flood_exchange_comm(p, oc, cm, counts, cells)) return throw_wo(env, "planetSetFloodExchange");
    return nullptr;
}

napi_value Init(napi_env env, napi_value exports) {
    struct { const char* name; napi_callback fn; } fns[] = {
        {"fibSpherePoints", FibSpherePoints}, {"sphereDelaunay", SphereDelaunay}, {"meshCsr", MeshCsr}, {"neighborDist", NeighborDist},
        {"triangleElevations", TriangleElevations}, {"noiseTables", NoiseTables}, {"noiseEval", NoiseEval}, {"noisePoint", NoisePoint},
        {"deviceCount", DeviceCount}, {"ctxCreate", CtxCreate}, {"planetCreate", PlanetCreate}, {"planetDestroy", PlanetDestroy},
        {"warpTerrain", WarpTerrain}, {"smoothElevation", SmoothElevation}, {"erodeComposite", ErodeComposite},
        {"sharpenRidges", SharpenRidges}, {"applySoilCreep", ApplySoilCreep},
        {"planetUpload", PlanetUpload}, {"planetDownload", PlanetDownload}, {"planetDownloadOcean", PlanetDownloadOcean},
        {"planetUploadHotspot", PlanetUploadHotspot}, {"planetOceanFromElevation", PlanetOceanFromElevation}, {"planetSync", PlanetSync},
        {"planetSaveState", PlanetSaveState}, {"planetRestoreState", PlanetRestoreState}, {"planetSyntheticTerrain", PlanetSyntheticTerrain},
        {"warpTerrainResident", WarpTerrainResident}, {"smoothElevationResident", SmoothElevationResident},
        {"erodeCompositeResident", ErodeCompositeResident}, {"sharpenRidgesResident", SharpenRidgesResident},
        {"applySoilCreepResident", ApplySoilCreepResident}, {"timerStart", TimerStart}, {"timerStopMs", TimerStopMs},
        {"lastStageTiming", LastStageTiming}, {"lastErodeStats", LastErodeStats}, {"assignElevation", AssignElevation},
        {"projectCoarsePlates", ProjectCoarsePlates}, {"smoothField", SmoothField}, {"smoothAndReconnectPlates", SmoothAndReconnectPlates},
        {"diffuseOceanWarmth", DiffuseOceanWarmth}, {"computeWindConvergence", WindConvergence}, {"advectMoisture", AdvectMoisture},
        {"landComponents", LandComponents},
        {"commUniqueId", CommUniqueId}, {"commCreate", CommCreate}, {"planetSetHalo", PlanetSetHalo},
        {"planetExchangeAllgather", PlanetExchangeAllgather}, {"planetExchangeNeighbors", PlanetExchangeNeighbors}, {"planetSetFloodExchange", PlanetSetFloodExchange},
    };
    for (auto& f : fns) {
        napi_value v;
        if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &v) != napi_ok) return nullptr;
        napi_set_named_property(env, exports, f.name, v);
    }
    napi_value ver; napi_create_int32(env, wo_abi_version(), &ver); napi_set_named_property(env, exports, "abiVersion", ver);
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
