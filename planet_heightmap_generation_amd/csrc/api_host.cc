// extern "C" entry points that need no GPU: status, mesh producers, noise tables.
#include <cmath>
#include <cstring>
#include <string>

#include "../../include/worogen.h"
#include "host_util.h"
#include "wo_internal.h"
#include "noise.h"

namespace wo {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace wo

extern "C" {

int wo_abi_version(void) { return WO_ABI_VERSION; }
const char* wo_last_error(void) { return wo::g_err.c_str(); }

int wo_fib_sphere_points(int32_t N, double jitter, double seed, float* r_xyz) {
    if (N < 1 || !r_xyz) { wo::set_error("wo_fib_sphere_points: bad arguments"); return 1; }
    wo::fib_sphere_points(N, jitter, seed, r_xyz);
    return 0;
}

int wo_sphere_delaunay(int32_t numRegions, const float* r_xyz, int32_t* triangles, int32_t* halfedges) {
    if (!r_xyz || !triangles || !halfedges) { wo::set_error("wo_sphere_delaunay: null pointer"); return 1; }
    std::string err;
    int rc = wo::sphere_delaunay(numRegions, r_xyz, triangles, halfedges, err);
    if (rc) wo::set_error(err);
    return rc;
}

int wo_mesh_csr(int32_t numRegions, int32_t numSides, const int32_t* triangles, const int32_t* halfedges,
                int32_t* adjOffset, int32_t* adjList, int32_t* adjTriList) {
    if (!triangles || !halfedges || !adjOffset || !adjList) { wo::set_error("wo_mesh_csr: null pointer"); return 1; }
    std::string err;
    int rc = wo::mesh_csr(numRegions, numSides, triangles, halfedges, adjOffset, adjList, adjTriList, err);
    if (rc) wo::set_error(err);
    return rc;
}

int wo_neighbor_dist(int32_t numRegions, const int32_t* adjOffset, const int32_t* adjList,
                     const float* r_xyz, float* neighborDist) {
    if (!adjOffset || !adjList || !r_xyz || !neighborDist) { wo::set_error("wo_neighbor_dist: null pointer"); return 1; }
    wo::neighbor_dist(numRegions, adjOffset, adjList, r_xyz, neighborDist);
    return 0;
}

int wo_triangle_elevations(int32_t numTriangles, const int32_t* triangles, const float* r_elevation,
                           float* t_elevation) {
    if (!triangles || !r_elevation || !t_elevation) { wo::set_error("wo_triangle_elevations: null pointer"); return 1; }
    wo::parallel_ranges(numTriangles, [&](int64_t b, int64_t e, int) {
        for (int64_t t = b; t < e; ++t) {
            // (a + b + c) / 3 in double, stored as float32 (js/planet-worker.js:33-35)
            double s = (double)r_elevation[triangles[3 * t]] + (double)r_elevation[triangles[3 * t + 1]];
            s = s + (double)r_elevation[triangles[3 * t + 2]];
            t_elevation[t] = (float)(s / 3.0);
        }
    });
    return 0;
}

int wo_noise_tables(double seed, uint8_t* perm512, uint8_t* pm12_512) {
    if (!perm512 || !pm12_512) { wo::set_error("wo_noise_tables: null pointer"); return 1; }
    wo::noise_tables(seed, perm512, pm12_512);
    return 0;
}

int wo_noise_point(const uint8_t* perm512, const uint8_t* pm12_512, int32_t kind, int32_t octaves, double p0, double p1, double p2,
                   double x, double y, double z, double* out) {
    if (!perm512 || !pm12_512 || !out || kind < 0 || kind > 2) { wo::set_error("wo_noise_point: bad arguments"); return 1; }
    *out = kind == 0 ? wo::noise3d(perm512, pm12_512, x, y, z)
         : kind == 1 ? wo::fbm(perm512, pm12_512, x, y, z, octaves, p0)
                     : wo::ridged_fbm(perm512, pm12_512, x, y, z, octaves, p0, p1, p2);
    return 0;
}

int wo_smooth_reconnect_plates(int32_t numRegions, const int32_t* adjOffset, const int32_t* adjList, int32_t* r_plate,
                               const int32_t* plateSeeds, int32_t numPlateSeeds, int32_t numPasses) {
    if (numRegions < 1 || !adjOffset || !adjList || !r_plate || numPlateSeeds < 0 || (numPlateSeeds > 0 && !plateSeeds)) {
        wo::set_error("wo_smooth_reconnect_plates: bad arguments"); return 1;
    }
    try {
        wo::smooth_reconnect_plates_host(numRegions, adjOffset, adjList, r_plate, numPlateSeeds, plateSeeds, numPasses);
    } catch (const std::exception& e) { wo::set_error(std::string("wo_smooth_reconnect_plates: ") + e.what()); return 3; }
    return 0;
}

int wo_land_components(int32_t numRegions, const int32_t* adjOffset, const int32_t* adjList, const uint8_t* r_isOcean,
                       int32_t* label) {
    if (numRegions < 1 || !adjOffset || !adjList || !r_isOcean || !label) { wo::set_error("wo_land_components: bad arguments"); return 1; }
    if (adjOffset[0] != 0) { wo::set_error("wo_land_components: adjOffset[0] != 0"); return 1; }
    for (int32_t r = 0; r < numRegions; ++r) if (adjOffset[r + 1] < adjOffset[r]) { wo::set_error("wo_land_components: adjOffset is not monotone"); return 1; }
    for (int32_t i = 0; i < adjOffset[numRegions]; ++i) if (adjList[i] < 0 || adjList[i] >= numRegions) { wo::set_error("wo_land_components: adjList entry out of range"); return 1; }
    try {
        wo::mesh_components(numRegions, adjOffset, adjList, [&](int32_t r) { return r_isOcean[r] == 0; }, [](int32_t, int32_t) { return true; }, label);
        wo::parallel_ranges(numRegions, [&](int64_t b, int64_t e, int) { for (int64_t r = b; r < e; ++r) if (r_isOcean[r]) label[r] = -1; });
    } catch (const std::exception& e) { wo::set_error(std::string("wo_land_components: ") + e.what()); return 3; }
    return 0;
}

}  // extern "C"
