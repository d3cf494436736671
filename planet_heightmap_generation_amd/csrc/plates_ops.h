// Plate projection bodies (WO_HD: device kernels and the test-only CPU emulator drive the same code).
// Reference: js/coarse-plates.js:51-117 projectCoarsePlates.
#pragma once
#include <cstdint>
#include <cmath>

#include "noise.h"

namespace wo {

struct CoarsePlates {
    int32_t NC;                 // coarseMesh.numRegions
    const int32_t* off;         // coarseMesh.adjOffset [NC+1]
    const int32_t* adj;         // coarseMesh.adjList
    const float* xyz;           // coarse_xyz [3*NC]
    const int32_t* plate;       // coarse_r_plate [NC]
    const int32_t* grid;        // start cells: nearest coarse region to the centre of each (z, longitude) bucket
    int32_t gridZ, gridLon;     // bucket counts
};

// bucket of a unit vector: equal-area bands in z, equal sectors in longitude
WO_HD inline int32_t plate_grid_bucket(int32_t gridZ, int32_t gridLon, double x, double y, double z) {
    int32_t iz = (int32_t)((z + 1.0) * 0.5 * gridZ);
    if (iz < 0) iz = 0;
    if (iz >= gridZ) iz = gridZ - 1;
    int32_t il = (int32_t)((atan2(y, x) + 3.141592653589793) * (0.5 / 3.141592653589793) * gridLon);
    if (il < 0) il = 0;
    if (il >= gridLon) il = gridLon - 1;
    return iz * gridLon + il;
}
WO_HD inline void plate_grid_centre(int32_t gridZ, int32_t gridLon, int32_t b, double& x, double& y, double& z) {
    const int32_t iz = b / gridLon, il = b % gridLon;
    z = (iz + 0.5) / gridZ * 2.0 - 1.0;
    const double lon = (il + 0.5) / gridLon * 2.0 * 3.141592653589793 - 3.141592653589793;
    const double rr = sqrt(1.0 - z * z);
    x = rr * cos(lon); y = rr * sin(lon);
}

// the perturbed lookup point of hi-res cell (ox, oy, oz)            :69-86
WO_HD inline void plate_lookup_point(const uint8_t* P, const uint8_t* M, double ox, double oy, double oz, double perturbAmp,
                                     double& px, double& py, double& pz) {
    double dx = 0, dy = 0, dz = 0, amp = perturbAmp, freq = 8;
    for (int oct = 0; oct < 4; ++oct) {
        dx += noise3d(P, M, ox * freq, oy * freq, oz * freq) * amp;
        dy += noise3d(P, M, ox * freq + 100, oy * freq + 100, oz * freq + 100) * amp;
        dz += noise3d(P, M, ox * freq + 200, oy * freq + 200, oz * freq + 200) * amp;
        amp *= 0.5; freq *= 2;
    }
    px = ox + dx; py = oy + dy; pz = oz + dz;
    double len = sqrt(px * px + py * py + pz * pz);
    if (len == 0 || len != len) len = 1;
    px /= len; py /= len; pz /= len;
}

// Greedy ascent of p·c over the coarse Delaunay graph (:88-103).  The reference warm-starts from the previous cell's
// answer and falls back to a brute-force scan when a walk reaches ceil(sqrt(NC)) steps (:106-111); on a Delaunay
// graph the ascent has a single local maximum, the nearest site, so any start gives the region the reference finds.
// The neighbour scan keeps the reference's form (bounds fixed at entry, `cur` may move inside the scan).
WO_HD inline int32_t plate_nearest_coarse(const CoarsePlates& C, double px, double py, double pz, int32_t cur) {
    double bestDot = px * C.xyz[3 * cur] + py * C.xyz[3 * cur + 1] + pz * C.xyz[3 * cur + 2];
    for (int32_t steps = 0; steps <= C.NC; ++steps) {
        bool improved = false;
        const int32_t iEnd = C.off[cur + 1];
        for (int32_t i = C.off[cur]; i < iEnd; ++i) {
            const int32_t nb = C.adj[i];
            const double d = px * C.xyz[3 * nb] + py * C.xyz[3 * nb + 1] + pz * C.xyz[3 * nb + 2];
            if (d > bestDot) { bestDot = d; cur = nb; improved = true; }
        }
        if (!improved) break;
    }
    return cur;
}

WO_HD inline int32_t plate_project_cell(const CoarsePlates& C, const uint8_t* P, const uint8_t* M, const float* r_xyz, int32_t r, double perturbAmp) {
    double px, py, pz;
    plate_lookup_point(P, M, r_xyz[3 * r], r_xyz[3 * r + 1], r_xyz[3 * r + 2], perturbAmp, px, py, pz);
    const int32_t start = C.grid ? C.grid[plate_grid_bucket(C.gridZ, C.gridLon, px, py, pz)] : 0;
    return C.plate[plate_nearest_coarse(C, px, py, pz, start)];
}

// start cell of one bucket: brute-force nearest coarse region to the bucket centre
WO_HD inline int32_t plate_grid_cell(const CoarsePlates& C, int32_t b) {
    double x, y, z;
    plate_grid_centre(C.gridZ, C.gridLon, b, x, y, z);
    int32_t best = 0; double bestDot = -2;
    for (int32_t c = 0; c < C.NC; ++c) {
        const double d = x * C.xyz[3 * c] + y * C.xyz[3 * c + 1] + z * C.xyz[3 * c + 2];
        if (d > bestDot) { bestDot = d; best = c; }
    }
    return best;
}

}  // namespace wo
