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
struct FloodHeapItem { float key; int32_t cell; };
struct FloodCell { float e; int32_t drain; float surface; int32_t root; };   // one 16-byte record per land cell: a pop touches one line for all four
struct FloodScratch {
    // static per (mesh, positions, ocean mask)
    bool staticValid = false; int32_t staticN = -1; int32_t L = 0; int64_t staticVersion = 0;
    hvec<int32_t> landCell, landIndex, offL, adjL, seedCell, landByR;   // landByR: land indices in ascending original id
    // per call (land-index space)
    hvec<float> surface, eL;                 // surface / root: compact copies of the pass-1 results for passes 2 and 3
    hvec<FloodCell> state;
    hvec<int32_t> root, order, order2, list2;
    hvec<uint32_t> bits, bits2;
    hvec<FloodHeapItem> heapStore;
};
// (re)builds the mask-dependent tables (Morton-ordered land list `landCell`, compact CSR, seeds)
void flood_build_static(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, const uint8_t* ocean, FloodScratch& S);
// xyz (3*N floats) orders the compact land arrays spatially; may be nullptr (index order)
void priority_flood_carve_host(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz, float* e,
                               const uint8_t* ocean, double carveStrength, FloodScratch& S);
// the pieces of the call above, used by the device flood (pass 1 on the GPU, passes 2/3 per drainage tree on the host)
void flood_cell_noise(const FloodScratch& S, double* out);            // cellNoise per land cell, compact order
void flood_gather(const float* e, FloodScratch& S);                   // land elevations -> compact arrays, pass-1 start state
void flood_pass1_host(FloodScratch& S);                               // serial heap walk (reference order incl. heap tie mechanics)
void flood_import_pass1(const int32_t* par, const float* surface, const int32_t* root, FloodScratch& S);
void flood_pass23_host(float* e, double carveStrength, FloodScratch& S);

// plates_host.cc — js/plates.js:241-348 (r_plate rewritten in place; plateSeeds in the Set's iteration order)
void smooth_reconnect_plates_host(int32_t N, const int32_t* off, const int32_t* adj, int32_t* r_plate, int32_t numSeeds,
                                  const int32_t* plateSeeds, int32_t numPasses);

// error slot used by every extern "C" entry point (thread-local)
void set_error(const std::string& msg);

}  // namespace wo
