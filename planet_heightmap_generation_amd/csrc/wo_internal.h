// Internal declarations shared between the translation units of libworogen (not part of the C ABI).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace wo {

// mesh_builder.cc
void fib_sphere_points(int N, double jitter, double seed, float* xyz);
int sphere_delaunay(int V, const float* xyz, int* triangles, int* halfedges, std::string& err);
int mesh_csr(int V, int numSides, const int* triangles, const int* halfedges,
             int* adjOffset, int* adjList, int* adjTri, std::string& err);
void neighbor_dist(int V, const int* adjOffset, const int* adjList, const float* xyz, float* out);

// flood_host.cc
struct FloodHeapItem { float key; int32_t cell; };
struct FloodCell { float e; int32_t drain; };
struct FloodScratch {
    std::vector<int32_t> drainTo, path, order, order2, seedCell, seedTarget, root;
    std::vector<uint8_t> visited;
    std::vector<float> surface;
    std::vector<uint32_t> bits, bits2;
    std::vector<FloodHeapItem> heapStore;
    std::vector<FloodCell> state;
    bool staticValid = false;       // seed list valid for the current (mesh, r_isOcean)
    void ensure(int32_t N);
};
void priority_flood_carve_host(int32_t N, const int32_t* off, const int32_t* adj, float* e,
                               const uint8_t* ocean, double carveStrength, FloodScratch& S);

// error slot used by every extern "C" entry point (thread-local)
void set_error(const std::string& msg);

}  // namespace wo
