// Internal declarations shared between the translation units of libworogen (not part of the C ABI).
#pragma once
#include <cstdint>
#include <string>

namespace wo {

// mesh_builder.cc
void fib_sphere_points(int N, double jitter, double seed, float* xyz);
int sphere_delaunay(int V, const float* xyz, int* triangles, int* halfedges, std::string& err);
int mesh_csr(int V, int numSides, const int* triangles, const int* halfedges,
             int* adjOffset, int* adjList, int* adjTri, std::string& err);
void neighbor_dist(int V, const int* adjOffset, const int* adjList, const float* xyz, float* out);

// error slot used by every extern "C" entry point (thread-local)
void set_error(const std::string& msg);

}  // namespace wo
