// Host-side state of assignElevation (see elevation_host.cc).
#pragma once
#include <cstdint>
#include <functional>
#include <vector>

#include "host_util.h"

#include "elevation_ops.h"

namespace wo {

struct ElevMesh { int32_t N; const int32_t* off; const int32_t* adj; const float* xyz; };

struct ElevInputs {
    const int32_t* plate;            // r_plate [N]
    PlateTable plates;               // host pointers, dense by plate id
    const int32_t* plateSeeds; int32_t numPlateSeeds;   // plateSeeds in Set iteration order
    const int32_t* superPlate;       // r_superPlate [N] or nullptr
    PlateTable superPlates;
    double seed, spread, noiseMag;
    NoiseTab hsNoise3;               // SimplexNoise(seed + 503), host tables
};

struct CollisionHost {               // one layer's findCollisions output, downloaded from the device
    std::vector<float> stress, subduct;
    std::vector<int8_t> btype;
    std::vector<uint8_t> bothOcean, hasOcean, setCode;
    void resize(int32_t N) { stress.resize(N); subduct.resize(N); btype.resize(N); bothOcean.resize(N); hasOcean.resize(N); setCode.resize(N); }
};

struct ElevHostState {
    hvec<float> stress, subduct;
    std::vector<int8_t> btype;
    hvec<uint8_t> bothOcean, hasOcean, isOcean, coastConvergent;
    std::vector<int32_t> mountain, coastline, ocean;         // Sets in insertion order
    hvec<float> distMountain, distOcean, distCoastline, distCoast, distCoastLand;
    hvec<float> dBdry, coastStressMax, coastSubductMax, riftDist, ridgeDist, fractureDist, backArcDist, backArcStress, arcDist, arcStress;
};

void blend_collision_layers(int32_t N, const CollisionHost& S, const CollisionHost* P, ElevHostState& H);
// bfsOnDevice: called once the scalars are known and BEFORE the distance fields: the caller produces the FIFO BFS fields on the device
// (elevation_bfs.h; H.dBdry ... H.arcStress stay empty) while the distance fields run here.  Arguments: the scalars, maxCD, maxStress.
void elevation_host_stage(const ElevMesh& M, const ElevInputs& I, const CollisionHost& S, const CollisionHost* P,
                          ElevHostState& H, ElevParams& Q, std::vector<Dome>& domes,
                          const std::function<void(const ElevParams&, int32_t, double)>& bfsOnDevice);

}  // namespace wo
