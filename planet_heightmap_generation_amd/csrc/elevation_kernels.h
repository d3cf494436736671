// gfx950 kernels of assignElevation (bodies in elevation_ops.h).  Included by planet.hip only.
#pragma once
#include <hip/hip_runtime.h>

#include "device.h"
#include "elevation_ops.h"

namespace wo {

// nine SimplexNoise instances live in LDS (1 KiB each): order = EL_TAB_*
enum { EL_TAB_NOISE = 0, EL_TAB_RIFT, EL_TAB_FOLD, EL_TAB_C1, EL_TAB_C2, EL_TAB_C3, EL_TAB_ARC, EL_TAB_HS1, EL_TAB_HS2, EL_TAB_COUNT };

__global__ __launch_bounds__(WO_BLOCK) void k_collision(int32_t N, const int32_t* off, const int32_t* adj, const float* xyz,
                                                         const int32_t* plate, PlateTable T, const uint8_t* table, CollisionOut O) {
    __shared__ uint8_t sP[512], sM[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) { sP[i] = table[i]; sM[i] = table[512 + i]; }
    __syncthreads();
    NoiseTab nt{sP, sM};
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < N; r += gridDim.x * blockDim.x)
        collision_cell(r, N, off, adj, xyz, plate, T, nt, O);
}

// main loop + coastal roughening + island arcs + hotspots + compression: every stage only reads and writes
// the cell's own elevation, so the five reference loops fuse into one pass (one read of each input field).
__global__ __launch_bounds__(WO_BLOCK) void k_elevation(ElevFields F, ElevParams Q, PlateTable T, const uint8_t* tables, const Dome* domes) {
    __shared__ uint8_t sT[EL_TAB_COUNT * 1024];
    for (int i = threadIdx.x; i < EL_TAB_COUNT * 1024; i += blockDim.x) sT[i] = tables[i];
    __syncthreads();
    auto tab = [&](int k) { return NoiseTab{sT + k * 1024, sT + k * 1024 + 512}; };
    for (int32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < Q.N; r += gridDim.x * blockDim.x) {
        float e = elevation_main_cell(F, Q, T, r, tab(EL_TAB_NOISE), tab(EL_TAB_RIFT), tab(EL_TAB_FOLD));
        e = coastal_cell(F, Q, r, e, tab(EL_TAB_NOISE), tab(EL_TAB_C1), tab(EL_TAB_C2), tab(EL_TAB_C3));
        e = arc_cell(F, Q, r, e, tab(EL_TAB_ARC));
        e = hotspot_cell(F, Q, r, e, domes, tab(EL_TAB_HS1), tab(EL_TAB_HS2));
        F.elev[r] = compress_cell(e);
    }
}

}  // namespace wo
