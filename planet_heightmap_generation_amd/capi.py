"""ctypes binding of libworogen's C ABI (include/worogen.h).

Only plumbing lives here: it loads the in-tree shared library, declares argument types for every
symbol the header exports and converts a non-zero status into ``WorogenError`` (the reference reports
failures as JS exceptions, js/planet-worker.js:336-338).  There is no CPU fallback: if the library or a
HIP device is missing the calls raise.

Load order: PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  A process that uses both torch's
GPU side and this library must `import torch` first (bench.py does); the other way round torch finds the system runtime
already mapped and reports "No HIP GPUs are available".  This library works with either runtime.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ["WO_LIBWOROGEN"]) if os.environ.get("WO_LIBWOROGEN") else _PKG_DIR / "libworogen.so"      # (override: A/B builds of the library in experiments)


class WorogenError(RuntimeError):
    pass


# name -> (restype, argtypes); mirrors include/worogen.h one to one
_c_i32, _c_i64, _c_f64 = C.c_int32, C.c_int64, C.c_double
_p = C.c_void_p
SIGNATURES = {
    "wo_abi_version": (C.c_int, []),
    "wo_last_error": (C.c_char_p, []),
    "wo_device_count": (C.c_int, []),
    "wo_fib_sphere_points": (C.c_int, [_c_i32, _c_f64, _c_f64, _p]),
    "wo_sphere_delaunay": (C.c_int, [_c_i32, _p, _p, _p]),
    "wo_mesh_csr": (C.c_int, [_c_i32, _c_i32, _p, _p, _p, _p, _p]),
    "wo_neighbor_dist": (C.c_int, [_c_i32, _p, _p, _p, _p]),
    "wo_triangle_elevations": (C.c_int, [_c_i32, _p, _p, _p]),
    "wo_noise_tables": (C.c_int, [_c_f64, _p, _p]),
    "wo_noise_point": (C.c_int, [_p, _p, _c_i32, _c_i32, _c_f64, _c_f64, _c_f64, _c_f64, _c_f64, _c_f64, _p]),
    "wo_noise_eval": (C.c_int, [_p, _c_f64, _c_i32, _c_i32, _c_f64, _c_f64, _c_f64, _c_i64, _p, _p]),
    "wo_ctx_create": (_p, [_c_i32]),
    "wo_ctx_destroy": (None, [_p]),
    "wo_planet_create": (_p, [_p, _c_i32, _p, _p, _p, _p]),
    "wo_planet_destroy": (None, [_p]),
    "wo_planet_num_regions": (_c_i32, [_p]),
    "wo_warp_terrain": (C.c_int, [_p, _p, _c_f64, _c_f64, _p]),
    "wo_smooth_elevation": (C.c_int, [_p, _p, _p, _c_i32, _c_f64]),
    "wo_erode_composite": (C.c_int, [_p, _p, _p, _c_i32, _c_f64, _c_f64, _c_f64, _c_i32, _c_f64, _c_f64, _c_i32, _c_f64]),
    "wo_sharpen_ridges": (C.c_int, [_p, _p, _p, _c_i32, _c_f64]),
    "wo_soil_creep": (C.c_int, [_p, _p, _p, _c_i32, _c_f64]),
    "wo_assign_elevation": (C.c_int, [_p, _p, _p, _p, _c_i32, _p, _p, _p, _p, _c_f64, _c_f64, _c_f64, _p, _p, _p, _p, _p, _p, _p]),
    "wo_smooth_field": (C.c_int, [_p, _p, _c_i32]),
    "wo_project_coarse_plates": (C.c_int, [_p, _c_i32, _p, _p, _p, _p, _c_f64, _c_i32, _p]),
    "wo_smooth_reconnect_plates": (C.c_int, [_c_i32, _p, _p, _p, _p, _c_i32, _c_i32]),
    "wo_land_components": (C.c_int, [_c_i32, _p, _p, _p, _p]),
    "wo_diffuse_ocean_warmth": (C.c_int, [_p, _p, _p, _p, _c_i32, _p]),
    "wo_wind_convergence": (C.c_int, [_p, _p, _p, _p, _p]),
    "wo_advect_moisture": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _c_i32, _p]),
    "wo_planet_upload": (C.c_int, [_p, _p, _p]),
    "wo_planet_download": (C.c_int, [_p, _p]),
    "wo_planet_set_halo": (C.c_int, [_p, _p, _c_i32, _p, _c_i32]),
    "wo_planet_pack_halo": (C.c_int, [_p, _p, _p]),
    "wo_planet_unpack_halo": (C.c_int, [_p, _p, _p]),
    "wo_comm_unique_id": (C.c_int, [_p]),
    "wo_comm_create": (C.c_int, [_p, _p, _c_i32, _c_i32, _p]),
    "wo_comm_destroy": (C.c_int, [_p]),
    "wo_comm_rank": (C.c_int, [_p]),
    "wo_comm_size": (C.c_int, [_p]),
    "wo_planet_exchange_allgather": (C.c_int, [_p, _p, _p]),
    "wo_planet_exchange_neighbors": (C.c_int, [_p, _p, _c_i32, _c_i32]),
    "wo_planet_set_flood_exchange": (C.c_int, [_p, _p, _p, _p]),
    "wo_planet_set_flood_exchange_comm": (C.c_int, [_p, _p, _p, _p, _p]),
    "wo_planet_ocean_from_elevation": (C.c_int, [_p]),
    "wo_planet_download_ocean": (C.c_int, [_p, _p]),
    "wo_planet_sync": (C.c_int, [_p]),
    "wo_planet_save_state": (C.c_int, [_p]),
    "wo_planet_restore_state": (C.c_int, [_p]),
    "wo_warp_terrain_resident": (C.c_int, [_p, _c_f64, _c_f64, _c_i32]),
    "wo_planet_upload_hotspot": (C.c_int, [_p, _p]),
    "wo_smooth_elevation_resident": (C.c_int, [_p, _c_i32, _c_f64]),
    "wo_erode_composite_resident": (C.c_int, [_p, _c_i32, _c_f64, _c_f64, _c_f64, _c_i32, _c_f64, _c_f64, _c_i32, _c_f64]),
    "wo_sharpen_ridges_resident": (C.c_int, [_p, _c_i32, _c_f64]),
    "wo_soil_creep_resident": (C.c_int, [_p, _c_i32, _c_f64]),
    "wo_planet_synthetic_terrain": (C.c_int, [_p, _c_f64]),
    "wo_timer_start": (C.c_int, [_p]),
    "wo_timer_stop_ms": (C.c_int, [_p, _p]),
    "wo_profile_enable": (C.c_int, [_p, _c_i32]),
    "wo_profile_reset": (C.c_int, [_p]),
    "wo_profile_report": (C.c_int, [_p, _c_i32, _p, _p, _p, _p]),
    "wo_last_stage_timing": (C.c_int, [_p, _c_i32, _p, _p, _p]),
    "wo_last_erode_stats": (C.c_int, [_p, _c_i32, _p, _p, _p]),
}

# wo_flood_exchange_fn: int fn(void* user, int32_t phase, void* buf, int64_t n)
FLOOD_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64)
FLOOD_EXCHANGE_PROTOCOL = 2          # include/worogen.h: WO_FLOOD_EXCHANGE_PROTOCOL (the handshake of phase -1)

_lib = None
MISSING: list[str] = []


def lib() -> C.CDLL:
    """Load libworogen.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise WorogenError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the HIP path)")
        # RTLD_GLOBAL is not needed; keep symbols local to avoid clashing with torch's HIP runtime.
        _lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(_lib, name)
            except AttributeError:
                MISSING.append(name)   # tests assert this stays empty
                continue
            fn.restype = res
            fn.argtypes = args
    return _lib


def last_error() -> str:
    msg = lib().wo_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise WorogenError(f"{what} failed (status {rc}): {last_error()}")


def ptr(a: np.ndarray | None):
    """Pointer to a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("array must be C-contiguous")
    return a.ctypes.data_as(C.c_void_p)
