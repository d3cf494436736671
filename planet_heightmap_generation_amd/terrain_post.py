"""Python mirror of the reference's terrain-post call surface (js/terrain-post.js) over the C ABI.

Same function names (snake-cased), argument order and in-place mutation contract as the five exports
``warpTerrain / smoothElevation / erodeComposite / sharpenRidges / applySoilCreep``
(js/terrain-post.js:233,317,369,713,758): each takes the caller's ``r_elevation`` Float32 array and
mutates it in place, returns ``None`` and raises on failure (the reference throws).  All compute runs in
the HIP kernels of libworogen; there is no CPU path here.

``Planet`` is the device-resident handle (mirrors the worker's retained state ``W``,
js/planet-worker.js:277-292): mesh, r_xyz and neighborDist are uploaded once, the ``*_resident`` methods
keep the field in HBM between passes (the "reapply" pattern, js/planet-worker.js:341-440).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

_ctx_cache: dict[int, "Context"] = {}


class Context:
    def __init__(self, device: int = 0):
        L = capi.lib()
        self.device = device
        self.handle = L.wo_ctx_create(device)
        if not self.handle:
            raise capi.WorogenError(capi.last_error())

    def close(self):
        if self.handle:
            capi.lib().wo_ctx_destroy(self.handle)
            self.handle = None


def default_context(device: int = 0) -> Context:
    if device not in _ctx_cache:
        _ctx_cache[device] = Context(device)
    return _ctx_cache[device]


def _f32(a, n=None):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]):
        raise TypeError("expected a C-contiguous float32 numpy array (Float32Array in the reference)")
    if n is not None and a.size != n:
        raise ValueError(f"array has {a.size} elements, expected {n}")
    return a


def _u8(a, n):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.size != n:
        raise ValueError(f"r_isOcean has {a.size} elements, expected {n}")
    return a


class Comm:
    """RCCL communicator behind the C ABI (include/worogen.h: wo_comm_*): one per rank, created collectively from the 128-byte id
    that rank 0 obtains with Comm.unique_id() and the host distributes (bench.py: torch.distributed's object broadcast)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = np.zeros(128, np.uint8)
        capi.check(capi.lib().wo_comm_unique_id(capi.ptr(buf)), "wo_comm_unique_id")
        return buf.tobytes()

    def __init__(self, ctx: "Context", uid: bytes, size: int, rank: int):
        import ctypes as C
        if len(uid) != 128:
            raise ValueError("the communicator id is 128 bytes")
        idb = np.frombuffer(uid, np.uint8).copy()
        h = C.c_void_p()
        capi.check(capi.lib().wo_comm_create(ctx.handle, capi.ptr(idb), int(size), int(rank), C.byref(h)), "wo_comm_create")
        self.handle, self.size, self.rank, self.ctx = h, int(size), int(rank), ctx

    def close(self):
        if getattr(self, "handle", None):
            capi.lib().wo_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Planet:
    """Device-resident planet: CSR mesh + r_xyz + neighborDist, and the current r_elevation / r_isOcean."""

    def __init__(self, mesh, r_xyz, neighborDist=None, ctx: Context | None = None, device: int = 0):
        self.ctx = ctx or default_context(device)
        self.numRegions = int(mesh.numRegions)
        off = np.ascontiguousarray(mesh.adjOffset, dtype=np.int32)
        adj = np.ascontiguousarray(mesh.adjList, dtype=np.int32)
        xyz = np.ascontiguousarray(r_xyz, dtype=np.float32)
        if off.size != self.numRegions + 1 or xyz.size != 3 * self.numRegions:
            raise ValueError("mesh / r_xyz size mismatch")
        nd = None if neighborDist is None else np.ascontiguousarray(neighborDist, dtype=np.float32)
        if nd is not None and nd.size != adj.size:
            raise ValueError("neighborDist must be slot-aligned with adjList")
        self.handle = capi.lib().wo_planet_create(self.ctx.handle, self.numRegions, capi.ptr(off), capi.ptr(adj), capi.ptr(xyz),
                                                  capi.ptr(nd))
        if not self.handle:
            raise capi.WorogenError(capi.last_error())

    def close(self):
        if getattr(self, "handle", None):
            capi.lib().wo_planet_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- JS call surface (host array in, mutated in place) ----
    def warp_terrain(self, r_elevation, seed, strength, r_hotspot=None):
        e = _f32(r_elevation, self.numRegions)
        hot = None if r_hotspot is None else _f32(np.ascontiguousarray(r_hotspot, dtype=np.float32), self.numRegions)
        capi.check(capi.lib().wo_warp_terrain(self.handle, capi.ptr(e), float(seed), float(strength), capi.ptr(hot)), "warpTerrain")

    def smooth_elevation(self, r_elevation, r_isOcean, iterations, strength):
        e = _f32(r_elevation, self.numRegions)
        capi.check(capi.lib().wo_smooth_elevation(self.handle, capi.ptr(e), capi.ptr(_u8(r_isOcean, e.size)), int(iterations), float(strength)),
                   "smoothElevation")

    def _check_erode(self, rc: int) -> None:
        """capi.check for erodeComposite: a failure inside the flood-exchange callback (set_flood_exchange) is the cause, not just a status."""
        try:
            capi.check(rc, "erodeComposite")
        except capi.WorogenError as err:
            pending = getattr(self, "_flood_errors", None)
            if pending:
                cause = pending[-1]
                del pending[:]
                raise capi.WorogenError(f"{err} (flood exchange: {cause!r})") from cause
            raise

    def erode_composite(self, r_elevation, r_isOcean, hIters, K, m, dt, tIters, talusSlope, kThermal, gIters=0, glacialStrength=0.0):
        e = _f32(r_elevation, self.numRegions)
        self._check_erode(capi.lib().wo_erode_composite(self.handle, capi.ptr(e), capi.ptr(_u8(r_isOcean, e.size)), int(hIters), float(K), float(m),
                                                        float(dt), int(tIters), float(talusSlope), float(kThermal), int(gIters or 0),
                                                        float(glacialStrength or 0.0)))

    def sharpen_ridges(self, r_elevation, r_isOcean, iterations, strength):
        e = _f32(r_elevation, self.numRegions)
        capi.check(capi.lib().wo_sharpen_ridges(self.handle, capi.ptr(e), capi.ptr(_u8(r_isOcean, e.size)), int(iterations), float(strength)),
                   "sharpenRidges")

    def apply_soil_creep(self, r_elevation, r_isOcean, iterations, strength):
        e = _f32(r_elevation, self.numRegions)
        capi.check(capi.lib().wo_soil_creep(self.handle, capi.ptr(e), capi.ptr(_u8(r_isOcean, e.size)), int(iterations), float(strength)),
                   "applySoilCreep")

    # ---- resident variants ----
    def upload(self, r_elevation=None, r_isOcean=None):
        e = None if r_elevation is None else _f32(r_elevation, self.numRegions)
        oc = None if r_isOcean is None else _u8(r_isOcean, self.numRegions)
        capi.check(capi.lib().wo_planet_upload(self.handle, capi.ptr(e), capi.ptr(oc)), "wo_planet_upload")

    def download(self) -> np.ndarray:
        out = np.empty(self.numRegions, np.float32)
        capi.check(capi.lib().wo_planet_download(self.handle, capi.ptr(out)), "wo_planet_download")
        return out

    # ---- exchange over RCCL behind the C ABI (Comm below) ----
    def exchange_allgather(self, comm: "Comm", counts) -> None:
        """Every rank contributes the values of its send list and receives all the others' (set_halo lists): the landmass merge."""
        c = np.ascontiguousarray(counts, np.int32)
        if c.size != comm.size:
            raise ValueError("one count per rank")
        capi.check(capi.lib().wo_planet_exchange_allgather(self.handle, comm.handle, capi.ptr(c)), "wo_planet_exchange_allgather")

    def exchange_neighbors(self, comm: "Comm", n_to_prev: int, n_from_prev: int) -> None:
        """One-ring halo swap with rank - 1 / rank + 1 of a band chain (set_halo lists: [to prev | to next], [from prev | from next])."""
        capi.check(capi.lib().wo_planet_exchange_neighbors(self.handle, comm.handle, int(n_to_prev), int(n_from_prev)), "wo_planet_exchange_neighbors")

    def set_flood_exchange(self, true_ocean, exchange=None, comm: "Comm | None" = None, counts=None, cells_by_rank=None) -> None:
        """The flood exchange of the landmass decomposition (include/worogen.h: wo_planet_set_flood_exchange): every flood call of
        erodeComposite agrees with the other ranks whether any of them met an equal-key decision that matters and, if so, pools the
        heights of all land cells so that ONE undecided rank floods the whole planet like the unpartitioned run does and hands the
        result back.
        exchange: an object with allreduce_max(flag: int) -> int, allgather(field: np.ndarray[numRegions]) -> None (in place;
        own land cells valid on entry, every land cell on return) and broadcast(land: np.ndarray, sender: bool) -> None (in place:
        the one rank that flooded the whole planet sends its land heights, the others receive them) —
        decomposed.TorchFloodExchange / ThreadFloodExchange; or
        comm + counts + cells_by_rank: the same over RCCL behind the C ABI.  true_ocean None: off."""
        import ctypes as C
        L = capi.lib()
        if true_ocean is None:
            capi.check(L.wo_planet_set_flood_exchange(self.handle, None, None, None), "wo_planet_set_flood_exchange")
            self._flood_cb = None
            return
        oc = np.ascontiguousarray(true_ocean, np.uint8)
        if oc.size != self.numRegions:
            raise ValueError("true_ocean length must equal mesh.numRegions")
        if comm is not None:
            cnt = np.ascontiguousarray(counts, np.int32)
            cells = np.ascontiguousarray(cells_by_rank, np.int32)
            if cnt.size != comm.size or cells.size != int(cnt.sum()):
                raise ValueError("one count per rank, and the ranks' cells concatenated in rank order")
            capi.check(L.wo_planet_set_flood_exchange_comm(self.handle, capi.ptr(oc), comm.handle, capi.ptr(cnt), capi.ptr(cells)), "wo_planet_set_flood_exchange_comm")
            self._flood_cb = None
            return
        N = self.numRegions
        errors = []

        def _cb(_user, phase, buf, n):
            try:
                if phase == -1:             # handshake: this callback implements protocol 2 (phases 0-3) and nothing else
                    proto = C.cast(buf, C.POINTER(C.c_int32))
                    if int(proto[0]) != capi.FLOOD_EXCHANGE_PROTOCOL:
                        return 1
                    proto[0] = -int(proto[0])
                elif phase == 0:
                    flag = C.cast(buf, C.POINTER(C.c_int32))
                    flag[0] = int(exchange.allreduce_max(int(flag[0])))
                elif phase == 1:
                    field = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_float)), shape=(int(n),))
                    exchange.allgather(field)
                elif phase in (2, 3):       # 2: this rank flooded the whole planet and sends the land heights; 3: it receives them
                    land = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_float)), shape=(int(n),))
                    exchange.broadcast(land, phase == 2)
                else:
                    return 1                # a phase this callback does not know
                return 0
            except Exception as e:          # never let an exception cross the C boundary
                errors.append(e)
                return 1
        self._flood_cb = capi.FLOOD_EXCHANGE_FN(_cb)       # keep the thunk alive as long as the planet uses it
        self._flood_errors = errors
        capi.check(L.wo_planet_set_flood_exchange(self.handle, capi.ptr(oc), self._flood_cb, None), "wo_planet_set_flood_exchange")

    # ---- band-decomposed Jacobi passes (banded.py) ----
    def set_halo(self, send_idx, recv_idx):
        s = np.ascontiguousarray(send_idx, np.int32); r = np.ascontiguousarray(recv_idx, np.int32)
        self._halo = (int(s.size), int(r.size))
        capi.check(capi.lib().wo_planet_set_halo(self.handle, capi.ptr(s), s.size, capi.ptr(r), r.size), "wo_planet_set_halo")

    def pack_halo(self, device_ptr: int | None = None) -> np.ndarray | None:
        if device_ptr is not None:
            capi.check(capi.lib().wo_planet_pack_halo(self.handle, None, device_ptr), "wo_planet_pack_halo")
            return None
        out = np.empty(self._halo[0], np.float32)
        capi.check(capi.lib().wo_planet_pack_halo(self.handle, capi.ptr(out), None), "wo_planet_pack_halo")
        return out

    def unpack_halo(self, values: np.ndarray | None = None, device_ptr: int | None = None) -> None:
        if device_ptr is not None:
            capi.check(capi.lib().wo_planet_unpack_halo(self.handle, None, device_ptr), "wo_planet_unpack_halo")
            return
        v = np.ascontiguousarray(values, np.float32)
        if v.size != self._halo[1]:
            raise ValueError("halo value count mismatch")
        capi.check(capi.lib().wo_planet_unpack_halo(self.handle, capi.ptr(v), None), "wo_planet_unpack_halo")

    def download_ocean(self) -> np.ndarray:
        out = np.empty(self.numRegions, np.uint8)
        capi.check(capi.lib().wo_planet_download_ocean(self.handle, capi.ptr(out)), "wo_planet_download_ocean")
        return out

    def ocean_from_elevation(self):
        capi.check(capi.lib().wo_planet_ocean_from_elevation(self.handle), "wo_planet_ocean_from_elevation")

    def synthetic_terrain(self, seed):
        capi.check(capi.lib().wo_planet_synthetic_terrain(self.handle, float(seed)), "wo_planet_synthetic_terrain")

    def save_state(self):
        capi.check(capi.lib().wo_planet_save_state(self.handle), "wo_planet_save_state")

    def restore_state(self):
        capi.check(capi.lib().wo_planet_restore_state(self.handle), "wo_planet_restore_state")

    def sync(self):
        capi.check(capi.lib().wo_planet_sync(self.handle), "wo_planet_sync")

    def upload_hotspot(self, r_hotspot):
        capi.check(capi.lib().wo_planet_upload_hotspot(self.handle, capi.ptr(_f32(np.ascontiguousarray(r_hotspot, np.float32), self.numRegions))),
                   "wo_planet_upload_hotspot")

    def warp_terrain_resident(self, seed, strength, use_hotspot=False):
        capi.check(capi.lib().wo_warp_terrain_resident(self.handle, float(seed), float(strength), int(bool(use_hotspot))), "warpTerrain")

    def smooth_elevation_resident(self, iterations, strength):
        capi.check(capi.lib().wo_smooth_elevation_resident(self.handle, int(iterations), float(strength)), "smoothElevation")

    def erode_composite_resident(self, hIters, K, m, dt, tIters, talusSlope, kThermal, gIters=0, glacialStrength=0.0):
        self._check_erode(capi.lib().wo_erode_composite_resident(self.handle, int(hIters), float(K), float(m), float(dt), int(tIters), float(talusSlope),
                                                                 float(kThermal), int(gIters), float(glacialStrength)))

    def sharpen_ridges_resident(self, iterations, strength):
        capi.check(capi.lib().wo_sharpen_ridges_resident(self.handle, int(iterations), float(strength)), "sharpenRidges")

    def apply_soil_creep_resident(self, iterations, strength):
        capi.check(capi.lib().wo_soil_creep_resident(self.handle, int(iterations), float(strength)), "applySoilCreep")

    # ---- measurement ----
    def timer_start(self):
        capi.check(capi.lib().wo_timer_start(self.handle), "wo_timer_start")

    def timer_stop_ms(self) -> float:
        ms = C.c_double(0.0)
        capi.check(capi.lib().wo_timer_stop_ms(self.handle, C.byref(ms)), "wo_timer_stop_ms")
        return ms.value

    def profile_enable(self, on: bool):
        capi.check(capi.lib().wo_profile_enable(self.handle, int(on)), "wo_profile_enable")

    def profile_reset(self):
        capi.check(capi.lib().wo_profile_reset(self.handle), "wo_profile_reset")

    def _named(self, fn, with_counts=False):
        cap = 64
        names = (C.c_char_p * cap)()
        vals = (C.c_double * cap)()
        cnt = C.c_int32(0)
        if with_counts:
            launches = (C.c_int64 * cap)()
            capi.check(fn(self.handle, cap, names, vals, launches, C.byref(cnt)), "report")
            return {names[i].decode(): (vals[i], int(launches[i])) for i in range(cnt.value)}
        capi.check(fn(self.handle, cap, names, vals, C.byref(cnt)), "report")
        return {names[i].decode(): vals[i] for i in range(cnt.value)}

    def profile_report(self) -> dict:
        """{kernel family: (total ms, launches)} since the last reset (HIP events on the planet's stream)."""
        return self._named(capi.lib().wo_profile_report, with_counts=True)

    def last_stage_timing(self) -> dict:
        return self._named(capi.lib().wo_last_stage_timing)

    def last_erode_stats(self) -> dict:
        return self._named(capi.lib().wo_last_erode_stats)


# ---- module-level functions with the reference's signatures (mesh first) ----
_planet_cache: dict[int, tuple] = {}


def _planet_for(mesh, r_xyz=None, neighborDist=None) -> Planet:
    """One resident planet per mesh object (the worker keeps one mesh in W at a time)."""
    key = id(mesh)
    hit = _planet_cache.get(key)
    if hit is not None and hit[0] is mesh:
        return hit[1]
    if r_xyz is None:
        raise ValueError("first call for this mesh needs r_xyz (use bind_mesh(mesh, r_xyz, neighborDist))")
    _planet_cache.clear()
    pl = Planet(mesh, r_xyz, neighborDist)
    _planet_cache[key] = (mesh, pl)
    return pl


def bind_mesh(mesh, r_xyz, neighborDist=None) -> Planet:
    return _planet_for(mesh, r_xyz, neighborDist)


def warp_terrain(mesh, r_elevation, r_xyz, seed, strength, r_hotspot=None):
    if strength <= 0:
        return
    _planet_for(mesh, r_xyz).warp_terrain(r_elevation, seed, strength, r_hotspot)


def smooth_elevation(mesh, r_elevation, r_isOcean, iterations, strength):
    _planet_for(mesh).smooth_elevation(r_elevation, r_isOcean, iterations, strength)


def erode_composite(mesh, r_elevation, r_xyz, r_isOcean, hIters, K, m, dt, tIters, talusSlope, kThermal,
                    gIters=0, glacialStrength=0.0, neighborDist=None):
    _planet_for(mesh, r_xyz, neighborDist).erode_composite(r_elevation, r_isOcean, hIters, K, m, dt, tIters, talusSlope, kThermal,
                                                           gIters, glacialStrength)


def sharpen_ridges(mesh, r_elevation, r_isOcean, iterations, strength):
    _planet_for(mesh).sharpen_ridges(r_elevation, r_isOcean, iterations, strength)


def apply_soil_creep(mesh, r_elevation, r_isOcean, iterations, strength):
    _planet_for(mesh).apply_soil_creep(r_elevation, r_isOcean, iterations, strength)


def noise_eval(seed, kind, xyz, octaves=5, p0=2.0 / 3.0, p1=0.5, p2=1.0, ctx: Context | None = None) -> np.ndarray:
    """SimplexNoise(seed).noise3D / fbm / ridgedFbm at many points on the device (js/simplex-noise.js:17-53)."""
    ctx = ctx or default_context()
    xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
    out = np.empty(xyz.shape[0], np.float64)
    capi.check(capi.lib().wo_noise_eval(ctx.handle, float(seed), int(kind), int(octaves), float(p0), float(p1), float(p2), xyz.shape[0],
                                        capi.ptr(xyz), capi.ptr(out)), "wo_noise_eval")
    return out


def run_post_processing(planet: Planet, r_elevation, params: dict, seed, r_hotspot=None):
    """Counterpart of runPostProcessing (js/planet-worker.js:40-102): slider -> argument mapping, ocean mask
    taken after the warp, fixed soil creep.  Mutates r_elevation in place; returns (r_isOcean, erosionDelta)."""
    e = _f32(r_elevation, planet.numRegions)
    g = lambda k: float(params.get(k, 0.0))  # noqa: E731
    warp_s, smoothing, glac, hyd, therm, ridge = g("terrainWarp"), g("smoothing"), g("glacialErosion"), g("hydraulicErosion"), g("thermalErosion"), g("ridgeSharpening")
    planet.upload(e)
    if r_hotspot is not None:
        planet.upload_hotspot(r_hotspot)
    if warp_s > 0:
        planet.warp_terrain_resident(seed, warp_s, r_hotspot is not None)
    planet.ocean_from_elevation()
    pre = planet.download()
    if smoothing > 0:
        planet.smooth_elevation_resident(_js_round(1 + smoothing * 4), 0.2 + smoothing * 0.5)
    if glac > 0 or hyd > 0 or therm > 0:
        planet.erode_composite_resident(_js_round(hyd * 20), 0.0006 * hyd, 0.5, 1.0, _js_round(therm * 10), 1.2 - therm * 0.4,
                                        therm * 0.15, _js_round(glac * 10), glac)
    if ridge > 0:
        planet.sharpen_ridges_resident(_js_round(1 + ridge * 3), ridge * 0.08)
    planet.apply_soil_creep_resident(3, 0.1125)
    out = planet.download()
    e[:] = out
    return planet.download_ocean(), (out.astype(np.float64) - pre.astype(np.float64)).astype(np.float32)


def _js_round(x: float) -> int:
    import math
    return int(math.floor(x + 0.5))
