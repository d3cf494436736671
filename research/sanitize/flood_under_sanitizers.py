"""The host flood (csrc/flood_host.cc) under ASan+UBSan and under TSan: builds the test emulator with the sanitizer into /tmp, runs the flood
routes — landmass pipeline, ring / heap walks, chain form, replay of the single heap with stop levels and forced prefixes, really undecided
landmasses, the shares' exchange with one flooding share — against the oracle.  Usage: python research/sanitize/flood_under_sanitizers.py asan|tsan
(the script re-executes itself under LD_PRELOAD of the sanitizer runtime).  Not part of the test suite (minutes; needs the gcc runtimes)."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[2]
kind = sys.argv[1] if len(sys.argv) > 1 else "asan"
flags = {"asan": ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"], "tsan": ["-fsanitize=thread"]}[kind]
lib = Path(f"/tmp/libemu_{kind}.so")
if os.environ.get("WO_SAN_CHILD") != kind:
    src = [str(REPO / "tests/emu/emu_erode.cc")] + [str(REPO / "planet_heightmap_generation_amd/csrc" / f) for f in ("flood_host.cc", "noise_host.cc", "elevation_host.cc")]
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-pthread", *flags, *src, "-o", str(lib)], check=True)
    rt = subprocess.run(["g++", f"-print-file-name=lib{kind}.so"], capture_output=True, text=True, check=True).stdout.strip()
    env = dict(os.environ, WO_SAN_CHILD=kind, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0", TSAN_OPTIONS="report_signal_unsafe=0")
    sys.exit(subprocess.run([sys.executable, __file__, kind], env=env).returncode)

sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import numpy as np                                                                  # noqa: E402
from oracle import pyoracle as O                                                    # noqa: E402
from planet_heightmap_generation_amd import decomposed as D, sphere_mesh as S       # noqa: E402

L = C.CDLL(str(lib)); p = C.c_void_p
L.emu_flood_host.argtypes = [C.c_int32, p, p, p, p, p, C.c_double, C.c_int32, C.c_int32, p]
L.emu_flood_shares.argtypes = [C.c_int32, p, p, p, p, p, p, C.c_int32, C.c_double, C.c_int32, p]
P = lambda a: a.ctypes.data_as(p)                                                   # noqa: E731
cells = 60000 if kind == "asan" else 700000
mesh, xyz, nd = S.build_sphere(cells, 0.75, 2)
om = O.Mesh(mesh.adjOffset, mesh.adjList)
base = O.warp_terrain(om, O.synthetic_terrain(xyz, 2), xyz, 2, 0.75)
oc = (base <= 0).astype(np.uint8)
eroded = O.erode_composite(om, base, xyz, oc, 8, 3e-4, 0.5, 1.0, 8, 1.16, 0.015, 1, 0.5, nd)
N = mesh.numRegions
r = np.arange(N, dtype=np.float64)
h = np.mod(r * 2654435761.0, 4294967296.0).astype(np.uint64).astype(np.uint32)
x = ((h >> np.uint32(16)) ^ h).astype(np.int32).astype(np.float64)
h = np.mod(x * 73244475.0, 4294967296.0).astype(np.int64).astype(np.uint32)
h = (h >> np.uint32(16)) ^ h
noise = (h.astype(np.float64) / 4294967295.0 * 0.01).astype(np.float32)


def tied_band(lo, cut):
    e0 = np.where((base > lo) & (base < cut), np.float32(cut) - noise, base).astype(np.float32)
    return np.where((base > 0) & (e0 <= 0), np.float32(1e-3), e0).astype(np.float32)


fields = (("fresh", base), ("eroded", eroded), ("tied low", tied_band(0.0, 0.06)), ("tied band", tied_band(0.05, 0.08)))
routes = ({}, {"WO_FLOOD_RING_MIN": "1", "WO_FLOOD_CHAINS_MIN": "2"}, {"WO_FLOOD_FORCE_DIRTY": "0", "WO_FLOOD_REPLAY_STOP": "0.1"},
          {"WO_FLOOD_FORCE_DIRTY": "0", "WO_FLOOD_FORCE_PREFIX": "400"}, {"WO_FLOOD_FORCE_DIRTY": "1", "WO_FLOOD_FORCE_PREFIX": "1000", "WO_FLOOD_REPLAY_STOP": "0.05"},
          {"WO_FLOOD_PREFIX": "0"})
for envs in routes:
    os.environ.update(envs)
    for name, e0 in fields:
        ref = O.priority_flood_carve(om, e0, oc, 0.85)
        for mode in (1, 101, 0):
            e = e0.copy(); st = np.zeros(11)
            L.emu_flood_host(N, P(mesh.adjOffset), P(mesh.adjList), P(xyz), P(e), P(oc), 0.85, mode + 10 if mode < 100 else 111, 1, P(st))
            assert np.array_equal(e, ref), (envs, name, mode)
    for k in envs:
        del os.environ[k]
    print("route", envs or "default", "ok", flush=True)
for shares in (2, 5):
    plan = D.plan_landmasses(mesh, oc, shares)
    owner = np.ascontiguousarray(plan.owner, np.int32)
    for name, e0 in fields[2:]:
        ref = O.priority_flood_carve(om, e0, oc, 0.5)
        e = e0.copy(); st = np.zeros(4)
        L.emu_flood_shares(N, P(mesh.adjOffset), P(mesh.adjList), P(xyz), P(e), P(oc), P(owner), shares, 0.5, 1, P(st))
        assert np.array_equal(e, ref), (shares, name, st)
        print("shares", shares, name, "gathers / whole-planet floods / replays / received", st, flush=True)
print(kind, "run finished")
