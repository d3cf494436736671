"""Research driver for solve_schedule.cc: launches the patch solve needs under different patch orders (10 M-cell bench planet
dumped by /tmp/dump_planet.py: off/adj/xyz/oc/e0[/e1].npy)."""
import ctypes as C, sys, time, numpy as np
d = sys.argv[1]; which = sys.argv[2] if len(sys.argv) > 2 else "e0"
off = np.load(d + "/off.npy"); adj = np.load(d + "/adj.npy"); xyz = np.load(d + "/xyz.npy"); oc = np.load(d + "/oc.npy")
e = np.load(d + "/%s.npy" % which).copy()
N = off.size - 1
p = C.c_void_p; a = lambda x: x.ctypes.data_as(p)
emu = C.CDLL("/root/repo/tests/emu/_build/libemu.so")
emu.emu_flood_host.argtypes = [C.c_int32, p, p, p, p, p, C.c_double, C.c_int32, C.c_int32, p]
if "--noflood" not in sys.argv:
    emu.emu_flood_host(N, a(off), a(adj), a(xyz), a(e), a(oc), 0.5, 1, 1, None)
ss = C.CDLL("/tmp/libss.so")
target = np.empty(N, np.int32); rank = np.empty(N, np.int32)
ss.ss_receivers(C.c_int32(N), a(off), a(adj), a(e), a(oc), a(target), a(rank))
land = np.flatnonzero(oc == 0).astype(np.int32); L = land.size
def morton_order():
    q = np.clip(((xyz.reshape(-1, 3)[land] + 1.0) * 511.5).astype(np.int64), 0, 1023)
    def spread(v):
        v = v & 1023; v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249; return v
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return land[np.argsort(key, kind="stable")]
def evaluate(name, order, pc=1024):
    slot = np.full(N, -1, np.int32); slot[order] = np.arange(L, dtype=np.int32)
    out = np.zeros(4); hist = np.zeros(512, np.int64)
    ss.ss_launches(C.c_int32(N), a(off), a(adj), a(target), a(rank), a(oc), a(slot), C.c_int32(pc), a(out), a(hist), C.c_int32(512))
    cum = np.cumsum(hist) / L
    print(f"{name:28s} patch {pc}: launches {int(out[0])}, DAG depth {int(out[1])}, cross-patch edges {out[2]/out[3]:.3f}, done after 5/10/20/40 launches: "
          + " ".join(f"{cum[k-1]:.3f}" for k in (5, 10, 20, 40)), flush=True)
evaluate("morton", morton_order())
evaluate("ascending id", land)
for v, nm in ((0, "river siblings-first"), (4, "early-edge forest, siblings-first"), (2, "river preorder")):
    order = np.empty(L, np.int32); info = np.zeros(4)
    ss.ss_river_order(C.c_int32(N), a(target), a(oc), C.c_int32(v), a(order), a(info), a(rank))
    assert np.array_equal(np.sort(order), land)
    print("  roots %d unreached %d largest tree %d" % tuple(info[:3]))
    evaluate(nm, order)
for rounds in (12,):
    order = np.empty(L, np.int32); info = np.zeros(4)
    ss.ss_river_keys(C.c_int32(N), a(off), a(adj), a(target), a(rank), a(oc), C.c_int32(rounds), a(order), a(info))
    assert np.array_equal(np.sort(order), land)
    print("  device-style keys: roots %d, still jumping after %d rounds: %d, sum of root sizes %d, largest key %d" % (info[0], rounds, info[1], info[2], info[3]))
    evaluate("device-style keys + sort", order)
    device_order = order
if "--stale" in sys.argv:        # order from this field's forest, evaluated on another field's dependencies
    other = np.load(d + "/%s.npy" % sys.argv[sys.argv.index("--stale") + 1])
    order = np.empty(L, np.int32); info = np.zeros(4)
    ss.ss_river_order(C.c_int32(N), a(target), a(oc), C.c_int32(4), a(order), a(info), a(rank))
    ss.ss_receivers(C.c_int32(N), a(off), a(adj), a(other), a(oc), a(target), a(rank))
    evaluate("stale river order", order)
    evaluate("morton on the other field", morton_order())
