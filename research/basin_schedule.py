"""Research driver for basin_schedule.cc: drainage components of the implicit solve on a dumped planet (/tmp/dump_planet.py)."""
import ctypes as C, sys, numpy as np
d = sys.argv[1]; which = sys.argv[2] if len(sys.argv) > 2 else "e0"
off = np.load(d + "/off.npy"); adj = np.load(d + "/adj.npy"); xyz = np.load(d + "/xyz.npy"); oc = np.load(d + "/oc.npy")
e = np.load(d + "/%s.npy" % which).copy()
N = off.size - 1
p = C.c_void_p; a = lambda x: x.ctypes.data_as(p)
if "--noflood" not in sys.argv:
    emu = C.CDLL("/root/repo/tests/emu/_build/libemu.so")
    emu.emu_flood_host.argtypes = [C.c_int32, p, p, p, p, p, C.c_double, C.c_int32, C.c_int32, p]
    emu.emu_flood_host(N, a(off), a(adj), a(xyz), a(e), a(oc), 0.5, 1, 1, None)
ss = C.CDLL("/tmp/libss.so"); bs = C.CDLL("/tmp/libbs.so")
target = np.empty(N, np.int32); rank = np.empty(N, np.int32)
ss.ss_receivers(C.c_int32(N), a(off), a(adj), a(e), a(oc), a(target), a(rank))
L = int((oc == 0).sum())
for W in (1024, 2048):
    out = np.zeros(16); sizes = np.zeros(64, np.int32)
    bs.bs_analyse(C.c_int32(N), a(off), a(adj), a(target), a(rank), a(oc), C.c_int32(W), C.c_double(3.0), C.c_double(0.4), a(out), a(sizes), C.c_int32(64), None, None)
    print(f"W={W}: land {L}, components {int(out[0])}, largest {int(out[1])}, cells in components <= W: {out[2]/L:.3f}, DAG depth {int(out[3])}, "
          f"windows {int(out[5])} (ideal {L/W:.0f}), mean in-window depth {out[6]/out[5]:.1f}, largest component: {int(out[7])} windows, summed in-window depth {int(out[8])}, "
          f"est {out[4]:.0f} us; cross-component preds {int(out[9])}")
    print("  largest components:", sizes[:32].tolist())
