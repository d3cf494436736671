"""Research driver (solve_schedule.cc: ss_launches): how many hand-off rounds K would the implicit solve need per iteration if
the mesh were cut into B contiguous index bands (north_star / SURVEY 8(e): latitude bands of the Fibonacci spiral) and a
dependency chain could only cross a band boundary at an exchange?  K + 1 = the launches-needed figure with patch = band.
usage: python research/band_handoffs.py <planet dump dir> <field> [--noflood]"""
import ctypes as C, json, sys, numpy as np
d = sys.argv[1]; which = sys.argv[2] if len(sys.argv) > 2 else "e0"
off = np.load(d + "/off.npy"); adj = np.load(d + "/adj.npy"); xyz = np.load(d + "/xyz.npy"); oc = np.load(d + "/oc.npy")
e = np.load(d + "/%s.npy" % which).copy()
N = off.size - 1
p = C.c_void_p; a = lambda x: x.ctypes.data_as(p)
if "--noflood" not in sys.argv:
    emu = C.CDLL("/root/repo/tests/emu/_build/libemu.so")
    emu.emu_flood_host.argtypes = [C.c_int32, p, p, p, p, p, C.c_double, C.c_int32, C.c_int32, p]
    emu.emu_flood_host(N, a(off), a(adj), a(xyz), a(e), a(oc), 0.5, 1, 1, None)
ss = C.CDLL("/tmp/libss.so")
target = np.empty(N, np.int32); rank = np.empty(N, np.int32)
ss.ss_receivers(C.c_int32(N), a(off), a(adj), a(e), a(oc), a(target), a(rank))
L = int((oc == 0).sum())
slot = np.arange(N, dtype=np.int32)
out_all = {}
for B in (2, 4, 8):
    per = (N + B - 1) // B
    out = np.zeros(4); hist = np.zeros(512, np.int64)
    ss.ss_launches(C.c_int32(N), a(off), a(adj), a(target), a(rank), a(oc), a(slot), C.c_int32(per), a(out), a(hist), C.c_int32(512))
    K = int(out[0]) - 1
    h = hist[:int(out[0])].tolist()
    # receiver edges that cross a band boundary (what a halo exchange of the receivers pass carries)
    land = np.flatnonzero(oc == 0)
    t = target[land]; ok = t >= 0
    cross_edges = int(((land[ok] // per) != (t[ok] // per)).sum())
    out_all[str(B)] = dict(bands=B, hand_off_rounds_K=K, dag_depth=int(out[1]), tasks_by_round=h, dependency_edges=int(out[3]), edges_crossing_a_band=int(out[2]),
                           receiver_edges_crossing_a_band=cross_edges, land_cells=L)
    print(f"{which}: {B} bands: K = {K} hand-off rounds (tasks finishing in round 1..: {h[:12]}{'...' if len(h) > 12 else ''}), DAG depth {int(out[1])}, "
          f"crossing dependency edges {int(out[2])} of {int(out[3])}, crossing receiver edges {cross_edges}", flush=True)
print(json.dumps({"field": which, "cells": int(N), "solve": out_all}))
