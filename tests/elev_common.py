"""Shared helpers for the assignElevation tests: rebuild the reference's argument objects from a golden file."""
import json

import numpy as np


def load_case(g):
    meta = json.loads(bytes(g["meta_json"]).decode())
    ids = g["plateSeeds"].tolist()
    vec = {pid: {"pole": g["plateVec"][4 * i:4 * i + 3].tolist(), "omega": float(g["plateVec"][4 * i + 3])} for i, pid in enumerate(ids)}
    dens = {pid: float(g["plateDensity"][i]) for i, pid in enumerate(ids)}
    is_ocean = [pid for i, pid in enumerate(ids) if g["plateIsOcean"][i]]
    sup = None
    if meta["hasSuper"]:
        ns = meta["numSuperPlates"]
        sup = {"r_superPlate": g["r_superPlate"],
               "superPlateVec": {s: {"pole": g["superPlateVec"][4 * s:4 * s + 3].tolist(), "omega": float(g["superPlateVec"][4 * s + 3])} for s in range(ns)},
               "superPlateIsOcean": [s for s in range(ns) if g["superPlateIsOcean"][s]],
               "superPlateDensity": {s: float(g["superPlateDensity"][s]) for s in range(ns)}}
    return meta, ids, vec, dens, is_ocean, sup


def dense_table(ids, vec4, dens, isoc):
    n = int(np.max(ids)) + 1
    has = np.zeros(n, np.uint8); pole = np.zeros(3 * n); om = np.zeros(n); oc = np.zeros(n, np.uint8); de = np.full(n, np.nan)
    for i, pid in enumerate(ids):
        has[pid] = 1; pole[3 * pid:3 * pid + 3] = vec4[4 * i:4 * i + 3]; om[pid] = vec4[4 * i + 3]; oc[pid] = isoc[i]; de[pid] = dens[i]
    return n, has, pole, om, oc, de
