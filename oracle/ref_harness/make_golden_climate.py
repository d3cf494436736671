#!/usr/bin/env python3
"""Golden vectors for smoothField (js/climate-util.js:5-25) and the climate sweeps diffuseOceanWarmth (js/temperature.js:19-66),
computeWindConvergence (js/precipitation.js:18-52), advectMoisture (js/precipitation.js:59-195), produced by running the
REFERENCE JavaScript under Node (run_smooth_field.mjs, run_climate_sweeps.mjs) on the mesh and start elevation of
tests/golden/post_N10000_s1.npz; the sweeps' other inputs are rebuilt deterministically by tests/climate_common.py.
Only outputs are stored.

Usage: python oracle/ref_harness/make_golden_climate.py [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import json
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from oracle.ref_harness.make_golden import prepare_reference  # noqa: E402

GOLD = REPO / "tests" / "golden"
HARNESS = Path(__file__).resolve().parent / "run_smooth_field.mjs"
HARNESS_SWEEPS = Path(__file__).resolve().parent / "run_climate_sweeps.mjs"
sys.path.insert(0, str(REPO / "tests"))
from climate_common import SWEEP_CASES, sweep_inputs  # noqa: E402
PASSES = (0, 1, 4, 7)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    g = np.load(GOLD / "post_N10000_s1.npz")
    with tempfile.TemporaryDirectory(prefix="wo_golden_cl_") as td:
        work = Path(td)
        ref_js = prepare_reference(Path(args.ref), work)
        for k in ("adjOffset", "adjList", "elevation0"):
            np.ascontiguousarray(g[k]).tofile(work / f"{k}.bin")
        job = dict(numRegions=int(g["adjOffset"].size - 1), adjOffset=str(work / "adjOffset.bin"), adjList=str(work / "adjList.bin"),
                   field=str(work / "elevation0.bin"), cases=[dict(passes=p, out=str(work / f"out_{p}.bin")) for p in PASSES])
        (work / "job.json").write_text(json.dumps(job))
        subprocess.run(["node", str(HARNESS), str(ref_js), str(work / "job.json")], check=True)
        data = {f"ref_smoothField_{p}": np.fromfile(work / f"out_{p}.bin", np.float32) for p in PASSES}
        # ---- sweeps: the module-private functions are exported from the scratch copies only
        for mod, names in (("temperature.js", "diffuseOceanWarmth"), ("precipitation.js", "computeWindConvergence, advectMoisture")):
            f = ref_js / mod
            f.write_text(f.read_text() + f"\nexport {{ {names} }};\n")
        inp = sweep_inputs(g["adjOffset"], g["adjList"], g["xyz"], g["elevation0"])
        files = {}
        for k, v in dict(adjOffset=g["adjOffset"], adjList=g["adjList"], xyz=g["xyz"], **inp).items():
            np.ascontiguousarray(v).tofile(work / f"sw_{k}.bin"); files[k] = str(work / f"sw_{k}.bin")
        cases = []
        for p_ in SWEEP_CASES["diffuse_passes"]:
            cases.append(dict(fn="diffuseOceanWarmth", passes=p_, key=f"ref_diffuse_{p_}"))
        cases.append(dict(fn="diffuseOceanWarmth", passes=SWEEP_CASES["diffuse_no_cont_passes"], noCont=True, noWarmth=True, key="ref_diffuse_nulls"))
        cases.append(dict(fn="computeWindConvergence", key="ref_convergence"))
        for h_ in SWEEP_CASES["advect_hops"]:
            cases.append(dict(fn="advectMoisture", maxHops=h_, key=f"ref_advect_{h_}"))
        cases.append(dict(fn="advectMoisture", maxHops=SWEEP_CASES["advect_hops"][0], noWarmth=True, key="ref_advect_nowarmth"))
        for c in cases:
            c["out"] = str(work / (c["key"] + ".bin"))
        (work / "job_sw.json").write_text(json.dumps(dict(numRegions=int(g["adjOffset"].size - 1), **{"in": files}, cases=cases)))
        subprocess.run(["node", str(HARNESS_SWEEPS), str(ref_js), str(work / "job_sw.json")], check=True)
        sweeps = {c["key"]: np.fromfile(c["out"], np.float32) for c in cases}
    np.savez_compressed(GOLD / "climate_sweeps_N10000_s1.npz", **sweeps)
    print(f"wrote tests/golden/climate_sweeps_N10000_s1.npz ({(GOLD / 'climate_sweeps_N10000_s1.npz').stat().st_size / 1024:.0f} KiB)")
    np.savez_compressed(GOLD / "climate_N10000_s1.npz", **data)
    print(f"wrote tests/golden/climate_N10000_s1.npz ({(GOLD / 'climate_N10000_s1.npz').stat().st_size / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
