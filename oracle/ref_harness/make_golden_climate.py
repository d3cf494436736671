#!/usr/bin/env python3
"""Golden vectors for smoothField (js/climate-util.js:5-25), produced by running the REFERENCE JavaScript under Node
(run_smooth_field.mjs) on the mesh and start elevation of tests/golden/post_N10000_s1.npz.  Only outputs are stored.

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
    np.savez_compressed(GOLD / "climate_N10000_s1.npz", **data)
    print(f"wrote tests/golden/climate_N10000_s1.npz ({(GOLD / 'climate_N10000_s1.npz').stat().st_size / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
