#!/usr/bin/env python3
"""Golden vectors for projectCoarsePlates + smoothAndReconnectPlates, produced by running the REFERENCE JavaScript
under Node (see run_plates.mjs) on the build's triangulations.  Only what the tests cannot regenerate is stored:
the coarse plate table (reference host logic on the 20 000-cell mesh), the seeds and the two hi-res results; the
meshes themselves come from the build's mesh producer, whose CSR/points are pinned to the reference's elsewhere
(tests/test_mesh_builder.py) and are re-checked here by checksum.

Usage: python oracle/ref_harness/make_golden_plates.py [--ref /root/reference]
"""
from __future__ import annotations

import argparse
import json
import subprocess
import sys
import tempfile
import zlib
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from oracle.ref_harness.make_golden import prepare_reference  # noqa: E402
from oracle.ref_harness.make_golden_elevation import planar_triangulation  # noqa: E402

GOLD = REPO / "tests" / "golden"
HARNESS = Path(__file__).resolve().parent / "run_plates.mjs"


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def run_case(ref_js: Path, work: Path, name: str, N: int, seed: int, P: int, passes: int = 3):
    d = work / name
    d.mkdir()
    trs = []
    for n, jit, sd in ((N, 0.75, seed), (20000, 0.75, seed + 137)):
        t, h = planar_triangulation(n, jit, sd)
        t.tofile(d / f"tri_{n}.bin"); h.tofile(d / f"he_{n}.bin")
        trs.append({"n": n, "triangles": str(d / f"tri_{n}.bin"), "halfedges": str(d / f"he_{n}.bin")})
    job = dict(triangulations=trs, N=N, P=P, jitter=0.75, numContinents=4, seed=seed, passes=passes, out=str(d) + "/o_")
    (d / "job.json").write_text(json.dumps(job))
    subprocess.run(["node", "--max-old-space-size=6000", str(HARNESS), str(ref_js), str(d / "job.json")], check=True)
    meta = json.loads((d / "o_meta.json").read_text())
    rd = lambda k, dt: np.fromfile(d / f"o_{k}.bin", dtype=dt)
    for k, dt in (("coarse_xyz", np.float32), ("coarse_adjOffset", np.int32), ("coarse_adjList", np.int32), ("xyz", np.float32),
                  ("adjOffset", np.int32), ("adjList", np.int32)):
        meta["crc_" + k] = crc(rd(k, dt))
    data = {"meta_json": np.frombuffer(json.dumps(meta).encode(), np.uint8),
            "coarse_r_plate": rd("coarse_r_plate", np.int32), "plateSeeds": rd("plateSeeds", np.int32),
            "r_plate_projected": rd("r_plate_projected", np.int32), "r_plate_smoothed": rd("r_plate_smoothed", np.int32)}
    np.savez_compressed(GOLD / f"{name}.npz", **data)
    diff = int((data["r_plate_projected"] != data["r_plate_smoothed"]).sum())
    print(f"wrote tests/golden/{name}.npz ({(GOLD / (name + '.npz')).stat().st_size / 1024:.0f} KiB); reference: project {meta['msProject']:.0f} ms, "
          f"smooth {meta['msSmooth']:.0f} ms; smoothing changed {diff} cells")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory(prefix="wo_golden_pl_") as td:
        work = Path(td)
        ref_js = prepare_reference(Path(args.ref), work)
        run_case(ref_js, work, "plates_N10000_s1_P80", 10000, 1, 80)
        run_case(ref_js, work, "plates_N5000_s3_P24", 5000, 3, 24)
        run_case(ref_js, work, "plates_N200000_s5_P12", 200000, 5, 12)


if __name__ == "__main__":
    main()
