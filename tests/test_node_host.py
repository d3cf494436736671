"""The JavaScript host (planet_heightmap_generation_amd/js, N-API shim over the C ABI) driven under Node the
way the reference's worker drives terrain-post.js.  CPU part: the addon loads, exposes the call surface, the
host-side producers and the scalar SimplexNoise agree with the reference goldens, device calls throw a JS
Error without a GPU.  GPU part (-m gpu): the five exports and runPostProcessing reproduce the goldens."""
import json
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import REPO, golden_cases, load_golden

NODE = shutil.which("node")
ADDON = REPO / "planet_heightmap_generation_amd" / "worogen.node"
DRIVER = REPO / "tests" / "node" / "run_cases.mjs"
pytestmark = pytest.mark.skipif(NODE is None or not ADDON.exists(), reason="node or worogen.node not available")


def run_node(tmp: Path, jobs):
    (tmp / "jobs.json").write_text(json.dumps({"jobs": jobs}))
    r = subprocess.run([NODE, "--no-warnings", str(DRIVER), str(tmp)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads((tmp / "result.json").read_text())


def test_addon_surface_and_host_side(tmp_path):
    g = load_golden("noise_seed1")
    pts = g["points"][:256]
    pts.tofile(tmp_path / "pts.bin")
    res = run_node(tmp_path, [
        {"op": "exports"},
        {"op": "noise_scalar", "seed": 1, "points": "pts.bin", "out": "noise.bin"},
        {"op": "build_sphere", "N": 2000, "jitter": 0.75, "seed": 1, "out": "mesh"},
    ])
    for name in ("warpTerrain", "smoothElevation", "erodeComposite", "sharpenRidges", "applySoilCreep", "planetCreate", "noiseTables"):
        assert name in res["exports"]
    o = np.fromfile(tmp_path / "noise.bin", np.float64).reshape(-1, 4)
    assert np.array_equal(o[:, 0], g["ref_noise3D"][:256]) and np.array_equal(o[:, 1], g["ref_fbm5"][:256])
    assert np.array_equal(o[:, 2], g["ref_ridged6"][:256]) and np.array_equal(o[:, 3], g["ref_ridged3h"][:256])
    assert np.array_equal(np.fromfile(tmp_path / "noise.bin.perm", np.uint8), g["ref_perm"])
    m = load_golden("mesh_N2000_s1")
    assert np.array_equal(np.fromfile(tmp_path / "mesh.xyz", np.float32), m["xyz"])
    assert np.array_equal(np.fromfile(tmp_path / "mesh.tri", np.int32), m["triangles"])
    assert np.array_equal(np.fromfile(tmp_path / "mesh.off", np.int32), m["ref_adjOffset"])
    assert np.array_equal(np.fromfile(tmp_path / "mesh.adj", np.int32), m["ref_adjList"])
    assert np.array_equal(np.fromfile(tmp_path / "mesh.nd", np.float32), m["ref_neighborDist"])
    if res["deviceCount"] == 0:
        res2 = run_node(tmp_path, [{"op": "device_must_throw"}])
        assert res2["threw"] and "no usable HIP device" in res2["threw"]


@pytest.mark.gpu
def test_js_drop_in_matches_goldens(tmp_path):
    g = load_golden("post_N10000_s1")
    m = load_golden("mesh_N10000_s1")
    for k, arr in (("tri", m["triangles"]), ("he", m["halfedges"]), ("xyz", g["xyz"]), ("nd", g["neighborDist"]), ("e0", g["elevation0"]),
                   ("oc", g["isOcean"]), ("hot", g["hotspot"])):
        np.ascontiguousarray(arr).tofile(tmp_path / f"{k}.bin")
    jobs = [{"op": "load_mesh", "tri": "tri.bin", "he": "he.bin", "xyz": "xyz.bin", "nd": "nd.bin", "numRegions": int(g["numRegions"])}]
    cases = {k: v for k, v in golden_cases(g).items() if v["fn"] != "priorityFloodCarve"}
    for name, c in cases.items():
        jobs.append({"op": "post", "fn": c["fn"], "args": c["args"], "elevation": "e0.bin", "isOcean": "oc.bin",
                     "hotspot": "hot.bin" if "hot" in name else None, "out": f"out_{name}.bin"})
    params = dict(terrainWarp=0.75, smoothing=0.10, glacialErosion=0.5, hydraulicErosion=0.5, thermalErosion=0.1, ridgeSharpening=0.5)
    jobs.append({"op": "pipeline", "elevation": "e0.bin", "hotspot": "hot.bin", "params": params, "seed": 1, "out": "pipe.bin"})
    jobs.append({"op": "error_paths", "elevation": "e0.bin", "isOcean": "oc.bin"})
    jobs.append({"op": "smooth_field", "field": "e0.bin", "passes": 4, "out": "sf4.bin"})
    from climate_common import SWEEP_CASES, sweep_inputs
    I = sweep_inputs(g["adjOffset"], g["adjList"], g["xyz"], g["elevation0"])
    for k, v in I.items():
        v.tofile(tmp_path / f"cs_{k}.bin")
    jobs.append({"op": "climate_sweeps", "in": {k: f"cs_{k}.bin" for k in I}, "passes": SWEEP_CASES["diffuse_passes"][-1], "passesNulls": SWEEP_CASES["diffuse_no_cont_passes"],
                 "maxHops": SWEEP_CASES["advect_hops"][-1], "out": {"diffuse": "cs_d.bin", "diffuseNulls": "cs_dn.bin", "convergence": "cs_c.bin", "advect": "cs_a.bin"}})
    pts = load_golden("noise_seed78")["points"]
    pts.tofile(tmp_path / "pts.bin")
    jobs.append({"op": "noise_batch", "seed": 78, "kind": "ridgedFbm", "points": "pts.bin", "octaves": 3, "p0": 0.5, "p1": 0.5, "p2": 1.0, "out": "nb.bin"})
    res = run_node(tmp_path, jobs)
    for name, c in cases.items():
        got = np.fromfile(tmp_path / f"out_{name}.bin", np.float32)
        ref = g["ref_" + name]
        d = got.astype(np.float64) - ref.astype(np.float64)
        assert float(np.sqrt((d * d).mean())) < 1e-5, name
        if not (c["fn"] == "erodeComposite" and (c["args"]["gIters"] > 0 or c["args"]["m"] != 0.5)):
            assert np.array_equal(got, ref), name
    assert np.array_equal(np.fromfile(tmp_path / "nb.bin", np.float64), load_golden("noise_seed78")["ref_ridged3h"])
    assert res["errors"] == ["TypeError", "TypeError", "RangeError", "RangeError", "RangeError"]
    assert np.array_equal(np.fromfile(tmp_path / "sf4.bin", np.float32), load_golden("climate_N10000_s1")["ref_smoothField_4"])
    cs = load_golden("climate_sweeps_N10000_s1")
    for f, k in (("cs_d.bin", f"ref_diffuse_{SWEEP_CASES['diffuse_passes'][-1]}"), ("cs_dn.bin", "ref_diffuse_nulls"), ("cs_c.bin", "ref_convergence"),
                 ("cs_a.bin", f"ref_advect_{SWEEP_CASES['advect_hops'][-1]}")):
        assert np.array_equal(np.fromfile(tmp_path / f, np.float32), cs[k]), k
    assert res["climateErrors"] == ["RangeError"]
    assert res["postTiming"][0].startswith("Terrain warp") and res["postTiming"][-1] == "Soil creep (3 iters)"
    # pipeline == the Python mirror's pipeline (same C ABI underneath)
    from planet_heightmap_generation_amd import terrain_post as TP

    class _M:
        adjOffset, adjList, numRegions = g["adjOffset"], g["adjList"], int(g["numRegions"])
    pl = TP.Planet(_M, g["xyz"], g["neighborDist"])
    e = g["elevation0"].copy()
    _, delta = TP.run_post_processing(pl, e, params, 1.0, g["hotspot"])
    assert np.array_equal(np.fromfile(tmp_path / "pipe.bin", np.float32), e)
    assert np.array_equal(np.fromfile(tmp_path / "pipe.bin.delta", np.float32), delta)
    pl.close()


@pytest.mark.gpu
def test_js_assign_elevation(tmp_path):
    import json as _json
    g = load_golden("elev_config1_N10000_s1")
    meta = _json.loads(bytes(g["meta_json"]).decode())
    for k in ("triangles", "halfedges", "xyz", "neighborDist", "r_plate", "plateSeeds", "plateVec", "plateDensity", "plateIsOcean", "r_superPlate",
              "superPlateVec", "superPlateDensity", "superPlateIsOcean"):
        np.ascontiguousarray(g[k]).tofile(tmp_path / f"{k}.bin")
    jobs = [{"op": "load_mesh", "tri": "triangles.bin", "he": "halfedges.bin", "xyz": "xyz.bin", "nd": "neighborDist.bin", "numRegions": meta["numRegions"]},
            {"op": "assign_elevation", "r_plate": "r_plate.bin", "plateSeeds": "plateSeeds.bin", "plateVec": "plateVec.bin", "plateDensity": "plateDensity.bin",
             "plateIsOcean": "plateIsOcean.bin", "r_superPlate": "r_superPlate.bin", "superPlateVec": "superPlateVec.bin",
             "superPlateDensity": "superPlateDensity.bin", "superPlateIsOcean": "superPlateIsOcean.bin", "seed": meta["seed"], "nMag": meta["nMag"],
             "spread": meta["spread"], "out": "ae"}]
    res = run_node(tmp_path, jobs)
    e = np.fromfile(tmp_path / "ae.elev", np.float32)
    d = e.astype(np.float64) - g["ref_elevation"].astype(np.float64)
    assert float(np.sqrt((d * d).mean())) < 1e-5
    assert np.array_equal(np.fromfile(tmp_path / "ae.mountain", np.int32), g["ref_mountain"])
    assert np.array_equal(np.fromfile(tmp_path / "ae.coastline", np.int32), g["ref_coastline"])
    assert np.array_equal(np.fromfile(tmp_path / "ae.ocean", np.int32), g["ref_ocean"])
    assert res["elevKeys"] == sorted(["r_elevation", "mountain_r", "coastline_r", "ocean_r", "r_stress", "debugLayers", "_timing"])
    assert "superPlates" in res["layerKeys"] and len(res["layerKeys"]) == 13


def _dump_plate_case(tmp_path, c):
    for k, arr in (("off", c["mesh"].adjOffset), ("adj", c["mesh"].adjList), ("xyz", c["xyz"]), ("coff", c["cmesh"].adjOffset), ("cadj", c["cmesh"].adjList),
                   ("cxyz", c["cxyz"]), ("cplate", c["coarse_r_plate"]), ("seeds", c["seeds"]), ("proj", c["projected"])):
        np.ascontiguousarray(arr).tofile(tmp_path / f"{k}.bin")


def test_js_smooth_and_reconnect_plates(tmp_path):
    """plates.js drop-in (host stage, runs without a GPU): mutates r_plate in place like the reference."""
    from plates_common import plate_case
    c = plate_case("plates_N10000_s1_P80")
    _dump_plate_case(tmp_path, c)
    run_node(tmp_path, [{"op": "smooth_plates", "numRegions": c["mesh"].numRegions, "off": "off.bin", "adj": "adj.bin", "r_plate": "proj.bin",
                         "seeds": "seeds.bin", "passes": c["meta"]["passes"], "out": "smoothed.bin"}])
    assert np.array_equal(np.fromfile(tmp_path / "smoothed.bin", np.int32), c["smoothed"])


@pytest.mark.gpu
def test_js_project_coarse_plates(tmp_path):
    from plates_common import plate_case
    c = plate_case("plates_N10000_s1_P80")
    _dump_plate_case(tmp_path, c)
    run_node(tmp_path, [{"op": "project_plates", "numRegions": c["mesh"].numRegions, "off": "off.bin", "adj": "adj.bin", "xyz": "xyz.bin",
                         "coarseRegions": c["cmesh"].numRegions, "coff": "coff.bin", "cadj": "cadj.bin", "cxyz": "cxyz.bin", "cplate": "cplate.bin",
                         "seed": c["meta"]["seed"], "P": c["meta"]["P"], "out": "rp.bin"}])
    assert np.array_equal(np.fromfile(tmp_path / "rp.bin", np.int32), c["projected"])


@pytest.mark.gpu
def test_worker_threads_counterpart(tmp_path):
    """planet_heightmap_generation_amd/js/planet-worker.js: the reference worker's retained state + `reapply` message
    (js/planet-worker.js:277-292, 341-440, 944-954) as a Node worker thread.  The reapply result equals the direct
    pipeline bit for bit, repeats exactly from the retained (device-resident) pre-erosion field, and errors / progress
    messages have the reference's shapes."""
    g = load_golden("post_N10000_s1")
    m = load_golden("mesh_N10000_s1")
    for k, arr in (("tri", m["triangles"]), ("off", g["adjOffset"]), ("adj", g["adjList"]), ("xyz", g["xyz"]), ("nd", g["neighborDist"]), ("e0", g["elevation0"]),
                   ("hot", g["hotspot"])):
        np.ascontiguousarray(arr).tofile(tmp_path / f"{k}.bin")
    params = dict(terrainWarp=0.75, smoothing=0.10, glacialErosion=0.5, hydraulicErosion=0.5, thermalErosion=0.1, ridgeSharpening=0.5)
    params2 = dict(terrainWarp=0.3, smoothing=0.0, glacialErosion=0.0, hydraulicErosion=0.8, thermalErosion=0.4, ridgeSharpening=0.0)
    job = dict(numRegions=int(g["numRegions"]), adjOffset="off.bin", adjList="adj.bin", triangles="tri.bin", xyz="xyz.bin", neighborDist="nd.bin",
               elevation="e0.bin", hotspot="hot.bin", seed=1, params=params, params2=params2)
    (tmp_path / "worker_job.json").write_text(json.dumps(job))
    r = subprocess.run([NODE, "--no-warnings", str(REPO / "tests" / "node" / "run_worker.mjs"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads((tmp_path / "worker_result.json").read_text())
    assert res["beforeRetain"] == {"type": "error", "message": "No retained state for reapply"}
    assert res["unknown"] == {"type": "error", "message": "Unknown command: frobnicate"}
    assert res["hostStage"]["type"] == "error" and "generate" in res["hostStage"]["message"]
    assert res["retained"] == {"type": "retained", "numRegions": int(g["numRegions"])}
    f = res["first"]
    assert f["type"] == "reapplyDone" and f["skipClimate"] is True and f["n"] == int(g["numRegions"]) and f["nt"] == m["triangles"].size // 3
    assert {"r_elevation", "t_elevation", "erosionDelta", "_reapplyTiming", "_postTiming", "skipClimate", "type"} <= set(f["keys"])
    assert f["timingKeys"] == sorted(["clone", "postProcessing", "wind", "ocean", "precipitation", "temperature", "triangleElevations", "workerTotal"])
    assert f["postTiming"][0].startswith("Terrain warp") and f["postTiming"][-1] == "Soil creep (3 iters)"
    assert [p["pct"] for p in res["progress"][:3]] == [0, 20, 70] and res["progress"][0]["label"].startswith("Reapplying terrain")
    # reapply == the direct pipeline on the same inputs, bit for bit; the triangle elevations and the delta belong to it
    from planet_heightmap_generation_amd import sphere_mesh as S, terrain_post as TP

    class _M:
        adjOffset, adjList, numRegions = g["adjOffset"], g["adjList"], int(g["numRegions"])
    pl = TP.Planet(_M, g["xyz"], g["neighborDist"])
    e = g["elevation0"].copy()
    oc, delta = TP.run_post_processing(pl, e, params, 1.0, g["hotspot"])
    e1 = np.fromfile(tmp_path / "w_elev1.bin", np.float32)
    assert np.array_equal(e1, e)
    assert np.array_equal(np.fromfile(tmp_path / "w_delta1.bin", np.float32), delta)
    tri = np.fromfile(tmp_path / "w_tri1.bin", np.float32)
    t = m["triangles"].reshape(-1, 3)
    assert np.allclose(tri, (e1[t[:, 0]].astype(np.float64) + e1[t[:, 1]] + e1[t[:, 2]]) / 3, atol=1e-6)
    e2 = g["elevation0"].copy()
    TP.run_post_processing(pl, e2, params2, 1.0, g["hotspot"])
    assert np.array_equal(np.fromfile(tmp_path / "w_elev2.bin", np.float32), e2) and not np.array_equal(e2, e)
    assert np.array_equal(np.fromfile(tmp_path / "w_elev3.bin", np.float32), e)          # back to the first sliders: same field again
    assert res["disposed"] == {"type": "disposed"} and res["afterDispose"]["message"] == "No retained state for reapply"
    pl.close()


@pytest.mark.gpu
def test_js_comm_bindings_single_rank(tmp_path):
    """include/worogen.h's wo_comm_* / wo_planet_exchange_* through the addon, the way a multi-GPU JS host would call them
     — with the one-rank communicator a one-GPU box allows: the id is 128 bytes, the communicator
    comes up, both exchange shapes run and leave the field alone, wrong counts throw."""
    g = load_golden("post_N10000_s1")
    m = load_golden("mesh_N10000_s1")
    for k, arr in (("tri", m["triangles"]), ("he", m["halfedges"]), ("xyz", g["xyz"]), ("nd", g["neighborDist"]), ("e0", g["elevation0"])):
        np.ascontiguousarray(arr).tofile(tmp_path / f"{k}.bin")
    res = run_node(tmp_path, [
        {"op": "load_mesh", "tri": "tri.bin", "he": "he.bin", "xyz": "xyz.bin", "nd": "nd.bin", "numRegions": int(g["numRegions"])},
        {"op": "comm_single_rank", "elevation": "e0.bin"},
    ])
    c = res["comm"]
    assert c["idBytes"] == 128 and c["idNonZero"] and c["unchanged"]
    assert c["errors"] == ["RangeError", "Error"], c


@pytest.mark.gpu
def test_js_planet_destroy(tmp_path):
    """planetDestroy frees a planet's device memory at once (the worker's dispose / re-retain use it): the handle stays behind
    dead — every entry point then throws instead of touching freed memory — and destroying twice is harmless."""
    g = load_golden("post_N10000_s1")
    m = load_golden("mesh_N10000_s1")
    for k, arr in (("tri", m["triangles"]), ("he", m["halfedges"]), ("xyz", g["xyz"]), ("nd", g["neighborDist"]), ("e0", g["elevation0"])):
        np.ascontiguousarray(arr).tofile(tmp_path / f"{k}.bin")
    res = run_node(tmp_path, [
        {"op": "load_mesh", "tri": "tri.bin", "he": "he.bin", "xyz": "xyz.bin", "nd": "nd.bin", "numRegions": int(g["numRegions"])},
        {"op": "planet_destroy", "elevation": "e0.bin"},
    ])
    d = res["destroy"]
    assert d["roundTrip"] and d["afterDestroy"] in ("TypeError", "Error"), d
