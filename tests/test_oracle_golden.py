"""The CPU oracle (oracle/*.c) against outputs of the reference JavaScript itself (tests/golden/*.npz,
made by oracle/ref_harness/make_golden.py).  Bit-exact: this is what pins the oracle."""
import numpy as np
import pytest

from conftest import POST_TAGS, golden_cases, load_golden

NOISE_KINDS = {  # golden key -> (kind, octaves, p0, p1, p2)
    "noise3D": (0, 5, 2 / 3, 0.5, 1.0), "fbm5": (1, 5, 2 / 3, 0.5, 1.0), "fbm4h": (1, 4, 0.5, 0.5, 1.0),
    "fbm2": (1, 2, 2 / 3, 0.5, 1.0), "ridged6": (2, 6, 2.0, 0.5, 1.0), "ridged3h": (2, 3, 0.5, 0.5, 1.0),
    "ridged4": (2, 4, 2.0, 0.5, 1.0)}


def test_cell_noise_known_answers(oracle):
    # SURVEY Appendix C, captured from the reference under Node 12 (JS double-rounded hash)
    rs = (0, 1, 12345, 3390000, 5000000, 9999999, 39999999)
    want = [0, 1667036860, 804843464, 1610598319, 2047435685, 1803284699, 3385718197]
    assert [oracle.lib().wo_or_cell_noise_hash(r) for r in rs] == want


@pytest.mark.parametrize("seed", [1, 10000, 1.5, 778])
def test_rng(oracle, seed):
    g = load_golden(f"rng_seed{seed}")
    assert np.array_equal(oracle.rng_values(seed, 64), g["ref_values"])
    assert np.array_equal(np.floor(oracle.rng_values(seed, 64) * 1000).astype(np.int32), g["ref_ints"])


def test_rng_known_answers(oracle):
    assert oracle.rng_values(1, 3).tolist() == [0.45861741198098044, 0.9828474013906414, 0.7162753066199602]
    assert oracle.rng_values(1.5, 1).tolist() == [0.4950100290542562]


@pytest.mark.parametrize("seed", [1, 10000, 78, 420])
def test_noise(oracle, seed):
    g = load_golden(f"noise_seed{seed}")
    p, m = oracle.noise_tables(seed)
    assert np.array_equal(p, g["ref_perm"]) and np.array_equal(m, g["ref_pm12"])
    for key, (kind, octv, p0, p1, p2) in NOISE_KINDS.items():
        got = oracle.noise_batch(seed, kind, g["points"], octv, p0, p1, p2)
        assert np.array_equal(got, g["ref_" + key]), key


def run_oracle_case(oracle, g, name, case):
    mesh = oracle.Mesh(g["adjOffset"], g["adjList"])
    e0, oc, xyz, nd = g["elevation0"], g["isOcean"], g["xyz"], g["neighborDist"]
    a, fn = case["args"], case["fn"]
    if fn == "warpTerrain":
        return oracle.warp_terrain(mesh, e0, xyz, a["seed"], a["strength"], g["hotspot"] if "hot" in name else None)
    if fn == "smoothElevation":
        return oracle.smooth_elevation(mesh, e0, oc, a["iterations"], a["strength"])
    if fn == "sharpenRidges":
        return oracle.sharpen_ridges(mesh, e0, oc, a["iterations"], a["strength"])
    if fn == "applySoilCreep":
        return oracle.soil_creep(mesh, e0, oc, a["iterations"], a["strength"])
    if fn == "priorityFloodCarve":
        return oracle.priority_flood_carve(mesh, e0, oc, a["carveStrength"])
    assert fn == "erodeComposite"
    return oracle.erode_composite(mesh, e0, xyz, oc, a["hIters"], a["K"], a["m"], a["dt"], a["tIters"], a["talusSlope"],
                                  a["kThermal"], a["gIters"], a["glacialStrength"], nd)


@pytest.mark.parametrize("tag", POST_TAGS)
def test_terrain_post_bit_exact(oracle, tag):
    g = load_golden(f"post_{tag}")
    assert np.array_equal(oracle.synthetic_terrain(g["xyz"], float(g["seed"])), g["elevation0"])
    for name, case in golden_cases(g).items():
        got = run_oracle_case(oracle, g, name, case)
        ref = g["ref_" + name]
        assert np.array_equal(got, ref), f"{tag}/{name}: {(got != ref).sum()} cells differ, max {np.abs(got - ref).max():.3e}"
        assert (ref != g["elevation0"]).any(), f"{tag}/{name}: case is a no-op"


@pytest.mark.parametrize("name", __import__("plates_common").PLATE_CASES)
def test_plate_projection_bit_exact(oracle, name):
    """projectCoarsePlates + smoothAndReconnectPlates vs the reference's own outputs (plate ids are integers: exact)."""
    from plates_common import plate_case
    c = plate_case(name)
    om, oc = oracle.Mesh(c["mesh"].adjOffset, c["mesh"].adjList), oracle.Mesh(c["cmesh"].adjOffset, c["cmesh"].adjList)
    proj = oracle.project_coarse_plates(om, c["xyz"], oc, c["cxyz"], c["coarse_r_plate"], c["meta"]["seed"], c["meta"]["P"])
    assert np.array_equal(proj, c["projected"]), int((proj != c["projected"]).sum())
    sm = oracle.smooth_reconnect_plates(om, c["projected"], c["seeds"], c["meta"]["passes"])
    assert np.array_equal(sm, c["smoothed"]), int((sm != c["smoothed"]).sum())
