"""assignElevation through the C ABI on gfx950 against the reference's golden vectors, and BASELINE config 1
end to end (10 k cells, seed 1, UI defaults: assignElevation -> runPostProcessing) against the reference's
final elevation.  Integer outputs (Sets, boundary-derived indices) bit-exact; elevation within 1e-5 RMS
(device tanh/exp/sin/cos/atan2/pow vs V8's), and the number of non-identical cells is reported."""
import numpy as np
import pytest

from conftest import load_golden
from elev_common import load_case

pytestmark = pytest.mark.gpu
CASES = ("elev_N5000_s3_nosuper", "elev_config1_N10000_s1", "elev_N10000_s2")


class _Mesh:
    def __init__(self, off, adj):
        self.adjOffset, self.adjList, self.numRegions = off, adj, off.size - 1


def rms(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt((d * d).mean()))


@pytest.mark.parametrize("name", CASES)
def test_assign_elevation(name):
    from planet_heightmap_generation_amd import elevation as EL, terrain_post as TP
    g = load_golden(name)
    meta, ids, vec, dens, is_ocean, sup = load_case(g)
    mesh = _Mesh(g["adjOffset"], g["adjList"])
    pl = TP.Planet(mesh, g["xyz"], g["neighborDist"])
    res = EL.assign_elevation(mesh, g["xyz"], is_ocean, g["r_plate"], vec, ids, EL.SimplexNoise(meta["seed"]), meta["nMag"], meta["seed"],
                              meta["spread"], dens, sup, planet=pl)
    assert res["mountain_r"] == g["ref_mountain"].tolist() and res["coastline_r"] == g["ref_coastline"].tolist() and res["ocean_r"] == g["ref_ocean"].tolist()
    nbad = int((res["r_elevation"] != g["ref_elevation"]).sum())
    print(f"{name}: elevation non-identical cells {nbad}, rms {rms(res['r_elevation'], g['ref_elevation']):.2e}; "
          f"stress non-identical {int((res['r_stress'] != g['ref_stress']).sum())}; stages {[(t['stage'], round(t['ms'], 2)) for t in res['_timing']]}")
    assert rms(res["r_elevation"], g["ref_elevation"]) < 1e-5
    assert rms(res["r_stress"], g["ref_stress"]) < 1e-5
    for layer in meta["layers"]:
        assert rms(res["debugLayers"][layer], g["ref_dl_" + layer]) < 1e-5, layer
    if name == "elev_config1_N10000_s1":
        # BASELINE config 1: the whole hot path, end to end
        e = res["r_elevation"].copy()
        params = dict(terrainWarp=0.75, smoothing=0.10, glacialErosion=0.5, hydraulicErosion=0.5, thermalErosion=0.1, ridgeSharpening=0.5)
        oc, _ = TP.run_post_processing(pl, e, params, float(meta["seed"]), res["debugLayers"]["hotspot"])
        print(f"config 1 end to end: final elevation non-identical cells {int((e != g['ref_final_elevation']).sum())}, rms {rms(e, g['ref_final_elevation']):.2e}")
        assert np.array_equal(oc, g["ref_final_isOcean"])
        assert rms(e, g["ref_final_elevation"]) < 1e-5
    pl.close()


def test_assign_elevation_large():
    """250 k cells (past the reference's N > 200 000 switches) on the device against the reference's output."""
    from planet_heightmap_generation_amd import elevation as EL, terrain_post as TP
    from test_elevation_emulated import large_case
    g, meta, mesh, xyz, nd, crc = large_case()
    _, ids, vec, dens, is_ocean, sup = load_case(g)
    pl = TP.Planet(mesh, xyz, nd)
    res = EL.assign_elevation(mesh, xyz, is_ocean, g["r_plate"], vec, ids, EL.SimplexNoise(meta["seed"]), meta["nMag"], meta["seed"], meta["spread"],
                              dens, sup, planet=pl)
    assert res["mountain_r"] == g["ref_mountain"].tolist() and res["coastline_r"] == g["ref_coastline"].tolist() and res["ocean_r"] == g["ref_ocean"].tolist()
    nbad = int((res["r_elevation"] != g["ref_elevation"]).sum())
    nlay = sum(crc(res["debugLayers"][l]) != meta["crc_dl_" + l] for l in meta["layers"])
    print(f"250k: elevation non-identical cells {nbad}, rms {rms(res['r_elevation'], g['ref_elevation']):.2e}; layers with a different checksum {nlay}; "
          f"stages {[(t['stage'], round(t['ms'], 1)) for t in res['_timing']]}")
    assert rms(res["r_elevation"], g["ref_elevation"]) < 1e-5 and rms(res["r_stress"], g["ref_stress"]) < 1e-5
    pl.close()
