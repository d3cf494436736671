"""bench.py's N>1 path on CPU: two ranks over gloo (the GPU run uses the same code over RCCL).  The path shards
by planet (ensemble, no data-path collective): ranks must get distinct seeds, the job time must be the MAX over
ranks and the value must aggregate all ranks' cells."""
import json
import socket
import subprocess
import sys

import pytest

from conftest import REPO


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(REPO / "tests" / "dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [json.loads((tmp_path / f"rank{i}.json").read_text()) for i in range(2)]
    assert [x["seed"] for x in res] == [1, 2]
    assert all(abs(x["wall"] - 3.0) < 1e-12 for x in res)               # max over ranks
    assert all(abs(x["value"] - 1000 * 200 * 3 * 2 / 3.0 / 1e6) < 1e-12 for x in res)


def test_bench_line_schema_helpers():
    import bench
    assert bench.whole_job_value(10_000_001, 200, 2, 1, 10.0) == 10_000_001 * 200 * 2 / 10.0 / 1e6
    # the kernels of a pass share the pass's byte budget, and the five passes add up to 243 B x land + 12 B x cells
    kernels = {k for p in bench.PASSES.values() for k in p["kernels"]}
    assert kernels >= {"solve_basin", "solve_setup", "thermal_apply", "receivers", "sort_radix"}
    assert sum(p["budget"][0] for p in bench.PASSES.values()) == 243.0 and sum(p["budget"][1] for p in bench.PASSES.values()) == 12.0
    alt = set(bench.ALTERNATIVES)                             # alternatives of solve_basin / flow_tiles: only one of them runs in a pass
    for name, p in bench.PASSES.items():
        assert sum(v[0] for k, v in p["kernels"].items() if k not in alt) == p["budget"][0], name


@pytest.mark.gpu
def test_bench_two_ranks_emit_the_ensemble_line_and_the_one_planet_leg(tmp_path):
    """bench.py --gpus 2 as the driver launches it, rehearsed on a one-GPU box (gloo, both ranks on device 0, small planets): the
    line's `value` is the ensemble (weak scaling, one planet per rank, no collective) and `one_planet` holds north_star's workload
    — ONE planet over all ranks by landmass with the flood exchange (strong) — each with its own workload, scaling and parity."""
    import os
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           str(REPO / "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--cells", "200000", "--iters", "8", "--steps", "1", "--warmup", "1",
           "--no-cpu", "--no-profile", "--in-flight", "0", "--one-planet-cells", "300000", "--one-planet-iters", "8", "--one-planet-parity-iters", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "one planet per GPU" in out["config"]["workload"]
    leg = out["one_planet"]
    assert leg and leg["scaling"] == "strong" and leg["n_gpus"] == 2 and "ONE planet over 2 GPUs" in leg["config"]["workload"]
    assert leg["config"]["cells"] == 300001 and leg["value"] > 0 and "crc32" in leg["parity"]
    assert len(leg["per_rank"]) == 2 and sum(x["land_cells"] for x in leg["per_rank"]) > 0
    assert abs(out["value"] - 200001 * 8 * 1 * 2 / (out["ms_per_step"] / 1e3) / 1e6) < 1e-6 * out["value"]


def test_committed_pmc_file_covers_the_default_paths_kernels():
    """bench.py refuses (SystemExit) a PMC file without counters for a kernel family the profiled step launched.  The committed file
    must therefore name the kernels of the default route's per-iteration families — checked here, where a stale file costs a test
    and not the round's bench line."""
    import bench
    have = [k.replace("(anonymous namespace)::", "") for k, v in json.loads(bench.PMC_FILE.read_text()).items() if isinstance(v, dict)]
    for fam in ("sort_radix", "sort_keys", "receivers", "flow_final", "solve_setup", "solve_basin", "thermal_excess", "thermal_apply"):
        if fam not in bench.FAMILY_KERNEL:
            continue
        names = bench.FAMILY_KERNEL[fam] if isinstance(bench.FAMILY_KERNEL[fam], tuple) else (bench.FAMILY_KERNEL[fam],)
        assert any(h.startswith(names) for h in have), (fam, names)
