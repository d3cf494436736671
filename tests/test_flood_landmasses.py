"""priorityFloodCarve's host stage, pass 1 with one heap per landmass (csrc/flood_host.cc: flood_pass1_landmasses)
against the oracle's single-heap walk (reference: js/terrain-post.js:59-215).

The landmass route must give the reference's elevations bit for bit whatever the reference's heap does with equal
keys: it either proves that no equal-key decision can matter (tie groups / contested cells / open parents) or hands
pass 1 to the serial walk.  These tests drive it through the test-only emulator library (the same flood_host.cc the
product links) on CPU: ordinary terrain (no fallback expected), islands and lakes, and terrain quantised so that
thousands of keys collide (contested cells, fallbacks)."""
import ctypes as C
import subprocess

import numpy as np

from hooks import del_hook, set_hook
import pytest

from conftest import POST_TAGS, REPO, golden_cases, load_golden

EMU_DIR = REPO / "tests" / "emu"

STAT_NAMES = "calls serialPass1 tieGroups contested openParents unresolved pathRedo pass1Ms pass23Ms replays replayedLandmasses".split()


@pytest.fixture(scope="module")
def emu():
    subprocess.run(["make", "-s", "-C", str(EMU_DIR)], check=True)
    L = C.CDLL(str(EMU_DIR / "_build" / "libemu.so"))
    p = C.c_void_p
    L.emu_flood_host.argtypes = [C.c_int32, p, p, p, p, p, C.c_double, C.c_int32, C.c_int32, p]
    L.emu_flood_shares.argtypes = [C.c_int32, p, p, p, p, p, p, C.c_int32, C.c_double, C.c_int32, p]
    return L


def P(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def flood(emu, mesh_off, mesh_adj, xyz, e0, oc, cs, mode):
    e = e0.copy()
    st = np.zeros(11)
    emu.emu_flood_host(mesh_off.size - 1, P(mesh_off), P(mesh_adj), P(xyz), P(e), P(oc), cs, mode + 10, 1, P(st))      # + 10: all eleven statistics
    return e, dict(zip(STAT_NAMES, st.tolist()))


@pytest.mark.parametrize("tag", POST_TAGS)
def test_landmass_flood_on_reference_goldens(emu, tag):
    g = load_golden(f"post_{tag}")
    off, adj, e0, oc, xyz = (g[k] for k in ("adjOffset", "adjList", "elevation0", "isOcean", "xyz"))
    n = 0
    for name, c in golden_cases(g).items():
        if c["fn"] != "priorityFloodCarve":
            continue
        e, st = flood(emu, off, adj, xyz, e0, oc, c["args"]["carveStrength"], 1)
        assert np.array_equal(e, g["ref_" + name]), (tag, name, st)
        n += 1
    assert n > 0


@pytest.mark.parametrize("cells,seed", [(20000, 3), (200000, 1), (200000, 7)])
def test_landmass_flood_equals_oracle(emu, oracle, cells, seed):
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(cells, 0.75, seed)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    e0 = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, seed), xyz, seed, 0.75)
    oc = (e0 <= 0).astype(np.uint8)
    for cs in (0.5, 0.85):
        ref = oracle.priority_flood_carve(om, e0, oc, cs)
        e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, cs, 1)
        assert np.array_equal(e, ref), (cs, int((e != ref).sum()), st)
        assert st["serialPass1"] == 0, st          # ordinary terrain: the landmass route vouches for itself
        e_serial, _ = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, cs, 0)
        assert np.array_equal(e_serial, ref)


def test_big_landmass_hands_its_trees_out_in_chunks(emu, oracle):
    """A landmass of >= 32 768 cells hands its trees out in chunks to the workers that have run out of landmasses (flood_landmass_pipeline: BigJob).
    700 k cells: the largest landmass is well above the threshold; fresh and eroded terrain, both carve strengths; == oracle bit for bit, several
    times over (the workers race for chunks differently every time)."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(700000, 0.75, 3)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    e0 = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, 3), xyz, 3, 0.75)
    oc = (e0 <= 0).astype(np.uint8)
    from planet_heightmap_generation_amd import decomposed as D
    plan = D.plan_landmasses(mesh, oc, 1)
    assert plan.largest >= 32768, plan.largest
    eroded = oracle.erode_composite(om, e0, xyz, oc, 6, 3e-4, 0.5, 1.0, 6, 1.16, 0.015, 1, 0.5, nd)
    for field, cs in ((e0, 0.5), (eroded, 0.85)):
        ref = oracle.priority_flood_carve(om, field, oc, cs)
        for rep in range(3):
            e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, field, oc, cs, 1)
            assert np.array_equal(e, ref), (cs, rep, int((e != ref).sum()), st)
            assert st["serialPass1"] == 0, st


@pytest.mark.parametrize("quant", [64, 1024, 1 << 16])
def test_landmass_flood_under_key_collisions(emu, oracle, quant):
    """Quantised heights: cells of one level differ only by their noise term, pits fill in EPS steps, and equal f32
    keys sit in the heap together all the time.  Whatever the route decides (vouch, open parents, serial walk) the
    elevations must be the oracle's."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(30000, 0.75, 5)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    e0 = oracle.synthetic_terrain(xyz, 5)
    eq = (np.round(e0 * quant) / quant).astype(np.float32)
    oc = (eq <= 0).astype(np.uint8)
    ref = oracle.priority_flood_carve(om, eq, oc, 0.5)
    e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, eq, oc, 0.5, 1)
    assert np.array_equal(e, ref), (quant, int((e != ref).sum()), st)
    assert st["serialPass1"] == 0, st              # undecided landmasses go through the replay of the single heap, never the serial walk
    print(quant, st)


@pytest.mark.parametrize("cells,seed", [(60000, 4), (200000, 1)])
def test_replay_of_the_single_heap(emu, oracle, monkeypatch, cells, seed):
    """The decision procedure for equal keys that matter (flood_host.cc: replay_dirty_landmasses): the landmass given by
    hook flood_force_dirty is treated as undecided, i.e. walked again inside a replay of the reference's single heap in which
    every other landmass only repeats its known pushes.  Whichever landmass is redone that way — the largest, a middle one,
    a tiny one — the elevations are the oracle's, and nothing falls back to the serial walk."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(cells, 0.75, seed)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    e0 = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, seed), xyz, seed, 0.75)
    oc = (e0 <= 0).astype(np.uint8)
    ref = oracle.priority_flood_carve(om, e0, oc, 0.5)
    for k in (0, 1, 5, 40):
        set_hook(monkeypatch, "flood_force_dirty", str(k))
        e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, 0.5, 1)
        assert np.array_equal(e, ref), (k, int((e != ref).sum()), st)
        assert st["serialPass1"] == 0 and st["replays"] == 1 and st["replayedLandmasses"] >= 1, (k, st)
    del_hook(monkeypatch, "flood_force_dirty")


def test_landmass_flood_constructed_equal_keys(emu, oracle):
    """Equal keys forced next to each other: the noise term is a function of the cell id only, so giving every land
    cell the height K - noise(cell) (rounded) makes whole neighbourhoods pop with (nearly) one key.  Exercises
    contested cells; the result must still be the single heap's."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(12000, 0.75, 2)
    N = mesh.numRegions
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    base = oracle.synthetic_terrain(xyz, 2)
    r = np.arange(N, dtype=np.float64)
    h = np.mod(r * 2654435761.0, 4294967296.0).astype(np.uint64).astype(np.uint32)
    x = ((h >> np.uint32(16)) ^ h).astype(np.int32).astype(np.float64)
    h = np.mod(x * 73244475.0, 4294967296.0).astype(np.int64).astype(np.uint32)
    h = (h >> np.uint32(16)) ^ h
    noise = h.astype(np.float64) / 4294967295.0 * 0.01
    total = replays = 0
    for level in (0.05, 0.3):
        e0 = np.where(base > 0, np.float32(level) - noise.astype(np.float32), np.float32(-0.1)).astype(np.float32)
        e0 = np.where((base > 0) & (e0 <= 0), np.float32(1e-3), e0).astype(np.float32)
        oc = (e0 <= 0).astype(np.uint8)
        ref = oracle.priority_flood_carve(om, e0, oc, 0.5)
        e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, 0.5, 1)
        assert np.array_equal(e, ref), (level, int((e != ref).sum()), st)
        assert st["serialPass1"] == 0, st
        total += st["tieGroups"]
        replays += st["replays"]
        print(level, st)
    assert total > 0 and replays > 0          # thousands of contested cells: decided by the replay of the single heap


def _cell_noise(r):
    r = np.asarray(r, dtype=np.float64)
    h = np.mod(r * 2654435761.0, 4294967296.0).astype(np.uint64).astype(np.uint32)
    x = ((h >> np.uint32(16)) ^ h).astype(np.int32).astype(np.float64)
    h = np.mod(x * 73244475.0, 4294967296.0).astype(np.int64).astype(np.uint32)
    h = (h >> np.uint32(16)) ^ h
    return h.astype(np.float64) / 4294967295.0 * 0.01


def _height_with_key(key, cell):
    """an f32 height h with f32(h + noise(cell)) == key, or None"""
    nz = float(_cell_noise(cell))
    h = np.float32(float(key) - nz)
    for _ in range(8):
        for cand in (h, np.nextafter(h, np.float32(1)), np.nextafter(h, np.float32(-1))):
            if np.float32(float(cand) + nz) == key:
                return np.float32(cand)
        h = np.nextafter(h, np.float32(1))
    return None


def test_open_parents_are_vouched_for_or_redone(emu, oracle):
    """Two coastal seeds A, B with EQUAL keys next to a higher inland cell x: the reference's heap decides which of
    them claims x; a per-landmass heap cannot know.  Without a pit behind x the choice changes only drainTo[x] and the
    landmass route must accept it (open parent, no serial walk); with a pit behind x the carve path runs through x
    towards A or B, the elevations depend on the choice, and the route must redo the call with the serial walk.
    Either way the elevations are the oracle's."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(20000, 0.75, 3)
    off, adj = mesh.adjOffset, mesh.adjList
    om = oracle.Mesh(off, adj)
    e0 = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, 3), xyz, 3, 0.75)
    oc = (e0 <= 0).astype(np.uint8)
    nbs = lambda r: adj[off[r]:off[r + 1]]
    coastal = np.array([oc[r] == 0 and oc[nbs(r)].any() for r in range(mesh.numRegions)])
    accepted = redone = tried = 0
    for x in range(mesh.numRegions):
        if oc[x] or coastal[x]:
            continue
        cn = [int(r) for r in nbs(x) if coastal[r]]
        if len(cn) < 2:
            continue
        A, B = cn[0], cn[1]
        for with_pit in (False, True):
            e = e0.copy()
            e[A] = np.float32(0.02)
            key = np.float32(float(e[A]) + float(_cell_noise(A)))
            hb = _height_with_key(key, B)
            if hb is None or hb <= 0:
                continue
            e[B] = hb
            e[x] = np.float32(0.05)
            if with_pit:
                ys = [int(r) for r in nbs(x) if not oc[r] and not coastal[r]]
                if not ys:
                    continue
                y = ys[0]
                for r in nbs(y):
                    if r != x and not oc[r]:
                        e[r] = max(e[r], np.float32(0.2))
                e[y] = np.float32(0.03)
            ref = oracle.priority_flood_carve(om, e, oc, 0.5)
            got, st = flood(emu, off, adj, xyz, e, oc, 0.5, 1)
            assert np.array_equal(got, ref), (x, with_pit, int((got != ref).sum()), st)
            if st["tieGroups"] > 0 and st["contested"] > 0:
                if st["serialPass1"] == 0 and st["openParents"] > 0:
                    accepted += 1
                if st["pathRedo"] > 0:
                    redone += 1
        tried += 1
        if tried >= 25:
            break
    assert accepted > 0 and redone > 0, (accepted, redone, tried)


def flood_shares(emu, mesh, xyz, e0, oc, shares, cs, exchange=True):
    """one flood call of a planet dealt to `shares` landmass shares (decomposed.plan_landmasses), merged"""
    from planet_heightmap_generation_amd import decomposed as D
    plan = D.plan_landmasses(mesh, oc, shares)
    e = e0.copy()
    st = np.zeros(4)
    owner = np.ascontiguousarray(plan.owner, np.int32)
    emu.emu_flood_shares(mesh.numRegions, P(mesh.adjOffset), P(mesh.adjList), P(xyz), P(e), P(oc), P(owner), shares, cs, 1 if exchange else 0, P(st))
    return e, dict(gathers=st[0], whole_planet_floods=st[1], replays=st[2], received=st[3])


@pytest.mark.parametrize("shares", [2, 5])
def test_shares_pool_their_heights_when_equal_keys_matter(emu, oracle, shares):
    """The landmass decomposition's flood (flood_host.cc: flood_host_passes_exchange).  A share's flood is the planet's as long
    as no equal-key decision matters; when one does, the answer lies in the reference's single heap over the WHOLE planet
    (js/terrain-post.js:131-147), other shares' landmasses included.  Constructed terrain on which thousands of decisions
    matter: with the exchange the merged field is the oracle's bit for bit; a share that replays only its own landmasses
    (no exchange) is NOT — the case round 3 measured at 40 M cells."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(12000, 0.75, 2)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    base = oracle.synthetic_terrain(xyz, 2)
    noise = _cell_noise(np.arange(mesh.numRegions))
    differs_without = 0
    for level in (0.05, 0.3):
        e0 = np.where(base > 0, np.float32(level) - noise.astype(np.float32), np.float32(-0.1)).astype(np.float32)
        e0 = np.where((base > 0) & (e0 <= 0), np.float32(1e-3), e0).astype(np.float32)
        oc = (e0 <= 0).astype(np.uint8)
        ref = oracle.priority_flood_carve(om, e0, oc, 0.5)
        got, st = flood_shares(emu, mesh, xyz, e0, oc, shares, 0.5)
        assert np.array_equal(got, ref), (level, int((got != ref).sum()), st)
        # ONE share floods the whole planet — the undecided one with the lowest land cell — and hands the land heights back to the others
        assert st["gathers"] == shares and st["whole_planet_floods"] == 1 and st["replays"] == 1 and st["received"] >= 1, st
        alone, _ = flood_shares(emu, mesh, xyz, e0, oc, shares, 0.5, exchange=False)
        differs_without += int((alone != ref).sum())
    assert differs_without > 0


@pytest.mark.parametrize("shares", [3, 8])
def test_shares_without_open_decisions_do_not_gather(emu, oracle, shares):
    """ordinary terrain: every share vouches for its landmasses, one flag round, no heights move"""
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(60000, 0.75, 4)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    e0 = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, 4), xyz, 4, 0.75)
    oc = (e0 <= 0).astype(np.uint8)
    ref = oracle.priority_flood_carve(om, e0, oc, 0.85)
    got, st = flood_shares(emu, mesh, xyz, e0, oc, shares, 0.85)
    assert np.array_equal(got, ref), (int((got != ref).sum()), st)
    assert st["gathers"] == 0 and st["whole_planet_floods"] == 0, st


def test_more_shares_than_landmasses(emu, oracle):
    """Shares without any land still answer the flood exchange's collectives (phase 0 always, phase 1 when somebody is undecided):
    a planet with ~46 landmasses dealt to 64 shares, on terrain where equal keys matter, merged == oracle."""
    from planet_heightmap_generation_amd import decomposed as D
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(12000, 0.75, 2)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    base = oracle.synthetic_terrain(xyz, 2)
    noise = _cell_noise(np.arange(mesh.numRegions))
    land = base > 0.35                                     # only the highest ground stays land: few landmasses
    e0 = np.where(land, np.float32(0.3) - noise.astype(np.float32), np.float32(-0.1)).astype(np.float32)
    oc = (e0 <= 0).astype(np.uint8)
    shares = 64
    plan = D.plan_landmasses(mesh, oc, shares)
    assert 0 < plan.num_landmasses < shares and any(c.size == 0 for c in plan.cells)
    ref = oracle.priority_flood_carve(om, e0, oc, 0.5)
    got, st = flood_shares(emu, mesh, xyz, e0, oc, shares, 0.5)
    assert np.array_equal(got, ref), (int((got != ref).sum()), st)


@pytest.mark.parametrize("chains_min", [2, 64, 0])
def test_chain_form_of_the_carve_pass_equals_oracle(emu, oracle, chains_min, monkeypatch):
    """Pass 2 of a big drainage tree runs on a chain-ordered copy of the heights (flood_host.cc: tree_pass2_chains; by default for trees of
    >= 2048 cells, which small test planets rarely hold).  hook flood_chains_min is read per call: every tree of >= 2 / >= 64 cells takes
    the chain form, 0 = none does; the elevations must be the oracle's either way, on ordinary and on quantised (tie-heavy) terrain,
    through the landmass pipeline, the two-phase route and the serial walk."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    set_hook(monkeypatch, "flood_chains_min", str(chains_min))
    mesh, xyz, nd = S.build_sphere(60000, 0.75, 11)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    base = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, 11), xyz, 11, 0.75)
    for e0 in (base, (np.floor(base * 256) / 256).astype(np.float32)):
        oc = (e0 <= 0).astype(np.uint8)
        for cs in (0.5, 0.85):
            ref = oracle.priority_flood_carve(om, e0, oc, cs)
            for mode in (1, 0):
                e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, cs, mode)
                assert np.array_equal(e, ref), (chains_min, cs, mode, int((e != ref).sum()), st)


@pytest.mark.parametrize("ring_min", [1, 1 << 30])
def test_both_walk_queues_equal_oracle(emu, oracle, ring_min, monkeypatch):
    """The landmass walks pop from a ring of key buckets (flood_host.cc: RingQueue; landmasses of >= 4096 cells by default) or from the
    4-ary heap.  hook flood_ring_min is read per call: every landmass / none on the ring.  Fresh terrain, terrain after erosion (most keys
    raised: the heap grows large) and quantised terrain (equal keys everywhere: the tie bookkeeping reads the queue's next key)."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    set_hook(monkeypatch, "flood_ring_min", str(ring_min))
    mesh, xyz, nd = S.build_sphere(50000, 0.75, 13)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    fresh = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, 13), xyz, 13, 0.75)
    oc = (fresh <= 0).astype(np.uint8)
    eroded = oracle.erode_composite(om, fresh, xyz, oc, 12, 3e-4, 0.5, 1.0, 12, 1.16, 0.015, 2, 0.5, nd)
    quant = np.where(oc == 1, fresh, np.maximum(np.floor(fresh * 512) / 512, 1.0 / 512)).astype(np.float32)
    for name, e0 in (("fresh", fresh), ("eroded", eroded), ("quantised", quant)):
        for cs in (0.5, 0.85):
            ref = oracle.priority_flood_carve(om, e0, oc, cs)
            e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, cs, 1)
            assert np.array_equal(e, ref), (name, ring_min, cs, int((e != ref).sum()), st)


@pytest.mark.parametrize("stop", [0.02, 0.1, 0.35, 5.0])
def test_replay_stopped_at_a_level_and_resumed_per_landmass(emu, oracle, monkeypatch, stop):
    """The replay of the single heap ends once the heap's smallest key has passed the last tie level that matters; each undecided landmass
    then finishes its walk on a queue of its own, seeded with the entries the single heap held for it (flood_host.cc:
    flood_landmass_pipeline).  hook flood_replay_stop puts that level anywhere: for landmasses that are only FORCED to be undecided
    (hook flood_force_dirty: no equal-key decision of theirs matters) every level must give the oracle's elevations — below the first seed
    (the replay pops nothing), in the middle, above every key (never stops).  On ordinary and on quantised terrain."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(80000, 0.75, 6)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    base = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, 6), xyz, 6, 0.75)
    oc = (base <= 0).astype(np.uint8)
    quant = np.where(oc == 1, base, np.maximum(np.floor(base * 2048) / 2048, 1.0 / 2048)).astype(np.float32)
    set_hook(monkeypatch, "flood_replay_stop", str(stop))
    for e0 in (base, quant):
        ref = oracle.priority_flood_carve(om, e0, oc, 0.85)
        for k in (0, 2):
            set_hook(monkeypatch, "flood_force_dirty", str(k))
            e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, 0.85, 1)
            assert np.array_equal(e, ref), (stop, k, int((e != ref).sum()), st)
            assert st["serialPass1"] == 0 and st["replays"] == 1, st


@pytest.mark.parametrize("permille", [0, 250, 600, 1000])
def test_replay_keeps_the_decided_prefix_of_an_undecided_landmass(emu, oracle, monkeypatch, permille):
    """Inside the replay of the single heap an undecided landmass is walked for real only from its first tie group that holds a contested
    cell; the cells its own walk popped before that (flood_host.cc: PopLog) keep their claims and push like cells of a decided landmass.
    For a landmass that is only FORCED to be undecided every cut of its pops must do (hook flood_force_prefix, in permille of the pops; 1000 =
    nothing is walked for real), with and without a stop level in the middle of the prefix, on ordinary and on quantised terrain."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(80000, 0.75, 6)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    base = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, 6), xyz, 6, 0.75)
    oc = (base <= 0).astype(np.uint8)
    quant = np.where(oc == 1, base, np.maximum(np.floor(base * 2048) / 2048, 1.0 / 2048)).astype(np.float32)
    set_hook(monkeypatch, "flood_force_prefix", str(permille))
    for e0 in (base, quant):
        ref = oracle.priority_flood_carve(om, e0, oc, 0.85)
        for k, stop in ((0, None), (0, 0.05), (2, 0.1), (7, None)):
            set_hook(monkeypatch, "flood_force_dirty", str(k))
            if stop is None:
                del_hook(monkeypatch, "flood_replay_stop")
            else:
                set_hook(monkeypatch, "flood_replay_stop", str(stop))
            e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, 0.85, 1)
            assert np.array_equal(e, ref), (permille, k, stop, int((e != ref).sum()), st)
            assert st["serialPass1"] == 0 and st["replays"] == 1, st


def test_prefix_of_a_really_undecided_landmass(emu, oracle, monkeypatch, capfd):
    """Equal keys that matter (lowlands of one key, see the next test): the undecided landmasses have a first contested tie group of their
    own, their pops before it are kept (the log names them), and the elevations are the oracle's with the prefix rule on and off."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    import re
    monkeypatch.setenv("WO_FLOOD_TIMING", "1")
    mesh, xyz, nd = S.build_sphere(60000, 0.75, 2)
    N = mesh.numRegions
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    base = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, 2), xyz, 2, 0.75)
    noise = _cell_noise(np.arange(N)).astype(np.float32)
    oc = (base <= 0).astype(np.uint8)
    kept, rows = 0, []
    for lo, cut in ((0.05, 0.08), (0.10, 0.12), (0.0, 0.06)):
        band = (base > lo) & (base < cut)
        e0 = np.where(band, np.float32(cut) - noise, base).astype(np.float32)
        e0 = np.where((base > 0) & (e0 <= 0), np.float32(1e-3), e0).astype(np.float32)
        ref = oracle.priority_flood_carve(om, e0, oc, 0.5)
        for on in ("1", "0"):
            set_hook(monkeypatch, "flood_prefix", on)
            capfd.readouterr()
            e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, 0.5, 1)
            log = capfd.readouterr().err
            assert np.array_equal(e, ref), (lo, cut, on, int((e != ref).sum()), st)
            assert st["replays"] == 1 and st["unresolved"] > 0 and st["serialPass1"] == 0, st
            lens = [int(x) for x in re.findall(r"(\d+) pops before its first tie group", log)]
            assert (len(lens) > 0) == (on == "1"), log[-400:]
            real = [int(x) for x in re.findall(r"replay: (\d+) cells walked for real", log)]
            rows.append((lo, cut, on, "prefix pops", sum(lens), "of", st["replayedLandmasses"], "landmasses; walked for real", real))
            if on == "1":
                kept += sum(lens)
                real_on = real[0]
            else:
                assert real_on <= real[0] and (sum(lens_on) == 0 or real_on < real[0]), (real_on, real[0], lens_on)
            lens_on = lens
    print(*rows, sep="\n")
    assert kept > 0, "no undecided landmass had a decided prefix"


def test_replay_stops_after_the_last_tie_that_matters(emu, oracle, monkeypatch, capfd):
    """Equal keys that matter, all of them LOW: the land below a cut gets the height cut - noise(cell) (one key for whole lowlands: hundreds
    of undecided contested cells), the rest keeps its ordinary terrain.  The replay of the single heap must stop at the cut and the ~30
    undecided landmasses finish on their own queues (cut 0.03, 0.06); with the cut at 0.12 a resumed walk meets a contested cell and the
    replay is run again to the end.  Either way: the oracle's elevations."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    monkeypatch.setenv("WO_FLOOD_TIMING", "1")
    mesh, xyz, nd = S.build_sphere(60000, 0.75, 2)
    N = mesh.numRegions
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    base = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, 2), xyz, 2, 0.75)
    r = np.arange(N, dtype=np.float64)
    h = np.mod(r * 2654435761.0, 4294967296.0).astype(np.uint64).astype(np.uint32)
    x = ((h >> np.uint32(16)) ^ h).astype(np.int32).astype(np.float64)
    h = np.mod(x * 73244475.0, 4294967296.0).astype(np.int64).astype(np.uint32)
    h = (h >> np.uint32(16)) ^ h
    noise = (h.astype(np.float64) / 4294967295.0 * 0.01).astype(np.float32)
    oc = (base <= 0).astype(np.uint8)
    stopped_and_resumed = ran_again = 0
    for cut in (0.03, 0.06, 0.12):
        e0 = np.where((base > 0) & (base < cut), np.float32(cut) - noise, base).astype(np.float32)
        e0 = np.where((base > 0) & (e0 <= 0), np.float32(1e-3), e0).astype(np.float32)
        ref = oracle.priority_flood_carve(om, e0, oc, 0.5)
        capfd.readouterr()
        e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, 0.5, 1)
        log = capfd.readouterr().err
        assert np.array_equal(e, ref), (cut, int((e != ref).sum()), st)
        assert st["replays"] == 1 and st["unresolved"] > 0 and st["serialPass1"] == 0, st
        assert "stopped after" in log and "resumed on their own queues" in log, log[-600:]
        if "full replay" in log:
            ran_again += 1
            assert "ran to the end" in log
        else:
            stopped_and_resumed += 1
    assert stopped_and_resumed >= 1, "no case finished on the landmasses' own queues"


def test_heights_handed_over_in_land_order(emu, oracle, monkeypatch):
    """The mirrored planet keeps its land cells first and in the flood's own (Morton) order, so the flood stage copies just those heights
    and the host passes index them by land index (FloodScratch::landOrder; emulator mode + 100): same elevations as through the full
    array, with and without a replay of the single heap (which re-reads the start heights of the landmasses it walks again)."""
    from planet_heightmap_generation_amd import sphere_mesh as S
    mesh, xyz, nd = S.build_sphere(70000, 0.75, 8)
    om = oracle.Mesh(mesh.adjOffset, mesh.adjList)
    e0 = oracle.warp_terrain(om, oracle.synthetic_terrain(xyz, 8), xyz, 8, 0.75)
    oc = (e0 <= 0).astype(np.uint8)
    for cs in (0.5, 0.85):
        ref = oracle.priority_flood_carve(om, e0, oc, cs)
        e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, cs, 101)
        assert np.array_equal(e, ref), (cs, int((e != ref).sum()), st)
        set_hook(monkeypatch, "flood_force_dirty", "1")
        e, st = flood(emu, mesh.adjOffset, mesh.adjList, xyz, e0, oc, cs, 101)
        del_hook(monkeypatch, "flood_force_dirty")
        assert np.array_equal(e, ref) and st["replays"] == 1, (cs, int((e != ref).sum()), st)



def test_ring_queue_pops_in_key_order_also_after_running_empty(emu):
    """flood_host.cc: RingQueue against the 4-ary heap on random operation sequences shaped like a walk's (keys a little above the level
    reached, some below it, a few far above the ring's window) in which the queue runs EMPTY again and again and is refilled — as in a
    landmass that touches the open ocean in one cell: the popped keys must be the heap's, one for one."""
    emu.emu_flood_queues_differ.restype = C.c_int64
    emu.emu_flood_queues_differ.argtypes = [C.c_int64, C.c_uint64]
    for seed in (1, 2, 3):
        assert emu.emu_flood_queues_differ(400000, seed) == 0
