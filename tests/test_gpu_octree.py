"""GPU parity: octree Barnes-Hut (the reference's default --algorithm, src/octree.h) built without locks
(path keys -> radix sort -> breadth-first split) vs the oracle's serial restatement of the reference's lock-based
insertion.  Node numbering differs by construction; everything observable must not: tree size, root monopole
(bit-exact), per-body visit counters (bit-exact: every opening decision equals the reference's) and the forces
(tolerance: polished hardware seeds, terms summed per child slot)."""
import numpy as np
import pytest

from conftest import DT

pytestmark = pytest.mark.gpu

import json
import os

FORCE_TOL = {0: 2e-5, 1: 1e-12}
# Float trajectories of the tree algorithms: MEASURED on the reference itself (tests/golden/calibrate_float_tolerance.py builds it
# -O2 and -Ofast -march=native and runs bvh / octree, theta 0 and 0.5, 10 steps: its two builds drift apart by up to 8.0e-4 of the
# position scale in 2D and 2.6e-6 in 3D); the tests allow 4x that spread, never less than the one-pass float floor 2e-5.
_FLOAT_TOL = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "float_tolerance.json")))
TRAJ_TOL = {0: {2: _FLOAT_TOL["float_tree_trajectory_rel_2d"], 3: _FLOAT_TOL["float_tree_trajectory_rel_3d"]}, 1: {2: 1e-11, 3: 1e-11}}


def maxrel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("dim", [3, 2])
def test_octree_force_phase_vs_oracle(nb, oracle, dtype, dim):
    cases = [("uniform", 1), ("uniform", 2), ("uniform", 3), ("galaxy", 10), ("uniform", 257), ("galaxy", 1000), ("uniform", 4099),
             ("galaxy", 20000)]
    if dim == 3:
        cases.append(("plummer", 1000))
    for wl, n in cases:
        for theta in (0.0, 0.5, 1.0):
            ref = oracle.build_model(dtype, dim, wl, n)
            dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, wl, n))
            dev.octree.enable_counters(True)
            dev.octree_force(theta)
            dev.sync()
            size, mass = dev.octree.info(dev.stream)
            ocnt, osize, omass = oracle.octree_step_force(ref, theta, want_counts=True)
            assert size == osize, (wl, n, "tree size")
            assert mass == omass, (wl, n, "root mass")
            assert np.array_equal(dev.octree.read_counters(dev.stream), ocnt), (wl, n, theta, "visit counters")
            out = dev.download()
            assert maxrel(out.a, ref.a) <= FORCE_TOL[dtype], (wl, n, theta)
            for k in ("m", "x", "v"):  # unlike bvh, the octree does not permute the bodies
                assert np.array_equal(getattr(out, k), getattr(ref, k))
            dev.close()


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("dim", [3, 2])
def test_octree_small_systems_one_block_step(nb, oracle, dtype, dim):
    """Systems of up to 1024 bodies (the reference's default run is 1000) are built by ONE block in one launch (bounds in one, keys
    + sort + numbering + cells + deep cells in one, monopoles in one): tree size, root monopole and per-body visit counters bit for
    bit against the oracle on both sides of that limit, the accelerations within the force tolerance and bitwise equal to the
    breadth-first build's, and 30 steps of the step loop bitwise equal between the two build forms (the tree object is reused:
    whatever one step leaves in the buffers the next must not depend on)."""
    for wl, n in (("uniform", 1), ("uniform", 2), ("uniform", 3), ("galaxy", 257), ("galaxy", 1000), ("uniform", 1024), ("uniform", 1025),
                  ("galaxy", 2048)):
        ref = oracle.build_model(dtype, dim, wl, n)
        ocnt, osize, omass = oracle.octree_step_force(ref, 0.5, want_counts=True)
        acc = []
        for form in (3, 1):
            dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, wl, n))
            dev.octree.set_build(form)
            dev.octree.enable_counters(True)
            dev.octree_force(0.5)
            dev.sync()
            assert dev.octree.info(dev.stream) == (osize, omass), (wl, n, form)
            assert np.array_equal(dev.octree.read_counters(dev.stream), ocnt), (wl, n, form)
            acc.append(dev.download().a.copy())
            assert maxrel(acc[-1], ref.a) <= FORCE_TOL[dtype] or n == 1, (wl, n, form)
            dev.close()
        assert np.array_equal(acc[0], acc[1]), (wl, n)
    if dim == 3:
        # 1300 bodies, 300 of them 2^-7 beside another one: 1094 cells — the insert goes the large path, the monopoles are still one
        # block's (at most 3 rank chunks), and in float that block must take its chunk-by-chunk form (more cells than its 1024 threads)
        rng = np.random.default_rng(3)
        n, t = 1300, (np.float64 if dtype == 1 else np.float32)
        x = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        x[1:600:2] = x[0:600:2] + np.float32(2.0 ** -7)
        hs, ref = nb.HostSystem(dtype, dim, n), oracle.State(dtype, dim, n)
        hs.x[:], hs.m[:], hs.c, hs.dt = x.astype(t), 1.0, 1.0, 1e-3
        ref.x[:], ref.m[:], ref.c, ref.dt = hs.x, hs.m, hs.c, hs.dt
        ocnt, osize, omass = oracle.octree_step_force(ref, 0.5, want_counts=True)
        assert 1024 < (osize - 1) // 8 <= n
        dev = nb.DeviceSystem.from_host(hs)
        dev.octree.enable_counters(True)
        dev.octree_force(0.5)
        dev.sync()
        assert dev.octree.info(dev.stream) == (osize, omass)
        assert np.array_equal(dev.octree.read_counters(dev.stream), ocnt)
        assert maxrel(dev.download().a, ref.a) <= FORCE_TOL[dtype]
        dev.close()
    for n in (1000, 1024):
        out = []
        for form in (3, 1):
            dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, "galaxy", n))
            dev.octree.set_build(form)
            for _ in range(30):
                dev.octree_force(0.5)
                dev.accelerate_step()
            dev.sync()
            dev.octree.info(dev.stream)
            out.append(dev.download())
            dev.close()
        assert np.array_equal(out[0].x, out[1].x) and np.array_equal(out[0].v, out[1].v), n


def test_octree_force_is_deterministic(nb):
    """Two runs on fresh trees give the same bits (the build's node numbering may differ between runs; the walk does not
    depend on it), and a body's result does not depend on which other bodies share its wave."""
    n = 30000
    runs = []
    for _ in range(2):
        dev = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
        dev.octree_force(0.5)
        runs.append(dev.download().a.copy())
        dev.close()
    assert np.array_equal(runs[0], runs[1])


ISA_WALKS = [(1, 3), (0, 3), (0, 2), (1, 2)]


@pytest.mark.parametrize("dtype,dim", ISA_WALKS)
def test_octree_walk_forms_stress(nb, oracle, dtype, dim):
    """Hand-written ISA is only as good as its hazards: repeated walks of one tree of 200 000 uniformly random bodies per theta,
    by BOTH instantiations of the ISA round (with and without the counters: different register allocations around the same
    text), must reproduce the compiler-scheduled kernel's counters and accelerations bit for bit EVERY time.  (Round 3: a
    v_readfirstlane issued by an asm block directly after the VALU write of its source returned the register's previous content —
    csrc/to_sgpr.hpp — which made the float 2D walk without counters wrong and the double 2D walk flip a few decisions per
    walk, different ones each time.)"""
    rng = np.random.default_rng(5 + 2 * dtype + dim)
    n = 200000
    for theta in (0.5, 1.0):
        hs = nb.build_model(dtype, dim, "uniform", n)
        hs.x[:] = rng.uniform(-1, 1, hs.x.shape).astype(hs.x.dtype)
        dev = nb.DeviceSystem.from_host(hs)
        t, st = dev.octree, dev.state()
        t.enable_counters(True)
        t.clear(dev.stream); t.compute_bounds(st, dev.stream); t.insert(st, dev.stream); t.compute_tree(dev.stream)
        t.set_walk(1)
        t.compute_force(st, theta, dev.stream)
        dev.sync()
        cnt, acc = t.read_counters(dev.stream).copy(), dev.download().a.copy()
        ref = oracle.State(dtype, dim, n)   # the walk the forms must reproduce is the reference's, not merely each other's
        ref.m[:], ref.x[:], ref.c, ref.dt = hs.m, hs.x, hs.c, hs.dt
        ocnt, osize, omass = oracle.octree_step_force(ref, theta, want_counts=True)
        assert t.info(dev.stream) == (osize, omass)
        assert np.array_equal(cnt, ocnt), (theta, "visit counters vs the oracle")
        assert maxrel(acc, ref.a) <= FORCE_TOL[dtype], theta
        t.set_walk(2)
        for rep in range(4):
            t.compute_force(st, theta, dev.stream)
            dev.sync()
            assert np.array_equal(t.read_counters(dev.stream), cnt), (theta, rep)
            assert np.array_equal(dev.download().a, acc), (theta, rep)
        t.enable_counters(False)
        for form in (2, 1, 0):
            t.set_walk(form)
            for rep in range(4):
                t.compute_force(st, theta, dev.stream)
                dev.sync()
                assert np.array_equal(dev.download().a, acc), (theta, form, rep, "no counters")
        dev.close()


@pytest.mark.parametrize("dtype,dim", ISA_WALKS)
def test_octree_walk_forms_are_bitwise_equal(nb, dtype, dim):
    """The visit round written as ISA (double and float, 3D and 2D — float octree is the reference's DEFAULT run,
    src/arguments.h:23-30: what runs by default) and the compiler-scheduled kernel
    (nbody_octree_set_walk: 2 / 1) perform the same tests and the same arithmetic in the same order: accelerations and counters equal
    bit for bit — fresh and clustered systems, theta 0 (every node opened: the deepest stacks), a shard window."""
    import os
    rng = np.random.default_rng(7)
    cases = [("galaxy", 50000, 0.5), ("uniform", 20011, 0.0), ("galaxy", 4096, 1.2), ("plummer", 30000, 0.3), ("uniform", 60000, 0.7),
             ("galaxy", 300000, 0.5)]   # theta 0.7 on a dense cube: many lanes inside the guard band of the quick test
    if dim == 2:
        cases = [c for c in cases if c[0] != "plummer"]   # (the reference builds Plummer spheres in 3D only)
    for wl, n, theta in cases:
        res = []
        for form in (2, 1):
            hs = nb.build_model(dtype, dim, wl, n)
            if wl == "uniform":  # tight pairs and a far escaper: near-pair path, cells below the usual depth
                hs.x[1] = hs.x[0] + (1e-9 if dtype == 1 else 1e-5)
                hs.x[3] = 1e3
            dev = nb.DeviceSystem.from_host(hs)
            dev.octree.set_walk(form)
            dev.octree.enable_counters(True)
            dev.octree_force(theta)
            dev.sync()
            full = dev.download().a.copy()
            cnt = dev.octree.read_counters(dev.stream).copy()
            dev.octree.compute_force(dev.state(n // 3, n // 2), theta, dev.stream)
            dev.sync()
            win = dev.download().a[n // 3:n // 3 + n // 2].copy()
            res.append((full, cnt, win))
            dev.close()
        assert np.array_equal(res[0][0], res[1][0]), (wl, n, theta)
        assert np.array_equal(res[0][1], res[1][1]), (wl, n, theta)
        assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][2], res[0][0][n // 3:n // 3 + n // 2]), (wl, n, theta)


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("dim", [3, 2])
def test_octree_deep_and_clustered_trees_vs_oracle(nb, oracle, dtype, dim):
    """Hand-made inputs that drive the build deep: close pairs (split ~19 levels down), two tight far-apart clusters,
    negative coordinates, unequal masses.  Same checks as the generator cases: tree size, root monopole and visit
    counters bit-exact, forces within tolerance."""
    rng = np.random.default_rng(7 + dim + 10 * dtype)
    t = np.float64 if dtype == 1 else np.float32
    n = 600
    x = rng.uniform(-1.0, 1.0, (n, dim))
    x[100:300] = 0.37 + 1e-3 * rng.standard_normal((200, dim))     # tight cluster
    x[300:400] = -0.81 + 2e-4 * rng.standard_normal((100, dim))    # a tighter one, far away
    for k in range(5):                                              # pairs ~2^-18 of the root side apart
        x[400 + 2 * k + 1] = x[400 + 2 * k] + 1.2e-5 * (k + 1)
    m = rng.uniform(0.1, 3.0, n)
    m[::7] *= 50.0
    for theta in (0.0, 0.4, 1.1):
        hs = nb.HostSystem(dtype, dim, n)
        hs.m[:], hs.x[:] = m.astype(t), x.astype(t)
        hs.c, hs.dt = 1.0, 0.01
        ref = oracle.State(dtype, dim, n)
        ref.m[:], ref.x[:] = hs.m, hs.x
        ref.c, ref.dt = hs.c, hs.dt
        dev = nb.DeviceSystem.from_host(hs)
        dev.octree.enable_counters(True)
        dev.octree_force(theta)
        dev.sync()
        size, mass = dev.octree.info(dev.stream)
        ocnt, osize, omass = oracle.octree_step_force(ref, theta, want_counts=True)
        assert (size, mass) == (osize, omass), (theta, size, osize)
        assert np.array_equal(dev.octree.read_counters(dev.stream), ocnt), theta
        assert maxrel(dev.download().a, ref.a) <= FORCE_TOL[dtype], theta
        dev.close()


@pytest.mark.parametrize("dtype,dim", [(1, 3), (1, 2), (0, 3)])
def test_octree_below_the_key_depth_vs_oracle(nb, oracle, dtype, dim):
    """Cells finer than the 21 (3D) / 32 (2D) key levels — what a long run produces once escapers have inflated the root
    cube — are finished by the deep build the reference's way: same tree size, root monopole, visit counters, forces."""
    rng = np.random.default_rng(11 + dim)
    t = np.float64 if dtype == 1 else np.float32
    n = 2000                           # pool = 2^dim * n nodes (src/system.h:30): room for the deep chains below
    x = rng.uniform(-1.0, 1.0, (n, dim))
    x[0] = 9000.0                      # an escaper: root side ~ 1.8e4, key resolution ~ 1e-2 (3D)
    x[1] = -9000.0
    x[100:200] = 0.25 + 2e-4 * rng.standard_normal((100, dim))   # a core far below that resolution
    if dtype == 1:
        for k in range(6):             # pairs 1e-7 .. 1e-9 apart: 35 - 45 levels deep
            x[300 + 2 * k + 1] = x[300 + 2 * k] + 10.0 ** (-7 - k // 2)
    m = rng.uniform(0.5, 2.0, n)
    for theta in (0.0, 0.5):
        hs = nb.HostSystem(dtype, dim, n)
        hs.m[:], hs.x[:] = m.astype(t), x.astype(t)
        hs.c, hs.dt = 1.0, 0.01
        ref = oracle.State(dtype, dim, n)
        ref.m[:], ref.x[:] = hs.m, hs.x
        ref.c, ref.dt = hs.c, hs.dt
        dev = nb.DeviceSystem.from_host(hs)
        dev.octree.enable_counters(True)
        dev.octree_force(theta)
        dev.sync()
        size, mass = dev.octree.info(dev.stream)
        ocnt, osize, omass = oracle.octree_step_force(ref, theta, want_counts=True)
        assert osize > 1 + (1 << dim) * 60, "the case is meant to be deep"
        assert (size, mass) == (osize, omass), (theta, size, osize)
        assert np.array_equal(dev.octree.read_counters(dev.stream), ocnt), theta
        assert maxrel(dev.download().a, ref.a) <= FORCE_TOL[dtype], theta
        dev.close()


def test_octree_randomized_systems_vs_oracle(nb, oracle):
    """Two dozen seeded random systems (clusters over eight decades of scale, far outliers — i.e. cells below the key depth —,
    random theta, both precisions and dimensions): tree size, root monopole and counters bit-exact, force within tolerance."""
    from test_gpu_bvh import _random_system
    rng = np.random.default_rng(20240602)
    for case in range(24):
        dtype, dim = int(rng.integers(0, 2)), int(rng.integers(2, 4))
        n = int(rng.integers(2, 3000))
        theta = float(rng.choice([0.0, 0.2, 0.5, 0.9, 1.4]))
        hs, ref = _random_system(nb, oracle, rng, dtype, dim, n)
        try:
            ocnt, osize, omass = oracle.octree_step_force(ref, theta, want_counts=True)
        except RuntimeError:   # the reference's node pool overflows on this one (near-coincident floats): both must fail
            dev = nb.DeviceSystem.from_host(hs)
            dev.octree_force(theta)
            with pytest.raises(nb.NbodyError, match="node pool exhausted|depth limit"):
                dev.octree.info(dev.stream)
            dev.close()
            continue
        dev = nb.DeviceSystem.from_host(hs)
        dev.octree.enable_counters(True)
        dev.octree_force(theta)
        dev.sync()
        size, mass = dev.octree.info(dev.stream)
        assert (size, mass) == (osize, omass), (case, dtype, dim, n, theta, size, osize)
        assert np.array_equal(dev.octree.read_counters(dev.stream), ocnt), (case, dtype, dim, n, theta)
        assert maxrel(dev.download().a, ref.a) <= FORCE_TOL[dtype], (case, dtype, dim, n, theta)
        dev.close()


def test_octree_theta0_equals_all_pairs(nb):
    """theta = 0 never approximates: the walk reaches every body leaf (README.md:122-129)."""
    n = 3000
    d1 = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    d2 = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    d1.all_pairs_force()
    d2.octree_force(0.0)
    # the octree term is m*d/(sqrt(r2)+eps)^3, all-pairs is m*d/(r2^1.5 + eps): equal to ~eps/r
    assert maxrel(d2.download().a, d1.download().a) <= 1e-9


def test_trajectories_vs_reference_fixtures(nb, golden_positions):
    meta, data = golden_positions
    ran = 0
    for name, case in meta.items():
        if case["algorithm"] != "octree":
            continue
        dtype = DT[case["precision"]]
        ref = data[name + "__frames"]
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, case["dim"], case["workload"], case["n"]))
        scale = np.abs(ref[0]).max()
        k = 1
        for step in range(1, case["steps"] + 1):
            nb.run(dev, "octree", 1, case["theta"])
            if step in case["frame_ids"]:
                assert np.abs(dev.download().x - ref[k]).max() <= TRAJ_TOL[dtype][case["dim"]] * scale, (name, step)
                k += 1
        dev.octree.info(dev.stream)
        dev.close()
        ran += 1
    assert ran >= 30


def test_octree_shard_windows_and_graph(nb):
    n = 9000
    dev = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    dev.octree_force(0.5)
    full = dev.download().a.copy()
    hs = dev.download()
    hs.a[:] = 0
    dev.upload(hs)
    for f, c in ((0, 1000), (1000, 4001), (5001, 3999)):
        dev.octree.compute_force(dev.state(f, c), 0.5, dev.stream)
    assert np.array_equal(dev.download().a, full)
    # a recorded step replays to the same state as direct calls
    d1 = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    nb.run(d1, "octree", 3, 0.5)
    d2 = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    _ = d2.octree
    g = nb.StepGraph(d2, lambda: (d2.octree_force(0.5), d2.accelerate_step()))
    for _ in range(3):
        g.launch()
    d2.sync()
    a, b = d1.download(), d2.download()
    assert np.array_equal(a.x, b.x) and np.array_equal(a.a, b.a)


def test_octree_errors(nb):
    hs = nb.HostSystem(1, 3, 4)
    hs.m[:] = 1
    hs.x[:] = [[0, 0, 0], [1, 1, 1], [1, 1, 1], [2, 0, 1]]  # two coincident bodies: the reference would split until its pool overflows
    hs.c, hs.dt = 1.0, 0.1
    dev = nb.DeviceSystem.from_host(hs)
    dev.octree_force(0.5)
    dev.sync()
    with pytest.raises(nb.NbodyError, match="node pool exhausted|depth limit"):
        dev.octree.info(dev.stream)
    # the flag is sticky across builds (a run that replays recorded steps checks once, at the end) and cleared by the report
    hs2 = dev.download()
    dev.octree_force(0.5)      # bad build
    hs2.x[2] = [1.5, 1.0, 1.25]
    dev.upload(hs2)
    dev.octree_force(0.5)      # good build afterwards
    with pytest.raises(nb.NbodyError, match="node pool exhausted|depth limit"):
        dev.octree.info(dev.stream)
    size, mass = dev.octree.info(dev.stream)
    assert size == 1 + 8 * ((size - 1) // 8) and mass == 4.0
    d2 = nb.DeviceSystem.from_host(nb.build_model(1, 3, "uniform", 50))
    with pytest.raises(nb.NbodyError, match="before nbody_octree_compute_bounds"):
        d2.octree.insert(d2.state(), d2.stream)
    with pytest.raises(nb.NbodyError, match="before nbody_octree_compute_tree"):
        d2.octree.compute_force(d2.state(), 0.5, d2.stream)


def test_octree_one_pass_build_beyond_the_round_limit(nb):
    """Above 4.2 * 10^6 cells the one-pass build's multipole pass takes one launch per level over all ranks (the compacted lists
    of its rounds no longer fit a block's LDS).  4.3 * 10^6 bodies in float: the same tree size, root monopole and accelerations
    as the breadth-first build, bit for bit."""
    n = 4300000
    res = []
    for form in (1, 3):
        dev = nb.DeviceSystem.from_host(nb.build_model(0, 3, "galaxy", n))
        dev.octree.set_build(form)
        dev.octree_force(0.7)
        dev.sync()
        res.append((dev.octree.info(dev.stream), dev.download().a.copy()))
        dev.close()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("dtype", [1, 0])
def test_octree_build_forms_over_many_steps(nb, dtype):
    """Sixty recorded-and-replayed steps of an evolving galaxy: the one-pass build reuses its lists, marks and masks from step to
    step (nothing is cleared in between), so the trajectory must stay bit-identical to the breadth-first build's."""
    outs = []
    for form in (1, 3):
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", 30000))
        dev.octree.set_build(form)
        nb.run(dev, "octree", 60, 0.5)
        dev.sync()
        out = dev.download()
        outs.append((out.x.copy(), out.v.copy(), dev.octree.info(dev.stream)))
        dev.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]


@pytest.mark.parametrize("dtype,dim", [(1, 3), (0, 3), (1, 2), (0, 2)])
def test_octree_build_forms_fuzz(nb, oracle, dtype, dim):
    """The one-pass build against the breadth-first one on geometry that random clouds do not produce: bodies exactly on cell
    boundaries (grid-aligned coordinates: `pos > divide` is false on the boundary), on a line, on a plane, in tight clusters with
    a far escaper, with many equal coordinates, in sizes around the kernels' block sizes.  Tree size, root monopole, counters and
    accelerations bit for bit; a case one form refuses (node pool / depth) must be refused by the others.  (Round 3: the first run
    of this test crashed the BREADTH-FIRST forms — when a level's cell list ran out of room the cells were counted but not stored,
    and the deep build and the multipole pass read list entries past the allocation.)"""
    rng = np.random.default_rng(11 + 3 * dtype + dim)
    t = np.float64 if dtype == 1 else np.float32
    sizes = [2, 3, 5, 8, 9, 63, 64, 65, 255, 256, 257, 511, 1023, 1024, 1025, 2047, 2048, 3000, 4097]
    for case in range(40):
        n = int(sizes[case % len(sizes)])
        kind = case % 8
        x = rng.uniform(-1, 1, (n, dim))
        if kind == 1:    # grid-aligned: multiples of 2^-6, distinct
            side = int(np.ceil(n ** (1.0 / dim))) + 1
            idx = rng.permutation(side ** dim)[:n]
            x = np.stack([(idx // side ** k) % side for k in range(dim)], axis=1) / 64.0
        elif kind == 2:  # a line
            x = np.outer(np.sort(rng.uniform(-1, 1, n)), np.ones(dim))
            x += np.arange(n)[:, None] * 1e-9
        elif kind == 3:  # a plane / an axis-aligned segment in 2D
            x[:, 0] = 0.25
        elif kind == 4:  # tight clusters and an escaper
            centres = rng.uniform(-1, 1, (max(1, n // 16), dim))
            x = centres[rng.integers(0, len(centres), n)] + rng.normal(0, 1e-4, (n, dim))
            x[0] = 50.0
        elif kind == 5:  # many equal coordinates
            x = np.round(x * 4) / 4 + rng.uniform(0, 1e-3, (n, dim)) * (rng.uniform(0, 1, (n, dim)) < 0.3)
            x += np.arange(n)[:, None] * 1e-7
        elif kind == 6:  # powers of two and their negatives
            x = np.ldexp(1.0, rng.integers(-20, 2, (n, dim))) * rng.choice([-1.0, 1.0], (n, dim))
            x += np.arange(n)[:, None] * 1e-6
        hs = nb.HostSystem(dtype, dim, n)
        hs.m[:] = rng.uniform(0.5, 1.5, n).astype(t)
        hs.x[:] = x.astype(t)
        hs.c, hs.dt = 1.0, 1e-3
        res = []
        for form in (1, 3):
            dev = nb.DeviceSystem.from_host(hs)
            dev.octree.set_build(form)
            dev.octree.enable_counters(True)
            dev.octree_force(0.4)
            dev.sync()
            try:
                info = dev.octree.info(dev.stream)
            except nb.NbodyError as e:
                info = ("refused", "node pool" in str(e) or "depth limit" in str(e))
            res.append((info, dev.octree.read_counters(dev.stream).copy(), dev.download().a.copy()))
            dev.close()
        for r in res[1:]:
            assert res[0][0] == r[0], (case, n, kind, res[0][0], r[0])
            if res[0][0][0] != "refused":
                assert np.array_equal(res[0][1], r[1]) and np.array_equal(res[0][2], r[2]), (case, n, kind)
        # ... and against the reference's own insertion (src/octree.h:114-180, restated by the oracle): the same tree size, root
        # monopole and per-body counters for every tree the product accepts, a pool overflow for every one it refuses
        ref = oracle.State(dtype, dim, n)
        ref.m[:], ref.x[:], ref.c, ref.dt = hs.m, hs.x, hs.c, hs.dt
        if res[0][0][0] == "refused":
            with pytest.raises(RuntimeError):
                oracle.octree_step_force(ref, 0.4)
        else:
            ocnt, osize, omass = oracle.octree_step_force(ref, 0.4, want_counts=True)
            assert res[0][0] == (osize, omass), (case, n, kind, res[0][0], (osize, omass))
            assert np.array_equal(res[0][1], ocnt), (case, n, kind, "visit counters vs the oracle")
            assert maxrel(res[0][2], ref.a) <= FORCE_TOL[dtype], (case, n, kind)
        if res[0][0][0] != "refused":
            walks = []
            for walk in (1, 2):   # the compiler-scheduled walk and the visit round written as ISA, on the same degenerate tree
                dev = nb.DeviceSystem.from_host(hs)
                dev.octree.set_walk(walk)
                dev.octree.enable_counters(True)
                dev.octree_force(0.4)
                dev.sync()
                walks.append((dev.octree.read_counters(dev.stream).copy(), dev.download().a.copy()))
                dev.close()
            assert np.array_equal(walks[0][0], walks[1][0]) and np.array_equal(walks[0][1], walks[1][1]), (case, n, kind, "walk forms")
            assert np.array_equal(walks[0][1], res[0][2]), (case, n, kind)


@pytest.mark.parametrize("form", [3, 1])
def test_octree_node_pool_exhausted_above_the_key_depth(nb, oracle, form):
    """Ten pairs 2^-18 of the root side apart make ten chains of ~17 nested cells: 170 sibling groups against the 125 the
    reference's pool holds for 20 bodies (System::max_tree_node_size = max(1000, 8 n) nodes, src/system.h:30).  The oracle's
    insertion overflows; every build form reports it through nbody_octree_info — the one-pass build through the check each cell
    and its parent make on the cell's rank — leaves the cells it cannot place as empty leaves, and the walk of what was built
    terminates with finite accelerations inside its buffer."""
    rng = np.random.default_rng(3)
    hs = nb.HostSystem(1, 3, 20)
    hs.m[:] = 1.0
    base = rng.uniform(-1, 1, (10, 3))
    hs.x[0::2] = base
    hs.x[1::2] = base + 4.0 / (1 << 18)
    hs.c, hs.dt = 1.0, 1e-3
    ref = oracle.State(1, 3, 20)
    ref.m[:], ref.x[:], ref.c = hs.m, hs.x, 1.0
    with pytest.raises(RuntimeError):
        oracle.octree_step_force(ref, 0.5)
    dev = nb.DeviceSystem.from_host(hs)
    dev.octree.set_build(form)
    dev.octree_force(0.5)
    dev.sync()
    with pytest.raises(nb.NbodyError, match="node pool exhausted"):
        dev.octree.info(dev.stream)
    assert np.all(np.isfinite(dev.download().a))
    dev.close()


def test_octree_run_paths_raise_on_device_flags(nb):
    """A flagged build drops mass (cells still holding >= 2 bodies stay empty leaves): the step loops must not integrate on.
    nb.run() reads the sticky device-side flag every OCTREE_CHECK_EVERY steps and after the last step; ShardedOctree does
    the same (check()); the CLI checks after its warm-up steps and at the end (non-zero exit, message on stderr)."""
    import os
    import subprocess
    import tempfile
    import torch
    from conftest import ROOT
    hs = nb.HostSystem(1, 3, 40)
    rng = np.random.default_rng(2)
    hs.m[:] = 1
    hs.x[:] = rng.uniform(-1, 1, hs.x.shape)
    hs.x[7] = hs.x[3]  # two coincident bodies
    hs.c, hs.dt = 1e-6, 1e-3
    dev = nb.DeviceSystem.from_host(hs)
    with pytest.raises(nb.NbodyError, match="node pool exhausted|depth limit"):
        nb.run(dev, "octree", 3, 0.5)
    saved = nb.OCTREE_CHECK_EVERY
    try:
        nb.OCTREE_CHECK_EVERY = 2   # the periodic check fires inside the loop, before the final one
        dev2 = nb.DeviceSystem.from_host(hs)
        calls = []
        orig = dev2.octree.info
        dev2.octree.info = lambda stream=None: (calls.append(1), orig(stream))[1]
        with pytest.raises(nb.NbodyError):
            nb.run(dev2, "octree", 5, 0.5)
        assert len(calls) == 1   # raised at step 2, not at the end
    finally:
        nb.OCTREE_CHECK_EVERY = saved
    sim = nb.parallel.ShardedOctree(hs, 0, 1, theta=0.5, torch_device=torch.device("cuda", 0))
    sim.step()
    with pytest.raises(nb.NbodyError, match="node pool exhausted|depth limit"):
        sim.check()
    # the same system through the CLI (--workload load): f32 file format of src/saving.h:25-68
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "bad.bin")
        with open(path, "wb") as f:
            f.write(np.array([hs.n, 3], np.uint32).tobytes() + np.array([hs.dt, hs.c], np.float32).tobytes())
            rec = np.zeros((hs.n, 7), np.float32)
            rec[:, 0], rec[:, 1:4] = hs.m, hs.x
            f.write(rec.tobytes())
        exe = os.path.join(ROOT, "stdpar-nbody_amd", "bin", "nbody_hip_d3")
        r = subprocess.run([exe, "--workload", "load", path, "--algorithm", "octree", "--precision", "double", "-s", "12"],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and ("node pool exhausted" in r.stderr or "depth limit" in r.stderr), r.stderr


def test_octree_full_size(nb, oracle):
    """N = 1e6 galaxy, theta = 0.5 (the reference's tree benchmark size): whole force phase vs the oracle."""
    n = 1000000
    ref = oracle.build_model(1, 3, "galaxy", n)
    dev = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    dev.octree.enable_counters(True)
    dev.octree_force(0.5)
    dev.sync()
    size, mass = dev.octree.info(dev.stream)
    ocnt, osize, omass = oracle.octree_step_force(ref, 0.5, want_counts=True)
    assert (size, mass) == (osize, omass)
    assert np.array_equal(dev.octree.read_counters(dev.stream), ocnt)
    assert maxrel(dev.download().a, ref.a) <= FORCE_TOL[1]


@pytest.mark.parametrize("n,theta", [(768 * 2048 + 1, 0.5), (1 << 21, 0.8)])
def test_octree_above_the_splitter_sort_range(nb, oracle, n, theta):
    """The octree's path keys go through the same sort entry as the bvh's; above 1 572 864 bodies that is the eight-pass radix
    sort (radix_sort.hpp).  Tree size, root monopole and every body's visit counters bit-exact against the oracle's serial
    insertion (src/octree.h:114-180), force within tolerance."""
    ref = oracle.build_model(1, 3, "galaxy", n)
    dev = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    dev.octree.enable_counters(True)
    dev.octree_force(theta)
    dev.sync()
    size, mass = dev.octree.info(dev.stream)
    ocnt, osize, omass = oracle.octree_step_force(ref, theta, want_counts=True)
    assert (size, mass) == (osize, omass)
    assert np.array_equal(dev.octree.read_counters(dev.stream), ocnt)
    assert maxrel(dev.download().a, ref.a) <= FORCE_TOL[1]
    dev.close()


def test_sharded_octree_torch_path(nb):
    """ShardedOctree (torch tensors' data_ptr() and torch's current stream through the C ABI; every 'rank' rebuilds the
    whole tree and walks it for its own bodies): one rank and two emulated ranks give bitwise the single-context run."""
    import torch
    n, steps = 7001, 3
    dev = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    nb.run(dev, "octree", steps, 0.5)
    ref = dev.download()
    sim = nb.parallel.ShardedOctree(nb.build_model(1, 3, "galaxy", n), 0, 1, theta=0.5, torch_device=torch.device("cuda", 0))
    for _ in range(steps):
        sim.step()
    torch.cuda.synchronize()
    x, v, a = sim.gather_state()
    assert np.array_equal(x, ref.x) and np.array_equal(v, ref.v) and np.array_equal(a, ref.a)
    sims = [nb.parallel.ShardedOctree(nb.build_model(1, 3, "galaxy", n), r, 2, theta=0.5, torch_device=torch.device("cuda", 0))
            for r in range(2)]
    for s in sims:
        s.exchange = False  # no process group here; the exchange is emulated below
    for _ in range(steps):
        for s in sims:
            s.step()
        torch.cuda.synchronize()
        for s in sims:      # what all_gather_into_tensor does
            for o in sims:
                s.x[o.first:o.first + o.count] = o.x[o.first:o.first + o.count]
    torch.cuda.synchronize()
    assert np.array_equal(sims[0].x.cpu().numpy(), ref.x) and np.array_equal(sims[1].x.cpu().numpy(), ref.x)
    assert np.array_equal(np.concatenate([s.v.cpu().numpy() for s in sims]), ref.v)
    assert np.array_equal(np.concatenate([s.a.cpu().numpy() for s in sims]), ref.a)


def _deep_chain(nb, levels):
    """A legal but pathological system: the root cube is [-8, 8]^3, centred on the origin (bodies at -7 and +7 in every
    coordinate are the scalar extremes, +-1: src/octree.h:93-112), and a chain of cells [0, 16/2^k]^3, k = 1..levels, converges
    on the origin.  In every cell of the chain two octants hold a
    pair of bodies (cells that a theta = 0 walk must open) and the low octant continues the chain, so a walk that descends
    the chain first — child 0 is popped first — holds 2 more pending cells per level."""
    pts = [(-7.0, -7.0, -7.0), (7.0, 7.0, 7.0)]
    for k in range(1, levels + 1):
        s = 16.0 / 2.0 ** k
        pts += [(0.75 * s, 0.25 * s, 0.25 * s), (0.80 * s, 0.20 * s, 0.20 * s),
                (0.25 * s, 0.75 * s, 0.25 * s), (0.20 * s, 0.80 * s, 0.20 * s)]
    s = 16.0 / 2.0 ** levels
    pts.append((0.1 * s, 0.1 * s, 0.1 * s))
    hs = nb.HostSystem(nb.F64, 3, len(pts))
    hs.x[:] = np.array(pts)
    hs.m[:] = 1.0
    hs.c, hs.dt = 1.0, 1e-3
    return hs


def _walk_with_canaries(nb, hs, theta, form, budget=0):
    """One octree force phase with the accelerations written into the middle of a larger canary-filled buffer: returns
    (a, the NbodyError message of nbody_octree_info or None) after checking that nothing outside a's rows was written."""
    import ctypes as C
    import torch
    dev = torch.device("cuda", 0)
    n, pad, canary = hs.n, 4096, -7.25
    to = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    m, x, v = to(hs.m), to(hs.x), to(hs.v)
    big = torch.full((n + 2 * pad, 3), canary, dtype=torch.float64, device=dev)
    ao = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    st = nb.nbody_state()
    st.m, st.x, st.v, st.ao = m.data_ptr(), x.data_ptr(), v.data_ptr(), ao.data_ptr()
    st.a = big[pad:].data_ptr()
    st.dt, st.c, st.sz, st.first, st.count, st.dtype, st.dim = hs.dt, hs.c, n, 0, n, nb.F64, 3
    stream = torch.cuda.current_stream(dev).cuda_stream
    tree = nb.Octree(nb.F64, 3, n, 0)
    tree.set_walk(form)
    tree.set_step_budget(budget)
    tree.enable_counters(True)
    tree.clear(stream); tree.compute_bounds(st, stream); tree.insert(st, stream); tree.compute_tree(stream)
    tree.compute_force(st, theta, stream)
    torch.cuda.synchronize()
    err = None
    try:
        tree.info(stream)
    except nb.NbodyError as ex:
        err = str(ex)
    out = big.cpu().numpy()
    assert np.all(out[:pad] == canary) and np.all(out[pad + n:] == canary), "the walk wrote outside a"
    cnt = tree.read_counters(stream)
    tree.close()
    return out[pad:pad + n].copy(), cnt, err


@pytest.mark.parametrize("form", [2, 1])   # the visit round as ISA, the compiler-scheduled kernel
def test_octree_walk_stack_full_and_step_budget_exits(nb, oracle, form):
    """the two emergency exits of the walk, which no test took.  (1) Stack full: a chain of 90 nested cells
    with two more cells to open at every level makes a theta = 0 walk hold > 155 pending cells ((2^D - 1) * 21 + 2^D slots per
    body); the kernel must leave, raise kFlagStack (reported by nbody_octree_info) and write nothing outside a — in the ISA form
    the overflowing push goes past the end of the block's LDS and is dropped by the hardware.  The same chain 60 levels deep
    (120 pending) fits: forces within tolerance of the oracle, counters bit-exact.  (2) Step budget: with a budget of 5 visit
    rounds (nbody_octree_set_step_budget) on an ordinary tree, bodies are abandoned and kFlagWalk is reported; with the default
    budget the same tree is clean."""
    a, cnt, err = _walk_with_canaries(nb, _deep_chain(nb, 90), 0.0, form)
    assert err is not None and "per-body stack" in err, err
    assert np.all(np.isfinite(a))
    hs = _deep_chain(nb, 60)
    a, cnt, err = _walk_with_canaries(nb, hs, 0.0, form)
    assert err is None, err
    ref = oracle.State(oracle.F64, 3, hs.n)
    ref.m[:], ref.x[:], ref.c, ref.dt = hs.m, hs.x, hs.c, hs.dt
    ocnt, osize, omass = oracle.octree_step_force(ref, 0.0, want_counts=True)
    assert np.array_equal(cnt, ocnt)
    err_a = np.abs(a - ref.a).max(axis=1)
    assert np.all(err_a <= 1e-12 * np.maximum(np.abs(ref.a).max(axis=1), 1.0))
    gal = nb.build_model(nb.F64, 3, "galaxy", 20000)
    a5, _, err = _walk_with_canaries(nb, gal, 0.5, form, budget=5)
    assert err is not None and "visit rounds" in err and "nbody_octree_set_step_budget" in err, err
    assert np.all(np.isfinite(a5))
    afull, _, err = _walk_with_canaries(nb, gal, 0.5, form)
    assert err is None and not np.array_equal(a5, afull)


@pytest.mark.parametrize("dtype,dim", [(1, 3), (0, 3), (1, 2), (0, 2)])
def test_octree_build_forms_are_bitwise_equal(nb, dtype, dim):
    """The two shipped ways to build the tree from the sorted keys: breadth-first with one launch per level (1), and in one pass
    from the common key prefixes of neighbouring bodies (3 = auto, 0), which numbers the sibling groups in pre-order instead of
    breadth-first.  Same tree size, same root monopole, same per-body counters and accelerations, bit for bit — also with cells
    below the key depth, with a 60-level chain of nested cells (in the one-pass build one position starts 20 cells), for two and
    three bodies, and for sizes around the 1024-position blocks of the prefix sum.  (The grid-barrier forms 2 and 4 live in the
    experiments build: tests/test_gpu_bvh.py::test_experiment_forms_are_bitwise_equal.)"""
    cases = [("galaxy", 100000), ("uniform", 30011), ("galaxy", 2), ("uniform", 1), ("uniform", 3), ("uniform", 1023), ("uniform", 1024),
             ("uniform", 1025), ("galaxy", 2049), ("galaxy", 300000)]   # 300 000: the multipole pass takes its second round
    if dim == 3:
        cases.append(("plummer", 5000))
    for wl, n in cases:
        res = []
        for form in (1, 3, 0):
            hs = nb.build_model(dtype, dim, wl, n)
            if wl == "uniform" and n > 1000:  # pairs far below the key resolution and an escaper that inflates the root cube
                hs.x[1] = hs.x[0] + (1e-9 if dtype == 1 else 1e-6)
                hs.x[5] = 4e3
            dev = nb.DeviceSystem.from_host(hs)
            dev.octree.set_build(form)
            dev.octree.enable_counters(True)
            for _ in range(2):   # twice: nothing is cleared between builds
                dev.octree_force(0.5)
                dev.sync()
                size, mass = dev.octree.info(dev.stream)
            res.append((size, mass, dev.octree.read_counters(dev.stream).copy(), dev.download().a.copy()))
            dev.close()
        for r in res[1:]:
            assert res[0][0] == r[0] and res[0][1] == r[1], (wl, n)
            assert np.array_equal(res[0][2], r[2]) and np.array_equal(res[0][3], r[3]), (wl, n)
    if dtype == 1 and dim == 3:
        outs = []
        for form in (1, 3, 0):
            # a shallow tree first (a galaxy of as many bodies), then the chain in the same buffers
            shallow = nb.build_model(1, 3, "galaxy", _deep_chain(nb, 60).n)
            dev = nb.DeviceSystem.from_host(shallow)
            dev.octree.set_build(form)
            dev.octree_force(0.5)
            dev.octree.info(dev.stream)
            dev.upload(_deep_chain(nb, 60))
            dev.octree_force(0.0)
            dev.sync()
            outs.append((dev.octree.info(dev.stream), dev.download().a.copy()))
            dev.close()
        for o in outs[1:]:
            assert outs[0][0] == o[0] and np.array_equal(outs[0][1], o[1])
