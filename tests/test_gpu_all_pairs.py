"""GPU parity: K1 all-pairs, K2 all-pairs-collapsed, K3 leapfrog through the C ABI vs the oracle and the
reference-generated fixtures.

Tolerances (stated once, used everywhere below):
  * integer / copy / K3 leapfrog: bit-exact.
  * double forces: |a_gpu - a_ref| <= 1e-12 * max|a_ref|  (the kernel's pair term is within 4.1e-15; the rest is
    summation order: the reference sums j ascending in one chain, the kernel in 4 wave-partials).
    Two legitimate builds of the reference (-O2 vs -Ofast) differ by 3.6e-15 after one step.
  * double trajectories vs reference frames: rel 1e-11 of the position scale after <= 20 steps.
  * float forces (one pass): 2e-5 * max|a_ref|.  Float trajectories and forces after several steps: the reference is
    not reproducible across its own builds in float, so the tolerance is MEASURED on it —
    tests/golden/calibrate_float_tolerance.py builds the reference -O2 and -Ofast, runs config 1 and smaller cases and
    records their spread; tests/golden/float_tolerance.json holds 4x that spread, imported below.
"""
import json
import os
import numpy as np
import pytest

from conftest import DT

pytestmark = pytest.mark.gpu

FLOAT_TOL = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "float_tolerance.json")))
FORCE_TOL = {0: 2e-5, 1: 1e-12}
TRAJ_TOL = {0: FLOAT_TOL["float_trajectory_rel_small_n"], 1: 1e-11}


def maxrel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("dim", [3, 2])
def test_all_pairs_force_vs_oracle(nb, oracle, dtype, dim):
    for wl, n in (("uniform", 1), ("uniform", 2), ("uniform", 63), ("uniform", 64), ("uniform", 65), ("galaxy", 511),
                  ("uniform", 512), ("galaxy", 513), ("uniform", 1025), ("galaxy", 4099)):
        ref = oracle.build_model(dtype, dim, wl, n)
        hs = nb.build_model(dtype, dim, wl, n)
        dev = nb.DeviceSystem.from_host(hs)
        oracle.all_pairs_force(ref)
        for split, tpt in ((0, 0), (1, 1), (1, 2), (2, 1), (2, 2), (4, 1), (4, 2)):
            nb.configure_all_pairs(split, tpt)
            dev.all_pairs_force()
            out = dev.download()
            assert maxrel(out.a, ref.a) <= FORCE_TOL[dtype], (wl, n, split, tpt)
        nb.configure_all_pairs(0, 0)
        dev.close()


def test_pair_term_accuracy(nb, oracle):
    """split=1 sums in the reference's order, so what is left is the pair math.  K1's far form (r2 >= 2^-16, third-order
    step on the cube of the v_rsq seed, no reciprocal) carries rounding error only: <= 2.5 ulp per term (csrc/common.hpp);
    the sum over 4098 terms in the same order then agrees with the oracle to ~1e-15 with no bias."""
    ref = oracle.build_model(1, 3, "galaxy", 4099)
    dev = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", 4099))
    nb.configure_all_pairs(1, 1)
    dev.all_pairs_force()
    nb.configure_all_pairs(0, 0)
    oracle.all_pairs_force(ref)
    a = dev.download().a
    assert maxrel(a, ref.a) <= 2e-15
    na, nr = np.linalg.norm(a, axis=1), np.linalg.norm(ref.a, axis=1)
    assert abs(np.mean((na - nr) / nr)) <= 2e-16  # no systematic bias


def test_pair_term_two_body_ulps(nb, oracle):
    """One source, one target: the kernel's pair term against the oracle's pow/divide expression, over separations that
    straddle the far/near switch of K1 (r2 = 2^-16, r = 2^-8) and the range where the reference's `+ eps` matters."""
    rng = np.random.default_rng(23)
    worst = 0.0
    for _ in range(6):
        n = 2048
        hs = nb.HostSystem(1, 3, n)
        # pairs (2k, 2k+1) at separation r_k, pairs far apart from each other so that one term dominates each sum
        r = np.concatenate([2.0 ** rng.uniform(-14, 6, n // 2 - 64), 2.0 ** -8 * (1 + rng.uniform(-1e-6, 1e-6, 64))])
        base = np.arange(n // 2)[:, None] * np.array([[1e9, 0, 0]])
        dirs = rng.standard_normal((n // 2, 3))
        dirs /= np.linalg.norm(dirs, axis=1)[:, None]
        hs.x[0::2] = base
        hs.x[1::2] = base + dirs * r[:, None]
        hs.m[:] = 10.0 ** rng.uniform(-3, 3, n)
        hs.c, hs.dt = 1.0, 0.1
        ref = oracle.State(1, 3, n)
        ref.m[:], ref.x[:], ref.c = hs.m, hs.x, 1.0
        dev = nb.DeviceSystem.from_host(hs)
        nb.configure_all_pairs(1, 1)
        dev.all_pairs_force()
        nb.configure_all_pairs(0, 0)
        oracle.all_pairs_force(ref)
        a = dev.download().a
        mag = np.abs(ref.a).max(axis=1)
        worst = max(worst, (np.abs(a - ref.a).max(axis=1) / mag).max())
        dev.close()
    assert worst <= 1.2e-15, worst  # a few ulp (2.2e-16 each, kernel + oracle): rounding only, on both sides of the switch


# 70001, 100000: the 8-slice scalar-stream form with source chunks (sz >= 65536), windows of every size class
@pytest.mark.parametrize("n", [6000, 70001, 100000])
def test_shard_windows_are_bitwise_identical(nb, n):
    """Multi-GPU property on one GPU: any split of the targets into shard windows gives bitwise the full result."""
    dev = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    dev.all_pairs_force()
    full = dev.download().a.copy()
    for parts in (2, 4, 8, 7) if n < 10000 else (2, 8, 70):
        edges = [n * r // parts for r in range(parts + 1)]
        hs0 = nb.build_model(1, 3, "galaxy", n)
        d2 = nb.DeviceSystem.from_host(hs0)
        for f, e in zip(edges[:-1], edges[1:]):
            d2.all_pairs_force(f, e - f)
        assert np.array_equal(d2.download().a, full), parts
        d2.close()


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("dim", [3, 2])
def test_source_paths_are_bitwise_identical(nb, dtype, dim):
    """K1's two ways of bringing a source record to the lanes (LDS tiles / scalar stream into SGPRs) perform the same
    arithmetic in the same order, for every (split, targets-per-lane) configuration and for shard windows."""
    n = 5000 + 37 * dim
    try:
        for split, tpt in ((1, 1), (2, 2), (4, 1), (4, 2), (1, 2)):  # explicit splits: one source chunk in both forms
            res = []
            for path in (1, 2):
                nb.configure_all_pairs(split, tpt, source_path=path)
                dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, "galaxy", n))
                dev.all_pairs_force()
                full = dev.download().a.copy()
                dev.all_pairs_force(1000, 2049)  # a shard window: rows [1000, 3049) are recomputed in place
                assert np.array_equal(dev.download().a, full), (split, tpt, path)
                res.append(full)
                dev.close()
            assert np.array_equal(res[0], res[1]), (split, tpt)
        # the automatic launch (8 slices, source chunks over grid.y): shard windows and both targets-per-lane settings agree
        nb.configure_all_pairs(0, 0, source_path=0)
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, "galaxy", n))
        assert "chunks=10" in nb.describe_all_pairs(dev.state()) and "JS=8" in nb.describe_all_pairs(dev.state())
        dev.all_pairs_force()
        full = dev.download().a.copy()
        dev.all_pairs_force(1000, 2049)
        assert np.array_equal(dev.download().a, full)
        for tpt in (1, 2):
            nb.configure_all_pairs(0, tpt)
            dev.all_pairs_force()
            assert np.array_equal(dev.download().a, full), tpt
        dev.close()
    finally:
        nb.configure_all_pairs(0, 0, source_path=0)


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("dim", [3, 2])
def test_accelerate_step_bit_exact(nb, oracle, dtype, dim):
    rng = np.random.default_rng(5)
    for n in (1, 7, 1000, 100003):
        hs = nb.build_model(dtype, dim, "uniform", n)
        hs.a[:] = rng.standard_normal(hs.a.shape)
        hs.ao[:] = rng.standard_normal(hs.a.shape)
        ref = oracle.State(dtype, dim, n)
        for k in ("m", "x", "v", "a", "ao"):
            getattr(ref, k)[:] = getattr(hs, k)
        ref.dt, ref.c = hs.dt, hs.c
        dev = nb.DeviceSystem.from_host(hs)
        dev.accelerate_step()
        out = dev.download()
        oracle.accelerate_step(ref)
        for k in ("x", "v", "a", "ao"):
            assert np.array_equal(getattr(out, k), getattr(ref, k)), (n, k)
        dev.close()


def test_trajectories_vs_reference_fixtures(nb, golden_positions):
    """positions.bin frames written by the real reference (--save pos --csv-detailed) for all-pairs cases."""
    meta, data = golden_positions
    ran = 0
    for name, case in meta.items():
        if case["algorithm"] != "all-pairs":
            continue
        dtype = DT[case["precision"]]
        ref = data[name + "__frames"]
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, case["dim"], case["workload"], case["n"]))
        scale = np.abs(ref[0]).max()
        k = 1
        for step in range(1, case["steps"] + 1):
            nb.run(dev, "all-pairs", 1)
            if step in case["frame_ids"]:
                x = dev.download().x
                assert np.abs(x - ref[k]).max() <= TRAJ_TOL[dtype] * scale, (name, step)
                k += 1
        dev.close()
        ran += 1
    assert ran >= 25


def test_print_state_text_double_exact(nb, oracle, golden_print_state):
    """north_star: --print-state output matches the reference exactly (double, small N): all-pairs cases,
    default mode => max(steps, 10) steps."""
    ran = 0
    for name, case in golden_print_state.items():
        if case["algorithm"] != "all-pairs" or case["precision"] != "double":
            continue
        hs = nb.build_model(1, case["dim"], case["workload"], case["n"])
        dev = nb.DeviceSystem.from_host(hs)
        nb.run(dev, "all-pairs", nb.executed_steps(case["steps"], False))
        out = dev.download()
        s = oracle.State(1, case["dim"], hs.n)
        for k in ("m", "x", "v", "a", "ao"):
            getattr(s, k)[:] = getattr(out, k)
        rows = oracle.format_state_rows(s)
        diff = [i for i, (r, g) in enumerate(zip(rows, case["final"])) if r != g]
        assert not diff, f"{name}: {len(diff)} rows differ, first: {rows[diff[0]]} vs {case['final'][diff[0]]}"
        dev.close()
        ran += 1
    assert ran >= 12


def test_print_state_text_double_exact_n4096(nb, oracle, golden_print_state_n4096):
    """double `--print-state` identity "for N <= 4096, theta = 0 and all-pairs" — the reference's own printed
    final rows at n = 4096 (tests/golden/generate_golden_n4096.py), all-pairs and bvh theta = 0, row order included."""
    for name, case in golden_print_state_n4096.items():
        hs = nb.build_model(1, case["dim"], case["workload"], case["n"])
        dev = nb.DeviceSystem.from_host(hs)
        nb.run(dev, case["algorithm"], nb.executed_steps(case["steps"], False), case["theta"] if case["theta"] is not None else 0.5)
        out = dev.download()
        s = oracle.State(1, case["dim"], hs.n)
        for k in ("m", "x", "v", "a", "ao"):
            getattr(s, k)[:] = getattr(out, k)
        rows = oracle.format_state_rows(s)
        diff = [i for i, (r, g) in enumerate(zip(rows, case["final"])) if r != g]
        assert len(rows) == len(case["final"]) and not diff, f"{name}: {len(diff)} rows differ, first: {rows[diff[0]]} vs {case['final'][diff[0]]}"
        dev.close()


def test_config1_2d_float_n10000(nb, oracle):
    """BASELINE config[0]: all-pairs 2D float -n 10000 -s 5 (=> 10 steps).  Close encounters amplify the last float bit, so
    after 10 steps neither positions nor forces are comparable body by body at rounding level — not even between the
    reference's own -O2 and -Ofast builds.  The assertions use that measured spread (x4): positions, the worst force and
    the 99th percentile of the per-body force deviation; plus one fresh force pass at the one-pass tolerance."""
    ref = oracle.build_model(0, 2, "uniform", 10000)
    dev = nb.DeviceSystem.from_host(nb.build_model(0, 2, "uniform", 10000))
    nsteps = nb.executed_steps(5, False)
    nb.run(dev, "all-pairs", nsteps)
    oracle.run(ref, "all-pairs", nsteps)
    out = dev.download()
    assert np.abs(out.x - ref.x).max() <= FLOAT_TOL["config1_trajectory_rel"] * np.abs(ref.x).max()
    ferr = np.abs(out.a.astype(np.float64) - ref.a).max(axis=1) / np.abs(ref.a).max()
    assert ferr.max() <= FLOAT_TOL["config1_force_after_10_steps_max_rel"], ferr.max()
    assert np.quantile(ferr, 0.99) <= FLOAT_TOL["config1_force_after_10_steps_p99_rel"], np.quantile(ferr, 0.99)
    ref1 = oracle.build_model(0, 2, "uniform", 10000)
    dev1 = nb.DeviceSystem.from_host(nb.build_model(0, 2, "uniform", 10000))
    dev1.all_pairs_force()
    oracle.all_pairs_force(ref1)
    assert maxrel(dev1.download().a, ref1.a) <= FORCE_TOL[0]


def _sample_check(nb, oracle, dtype, dim, wl, n, algo, nsample=192, tol=None):
    """Full-size run on the GPU; a random sample of targets is recomputed by the oracle against all n sources."""
    hs = nb.build_model(dtype, dim, wl, n)
    dev = nb.DeviceSystem.from_host(hs)
    getattr(dev, algo)()
    out = dev.download()
    rng = np.random.default_rng(11)
    ref = oracle.State(dtype, dim, hs.n)
    for k in ("m", "x", "v"):
        getattr(ref, k)[:] = getattr(hs, k)
    ref.c, ref.dt = hs.c, hs.dt
    picks = np.unique(np.concatenate([rng.integers(0, hs.n, nsample), [0, hs.n // 2, hs.n - 1]]))
    scale = 0.0
    worst = 0.0
    for i in picks:
        oracle.all_pairs_force(ref, int(i), 1)
    scale = np.abs(ref.a[picks]).max()
    worst = np.abs(out.a[picks] - ref.a[picks]).max() / scale
    assert worst <= (tol or FORCE_TOL[dtype]), worst
    return hs, dev, out


def test_config2_full_size_properties(nb, oracle):
    """BASELINE config[1]: 3D double N=65536.  Oracle on a target sample + size-independent properties:
    total momentum change sum(m_i a_i) = 0 (Newton's third law) and exact linearity in the masses."""
    hs, dev, out = _sample_check(nb, oracle, 1, 3, "galaxy", 65536, "all_pairs_force")
    mom = (hs.m[:, None] * out.a).sum(axis=0)
    assert np.abs(mom).max() <= 1e-10 * np.abs(hs.m[:, None] * out.a).sum()
    hs2 = nb.build_model(1, 3, "galaxy", 65536)
    hs2.m *= 2.0  # power of two: every product scales exactly
    d2 = nb.DeviceSystem.from_host(hs2)
    d2.all_pairs_force()
    assert np.array_equal(d2.download().a, 2.0 * out.a)


def test_config5_n_2pow20_sample(nb, oracle):
    """BASELINE metric size: 3D double N=2^20 on one GPU, oracle on a target sample."""
    _sample_check(nb, oracle, 1, 3, "galaxy", 1 << 20, "all_pairs_force", nsample=64)


def test_config5_rank_windows_at_2pow20(nb, oracle):
    """BASELINE config[4] as each of its 8 ranks launches it: a 131 072-target window with first = k * 131072 at
    sz = 2^20.  Bitwise the rows of the single-GPU run (so any GPU count gives one trajectory), and within 1e-12 of the
    oracle on a sample of the window's targets against all 2^20 sources."""
    n, w = 1 << 20, 1 << 17
    hs = nb.build_model(1, 3, "galaxy", n)
    dev = nb.DeviceSystem.from_host(hs)
    dev.all_pairs_force()
    full = dev.download().a.copy()
    dev.close()
    ref = oracle.State(1, 3, n)
    ref.m[:], ref.x[:], ref.c = hs.m, hs.x, hs.c
    rng = np.random.default_rng(5)
    for world, k in ((2, 1), (4, 2)):   # the windows of a 2- and a 4-GPU run: bitwise the single-GPU rows as well
        f, e = nb.shard_range(n, k, world)
        d2 = nb.DeviceSystem.from_host(hs)
        d2.all_pairs_force(f, e - f)
        assert np.array_equal(d2.download().a[f:e], full[f:e]), (world, k)
        d2.close()
    for k in (0, 3, 7):
        assert nb.shard_range(n, k, 8) == (k * w, (k + 1) * w)
        d2 = nb.DeviceSystem.from_host(hs)
        st = d2.state(k * w, w)
        assert "R=2,JS=8" in nb.describe_all_pairs(st)
        d2.all_pairs_force(k * w, w)
        got = d2.download().a
        assert np.array_equal(got[k * w:(k + 1) * w], full[k * w:(k + 1) * w]), k
        assert not got[:k * w].any() and not got[(k + 1) * w:].any()   # nothing outside the window was written
        picks = np.unique(np.concatenate([rng.integers(k * w, (k + 1) * w, 24), [k * w, (k + 1) * w - 1]]))
        for i in picks:
            oracle.all_pairs_force(ref, int(i), 1)
        scale = np.abs(ref.a[picks]).max()
        assert np.abs(got[picks] - ref.a[picks]).max() <= FORCE_TOL[1] * scale, k
        d2.close()


def test_per_context_tuning_and_describe(nb):
    """The K1 launch shape is a property of the context (nbody_state.tuning), not of the process: two contexts with
    different settings coexist, each equals the process-wide setting of the same values, and describe() names the launch."""
    n = 5000
    res = {}
    for split, tpt, path in ((1, 1, 1), (4, 2, 2)):
        nb.configure_all_pairs(split, tpt, source_path=path)
        d = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
        d.all_pairs_force()
        res[(split, tpt, path)] = d.download().a.copy()
        d.close()
    nb.configure_all_pairs(0, 0, source_path=0)
    da = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    db = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    da.configure_all_pairs(1, 1, 1)
    db.configure_all_pairs(4, 2, 2)
    assert da.state().tuning == nb.tuning(1, 1, 1) and db.state().tuning == nb.tuning(4, 2, 2)
    assert nb.describe_all_pairs(da.state()).startswith("all_pairs_force_kernel<double,3,R=1,JS=1>")
    assert nb.describe_all_pairs(db.state()).startswith("all_pairs_force_sgpr_kernel<double,3,R=2,JS=4>")
    st_a, st_b = da.state(), db.state()
    lib = nb.lib()
    import ctypes as C
    assert lib.nbody_all_pairs_force(C.byref(st_b), C.c_void_p(db.stream)) == 0
    assert lib.nbody_all_pairs_force(C.byref(st_a), C.c_void_p(da.stream)) == 0
    assert np.array_equal(da.download().a, res[(1, 1, 1)]) and np.array_equal(db.download().a, res[(4, 2, 2)])
    big = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", 1 << 17))
    assert "chunks=" in nb.describe_all_pairs(big.state()) and "JS=8" in nb.describe_all_pairs(big.state())


def test_edge_cases(nb):
    hs = nb.HostSystem(1, 3, 4)
    hs.m[:] = [1, 2, 3, 0]
    hs.x[:] = [[0, 0, 0], [0, 0, 0], [1, 0, 0], [5, 5, 5]]  # coincident pair, zero-mass body
    hs.c, hs.dt = 1.0, 0.1
    dev = nb.DeviceSystem.from_host(hs)
    dev.all_pairs_force()
    a = dev.download().a
    assert np.all(np.isfinite(a))
    expect = 3.0 / (1.0 + np.finfo(np.float64).eps)
    assert abs(a[0, 0] - expect) <= 4e-16 * expect and a[0, 0] == a[1, 0] and a[0, 1] == 0 and a[0, 2] == 0
    one = nb.HostSystem(0, 2, 1)
    one.m[:] = 1
    one.c = 1.0
    d1 = nb.DeviceSystem.from_host(one)
    d1.all_pairs_force()
    d1.all_pairs_collapsed_force()
    assert not d1.download().a.any()


# ---- K2 -------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("dim", [3, 2])
def test_collapsed_force_vs_oracle(nb, oracle, dtype, dim):
    """Intended semantics (all D components, 64-bit pair space) == all-pairs up to rounding; a - ao reset included."""
    for wl, n in (("uniform", 2), ("uniform", 65), ("galaxy", 1000), ("uniform", 4099)):
        ref = oracle.build_model(dtype, dim, wl, n)
        hs = nb.build_model(dtype, dim, wl, n)
        rng = np.random.default_rng(3)
        hs.a[:] = rng.standard_normal(hs.a.shape)
        hs.ao[:] = hs.a  # steady state of the driver: ao == a after accelerate_step
        dev = nb.DeviceSystem.from_host(hs)
        dev.all_pairs_collapsed_force()
        oracle.all_pairs_force(ref)
        assert maxrel(dev.download().a, ref.a) <= 4 * FORCE_TOL[dtype], (wl, n)
        dev.close()


@pytest.mark.parametrize("dim", [3, 2])
def test_collapsed_streamed_form_at_its_boundaries(nb, oracle, dim):
    """Float takes K2's streamed form (round 6: a wave owns 16 targets for its whole source chunk, the packed records come in
    batches of 512 (3D) / 256 (2D) per wave, two batches per trip, the record array padded with zero-mass records to whole pairs
    of batches): sizes on both sides of every boundary of that shape — one body, a group of 16, a wave's 64 lanes, one and two
    batches, the padding — with a coincident pair, a pair closer than the float epsilon's cube root and the `a - ao` reset,
    against the oracle's all-pairs sums.  3D runs the hand-scheduled pair, 2D the compiled one."""
    for n in (1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2049, 5000, 16385):
        hs = nb.build_model(0, dim, "uniform", n)
        if n >= 17:
            hs.x[5] = hs.x[11]                                  # coincident: contributes exactly 0
            hs.x[7] = hs.x[3] + np.float32(1e-4)                 # r^3 ~ 1e-12 << eps: the `+ eps` dominates the denominator
        ref = oracle.State(0, dim, n)
        for k in ("m", "x", "v"):
            getattr(ref, k)[:] = getattr(hs, k)
        ref.dt, ref.c = hs.dt, hs.c
        rng = np.random.default_rng(n)
        hs.a[:] = rng.standard_normal(hs.a.shape)
        hs.ao[:] = hs.a
        dev = nb.DeviceSystem.from_host(hs)
        dev.all_pairs_collapsed_force()
        got = dev.download().a
        oracle.all_pairs_force(ref)
        assert np.isfinite(got).all(), n
        scale = max(np.abs(ref.a).max(), 1e-30)
        assert np.abs(got - ref.a).max() <= 4 * FORCE_TOL[0] * scale, (n, np.abs(got - ref.a).max() / scale)
        dev.close()


def test_collapsed_vs_reference_as_written_2d(nb, golden_positions):
    """D=2, N < 65536: the reference's own collapsed output is well defined; match its frames."""
    meta, data = golden_positions
    ran = 0
    for name, case in meta.items():
        if case["algorithm"] != "all-pairs-collapsed" or case["dim"] != 2:
            continue
        dtype = DT[case["precision"]]
        ref = data[name + "__frames"]
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 2, case["workload"], case["n"]))
        for step in range(1, case["steps"] + 1):
            nb.run(dev, "all-pairs-collapsed", 1)
            assert np.abs(dev.download().x - ref[step]).max() <= TRAJ_TOL[dtype] * np.abs(ref[0]).max(), (name, step)
        ran += 1
    assert ran >= 2


def test_config3_collapsed_3d_float_262144(nb, oracle):
    """BASELINE config[2]: the reference computes nothing here (pair count wraps to 0); parity is pinned to
    all-pairs on a target sample.  Tolerance: the one-pass float tolerance plus 4x the run-to-run spread of the
    kernel's float atomics, MEASURED on the product (tests/golden/calibrate_config3_atomics.py: 0 over three runs at this size —
    four partial sums per target — and 5.3e-7 between K2 and K1)."""
    tol = FORCE_TOL[0] + 4.0 * FLOAT_TOL["config3_collapsed"]["run_to_run_spread_max_rel"]
    hs, dev, out = _sample_check(nb, oracle, 0, 3, "uniform", 262144, "all_pairs_collapsed_force", nsample=64, tol=tol)
    dev.all_pairs_collapsed_force()   # K2 accumulates: (a - ao) + sum with ao = 0 ADDS the same forces once more ...
    hs2 = dev.download()
    assert maxrel(hs2.a - out.a, out.a) <= 1e-6   # ... so the increment of the second pass reproduces the first within the spread


@pytest.mark.parametrize("dtype", [1, 0])
def test_pair_math_adversarial_separations(nb, oracle, dtype):
    """Separations from far below eps^(1/3) (where the reference's `+ eps` dominates the denominator) up to 1e6, exact
    coincidences, zero masses and 12 decades of mass ratio: per-target error against the oracle stays at rounding level."""
    rng = np.random.default_rng(17)
    t = np.float64 if dtype == 1 else np.float32
    n = 3000
    hs = nb.HostSystem(dtype, 3, n)
    centers = rng.standard_normal((30, 3)) * 100.0
    scale = 10.0 ** rng.uniform(-13 if dtype == 1 else -6, 3, size=n)          # cluster radii over many decades
    x = centers[rng.integers(0, 30, n)] + rng.standard_normal((n, 3)) * scale[:, None]
    x[100:110] = x[90:100]                                                      # exact coincidences
    x[200] = 1e6
    hs.x[:] = x.astype(t)
    hs.m[:] = (10.0 ** rng.uniform(-6, 6, n)).astype(t)
    hs.m[300:310] = 0
    hs.c, hs.dt = 1.0, 0.1
    ref = oracle.State(dtype, 3, n)
    ref.m[:], ref.x[:], ref.c = hs.m, hs.x, 1.0
    dev = nb.DeviceSystem.from_host(hs)
    oracle.all_pairs_force(ref)
    for split in (1, 4):
        nb.configure_all_pairs(split, 1)
        dev.all_pairs_force()
        a = dev.download().a.astype(np.float64)
        assert np.all(np.isfinite(a))
        # per target: error relative to the sum of the magnitudes of its terms ~ |a| scale of the dominant term
        err = np.abs(a - ref.a).max(axis=1)
        mag = np.abs(ref.a).max(axis=1) + 1e-300
        tol = 2e-13 if dtype == 1 else 1e-4
        bad = np.where(err > tol * np.maximum(mag, np.median(mag)))[0]
        assert bad.size == 0, (split, bad[:5], err[bad[:5]], mag[bad[:5]])
    nb.configure_all_pairs(0, 0)


def test_torch_plumbing_path_equals_ctx_path(nb):
    """bench.py's path (torch tensors' data_ptr() and torch's current stream passed through the C ABI, including the
    shard-window form with v/a/ao holding only the owned rows) gives bitwise the result of the owning-context path."""
    import torch
    n, steps = 6000, 3
    dev = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
    nb.run(dev, "all-pairs", steps)
    ref = dev.download()
    hs = nb.build_model(1, 3, "galaxy", n)
    sim = nb.parallel.ShardedAllPairs(hs, 0, 1, torch_device=torch.device("cuda", 0))
    for _ in range(steps):
        sim.step()
    torch.cuda.synchronize()
    x, v, a = sim.gather_state()
    assert np.array_equal(x, ref.x) and np.array_equal(v, ref.v) and np.array_equal(a, ref.a)
    # two "ranks" emulated one after the other on one GPU: each owns half of the targets, positions merged by hand
    sims = [nb.parallel.ShardedAllPairs(nb.build_model(1, 3, "galaxy", n), r, 2, torch_device=torch.device("cuda", 0)) for r in range(2)]
    for s in sims:
        s.exchange = False  # no process group here; the exchange is emulated below
    for _ in range(steps):
        for s in sims:
            s.force_phase()
            s.integrate_phase()
        torch.cuda.synchronize()
        for s in sims:      # what all_gather_into_tensor does
            for o in sims:
                s.x[o.first:o.first + o.count] = o.x[o.first:o.first + o.count]
    torch.cuda.synchronize()
    assert np.array_equal(sims[0].x.cpu().numpy(), ref.x) and np.array_equal(sims[1].x.cpu().numpy(), ref.x)
    assert np.array_equal(np.concatenate([s.v.cpu().numpy() for s in sims]), ref.v)


def _planted_system(nb, dtype, dim, extreme):
    """6000 well-separated bodies (spacing ~0.5) plus partners planted in a DIFFERENT source tile than their target (index
    distance >= 2100: beyond K2's 2048-record tile, K1's and the energies' 512): pairs closer than 2^-8 over many decades;
    with `extreme` also exact coincidences, a zero-mass partner and an r^2 that is denormal in T (target at the origin)."""
    rng = np.random.default_rng(23 + dtype * 2 + dim)
    t = np.float64 if dtype == 1 else np.float32
    n = 6000
    hs = nb.HostSystem(dtype, dim, n)
    side = 9.0 if dim == 3 else 40.0
    hs.x[:] = rng.uniform(-side, side, (n, dim)).astype(t)
    hs.m[:] = rng.uniform(0.5, 2.0, n).astype(t)
    hs.v[:] = rng.standard_normal((n, dim)).astype(t)
    gaps = [3e-3, 2.0 ** -9, 1e-4, 1e-6] + ([1e-9, 1e-12] if dtype == 1 else [])
    planted = []
    for k, gap in enumerate(gaps):
        i, j = 17 + 31 * k, 17 + 31 * k + 2100 + 613 * k
        d = np.zeros(dim)
        d[k % dim] = gap
        hs.x[j] = (hs.x[i].astype(np.float64) + d).astype(t)
        planted += [i, j]
    if extreme:
        hs.x[300] = hs.x[2900]                          # coincident, distinct bodies, tiles apart
        hs.x[301] = hs.x[5000]
        hs.x[400] = 0                                    # target at the origin ...
        hs.x[3000] = 0
        hs.x[3000, 0] = 1e-160 if dtype == 1 else 1e-22  # ... partner at r^2 = 1e-320 / 1e-44: denormal in T
        hs.x[500] = hs.x[4000]
        hs.x[500, dim - 1] += t(1e-3)
        hs.m[500] = 0                                    # zero-mass partner of a near pair
        planted += [300, 2900, 301, 5000, 400, 3000, 500, 4000]
    hs.c, hs.dt = 1.0, 0.01
    return hs, np.array(planted)


def _as_oracle_state(oracle, hs):
    ref = oracle.State(hs.dtype, hs.dim, hs.n)
    ref.m[:], ref.x[:], ref.v[:], ref.c, ref.dt = hs.m, hs.x, hs.v, hs.c, hs.dt
    return ref


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("dim", [3, 2])
@pytest.mark.parametrize("extreme", [False, True])
def test_near_pairs_in_another_source_tile(nb, oracle, dtype, dim, extreme):
    """K2's two-pass path (a (group, tile) block is first evaluated with the reciprocal-free weight and redone
    guarded only if it held a pair closer than 2^-8) and the energies' near path were reached only through the self tile.
    Here the near partner sits >= 2100 indices away from its target — another source tile for every kernel — at gaps from
    3e-3 down to 1e-12, plus (extreme) coincident bodies, a zero-mass partner and a denormal r^2.  K1, K2 and calc_energies
    against the oracle, per target relative to that target's own scale (the planted targets' forces are 10^2..10^18 times
    the typical force, so a global max-norm would hide everything else)."""
    hs, planted = _planted_system(nb, dtype, dim, extreme)
    ref = _as_oracle_state(oracle, hs)
    oracle.all_pairs_force(ref)
    ra = ref.a.astype(np.float64)
    mag = np.abs(ra).max(axis=1)
    floor = np.median(mag)
    tol = 2e-13 if dtype == 1 else 4e-5
    dev = nb.DeviceSystem.from_host(hs)
    for name, call in (("K1", dev.all_pairs_force), ("K2", dev.all_pairs_collapsed_force)):
        call()
        a = dev.download().a.astype(np.float64)
        assert np.all(np.isfinite(a)), name
        err = np.abs(a - ra).max(axis=1)
        bad = np.where(err > tol * np.maximum(mag, floor))[0]
        assert bad.size == 0, (name, bad[:6], err[bad[:6]], mag[bad[:6]])
        assert np.all(err[planted] <= tol * np.maximum(mag[planted], floor)), name
        hs0 = dev.download()          # K2 accumulates into a - ao: start it from the reference's steady state a == ao
        hs0.ao[:] = hs0.a
        dev.upload(hs0)
    # K2 once more from that state: (a - ao) + sum must give the same forces (the diagonal pass of src/all_pairs.h:35-40)
    dev.all_pairs_collapsed_force()
    a = dev.download().a.astype(np.float64)
    assert np.all(np.abs(a - ra).max(axis=1) <= 2 * tol * np.maximum(mag, floor))
    ke, pe = dev.calc_energies()
    oke, ope = oracle.calc_energies(ref)
    assert abs(ke - oke) <= (1e-12 if dtype == 1 else 2e-5) * abs(oke)
    if dtype == 1:
        assert np.isfinite(pe) and abs(pe - ope) <= 1e-12 * abs(ope), (pe, ope)
    else:
        # float: the oracle (like the reference's transform_reduce on a serial backend) adds 3.6e7 terms into ONE float
        # accumulator, which by itself drifts 3e-5 from the exact sum of the same terms (measured: -58563080 against
        # -58564799.4); the kernel's tree-shaped sum does not.  The yardstick is therefore the oracle's own terms summed in
        # double (oracle.calc_energies_wide, src/system.h:62-79), against which the oracle's float sum is held too.
        _, pe64 = oracle.calc_energies_wide(ref)
        assert np.isfinite(pe) and abs(pe - pe64) <= 2e-6 * abs(pe64), (pe, pe64)
        assert abs(ope - pe64) <= 1e-4 * abs(pe64), (ope, pe64)
    dev.close()


@pytest.mark.parametrize("dim", [3, 2])
def test_sparse_system_pair_rules_with_planted_pairs(nb, oracle, dim):
    """K1's launch-level far mode (double, >= 32 768 bodies whose bounding box says "sparse"): pairs at r^2 >= 4 drop the eps term
    of the weight, a batch that holds a closer pair selects per lane, r^2 < 2^-16 takes the guarded form.  40 000 bodies in a box
    of 2 * 10^6 (3D) / 6 * 10^6 (2D) units of volume, with partners planted tiles apart from their targets: r^2 EXACTLY 4 and one
    ulp to either side, gaps from 1.5 down to 1e-12, coincident bodies, a zero-mass partner, a denormal r^2.  Every target against
    the oracle relative to its own scale; then the same launch over shard windows, bit for bit (the rule a pair takes depends on its
    own r^2 only, never on its batch)."""
    rng = np.random.default_rng(77 + dim)
    n = 40000
    hs = nb.HostSystem(1, dim, n)
    side = 63.0 if dim == 3 else 1250.0
    hs.x[:] = rng.uniform(-side, side, (n, dim))
    hs.m[:] = rng.uniform(0.5, 2.0, n)
    gaps = [2.0, np.nextafter(2.0, 3.0), np.nextafter(2.0, 1.0), 1.5, 0.3, 2.0 ** -8, np.nextafter(2.0 ** -8, 0.0), 3e-3, 1e-6, 1e-12]
    planted = []
    for k, gap in enumerate(gaps):
        i, j = 101 + 977 * k, 101 + 977 * k + 4099 + 1033 * k   # another source tile, another slice, usually another chunk
        d = np.zeros(dim)
        d[k % dim] = gap
        hs.x[i] = np.round(hs.x[i])          # exact coordinates: the planted r^2 is what it says
        hs.x[j] = hs.x[i] + d
        planted += [i, j]
    hs.x[300] = hs.x[29000]                  # coincident, distinct bodies
    hs.x[400] = 0
    hs.x[31000] = 0
    hs.x[31000, 0] = 1e-160                  # r^2 = 1e-320: denormal
    hs.x[500] = hs.x[24000]
    hs.x[500, dim - 1] += 1e-3
    hs.m[500] = 0                            # zero-mass partner of a near pair
    planted += [300, 29000, 400, 31000, 500, 24000]
    planted = np.array(planted)
    hs.c, hs.dt = 1.0, 0.01
    dev = nb.DeviceSystem.from_host(hs)
    sparse, vol = nb.all_pairs_pair_rule(dev.state(), dev.stream)   # the rule in force is a property of the system's extent
    assert sparse and vol >= (1.7e5 if dim == 3 else 6.4e4), (sparse, vol)
    assert nb.all_pairs_pair_rule(dev.state(100, 5000), dev.stream) == (sparse, vol)   # ... the same for every shard window
    hd = nb.build_model(nb.F64, dim, "uniform", n)                                     # the unit cube: the dense rule
    dense = nb.DeviceSystem.from_host(hd)
    assert nb.all_pairs_pair_rule(dense.state(), dense.stream)[0] is False
    dense.close()
    # ... and the cube with a few escapers far outside (what config 2 as written becomes: close encounters eject bodies at ten
    # times the bulk's speed): the bounding box would say "sparse" (rounds 3-4: + 8.5 % on config 2), the variances do not
    hd.x[rng.integers(0, n, 40)] *= 400.0
    dense = nb.DeviceSystem.from_host(hd)
    sp, v = nb.all_pairs_pair_rule(dense.state(), dense.stream)
    assert sp is False and np.prod(np.ptp(hd.x, axis=0)) > (1.7e5 if dim == 3 else 6.4e4) > v, (sp, v, np.ptp(hd.x, axis=0))
    side_eff = np.sqrt(12.0 * hd.x.astype(np.float64).var(axis=0))
    assert abs(v - np.prod(side_eff)) <= 1e-9 * v, (v, side_eff)                       # the documented quantity: prod sqrt(12 var_k)
    again = nb.all_pairs_pair_rule(dense.state(7, 1000), dense.stream)
    assert again == (sp, v)                                                            # the same BITS from another window, another launch
    dense.close()
    dev.all_pairs_force()
    dev.sync()
    a = dev.download().a.astype(np.float64)
    assert np.all(np.isfinite(a))
    ref = _as_oracle_state(oracle, hs)
    oracle.all_pairs_force(ref)
    ra = ref.a.astype(np.float64)
    mag = np.abs(ra).max(axis=1)
    floor = np.median(mag)
    err = np.abs(a - ra).max(axis=1)
    bad = np.where(err > 2e-13 * np.maximum(mag, floor))[0]
    assert bad.size == 0, (bad[:6], err[bad[:6]], mag[bad[:6]])
    assert np.all(err[planted] <= 2e-13 * np.maximum(mag[planted], floor))
    for first, count in ((0, n // 8), (n // 8, n // 8), (n - 5000, 5000), (12345, 7777)):
        dev.all_pairs_force(first, count)
        dev.sync()
        w = dev.download().a[first:first + count]
        assert np.array_equal(w, a[first:first + count].astype(w.dtype)), (first, count)
    dev.close()


def test_config2_as_written_100_steps(nb, oracle):
    """BASELINE config[1] as written: `-n 65536 -s 100 --precision double --algorithm all-pairs` — no workload flag, so the
    reference's default, uniform (src/arguments.h:27) — 100 steps of K1 + K3 (src/all_pairs.h:86-97).  The oracle cannot run
    4.3e11 pairs in a test, so the trajectory is pinned link by link: at steps 1, 37 and 100 the state before the step
    (x, v, a, ao as the device holds them) is handed to the oracle, which performs that one step for a sample of targets against
    all 65536 sources; positions, velocities and accelerations after the step must agree to 1e-11 / 1e-12.  Momentum drift over
    the run bounds the rest.  tests/test_gpu_cli.py runs the same command through the CLI and compares its printed rows."""
    n, steps = 65536, 100
    hs = nb.build_model(nb.F64, 3, "uniform", n)
    dev = nb.DeviceSystem.from_host(hs)
    rng = np.random.default_rng(2)
    picks = np.unique(np.concatenate([rng.integers(0, n, 96), [0, n - 1]]))
    p0 = (hs.m[:, None] * hs.v).sum(axis=0)
    for step in range(1, steps + 1):
        before = dev.download() if step in (1, 37, 100) else None
        nb.run(dev, "all-pairs", 1)
        if before is None:
            continue
        after = dev.download()
        ref = _as_oracle_state(oracle, before)
        ref.a[:], ref.ao[:] = before.a, before.ao
        for i in picks:
            oracle.all_pairs_force(ref, int(i), 1)
        assert np.abs(after.a[picks] - ref.a[picks]).max() <= 1e-12 * np.abs(ref.a[picks]).max(), step
        ref.a[picks] = after.a[picks]        # then the leapfrog of exactly those rows, bit for bit from the same inputs
        sub = oracle.State(oracle.F64, 3, len(picks))
        sub.x[:], sub.v[:], sub.a[:], sub.ao[:], sub.dt = before.x[picks], before.v[picks], after.a[picks], before.ao[picks], before.dt
        oracle.accelerate_step(sub)
        assert np.array_equal(sub.x, after.x[picks]) and np.array_equal(sub.v, after.v[picks]) and np.array_equal(sub.ao, after.ao[picks]), step
    out = dev.download()
    p1 = (out.m[:, None] * out.v).sum(axis=0)
    assert np.abs(p1 - p0).max() <= 1e-11 * np.abs(out.m[:, None] * out.v).sum()
    dev.close()


def test_k1_handoff_status_of_ordinary_runs(nb):
    """K1's chunks add their sums into `a` in turn (all_pairs.hip, all_pairs_force_sgpr_kernel; launches of up to 2048 blocks collect
    them instead: same bits, no waiting).  The stream's status block must say
    after ordinary launches — whole systems and rank windows, both precisions — that no hand-off failed; how many waves had to
    poll for their turn is reported, not asserted (normally none: the predecessor finished a round of blocks earlier)."""
    for dtype, n, first, count in ((1, 8192, 0, None), (0, 8192, 0, None), (1, 70001, 0, None), (1, 1 << 18, 1 << 17, 1 << 15)):
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
        desc = nb.describe_all_pairs(dev.state(first, count))
        assert ("by the last to arrive" if n == 8192 else "summed in turn") in desc, desc
        for _ in range(3):
            dev.all_pairs_force(first, count)
        dev.sync()                                   # would raise NBODY_ERR_STATE
        st = nb.all_pairs_status(dev.stream)
        assert not st["failed"] and st["rc"] == 0, st
        a = dev.download().a
        lo, hi = first, first + (count if count is not None else n)
        assert np.isfinite(a[lo:hi]).all()
        dev.close()
    # a stream that never ran the chunked kernel reports zeros
    dev = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", 600))
    dev.all_pairs_force()
    assert nb.all_pairs_status(dev.stream, clear=True) == {"failed": False, "block": 0, "group": 0, "chunk": 0, "polls": 0, "waits": 0, "rc": 0}
    dev.close()


@pytest.mark.parametrize("dtype", [1, 0])
def test_k1_handoff_made_to_wait_and_made_to_fail(dtype):
    """The hand-off under stress, through libnbody_hip_exp.so (the same kernel; only that build reads the two switches) in a child
    process.  (1) NBODY_K1_HANDOFF_DELAY: every block sleeps before it passes the turn on, so its successors really wait: polls > 0
    and `a` bitwise equal to the undelayed run — for the whole system, for a rank window and with two targets per lane.
    (2) A turn that does not come within the budget (delay of milliseconds, NBODY_K1_TURN_SPINS = 3): nbody_stream_sync and
    nbody_download return NBODY_ERR_STATE naming block and chunk, the status block holds the failure, and EVERY row of `a` is either
    NaN or bitwise the right value — never a finite partial sum — both when the holder of the turn is late in passing it on and
    (round 6) when a predecessor has not begun to poll at all while its successors give up.  (3) The flag is sticky until cleared; after the clear the same
    context computes the right forces again."""
    import subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "stdpar-nbody_amd", "libnbody_hip_exp.so")
    if not os.path.exists(lib):
        pytest.skip("libnbody_hip_exp.so not built (make -C stdpar-nbody_amd experiments)")
    code = textwrap.dedent(f"""
        import os, sys, numpy as np
        sys.path.insert(0, {os.path.join(root, 'tests')!r})
        from conftest import load_package
        nb = load_package()
        nb.LIB_PATH = {lib!r}
        dtype = {dtype}
        def force(n, first=0, count=None, tpt=0, delay=None, spins=None, expect_failure=False, collect=None):
            # launches of up to 2048 blocks collect their chunks' sums instead of passing turns (no waiting there at all): the
            # hand-off is tested with NBODY_K1_COLLECT=0, which makes every size pass turns
            for k, v in (("NBODY_K1_HANDOFF_DELAY", delay), ("NBODY_K1_TURN_SPINS", spins), ("NBODY_K1_COLLECT", collect)):
                os.environ.pop(k, None)
                if v is not None:
                    os.environ[k] = str(v)
            dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
            if tpt:
                dev.configure_all_pairs(0, tpt, 0)
            dev.all_pairs_force(first, count)
            if not expect_failure:
                dev.sync()
                st = nb.all_pairs_status(dev.stream)
                a = dev.download().a.copy()
                dev.close()
                return a, st
            return dev
        for n, first, count, tpt in ((8192, 0, None, 0), (8192, 0, None, 2), (50000, 20000, 9000, 0), (3000, 0, None, 0), (21000, 0, None, 0)):
            a0, st0 = force(n, first, count, tpt, collect=0)
            a1, st1 = force(n, first, count, tpt, delay=50, collect=0)
            a2, st2 = force(n, first, count, tpt)            # as shipped: the small ones collect, the large one passes turns
            assert not st0["failed"] and not st1["failed"] and not st2["failed"], (st0, st1, st2)
            assert st1["polls"] > 0 and st1["waits"] > 0, ("nobody waited", n, st1)
            assert np.array_equal(a0, a1), ("delayed hand-off changed a", n, first, count, tpt)
            assert np.array_equal(a0, a2), ("collected sums differ from sums passed in turn", n, first, count, tpt)
            print("waited", n, first, count, tpt, st1["waits"], st1["polls"])
        # the turn that never comes in time
        n = 8192
        a0, _ = force(n)
        dev = force(n, delay=3000, spins=3, expect_failure=True, collect=0)
        try:
            dev.sync()
            raise SystemExit("nbody_stream_sync returned success after a failed hand-off")
        except nb.NbodyError as e:
            assert "error 3" in str(e) and "hand-off" in str(e) and "chunk" in str(e), str(e)
        st = nb.all_pairs_status(dev.stream, check=False)
        assert st["failed"] and st["rc"] == 3 and 1 <= st["chunk"] <= 15 and st["block"] < n // 64, st
        hs = nb.HostSystem(dtype, 3, n)
        try:
            dev.download(hs)
            raise SystemExit("nbody_download returned success after a failed hand-off")
        except nb.NbodyError as e:
            assert "hand-off" in str(e)
        bad = np.isnan(hs.a).any(axis=1)
        assert bad.any(), "a failed hand-off left no NaN"
        assert np.array_equal(hs.a[~bad], a0[~bad]), "a finite row differs from the undisturbed run"
        assert np.isnan(hs.a[bad]).all(), "a row is partly NaN"
        print("failed as it should:", int(bad.sum()), "of", n, "rows NaN;", st)
        dev.close()
        # the case the poison exists for (round 6): a predecessor that has not even begun to poll when its successors give up.  Chunk
        # 1's waves sleep before their FIRST look at the turn word (NBODY_K1_HANDOFF_LATE); chunk 0 has handed on at once, so the
        # word holds 1 while chunks 2..15 run out of polls and swap the poison in over it; chunk 1 then finds the poison at its first
        # poll.  Until round 6 both kinds of wave left `a` alone and the rows kept the finite partial sum s_0: now each writes NaN.
        for k in ("NBODY_K1_HANDOFF_DELAY", "NBODY_K1_TURN_SPINS", "NBODY_K1_COLLECT"):
            os.environ.pop(k, None)
        os.environ.update(NBODY_K1_HANDOFF_LATE="3000", NBODY_K1_TURN_SPINS="3", NBODY_K1_COLLECT="0")
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
        dev.all_pairs_force()
        st = nb.all_pairs_status(dev.stream, check=False)
        assert st["failed"] and st["rc"] == 3 and st["chunk"] >= 2, st
        hs = nb.HostSystem(dtype, 3, n)
        try:
            dev.download(hs)
            raise SystemExit("nbody_download returned success after a failed hand-off (late predecessor)")
        except nb.NbodyError:
            pass
        bad = np.isnan(hs.a).any(axis=1)
        assert bad.sum() >= n // 2, ("a late predecessor poisoned almost nothing", int(bad.sum()))
        assert np.array_equal(hs.a[~bad], a0[~bad]), "late predecessor: a finite row differs from the undisturbed run"
        assert np.isnan(hs.a[bad]).all(), "late predecessor: a row is partly NaN"
        print("late predecessor:", int(bad.sum()), "of", n, "rows NaN;", st)
        os.environ.pop("NBODY_K1_HANDOFF_LATE")
        os.environ.update(NBODY_K1_HANDOFF_DELAY="3000", NBODY_K1_TURN_SPINS="3", NBODY_K1_COLLECT="0")
        nb.all_pairs_status(dev.stream, clear=True, check=False)
        dev.all_pairs_force()   # the first failure again, on this context: the sticky flag and the recovery are checked on it below
        # sticky, then cleared; the context works again
        os.environ.pop("NBODY_K1_HANDOFF_DELAY"); os.environ.pop("NBODY_K1_TURN_SPINS"); os.environ.pop("NBODY_K1_COLLECT")
        try:
            dev.sync()
            raise SystemExit("the failure flag is not sticky")
        except nb.NbodyError:
            pass
        assert nb.all_pairs_status(dev.stream, clear=True, check=False)["failed"]
        dev.sync()
        dev.all_pairs_force()
        dev.sync()
        assert np.array_equal(dev.download().a, a0)
        assert not nb.all_pairs_status(dev.stream)["failed"]
        dev.close()
        print("ok")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
