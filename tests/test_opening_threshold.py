"""K9's opening test as one compare.  The reference accepts a node iff  bw * bw < theta^2 * dist2  (src/bvh.h:246-248); the
sweep kernels test  v < dist2  against a threshold v stored in the node's record.  These tests hold the two forms equal, decision
by decision, with the reference's expression evaluated by the oracle (oracle.can_approximate):

* CPU (no GPU needed): the threshold function itself — the library exports its host instance — over widths and angles that
  cover normal, denormal, zero, huge and non-finite values.  The product is monotone in dist2, so a threshold is right for EVERY
  distance iff the reference rejects at v and accepts at the next number above v; that pair is checked for every input, together
  with a cloud of distances around v and random ones.
* GPU: the thresholds the build kernels wrote into a tree (nbody_bvh_read what = 6) are the host function's, the pair test holds
  for the tree's own widths, and a traversal with another angle rewrites them."""
import numpy as np
import pytest

UINT = {0: np.uint32, 1: np.uint64}
FLT = {0: np.float32, 1: np.float64}


def _succ(v, dtype, k=1):
    """k numbers above v >= 0 (bit pattern order)."""
    return (v.view(UINT[dtype]) + UINT[dtype](k)).view(FLT[dtype])


def _widths(rng, dtype, n):
    t, info = FLT[dtype], np.finfo(FLT[dtype])
    w = np.concatenate([
        10.0 ** rng.uniform(-6, 4, n),                                  # what trees hold
        10.0 ** rng.uniform(np.log10(float(info.tiny)) / 2 - 3, np.log10(float(info.tiny)) / 2 + 3, n // 4),  # squares around the denormal border
        10.0 ** rng.uniform(np.log10(float(info.max)) / 2 - 3, np.log10(float(info.max)) / 2 - 0.01, n // 4),  # squares near overflow
        [0.0, float(info.tiny), float(info.smallest_subnormal), 1.0, 0.5, 2.0 ** -30, 3.0, 1e-3],
    ]).astype(t)
    return w


def _check_pairs(oracle, nb, dtype, bw, theta):
    """For every width: the reference rejects at d2 = v and accepts just above it (monotone => equal for every d2), plus a cloud."""
    t = FLT[dtype]
    with np.errstate(over="ignore", under="ignore"):
        w2 = (bw * bw).astype(t)  # rounded once in T, as the record's width^2 (src/bvh.h:247 evaluates bw * bw)
    v = nb.bvh_opening_thresholds(dtype, w2, theta)
    finite = np.isfinite(v)
    assert (v[finite] >= 0).all()
    # no distance accepts: the reference must reject even at the largest finite distance and at +inf
    if (~finite).any():
        big = np.full((~finite).sum(), np.finfo(t).max, t)
        assert not oracle.can_approximate(dtype, bw[~finite], theta, big).any()
        assert not oracle.can_approximate(dtype, bw[~finite], theta, np.full_like(big, np.inf)).any()
    bwf, vf = bw[finite], v[finite]
    assert not oracle.can_approximate(dtype, bwf, theta, vf).any(), "the reference accepts at the threshold itself"
    assert oracle.can_approximate(dtype, bwf, theta, _succ(vf, dtype)).all(), "the reference rejects just above the threshold"
    rng = np.random.default_rng(7)
    for k in range(-6, 7):  # a cloud of neighbours, and random distances over the whole range
        bits = vf.view(UINT[dtype]).astype(np.int64 if dtype == 0 else np.uint64)
        if k < 0:
            ok = bits >= -k
            d2 = np.where(ok, bits - np.where(ok, -k, 0).astype(bits.dtype), 0).astype(UINT[dtype]).view(t)
        else:
            d2 = (bits + bits.dtype.type(k)).astype(UINT[dtype]).view(t)
        d2 = np.where(np.isnan(d2), np.inf, d2).astype(t)
        assert np.array_equal(oracle.can_approximate(dtype, bwf, theta, d2), vf < d2), k
    for _ in range(4):
        with np.errstate(over="ignore", under="ignore"):
            d2 = (vf.astype(np.float64) * 10.0 ** rng.uniform(-3, 3, vf.size)).astype(t)
        assert np.array_equal(oracle.can_approximate(dtype, bwf, theta, d2), vf < d2)


@pytest.mark.parametrize("dtype", [1, 0])
def test_threshold_equals_the_reference_opening_test(nb, oracle, dtype):
    rng = np.random.default_rng(20241004 + dtype)
    bw = _widths(rng, dtype, 20000)
    thetas = [0.5, 0.0, 1.0, 0.3, 0.7, 1.4, 2.5, 1e-3, 1e-160 if dtype else 1e-20, 1e150 if dtype else 1e18, 0.1, 1 / 3]
    thetas += list(10.0 ** rng.uniform(-2, 1, 6))
    for theta in thetas:
        _check_pairs(oracle, nb, dtype, bw, float(theta))


@pytest.mark.parametrize("dtype", [1, 0])
def test_threshold_special_values(nb, dtype):
    t = FLT[dtype]
    w2 = np.array([-1.0, np.nan, np.inf, 0.0], t)
    v = nb.bvh_opening_thresholds(dtype, w2, 0.5)
    assert v[0] == -1.0           # body entries: always accepted
    assert np.isnan(v[1])         # a NaN width keeps accepting, as !(NaN >= x) did
    assert np.isposinf(v[2])      # nothing is farther than an infinite node is wide
    assert v[3] >= 0 and np.isfinite(v[3])
    assert np.isposinf(nb.bvh_opening_thresholds(dtype, np.array([0.0, 1.0], t), 0.0)).all()  # theta = 0 opens everything


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,dim", [(1, 3), (0, 3), (1, 2), (0, 2)])
def test_tree_records_hold_the_thresholds(nb, oracle, dtype, dim):
    """What the build kernels wrote is what the host function computes from the tree's widths — for the default angle, for the
    angle of the last traversal, and after a traversal with another angle (rewritten in place)."""
    for wl, n in (("galaxy", 5000), ("uniform", 1000)):
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, wl, n))
        st, t = dev.state(), dev.bvh
        t.bounding_box(st, dev.stream); t.hilbert_sort(st, dev.stream); t.build_tree(st, dev.stream)
        bw = t.read(3, dev.stream)
        with np.errstate(over="ignore", under="ignore"):
            w2 = (bw * bw).astype(FLT[dtype])

        def same(theta):
            got, want = t.read(6, dev.stream), nb.bvh_opening_thresholds(dtype, w2, theta)
            return np.array_equal(got.view(UINT[dtype]), want.view(UINT[dtype]))
        assert same(0.5), "built for the reference's default angle"
        t.compute_force(st, 0.8, dev.stream); dev.sync()
        assert same(0.8), "rewritten by a traversal with another angle"
        t.build_tree(st, dev.stream); dev.sync()
        assert same(0.8), "the next build writes them for the last angle"
        _check_pairs(oracle, nb, dtype, bw, 0.8)
        dev.close()
