"""The one-pass octree build (csrc/octree.hip: ot_lcp_kernel / ot_build_lcp_kernel) rests on one counting argument: with the
bodies sorted by path key and l_i = the number of leading key digits bodies i-1 and i share (l_0 = l_n = -1), sorted position i
starts exactly the cells of levels l_i + 1 ... l_(i+1), so sum max(0, l_(i+1) - l_i) is the number of cells, an exclusive prefix
sum of it numbers them in pre-order, and a cell's subtree is a rank interval.  This file restates that argument in numpy (no
GPU) and checks it against the ORACLE's octree — the reference's insertion algorithm restated in C (src/octree.h:114-174) —
through the one number both expose: the tree size (next_free_child_group).  The kernels themselves are compared with the
breadth-first build, bit for bit, in tests/test_gpu_octree.py::test_octree_build_forms_are_bitwise_equal."""
import numpy as np
import pytest

import oracle as O
from conftest import ROOT  # noqa: F401  (sys.path set-up)


def path_keys(x, dim, levels):
    """ot_keys_kernel: the reference's `pos > divide` chain (src/octree.h:127-138) from the bounds of src/octree.h:93-112."""
    t = x.dtype.type
    mn, mx = min(t(0), x.min()), max(t(0), x.max())
    mx, mn = t(mx + t(1)), t(mn - t(1))
    divide = np.full(x.shape, t((mx + mn) / t(2)), x.dtype)
    side = t(mx - mn)
    keys = np.zeros(len(x), np.uint64)
    for _ in range(levels):
        half = t(side / t(4))
        gt = x > divide
        cp = np.zeros(len(x), np.uint64)
        for k in range(dim):
            cp |= gt[:, k].astype(np.uint64) << np.uint64(k)
        divide = (divide + np.where(gt, half, -half).astype(x.dtype)).astype(x.dtype)
        side = t(side / t(2))
        keys = (keys << np.uint64(dim)) | cp
    return keys


def common_levels(a, b, dim, levels):
    out = np.full(len(a), levels, np.int64)
    for lvl in range(levels):          # first level (from the root) at which the digits differ
        shift = np.uint64(dim * (levels - 1 - lvl))
        differ = ((a >> shift) != (b >> shift)) & (out == levels)
        out[differ] = lvl
    return out


def cells_by_prefix_counting(keys, dim, levels):
    """(number of cells, per-level counts) from the sorted keys, by the boundary argument."""
    k = np.sort(keys)
    n = len(k)
    l = np.full(n + 1, -1, np.int64)
    if n > 1:
        l[1:n] = common_levels(k[:-1], k[1:], dim, levels)
    starts = np.maximum(0, l[1:] - l[:-1])           # cells that start at each sorted position
    per_level = np.zeros(levels + 1, np.int64)
    for i in np.nonzero(starts)[0]:
        per_level[l[i] + 1:l[i + 1] + 1] += 1
    # the same count the direct way: a cell of level d = a d-digit prefix shared by >= 2 bodies
    direct = 0
    for d in range(levels + 1):
        shift = np.uint64(dim * (levels - d))
        pref = (k >> shift) if d > 0 and int(shift) < 64 else np.zeros(n, np.uint64)
        _, cnt = np.unique(pref, return_counts=True)
        direct += int((cnt >= 2).sum())
    assert int(starts.sum()) == direct == int(per_level.sum())
    # pre-order: the rank of cell (i, d) is P[i] + d - l_i - 1 and its subtree ends at P[end of its range]
    P = np.concatenate([[0], np.cumsum(starts)])
    for i in np.nonzero(starts)[0][:200]:
        for d in range(l[i] + 1, l[i + 1] + 1):
            shift = np.uint64(dim * (levels - d))
            inside = (k >> shift) == (k[i] >> shift) if d > 0 and int(shift) < 64 else np.ones(n, bool)
            e = i + int(inside[i:].argmin()) if not inside[i:].all() else n
            rank = P[i] + d - l[i] - 1
            assert P[i] <= rank < P[e]               # the cell and everything below it: ranks [rank, P[e])
            if d > 0:
                assert l[i] < d <= l[i + 1]
    return int(starts.sum()), per_level


@pytest.mark.parametrize("dim,dtype", [(3, np.float64), (3, np.float32), (2, np.float64)])
@pytest.mark.parametrize("n,kind", [(2, "uniform"), (3, "uniform"), (257, "uniform"), (5000, "uniform"), (4000, "clustered")])
def test_prefix_counting_gives_the_oracles_tree_size(dim, dtype, n, kind):
    rng = np.random.default_rng(n * 7 + dim)
    x = rng.uniform(-1, 1, (n, dim)).astype(dtype)
    if kind == "clustered":   # a dense core inside a sparse halo: deep chains of nested cells
        x[: n // 2] = (x[: n // 2] * dtype(1e-3)).astype(dtype)
        x[0] = dtype(30.0)
    levels = 21 if dim == 3 else 32
    keys = path_keys(x, dim, levels)
    assert len(np.unique(keys)) == n, "the case is meant to stay above the key depth"
    cells, per_level = cells_by_prefix_counting(keys, dim, levels)
    s = O.State(0 if dtype == np.float32 else 1, dim, n)
    s.c = 1.0
    s.x[:] = x
    s.m[:] = dtype(1.0 / n)
    _, size, _ = O.octree_step_force(s, 0.5)
    assert size == 1 + cells * (1 << dim), (size, cells, per_level)
