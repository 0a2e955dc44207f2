"""pytest configuration: `gpu` marker, package loader (the package directory has a hyphen), fixtures."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_package():
    """Import stdpar-nbody_amd/ under the module name stdpar_nbody_amd."""
    if "stdpar_nbody_amd" in sys.modules:
        return sys.modules["stdpar_nbody_amd"]
    path = os.path.join(ROOT, "stdpar-nbody_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("stdpar_nbody_amd", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["stdpar_nbody_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def nb():
    return load_package()


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.lib()  # builds liboracle.so on first use if missing
    return O


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_positions():
    meta = json.load(open(os.path.join(GOLDEN, "positions_meta.json")))
    data = np.load(os.path.join(GOLDEN, "positions.npz"))
    return meta, data


@pytest.fixture(scope="session")
def golden_print_state():
    return json.load(open(os.path.join(GOLDEN, "print_state.json")))


@pytest.fixture(scope="session")
def golden_print_state_n4096():
    """The reference's final `--print-state` rows at n = 4096 (tests/golden/generate_golden_n4096.py), SURVEY §8(c)."""
    import gzip
    with gzip.open(os.path.join(GOLDEN, "print_state_n4096.json.gz"), "rt") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def float_tolerance():
    """Float tolerances measured on two builds of the reference (tests/golden/calibrate_float_tolerance.py)."""
    return json.load(open(os.path.join(GOLDEN, "float_tolerance.json")))


@pytest.fixture(scope="session")
def golden_hilbert():
    return json.load(open(os.path.join(GOLDEN, "hilbert.json")))


DT = {"float": 0, "double": 1}


@pytest.fixture(scope="session")
def golden_bvh_ties():
    """Systems with equal Hilbert keys and the reference's outputs for them (tests/golden/generate_golden_ties.py)."""
    return json.load(open(os.path.join(GOLDEN, "bvh_ties.json")))


def tie_case_arrays(case):
    """(m, x, v, dt, G) of a bvh_ties.json case in float64 (the reference reads float32 and converts, src/saving.h:25-68)."""
    body = np.array(case["body_f32"], dtype=np.float64)
    d = case["dim"]
    return body[:, 0].copy(), body[:, 1:1 + d].copy(), body[:, 1 + d:].copy(), case["dt"], case["G"]


def rows_multiset(rows):
    """`--print-state` rows without their index prefix, sorted: the order-free form (bvh permutes the bodies)."""
    return sorted(r.split(": ", 1)[1] for r in rows)


def assert_frames_equal_as_multisets(got, want, rtol):
    """Two (n, D) position frames hold the same bodies in any order: rows sorted lexicographically on coordinates rounded well
    above the tolerance, then compared within rtol * max|want|."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    scale = np.abs(want).max()
    key = lambda a: np.lexsort(np.round(a / (scale * 1e-9)).T[::-1])
    g, w = got[key(got)], want[key(want)]
    assert np.abs(g - w).max() <= rtol * scale, np.abs(g - w).max() / scale


@pytest.fixture(scope="session")
def golden_long_runs():
    """100- and 1000-step runs of the real reference at N = 1024, 3D double (tests/golden/generate_golden_long.py): SURVEY §8(c)'s
    criterion at its stated length, with the distance of the reference's own other builds beside every case."""
    meta = json.load(open(os.path.join(GOLDEN, "long_runs_meta.json")))
    data = np.load(os.path.join(GOLDEN, "long_runs.npz"))
    return meta, data
