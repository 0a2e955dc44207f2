"""CPU: the product's ISO-C++ workload generators (stdpar-nbody_amd/host/models.hpp via libnbody_host.so)
against the oracle and the reference-generated fixtures, bit-exact, plus CLI plumbing that needs no GPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import DT, ROOT


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("prec", ["float", "double"])
def test_models_bit_exact_vs_oracle(nb, oracle, dim, prec):
    if not os.path.exists(nb.HOST_LIB_PATH):
        nb.build()
    for wl in ("uniform", "galaxy") + (("plummer",) if dim == 3 else ()):
        for n in (2, 10, 11, 1000, 4097):
            h = nb.build_model(DT[prec], dim, wl, n)
            o = oracle.build_model(DT[prec], dim, wl, n)
            assert h.n == o.n and h.dt == o.dt and h.c == o.c
            assert np.array_equal(h.m, o.m) and np.array_equal(h.x, o.x) and np.array_equal(h.v, o.v), (wl, n)


def test_models_vs_reference_frame0(nb, golden_positions):
    meta, data = golden_positions
    for name, case in meta.items():
        h = nb.build_model(DT[case["precision"]], case["dim"], case["workload"], case["n"])
        assert np.array_equal(h.x, data[name + "__frames"][0]), name


def test_plummer_needs_3d(nb):
    with pytest.raises(nb.NbodyError):
        nb.build_model(nb.F64, 2, "plummer", 8)


def test_one_body_galaxy_is_refused(nb, oracle):
    """models.h:112-136 places one central mass per disc — index 1 of a one-body system is past its end (the reference writes there).
    The generator and the oracle refuse instead; two bodies are the two central masses."""
    with pytest.raises(nb.NbodyError):
        nb.build_model(nb.F64, 3, "galaxy", 1)
    with pytest.raises(RuntimeError):
        oracle.build_model(1, 3, "galaxy", 1)
    h, o = nb.build_model(nb.F64, 3, "galaxy", 2), oracle.build_model(1, 3, "galaxy", 2)
    assert h.m.tolist() == [1e4, 1e3] and np.array_equal(h.x, o.x)


def _cli(dim):
    p = os.path.join(ROOT, "stdpar-nbody_amd", "bin", f"nbody_hip_d{dim}")
    if not os.path.exists(p):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "stdpar-nbody_amd"), "all"])
    return p


def test_cli_flag_errors_match_reference_text():
    """Reference behaviour [SURVEY §5]: `--bogus` -> stdout "Unknown argument: '--bogus'", exit 1; both csv
    flags -> stderr message, exit 1; bad enum values list the options; --help exits 0."""
    exe = _cli(3)
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout == "Unknown argument: '--bogus'\n"
    r = subprocess.run([exe, "--csv-total", "--csv-detailed"], capture_output=True, text=True)
    assert r.returncode == 1 and "Cannot capture a CSV detailed and coarse trace in the same run" in r.stderr
    r = subprocess.run([exe, "--precision", "half"], capture_output=True, text=True)
    assert r.returncode == 1 and 'Unknown precision: "half".' in r.stderr and "Options are: double, float (default)." in r.stderr
    r = subprocess.run([exe, "--algorithm", "fmm"], capture_output=True, text=True)
    assert r.returncode == 1 and 'Unknown algorithm: "fmm".' in r.stderr
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("Help:\n-n size\t\tNumber of particles to simulate\n")


def test_cli_help_matches_reference_binary(oracle):
    if oracle.ref_binary(3) is None:
        pytest.skip("oracle/_ref not built here")
    ours = subprocess.run([_cli(3), "--help"], capture_output=True, text=True).stdout
    assert ours == oracle.ref_run(3, ["--help"])
