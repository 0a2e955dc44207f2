"""CPU: the oracle (plain-C restatement) against the golden fixtures generated from the REAL reference
(tests/golden/generate_golden.py).  Bit-exact, every case."""
import numpy as np
import pytest

from conftest import DT


def _run_oracle(O, case, frames):
    s = O.build_model(DT[case["precision"]], case["dim"], case["workload"], case["n"])
    init = s.x.copy()
    mode = 0 if case["algorithm"] == "all-pairs-collapsed" else 1  # fixture = reference as written (defects incl.)
    O.run(s, case["algorithm"], case["steps"], theta=case["theta"] if case["theta"] is not None else 0.5,
          collapsed_mode=mode, frames=frames)
    return init, s


def test_positions_bit_exact(oracle, golden_positions):
    meta, data = golden_positions
    assert len(meta) >= 70
    for name, case in meta.items():
        ref = data[name + "__frames"]
        frames = []
        init, _ = _run_oracle(oracle, case, frames)
        allf = [init] + frames
        for k, fid in enumerate(case["frame_ids"]):
            assert np.array_equal(ref[k], allf[fid]), f"{name}: frame {fid} differs from the reference"


def test_energy_bit_exact(oracle, golden_positions):
    meta, data = golden_positions
    checked = 0
    for name, case in meta.items():
        key = name + "__energy"
        if key not in data.files:
            continue
        s = oracle.build_model(DT[case["precision"]], case["dim"], case["workload"], case["n"])
        mode = 0 if case["algorithm"] == "all-pairs-collapsed" else 1
        en = [oracle.calc_energies(s)]
        for _ in range(case["steps"]):
            oracle.run(s, case["algorithm"], 1, theta=case["theta"] if case["theta"] is not None else 0.5, collapsed_mode=mode)
            en.append(oracle.calc_energies(s))
        assert np.array_equal(np.array(en, dtype=data[key].dtype), data[key]), name
        checked += 1
    assert checked >= 20


def test_print_state_text_exact(oracle, golden_print_state):
    """--print-state rows (4 significant digits) incl. the step-count semantics: default mode runs max(steps, 10)."""
    for name, case in golden_print_state.items():
        if case["n"] > 64:
            continue  # the n=1000 cases are exercised by the GPU parity tests; keep the CPU suite fast
        s = oracle.build_model(DT[case["precision"]], case["dim"], case["workload"], case["n"])
        assert oracle.format_state_rows(s) == case["start"], f"{name}: starting state"
        mode = 0 if case["algorithm"] == "all-pairs-collapsed" else 1
        oracle.run(s, case["algorithm"], oracle.executed_steps(case["steps"], False),
                   theta=case["theta"] if case["theta"] is not None else 0.5, collapsed_mode=mode)
        assert oracle.format_state_rows(s) == case["final"], f"{name}: final state"


def test_print_state_text_exact_n4096(oracle, golden_print_state_n4096):
    """The oracle prints the reference's rows at n = 4096 too (double, all-pairs and bvh theta = 0)."""
    import hashlib
    for name, case in golden_print_state_n4096.items():
        s = oracle.build_model(oracle.F64, case["dim"], case["workload"], case["n"])
        assert hashlib.md5("\n".join(oracle.format_state_rows(s)).encode()).hexdigest() == case["start_md5"], name
        oracle.run(s, case["algorithm"], max(case["steps"], 10), case["theta"] if case["theta"] is not None else 0.5)
        assert oracle.format_state_rows(s) == case["final"], name


def test_hilbert_known_answers(oracle, golden_hilbert):
    assert len(golden_hilbert) >= 400
    for dim, c0, c1, c2, h, il in golden_hilbert:
        cell = [c0, c1] if dim == 2 else [c0, c1, c2]
        assert oracle.hilbert_cell(dim, cell) == h
        assert oracle.interleave_bits(dim, cell) == il


def test_bvh_theta0_equals_all_pairs_as_multiset(oracle):
    """README.md:122-129: theta=0 BVH and all-pairs agree — as a multiset of rows (bvh permutes bodies)."""
    a = oracle.build_model(oracle.F64, 2, "uniform", 10)
    b = oracle.build_model(oracle.F64, 2, "uniform", 10)
    oracle.run(a, "all-pairs", 10)
    oracle.run(b, "bvh", 10, theta=0.0)
    strip = lambda rows: sorted(r.split(": ", 1)[1] for r in rows)
    assert strip(oracle.format_state_rows(a)) == strip(oracle.format_state_rows(b))


def test_collapsed_reference_defects(oracle):
    """as written, the collapsed variant drops component 2 and wraps the pair count at 2^32;
    the intended semantics (mode 1) equal all-pairs up to rounding."""
    s0 = oracle.build_model(oracle.F64, 3, "uniform", 6)
    s1 = s0.copy()
    s2 = s0.copy()
    oracle.all_pairs_collapsed_force(s0, mode=0)
    oracle.all_pairs_collapsed_force(s1, mode=1)
    oracle.all_pairs_force(s2)
    assert np.all(s0.a[:, 2] == 0) and np.any(s2.a[:, 2] != 0)
    np.testing.assert_allclose(s1.a, s2.a, rtol=1e-13)
    big = oracle.State(oracle.F32, 2, 65536)  # 65536^2 wraps to 0 pairs: no force at all
    big.m[:] = 1
    big.x[:] = np.random.default_rng(0).random((65536, 2), dtype=np.float32)
    big.c = 1.0
    oracle.all_pairs_collapsed_force(big, mode=0)
    assert not big.a.any()


def test_edge_cases(oracle):
    # coincident bodies and the self term contribute exactly zero
    s = oracle.State(oracle.F64, 3, 3)
    s.m[:] = [1, 2, 3]
    s.x[:] = [[0, 0, 0], [0, 0, 0], [1, 0, 0]]
    s.c = 1.0
    oracle.all_pairs_force(s)
    assert np.all(np.isfinite(s.a))
    assert s.a[0, 0] == s.a[1, 0] == 3.0 / (1.0 + np.finfo(np.float64).eps)
    # n = 2 (smallest tree), odd n (dead node), bbox always contains the origin
    for n in (2, 3, 5):
        t = oracle.build_model(oracle.F64, 3, "uniform", n)
        t.x += 10.0
        lo, hi = oracle.bounding_box(t)
        assert np.all(lo < 0) and np.all(hi > t.x.max(axis=0))  # origin is always inside the box
        u = t.copy()
        oracle.bvh_step_force(t, 0.0)
        oracle.all_pairs_force(u)
        order = np.lexsort(t.x.T[::-1])
        order_u = np.lexsort(u.x.T[::-1])
        np.testing.assert_allclose(t.a[order], u.a[order_u], rtol=1e-12, atol=1e-18)


def test_bvh_key_ties_match_the_reference_as_multisets(oracle, golden_bvh_ties):
    """Two bodies in one Hilbert cell: the reference's std::sort is unstable (src/bvh.h:55-94), so only the multiset of final
    rows is specified.  The oracle (stable order) reproduces the reference's rows — printed and full precision — as multisets,
    and at theta = 0 they are also the all-pairs rows."""
    from conftest import assert_frames_equal_as_multisets, rows_multiset, tie_case_arrays
    for name, case in golden_bvh_ties.items():
        assert case["bodies_with_tied_keys"] >= 6
        for algo, steps, detailed in (("bvh", 12, False), ("all-pairs", 12, False), ("bvh", 4, True)):
            s = oracle.State(oracle.F64, case["dim"], case["n"])
            s.m[:], s.x[:], s.v[:], s.dt, s.c = tie_case_arrays(case)
            if algo == "bvh" and not detailed:   # the ties are real in this restatement of the key function too
                lo, hi = oracle.bounding_box(s)
                keys = oracle.hilbert_keys(s, lo, hi)
                assert len(np.unique(keys)) <= case["n"] - 3
            oracle.run(s, algo, steps, theta=0.0)
            if detailed:
                assert_frames_equal_as_multisets(s.x, case["bvh_last_frame"], 1e-11)
            else:
                assert rows_multiset(oracle.format_state_rows(s)) == rows_multiset(case[f"final_rows_{algo}"]), (name, algo)
        assert rows_multiset(case["final_rows_bvh"]) == rows_multiset(case["final_rows_all-pairs"]), name


def test_long_runs_bit_exact(oracle, golden_long_runs):
    """SURVEY §8(c) at its stated length: 100 steps at N = 1024 (3D double; galaxy and uniform; all-pairs, bvh theta 0 and 0.5, octree
    theta 0.5), the galaxy in float (the reference's default precision) and in 2D double (all-pairs, bvh and octree at theta 0.5
    each), and one 1000-step
    all-pairs run that goes through the discs' collision (energy jumps by 5e5 at step 366): the oracle holds the reference's
    frames AND every (KE, PE) pair bit for bit (reference: src/all_pairs.h:72-83, src/saving.h:100-122, src/system.h:62-79)."""
    meta, data = golden_long_runs
    assert len(meta) == 15
    for name, case in meta.items():
        s = oracle.build_model(DT[case["precision"]], case["dim"], case["workload"], case["n"])
        ref, ref_en = data[name + "__frames"], data[name + "__energy"]
        keep = {fid: k for k, fid in enumerate(case["frame_ids"])}
        assert ref.dtype == s.x.dtype and np.array_equal(s.x, ref[keep[0]]), name
        en = [oracle.calc_energies(s)]
        for step in range(1, case["steps"] + 1):
            oracle.run(s, case["algorithm"], 1, theta=case["theta"] if case["theta"] is not None else 0.5)
            en.append(oracle.calc_energies(s))
            if step in keep:
                assert np.array_equal(s.x, ref[keep[step]]), f"{name}: frame {step}"
        assert np.array_equal(np.array(en, dtype=ref_en.dtype), ref_en), name
