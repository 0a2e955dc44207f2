"""GPU: the ISO-C++ CLI host (stdpar-nbody_amd/bin/nbody_hip_d{2,3}) end to end against the reference-generated
fixtures: --print-state text, positions.bin / energy.bin, CSV rows and the step-count semantics."""
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import DT, ROOT

pytestmark = pytest.mark.gpu


def cli(dim, args, cwd=None):
    exe = os.path.join(ROOT, "stdpar-nbody_amd", "bin", f"nbody_hip_d{dim}")
    return subprocess.run([exe] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True, timeout=600)


def test_print_state_text_matches_reference(oracle, golden_print_state):
    """Double precision: starting AND final rows are textually identical to the reference's stdout."""
    ran = 0
    for name, case in golden_print_state.items():
        if case["precision"] != "double" or case["algorithm"] == "all-pairs-collapsed":
            continue
        r = cli(case["dim"], case["args"])
        assert r.returncode == 0, r.stderr
        start, final = oracle.parse_print_state(r.stdout)
        assert start == case["start"], f"{name}: starting state"
        bad = [i for i, (a, b) in enumerate(zip(final, case["final"])) if a != b]
        assert len(final) == len(case["final"]) and not bad, f"{name}: {len(bad)} final rows differ"
        assert "Starting simulation\n" in r.stdout and re.search(r"Done simulation\nTotal time: \d+\.\d\d ms\n", r.stdout)
        ran += 1
    assert ran >= 36


def test_float_n10_text(oracle, golden_print_state):
    """Float agrees textually only at toy N: the README check, -s 5 -n 10, D=2."""
    for name, case in golden_print_state.items():
        if case["precision"] == "float" and case["n"] == 10 and case["dim"] == 2 and case["algorithm"] != "all-pairs-collapsed":
            r = cli(2, case["args"])
            start, final = oracle.parse_print_state(r.stdout)
            assert start == case["start"]
            num = lambda rows: np.array([[float(v) for v in re.findall(r"[-+]?\d\.\d+e[-+]\d+", row)] for row in rows])
            np.testing.assert_allclose(num(final), num(case["final"]), rtol=2e-3, atol=1e-9)


def test_step_count_semantics():
    """-s 1, 5, 10 all run 10 steps; -s 11 runs 11 (default mode warm-up, src/arguments.h:26, src/all_pairs.h:93-97)."""
    outs = {}
    for s in (1, 5, 10, 11):
        r = cli(3, ["-n", 64, "-s", s, "--precision", "double", "--algorithm", "all-pairs", "--workload", "galaxy", "--print-state"])
        outs[s] = r.stdout.split("Final state:")[1].split("Done simulation")[0]
    assert outs[1] == outs[5] == outs[10] and outs[10] != outs[11]


def test_csv_rows():
    r = cli(3, ["-n", 1000, "-s", 5, "--precision", "double", "--algorithm", "all-pairs", "--csv-total"])
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "algorithm,dim,precision,nsteps,nbodies,total [s]"
    # steps - warmup wraps as size_t, as printed by the reference
    assert re.fullmatch(r"all-pairs,3,64,18446744073709551611,1000,\d+\.\d\d", lines[1])
    r = cli(2, ["-n", 500, "-s", 3, "--algorithm", "all-pairs-collapsed", "--csv-detailed"])
    assert re.fullmatch(r"all-pairs-collapsed,2,32,3,500,\d+\.\d\d,\d+\.\d\d,\d+\.\d\d", r.stdout.strip())  # no header in detailed mode
    r = cli(3, ["-n", 500, "-s", 3, "--algorithm", "bvh", "--precision", "double", "--csv-detailed", "--print-info"])
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "algorithm,dim,precision,nsteps,nbodies,total [s],force [s],accel [s],bbox [s],sort [s],multipoles [s],force approx [s]"
    assert sum(1 for l in lines if l.startswith("Total mass:  1.00000")) == 3
    assert re.fullmatch(r"bvh,3,64,3,500(,\d+\.\d\d){7}", lines[-1])
    r = cli(3, ["-n", 100, "--csv-total", "--print-state", "--algorithm", "bvh"])
    assert r.returncode != 0  # abort(), as the reference (src/bvh.h:334-339)
    # octree is the DEFAULT algorithm (src/arguments.h:28): no --algorithm flag
    r = cli(3, ["-n", 100, "-s", 2, "--precision", "double", "--workload", "galaxy", "--print-info", "--csv-detailed"])
    lines = r.stdout.strip().splitlines()
    assert lines[0] == ("algorithm,dim,precision,nsteps,nbodies,total [s],force [s],accel [s],clear [s],bbox [s],insert [s],"
                        "multipoles [s],force approx [s]")
    # same text as the reference for this input (oracle/_ref: "Tree size: 697 / 681", "Total mass:  11002.00000")
    assert lines[1:6] == ["Tree init complete", "Tree size: 697", "Total mass:  11002.00000", "Tree size: 681", "Total mass:  11002.00000"]
    assert re.fullmatch(r"octree,3,64,2,100(,\d+\.\d\d){8}", lines[6])


def test_saved_frames_and_energy_vs_reference(oracle, golden_positions):
    """--save all --csv-detailed writes steps+1 frames and energy pairs in the reference's binary formats."""
    meta, data = golden_positions
    ran = 0
    for name, case in meta.items():
        if name + "__energy" not in data.files or case["algorithm"] == "all-pairs-collapsed":
            continue
        dbl = case["precision"] == "double"
        with tempfile.TemporaryDirectory() as d:
            args = ["-n", case["n"], "-s", case["steps"], "--precision", case["precision"], "--algorithm", case["algorithm"],
                    "--workload", case["workload"], "--save", "all", "--csv-detailed"]
            if case["theta"] is not None:
                args += ["--theta", case["theta"]]
            r = cli(case["dim"], args, cwd=d)
            assert r.returncode == 0, r.stderr
            frames, hdr_steps = oracle.read_positions_bin(os.path.join(d, "positions.bin"))
            en, _ = oracle.read_energy_bin(os.path.join(d, "energy.bin"))
        ref, ref_en = data[name + "__frames"], data[name + "__energy"]
        assert frames.shape == ref.shape and hdr_steps == case["steps"] and frames.dtype == ref.dtype
        assert np.array_equal(frames[0], ref[0])
        assert np.abs(frames - ref).max() <= (1e-11 if dbl else 2e-3) * np.abs(ref[0]).max(), name
        assert en.shape == ref_en.shape
        np.testing.assert_allclose(en, ref_en, rtol=1e-11 if dbl else 2e-3, err_msg=name)
        ran += 1
    assert ran >= 20


@pytest.mark.parametrize("dtype", [1, 0])
@pytest.mark.parametrize("dim", [3, 2])
def test_calc_energies_vs_oracle(nb, oracle, dtype, dim):
    for wl, n in (("uniform", 1), ("uniform", 2), ("galaxy", 255), ("uniform", 256), ("galaxy", 257), ("uniform", 5000)):
        ref = oracle.build_model(dtype, dim, wl, n)
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, wl, n))
        ke, pe = dev.calc_energies()
        oke, ope = oracle.calc_energies(ref)
        tol = 1e-12 if dtype == 1 else 2e-4  # float: the reference's sequential float sum carries ~1e-4 itself
        assert abs(ke - oke) <= tol * max(abs(oke), 1e-30) and abs(pe - ope) <= tol * max(abs(ope), 1e-30), (wl, n, ke, oke, pe, ope)
    s = nb.HostSystem(1, 3, 3)  # coincident distinct bodies keep the reference's m_i*m_j/eps term; self pairs are excluded
    s.m[:] = [1, 2, 3]
    s.c = 1.0
    d = nb.DeviceSystem.from_host(s)
    _, pe = d.calc_energies()
    assert abs(pe + 0.5 * 2 * (1 * 2 + 1 * 3 + 2 * 3) / np.finfo(np.float64).eps) <= 1e-12 * abs(pe)


def test_step_graph_replay_equals_direct_calls(nb):
    """A recorded step (HIP stream capture -> graph) replays to bitwise the same state as direct phase calls."""
    for algo in ("all-pairs", "bvh"):
        n, steps = 5000, 4
        d1 = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
        nb.run(d1, algo, steps, 0.5)
        d2 = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", n))
        if algo == "bvh":
            _ = d2.bvh  # allocate the tree outside the capture
        g = nb.StepGraph(d2, lambda: (d2.all_pairs_force() if algo == "all-pairs" else d2.bvh_force(0.5), d2.accelerate_step()))
        for _ in range(steps):
            g.launch()
        d2.sync()
        a, b = d1.download(), d2.download()
        for k in ("m", "x", "v", "a", "ao"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), (algo, k)
        g.close()


def test_workload_load_bin(nb, oracle):
    """--workload load f.bin: u32 n, u32 dim, f32 dt, f32 G, then n x (m, pos[D], vel[D]) as f32 (src/saving.h:25-68)."""
    import struct
    rng = np.random.default_rng(9)
    n, dim = 200, 3
    body = rng.standard_normal((n, 1 + 2 * dim)).astype(np.float32)
    body[:, 0] = np.abs(body[:, 0]) + 0.1
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sys.bin")
        with open(path, "wb") as f:
            f.write(struct.pack("<IIff", n, dim, 0.05, 2.0))
            f.write(body.tobytes())
        r = cli(3, ["--workload", "load", path, "--precision", "double", "--algorithm", "all-pairs", "-s", 3, "--csv-detailed", "--save", "pos"], cwd=d)
        assert r.returncode == 0, r.stderr
        frames, _ = oracle.read_positions_bin(os.path.join(d, "positions.bin"))
        r2 = cli(2, ["--workload", "load", path])
        assert r2.returncode != 0 and "compiled with D=2" in r2.stderr  # uncaught std::runtime_error, as the reference
    s = oracle.State(oracle.F64, 3, n)
    s.m[:], s.x[:], s.v[:] = body[:, 0], body[:, 1:4], body[:, 4:7]
    s.dt, s.c = float(np.float32(0.05)), 2.0
    ref = [s.x.copy()]
    oracle.run(s, "all-pairs", 3, frames=ref)
    assert frames.shape == (4, n, 3) and np.array_equal(frames[0], ref[0])
    assert np.abs(frames - np.array(ref)).max() <= 1e-11 * np.abs(ref[0]).max()


def test_call_sequence_and_shard_errors(nb):
    """Error behaviour of the ABI on a live device: state errors, empty shards, whole-system-only phases."""
    import ctypes as C
    dev = nb.DeviceSystem.from_host(nb.build_model(1, 3, "uniform", 100))
    t, st = dev.bvh, dev.state()
    with pytest.raises(nb.NbodyError, match="before nbody_bvh_bounding_box"):
        t.hilbert_sort(st, dev.stream)
    with pytest.raises(nb.NbodyError, match="before nbody_bvh_build_tree"):
        t.compute_force(st, 0.5, dev.stream)
    t.bounding_box(st, dev.stream)
    with pytest.raises(nb.NbodyError, match="whole system"):
        t.hilbert_sort(dev.state(10, 20), dev.stream)
    with pytest.raises(nb.NbodyError, match="single-GPU"):
        s2 = dev.state(10, 20)
        nb._check(nb.lib().nbody_all_pairs_collapsed_force(C.byref(s2), C.c_void_p(dev.stream)))
    before = dev.download().a.copy()
    dev.all_pairs_force(50, 0)  # empty shard: nothing launched, nothing written
    dev.accelerate_step(50, 0)
    assert np.array_equal(dev.download().a, before)
    with pytest.raises(nb.NbodyError, match="exceeds"):
        dev.all_pairs_force(90, 20)
    with pytest.raises(nb.NbodyError):
        nb.Bvh(1, 3, 1)  # the reference's tree needs at least one body pair


def test_abi_from_plain_c(tmp_path):
    """examples/abi_from_c.c: the boundary used from C99 with nothing but include/nbody_hip.h and -lnbody_hip."""
    import subprocess
    exe = str(tmp_path / "abi_from_c")
    libdir = os.path.join(ROOT, "stdpar-nbody_amd")
    build = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", "abi_from_c.c"), "-L" + libdir, "-lnbody_hip", "-Wl,-rpath," + libdir, "-lm",
                            "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "tree size" in r.stdout and "root mass" in r.stdout
    assert "K1 hand-off: failed 0" in r.stdout, r.stdout   # nbody_all_pairs_status from C


def cli_env(dim, args, env_extra, cwd=None):
    exe = os.path.join(ROOT, "stdpar-nbody_amd", "bin", f"nbody_hip_d{dim}")
    env = dict(os.environ)
    env.update(env_extra)
    return subprocess.run([exe] + [str(a) for a in args], cwd=cwd, env=env, capture_output=True, text=True, timeout=600)


def test_gpus_flag_multi_device_path_on_one_gpu():
    """--gpus N (not in the reference): one context per device, the ABI's shard per context, nbody_comm_create_all and an
    nbody_allgather_positions per step.  An explicit --gpus takes that path for any N, so on the one-GPU box the whole path
    runs with `--gpus 1` and must print exactly what the plain single-device run prints, in the default and in the --csv-detailed mode with saved
    frames; asking for more devices than are visible, or for a non-sharding algorithm, fails with a message."""
    args = ["-n", 3000, "-s", 12, "--precision", "double", "--algorithm", "all-pairs", "--workload", "galaxy", "--print-state"]
    plain = cli(3, args)
    forced = cli(3, args + ["--gpus", 1])
    assert plain.returncode == 0 and forced.returncode == 0, forced.stderr
    strip = lambda out: re.sub(r"Total time: .*", "", out)
    assert strip(plain.stdout) == strip(forced.stdout)
    with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d2:
        a2 = ["-n", 500, "-s", 4, "--precision", "float", "--algorithm", "all-pairs", "--csv-detailed", "--save", "pos"]
        r1, r2 = cli(2, a2, cwd=d1), cli(2, a2 + ["--gpus", 1], cwd=d2)
        assert r1.returncode == 0 and r2.returncode == 0, r2.stderr
        assert open(os.path.join(d1, "positions.bin"), "rb").read() == open(os.path.join(d2, "positions.bin"), "rb").read()
    import torch
    too_many = torch.cuda.device_count() + 1
    r = cli(3, ["-n", 1000, "--algorithm", "all-pairs", "--gpus", too_many])
    assert r.returncode != 0 and "HIP devices visible" in r.stderr
    r = cli(3, ["-n", 1000, "--algorithm", "bvh", "--gpus", 2])
    assert r.returncode != 0 and "all-pairs only" in r.stderr
    r = cli(3, ["-n", 1000, "-s", 2, "--algorithm", "all-pairs", "--gpus", 1, "--save", "energy", "--csv-detailed"])
    assert r.returncode != 0 and "one GPU only" in r.stderr


def test_multi_gpu_abi_from_plain_c(tmp_path):
    """examples/abi_multi_gpu.c: contexts with shard windows, nbody_comm_create_all and nbody_allgather_positions from C99,
    compared bitwise with a single whole-system context (run with the one device this box has)."""
    libdir = os.path.join(ROOT, "stdpar-nbody_amd")
    exe = str(tmp_path / "abi_multi_gpu")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "abi_multi_gpu.c"), "-L" + libdir, "-lnbody_hip", "-Wl,-rpath," + libdir,
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "bitwise equal" in r.stdout, r.stdout + r.stderr


def test_scheduling_forms_give_the_same_trajectory(nb):
    """K9's per-lane walks, its compiler-scheduled sweep and the sweep written as ISA — and the octree walk's two forms — are
    bitwise equal per force phase (tests/test_gpu_bvh.py, test_gpu_octree.py); here over a few hundred steps of an evolving
    system (trees deepen as escapers inflate the box): the final x, v, a are identical bit for bit.  The forms are selected
    through the ABI setters (nbody_bvh_set_traversal, nbody_octree_set_walk); the shipped library reads no environment."""
    def final(dtype, n, steps, algo, select):
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
        select(dev)
        nb.run(dev, algo, steps, 0.5)
        out = dev.download()
        dev.close()
        return out
    for dtype, n, steps in ((nb.F64, 60000, 300), (nb.F32, 70000, 150)):
        outs = [final(dtype, n, steps, "bvh", lambda d, m=m: d.bvh.set_traversal(m)) for m in (1, 5)]
        for o in outs[1:]:
            assert np.array_equal(o.x, outs[0].x) and np.array_equal(o.v, outs[0].v) and np.array_equal(o.a, outs[0].a), dtype
    a, b = (final(nb.F64, 50000, 400, "octree", lambda d, m=m: d.octree.set_walk(m)) for m in (2, 1))
    assert np.array_equal(a.x, b.x) and np.array_equal(a.v, b.v) and np.array_equal(a.a, b.a)


def test_config2_command_as_written(nb, oracle):
    """BASELINE config[1] typed as the reference would be run: `-n 65536 -s 100 --precision double --algorithm all-pairs` (no
    --workload: uniform, src/arguments.h:27; 10 warm-up + 90 timed steps, src/all_pairs.h:86-97).  The --csv-total row has the
    reference's shape with nsteps = 90, and the rows `--print-state` prints after the 100 steps are, character for character,
    those of the step loop whose steps tests/test_gpu_all_pairs.py::test_config2_as_written_100_steps checks against the oracle."""
    args = ["-n", 65536, "-s", 100, "--precision", "double", "--algorithm", "all-pairs"]
    r = cli(3, args + ["--csv-total"])
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "algorithm,dim,precision,nsteps,nbodies,total [s]"
    assert re.fullmatch(r"all-pairs,3,64,90,65536,\d+\.\d\d", lines[1]), lines[1]
    r = cli(3, args + ["--print-state"])
    assert r.returncode == 0, r.stderr
    start, final = oracle.parse_print_state(r.stdout)
    hs = nb.build_model(nb.F64, 3, "uniform", 65536)
    assert start == oracle.format_state_rows(hs)
    dev = nb.DeviceSystem.from_host(hs)
    nb.run(dev, "all-pairs", 100)
    assert final == oracle.format_state_rows(dev.download())
    dev.close()


def test_reference_program_with_the_backend_dropped_in(oracle, golden_print_state, golden_positions):
    """oracle/_ref/nbody_ref_hip_d{2,3} = the reference's UNMODIFIED main.cpp (CLI, generators, run_simulation, run_all_pairs, Saver,
    print) + the stub of INTEGRATION.md (examples/hip_backend.h), built by `make -C oracle ref_hip` in the build container
    (tests/test_integration_stub.py) — here it RUNS: row (b)'s binding meets the reference on the GPU.
      * all-pairs: the reference's own step driver with the GPU force as its Force&& f (src/all_pairs.h:52-53,77,88) — every
        double case prints the reference's golden text, start and final; --save all --csv-detailed writes the reference's frames;
      * NBODY_HIP_RESIDENT=1: the device-resident loop of the stub (force phases + leapfrog on the GPU) for all-pairs, bvh and
        octree — the reference's golden final rows again."""
    exe = {d: os.path.join(ROOT, "oracle", "_ref", f"nbody_ref_hip_d{d}") for d in (2, 3)}
    if not all(os.path.exists(e) for e in exe.values()):
        pytest.skip("oracle/_ref/nbody_ref_hip_d* not built (make -C oracle ref_hip needs /root/reference)")

    def run(dim, args, cwd=None, resident=False):
        env = dict(os.environ)
        env.pop("NBODY_HIP_RESIDENT", None)
        if resident:
            env["NBODY_HIP_RESIDENT"] = "1"
        return subprocess.run([exe[dim]] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True, timeout=600, env=env)

    thin = resident = 0
    for name, case in golden_print_state.items():
        if case["precision"] != "double" or case["algorithm"] == "all-pairs-collapsed":
            continue
        for res in (False, True):
            if not res and case["algorithm"] != "all-pairs":
                continue
            r = run(case["dim"], case["args"], resident=res)
            assert r.returncode == 0, (name, r.stderr)
            start, final = oracle.parse_print_state(r.stdout)
            assert start == case["start"], f"{name}: starting state"
            bad = [i for i, (a, b) in enumerate(zip(final, case["final"])) if a != b]
            assert len(final) == len(case["final"]) and not bad, f"{name} resident={res}: {len(bad)} final rows differ"
            assert re.search(r"Done simulation\nTotal time: \d+\.\d\d ms\n", r.stdout)
            thin, resident = thin + (not res), resident + res
    assert thin >= 12 and resident >= 36
    # the reference's Saver and CSV row under the reference's own driver, forces from the GPU
    meta, data = golden_positions
    name = "d3_double_all-pairs_galaxy_n1024_s20"
    case, ref = meta[name], data[name + "__frames"]
    with tempfile.TemporaryDirectory() as d:
        r = run(3, ["-n", 1024, "-s", 20, "--precision", "double", "--algorithm", "all-pairs", "--workload", "galaxy", "--save", "pos",
                    "--csv-detailed"], cwd=d)
        assert r.returncode == 0, r.stderr
        assert re.fullmatch(r"all-pairs,3,64,20,1024,\d+\.\d\d,\d+\.\d\d,\d+\.\d\d", r.stdout.strip())
        frames, hdr_steps = oracle.read_positions_bin(os.path.join(d, "positions.bin"))
    assert frames.shape == (21, 1024, 3) and hdr_steps == 20
    assert np.array_equal(frames[0], ref[0])
    for k, fid in enumerate(case["frame_ids"]):
        assert np.abs(frames[fid] - ref[k]).max() <= 1e-11 * np.abs(ref[0]).max(), fid
    # a backend error ends the program the reference's way: message on stderr, EXIT_FAILURE
    r = run(3, ["-n", 100, "--algorithm", "bvh", "--precision", "double"])
    assert r.returncode != 0 and "NBODY_HIP_RESIDENT" in r.stderr
