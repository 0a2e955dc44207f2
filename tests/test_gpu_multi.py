"""GPU: the multi-rank path of BASELINE config[4] exercised on the one-GPU box — bench.py's self-launcher (parent starts
fresh rank processes through torch.distributed.run and relays rank 0's JSON line), the RCCL process group, and the
library's own collective (nbody_comm_* / nbody_allgather_positions) at world size 1.  More than one rank needs more than
one GPU (RCCL refuses two ranks on one device); the partition logic for world > 1 is covered on CPU by
tests/test_sharded_gloo.py and the per-rank launch shape by test_config5_rank_windows_at_2pow20."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench_raw(extra_env, *args):
    env = dict(os.environ)
    env.update(extra_env)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=900)


def _bench(extra_env, *args):
    r = _bench_raw(extra_env, *args)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # stdout carries exactly one line
    return json.loads(lines[0])


def test_bench_self_launch_forced_dist_matches_plain_run():
    """`python bench.py --gpus 1` through the launcher (NBODY_BENCH_FORCE_DIST=1: parent -> torch.distributed.run -> rank 0,
    RCCL process group, nbody_comm, barrier, all-gather, max-reduce) reports the same workload, kernel and — within the
    box's run-to-run spread — the same rate as the plain single-process run."""
    args = ("--gpus", "1", "--steps", "2", "--warmup", "1", "--bodies", str(1 << 17), "--no-cpu-baseline", "--no-other-configs")
    plain = _bench({}, *args)
    forced = _bench({"NBODY_BENCH_FORCE_DIST": "1"}, *args)
    assert plain["rccl_world"] is None and plain["n_gpus"] == 1
    rw = forced["rccl_world"]
    assert rw["world"] == 1 and rw["backend"] == "nccl" and "nbody_comm" in rw["data_path"]
    assert forced["shards"] == [1 << 17] and plain["shards"] == [1 << 17]
    assert forced["allgather"]["sent_bytes_per_rank_per_step"] == (1 << 17) * 24
    assert forced["allgather"]["avg_ms"] is not None and forced["allgather"]["avg_ms"] < 5.0
    assert rw["exchange"] == "nbody" and rw["bitwise_vs_single"]["equal"] and rw["bitwise_vs_single"]["checked_ranks"] == [0]
    assert len(rw["per_rank"]["k1_ms"]["by_rank"]) == 1 and rw["per_rank"]["k1_ms"]["max"] > 0
    assert "after the last timed step" in rw["exchange_check"]
    for k in ("metric", "unit", "dtype", "scaling", "steps", "warmup"):
        assert plain[k] == forced[k], k
    assert plain["roofline"]["kernel"] == forced["roofline"]["kernel"]
    assert plain["config"]["n_bodies"] == forced["config"]["n_bodies"] == 1 << 17
    assert abs(forced["value"] / plain["value"] - 1.0) < 0.3, (forced["value"], plain["value"])
    assert 0.2 < plain["roofline"]["frac"] < 0.7


def test_bench_line_carries_configs_1_to_4_and_the_handoff_status():
    """The line the driver records also times BASELINE.json configs[0..3] (after the headline's clock has stopped) with each one's
    dominant kernel, its average launch time from HIP events, and achieved / peak = frac against the roofline that bounds it (one
    meaning of `frac` in the whole list); config 1 — the one that names the CPU path — carries both legs, the CPU port timed over
    the whole config; and K1's chunk hand-off status over the run (no failure; how many waves had to wait)."""
    line = _bench({}, "--gpus", "1", "--steps", "1", "--warmup", "1", "--bodies", str(1 << 16), "--no-cpu-baseline")
    assert line["roofline"]["handoff"]["failed"] is False and line["roofline"]["handoff"]["polls"] >= 0
    cfg = line["configs"]
    assert len(cfg) == 4 and not any("failed" in c for c in cfg), cfg
    c1, c2, c3, c4 = cfg
    assert "2D float" in c1["workload"] and "10000" in c1["workload"] and c1["bound"] == "valu_fp32", c1
    assert "all_pairs_force" in c1["kernel"] and 0.1 < c1["frac"] < 0.6 and c1["avg_kernel_ms"] <= c1["ms_per_step"] + 0.01, c1
    assert c1["cpu_kind"] == "port" and c1["cpu_cores"] >= 1 and c1["cpu_ms_per_step"] > c1["ms_per_step"], c1
    assert "65536" in c2["workload"] and "all_pairs_force_sgpr_kernel<double,3" in c2["kernel"] and c2["bound"] == "valu_fp64"
    assert 0.25 < c2["frac"] < 0.6 and 0.97 * c2["avg_kernel_ms"] <= c2["ms_per_step"] <= c2["avg_kernel_ms"] + 0.08, c2   # the step IS its K1 (+ 40 us; the two are timed over different steps)
    assert c2["handoff"]["failed"] is False
    assert "262144" in c3["workload"] and "collapsed" in c3["kernel"] and c3["bound"] == "valu_fp32" and 0.2 < c3["frac"] < 0.6, c3
    assert c3["avg_kernel_ms"] <= c3["ms_per_step"] * 1.02
    if c3["rocprof"]:   # an f32 kernel: rocprofv3's VALUBusy formula is no fraction there and must not be published as one
        assert c3["rocprof"]["valu_busy_frac"] is None or c3["rocprof"]["valu_busy_frac"] <= 1.0
    assert "1000000" in c4["workload"] and "bvh_force_sweep_isa_kernel" in c4["kernel"] and c4["bound"] == "valu_fp64", c4
    assert 2e11 < c4["node_tests_per_s"] < 2e12 and 2000 < c4["node_tests_per_body"] < 8000, c4
    assert c4["avg_kernel_ms"] < c4["ms_per_step"] < c4["avg_kernel_ms"] + 1.0
    assert c4["peak"] == 78.6 and abs(c4["frac"] - c4["achieved"] / c4["peak"]) < 1e-12 and 0.05 < c4["frac"] < 0.4, c4
    assert c4["valu_issue_frac"] is None or 0.2 < c4["valu_issue_frac"] < 0.8
    assert c4["not_the_bound"]["hbm_cold_bytes_per_s"] > c4["not_the_bound"]["hbm_peak_bytes_per_s"]   # why 8(d)'s HBM model is not the bound
    for c in cfg:
        assert c["ms_per_step"] > 0 and c["body_steps_per_s"] > 0 and abs(c["frac"] - c["achieved"] / c["peak"]) < 1e-12


def test_bench_refuses_more_gpus_than_visible():
    import torch
    have = torch.cuda.device_count()
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "visible" in r.stderr and not r.stdout.strip()


def test_comm_world1_allgather_and_partition(nb):
    """nbody_comm at world 1 on RCCL: create from a unique id, the all-gather is the identity, a state whose window is not
    the rank's partition is rejected; nbody_shard_range is the partition sharded.py uses."""
    import torch
    assert nb.Comm.rccl_version() > 20000
    comm = nb.Comm(1, 0, nb.Comm.unique_id(), 0)
    hs = nb.build_model(1, 3, "galaxy", 5000)
    sim = nb.parallel.ShardedAllPairs(hs, 0, 1, torch_device=torch.device("cuda", 0), force_exchange=True, comm=comm)
    ref = nb.DeviceSystem.from_host(nb.build_model(1, 3, "galaxy", 5000))
    for _ in range(3):
        sim.step()
    nb.run(ref, "all-pairs", 3)
    torch.cuda.synchronize()
    x, v, a = sim.gather_state()
    out = ref.download()
    assert np.array_equal(x, out.x) and np.array_equal(v, out.v) and np.array_equal(a, out.a)
    st = sim.state()
    st.first, st.count = 10, 100
    rc = nb.lib().nbody_allgather_positions(comm.h, C.byref(st), None)
    assert rc == 1 and b"owns" in nb.lib().nbody_last_error()
    comm.close()
    for n in (7, 5000, 1 << 20, 1000003):
        for w in (1, 2, 3, 8):
            for r in range(w):
                assert nb.shard_range(n, r, w) == nb.parallel.shard_range(n, r, w)


def test_sharded_context_download_places_rows(nb):
    """nbody_ctx_set_shard: the context's view is the window, v/a/ao point at the owned rows, download writes the owned rows
    only — the form the CLI's --gpus N drives one context per device with."""
    n = 6000
    hs = nb.build_model(1, 3, "galaxy", n)
    whole = nb.DeviceSystem.from_host(hs)
    nb.run(whole, "all-pairs", 1)
    ref = whole.download()
    parts = []
    for r in range(3):
        d = nb.DeviceSystem.from_host(hs)
        f, e = nb.shard_range(n, r, 3)
        d.set_shard(f, e - f)
        st = d.state()
        assert (st.first, st.count, st.sz) == (f, e - f, n)
        assert nb.lib().nbody_all_pairs_force(C.byref(st), C.c_void_p(d.stream)) == 0
        assert nb.lib().nbody_accelerate_step(C.byref(st), C.c_void_p(d.stream)) == 0
        parts.append((d, f, e))
    out = nb.HostSystem(1, 3, n)
    for d, f, e in parts:
        nb._check(nb.lib().nbody_download(d.h, nb._p(out.m), nb._p(out.x), nb._p(out.v), nb._p(out.a), nb._p(out.ao)))
    assert np.array_equal(out.x, ref.x) and np.array_equal(out.v, ref.v) and np.array_equal(out.a, ref.a)


def _two_rank_worker(rank, world, port, n, steps, q, algo="all-pairs"):
    """One of two real rank processes sharing the box's one GPU: HIP kernels on its shard window, K3, exchange."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    dist.init_process_group("gloo", rank=rank, world_size=world)   # RCCL refuses two ranks on one device: CPU group, staged exchange
    nb = load_package()
    torch.cuda.set_device(0)
    hs = nb.build_model(nb.F64, 3, "galaxy", n)
    if algo == "octree":
        sim = nb.parallel.ShardedOctree(hs, rank, world, theta=0.5, torch_device=torch.device("cuda", 0))
    else:
        sim = nb.parallel.ShardedAllPairs(hs, rank, world, torch_device=torch.device("cuda", 0))
    for _ in range(steps):
        sim.step()
    torch.cuda.synchronize()
    x, v, a = sim.gather_state()
    if rank == 0:
        q.put((x, v, a))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world,algo", [(70001, 2, "all-pairs"), (8192, 3, "all-pairs"), (30001, 2, "octree")])
def test_two_rank_processes_on_one_gpu_equal_single_gpu(nb, n, world, algo):
    """ranks as real processes, each launching the HIP kernels on its own shard window and exchanging positions
    every step, against the single-GPU trajectory — bitwise in x, v, a.  Both ranks use the box's one MI355X, so the
    process group is gloo and the shards are staged through the host (the RCCL exchange itself needs one GPU per rank);
    70001 bodies: uneven shards, the 8-slice scalar-stream K1 with source chunks; the octree case: every rank rebuilds the
    whole tree from the gathered positions and walks it for its own bodies (ShardedOctree)."""
    import socket
    import torch.multiprocessing as mp
    steps = 3
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, n, steps, q, algo)) for r in range(world)]
    for p in procs:
        p.start()
    x, v, a = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "galaxy", n))
    nb.run(dev, algo, steps, 0.5)
    ref = dev.download()
    assert np.array_equal(x, ref.x) and np.array_equal(v, ref.v) and np.array_equal(a, ref.a)


def test_bench_multi_rank_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` as typed, rehearsed on the one-GPU box (NBODY_BENCH_SHARE_GPU=1: both ranks on device 0,
    gloo group, staged exchange): the launcher starts two ranks, every rank-dependent branch of bench.py runs (shards, the
    cross-rank position check, max-over-ranks of the time), rank 0's single line is relayed and marked as a rehearsal."""
    out = _bench({"NBODY_BENCH_SHARE_GPU": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--bodies", str(1 << 16), "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["shards"] == [1 << 15, 1 << 15] and "REHEARSAL" in out["rehearsal"]
    assert out["rccl_world"]["world"] == 2 and out["rccl_world"]["backend"] == "gloo"
    assert "identical on all 2 rank(s)" in out["rccl_world"]["exchange_check"]
    assert out["allgather"]["sent_bytes_per_rank_per_step"] == (1 << 15) * 24 and out["value"] > 0
    rw = out["rccl_world"]
    assert rw["exchange"] == "rehearsal" and "after the last timed step" in rw["exchange_check"]
    assert rw["bitwise_vs_single"]["equal"] and rw["bitwise_vs_single"]["checked_ranks"] == [0, 1]
    assert rw["bitwise_vs_single"]["targets_per_window"] == 4096
    pr = rw["per_rank"]
    assert len(pr["k1_ms"]["by_rank"]) == 2 and 0 < pr["k1_ms"]["min"] <= pr["k1_ms"]["max"]
    assert len(pr["allgather_ms"]["by_rank"]) == 2 and pr["allgather_ms"]["max"] >= pr["allgather_ms"]["min"] >= 0


@pytest.mark.parametrize("fault,needle", [
    ("stale_exchange:1", "after the warm-up exchange"),   # a rank whose exchange delivers nothing: caught before the clock starts
    ("stale_late:1", "after the timed steps"),            # ... from the first timed step on: caught by the end-of-run checksum
    ("corrupt_a:1", "bitwise_vs_single failed"),          # a rank whose rows differ from the single-GPU sum
    ("nan_a:1", "NaN accelerations or reports a failed K1 chunk hand-off"),   # a NaN row OUTSIDE the sampled windows (what a failed hand-off leaves)
])
def test_bench_multi_rank_run_fails_on_a_broken_exchange(fault, needle):
    """a multi-rank line cannot be green with a dead exchange.  Rehearsed on the one-GPU box (two ranks on
    device 0): with an injected fault on rank 1 every rank exits non-zero, the launcher relays the status, stdout carries NO
    result line and stderr names the check that failed."""
    r = _bench_raw({"NBODY_BENCH_SHARE_GPU": "1", "NBODY_BENCH_FAULT": fault}, "--gpus", "2", "--steps", "2", "--warmup", "1",
                   "--bodies", str(1 << 14), "--no-cpu-baseline")
    assert r.returncode != 0, r.stdout
    assert not r.stdout.strip(), r.stdout
    assert needle in r.stderr, r.stderr[-3000:]


def test_bench_fails_without_the_library_collective_unless_asked():
    """The communicator of the library (nbody_comm over RCCL) is the data path of the metric: if it cannot be created on some
    rank the run fails on every rank before anyone enters ncclCommInitRank — no silent torch.distributed fallback.  The torch
    exchange exists only behind an explicit --exchange torch, and the line it prints is marked."""
    args = ("--gpus", "1", "--steps", "1", "--warmup", "1", "--bodies", str(1 << 14), "--no-cpu-baseline")
    r = _bench_raw({"NBODY_BENCH_FORCE_DIST": "1", "NBODY_BENCH_FAULT": "comm_create:0"}, *args)
    assert r.returncode != 0 and not r.stdout.strip(), r.stdout
    assert "nbody_comm cannot be created" in r.stderr and "no fallback" in r.stderr, r.stderr[-3000:]
    out = _bench({"NBODY_BENCH_FORCE_DIST": "1"}, *args, "--exchange", "torch")
    assert out["rccl_world"]["exchange"] == "torch" and "NOT the library's collective" in out["not_the_library_collective"]
    assert out["rccl_world"]["bitwise_vs_single"]["equal"]


def test_rccl_uneven_shards_two_gpus(nb):
    """the grouped ncclSend/ncclRecv form of nbody_allgather_positions (sz % W != 0) between real devices.  Needs
    two GPUs in this process; skipped on the one-GPU box (the driver's 8-GPU node runs bench.py, whose self-check covers the
    even form)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    exe_src = os.path.join(ROOT, "examples", "abi_multi_gpu.c")
    libdir = os.path.join(ROOT, "stdpar-nbody_amd")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "abi_multi_gpu")
        r = subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), exe_src, "-L" + libdir, "-lnbody_hip",
                            "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for n in ("7001", "8192"):   # uneven shards: grouped ncclSend/ncclRecv; even: one in-place ncclAllGather
            r = subprocess.run([exe, "2", n], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0 and "bitwise equal" in r.stdout, r.stdout + r.stderr
