"""CPU, world_size 2, gloo: the body-sharding + position all-gather logic of stdpar-nbody_amd/sharded.py.
The two phase calls are overridden in test-only subclasses that run the oracle, so what is under test is the
partition, the exchange and that sharded == unsharded bit-for-bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_package


def oracle_sims(nb):
    """Test-only subclasses: the two phase calls of a step are replaced by the oracle acting on the CPU tensors, so what
    runs from the product is the partition, the exchange and the step sequence of sharded.py."""
    import oracle as O

    def o_state(sim):
        s = O.State(sim.dtype, sim.dim, sim.n)
        s.m, s.x = sim.m.numpy(), sim.x.numpy()
        s.dt, s.c = sim.dt, sim.c
        return s

    def integrate(sim):
        sub = O.State(sim.dtype, sim.dim, sim.count)
        sub.x = sim.x.numpy()[sim.first:sim.first + sim.count]  # view: updated in place
        sub.v, sub.a, sub.ao = sim.v.numpy(), sim.a.numpy(), sim.ao.numpy()
        sub.dt = sim.dt
        O.accelerate_step(sub)

    class OracleAllPairs(nb.parallel.ShardedAllPairs):
        def force_phase(self):
            s = o_state(self)
            s.a = np.zeros_like(s.x)
            O.all_pairs_force(s, self.first, self.count)
            self.a.numpy()[:] = s.a[self.first:self.first + self.count]

        integrate_phase = integrate

    class OracleOctree(nb.parallel.ShardedOctree):
        def make_tree(self):
            return None

        def check(self):
            pass

        def force_phase(self):
            s = o_state(self)
            s.a = np.zeros_like(s.x)
            O.octree_step_force(s, self.theta)
            self.a.numpy()[:] = s.a[self.first:self.first + self.count]

        integrate_phase = integrate

    return OracleAllPairs, OracleOctree


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, steps, q, algorithm="all-pairs"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nb = load_package()
    import oracle as O
    o = O.build_model(O.F64, 3, "galaxy", n)
    hs = nb.HostSystem(nb.F64, 3, o.n)
    for k in ("m", "x", "v", "a", "ao"):
        getattr(hs, k)[:] = getattr(o, k)
    hs.dt, hs.c = o.dt, o.c
    OracleAllPairs, OracleOctree = oracle_sims(nb)
    if algorithm == "octree":
        sim = OracleOctree(hs, rank, world, theta=0.5)
    else:
        sim = OracleAllPairs(hs, rank, world)
    for _ in range(steps):
        sim.step()
    x, v, a = sim.gather_state()
    if rank == 0:
        q.put((x, v, a))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [256, 301])  # 301: uneven shards -> per-owner broadcast path
def test_two_rank_sharded_equals_single(n, oracle):
    steps = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    x, v, a = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = oracle.build_model(oracle.F64, 3, "galaxy", n)
    oracle.run(ref, "all-pairs", steps)
    assert np.array_equal(x, ref.x) and np.array_equal(v, ref.v) and np.array_equal(a, ref.a)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_octree_equals_single(oracle, world):
    n, steps = 301, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, steps, q, "octree")) for r in range(world)]
    for p in procs:
        p.start()
    x, v, a = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = oracle.build_model(oracle.F64, 3, "galaxy", n)
    oracle.run(ref, "octree", steps, 0.5)
    assert np.array_equal(x, ref.x) and np.array_equal(v, ref.v) and np.array_equal(a, ref.a)


def test_shard_ranges_cover_everything(nb):
    for n in (1, 7, 1 << 20, 1000003):
        for w in (1, 2, 3, 8):
            edges = [nb.parallel.shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges[:-1], edges[1:]))
            sizes = [e - f for f, e in edges]
            assert max(sizes) - min(sizes) <= 1
