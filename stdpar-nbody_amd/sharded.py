"""Body sharding for multi-GPU all-pairs: one process per GPU, targets split contiguously by rank, every
rank keeps all N positions/masses, and ONE exchange per step — an all-gather of the updated position
shards (north_star / SURVEY §8e).  Nothing else moves: v, a, ao stay local; masses are constant and replicated
at start-up.  On GPUs the exchange is the library's own collective, nbody_allgather_positions (RCCL over xGMI
behind the C ABI, in place on the x array, include/nbody_hip.h); the launcher only hands the RCCL unique id to the
ranks.  The torch.distributed form of the same exchange (one all_gather over padded shards) remains for the
CPU/gloo tests of the partition logic and as `exchange="torch"`.

The reference has no counterpart (single process, single device).  The octree shards the same way (ShardedOctree:
every rank rebuilds the whole tree from the gathered positions, walks it for its own bodies only).  bvh and
all-pairs-collapsed do not shard ("replicas only": the bvh sort permutes all five arrays every step).  The per-target summation order does not depend on the shard window, so any
world size gives bitwise the same trajectory as one GPU (tests/test_gpu_all_pairs.py checks this with
shard windows on one GPU; tests/test_sharded_gloo.py checks the exchange logic with world_size 2).

torch is used for device memory, the stream handle and the process group only.
"""
import ctypes as C

import numpy as np


def shard_range(n, rank, world):
    """Contiguous balanced split: rank r owns [n*r//world, n*(r+1)//world)."""
    return (n * rank) // world, (n * (rank + 1)) // world


class ShardedAllPairs:
    """run_all_pairs' step (force, then accelerate_step; src/all_pairs.h:86-91) over a shard of targets."""

    def __init__(self, hs, rank, world, torch_device=None, pkg=None, force_exchange=False, comm=None):
        import sys
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = rank, world
        self.n, self.dim, self.dtype = hs.n, hs.dim, hs.dtype
        self.first, end = shard_range(hs.n, rank, world)
        self.count = end - self.first
        self.device = torch_device if torch_device is not None else torch.device("cpu")
        self.pkg = pkg or sys.modules["stdpar_nbody_amd"]
        self.lib = self.pkg.lib()
        to = lambda arr: torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
        self.m = to(hs.m)                       # all bodies
        self.x = to(hs.x)                       # all bodies; rows [first, first+count) are ours to update
        self.v = to(hs.v[self.first:end])       # owned bodies only
        self.a = to(hs.a[self.first:end])
        self.ao = to(hs.ao[self.first:end])
        self.dt, self.c = float(hs.dt), float(hs.c)
        self.equal = hs.n % world == 0
        self.exchange = world > 1 or force_exchange  # force_exchange: run the collective even with one rank (smoke test)
        self.comm = comm                             # nbody_comm (C ABI): the exchange of the GPU path
        self.shards = [shard_range(hs.n, r, world) for r in range(world)]
        self.maxc = max(e - f for f, e in self.shards)
        if self.exchange and comm is None:
            self.use_torch_exchange()

    def use_torch_exchange(self):
        """Switch to the torch.distributed form of the exchange: one all_gather over padded shards (the CPU/gloo tests of
        the partition logic, and bench.py only on an explicit `--exchange torch`)."""
        torch = self.torch
        self.comm = None
        self.send = torch.zeros((self.maxc,) + tuple(self.x.shape[1:]), dtype=self.x.dtype, device=self.device)
        self.recv = torch.empty((self.world * self.maxc,) + tuple(self.x.shape[1:]), dtype=self.x.dtype, device=self.device)

    def state(self, whole=False):
        """The shard's view; whole=True: the window covers every body (phases that read only m and x)."""
        st = self.pkg.nbody_state()
        st.m, st.x = self.m.data_ptr(), self.x.data_ptr()
        st.v, st.a, st.ao = self.v.data_ptr(), self.a.data_ptr(), self.ao.data_ptr()
        st.dt, st.c = self.dt, self.c
        st.sz, st.first, st.count = self.n, (0 if whole else self.first), (self.n if whole else self.count)
        st.dtype, st.dim = self.dtype, self.dim
        return st

    def _stream(self):
        if self.device.type == "cuda":
            return self.torch.cuda.current_stream(self.device).cuda_stream
        return None

    def exchange_positions(self):
        """The one collective of the path: all ranks end up with every rank's updated position shard."""
        if not self.exchange:
            return
        if self.comm is not None:   # in place on x, stream-ordered after K3 (RCCL behind the C ABI)
            self.comm.allgather_positions(self.state(), self._stream())
            return
        end = self.first + self.count
        self.send[:self.count].copy_(self.x[self.first:end])
        if self.x.is_cuda and self.dist.get_backend() == "gloo":
            # device tensors under a CPU process group (the two-ranks-on-one-GPU test): stage the shards through the host
            send = self.send.cpu()
            recv = [self.torch.empty_like(send) for _ in range(self.world)]
            self.dist.all_gather(recv, send)
            for r, (f, e) in enumerate(self.shards):
                if r != self.rank:
                    self.x[f:e].copy_(recv[r][:e - f].to(self.device))
            return
        self.dist.all_gather_into_tensor(self.recv, self.send)
        for r, (f, e) in enumerate(self.shards):
            if r != self.rank:
                self.x[f:e].copy_(self.recv[r * self.maxc:r * self.maxc + (e - f)])

    def _call(self, rc):
        if rc:
            raise self.pkg.NbodyError(self.lib.nbody_last_error().decode())

    # the two phase calls of the step, through the C ABI on the tensors' device pointers (include/nbody_hip.h)
    def force_phase(self):
        st = self.state()
        self._call(self.lib.nbody_all_pairs_force(C.byref(st), C.c_void_p(self._stream())))

    def integrate_phase(self):
        st = self.state()
        self._call(self.lib.nbody_accelerate_step(C.byref(st), C.c_void_p(self._stream())))

    def step(self, force_events=None, exchange_events=None):
        if force_events:
            force_events[0].record()
        self.force_phase()
        if force_events:
            force_events[1].record()
        self.integrate_phase()
        if exchange_events:
            exchange_events[0].record()
        self.exchange_positions()
        if exchange_events:
            exchange_events[1].record()

    def gather_state(self):
        """Full (x, v, a) on every rank as numpy, for checks: v and a are gathered too (test/diagnostic only)."""
        torch, dist = self.torch, self.dist
        x = self.x.cpu().numpy().copy()
        if self.world == 1:
            return x, self.v.cpu().numpy().copy(), self.a.cpu().numpy().copy()
        outs = []
        for t in (self.v, self.a):
            parts = [None] * self.world
            dist.all_gather_object(parts, t.cpu().numpy())
            outs.append(np.concatenate(parts, axis=0))
        return x, outs[0], outs[1]

    def describe(self):
        if self.world == 1 and self.comm is None:
            how = "none (one rank, nothing is exchanged)"
        elif self.comm is not None:
            how = "nbody_allgather_positions (" + ("ncclAllGather in place" if self.equal else "grouped ncclSend/ncclRecv") + ")"
        else:
            how = "torch.distributed all_gather over padded shards"
        return f"rank shard {self.count} of {self.n} targets, exchange: {how}"


class ShardedOctree(ShardedAllPairs):
    """run_octree's step (src/octree.h:321-327) over a shard of targets.  The tree needs every body, so each rank rebuilds
    it from the gathered positions (bounds + insert + multipoles: 0.65 ms of a 4.5 ms step at N=1e6 on one MI355X) and
    walks it for its own bodies only; the octree does not permute bodies, so v, a, ao stay local and the one exchange per
    step is the same all-gather of positions as for all-pairs.  A body's force depends on the tree and that body alone:
    any world size gives bitwise the single-GPU trajectory.

    FROZEN (round 5): north_star scopes the multi-GPU path to all-pairs and SURVEY 8(e) says "replicas only" for the trees.  This
    class stays because the multi-process tests use it as a second client of the exchange; it is not benchmarked and not grown."""

    def __init__(self, hs, rank, world, theta=0.5, **kw):
        super().__init__(hs, rank, world, **kw)
        self.theta = float(theta)
        self.tree = self.make_tree()
        self.steps_done = 0

    def make_tree(self):
        dev = self.device.index if self.device.type == "cuda" and self.device.index is not None else -1
        return self.pkg.Octree(self.dtype, self.dim, self.n, dev)

    def force_phase(self):
        """The force phase of run_octree (src/octree.h:321-326): build from ALL bodies, walk for the rank's window."""
        st, whole, stream, tree = self.state(), self.state(whole=True), self._stream(), self.tree
        tree.clear(stream)
        tree.compute_bounds(whole, stream)
        tree.insert(whole, stream)
        tree.compute_tree(stream)
        tree.compute_force(st, self.theta, stream)

    CHECK_EVERY = 64

    def check(self):
        """Raises if any build or walk since the last check flagged trouble (nbody_octree_info): a flagged build drops
        mass, so a run must not integrate on.  Called every CHECK_EVERY steps; call it once more after the last step."""
        self.tree.info(self._stream())

    def step(self, force_events=None):
        if force_events:
            force_events[0].record()
        self.force_phase()
        if force_events:
            force_events[1].record()
        self.integrate_phase()
        self.exchange_positions()
        self.steps_done += 1
        if self.steps_done % self.CHECK_EVERY == 0:
            self.check()

    def describe(self):
        return "octree: tree rebuilt on every rank, " + super().describe()
