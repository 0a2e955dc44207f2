"""Python host-side mirror of the reference's step-driver interface over the gfx950 backend.

This module is PLUMBING: it binds the C ABI of libnbody_hip.so (include/nbody_hip.h) with ctypes and
exposes the same phase calls the reference's drivers make (run_all_pairs, src/all_pairs.h:52-106;
run_bvh, src/bvh.h:327-418).  All arithmetic happens in the HIP kernels; there is NO CPU fallback —
if the shared library or a HIP device is missing every entry point raises.

The directory name contains a hyphen, so import it with the loader in `tests/conftest.py`,
`bench.py` or `__graft_entry__.py` (importlib by path, module name `stdpar_nbody_amd`).
"""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libnbody_hip.so")
HOST_LIB_PATH = os.path.join(HERE, "libnbody_host.so")

F32, F64 = 0, 1
UNIFORM, PLUMMER, GALAXY = 0, 1, 2
WORKLOADS = {"uniform": UNIFORM, "plummer": PLUMMER, "galaxy": GALAXY}


class NbodyError(RuntimeError):
    pass


class nbody_state(C.Structure):
    """include/nbody_hip.h: device-pointer mirror of System<T,N>::state_t (src/system.h:41-50)."""
    _fields_ = [("m", C.c_void_p), ("x", C.c_void_p), ("v", C.c_void_p), ("a", C.c_void_p), ("ao", C.c_void_p),
                ("dt", C.c_double), ("c", C.c_double), ("sz", C.c_uint32), ("first", C.c_uint32), ("count", C.c_uint32),
                ("dtype", C.c_int32), ("dim", C.c_int32), ("tuning", C.c_uint32)]


def build(verbose=False):
    """Compile libnbody_hip.so (hipcc --offload-arch=gfx950), libnbody_host.so and the CLI, in-tree."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", HERE, "all"], stdout=out)


# every symbol include/nbody_hip.h declares (tests/test_abi.py checks the .so exports them all)
ABI_SYMBOLS = [
    "nbody_abi_version", "nbody_last_error", "nbody_device_info", "nbody_all_pairs_force", "nbody_all_pairs_collapsed_force",
    "nbody_accelerate_step", "nbody_calc_energies", "nbody_all_pairs_configure", "nbody_all_pairs_source_path", "nbody_bvh_create", "nbody_bvh_destroy",
    "nbody_bvh_bounding_box", "nbody_bvh_get_bounding_box", "nbody_bvh_hilbert_sort", "nbody_bvh_build_tree",
    "nbody_octree_create", "nbody_octree_destroy", "nbody_octree_clear", "nbody_octree_compute_bounds", "nbody_octree_insert",
    "nbody_octree_compute_tree", "nbody_octree_compute_force", "nbody_octree_info", "nbody_octree_enable_counters",
    "nbody_octree_read_counters", "nbody_bvh_compute_force", "nbody_bvh_read", "nbody_bvh_enable_counters", "nbody_bvh_set_traversal", "nbody_bvh_nnodes", "nbody_create",
    "nbody_bvh_create_on", "nbody_octree_create_on", "nbody_octree_set_walk", "nbody_octree_set_build", "nbody_octree_set_step_budget", "nbody_bvh_set_launch_order",
    "nbody_destroy", "nbody_upload", "nbody_download", "nbody_ctx_state", "nbody_ctx_stream", "nbody_stream_sync",
    "nbody_graph_begin", "nbody_graph_end", "nbody_graph_launch", "nbody_graph_destroy",
    "nbody_ctx_configure_all_pairs", "nbody_ctx_set_shard", "nbody_all_pairs_describe",
    "nbody_comm_get_unique_id", "nbody_comm_create", "nbody_comm_create_all", "nbody_comm_destroy", "nbody_comm_world",
    "nbody_comm_rank", "nbody_comm_rccl_version", "nbody_shard_range", "nbody_comm_group_begin", "nbody_comm_group_end",
    "nbody_allgather_positions", "nbody_bvh_opening_thresholds", "nbody_all_pairs_pair_rule", "nbody_all_pairs_status",
]
ABI_MAJOR = 2
COMM_ID_BYTES = 128

_lib = None
_host = None


def lib():
    """Load libnbody_hip.so; raises NbodyError when it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NbodyError(f"{LIB_PATH} is missing: run `make -C stdpar-nbody_amd` (or __graft_entry__.build())")
        # One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64; if this library pulled in /opt/rocm's
        # first, a later `import torch` (sharded.py, bench.py) would bring a second runtime that finds no devices
        # ("No HIP GPUs are available").  Loaded in this order, both resolve to the copy torch brought.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        if L.nbody_abi_version() // 1000 != ABI_MAJOR:
            raise NbodyError(f"{LIB_PATH} has ABI version {L.nbody_abi_version()}, this binding needs major version {ABI_MAJOR}")
        L.nbody_last_error.restype = C.c_char_p
        L.nbody_ctx_stream.restype = C.c_void_p
        L.nbody_bvh_nnodes.restype = C.c_uint32
        L.nbody_shard_range.restype = None
        _lib = L
    return _lib


def host_lib():
    """libnbody_host.so: the product's ISO-C++ workload generators (host/models.hpp) behind a C entry."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise NbodyError(f"{HOST_LIB_PATH} is missing: run `make -C stdpar-nbody_amd host`")
        H = C.CDLL(HOST_LIB_PATH)
        H.nbody_host_build_model.restype = C.c_int64
        _host = H
    return _host


def _check(rc):
    if rc != 0:
        raise NbodyError(f"nbody backend error {rc}: {lib().nbody_last_error().decode()}")


def np_dtype(dtype):
    return np.float32 if dtype == F32 else np.float64


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def bvh_opening_thresholds(dtype, width2, theta):
    """The opening thresholds nbody_bvh_build_tree stores (nbody_bvh_read what = 6) for given width^2 values, computed on the
    host by the same function (no device needed)."""
    w2 = np.ascontiguousarray(width2, np_dtype(dtype))
    out = np.empty_like(w2)
    _check(lib().nbody_bvh_opening_thresholds(dtype, _p(w2), C.c_double(theta), C.c_size_t(w2.size), _p(out)))
    return out


def device_info(device=0):
    arch = C.create_string_buffer(128)
    cus = C.c_int()
    _check(lib().nbody_device_info(device, arch, C.c_size_t(128), C.byref(cus)))
    return arch.value.decode(), cus.value


def configure_all_pairs(split=0, targets_per_thread=0, source_path=None):
    """K1 knobs (include/nbody_hip.h): source split, targets per lane, and — when given — how a source record reaches
    the lanes (0 auto, 1 LDS tiles, 2 scalar stream)."""
    _check(lib().nbody_all_pairs_configure(split, targets_per_thread))
    if source_path is not None:
        _check(lib().nbody_all_pairs_source_path(int(source_path)))


def tuning(split=0, targets_per_thread=0, source_path=0):
    """NBODY_TUNING(split, targets_per_thread, source_path) of include/nbody_hip.h; 0 = the library default."""
    if not (split or targets_per_thread or source_path):
        return 0
    return (split & 15) | ((targets_per_thread & 3) << 4) | ((source_path & 3) << 6) | 0x100


def describe_all_pairs(st):
    """The K1 launch the library will make for this view (kernel template arguments, tile, chunks, pair math)."""
    buf = C.create_string_buffer(256)
    _check(lib().nbody_all_pairs_describe(C.byref(st), buf, C.c_size_t(256)))
    return buf.value.decode()


def all_pairs_pair_rule(st, stream=None):
    """(sparse, volume): the per-pair rounding form K1 takes for this state (nbody_all_pairs_pair_rule)."""
    sparse, vol = C.c_int(), C.c_double()
    _check(lib().nbody_all_pairs_pair_rule(C.byref(st), C.c_void_p(stream), C.byref(sparse), C.byref(vol)))
    return bool(sparse.value), vol.value


def all_pairs_status(stream=None, clear=False, check=True):
    """K1's chunk hand-off status of a stream (nbody_all_pairs_status; waits for the stream): a dict with `failed`, the
    (block, group, chunk) of the first failure, and `polls` / `waits` — how often waves had to wait for their turn.  Raises
    NbodyError while a failure is recorded, unless check is False."""
    out = (C.c_uint64 * 6)()
    rc = lib().nbody_all_pairs_status(C.c_void_p(stream), out, 1 if clear else 0)
    if check:
        _check(rc)
    return {"failed": bool(out[0]), "block": int(out[1]), "group": int(out[2]), "chunk": int(out[3]), "polls": int(out[4]),
            "waits": int(out[5]), "rc": rc}


def shard_range(n, rank, world):
    """nbody_shard_range: rank r of W owns bodies [n*r//W, n*(r+1)//W) — the partition the collective assumes."""
    f, c = C.c_uint32(), C.c_uint32()
    lib().nbody_shard_range(C.c_uint32(n), world, rank, C.byref(f), C.byref(c))
    return f.value, f.value + c.value


class Comm:
    """nbody_comm: the RCCL communicator of the per-step all-gather of positions (one process per GPU form)."""

    def __init__(self, world, rank, unique_id, device):
        self.h = C.c_void_p()
        assert len(unique_id) == COMM_ID_BYTES
        _check(lib().nbody_comm_create(C.byref(self.h), world, rank, C.c_char_p(bytes(unique_id)), device))
        self.world, self.rank = world, rank

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(lib().nbody_comm_get_unique_id(buf))
        return buf.raw

    @staticmethod
    def rccl_version():
        return int(lib().nbody_comm_rccl_version())

    def allgather_positions(self, st, stream=None):
        _check(lib().nbody_allgather_positions(self.h, C.byref(st), C.c_void_p(stream)))

    def close(self):
        if self.h:
            lib().nbody_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostSystem:
    """Host arrays of the reference's System<T,N> (src/system.h:13-19)."""

    def __init__(self, dtype, dim, n):
        t = np_dtype(dtype)
        self.dtype, self.dim, self.n = dtype, dim, n
        self.m = np.zeros(n, t)
        self.x = np.zeros((n, dim), t)
        self.v = np.zeros((n, dim), t)
        self.a = np.zeros((n, dim), t)
        self.ao = np.zeros((n, dim), t)
        self.dt = 0.0
        self.c = 0.0


def build_model(dtype, dim, workload, n):
    """Product workload generators (host/models.hpp, mirrors src/models.h). Returns a HostSystem."""
    wl = WORKLOADS[workload] if isinstance(workload, str) else workload
    s = HostSystem(dtype, dim, n)
    dt, c = C.c_double(), C.c_double()
    sz = host_lib().nbody_host_build_model(dtype, dim, wl, C.c_uint32(n), _p(s.m), _p(s.x), _p(s.v), C.byref(dt), C.byref(c))
    if sz < 0:
        raise NbodyError(f"nbody_host_build_model failed ({sz})")
    if sz != n:
        t = HostSystem(dtype, dim, int(sz))
        t.m[:], t.x[:], t.v[:] = s.m[:sz], s.x[:sz], s.v[:sz]
        s = t
    s.dt, s.c = dt.value, c.value
    return s


class Bvh:
    """bvh<T,N> (src/bvh.h:98-325) on the device."""

    def __init__(self, dtype, dim, n, device=-1):
        """device: where the tree's buffers live (-1: the calling thread's current device)."""
        self.h = C.c_void_p()
        self.dtype, self.dim, self.n = dtype, dim, n
        _check(lib().nbody_bvh_create_on(C.byref(self.h), dtype, dim, C.c_uint32(n), device))
        self.nnodes = int(lib().nbody_bvh_nnodes(self.h))

    def close(self):
        if self.h:
            lib().nbody_bvh_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bounding_box(self, st, stream=None):
        _check(lib().nbody_bvh_bounding_box(self.h, C.byref(st), C.c_void_p(stream)))

    def get_bounding_box(self, stream=None):
        t = np_dtype(self.dtype)
        lo, hi = np.zeros(self.dim, t), np.zeros(self.dim, t)
        _check(lib().nbody_bvh_get_bounding_box(self.h, _p(lo), _p(hi), C.c_void_p(stream)))
        return lo, hi

    def hilbert_sort(self, st, stream=None):
        _check(lib().nbody_bvh_hilbert_sort(self.h, C.byref(st), C.c_void_p(stream)))

    def build_tree(self, st, stream=None):
        _check(lib().nbody_bvh_build_tree(self.h, C.byref(st), C.c_void_p(stream)))

    def compute_force(self, st, theta, stream=None):
        _check(lib().nbody_bvh_compute_force(self.h, C.byref(st), C.c_double(theta), C.c_void_p(stream)))

    def set_traversal(self, mode):
        """0 auto, 1 per-lane walks, 2 wave-cooperative sweep (bitwise identical results)."""
        _check(lib().nbody_bvh_set_traversal(self.h, mode))

    def set_launch_order(self, mode):
        """Sweep: 0 work items (cut groups first), 1 one block per group in index order (bitwise identical results)."""
        _check(lib().nbody_bvh_set_launch_order(self.h, mode))

    def enable_counters(self, on=True):
        _check(lib().nbody_bvh_enable_counters(self.h, 1 if on else 0))

    def read(self, what, stream=None):
        t = np_dtype(self.dtype)
        shapes = {0: ((self.n,), np.uint64), 1: ((self.n,), np.uint32), 2: ((self.nnodes, self.dim + 1), t),
                  3: ((self.nnodes,), t), 4: ((self.nnodes, 2 * self.dim), t), 5: ((self.n, 4), np.uint32),
                  6: ((self.nnodes,), t)}
        shape, dt = shapes[what]
        out = np.zeros(shape, dt)
        _check(lib().nbody_bvh_read(self.h, what, _p(out), C.c_size_t(out.nbytes), C.c_void_p(stream)))
        return out


class Octree:
    """octree<T,N> (src/octree.h) on the device."""

    def __init__(self, dtype, dim, n, device=-1):
        """device: where the tree's buffers live (-1: the calling thread's current device)."""
        self.h = C.c_void_p()
        self.dtype, self.dim, self.n = dtype, dim, n
        _check(lib().nbody_octree_create_on(C.byref(self.h), dtype, dim, C.c_uint32(n), device))

    def set_build(self, mode):
        """0 auto (= 3), 1 one launch per tree level, 2 all levels behind a grid barrier, 3 one pass over the sorted keys, 4 per-level
        for the levels of the last info + one launch for the rest.  Same cells, monopoles, forces and counters bit for bit."""
        _check(lib().nbody_octree_set_build(self.h, mode))

    def set_step_budget(self, steps):
        """Visit rounds a body's walk may make before it is abandoned and info() raises (0 = the node pool size)."""
        _check(lib().nbody_octree_set_step_budget(self.h, C.c_uint32(steps)))

    def set_walk(self, mode):
        """0 auto, 1 the compiler-scheduled walk kernel, 2 the visit round as ISA (bitwise identical results)."""
        _check(lib().nbody_octree_set_walk(self.h, mode))

    def close(self):
        if self.h:
            lib().nbody_octree_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self, stream=None):
        _check(lib().nbody_octree_clear(self.h, C.c_void_p(stream)))

    def compute_bounds(self, st, stream=None):
        _check(lib().nbody_octree_compute_bounds(self.h, C.byref(st), C.c_void_p(stream)))

    def insert(self, st, stream=None):
        _check(lib().nbody_octree_insert(self.h, C.byref(st), C.c_void_p(stream)))

    def compute_tree(self, stream=None):
        _check(lib().nbody_octree_compute_tree(self.h, C.c_void_p(stream)))

    def compute_force(self, st, theta, stream=None):
        _check(lib().nbody_octree_compute_force(self.h, C.byref(st), C.c_double(theta), C.c_void_p(stream)))

    def info(self, stream=None):
        """(tree size = next_free_child_group, root mass); raises if the build hit the depth limit / node pool."""
        size = C.c_uint32()
        mass = np.zeros(1, np_dtype(self.dtype))
        _check(lib().nbody_octree_info(self.h, C.byref(size), _p(mass), C.c_void_p(stream)))
        return size.value, mass[0]

    def enable_counters(self, on=True):
        _check(lib().nbody_octree_enable_counters(self.h, 1 if on else 0))

    def read_counters(self, stream=None):
        out = np.zeros((self.n, 2), np.uint32)
        _check(lib().nbody_octree_read_counters(self.h, _p(out), C.c_size_t(out.nbytes), C.c_void_p(stream)))
        return out


class DeviceSystem:
    """Owning device mirror of a System<T,N>; phase methods mirror the calls of the reference drivers."""

    def __init__(self, dtype, dim, n, device=0):
        self.h = C.c_void_p()
        self.dtype, self.dim, self.n, self.device = dtype, dim, n, device
        _check(lib().nbody_create(C.byref(self.h), dtype, dim, C.c_uint32(n), device))
        self.stream = lib().nbody_ctx_stream(self.h)
        self._bvh = None
        self._octree = None

    @classmethod
    def from_host(cls, hs, device=0):
        d = cls(hs.dtype, hs.dim, hs.n, device)
        d.upload(hs)
        return d

    def close(self):
        if self._bvh is not None:
            self._bvh.close()
            self._bvh = None
        if self._octree is not None:
            self._octree.close()
            self._octree = None
        if self.h:
            lib().nbody_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, hs):
        for k in ("m", "x", "v", "a", "ao"):
            arr = getattr(hs, k)
            assert arr.flags["C_CONTIGUOUS"] and arr.dtype == np_dtype(self.dtype)
        _check(lib().nbody_upload(self.h, _p(hs.m), _p(hs.x), _p(hs.v), _p(hs.a), _p(hs.ao), C.c_double(hs.dt), C.c_double(hs.c)))

    def download(self, hs=None):
        hs = hs or HostSystem(self.dtype, self.dim, self.n)
        _check(lib().nbody_download(self.h, _p(hs.m), _p(hs.x), _p(hs.v), _p(hs.a), _p(hs.ao)))
        st = self.state()
        hs.dt, hs.c = st.dt, st.c
        return hs

    def state(self, first=0, count=None):
        """Device view; (first, count) selects a shard of targets (v/a/ao pointers are offset to it)."""
        st = nbody_state()
        _check(lib().nbody_ctx_state(self.h, C.byref(st)))
        if first or (count is not None and count != self.n):
            count = self.n - first if count is None else count
            esz = (4 if self.dtype == F32 else 8) * self.dim
            st.first, st.count = first, count
            st.v, st.a, st.ao = st.v + first * esz, st.a + first * esz, st.ao + first * esz
        return st

    def sync(self):
        _check(lib().nbody_stream_sync(C.c_void_p(self.stream)))

    def configure_all_pairs(self, split=0, targets_per_thread=0, source_path=0):
        """K1 launch shape of THIS context (nbody_ctx_configure_all_pairs); state() then carries it in `tuning`."""
        _check(lib().nbody_ctx_configure_all_pairs(self.h, split, targets_per_thread, source_path))

    def set_shard(self, first, count):
        """nbody_ctx_set_shard: this context owns targets [first, first+count) (multi-GPU all-pairs)."""
        _check(lib().nbody_ctx_set_shard(self.h, C.c_uint32(first), C.c_uint32(count)))

    # K1/K2/K3
    def all_pairs_force(self, first=0, count=None):
        st = self.state(first, count)
        _check(lib().nbody_all_pairs_force(C.byref(st), C.c_void_p(self.stream)))

    def all_pairs_collapsed_force(self):
        st = self.state()
        _check(lib().nbody_all_pairs_collapsed_force(C.byref(st), C.c_void_p(self.stream)))

    def accelerate_step(self, first=0, count=None):
        st = self.state(first, count)
        _check(lib().nbody_accelerate_step(C.byref(st), C.c_void_p(self.stream)))

    def calc_energies(self):
        """(kinetic, potential) as in System::calc_energies (src/system.h:62-79); blocking."""
        t = np_dtype(self.dtype)
        ke, pe = np.zeros(1, t), np.zeros(1, t)
        st = self.state()
        _check(lib().nbody_calc_energies(C.byref(st), _p(ke), _p(pe), C.c_void_p(self.stream)))
        return ke[0], pe[0]

    # K4..K9
    @property
    def bvh(self):
        if self._bvh is None:
            self._bvh = Bvh(self.dtype, self.dim, self.n, self.device)
        return self._bvh

    @property
    def octree(self):
        if self._octree is None:
            self._octree = Octree(self.dtype, self.dim, self.n, self.device)
        return self._octree

    def octree_force(self, theta):
        """One force phase of run_octree (src/octree.h:321-326)."""
        st, t = self.state(), self.octree
        t.clear(self.stream)
        t.compute_bounds(st, self.stream)
        t.insert(st, self.stream)
        t.compute_tree(self.stream)
        t.compute_force(st, theta, self.stream)

    def bvh_force(self, theta):
        """One force phase of run_bvh (src/bvh.h:382-393)."""
        st, b = self.state(), self.bvh
        b.bounding_box(st, self.stream)
        b.hilbert_sort(st, self.stream)
        b.build_tree(st, self.stream)
        b.compute_force(st, theta, self.stream)


def _load_sharded():
    import importlib.util
    import sys
    name = __name__ + ".sharded"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "sharded.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


parallel = _load_sharded()


class StepGraph:
    """Records the phase calls of one step on the device's stream (HIP stream capture) and replays them."""

    def __init__(self, dev, record):
        self.dev, self.h = dev, C.c_void_p()
        _check(lib().nbody_graph_begin(C.c_void_p(dev.stream)))
        try:
            record()
        finally:
            _check(lib().nbody_graph_end(C.c_void_p(dev.stream), C.byref(self.h)))

    def launch(self):
        _check(lib().nbody_graph_launch(self.h, C.c_void_p(self.dev.stream)))

    def close(self):
        if self.h:
            lib().nbody_graph_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def executed_steps(steps, csv_detailed, warmup=10):
    """Step-count semantics of the reference drivers."""
    return steps if csv_detailed else max(steps, warmup)


OCTREE_CHECK_EVERY = 64


def run(dev, algorithm, nsteps, theta=0.5):
    """The step loop of run_all_pairs / run_bvh / run_octree on the device.  Octree builds report trouble (bodies that
    never separate, node pool, walk stack or step budget exhausted) through a sticky device-side flag: it is read every
    OCTREE_CHECK_EVERY steps and after the last one, and raises — a flagged build drops mass, never integrate on."""
    for k in range(nsteps):
        if algorithm == "all-pairs":
            dev.all_pairs_force()
        elif algorithm == "all-pairs-collapsed":
            dev.all_pairs_collapsed_force()
        elif algorithm == "bvh":
            dev.bvh_force(theta)
        elif algorithm == "octree":
            dev.octree_force(theta)
            if (k + 1) % OCTREE_CHECK_EVERY == 0:
                dev.octree.info(dev.stream)
        else:
            raise ValueError(algorithm)
        dev.accelerate_step()
    if algorithm == "octree" and nsteps > 0:
        dev.octree.info(dev.stream)
    dev.sync()
