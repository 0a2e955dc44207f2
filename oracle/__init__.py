"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes front-end for the CPU oracle (oracle/liboracle.so, built by oracle/Makefile) and a runner for
the real reference binaries under oracle/_ref/ (built from /root/reference by `make -C oracle ref`).

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  Nothing under
stdpar-nbody_amd/ imports this package.
"""
import ctypes as C
import os
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")

F32, F64 = 0, 1
UNIFORM, PLUMMER, GALAXY = 0, 1, 2
WORKLOADS = {"uniform": UNIFORM, "plummer": PLUMMER, "galaxy": GALAXY}


def np_dtype(dtype):
    return np.float32 if dtype == F32 else np.float64


def build(ref=True):
    """Compile the oracle (and oracle/_ref when /root/reference is present). Building != using."""
    targets = ["oracle"] + (["ref"] if ref else [])
    subprocess.check_call(["make", "-s", "-C", HERE] + targets)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_build_model.restype = C.c_int64
        _lib.oracle_hilbert_cell.restype = C.c_uint64
        _lib.oracle_interleave_bits.restype = C.c_uint64
        _lib.oracle_bvh_nlevels.restype = C.c_uint32
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class State:
    """Host-side mirror of the reference's System<T,N> arrays (system.h:13-19): m[N], x/v/a/ao[N][D]."""

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

    def copy(self):
        s = State(self.dtype, self.dim, self.n)
        for k in ("m", "x", "v", "a", "ao"):
            setattr(s, k, getattr(self, k).copy())
        s.dt, s.c = self.dt, self.c
        return s


def build_model(dtype, dim, workload, n):
    """models.h generators; returns a State (galaxy may shrink n to 2*(n/2.0) truncated)."""
    wl = WORKLOADS[workload] if isinstance(workload, str) else workload
    s = State(dtype, dim, n)
    dt, c = C.c_double(), C.c_double()
    sz = lib().oracle_build_model(dtype, dim, wl, C.c_uint32(n), _p(s.m), _p(s.x), _p(s.v), C.byref(dt), C.byref(c))
    if sz < 0:
        raise RuntimeError(f"oracle_build_model failed: {sz}")
    if sz != n:
        t = State(dtype, dim, int(sz))
        t.m[:], t.x[:], t.v[:] = s.m[:sz], s.x[:sz], s.v[:sz]
        s = t
    s.dt, s.c = dt.value, c.value
    return s


def all_pairs_force(s, first=0, count=None):
    count = s.n - first if count is None else count
    r = lib().oracle_all_pairs_force(s.dtype, s.dim, _p(s.m), _p(s.x), _p(s.a), C.c_double(s.c), C.c_uint32(s.n),
                                     C.c_uint32(first), C.c_uint32(count))
    assert r == 0


def all_pairs_collapsed_force(s, mode=1):
    r = lib().oracle_all_pairs_collapsed_force(s.dtype, s.dim, _p(s.m), _p(s.x), _p(s.a), _p(s.ao), C.c_double(s.c),
                                               C.c_uint32(s.n), mode)
    assert r == 0


def accelerate_step(s):
    r = lib().oracle_accelerate_step(s.dtype, s.dim, _p(s.x), _p(s.v), _p(s.a), _p(s.ao), C.c_double(s.dt), C.c_uint32(s.n))
    assert r == 0


def calc_energies(s):
    t = np_dtype(s.dtype)
    ke, pe = np.zeros(1, t), np.zeros(1, t)
    r = lib().oracle_calc_energies(s.dtype, s.dim, _p(s.m), _p(s.x), _p(s.v), C.c_double(s.c), C.c_uint32(s.n), _p(ke), _p(pe))
    assert r == 0
    return ke[0], pe[0]


def calc_energies_wide(s):
    """calc_energies with the same terms (formed in T) summed in double: the yardstick for float systems."""
    ke, pe = C.c_double(), C.c_double()
    r = lib().oracle_calc_energies_wide(s.dtype, s.dim, _p(s.m), _p(s.x), _p(s.v), C.c_double(s.c), C.c_uint32(s.n), C.byref(ke),
                                        C.byref(pe))
    assert r == 0
    return ke.value, pe.value


def bounding_box(s):
    t = np_dtype(s.dtype)
    lo, hi = np.zeros(s.dim, t), np.zeros(s.dim, t)
    r = lib().oracle_bounding_box(s.dtype, s.dim, _p(s.x), C.c_uint32(s.n), _p(lo), _p(hi))
    assert r == 0
    return lo, hi


def hilbert_keys(s, lo, hi):
    keys = np.zeros(s.n, np.uint64)
    r = lib().oracle_hilbert_keys(s.dtype, s.dim, _p(s.x), C.c_uint32(s.n), _p(lo), _p(hi), _p(keys))
    assert r == 0
    return keys


def hilbert_cell(dim, cell):
    c = np.asarray(cell, np.uint32)
    return int(lib().oracle_hilbert_cell(dim, _p(c)))


def interleave_bits(dim, cell):
    c = np.asarray(cell, np.uint32)
    return int(lib().oracle_interleave_bits(dim, _p(c)))


def sort_keys(keys):
    perm = np.zeros(len(keys), np.uint32)
    lib().oracle_sort_keys(_p(keys), C.c_uint32(len(keys)), _p(perm))
    return perm


def apply_perm(s, perm):
    r = lib().oracle_apply_perm(s.dtype, s.dim, _p(s.m), _p(s.x), _p(s.v), _p(s.a), _p(s.ao), C.c_uint32(s.n), _p(perm))
    assert r == 0


def bvh_nlevels(n):
    return int(lib().oracle_bvh_nlevels(C.c_uint32(n)))


class Tree:
    def __init__(self, s):
        t = np_dtype(s.dtype)
        self.nlevels = bvh_nlevels(s.n)
        self.nnodes = (1 << self.nlevels) - 1
        self.nm = np.zeros((self.nnodes, s.dim + 1), t)   # monopole: x..., mass (monopole.h:7-23)
        self.nb = np.zeros((self.nnodes, 2 * s.dim), t)   # aabb: xmin, xmax
        self.nbw = np.zeros(self.nnodes, t)               # node width


def bvh_build(s):
    tr = Tree(s)
    r = lib().oracle_bvh_build(s.dtype, s.dim, _p(s.m), _p(s.x), C.c_uint32(s.n), _p(tr.nm), _p(tr.nb), _p(tr.nbw))
    assert r == 0
    return tr


def bvh_force(s, tr, theta, want_counts=False):
    counts = np.zeros((s.n, 4), np.uint32) if want_counts else None
    r = lib().oracle_bvh_force(s.dtype, s.dim, _p(s.m), _p(s.x), _p(s.a), C.c_double(s.c), C.c_uint32(s.n), C.c_double(theta),
                               _p(tr.nm), _p(tr.nbw), _p(counts))
    assert r == 0
    return counts


def can_approximate(dtype, bw, theta, d2):
    """bvh.h:246-248 on arrays: bw * bw < theta^2 * d2 in T (theta^2 formed in T, bvh.h:252)."""
    t = np_dtype(dtype)
    bw, d2 = np.ascontiguousarray(bw, t), np.ascontiguousarray(d2, t)
    assert bw.shape == d2.shape
    out = np.zeros(bw.shape, np.uint8)
    r = lib().oracle_can_approximate(dtype, _p(bw), C.c_double(theta), _p(d2), C.c_uint64(bw.size), _p(out))
    assert r == 0
    return out.astype(bool)


def bvh_step_force(s, theta):
    """bbox -> keys -> sort (permutes the state in place) -> build -> traversal (bvh.h:382-393)."""
    r = lib().oracle_bvh_step_force(s.dtype, s.dim, _p(s.m), _p(s.x), _p(s.v), _p(s.a), _p(s.ao), C.c_double(s.c),
                                    C.c_uint32(s.n), C.c_double(theta))
    assert r == 0


def octree_step_force(s, theta, want_counts=False):
    """clear, bounds, insert, multipoles, force (octree.h:321-326). Returns (counts or None, tree_size, root_mass)."""
    t = np_dtype(s.dtype)
    counts = np.zeros((s.n, 2), np.uint32) if want_counts else None
    size = C.c_uint32()
    root_mass = np.zeros(1, t)
    r = lib().oracle_octree_step_force(s.dtype, s.dim, _p(s.m), _p(s.x), _p(s.a), C.c_double(s.c), C.c_uint32(s.n),
                                       C.c_double(theta), _p(counts), C.byref(size), _p(root_mass))
    if r != 0:
        raise RuntimeError(f"oracle_octree_step_force failed ({r})")
    return counts, size.value, root_mass[0]


def executed_steps(steps, csv_detailed, warmup=10):
    """default mode runs max(steps, warmup) steps; --csv-detailed runs exactly `steps`
    (all_pairs.h:72-97, bvh.h:356-403, arguments.h:26)."""
    return steps if csv_detailed else max(steps, warmup)


def run(s, algorithm, nsteps, theta=0.5, collapsed_mode=1, frames=None):
    """Step loop of run_all_pairs / run_bvh (force phase then accelerate_step). Appends x after each
    step to `frames` when given (mirrors Saver::save_all in --csv-detailed mode)."""
    for _ in range(nsteps):
        if algorithm == "all-pairs":
            all_pairs_force(s)
        elif algorithm == "all-pairs-collapsed":
            all_pairs_collapsed_force(s, collapsed_mode)
        elif algorithm == "bvh":
            bvh_step_force(s, theta)
        elif algorithm == "octree":
            octree_step_force(s, theta)
        else:
            raise ValueError(algorithm)
        accelerate_step(s)
        if frames is not None:
            frames.append(s.x.copy())
    return s


# ---- the real reference (oracle/_ref) ------------------------------------------------------------

def ref_binary(dim):
    p = os.path.join(REF_DIR, f"nbody_ref_d{dim}")
    return p if os.path.exists(p) else None


def ref_run(dim, args, cwd=None, timeout=600):
    """Run the compiled reference with CLI `args`; returns stdout text."""
    exe = ref_binary(dim)
    if exe is None:
        raise FileNotFoundError("oracle/_ref not built (make -C oracle ref)")
    return subprocess.run([exe] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True, timeout=timeout, check=True).stdout


def read_positions_bin(path):
    """positions.bin: u32 nbodies, u32 steps, u32 sizeof(T), u32 dim, then frames of x (saving.h:85-114)."""
    raw = open(path, "rb").read()
    n, steps, tsz, dim = struct.unpack("<4I", raw[:16])
    t = np.float32 if tsz == 4 else np.float64
    data = np.frombuffer(raw[16:], dtype=t)
    nframes = data.size // (n * dim)
    return data[: nframes * n * dim].reshape(nframes, n, dim), steps


def read_energy_bin(path):
    raw = open(path, "rb").read()
    steps, tsz = struct.unpack("<2I", raw[:8])
    t = np.float32 if tsz == 4 else np.float64
    return np.frombuffer(raw[8:], dtype=t).reshape(-1, 2), steps


def ref_positions(dim, precision, algorithm, workload, n, steps, theta=None, energy=False):
    """Run the reference with --save pos --csv-detailed and return the (steps+1, n, dim) frames."""
    with tempfile.TemporaryDirectory() as d:
        args = ["-n", n, "-s", steps, "--precision", precision, "--algorithm", algorithm, "--workload", workload,
                "--save", "all" if energy else "pos", "--csv-detailed"]
        if theta is not None:
            args += ["--theta", theta]
        ref_run(dim, args, cwd=d)
        frames, _ = read_positions_bin(os.path.join(d, "positions.bin"))
        if energy:
            en, _ = read_energy_bin(os.path.join(d, "energy.bin"))
            return frames.copy(), en.copy()
        return frames.copy()


def parse_print_state(text):
    """Split `--print-state` stdout into (starting rows, final rows); `Total time` lines dropped."""
    start, final, cur = [], [], None
    for line in text.splitlines():
        if line.startswith("Starting state:"):
            cur = start
        elif line.startswith("Final state:"):
            cur = final
        elif line.startswith(("Starting simulation", "Done simulation", "Total time")):
            cur = None if line.startswith("Starting simulation") else cur
        elif cur is not None and ": m=" in line:
            cur.append(line)
    return start, final


def format_state_rows(s):
    """System::print (system.h:90-97): '{:02}: m={: .3e}, p=({: .3e}, {: .3e}), v=(...), f=(...)' — components 0,1 only."""
    rows = []
    for i in range(s.n):
        rows.append("%02d: m=% .3e, p=(% .3e, % .3e), v=(% .3e, % .3e), f=(% .3e, % .3e)" % (
            i, s.m[i], s.x[i, 0], s.x[i, 1], s.v[i, 0], s.v[i, 1], s.a[i, 0], s.a[i, 1]))
    return rows
