"""Times every BASELINE.json config (and the reference's own benchmark matrix sizes) on one MI355X via the C ABI.
Writes one JSON document to stdout.  Diagnostic tool (the headline metric is bench.py's)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()


def timed(dev, fn, reps, warm=1, warm_s=0.0):
    """ms per call: `warm` calls first, then fn for `warm_s` seconds (up to the clock the box sustains: a process's first case
    otherwise reads up to 12 % slow), then the best of three windows of `reps` calls when warmed by time, one window otherwise
    (a tree step integrates the system: hundreds of warm-up steps would time an evolved galaxy, not the initial one)."""
    for _ in range(warm):
        fn()
    dev.sync()
    if warm_s > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < warm_s:
            fn()
            dev.sync()
    best = 1e9
    for _ in range(3 if warm_s > 0 else 1):
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dev.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


def all_pairs(name, dtype, dim, wl, n, reps, collapsed=False, first=0, count=None):
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, wl, n))
    n = dev.n
    cnt = n if count is None else count
    def step():
        if collapsed:
            dev.all_pairs_collapsed_force()
        else:
            dev.all_pairs_force(first, cnt)
        dev.accelerate_step(first, cnt)
    t = timed(dev, step, reps, warm_s=0.2)
    flop = 20.0 if dim == 3 else 14.0
    peak = 157.3 if dtype == nb.F32 else 78.6
    tf = flop * cnt * (n - 1) / t / 1e12
    dev.close()
    return {"config": name, "n": n, "ms_per_step": t * 1e3, "body_steps_per_s": cnt / t, "pairs_per_s": cnt * (n - 1) / t,
            "algorithmic_tflops": tf, "pct_vector_peak": 100 * tf / peak}


def bvh(name, dtype, dim, wl, n, theta, reps):
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, wl, n))
    st, t = dev.state(), dev.bvh
    phases = {}
    def run_phase(key, fn):
        phases[key] = timed(dev, fn, reps, warm=0) * 1e3
    # warm
    dev.bvh_force(theta); dev.accelerate_step(); dev.sync()
    run_phase("bbox_ms", lambda: t.bounding_box(st, dev.stream))
    run_phase("sort_ms", lambda: t.hilbert_sort(st, dev.stream))
    run_phase("build_ms", lambda: t.build_tree(st, dev.stream))
    run_phase("traversal_ms", lambda: t.compute_force(st, theta, dev.stream))
    run_phase("accel_ms", lambda: dev.accelerate_step())
    g = nb.StepGraph(dev, lambda: (dev.bvh_force(theta), dev.accelerate_step()))
    tt = timed(dev, g.launch, reps)
    g.close(); n = dev.n; dev.close()
    return {"config": name, "n": n, "theta": theta, "ms_per_step": tt * 1e3, "body_steps_per_s": n / tt, **phases}


def octree(name, dtype, dim, wl, n, theta, reps):
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, wl, n))
    st, t = dev.state(), dev.octree
    phases = {}
    def run_phase(key, fn):
        phases[key] = timed(dev, fn, reps, warm=0) * 1e3
    dev.octree_force(theta); dev.accelerate_step(); dev.sync()
    run_phase("bounds_ms", lambda: t.compute_bounds(st, dev.stream))
    run_phase("insert_ms", lambda: t.insert(st, dev.stream))
    run_phase("multipoles_ms", lambda: t.compute_tree(dev.stream))
    run_phase("force_ms", lambda: t.compute_force(st, theta, dev.stream))
    run_phase("accel_ms", lambda: dev.accelerate_step())
    size, _ = t.info(dev.stream)
    g = nb.StepGraph(dev, lambda: (dev.octree_force(theta), dev.accelerate_step()))
    tt = timed(dev, g.launch, reps)
    g.close(); n = dev.n; dev.close()
    return {"config": name, "n": n, "theta": theta, "ms_per_step": tt * 1e3, "body_steps_per_s": n / tt, "tree_size": int(size), **phases}


out = []
out.append(all_pairs("C2 all-pairs 3D double n=65536 (uniform, the default workload)", nb.F64, 3, "uniform", 65536, 20))
out.append(all_pairs("C3 all-pairs-collapsed 3D float n=262144 (uniform)", nb.F32, 3, "uniform", 262144, 5, collapsed=True))
out.append(all_pairs("   all-pairs 3D float n=262144 (uniform), for comparison", nb.F32, 3, "uniform", 262144, 5))
out.append(bvh("C4 bvh 3D double n=1e6 galaxy theta=0.5", nb.F64, 3, "galaxy", 1000000, 0.5, 5))
out.append(all_pairs("C5 all-pairs 3D double n=2^20 galaxy, 1 GPU", nb.F64, 3, "galaxy", 1 << 20, 2))
out.append(all_pairs("C5 per-rank work at 8 GPUs: 131072-target shard of n=2^20", nb.F64, 3, "galaxy", 1 << 20, 4, count=(1 << 20) // 8))
out.append(all_pairs("C1-size all-pairs 2D float n=10000 (uniform)", nb.F32, 2, "uniform", 10000, 200))
out.append(all_pairs("ref matrix: all-pairs 3D double n=100000 galaxy", nb.F64, 3, "galaxy", 100000, 10))
out.append(all_pairs("ref matrix: all-pairs-collapsed 3D double n=100000 galaxy", nb.F64, 3, "galaxy", 100000, 10, collapsed=True))
out.append(bvh("ref matrix: bvh 3D double n=100000 galaxy theta=0.5", nb.F64, 3, "galaxy", 100000, 0.5, 20))
out.append(octree("ref matrix: octree 3D double n=100000 galaxy theta=0.5", nb.F64, 3, "galaxy", 100000, 0.5, 20))
out.append(octree("ref matrix: octree 3D double n=1e6 galaxy theta=0.5", nb.F64, 3, "galaxy", 1000000, 0.5, 5))
out.append(octree("octree 3D float n=1e6 galaxy theta=0.5", nb.F32, 3, "galaxy", 1000000, 0.5, 5))
out.append(bvh("bvh 3D float n=1e6 galaxy theta=0.5", nb.F32, 3, "galaxy", 1000000, 0.5, 5))
print(json.dumps({"device": nb.device_info()[0], "results": out}, indent=1))
