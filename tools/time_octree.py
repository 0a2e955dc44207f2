"""usage: time_octree.py [N] [float|double] [workload] [dim]
Diagnostic: per-phase time of the octree step (clear, bounds, insert, multipoles, force) and of the bvh step at the same N."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dtype = nb.F32 if (len(sys.argv) > 2 and sys.argv[2] == "float") else nb.F64
wl = sys.argv[3] if len(sys.argv) > 3 else "galaxy"
dim = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, wl, n))
st, t = dev.state(), dev.octree


def timed(name, fn, reps=5):
    fn(); dev.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    dev.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"n={n} {wl} dtype={dtype} dim={dim} {name}: {ms:.3f} ms", flush=True)
    return ms


tot = 0.0
tot += timed("octree clear", lambda: t.clear(dev.stream))
tot += timed("octree bounds", lambda: t.compute_bounds(st, dev.stream))
tot += timed("octree insert", lambda: t.insert(st, dev.stream))
tot += timed("octree multipoles", lambda: t.compute_tree(dev.stream))
tot += timed("octree force", lambda: t.compute_force(st, 0.5, dev.stream))
for form, label in ((1, "compiler-scheduled walk"), (2, "ISA visit round")):
    t.set_walk(form)
    timed(f"octree force, {label}", lambda: t.compute_force(st, 0.5, dev.stream), reps=20)
t.set_walk(0)
timed("octree force, 1/8 shard window", lambda: t.compute_force(dev.state(n // 2, n // 8), 0.5, dev.stream))
print(f"octree phases sum {tot:.3f} ms; tree info {t.info(dev.stream)}")
timed("octree whole step", lambda: nb.run(dev, "octree", 1, 0.5))
d2 = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, wl, n))
timed("bvh whole step", lambda: nb.run(d2, "bvh", 1, 0.5))
