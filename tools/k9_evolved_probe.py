import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
import numpy as np
nb = load_package()
n = 1000000
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "galaxy", n))
t = dev.bvh
nb.run(dev, "bvh", 1500, 0.5)
st = dev.state()
t.bounding_box(st, dev.stream); t.hilbert_sort(st, dev.stream); t.build_tree(st, dev.stream); dev.sync()
def timeit(first, count, reps=3):
    w = dev.state(first, count)
    t.compute_force(w, 0.5, dev.stream); dev.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        t.compute_force(w, 0.5, dev.stream)
    dev.sync()
    return (time.perf_counter() - t0) / reps * 1e3
for mode in (5, 1):
    t.set_traversal(mode)
    print("mode", mode, "whole %.2f ms" % timeit(0, n))
    if mode == 5:
        for parts in (2, 8, 32):
            ts = [timeit(n * k // parts, n // parts) for k in range(parts)]
            print(parts, "parts: sum %.2f ms, max %.2f min %.2f" % (sum(ts), max(ts), min(ts)))
        os.environ["NBODY_K9_ORDER"] = "0"
        print("index order whole %.2f ms" % timeit(0, n))
        os.environ.pop("NBODY_K9_ORDER")
t.set_traversal(5); t.enable_counters(True); t.compute_force(st, 0.5, dev.stream); dev.sync()
c = t.read(5, dev.stream).astype(np.float64)
walk = c[:, 0] + 2 * c[:, 1]
print("walk entries/body mean %.0f p50 %.0f p99 %.0f max %.0f" % (walk.mean(), np.median(walk), np.percentile(walk, 99), walk.max()))
# union length proxy per group: max walk in group
g = walk[: (n // 64) * 64].reshape(-1, 64)
print("per-group max walk: mean %.0f p99 %.0f max %.0f" % (g.max(1).mean(), np.percentile(g.max(1), 99), g.max(1).max()))
