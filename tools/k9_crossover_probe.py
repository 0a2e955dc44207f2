"""Diagnostic: along a run, time K9's per-lane and sweep forms on the SAME evolved state and print the walk statistics
(node tests + body terms per body) next to them.  Usage: k9_crossover_probe.py [n] [steps between probes] [probes]"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 200
probes = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "galaxy", n))
t = dev.bvh
done = 0
for p in range(probes):
    st = dev.state()
    t.bounding_box(st, dev.stream); t.hilbert_sort(st, dev.stream); t.build_tree(st, dev.stream); dev.sync()
    lo, hi = t.get_bounding_box(dev.stream)
    res = {}
    for mode in (1, 5):
        t.set_traversal(mode)
        t.enable_counters(False)
        t.compute_force(st, 0.5, dev.stream); dev.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            t.compute_force(st, 0.5, dev.stream)
        dev.sync()
        res[mode] = (time.perf_counter() - t0) / 5 * 1e3
    t.enable_counters(True); t.compute_force(st, 0.5, dev.stream); dev.sync()
    c = t.read(5, dev.stream).astype(np.float64)
    t.enable_counters(False)
    walk = c[:, 0] + 2 * c[:, 1]
    print(f"n={n} step={done}: per-lane {res[1]:.3f} ms, sweep {res[5]:.3f} ms; walk entries/body mean {walk.mean():.0f} p99 {np.percentile(walk, 99):.0f} max {walk.max():.0f}; "
          f"box diag {np.linalg.norm(hi - lo):.0f}", flush=True)
    t.set_traversal(0)
    nb.run(dev, "bvh", every, 0.5)
    done += every
