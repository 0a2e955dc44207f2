"""Diagnostic: time K2 (all-pairs-collapsed) against K1 at a given size/dtype."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
for n, dtype, dim in ((262144, nb.F32, 3), (100000, nb.F64, 3), (10000, nb.F32, 2), (4096, nb.F64, 3)):
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, "uniform", n))
    for name, fn in (("collapsed", dev.all_pairs_collapsed_force), ("all-pairs", dev.all_pairs_force)):
        fn(); dev.sync()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dev.sync()
        print(f"n={n} dtype={dtype} dim={dim} {name:10s}: {(time.perf_counter()-t0)/reps*1e3:.3f} ms", flush=True)
