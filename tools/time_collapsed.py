"""Diagnostic: time K2 (all-pairs-collapsed) against K1 at a given size/dtype, for both reduction group sizes
(NBODY_K2_NT=8|16 is an experiment knob: targets reduced together by the transposed wavefront reduction)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
for n, dtype, dim in ((262144, nb.F32, 3), (100000, nb.F64, 3), (100000, nb.F32, 3), (10000, nb.F32, 2), (4096, nb.F64, 3)):
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, "uniform", n))
    for name, fn, env in (("collapsed NT=8", dev.all_pairs_collapsed_force, "8"), ("collapsed NT=16", dev.all_pairs_collapsed_force, "16"),
                          ("all-pairs", dev.all_pairs_force, None)):
        if env:
            os.environ["NBODY_K2_NT"] = env
        fn(); dev.sync()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dev.sync()
        t = (time.perf_counter() - t0) / reps
        flop = 20.0 if dim == 3 else 14.0
        peak = 157.3 if dtype == nb.F32 else 78.6
        print(f"n={n} dtype={dtype} dim={dim} {name:16s}: {t*1e3:.3f} ms  {100 * flop * n * (n - 1) / t / 1e12 / peak:.1f}% of vector peak", flush=True)
    os.environ.pop("NBODY_K2_NT", None)
