"""Diagnostic: time K2 (all-pairs-collapsed) against K1 at a given size/dtype over its register-budget variants
(NBODY_K2_CFG is an experiment knob: (targets reduced together, source records in registers, pair chains in flight))."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
CFG = {20: "streamed, hand pair, KS 8 (f32 3D)", 21: "streamed, hand pair, KS 4 (f32 3D)", 22: "streamed, compiled pair, KS 8", 23: "streamed, compiled pair, KS 4", 24: "streamed, compiled pair, KS 2", 0: "(16,8,hand) f32 3D / (16,8,1)", 8: "(16,8,1)", 10: "(16,4,hand)", 11: "(8,8,hand)", 12: "(8,4,hand)", 1: "(8,4,4)", 2: "(8,4,1)", 3: "(16,8,4)", 4: "(8,4,2)", 5: "(8,8,1)", 6: "(8,2,1)", 7: "(8,2,2)"}
SIZES = ((262144, nb.F32, 3), (100000, nb.F64, 3), (100000, nb.F32, 3), (10000, nb.F32, 2))
for n, dtype, dim in SIZES[:int(os.environ.get("K2_SIZES", "4"))]:
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, "uniform", n))
    cfgs = [int(c) for c in os.environ["K2_CFGS"].split(",")] if "K2_CFGS" in os.environ else sorted(CFG)
    runs = [("collapsed cfg %d %s" % (c, CFG[c]), dev.all_pairs_collapsed_force, str(c)) for c in cfgs] + [("all-pairs", dev.all_pairs_force, None)]
    for name, fn, env in runs:
        if env is not None:
            os.environ["NBODY_K2_CFG"] = env
        fn(); dev.sync()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        dev.sync()
        t = (time.perf_counter() - t0) / reps
        flop = 20.0 if dim == 3 else 14.0
        peak = 157.3 if dtype == nb.F32 else 78.6
        print(f"n={n} dtype={dtype} dim={dim} {name:26s}: {t*1e3:.3f} ms  {100 * flop * n * (n - 1) / t / 1e12 / peak:.1f}% of vector peak", flush=True)
    os.environ.pop("NBODY_K2_CFG", None)
