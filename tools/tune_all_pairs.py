"""Diagnostic: time K1 (3D double all-pairs, galaxy) over (split, targets_per_thread) configs, full system and 1/8 shard."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
hs = nb.build_model(nb.F64, 3, "galaxy", n)
dev = nb.DeviceSystem.from_host(hs)
for (label, first, count) in (("full", 0, n), ("1/8 shard", 0, n // 8)):
    for js in (1, 2, 4):
        for r in (1, 2):
            nb.configure_all_pairs(js, r)
            dev.all_pairs_force(first, count); dev.sync()
            t0 = time.perf_counter()
            reps = 2 if count == n else 4
            for _ in range(reps):
                dev.all_pairs_force(first, count)
            dev.sync()
            t = (time.perf_counter() - t0) / reps
            tf = 20.0 * count * (n - 1) / t / 1e12
            print(f"n={n} {label:10s} split={js} tpt={r}: {t*1e3:9.2f} ms  {tf:6.2f} TFLOP/s ({100*tf/78.6:5.1f}% of FP64 vector peak)", flush=True)
