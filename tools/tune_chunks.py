"""Diagnostic: K1 time against the number of source chunks (grid.y) and targets per lane, to place ap_auto_chunks.
The NBODY_K1_CHUNKS override changes the rounding order; it exists for this experiment only.
usage: tune_chunks.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
sizes = [int(a) for a in sys.argv[1:]] or [65536, 100000, 262144]
for n in sizes:
    dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, os.environ.get("WL", "galaxy"), n))
    for count in (n,) if os.environ.get("FULL_ONLY") else (n, n // 8):
        for y in (1, 2, 4, 8, 16, 32, 64):
            for r in (1, 2):
                os.environ["NBODY_K1_CHUNKS"] = str(y)
                nb.configure_all_pairs(0, r)
                dev.all_pairs_force(0, count); dev.sync()
                reps = 3 if n > 300000 and count == n else 10
                t0 = time.perf_counter()
                for _ in range(reps):
                    dev.all_pairs_force(0, count)
                dev.sync()
                t = (time.perf_counter() - t0) / reps
                tf = 20.0 * count * (n - 1) / t / 1e12
                print(f"n={n} targets={count} chunks={y:2d} tpt={r}: {t*1e3:9.3f} ms {100*tf/78.6:5.1f}%  {nb.describe_all_pairs(dev.state(0, count))}", flush=True)
    dev.close()
os.environ.pop("NBODY_K1_CHUNKS", None)
nb.configure_all_pairs(0, 0)
