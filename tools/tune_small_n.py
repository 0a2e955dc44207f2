"""Diagnostic: K1 below 65 536 bodies — LDS-tile form (auto) against the scalar-stream form with 8 slices and source chunks.
usage: tune_small_n.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
sizes = [int(a) for a in sys.argv[1:]] or [4096, 10000, 20000, 30000, 50000]
for n in sizes:
    dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "uniform", n))
    def t(label):
        dev.all_pairs_force(); dev.sync()
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            dev.all_pairs_force()
        dev.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f"n={n} {label:34s}: {dt*1e3:8.4f} ms {100 * 20.0 * n * (n - 1) / dt / 1e12 / 78.6:5.1f}%  {nb.describe_all_pairs(dev.state())}", flush=True)
    os.environ.pop("NBODY_K1_CHUNKS", None)
    nb.configure_all_pairs(0, 0, source_path=0); t("auto")
    for split in (4, 8):
        for y in (1, 2, 4, 8, 16):
            for r in (1, 2):
                os.environ["NBODY_K1_CHUNKS"] = str(y)
                nb.configure_all_pairs(split, r, source_path=2); t(f"sgpr split={split} chunks={y} tpt={r}")
    os.environ.pop("NBODY_K1_CHUNKS", None)
    nb.configure_all_pairs(0, 0, source_path=0)
    dev.close()
