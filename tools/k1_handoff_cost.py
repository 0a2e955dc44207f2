"""Experiment (libnbody_hip_exp.so): what K1's chunk hand-off costs — the same launches with NBODY_K1_NO_HANDOFF=1 (no waiting, no
turn passed: the sums are WRONG, the time is what the kernel would take without the protocol)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
cases = [("f32 uniform n=262144", nb.F32, "uniform", 262144, 10), ("f32 galaxy n=262144", nb.F32, "galaxy", 262144, 10),
         ("f32 uniform n=100000", nb.F32, "uniform", 100000, 30), ("f64 uniform n=65536 (config 2)", nb.F64, "uniform", 65536, 60),
         ("f64 galaxy n=262144", nb.F64, "galaxy", 262144, 10), ("f64 galaxy n=2^20", nb.F64, "galaxy", 1 << 20, 2)]
for label, dtype, wl, n, reps in cases:
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, wl, n))
    row = []
    for rnd in range(2):
        for off in (None, "1"):
            os.environ.pop("NBODY_K1_NO_HANDOFF", None)
            if off:
                os.environ["NBODY_K1_NO_HANDOFF"] = off
            dev.all_pairs_force(); dev.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(reps):
                    dev.all_pairs_force()
                dev.sync()
                best = min(best, (time.perf_counter() - t0) / reps * 1e3)
            row.append(best)
    os.environ.pop("NBODY_K1_NO_HANDOFF", None)
    print(f"{label:32s} with the hand-off {row[0]:.3f} / {row[2]:.3f} ms   without {row[1]:.3f} / {row[3]:.3f} ms   ({100 * (row[0] + row[2]) / (row[1] + row[3]) - 100:+.2f} % for the protocol)", flush=True)
    dev.close()
