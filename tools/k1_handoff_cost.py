"""Experiment (libnbody_hip_exp.so): what meeting K1's chunk sums costs — as shipped (launches of up to 2048 blocks collect the sums, larger
ones pass turns), with turns at every size (NBODY_K1_COLLECT=0: round 4's form), and with NBODY_K1_NO_HANDOFF=1 (no waiting, no
turn passed: the sums are WRONG, the time is what the kernel would take without any protocol)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
cases = [("f64 galaxy n=2048", nb.F64, "galaxy", 2048, 200), ("f64 galaxy n=4096", nb.F64, "galaxy", 4096, 200), ("f64 galaxy n=8192", nb.F64, "galaxy", 8192, 200), ("f64 galaxy n=16384", nb.F64, "galaxy", 16384, 200),
         ("f64 galaxy n=32768", nb.F64, "galaxy", 32768, 100), ("f32 galaxy n=8192", nb.F32, "galaxy", 8192, 200), ("f32 uniform 2D n=10000 (C1 size)", nb.F32, "uniform2", 10000, 200),
         ("f32 uniform n=262144", nb.F32, "uniform", 262144, 10), ("f32 galaxy n=262144", nb.F32, "galaxy", 262144, 10),
         ("f32 uniform n=100000", nb.F32, "uniform", 100000, 30), ("f64 uniform n=65536 (config 2)", nb.F64, "uniform", 65536, 60),
         ("f64 galaxy n=262144", nb.F64, "galaxy", 262144, 10), ("f64 galaxy n=2^20", nb.F64, "galaxy", 1 << 20, 2)]
for label, dtype, wl, n, reps in cases:
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 2 if wl.endswith("2") else 3, wl.rstrip("2"), n))
    row = []
    for rnd in range(2):
        for mode in ("shipped", "turns", "none"):
            os.environ.pop("NBODY_K1_NO_HANDOFF", None); os.environ.pop("NBODY_K1_COLLECT", None)
            if mode == "none":
                os.environ["NBODY_K1_NO_HANDOFF"] = "1"
            if mode == "turns":
                os.environ["NBODY_K1_COLLECT"] = "0"
            dev.all_pairs_force(); dev.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(reps):
                    dev.all_pairs_force()
                dev.sync()
                best = min(best, (time.perf_counter() - t0) / reps * 1e3)
            row.append(best)
    os.environ.pop("NBODY_K1_NO_HANDOFF", None); os.environ.pop("NBODY_K1_COLLECT", None)
    print(f"{label:32s} shipped {row[0]:.4f} / {row[3]:.4f} ms   turns at every size {row[1]:.4f} / {row[4]:.4f}   no protocol {row[2]:.4f} / {row[5]:.4f}   "
          f"(shipped over none {100 * (row[0] + row[3]) / (row[2] + row[5]) - 100:+.1f} %, turns over none {100 * (row[1] + row[4]) / (row[2] + row[5]) - 100:+.1f} %)", flush=True)
    dev.close()
