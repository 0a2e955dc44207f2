"""Experiment (libnbody_hip_exp.so): K1 with ONE rule per kernel instantiation, forced from the host (NBODY_K1_RULE_FORCE=1 dense,
2 sparse), against the shipped kernel that holds both copies of the loop and branches on the device's rule.  Timing only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
cases = [("f64 galaxy n=2^20 (headline)", nb.F64, "galaxy", 1 << 20, 3, 2),
         ("f64 uniform n=65536 (config 2)", nb.F64, "uniform", 65536, 60, 1),
         ("f64 galaxy n=262144", nb.F64, "galaxy", 262144, 10, 2),
         ("f32 uniform n=262144", nb.F32, "uniform", 262144, 10, 1),
         ("f32 galaxy n=262144", nb.F32, "galaxy", 262144, 10, 2)]
for label, dtype, wl, n, reps, right in cases:
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, wl, n))
    sparse, _ = nb.all_pairs_pair_rule(dev.state(), dev.stream)
    assert (2 if sparse else 1) == right, (label, sparse)
    row = []
    for rnd in range(2):
        for force in (None, str(right)):
            os.environ.pop("NBODY_K1_RULE_FORCE", None)
            if force:
                os.environ["NBODY_K1_RULE_FORCE"] = force
            dev.all_pairs_force(); dev.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(reps):
                    dev.all_pairs_force()
                dev.sync()
                best = min(best, (time.perf_counter() - t0) / reps * 1e3)
            row.append(best)
    os.environ.pop("NBODY_K1_RULE_FORCE", None)
    print(f"{label:34s} rule {'sparse' if sparse else 'dense '}: both copies {row[0]:.3f} / {row[2]:.3f} ms   one rule per kernel {row[1]:.3f} / {row[3]:.3f} ms   ({100 * (row[1] + row[3]) / (row[0] + row[2]) - 100:+.2f} %)", flush=True)
    dev.close()
