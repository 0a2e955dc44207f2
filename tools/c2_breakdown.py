"""Where config 2's time per step goes beyond one K1 launch on the initial state: K1 (with its helper launches) and K3 timed
separately with a stream sync around each, every 10th step of the 100, with the pair rule in force and the system's extent."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
n = int(os.environ.get("C2_N", "65536"))
wl = os.environ.get("C2_WL", "uniform")
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, wl, n))
dev.all_pairs_force(); dev.sync()
def t(fn, reps=1):
    dev.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    dev.sync(); return (time.perf_counter() - t0) / reps * 1e3
print(f"n = {n} {wl}: K1 on the initial state, 20 launches back to back: {t(dev.all_pairs_force, 20):.4f} ms each")
for step in range(1, 101):
    k1 = t(dev.all_pairs_force)
    k3 = t(dev.accelerate_step)
    if step in (1, 2, 5) or step % 10 == 0:
        sparse, vol = nb.all_pairs_pair_rule(dev.state(), dev.stream)
        x = dev.download().x
        print(f"step {step:3d}: K1 {k1:.4f} ms  K3 {k3:.4f} ms  rule {'sparse' if sparse else 'dense'}  extent {np.ptp(x, axis=0).round(2).tolist()}  "
              f"closest-pair proxy: min |x_i - x_(i+1)| = {np.linalg.norm(np.diff(x, axis=0), axis=1).min():.2e}")
print(f"K1 on the final state, 20 launches back to back: {t(dev.all_pairs_force, 20):.4f} ms each")
