"""Is K1's time a function of the DATA at a fixed instruction stream?  The same launch (n = 65536, f64, dense rule, forced by keeping
the box volume below the sparse threshold) on: the initial uniform cube; the same cube scaled by 8 (same mantissas); the cube after
10 ballistic steps (x + v t: coordinates of mixed magnitudes); random positions in a flat slab; every case for ~1.5 s with the shader
clock and socket power sampled (rocm-smi) — a time that follows the clock is the power cap, not the kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from conftest import load_package
nb = load_package()
import bench
n = 65536
hs0 = nb.build_model(nb.F64, 3, "uniform", n)
rng = np.random.default_rng(1)
cases = [("initial uniform cube [-1,1]^3", hs0.x.copy()),
         ("the same cube x 8", hs0.x * 8.0),
         ("cube after 10 ballistic steps: x + v", hs0.x + hs0.v * 1.0),
         ("after 100 ballistic steps: x + 10 v (volume 8e3: still the dense rule)", hs0.x + hs0.v * 10.0),
         ("uniform in [-20,20]^2 x [-2,2] (volume 6.4e3)", rng.uniform(-1, 1, (n, 3)) * [20, 20, 2]),
         ("uniform in [-27,27]^3 (volume 1.6e5, just dense)", rng.uniform(-27, 27, (n, 3))),
         ("uniform in [-30,30]^3 (volume 2.2e5: the sparse rule)", rng.uniform(-30, 30, (n, 3))),
         ("uniform in [-300,300]^3 (sparse rule, no batch holds a close pair)", rng.uniform(-300, 300, (n, 3)))]
for label, x in cases:
    hs = nb.build_model(nb.F64, 3, "uniform", n)
    hs.x[:] = x
    dev = nb.DeviceSystem.from_host(hs)
    sparse, vol = nb.all_pairs_pair_rule(dev.state(), dev.stream)
    for _ in range(100):
        dev.all_pairs_force()
    dev.sync()
    tele = bench.Telemetry(0)
    with tele:
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < 1.5:
            for _ in range(50):
                dev.all_pairs_force()
            dev.sync(); k += 50
        ms = (time.perf_counter() - t0) / k * 1e3
    s = tele.summary() or {}
    clk = s.get("sclk_mhz_mean")
    print(f"{label}\n    rule {'sparse' if sparse else 'dense'} (volume {vol:.3g})  {ms:.4f} ms per launch  sclk {clk and round(clk)} MHz  "
          f"power {s.get('socket_power_w_mean') and round(s['socket_power_w_mean'])} W  ->  {ms * (clk or 0) / 2400:.4f} ms at 2400 MHz")
    dev.close()
