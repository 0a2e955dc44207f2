"""Config 2's 100 steps take 0.27 s: does the box reach its sustained clock within them?  A fresh process replays the recorded step
(K1 + K3, n = 65536 f64 uniform; dt = 0 so that the state — and with it the pair rule — does not change) 1200 times and prints
the time per step in windows of 20 steps, with the shader clock rocm-smi reports alongside (sampled in the background, ~4 per s)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
from conftest import load_package
nb = load_package()
import bench
n = 65536
hs = nb.build_model(nb.F64, 3, "uniform", n)
hs.dt = 0.0
dev = nb.DeviceSystem.from_host(hs)
dev.all_pairs_force(); dev.sync()          # scratch exists; nothing else has run on this GPU in this process
g = nb.StepGraph(dev, lambda: (dev.all_pairs_force(), dev.accelerate_step()))
tele = bench.Telemetry(0)
rows = []
with tele:
    t00 = time.perf_counter()
    for w in range(60):
        t0 = time.perf_counter()
        for _ in range(20):
            g.launch()
        dev.sync()
        rows.append((t0 - t00, (time.perf_counter() - t0) / 20 * 1e3, len(tele.samples)))
    total = time.perf_counter() - t00
clk = [c for c, _ in tele.samples]
print(f"n = {n} f64 uniform, recorded step (6 launches) replayed 1200 times in a fresh process, dt = 0 (dense rule throughout)")
print(f"{'steps':>11s} {'at s':>7s} {'ms per step':>12s}  sclk samples so far")
for w, (at, ms, k) in enumerate(rows):
    if w < 12 or w % 6 == 5:
        seen = clk[:k]
        print(f"{w * 20 + 1:5d}-{w * 20 + 20:5d} {at:7.3f} {ms:12.4f}  {[round(c) for c in seen[-3:]]}")
print(f"steps 1-100 (what the CLI's -s 100 times, 10 of them as warm-up): {sum(r[1] for r in rows[:5]) / 5:.4f} ms per step; "
      f"steps 601-1200: {sum(r[1] for r in rows[30:]) / 30:.4f} ms per step; rocm-smi sclk over the run: {round(min(clk)) if clk else None}-{round(max(clk)) if clk else None} MHz ({len(clk)} samples)")
