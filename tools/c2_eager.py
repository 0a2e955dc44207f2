"""BASELINE config 2 (all-pairs 3D double n = 65536, uniform) as a plain loop of asynchronous calls through the C ABI — what
tools/measure_configs.py times — or as the recorded step replayed (C2_MODE=graph), for tools/config2_gaps.sh's kernel traces."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
n, steps = int(os.environ.get("C2_N", "65536")), int(os.environ.get("C2_STEPS", "110"))
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "uniform", n))
one = lambda: (dev.all_pairs_force(), dev.accelerate_step())
one(); dev.sync()
if os.environ.get("C2_MODE") == "graph":
    g = nb.StepGraph(dev, one)
    one = g.launch
for _ in range(10):
    one()
dev.sync()
t0 = time.perf_counter()
for _ in range(steps - 10):
    one()
dev.sync()
print(f"{os.environ.get('C2_MODE', 'eager')}: {(time.perf_counter() - t0) / (steps - 10) * 1e3:.4f} ms per step (host clock, n = {n})")
