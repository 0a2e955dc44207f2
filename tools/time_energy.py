"""Diagnostic: System::calc_energies on the device (K10/K11) against K1 on the same system: pairs per second of both.
The energies call is blocking (it returns two host scalars), so its time includes one stream sync and a 16-byte copy."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
for dtype, name in ((nb.F64, "f64"), (nb.F32, "f32")):
    for n in (65536, 262144):
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
        dev.calc_energies(); dev.all_pairs_force(); dev.sync()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            ke, pe = dev.calc_energies()
        te = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            dev.all_pairs_force()
        dev.sync()
        tf = (time.perf_counter() - t0) / reps
        pairs = n * (n - 1)
        print(f"{name} n={n}: energies {te*1e3:8.3f} ms ({pairs/te:.3e} pairs/s)   K1 force {tf*1e3:8.3f} ms ({pairs/tf:.3e} pairs/s)   "
              f"energies/K1 pair rate = {tf/te:.2f}   KE={ke:.6e} PE={pe:.6e}", flush=True)
        dev.close()
