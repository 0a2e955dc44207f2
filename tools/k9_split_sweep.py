"""Diagnostic (experiments build): the ISA sweep at config 4 against the share of groups the work-item kernel cuts in two and starts
first (NBODY_K9_SPLIT = denominator: 16 ships), repeated launches on the initial tree, ms per traversal.
    python tools/k9_split_sweep.py [n] [float]"""
import os, sys, time
from _experiments import load_package
nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dtype = nb.F32 if "float" in sys.argv[1:] else nb.F64
dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
st, t = dev.state(), dev.bvh
t.bounding_box(st, dev.stream); t.hilbert_sort(st, dev.stream); t.build_tree(st, dev.stream); dev.sync()
t.set_traversal(5)
for den in (4, 8, 12, 16, 24, 32, 64, 128):
    os.environ["NBODY_K9_SPLIT"] = str(den)
    t.compute_force(st, 0.5, dev.stream); dev.sync()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(5):
            t.compute_force(st, 0.5, dev.stream)
        dev.sync()
        best = min(best, (time.perf_counter() - t0) / 5 * 1e3)
    print(f"n={n} dtype={dtype} 1/{den} of the groups cut: {best:.2f} ms", flush=True)
