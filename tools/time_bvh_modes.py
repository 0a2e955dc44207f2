"""Diagnostic: time K9 in its three scheduling forms at a given N (3D galaxy theta=0.5)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
if "exp" in sys.argv[1:]:   # the -DNBODY_EXPERIMENTS build as it lies in the tree (built with whatever EXPDEFS the caller chose)
    nb.LIB_PATH = os.path.join(os.path.dirname(nb.LIB_PATH), "libnbody_hip_exp.so")
    print("library:", nb.LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dtype = nb.F32 if (len(sys.argv) > 2 and sys.argv[2] == "float") else nb.F64
dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
st, t = dev.state(), dev.bvh
t.bounding_box(st, dev.stream); t.hilbert_sort(st, dev.stream); t.build_tree(st, dev.stream); dev.sync()
for mode in ((1, 3, 5, 6) if "exp" in sys.argv[1:] else (1, 5)):
    t.set_traversal(mode)
    t.compute_force(st, 0.5, dev.stream); dev.sync()
    acc = dev.download().a.copy()
    if mode == 1:
        base = acc
    print("   bitwise equal to mode 1:", bool((acc == base).all()))
    reps, best = 5, 1e9
    for _ in range(4):   # the first rounds also bring the clock back up after the host-side comparison above
        t0 = time.perf_counter()
        for _ in range(reps):
            t.compute_force(st, 0.5, dev.stream)
        dev.sync()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    print(f"n={n} dtype={dtype} traversal mode {mode}: {best:.2f} ms", flush=True)
