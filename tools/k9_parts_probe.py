"""Diagnostic (experiments build): K9's sweep with every group cut into 1/2/4/8/16 equal lane ranges (NBODY_K9_PARTS) over sizes,
and at N = 10^6 the key-jump cutter with 1/16 .. all of the groups cut (NBODY_K9_SPLIT).  ms per traversal of the initial galaxy.
    python tools/k9_parts_probe.py [float]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
dtype = nb.F32 if (len(sys.argv) > 1 and sys.argv[1] == "float") else nb.F64


def timeit(dev, t, st, reps=5):
    t.compute_force(st, 0.5, dev.stream); dev.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            t.compute_force(st, 0.5, dev.stream)
        dev.sync()
        best = min(best, (time.perf_counter() - t0) / reps * 1e3)
    return best


for n in (10000, 30000, 60000, 100000, 200000, 500000, 1000000):
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
    st, t = dev.state(), dev.bvh
    t.bounding_box(st, dev.stream); t.hilbert_sort(st, dev.stream); t.build_tree(st, dev.stream); dev.sync()
    row = []
    t.set_traversal(1)
    row.append("per-lane %.3f" % timeit(dev, t, st))
    t.set_traversal(5)
    for parts in (1, 2, 4, 8, 16):
        os.environ["NBODY_K9_PARTS"] = str(parts)
        row.append("parts=%d %.3f" % (parts, timeit(dev, t, st)))
    os.environ.pop("NBODY_K9_PARTS")
    row.append("auto %.3f" % timeit(dev, t, st))
    print("n=%d dtype=%d: " % (n, dtype) + "  ".join(row), flush=True)
    if n == 1000000:
        os.environ["NBODY_K9_PARTS"] = "1"   # parts=1 -> the key-jump cutter
        for den in (16, 8, 4, 2, 1):
            os.environ["NBODY_K9_SPLIT"] = str(den)
            print("   n=1e6 cutter: 1/%d of the groups cut in two: %.3f ms" % (den, timeit(dev, t, st)), flush=True)
        os.environ.pop("NBODY_K9_SPLIT"); os.environ.pop("NBODY_K9_PARTS")
    dev.close()
