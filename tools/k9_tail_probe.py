import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
n = 1000000
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "galaxy", n))
st, t = dev.state(), dev.bvh
t.bounding_box(st, dev.stream); t.hilbert_sort(st, dev.stream); t.build_tree(st, dev.stream); dev.sync()
t.set_traversal(5)
def timeit(first, count, reps=5):
    w = dev.state(first, count)
    t.compute_force(w, 0.5, dev.stream); dev.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        t.compute_force(w, 0.5, dev.stream)
    dev.sync()
    return (time.perf_counter() - t0) / reps * 1e3
print("whole", timeit(0, n))
for parts in (2, 4, 8, 16):
    ts = [timeit(n * k // parts, n // parts) for k in range(parts)]
    print(parts, "parts: sum %.2f ms, each" % sum(ts), " ".join("%.2f" % x for x in ts))
