"""K1 timing over the sizes the headline and the reference's matrix care about: ms per force pass, % of the vector peak,
and the launch the library picked (nbody_all_pairs_describe).  Diagnostic tool; one JSON document on stdout.
    python tools/time_all_pairs.py [reps]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cases = [("f64", nb.F64, "uniform", 65536, None), ("f64", nb.F64, "galaxy", 100000, None), ("f64", nb.F64, "galaxy", 262144, None),
         ("f64", nb.F64, "galaxy", 1 << 20, (1 << 20) // 8), ("f64", nb.F64, "galaxy", 1 << 20, None),
         ("f64", nb.F64, "uniform", 10000, None), ("f64", nb.F64, "uniform", 30000, None),
         ("f32", nb.F32, "uniform", 262144, None), ("f32", nb.F32, "uniform", 100000, None),
         ("f32", nb.F32, "galaxy", 262144, None), ("f64", nb.F64, "uniform", 262144, None), ("f64", nb.F64, "galaxy", 65536, None)]
out = []
for tname, dt, wl, n, count in cases:
    dev = nb.DeviceSystem.from_host(nb.build_model(dt, 3, wl, n))
    cnt = dev.n if count is None else count
    desc = nb.describe_all_pairs(dev.state(0, cnt))
    dev.all_pairs_force(0, cnt); dev.sync()
    t0 = time.perf_counter()          # warm up to the sustained clock: the first case of the process otherwise reads 12 % slow
    while time.perf_counter() - t0 < 0.3:
        dev.all_pairs_force(0, cnt); dev.sync()
    r = reps if n < (1 << 20) or count else max(2, reps // 2)
    t = 1e9
    for _ in range(3):                # best of three windows
        t0 = time.perf_counter()
        for _ in range(r):
            dev.all_pairs_force(0, cnt)
        dev.sync()
        t = min(t, (time.perf_counter() - t0) / r)
    peak = 157.3 if dt == nb.F32 else 78.6
    tf = 20.0 * cnt * (dev.n - 1) / t / 1e12
    out.append({"dtype": tname, "workload": wl, "n": dev.n, "targets": cnt, "ms": t * 1e3, "tflops": tf, "pct_peak": 100 * tf / peak, "launch": desc})
    dev.close()
print(json.dumps({"device": nb.device_info()[0], "results": out}, indent=1))
