"""Diagnostic: time K1 over (source path, split, targets_per_thread) configs, full system and 1/8 shard.
usage: tune_all_pairs.py [n] [double|float] [dim]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dtype = nb.F32 if (len(sys.argv) > 2 and sys.argv[2] == "float") else nb.F64
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 3
peak = 157.3 if dtype == nb.F32 else 78.6
flop = 20.0 if dim == 3 else 14.0
hs = nb.build_model(dtype, dim, "galaxy", n)
dev = nb.DeviceSystem.from_host(hs)
for (label, first, count) in (("full", 0, n), ("1/8 shard", 0, n // 8)):
    for path in (1, 2):
        for js in ((1, 2, 4) if path == 1 else (1, 2, 4, 8)):
            for r in (1, 2):
                nb.configure_all_pairs(js, r, source_path=path)
                dev.all_pairs_force(first, count); dev.sync()
                reps = 3 if n > 300000 else 20
                t0 = time.perf_counter()
                for _ in range(reps):
                    dev.all_pairs_force(first, count)
                dev.sync()
                t = (time.perf_counter() - t0) / reps
                tf = flop * count * (n - 1) / t / 1e12
                print(f"n={n} dim={dim} dtype={dtype} {label:10s} path={'lds' if path == 1 else 'sgpr'} split={js} tpt={r}: {t*1e3:9.3f} ms  {tf:6.2f} TFLOP/s ({100*tf/peak:5.1f}% of vector peak)", flush=True)
