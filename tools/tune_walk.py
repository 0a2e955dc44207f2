"""Diagnostic: K1 time against the launch shape (NBODY_K1_TAIL, experiments build): the number of target blocks launched as
one-chunk blocks behind the ones that walk all their chunks themselves.  0 = every block walks all chunks (no chunk-sum scratch,
no combine launch), a value >= the number of target blocks = one chunk per block throughout (rounds 2-3).  The result is
bitwise the same for every value; what changes is how ragged the end of the launch is.
usage: tune_walk.py [double|float] [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
args = sys.argv[1:]
dtype = nb.F64
if args and args[0] in ("double", "float"):
    dtype = nb.F32 if args[0] == "float" else nb.F64
    args = args[1:]
sizes = [int(a) for a in args] or [262144, 1 << 20]
for n in sizes:
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, os.environ.get("WL", "galaxy"), n))
    base = None
    for rnd in range(2):
        for w in (0, 128, 256, 512, 1024, 2048, 1 << 30):
            os.environ["NBODY_K1_TAIL"] = str(w)
            dev.all_pairs_force(); dev.sync()
            a = dev.download().a.copy()
            if base is None:
                base = a
            same = bool((a == base).all())
            reps = 3 if n > 300000 else 10
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                for _ in range(reps):
                    dev.all_pairs_force()
                dev.sync()
                best = min(best, (time.perf_counter() - t0) / reps)
            print(f"n={n} tail={w:10d}: {best*1e3:9.3f} ms  bitwise={same}  {nb.describe_all_pairs(dev.state())}", flush=True)
    dev.close()
os.environ.pop("NBODY_K1_TAIL", None)
