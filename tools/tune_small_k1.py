"""Experiment (libnbody_hip_exp.so): the launch shape of K1 at the reference's small benchmark sizes (ci/benchmark:64,137) — N = 10^4
(2D float: config 1's size; 3D double), 3 * 10^4 and 65 536 (config 2) — over source chunks (NBODY_K1_CHUNKS), targets per lane
and how far the collecting hand-off reaches (NBODY_K1_COLLECT_MAX: 2048 blocks shipped).  Chunks change the ROUNDING ORDER (a
function of sz alone in the shipped plan), so a row here is a what-if for the plan, not a setting.  ms per nbody_all_pairs_force
(pre-pass + K1, 100 back-to-back calls), % of the nominal vector peak at 20 (3D) / 14 (2D) flop per pair."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
CASES = [(10000, nb.F32, 2, "uniform"), (10000, nb.F64, 3, "uniform"), (30000, nb.F64, 3, "uniform"), (65536, nb.F64, 3, "uniform")]
for n, dtype, dim, wl in CASES:
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, dim, wl, n))
    flop, peak = (20.0 if dim == 3 else 14.0), (157.3 if dtype == nb.F32 else 78.6)
    def t(label):
        dev.all_pairs_force(); dev.sync()
        reps = 100 if n < 60000 else 30
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                dev.all_pairs_force()
            dev.sync()
            best = min(best, (time.perf_counter() - t0) / reps)
        print(f"n={n} {'f32' if dtype == nb.F32 else 'f64'} {dim}D {label:42s}: {best*1e3:8.4f} ms {100 * flop * n * (n - 1) / best / 1e12 / peak:5.1f}%  {nb.describe_all_pairs(dev.state())}", flush=True)
    for k in ("NBODY_K1_CHUNKS", "NBODY_K1_COLLECT_MAX"):
        os.environ.pop(k, None)
    nb.configure_all_pairs(0, 0, source_path=0); t("shipped")
    ntiles = (n + 511) // 512
    for y in sorted({c for c in (8, 10, 13, 16, 20, 26, 32, 40, 64, 128) if c <= ntiles}):
        for r in (1, 2):
            if dtype == nb.F32 and r == 2 and n < 60000:
                continue
            for cmax in (2048, 4096, 0) if n < 60000 else (2048,):
                os.environ["NBODY_K1_CHUNKS"] = str(y)
                os.environ["NBODY_K1_COLLECT_MAX"] = str(cmax)
                nb.configure_all_pairs(8, r, source_path=2)
                t(f"chunks={y} tpt={r} collect<={cmax}")
    for k in ("NBODY_K1_CHUNKS", "NBODY_K1_COLLECT_MAX"):
        os.environ.pop(k, None)
    nb.configure_all_pairs(0, 0, source_path=0)
    dev.close()
