"""Diagnostic: the f32 pair-weight forms of K1 and K2 (NBODY_F32_PAIR, csrc/common.hpp) against each other at config 3 (uniform,
N = 262 144: dense relative to float's eps) and on the galaxy (every pair far).  One process per variant library
(stdpar-nbody_amd/libnbody_hip_var_f32pair<mode>.so, built by hand with -DNBODY_F32_PAIR=<mode>); prints one line per case and
the worst per-target deviation of the forces from variant 0 (rsq + rcp).
    python tools/time_f32_pair_forms.py            (parent: runs every variant found)"""
import glob, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(lib):
    import numpy as np
    from conftest import load_package
    nb = load_package()
    nb.LIB_PATH, nb._lib = lib, None
    out = {}
    for wl, n in (("uniform", 262144), ("galaxy", 262144)):
        dev = nb.DeviceSystem.from_host(nb.build_model(nb.F32, 3, wl, n))
        for name, fn in (("K1", dev.all_pairs_force), ("K2", dev.all_pairs_collapsed_force)):
            hs = dev.download(); hs.a[:] = 0; hs.ao[:] = 0; dev.upload(hs)
            fn(); dev.sync()
            a = dev.download().a.copy()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            dev.sync()
            t = (time.perf_counter() - t0) / 10
            out[f"{wl} {name}"] = {"ms": t * 1e3, "pct_peak": 100 * 20.0 * n * (n - 1) / t / 1e12 / 157.3}
            np.save(f"/tmp/f32forms_{os.path.basename(lib)}_{wl}_{name}.npy", a)
        dev.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    import numpy as np
    libs = sorted(glob.glob(os.path.join(ROOT, "stdpar-nbody_amd", "libnbody_hip_var_f32pair*.so")))
    base = None
    for lib in libs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), lib], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(os.path.basename(lib), "FAILED", r.stderr[-500:]); continue
        res = json.loads(r.stdout.strip().splitlines()[-1])
        tag = os.path.basename(lib)[len("libnbody_hip_var_f32pair"):-3]
        for case, v in res.items():
            wl, name = case.split()
            a = np.load(f"/tmp/f32forms_{os.path.basename(lib)}_{wl}_{name}.npy").astype(np.float64)
            if tag == "0":
                base = base or {}
                base[case] = a
            dev = ""
            if base and case in base:
                dev = "  max |a - a(mode 0)| / max|a| = %.2e" % (np.abs(a - base[case]).max() / np.abs(base[case]).max())
            print(f"mode {tag:4s} {case:12s} {v['ms']:8.3f} ms  {v['pct_peak']:5.1f} % of FP32 peak{dev}", flush=True)
