"""Diagnostic: f32 K1 at config-3 size on the dense cube and on the galaxy, with 1 and 2 targets per lane, for one library
(path relative to stdpar-nbody_amd/, default the shipped one).  Boxes differ by several percent: compare libraries in ONE session.
    python tools/time_f32_k1.py [lib.so ...]"""
import os, subprocess, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(lib):
    from conftest import load_package
    nb = load_package()
    nb.LIB_PATH, nb._lib = lib, None
    out = {}
    for wl in ("uniform", "galaxy"):
        for n in (100000, 262144):
            dev = nb.DeviceSystem.from_host(nb.build_model(nb.F32, 3, wl, n))
            for tpt in (0, 2):
                nb.configure_all_pairs(0, tpt)
                dev.all_pairs_force(); dev.sync()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(10):
                        dev.all_pairs_force()
                    dev.sync()
                    best = min(best, (time.perf_counter() - t0) / 10)
                out[f"{wl} {n} tpt={tpt or 'auto'}"] = best * 1e3
            nb.configure_all_pairs(0, 0)
            dev.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2]); sys.exit(0)
    libs = sys.argv[1:] or ["libnbody_hip.so"]
    res = {}
    for rnd in range(2):
        for lib in libs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.join(ROOT, "stdpar-nbody_amd", lib)],
                               capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                print(lib, "FAILED", r.stderr[-600:]); continue
            res.setdefault(lib, []).append(json.loads(r.stdout.strip().splitlines()[-1]))
    keys = list(next(iter(res.values()))[0].keys())
    print("%-28s" % "f32 K1 (ms; rounds)" + "".join("%26s" % l[-24:] for l in res))
    for k in keys:
        print("%-28s" % k + "".join("%26s" % " / ".join("%.3f" % rr[k] for rr in res[l]) for l in res), flush=True)
