"""A/B of K1 (and K2) between library builds ON ONE BOX in one session: boxes of the pool differ by several percent (clock under
the power cap), so a number from another session proves nothing about a code change.
    python tools/ab_all_pairs.py libA.so libB.so ...       (paths relative to stdpar-nbody_amd/; each runs in its own process)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
CASES = [("f64", "uniform", 65536), ("f64", "galaxy", 65536), ("f64", "uniform", 262144), ("f64", "galaxy", 262144),
         ("f64", "galaxy", 1 << 20), ("f32", "uniform", 262144), ("f32", "galaxy", 262144), ("f32", "uniform", 100000)]
if os.environ.get("AB_CASES"):   # e.g. AB_CASES="f32:uniform:100000,f64:galaxy:65536"
    CASES = [(a, b, int(c)) for a, b, c in (x.split(":") for x in os.environ["AB_CASES"].split(","))]


def child(lib):
    from conftest import load_package
    nb = load_package()
    nb.LIB_PATH, nb._lib = lib, None
    out = {}
    for tname, wl, n in CASES:
        dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64 if tname == "f64" else nb.F32, 3, wl, n))
        for name, fn in (("K1", dev.all_pairs_force),) + ((("K2", dev.all_pairs_collapsed_force),) if n <= 262144 else ()):
            fn(); dev.sync()
            best = 1e9
            for _ in range(3):
                reps = 3 if n > 300000 else 10
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                dev.sync()
                best = min(best, (time.perf_counter() - t0) / reps)
            out[f"{tname} {wl} {n} {name}"] = best * 1e3
        dev.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    libs = sys.argv[1:]
    res = {}
    for rnd in range(2):            # A B A B: drift of the box over the session shows up as a difference between the rounds
        for lib in libs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.join(ROOT, "stdpar-nbody_amd", lib)],
                               capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                print(lib, "FAILED", r.stderr[-800:]); continue
            res.setdefault(lib, []).append(json.loads(r.stdout.strip().splitlines()[-1]))
    keys = list(next(iter(res.values()))[0].keys())
    print("%-28s" % "case (ms, best of 3; rounds)" + "".join("%28s" % l[-26:] for l in libs))
    for k in keys:
        print("%-28s" % k + "".join("%28s" % " / ".join("%.3f" % rr[k] for rr in res[l]) for l in libs), flush=True)
