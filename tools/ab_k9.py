import os, sys, subprocess, json, time
ROOT='/root/repo'
sys.path.insert(0, os.path.join(ROOT,'tests'))
def child(lib):
    from conftest import load_package
    nb = load_package(); nb.LIB_PATH, nb._lib = lib, None
    out = {}
    for label, dtype, n in (("f64 1e6", 1, 1000000), ("f32 1e6", 0, 1000000), ("f64 1e5", 1, 100000), ("f64 3e5", 1, 300000)):
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
        st, t = dev.state(), dev.bvh
        t.bounding_box(st, dev.stream); t.hilbert_sort(st, dev.stream); t.build_tree(st, dev.stream)
        t.set_traversal(2)
        for _ in range(3): t.compute_force(st, 0.5, dev.stream)
        dev.sync(); best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            for _ in range(10): t.compute_force(st, 0.5, dev.stream)
            dev.sync(); best = min(best, (time.perf_counter() - t0) / 10 * 1e3)
        out[label + " traversal"] = best
        # whole step loop, 40 steps recorded
        g = nb.StepGraph(dev, lambda: (dev.bvh_force(0.5), dev.accelerate_step()))
        for _ in range(10): g.launch()
        dev.sync(); t0 = time.perf_counter()
        for _ in range(40): g.launch()
        dev.sync(); out[label + " step (steps 11-50)"] = (time.perf_counter() - t0) / 40 * 1e3
        g.close(); dev.close()
    print(json.dumps(out))
if len(sys.argv) == 3 and sys.argv[1] == "--child":
    child(sys.argv[2]); sys.exit(0)
libs = ["libnbody_hip_var_pre8.so", "libnbody_hip.so"]
res = {}
for rnd in range(2):
    for lib in libs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", os.path.join(ROOT, "stdpar-nbody_amd", lib)], capture_output=True, text=True, timeout=600)
        if r.returncode: print(lib, "FAILED", r.stderr[-500:]); continue
        res.setdefault(lib, []).append(json.loads(r.stdout.strip().splitlines()[-1]))
print("%-30s" % "K9 (ms; rounds)" + "".join("%30s" % l for l in libs))
for k in res[libs[0]][0]:
    print("%-30s" % k + "".join("%30s" % " / ".join("%.3f" % rr[k] for rr in res[l]) for l in libs))
