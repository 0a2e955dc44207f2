import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from conftest import load_package
nb = load_package()
import oracle as O
print(nb.device_info())
def rel(a, b):
    return np.abs(a-b).max() / max(np.abs(b).max(), 1e-300)
ok = True
for dt in (O.F64, O.F32):
  for dim in (3, 2):
    for wl, n in (("galaxy", 1000), ("uniform", 777), ("uniform", 4099)):
        s = O.build_model(dt, dim, wl, n)
        hs = nb.build_model(dt, dim, wl, n)
        assert np.array_equal(hs.x, s.x) and np.array_equal(hs.v, s.v) and np.array_equal(hs.m, s.m), "model mismatch"
        dev = nb.DeviceSystem.from_host(hs)
        for (js, r) in ((0,0),(1,1),(2,2),(4,1)):
            nb.configure_all_pairs(js, r)
            dev.all_pairs_force(); dev.sync()
            out = dev.download()
            O.all_pairs_force(s)
            e = rel(out.a, s.a)
            tol = 1e-13 if dt == O.F64 else 2e-5
            print(f"K1 dtype={dt} dim={dim} {wl} n={n} js={js} r={r}: rel err {e:.3e}", "OK" if e < tol else "FAIL"); ok &= e < tol
        nb.configure_all_pairs(0,0)
        # shard consistency
        full = out.a.copy()
        dev.all_pairs_force(0, n//3); dev.all_pairs_force(n//3, n - n//3); dev.sync()
        out2 = dev.download()
        print("   shard bitwise:", np.array_equal(out2.a, full)); ok &= np.array_equal(out2.a, full)
        # K3 bit exact
        dev.accelerate_step(); dev.sync(); out3 = dev.download()
        s.a[:] = out2.a  # same a as device
        O.accelerate_step(s)
        b = np.array_equal(out3.x, s.x) and np.array_equal(out3.v, s.v) and np.array_equal(out3.ao, s.ao)
        print("   K3 bitwise:", b); ok &= b
        # K2
        s2 = O.build_model(dt, dim, wl, n); hs2 = nb.build_model(dt, dim, wl, n)
        dev2 = nb.DeviceSystem.from_host(hs2)
        dev2.all_pairs_collapsed_force(); dev2.sync(); o2 = dev2.download()
        O.all_pairs_force(s2)
        e = rel(o2.a, s2.a); tol = 1e-12 if dt == O.F64 else 5e-5
        print(f"   K2 rel err {e:.3e}", "OK" if e < tol else "FAIL"); ok &= e < tol
        # BVH
        for theta in (0.0, 0.5):
            s3 = O.build_model(dt, dim, wl, n); hs3 = nb.build_model(dt, dim, wl, n)
            dev3 = nb.DeviceSystem.from_host(hs3)
            st = dev3.state(); b3 = dev3.bvh; b3.enable_counters(True)
            b3.bounding_box(st, dev3.stream); lo, hi = b3.get_bounding_box(dev3.stream)
            olo, ohi = O.bounding_box(s3)
            okb = np.array_equal(lo, olo) and np.array_equal(hi, ohi)
            b3.hilbert_sort(st, dev3.stream); keys = b3.read(0, dev3.stream); perm = b3.read(1, dev3.stream)
            okeys = O.hilbert_keys(s3, olo, ohi); operm = O.sort_keys(okeys)
            okk = np.array_equal(keys, okeys); okp = np.array_equal(perm, operm)
            O.apply_perm(s3, operm)
            b3.build_tree(st, dev3.stream)
            tr = O.bvh_build(s3)
            nm = b3.read(2, dev3.stream); bw = b3.read(3, dev3.stream); bx = b3.read(4, dev3.stream)
            okt = np.array_equal(nm, tr.nm) and np.array_equal(bw, tr.nbw) and np.array_equal(bx, tr.nb)
            b3.compute_force(st, theta, dev3.stream); dev3.sync()
            cnt = b3.read(5, dev3.stream); o3 = dev3.download()
            ocnt = O.bvh_force(s3, tr, theta, want_counts=True)
            okc = np.array_equal(cnt, ocnt)
            e = rel(o3.a, s3.a); tol = 1e-13 if dt == O.F64 else 2e-5
            okx = np.array_equal(o3.x, s3.x) and np.array_equal(o3.m, s3.m)
            print(f"   BVH theta={theta}: bbox {okb} keys {okk} perm {okp} tree {okt} gather {okx} counts {okc} force rel {e:.3e}")
            ok &= okb and okk and okp and okt and okc and okx and e < tol
print("ALL OK" if ok else "SOME FAILED")
