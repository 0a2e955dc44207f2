"""Config 2 as written, state by state: K1 timed over 60 launches on the FROZEN state after 0, 3, 7, 10, 20, 30, 45, 70, 100 steps,
with what the pair rules see there: the rule in force, the box, where the bulk of the bodies sits (1st-99th percentile box), and —
from a sample of 1024 targets against all sources on the host — the share of 256-pair batches (64 consecutive targets x 2 x 2
consecutive sources is what a wave holds; estimated per pair) that hold a pair closer than 2 (the sparse rule's mixed path) or
closer than 2^-8 (both rules' guarded path)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import load_package
nb = load_package()
n = 65536
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "uniform", n))
done = 0
rng = np.random.default_rng(3)
for upto in (0, 3, 7, 10, 20, 30, 45, 70, 100):
    nb.run(dev, "all-pairs", upto - done); done = upto
    dev.sync()
    hs = dev.download()
    frozen = nb.DeviceSystem.from_host(hs)
    sparse, vol = nb.all_pairs_pair_rule(frozen.state(), frozen.stream)
    for _ in range(30): frozen.all_pairs_force()
    frozen.sync(); t0 = time.perf_counter()
    for _ in range(60): frozen.all_pairs_force()
    frozen.sync(); ms = (time.perf_counter() - t0) / 60 * 1e3
    x = hs.x
    lo, hi = np.percentile(x, 1, axis=0), np.percentile(x, 99, axis=0)
    # batches as the kernel forms them: a wave = 128 consecutive targets (R = 2), a batch = 2 consecutive sources
    t0s = rng.integers(0, n // 128, 8) * 128
    close2 = close8 = batches = 0
    for t in t0s:
        d = x[None, :, :] - x[t:t + 128, None, :]
        r2 = (d * d).sum(-1)                                   # (128, n)
        r2[np.arange(128), np.arange(t, t + 128)] = np.inf     # the self pair goes through the guarded path regardless; not counted
        b = r2.reshape(128, n // 2, 2).min(axis=(0, 2))         # min over the wave's 256 pairs of each batch
        close2 += (b < 4.0).sum(); close8 += (b < 2.0 ** -16).sum(); batches += b.size
    finite = np.isfinite(x).all()
    print(f"after {upto:3d} steps: K1 {ms:.4f} ms  rule {'sparse' if sparse else 'dense '} box volume {vol:9.3g}  extent {np.ptp(x, axis=0).round(1).tolist()}  "
          f"1-99 % box {(hi - lo).round(1).tolist()}  batches with a pair < 2: {100 * close2 / batches:5.1f} %  < 2^-8: {100 * close8 / batches:6.3f} %  finite {finite}")
    frozen.close()
