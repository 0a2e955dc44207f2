"""Diagnostic: how evenly the sweep's eight XCD ranges of groups share the work at config 4 (sum over a range of its groups' longest
per-body walks, from the per-lane form's counters), and what more, smaller ranges dealt round-robin would do.
    python tools/k9_xcd_balance.py [n]"""
import os, sys
sys.path.insert(0, os.path.join('/root/repo', 'tests'))
import numpy as np
from conftest import load_package
nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "galaxy", n))
dev.bvh.set_traversal(1)           # per-lane walk: per-body counters
dev.bvh.enable_counters(True)
dev.bvh_force(0.5); dev.sync()
cnt = dev.bvh.read(5, dev.stream).reshape(-1, 4).astype(np.int64)
steps = cnt[:, 0] + cnt[:, 3] + cnt[:, 1]      # nodes examined + body terms + leaf visits: ~ entries visited
groups = (n + 63) // 64
g = np.add.reduceat(steps, np.arange(0, n, 64))
gmax = np.maximum.reduceat(steps, np.arange(0, n, 64))
for parts in (8, 16, 32, 64, 128):
    # contiguous ranges dealt round-robin to the 8 XCDs
    bounds = np.linspace(0, groups, parts + 1).astype(int)
    per = np.array([gmax[bounds[i]:bounds[i+1]].sum() for i in range(parts)], dtype=np.float64)
    xcd = np.array([per[x::8].sum() for x in range(8)])
    print("ranges %4d: per-XCD work (sum of the groups' longest walks) max/mean = %.4f   min/mean = %.4f" % (parts, xcd.max() / xcd.mean(), xcd.min() / xcd.mean()))
dev.close()
