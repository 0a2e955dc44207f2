import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
import numpy as np
nb = load_package()
for wl in ("galaxy", "uniform"):
    n = 1000000
    dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, wl, n))
    dev.octree.enable_counters(True)
    dev.octree_force(0.5); dev.sync()
    c = dev.octree.read_counters(dev.stream).reshape(n, 2).astype(np.float64)
    print(wl, "octree nodes mean/max", c[:, 0].mean(), c[:, 0].max(), "terms mean/max", c[:, 1].mean(), c[:, 1].max(), "size", dev.octree.info(dev.stream))
