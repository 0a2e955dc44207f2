"""Diagnostic: step the octree CLI-equivalent loop until nbody_octree_info reports a depth-limit build; print the step,
the root cube side, NaN count and the closest pair among the offending region."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
import numpy as np
nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "galaxy", n))
for step in range(1, 1001):
    nb.run(dev, "octree", 1, 0.5)
    try:
        size, mass = dev.octree.info(dev.stream)
    except nb.NbodyError as e:
        x = dev.download().x
        side = x.max() - x.min() + 2
        print(f"step {step}: {e}")
        print("nan:", np.isnan(x).sum(), "root side:", side, "resolution side/2^21:", side / 2**21, "|x| max:", np.abs(x).max())
        r = np.linalg.norm(x, axis=1)
        print("bodies beyond 100:", (r > 100).sum(), "beyond 1000:", (r > 1000).sum())
        break
    if step % 100 == 0:
        x = dev.download().x
        print(f"step {step}: tree size {size}, root side {x.max() - x.min() + 2:.3g}", flush=True)
else:
    print("1000 steps without a depth-limit build")
