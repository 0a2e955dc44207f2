"""Diagnostic: ms per whole step of the tree algorithms when the step is recorded once and replayed (what the CLI's default and
--csv-total modes do), galaxy, theta 0.5 — the octree with every build form (nbody_octree_set_build).
    python tools/time_step_graph.py [float]      (STEP_GRAPH_N=1000,2048: other sizes)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package(experiments=bool(os.environ.get("STEP_GRAPH_EXP")))   # STEP_GRAPH_EXP=1: the experiments build (reads the NBODY_* switches)
dtype = nb.F32 if (len(sys.argv) > 1 and sys.argv[1] == "float") else nb.F64


def per_step(dev, step, steps=200):
    step(); dev.sync()                    # everything allocated before the capture
    g = nb.StepGraph(dev, step)
    for _ in range(10):
        g.launch()
    dev.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            g.launch()
        dev.sync()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    g.close()
    return best


for n in [int(v) for v in os.environ.get("STEP_GRAPH_N", "10000,100000,1000000").split(",")]:
    row = []
    for form in (1, 3):
        dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
        dev.octree.set_build(form)
        dev.octree_force(0.5); dev.octree.info(dev.stream)   # build 4: the tree's depth is known from here on
        row.append("octree build=%d %.3f" % (form, per_step(dev, lambda: (dev.octree_force(0.5), dev.accelerate_step()), 200 if n < 1000000 else 50)))
        dev.octree.info(dev.stream)
        dev.close()
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
    row.append("bvh %.3f" % per_step(dev, lambda: (dev.bvh_force(0.5), dev.accelerate_step()), 200 if n < 1000000 else 50))
    dev.close()
    print("n=%d dtype=%d ms/step (graph replay): " % (n, dtype) + "  ".join(row), flush=True)
