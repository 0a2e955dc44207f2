#!/usr/bin/env python3
"""Config 3 (all-pairs-collapsed 3D float, N = 262 144) has no reference output to compare with — the reference computes nothing
there (its 32-bit pair count wraps to 0) — so the GPU test pins it to the all-pairs oracle on a target sample.  The
kernel adds per-(target, source chunk) sums with float atomics, whose order changes from run to run: this script MEASURES that
spread (two runs of the product on one GPU, max |a1 - a2| / max |a|) and the distance of a run from the float all-pairs kernel,
and stores them in tests/golden/float_tolerance.json ("config3_collapsed"); tests/test_gpu_all_pairs.py allows the one-pass float
tolerance plus 4x the measured run-to-run spread.  Needs an MI355X:

    python tests/golden/calibrate_config3_atomics.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "float_tolerance.json")
GPU_OUT = os.path.join(ROOT, "gpurun_out", "config3_atomics.json")


def main():
    nb = load_package()
    n = 262144
    runs = []
    for _ in range(3):
        dev = nb.DeviceSystem.from_host(nb.build_model(nb.F32, 3, "uniform", n))
        dev.all_pairs_collapsed_force()
        runs.append(dev.download().a.astype(np.float64))
        dev.close()
    dev = nb.DeviceSystem.from_host(nb.build_model(nb.F32, 3, "uniform", n))
    dev.all_pairs_force()
    k1 = dev.download().a.astype(np.float64)
    scale = np.abs(k1).max()
    spread = max(np.abs(runs[0] - r).max() for r in runs[1:]) / scale
    res = {"n": n, "run_to_run_spread_max_rel": float(spread), "vs_all_pairs_kernel_max_rel": float(np.abs(runs[0] - k1).max() / scale),
           "_how": "tests/golden/calibrate_config3_atomics.py on one MI355X: three runs of K2, max |a_1 - a_k| / max |a|; and run 1 against K1"}
    print(json.dumps(res))
    os.makedirs(os.path.dirname(GPU_OUT), exist_ok=True)
    json.dump(res, open(GPU_OUT, "w"))
    try:   # in the build container the tracked file is writable; on the GPU box the result travels back through gpurun_out/
        tol = json.load(open(OUT))
        tol["config3_collapsed"] = res
        json.dump(tol, open(OUT, "w"), indent=1)
    except Exception as ex:
        print("could not update", OUT, ex)


if __name__ == "__main__":
    main()
