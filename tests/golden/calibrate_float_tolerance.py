#!/usr/bin/env python3
"""Float parity tolerances measured on the reference itself ("to be calibrated by the same O2-vs-Ofast
experiment").  Builds the unmodified reference twice — `-O2` (oracle/_ref/nbody_ref_d*) and with its own CPU flags
`-Ofast -march=native` (oracle/_ref/nbody_ref_ofast_d*) — runs BASELINE config[0] (all-pairs 2D float uniform n=10000,
10 steps) and smaller float cases with `--save pos --csv-detailed`, and records how far the two builds' positions drift
apart, and how far the forces recomputed (by the oracle, one pass, same code for both) from those positions differ.
Writes tests/golden/float_tolerance.json; tests/test_gpu_all_pairs.py imports the tolerances from it.

    make -C oracle ref ref_ofast && python tests/golden/calibrate_float_tolerance.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "float_tolerance.json")


def frames(exe, dim, n, steps, wl, algo="all-pairs", theta=None):
    with tempfile.TemporaryDirectory() as d:
        args = [exe, "-n", str(n), "-s", str(steps), "--precision", "float", "--algorithm", algo, "--workload", wl,
                "--save", "pos", "--csv-detailed"]
        if theta is not None:
            args += ["--theta", str(theta)]
        subprocess.run(args, cwd=d, check=True, capture_output=True, timeout=3600)
        fr, _ = O.read_positions_bin(os.path.join(d, "positions.bin"))
        return fr.copy()


def matched_spread(a, b):
    """max |a_i - b_pi(i)| / max|a| where pi pairs every row of a with its nearest row of b (bvh permutes the bodies every
    step, each build by its own sort; rows are matched by position, which is safe while the spread stays far below the
    spacing of the bodies — asserted: the match must be one to one)."""
    a, b = a.astype(np.float64), b.astype(np.float64)
    d2 = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
    nn = d2.argmin(axis=1)
    assert len(set(nn.tolist())) == len(a), "rows of the two builds cannot be matched one to one"
    return float(np.sqrt(d2[np.arange(len(a)), nn]).max() / np.abs(a).max())


def tree_cases():
    """The reference's own manual check is a float tree-vs-all-pairs comparison (README.md:122-129): how far do two builds of the
    reference drift apart on the tree algorithms in float?  bvh theta 0 / 0.5 and octree theta 0 / 0.5, 2D and 3D, 10 steps."""
    rows = []
    for dim, wl, n in ((2, "uniform", 100), (2, "uniform", 1000), (3, "uniform", 1000), (3, "galaxy", 1000), (3, "galaxy", 100)):
        for algo, theta in (("bvh", 0.0), ("bvh", 0.5), ("octree", 0.0), ("octree", 0.5)):
            a = frames(os.path.join(O.REF_DIR, f"nbody_ref_d{dim}"), dim, n, 10, wl, algo, theta)
            b = frames(os.path.join(O.REF_DIR, f"nbody_ref_ofast_d{dim}"), dim, n, 10, wl, algo, theta)
            per = [matched_spread(a[k], b[k]) for k in (4, 10)]
            rows.append({"dim": dim, "workload": wl, "n": n, "algorithm": algo, "theta": theta, "steps": 10,
                         "position_spread_after_4_steps": per[0], "position_spread_final": per[1]})
            print(dim, wl, n, algo, theta, "pos spread after 4 / 10 steps", per[0], per[1], flush=True)
    return rows


def forces(dim, wl, n, x):
    s = O.build_model(O.F32, dim, wl, n)
    s.x[:] = x
    O.all_pairs_force(s)
    return s.a.astype(np.float64)


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref", "ref_ofast"])
    rows = []
    for dim, wl, n, steps in ((2, "uniform", 10000, 10), (2, "uniform", 1000, 10), (3, "galaxy", 1000, 10), (3, "uniform", 4096, 10)):
        a = frames(os.path.join(O.REF_DIR, f"nbody_ref_d{dim}"), dim, n, steps, wl)
        b = frames(os.path.join(O.REF_DIR, f"nbody_ref_ofast_d{dim}"), dim, n, steps, wl)
        scale = np.abs(a[0]).max()
        pos = [float(np.abs(a[k].astype(np.float64) - b[k]).max() / scale) for k in range(steps + 1)]
        fa, fb = forces(dim, wl, n, a[steps]), forces(dim, wl, n, b[steps])
        err = np.abs(fa - fb).max(axis=1) / np.abs(fa).max()
        f1a, f1b = forces(dim, wl, n, a[1]), forces(dim, wl, n, b[1])
        rows.append({"dim": dim, "workload": wl, "n": n, "steps": steps,
                     "position_spread_per_frame": pos, "position_spread_final": pos[-1],
                     "force_spread_final_max": float(err.max()), "force_spread_final_p99": float(np.quantile(err, 0.99)),
                     "force_spread_final_median": float(np.median(err)),
                     "force_spread_after_1_step_max": float((np.abs(f1a - f1b).max(axis=1) / np.abs(f1a).max()).max())})
        print(rows[-1]["dim"], wl, n, "pos", pos[-1], "force max/p99/median", rows[-1]["force_spread_final_max"],
              rows[-1]["force_spread_final_p99"], rows[-1]["force_spread_final_median"], flush=True)
    trees = tree_cases()
    c1 = rows[0]
    tol = {
        "_how": "tests/golden/calibrate_float_tolerance.py: the reference built -O2 vs -Ofast -march=native (g++), all-pairs float; "
                "spreads are max |difference| / max |value| (positions: of frame 0; forces: of the -O2 build's forces)",
        "cases": rows,
        # what the tests assert: the product may differ from the oracle by as much as two builds of the reference differ from
        # each other, times a safety factor of 4 (different boxes / libm), never below the one-pass float rounding floor
        "config1_trajectory_rel": max(4.0 * c1["position_spread_final"], 2e-5),
        "config1_force_after_10_steps_max_rel": max(4.0 * c1["force_spread_final_max"], 2e-5),
        "config1_force_after_10_steps_p99_rel": max(4.0 * c1["force_spread_final_p99"], 2e-5),
        "float_trajectory_rel_small_n": max(4.0 * max(r["position_spread_final"] for r in rows[1:]), 2e-5),
        # bvh / octree, float, <= 10 steps, n <= 1000: the same rule on the tree runs of the two builds
        "tree_cases": trees,
        "float_tree_trajectory_rel": max(4.0 * max(r["position_spread_final"] for r in trees), 2e-5),
        "float_tree_trajectory_rel_2d": max(4.0 * max(r["position_spread_final"] for r in trees if r["dim"] == 2), 2e-5),
        "float_tree_trajectory_rel_3d": max(4.0 * max(r["position_spread_final"] for r in trees if r["dim"] == 3), 2e-5),
    }
    old = {}
    try:
        old = json.load(open(OUT))
    except Exception:
        pass
    for k in ("config3_collapsed",):   # measured on the GPU by tests/golden/calibrate_config3_atomics.py: kept across re-runs
        if k in old:
            tol[k] = old[k]
    json.dump(tol, open(OUT, "w"), indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
