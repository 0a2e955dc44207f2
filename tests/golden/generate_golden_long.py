#!/usr/bin/env python3
"""Long-run fixtures from the REAL reference (oracle/_ref): SURVEY §8(c)'s double criterion at its stated length —
"positions within rel 1e-11 after 100 steps at N = 1024" — and an energy trace long enough to show accumulated drift.

    make -C oracle ref ref_ofast && python tests/golden/generate_golden_long.py

Every case is 3D double, N = 1024, `--save all --csv-detailed` (exactly `-s` steps, steps + 1 frames and energy pairs:
/root/reference/src/all_pairs.h:72-83, src/saving.h:100-122, src/system.h:62-79):
  * 100 steps, galaxy and uniform x {all-pairs, bvh theta 0, bvh theta 0.5, octree theta 0.5 (the reference's default algorithm)}:
    frames 0, 50, 100 and all 101 (KE, PE) pairs;
  * 1000 steps, galaxy all-pairs: frames 0 and 1000 and all 1001 pairs;
  * float (the reference's default precision), galaxy, 100 steps: all-pairs, bvh and octree at theta 0.5;
  * 2D double, galaxy, 100 steps: the same three (yardstick: the -Ofast build alone).
Beside each case the meta file records how far the reference's OTHER legitimate builds — -Ofast -march=native (its own CPU
flags, ci/run:112-113) and -O2 -march=native (IEEE operations, FMA contraction) — are from the -O2 build at the same frames /
energies, the larger of the two: the yardstick the GPU tests scale their tolerances by.  (-O1 and -O3 produce the -O2 build's
bits.  At step 1000, behind the discs' collision at step 366, the two are 3.7e-11 and 1.9e-10 from -O2 in energy.)  Only CLI arguments and the reference's outputs are written — no reference source.

Writes long_runs.npz and long_runs_meta.json.
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

OUT = os.path.dirname(os.path.abspath(__file__))
N = 1024


def run(exe, args):
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([exe] + [str(a) for a in args], cwd=d, capture_output=True, text=True, check=True)
        frames, _ = O.read_positions_bin(os.path.join(d, "positions.bin"))
        en, _ = O.read_energy_bin(os.path.join(d, "energy.bin"))
        return frames.copy(), en.copy()


def total(en):
    return en[:, 0] + en[:, 1]


def main():
    o2 = os.path.join(O.REF_DIR, "nbody_ref_d3")
    others = [os.path.join(O.REF_DIR, "nbody_ref_ofast_d3"), os.path.join(O.REF_DIR, "nbody_ref_native_d3")]
    assert os.path.exists(o2) and all(os.path.exists(p) for p in others), "make -C oracle ref ref_ofast"
    cases = [(wl, algo, th, 100, [0, 50, 100]) for wl in ("galaxy", "uniform")
             for algo, th in (("all-pairs", None), ("bvh", 0.0), ("bvh", 0.5), ("octree", 0.5))]
    cases = [c + ("double",) for c in cases]
    cases.append(("galaxy", "all-pairs", None, 1000, [0, 1000], "double"))
    # float, the reference's default precision: the galaxy over 100 steps, all-pairs and the two trees at the default angle
    cases += [("galaxy", algo, th, 100, [0, 50, 100], "float") for algo, th in (("all-pairs", None), ("bvh", 0.5), ("octree", 0.5))]
    cases = [c + (3,) for c in cases]
    # 2D double (the reference is built once per dimension): the galaxy over 100 steps, frames 0 and 100
    cases += [("galaxy", algo, th, 100, [0, 100], "double", 2) for algo, th in (("all-pairs", None), ("bvh", 0.5), ("octree", 0.5))]
    arrays, meta = {}, {}
    for wl, algo, th, steps, keep, prec, dim in cases:
        o2 = os.path.join(O.REF_DIR, f"nbody_ref_d{dim}")
        others = [os.path.join(O.REF_DIR, f"nbody_ref_ofast_d{dim}")] + ([os.path.join(O.REF_DIR, "nbody_ref_native_d3")] if dim == 3 else [])
        name = f"d{dim}_{prec}_{algo}_{wl}_n{N}_s{steps}" + ("" if th is None else f"_th{th}")
        args = ["-n", N, "-s", steps, "--precision", prec, "--algorithm", algo, "--workload", wl, "--save", "all", "--csv-detailed"]
        if th is not None:
            args += ["--theta", th]
        f2, e2 = run(o2, args)
        assert f2.shape == (steps + 1, N, dim) and e2.shape == (steps + 1, 2)
        scale = float(np.abs(f2[0]).max())
        E2 = total(e2.astype(np.float64))
        pos_spread, en_spread = np.zeros(len(keep)), np.zeros(steps + 1)
        for exe in others:   # the larger distance of the two other builds, frame by frame and step by step
            fo, eo = run(exe, args)
            pos_spread = np.maximum(pos_spread, [np.abs(f2[k].astype(np.float64) - fo[k]).max() / scale for k in keep])
            en_spread = np.maximum(en_spread, np.abs(E2 - total(eo.astype(np.float64))) / np.abs(E2))
        arrays[name + "__frames"] = f2[keep]
        arrays[name + "__energy"] = e2
        arrays[name + "__build_energy_spread"] = en_spread   # per step, relative, E = KE + PE
        meta[name] = {
            "dim": dim, "precision": prec, "algorithm": algo, "workload": wl, "n": N, "steps": steps, "theta": th,
            "args": [str(a) for a in args], "frame_ids": keep, "position_scale": scale,
            # max |x_O2 - x_other| / scale at the kept frames, row by row (for bvh every build prints its own sorted order: the
            # spreads are small, so the orders agree)
            "build_position_spread": [float(v) for v in pos_spread],
            "build_energy_spread_last": float(en_spread[-1]),
            "build_energy_spread_max": float(en_spread.max()),
            "energy_drift_last": float((E2[-1] - E2[0]) / abs(E2[0])),
        }
        print(name, {k: meta[name][k] for k in ("build_position_spread", "build_energy_spread_last", "build_energy_spread_max",
                                                 "energy_drift_last")})
    np.savez_compressed(os.path.join(OUT, "long_runs.npz"), **arrays)
    json.dump(meta, open(os.path.join(OUT, "long_runs_meta.json"), "w"), indent=1)
    print(f"wrote {len(meta)} long-run cases")


if __name__ == "__main__":
    main()
