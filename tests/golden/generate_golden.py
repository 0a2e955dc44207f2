#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference (oracle/_ref, compiled
from /root/reference by `make -C oracle ref`).  Only inputs (CLI arguments, cell lists) and the
reference's outputs are written — no reference source.  Run in the build container:

    python tests/golden/generate_golden.py

Fixtures:
  print_state.json   --print-state text rows (start + final) per case          
  positions.npz      positions.bin frames (+ energy.bin pairs) per case        
  hilbert.json       hilbert<2>/<3>, interleave_bits known-answer vectors      
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def case_name(dim, prec, algo, wl, n, steps, theta):
    th = "" if theta is None else f"_th{theta}"
    return f"d{dim}_{prec}_{algo}_{wl}_n{n}_s{steps}{th}"


def print_state_cases():
    cases = []
    # D=2 float uniform -s 5 -n 10 (README.md:122-129 manual check) incl. collapsed
    for algo, theta in (("all-pairs", None), ("all-pairs-collapsed", None), ("bvh", 0.0), ("bvh", 0.5)):
        cases.append((2, "float", algo, "uniform", 10, 5, theta))
    for dim in (2, 3):
        for wl in ("uniform", "galaxy"):
            for n in (10, 64, 1000):
                for algo, theta in (("all-pairs", None), ("bvh", 0.0), ("bvh", 0.5)):
                    cases.append((dim, "double", algo, wl, n, 12 if n <= 64 else 5, theta))
    cases.append((2, "double", "all-pairs-collapsed", "uniform", 64, 12, None))
    for dim in (2, 3):  # octree: the reference's default algorithm (src/arguments.h:28)
        for wl in ("uniform", "galaxy"):
            for n in (10, 64):
                for theta in (0.0, 0.5):
                    cases.append((dim, "double", "octree", wl, n, 12, theta))
    cases.append((3, "double", "all-pairs", "plummer", 64, 12, None))
    cases.append((3, "float", "all-pairs", "galaxy", 10, 5, None))
    return cases


def positions_cases():
    cases = []
    for dim in (2, 3):
        for prec in ("float", "double"):
            for wl in ("uniform", "galaxy"):
                for n in (10, 100, 257):
                    for algo, theta in (("all-pairs", None), ("bvh", 0.0), ("bvh", 0.5)):
                        cases.append((dim, prec, algo, wl, n, 4, theta))
            cases.append((dim, prec, "all-pairs-collapsed", "uniform", 100, 4, None))
            for wl in ("uniform", "galaxy"):
                for n in (10, 257):
                    for theta in (0.0, 0.5):
                        cases.append((dim, prec, "octree", wl, n, 4, theta))
    cases.append((3, "double", "octree", "plummer", 257, 4, 0.5))
    cases.append((3, "double", "octree", "galaxy", 2000, 10, 0.5))
    cases.append((3, "double", "all-pairs", "plummer", 100, 4, None))
    cases.append((3, "double", "all-pairs", "galaxy", 1024, 20, None))
    cases.append((3, "double", "bvh", "galaxy", 1024, 20, 0.5))
    return cases


def main():
    ps = {}
    for (dim, prec, algo, wl, n, steps, theta) in print_state_cases():
        args = ["-n", n, "-s", steps, "--precision", prec, "--algorithm", algo, "--workload", wl, "--print-state"]
        if theta is not None:
            args += ["--theta", theta]
        start, final = O.parse_print_state(O.ref_run(dim, args))
        ps[case_name(dim, prec, algo, wl, n, steps, theta)] = {
            "dim": dim, "precision": prec, "algorithm": algo, "workload": wl, "n": n, "steps": steps, "theta": theta,
            "args": [str(a) for a in args], "start": start, "final": final}
    json.dump(ps, open(os.path.join(OUT, "print_state.json"), "w"), indent=0)

    arrays, meta = {}, {}
    for (dim, prec, algo, wl, n, steps, theta) in positions_cases():
        name = case_name(dim, prec, algo, wl, n, steps, theta)
        energy = n <= 100
        res = O.ref_positions(dim, prec, algo, wl, n, steps, theta, energy=energy)
        frames = res[0] if energy else res
        keep = frames if n <= 257 else frames[[0, 1, steps]]  # big cases: first, second, last frame only
        arrays[name + "__frames"] = keep
        if energy:
            arrays[name + "__energy"] = res[1]
        meta[name] = {"dim": dim, "precision": prec, "algorithm": algo, "workload": wl, "n": n, "steps": steps, "theta": theta,
                      "frame_ids": list(range(steps + 1)) if n <= 257 else [0, 1, steps]}
    np.savez_compressed(os.path.join(OUT, "positions.npz"), **arrays)
    json.dump(meta, open(os.path.join(OUT, "positions_meta.json"), "w"), indent=0)

    rng = np.random.default_rng(1234)
    cells = [(2, 0, 0, 0), (2, 1, 2, 0), (2, 0xffffffff, 0xffffffff, 0), (2, 0x80000000, 0x7fffffff, 0), (2, 0xffffffff, 0, 0),
             (3, 0, 0, 0), (3, 5, 6, 7), (3, 0x1fffff, 0x1fffff, 0x1fffff), (3, 0x100000, 0xfffff, 0x155555), (3, 1, 0, 0x1fffff)]
    for _ in range(200):
        cells.append((2, int(rng.integers(0, 2**32)), int(rng.integers(0, 2**32)), 0))
        cells.append((3, int(rng.integers(0, 2**21)), int(rng.integers(0, 2**21)), int(rng.integers(0, 2**21))))
    inp = "".join(f"{d} {a} {b} {c}\n" for d, a, b, c in cells)
    out = subprocess.run([os.path.join(O.REF_DIR, "ref_units")], input=inp, capture_output=True, text=True, check=True).stdout
    vecs = [[int(t) for t in line.split()] for line in out.splitlines()]  # dim c0 c1 c2 hilbert interleave
    json.dump(vecs, open(os.path.join(OUT, "hilbert.json"), "w"))
    print(f"wrote {len(ps)} print-state cases, {len(meta)} position cases, {len(vecs)} hilbert vectors")


if __name__ == "__main__":
    main()
