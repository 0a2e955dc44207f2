#!/usr/bin/env python3
"""double `--print-state` identity holds "for N <= 4096, theta = 0 and all-pairs".  This script runs the REAL
reference (oracle/_ref, `make -C oracle ref`) at n = 4096 and writes its printed final rows — outputs only — to
tests/golden/print_state_n4096.json.gz.  Kept apart from print_state.json because of its size (4 x 4096 rows).

    python tests/golden/generate_golden_n4096.py
"""
import gzip
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "print_state_n4096.json.gz")
CASES = [(3, "all-pairs", "galaxy", None), (3, "bvh", "galaxy", 0.0), (2, "all-pairs", "uniform", None), (2, "bvh", "uniform", 0.0)]


def main():
    out = {}
    for dim, algo, wl, theta in CASES:
        args = ["-n", 4096, "-s", 5, "--precision", "double", "--algorithm", algo, "--workload", wl, "--print-state"]
        if theta is not None:
            args += ["--theta", theta]
        start, final = O.parse_print_state(O.ref_run(dim, args, timeout=3600))
        assert len(start) == len(final) == 4096
        name = f"d{dim}_double_{algo}_{wl}_n4096_s5" + ("" if theta is None else f"_th{theta}")
        out[name] = {"dim": dim, "precision": "double", "algorithm": algo, "workload": wl, "n": 4096, "steps": 5, "theta": theta,
                     "args": [str(a) for a in args], "start_md5": hashlib.md5("\n".join(start).encode()).hexdigest(), "final": final}
        print(name, "final md5", hashlib.md5("\n".join(final).encode()).hexdigest(), flush=True)
    with gzip.open(OUT, "wt", compresslevel=9) as f:
        json.dump(out, f, indent=0)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
