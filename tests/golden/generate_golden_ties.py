#!/usr/bin/env python3
"""Key ties in the Hilbert sort (missing #5).  The reference sorts (key, body) with an unstable std::sort
(src/bvh.h:55-94), so the order of two bodies in ONE Hilbert cell is unspecified; what is specified is the multiset of final
rows.  This script builds small systems that DO contain equal keys (pairs closer than a cell, one exactly coincident pair),
feeds them to the REAL reference (oracle/_ref, `make -C oracle ref`) through `--workload load`, and writes inputs + the
reference's outputs — data only — to tests/golden/bvh_ties.json:
  * `--print-state` final rows after 12 steps (default mode), bvh theta = 0 and all-pairs;
  * the last full-precision frame of `--save pos --csv-detailed -s 4` for bvh theta = 0.

    python tests/golden/generate_golden_ties.py
"""
import json
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bvh_ties.json")


def make_system(dim, seed):
    """n x (m, pos[D], vel[D]) in float32, with planted pairs that share a Hilbert cell."""
    rng = np.random.default_rng(seed)
    n = 96
    body = np.zeros((n, 1 + 2 * dim), np.float32)
    body[:, 0] = rng.uniform(0.5, 1.5, n)
    body[:, 1:1 + dim] = rng.uniform(-1.0, 1.0, (n, dim))
    body[:, 1 + dim:] = rng.uniform(-0.05, 0.05, (n, dim))
    # 3D: 21 bits per dimension over an extent of ~2 -> cells of ~1e-6; 2D: 32 bits -> cells of ~5e-10 (below float32 spacing
    # except near the origin).  Partners are planted far from their twin in index space.
    if dim == 3:
        for i, j, d in ((3, 70, 1.2e-7), (10, 55, 1.8e-7), (20, 90, 0.6e-7)):
            body[j, 1:4] = body[i, 1:4]
            body[j, 1] = np.float32(body[i, 1] + np.float32(d))
    else:
        for k, (i, j) in enumerate(((3, 70), (10, 55))):
            base = np.float32(1e-12 * (k + 1))
            body[i, 1:3] = (base, 2 * base)
            body[j, 1:3] = (base * np.float32(1.5), 2 * base * np.float32(1.1))
    body[40, 1:1 + dim] = body[41, 1:1 + dim]     # exactly coincident positions, different velocities and masses
    return body


def main():
    out = {}
    for dim, seed in ((3, 5), (2, 6)):
        body = make_system(dim, seed)
        n = body.shape[0]
        dt, G = np.float32(0.01), np.float32(1.0)
        # the ties must be real: equal keys in the oracle's restatement of the reference's key function
        s = O.State(O.F64, dim, n)
        s.m[:], s.x[:], s.v[:] = body[:, 0], body[:, 1:1 + dim], body[:, 1 + dim:]
        lo, hi = O.bounding_box(s)
        keys = O.hilbert_keys(s, lo, hi)
        uniq, counts = np.unique(keys, return_counts=True)
        tied = int(counts[counts > 1].sum())   # bodies that share their key with another body
        assert tied >= 6, f"dim {dim}: only {tied} bodies with tied keys"
        case = {"dim": dim, "n": n, "dt": float(dt), "G": float(G), "body_f32": body.astype(np.float64).tolist(), "bodies_with_tied_keys": tied}
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "ties.bin")
            with open(path, "wb") as f:
                f.write(struct.pack("<IIff", n, dim, float(dt), float(G)))
                f.write(body.tobytes())
            for algo in ("bvh", "all-pairs"):
                args = ["--workload", "load", path, "--precision", "double", "--algorithm", algo, "-s", 12, "--print-state"]
                if algo == "bvh":
                    args += ["--theta", 0]
                start, final = O.parse_print_state(O.ref_run(dim, args, cwd=d))
                assert len(final) == n
                case[f"final_rows_{algo}"] = final
            O.ref_run(dim, ["--workload", "load", path, "--precision", "double", "--algorithm", "bvh", "--theta", 0, "-s", 4,
                            "--csv-detailed", "--save", "pos"], cwd=d)
            frames, _ = O.read_positions_bin(os.path.join(d, "positions.bin"))
            assert frames.shape == (5, n, dim)
            case["bvh_last_frame"] = frames[-1].tolist()
        out[f"d{dim}"] = case
        print(f"d{dim}: n={n}, {tied} bodies share a key with another")
    with open(OUT, "w") as f:
        json.dump(out, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
