#!/usr/bin/env python3
"""Registers, LDS, scratch and the waves per SIMD they allow, for every kernel of a built library (read from the code object's
metadata notes; no GPU needed).  usage: kernel_resources.py [lib.so] [name-filter ...]

Waves per SIMD on gfx950: min(8, by VGPRs, by SGPRs).  VGPRs: 512 per lane and SIMD in the unified file, allocated in blocks of 8,
accumulation registers included.  SGPRs: 800 per SIMD in granules of 16 with 16 MORE held back per wave — measured, round 5
(tools/microbench/cu_map.hip: a kernel that names up to s73, sgpr_count <= 80, runs 8 waves per SIMD — s72 / s73 / s74 measured
8 / 8 / 7 in round 6 —; up to s88 seven; above six):
floor(800 / (sgpr_count rounded up to 16 + 16)).  A `*` marks kernels that the SGPRs hold below what their VGPRs allow."""
import os
import re
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
ELF_AMDGPU = b"\x7fELF\x02\x01\x01\x40"


def kernels(lib_path):
    raw = open(lib_path, "rb").read()
    out, pos, k = [], 0, 0
    with tempfile.TemporaryDirectory() as d:
        while True:
            pos = raw.find(ELF_AMDGPU, pos)
            if pos < 0:
                break
            path = os.path.join(d, f"co{k}.elf")
            open(path, "wb").write(raw[pos:])
            notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"^\s+(?:- )?\.([a-z_]+):\s+(.*)$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip()
                if key in ("agpr_count", "args") and cur.get("name"):  # first key of the next kernel's record
                    pass
                if key == "name" and "symbol" not in cur and cur.get("vgpr_count") is not None:
                    pass
                cur[key] = val
                if key == "wavefront_size":  # last key of a kernel's record
                    if "symbol" in cur:
                        out.append(cur)
                    cur = {}
            pos += len(ELF_AMDGPU)
            k += 1
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(here, "..", "stdpar-nbody_amd", "libnbody_hip.so")
    filters = [a for a in sys.argv[1:] if not a.endswith(".so")]
    ks = kernels(lib)
    names = demangle([k["symbol"].replace(".kd", "") for k in ks])
    seen = set()
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>7} {'waves/SIMD':>10}  kernel")
    for k, n in sorted(zip(ks, names), key=lambda t: t[1]):
        n = re.sub(r"^void ", "", n)
        if n in seen or (filters and not any(f in n for f in filters)):
            continue
        seen.add(n)
        v, a = int(k.get("vgpr_count", 0)), int(k.get("agpr_count", 0))
        tot = (v + a + 7) // 8 * 8
        by_v = min(8, 512 // tot) if tot else 8
        sg = int(k.get("sgpr_count", 0))
        by_s = min(8, 800 // ((sg + 15) // 16 * 16 + 16))
        waves = min(by_v, by_s)
        print(f"{v:5d} {a:5d} {sg:5d} {int(k.get('group_segment_fixed_size', 0)):7d} "
              f"{int(k.get('private_segment_fixed_size', 0)):7d} {waves:9d}{'*' if by_s < by_v else ' '}  {n[:150]}")


if __name__ == "__main__":
    main()
