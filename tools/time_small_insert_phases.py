"""Diagnostic (experiments build): where the one-block octree insert of a small system spends its time — wall_clock64() stamps at
the phase boundaries of ot_insert_small_kernel (keys, sort, numbering, cells, deep cells), galaxy, 3D.
    python tools/time_small_insert_phases.py [float]"""
import ctypes, sys
from _experiments import load_package
nb = load_package()
dtype = nb.F32 if (len(sys.argv) > 1 and sys.argv[1] == "float") else nb.F64
lib = nb.lib()
out = (ctypes.c_uint64 * 8)()
for n in (257, 1000, 1024):
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, "galaxy", n))
    for _ in range(3):
        dev.octree_force(0.5)
    dev.sync()
    assert lib.nbody_exp_octree_small_stamps(out) == 0
    s = [out[k] for k in range(6)]
    names = ("keys", "sort", "numbering", "cells", "deep")
    print("n=%d dtype=%d  " % (n, dtype) + "  ".join("%s %.1f us" % (names[k], (s[k + 1] - s[k]) / 100.0) for k in range(5)) +
          "  total %.1f us" % ((s[5] - s[0]) / 100.0), flush=True)
    dev.close()
