"""Tools that use the tuning switches (NBODY_K1_CHUNKS, NBODY_K2_CFG, NBODY_K9_MODE/ORDER/LDS, NBODY_OT_FORM) load the
-DNBODY_EXPERIMENTS build of the library: the shipped libnbody_hip.so never reads the environment."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def load_package(experiments=True):
    from conftest import load_package as _load
    nb = _load()
    if experiments:
        pkg = os.path.join(ROOT, "stdpar-nbody_amd")
        subprocess.check_call(["make", "-s", "-C", pkg, "experiments"])
        nb.LIB_PATH = os.path.join(pkg, "libnbody_hip_exp.so")
        nb._lib = None
    return nb
