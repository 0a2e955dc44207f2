import sys, os
sys.path.insert(0, "tests")
from conftest import load_package
nb = load_package()
nb.LIB_PATH, nb._lib = os.path.abspath(sys.argv[1]), None
dt = nb.F32 if sys.argv[2] == "float" else nb.F64
n = int(sys.argv[3])
dev = nb.DeviceSystem.from_host(nb.build_model(dt, 3, "uniform", n))
for _ in range(20):
    dev.all_pairs_force()
dev.sync()
