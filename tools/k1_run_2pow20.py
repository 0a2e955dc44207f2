import sys, os
sys.path.insert(0, "tests")
from conftest import load_package
nb = load_package()
nb.LIB_PATH, nb._lib = os.path.abspath(sys.argv[1]), None
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "galaxy", 1 << 20))
for _ in range(2):
    dev.all_pairs_force()
dev.sync()
