"""How often K1's chunk hand-off makes a wave wait (nbody_all_pairs_status: waits = waves that polled at least once, polls = their
polls), per launch shape of the BASELINE configs — the 8-GPU rank shard of 2^20 above all (1024 target blocks x 16 chunks: one
chunk row per round of resident blocks, where waits are likeliest).  Shipped library.  profiles/<tag>/k1_handoff_polls.txt"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from conftest import load_package
nb = load_package()
cases = [("config 5, one of 8 ranks: 131072-target window of n = 2^20 (first = 3 * 131072)", nb.F64, "galaxy", 1 << 20, 3 << 17, 1 << 17, 6),
         ("config 5, one of 2 ranks: 524288-target window of n = 2^20", nb.F64, "galaxy", 1 << 20, 1 << 19, 1 << 19, 2),
         ("config 5 on one GPU: n = 2^20", nb.F64, "galaxy", 1 << 20, 0, None, 2),
         ("config 2: n = 65536 uniform", nb.F64, "uniform", 65536, 0, None, 50),
         ("float, n = 262144 uniform", nb.F32, "uniform", 262144, 0, None, 5),
         ("n = 8192 galaxy (every block resident at once)", nb.F64, "galaxy", 8192, 0, None, 50)]
for label, dtype, wl, n, first, count, reps in cases:
    dev = nb.DeviceSystem.from_host(nb.build_model(dtype, 3, wl, n))
    st = dev.state(first, count)
    desc = nb.describe_all_pairs(st)
    dev.all_pairs_force(first, count); dev.sync()
    nb.all_pairs_status(dev.stream, clear=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.all_pairs_force(first, count)
    dev.sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    s = nb.all_pairs_status(dev.stream)
    cnt = n if count is None else count
    print(f"{label}\n    {desc}\n    {reps} launches, {ms:.3f} ms each: failed = {s['failed']}, waves that waited = {s['waits']} "
          f"({s['waits'] / reps:.1f} per launch), polls = {s['polls']} ({s['polls'] / max(1, s['waits']):.1f} per waiting wave)")
    dev.close()
