"""Experiment (libnbody_hip_exp.so, NBODY_K9_TIMELINE=1): does an item's MEASURED duration carry from one step to the next?

Round 5 left one predictor of a sweep item's duration untried: the duration the item holding the same leading body had in the
LAST step (it carries the union's length and the contention the item met, which geometry and per-body walk lengths do not).
This tool takes the timeline of the sweep (every item's start and end by s_memrealtime) in two consecutive steps of the evolving
config-4 system, maps every item of step s + 1 to the item of step s that held its leading body (through the sort's permutation),
and reports the correlation of the two durations and what a longest-first order by the carried duration would give when the
true durations of step s + 1 are list-scheduled on the 8192 slots — next to the order that ran and the unreachable optimum
(longest-first by the true durations)."""
import os, sys, ctypes as C, heapq
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
n, theta = int(os.environ.get("K9_N", "1000000")), 0.5
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "galaxy", n))
st, t = dev.state(), dev.bvh
L = nb.lib()
L.nbody_exp_k9_timeline.restype = C.c_longlong


def timeline():
    """One traversal of the tree that is built, with the timeline on: (group, lo, hi, start us, duration us) per item."""
    for _ in range(2):
        t.compute_force(st, theta, dev.stream)
    dev.sync()
    os.environ["NBODY_K9_TIMELINE"] = "1"
    t.compute_force(st, theta, dev.stream); dev.sync()
    os.environ.pop("NBODY_K9_TIMELINE")
    buf = np.zeros(3 * (n // 64 + 4096) * 2, np.uint64)
    words = L.nbody_exp_k9_timeline(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
    assert words > 0, words
    tl = buf[:words].reshape(-1, 3)
    tl = tl[(tl[:, 2] >> np.uint64(63)) == 1]
    start, end = tl[:, 0].astype(np.int64), tl[:, 1].astype(np.int64)
    group = (tl[:, 2] & np.uint64(0xfffff)).astype(np.int64)
    lo = ((tl[:, 2] >> np.uint64(32)) & np.uint64(63)).astype(np.int64)
    hi = ((tl[:, 2] >> np.uint64(40)) & np.uint64(63)).astype(np.int64)
    steps = ((tl[:, 2] >> np.uint64(46)) & np.uint64(0x1ffff)).astype(np.int64)   # the sweep's own step count (experiments build)
    t0 = start.min()
    return group, lo, hi, (start - t0) * 0.01, (end - start) * 0.01, steps


def makespan(dur, order, slots=8192):
    heap = [0.0] * slots
    for i in order:
        heapq.heappush(heap, heapq.heappop(heap) + dur[i])
    return max(heap) / 1e3


for _ in range(int(os.environ.get("K9_STEPS", "3"))):
    dev.bvh_force(theta); dev.accelerate_step()
prev = None
for step in range(3):
    t.bounding_box(st, dev.stream); t.hilbert_sort(st, dev.stream); t.build_tree(st, dev.stream)
    perm = t.read(1, dev.stream).astype(np.int64)            # sorted position -> position before this step's sort
    g, lo, hi, start, dur, steps = timeline()
    if prev is not None:
        pg, plo, phi, pdur, psteps = prev
        # duration per BODY POSITION of the previous step: every body takes the duration of the item that held it
        body_dur = np.zeros(n)
        for i in range(len(pg)):
            body_dur[pg[i] * 64 + plo[i]: min(n, pg[i] * 64 + phi[i] + 1)] = pdur[i]
        lead = np.minimum(g * 64 + lo, n - 1)
        carried = body_dur[perm[lead]]                           # the leading body's item of last step
        carried_max = np.array([body_dur[perm[g[i] * 64 + lo[i]: min(n, g[i] * 64 + hi[i] + 1)]].max() for i in range(len(g))])
        moved = float(np.mean(perm[lead] // 64 != g))
        print(f"step {step}: {len(dur)} items; {100 * moved:.1f} % of the leading bodies sat in another group last step")
        print(f"  correlation of an item's duration with last step's duration of its leading body's item: {np.corrcoef(carried, dur)[0, 1]:.3f}; "
              f"with the longest of its bodies' items: {np.corrcoef(carried_max, dur)[0, 1]:.3f}")
        print(f"  list scheduling of THIS step's true durations on 8192 slots (ms): the order that ran {makespan(dur, np.argsort(start, kind='stable')):.3f};  "
              f"longest first by the carried duration {makespan(dur, np.argsort(-carried, kind='stable')):.3f};  "
              f"by the longest carried duration of its bodies {makespan(dur, np.argsort(-carried_max, kind='stable')):.3f};  "
              f"by the true durations {makespan(dur, np.argsort(-dur, kind='stable')):.3f}")
        # the union's LENGTH (steps of the sweep, counted by the kernel) as the predictor: this step's own, and last step's carried per body
        body_steps = np.zeros(n)
        for i in range(len(pg)):
            body_steps[pg[i] * 64 + plo[i]: min(n, pg[i] * 64 + phi[i] + 1)] = psteps[i]
        carried_steps = np.array([body_steps[perm[g[i] * 64 + lo[i]: min(n, g[i] * 64 + hi[i] + 1)]].max() for i in range(len(g))])
        print(f"  steps per item: min {steps.min()} median {int(np.median(steps))} max {steps.max()}; correlation of the duration with the item's own step count "
              f"{np.corrcoef(steps, dur)[0, 1]:.3f}, with last step's count carried by its bodies {np.corrcoef(carried_steps, dur)[0, 1]:.3f}; "
              f"own against carried count {np.corrcoef(steps, carried_steps)[0, 1]:.3f}")
        print(f"  list scheduling, longest first by the own step count {makespan(dur, np.argsort(-steps, kind='stable')):.3f};  by the carried step count "
              f"{makespan(dur, np.argsort(-carried_steps, kind='stable')):.3f}")
        # how much of a duration is the item and how much the moment it ran at: the same tree traversed twice
        g2, lo2, hi2, start2, dur2, steps2 = timeline()
        same = len(g2) == len(g) and (g2 == g).all() and (lo2 == lo).all()
        dur2 = dur2 if len(dur2) == len(dur) else dur
        print(f"  the SAME tree traversed again: correlation of the two runs' durations {np.corrcoef(dur, dur2)[0, 1]:.3f}" + ("" if same else " (items listed in another order)"))
    prev = (g, lo, hi, dur, steps)
    dev.sync()
    t.compute_force(st, theta, dev.stream)
    dev.accelerate_step()
