"""Experiment (libnbody_hip_exp.so, NBODY_K9_TIMELINE=1): when every work item of ONE sweep launch at config 4 started and ended
(s_memrealtime, 100 MHz), what the launch looks like over time, and a list-scheduling what-if with the measured durations:
the order that ran, longest-first by the true durations, and longest-first by predictors a same-step pre-pass could compute
(the group's bounding-box diagonal).  Durations are taken under the contention of the real launch (items in the tail ran alone
and faster), so the what-if is indicative, as the same model was for round 2's kernel."""
import os, sys, ctypes as C, heapq
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _experiments import load_package
nb = load_package()
n, theta = int(os.environ.get("K9_N", "1000000")), 0.5
dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "galaxy", n))
st, t = dev.state(), dev.bvh
for _ in range(int(os.environ.get("K9_STEPS", "3"))):       # a few steps of the evolving system, as the step loop sees it
    dev.bvh_force(theta); dev.accelerate_step()
t.bounding_box(st, dev.stream); t.hilbert_sort(st, dev.stream); t.build_tree(st, dev.stream)
for _ in range(3):
    t.compute_force(st, theta, dev.stream)
dev.sync()
os.environ["NBODY_K9_TIMELINE"] = "1"
t.compute_force(st, theta, dev.stream); dev.sync()
os.environ.pop("NBODY_K9_TIMELINE")
L = nb.lib()
L.nbody_exp_k9_timeline.restype = C.c_longlong
buf = np.zeros(3 * (n // 64 + 4096) * 2, np.uint64)
words = L.nbody_exp_k9_timeline(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
assert words > 0, words
tl = buf[:words].reshape(-1, 3)
ok = (tl[:, 2] >> np.uint64(63)) == 1
block_id = np.arange(len(tl))[ok]                             # launch block index: list (XCD) = id % 8, slot in the list = id // 8
tl = tl[ok]
start, end = tl[:, 0].astype(np.int64), tl[:, 1].astype(np.int64)
group = (tl[:, 2] & np.uint64(0xfffff)).astype(np.int64)
lo = ((tl[:, 2] >> np.uint64(32)) & np.uint64(63)).astype(np.int64); hi = ((tl[:, 2] >> np.uint64(40)) & np.uint64(63)).astype(np.int64)
t0 = start.min()
start, end = (start - t0) * 0.01, (end - t0) * 0.01          # microseconds
dur = end - start
span = end.max()
print(f"n = {n}: {len(dur)} work items ({int(((lo > 0) | (hi < 63)).sum())} of them half groups), launch {span / 1e3:.3f} ms from first start to last end")
print(f"item durations (us): min {dur.min():.0f}  p10 {np.percentile(dur, 10):.0f}  median {np.median(dur):.0f}  p90 {np.percentile(dur, 90):.0f}  max {dur.max():.0f};  sum / 8192 slots = {dur.sum() / 8192 / 1e3:.3f} ms")
ts = np.linspace(0, span, 41)
running = [(int(((start <= x) & (end > x)).sum())) for x in ts]
print("items running over the launch (41 samples):", " ".join(str(r) for r in running))
half = next((x for x, r in zip(ts[::-1], running[::-1]) if r >= 4096), 0.0)
print(f"last moment with >= 4096 items running: {half / 1e3:.3f} ms -> the drain takes {(span - half) / 1e3:.3f} ms; the last item started at {start.max() / 1e3:.3f} ms")

def makespan(order, slots=8192):
    heap = [0.0] * slots
    for i in order:
        heapq.heappush(heap, heapq.heappop(heap) + dur[i])
    return max(heap)

x = dev.download().x                                         # the state is in key order after the sort
diag = np.zeros(len(dur))
for i, (g, a, b) in enumerate(zip(group, lo, hi)):
    p = x[g * 64 + a: min(n, g * 64 + b + 1)]
    diag[i] = np.linalg.norm(p.max(axis=0) - p.min(axis=0)) if len(p) else 0.0
# per-body walk lengths of THIS traversal (a counted launch): what the best possible per-body predictor would know
t.enable_counters(True)
t.compute_force(st, theta, dev.stream); dev.sync()
cnt = t.read(5, dev.stream).reshape(-1, 4).astype(np.int64)
t.enable_counters(False)
body_len = cnt[:, 0] + cnt[:, 3]                              # nodes tested + body entries accepted: the entries the body stands on
gmax, gsum = np.zeros(len(dur)), np.zeros(len(dur))
for i, (g, a, b) in enumerate(zip(group, lo, hi)):
    w = body_len[g * 64 + a: min(n, g * 64 + b + 1)]
    gmax[i], gsum[i] = (w.max(), w.sum()) if len(w) else (0, 0)
print(f"correlation with the duration: the longest walk of the item's bodies {np.corrcoef(gmax, dur)[0, 1]:.3f}, the sum of their walks {np.corrcoef(gsum, dur)[0, 1]:.3f}, "
      f"longest walk x diagonal {np.corrcoef(gmax * (1 + diag), dur)[0, 1]:.3f}")
orders = {"the order that ran (by start time)": np.argsort(start, kind="stable"),
          "longest first by the longest walk of the item's bodies (this step's own counters: an upper bound for a per-body predictor)": np.argsort(-gmax, kind="stable"),
          "longest first by the sum of the bodies' walks": np.argsort(-gsum, kind="stable"),
          "group index order": np.argsort(group * 64 + lo, kind="stable"),
          "longest first, true durations": np.argsort(-dur, kind="stable"),
          "longest first by the group's bounding-box diagonal": np.argsort(-diag, kind="stable"),
          "shortest first (worst case)": np.argsort(dur, kind="stable")}
print(f"list scheduling of the measured durations on 8192 slots (ms); correlation of the diagonal with the duration: {np.corrcoef(diag, dur)[0, 1]:.3f}")
for name, o in orders.items():
    print(f"    {makespan(o) / 1e3:.3f}  {name}")

# What if the items that START LAST were swept as narrower lane ranges?  Slots are idle during the drain, so the extra work is free
# there; a 32-lane sweep takes 0.86 of the 64-lane sweep's steps at config 4, a 16-lane sweep 0.77 (DESIGN_APPENDIX, round 2 data).
slot = block_id // 8
order_ran = np.argsort(start, kind="stable")
print("what if the last fraction f of every XCD's list were cut into k lane ranges (durations x 0.86 for k = 2, x 0.77 for k = 4), list-scheduled in the order that ran:")
for k, factor in ((2, 0.86), (4, 0.77)):
    row = []
    for f in (1 / 32, 1 / 16, 1 / 8, 1 / 4, 1 / 2):
        heap = [0.0] * 8192
        thr = np.quantile(slot, 1 - f)
        for i in order_ran:
            if slot[i] >= thr and lo[i] == 0 and hi[i] == 63:
                for _ in range(k):
                    heapq.heappush(heap, heapq.heappop(heap) + dur[i] * factor)
            else:
                heapq.heappush(heap, heapq.heappop(heap) + dur[i])
        row.append(f"f = 1/{round(1 / f)}: {max(heap) / 1e3:.3f}")
    print(f"    k = {k}: " + "   ".join(row) + f"   (none: {makespan(order_ran) / 1e3:.3f})")
