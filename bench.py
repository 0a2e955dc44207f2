#!/usr/bin/env python3
"""Headline benchmark: body-steps/s (+ % of gfx950 FP64 vector peak) for 3D double all-pairs on the
synthetic galaxy init, N = 2^20 bodies, on 1/2/4/8 MI355X.

    python bench.py --gpus N --steps K --warmup W        (any N: for N > 1 it starts the N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W            (the driver's form: runs as one of the ranks)

One process per GPU.  A "step" is one pass of the hot path over all bodies: K1 all-pairs force on the
rank's shard of targets (all N sources), K3 leapfrog on the shard, then one RCCL all-gather of the
updated positions over xGMI (nbody_allgather_positions of the C ABI; torch.distributed, backend "nccl", carries the
rendezvous, the barriers and the max-over-ranks of the time).  Total work is fixed as N grows
(strong scaling).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

PyTorch is plumbing here (device memory, streams, the process group); all arithmetic is in
stdpar-nbody_amd/libnbody_hip.so through its C ABI.  The CPU baseline leg times the oracle's
restatement of the reference loop on this box's host cores (reported, not the target).
"""
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before any HIP runtime is loaded: dmabuf IPC only on this driver
import argparse
import ctypes as C
import importlib.util
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
FP32_VECTOR_PEAK_TFLOPS = 157.3  # the same lanes at 2 x 32 bits
FLOP_PER_INTERACTION = 20        # D=3: 3 sub + 5 r2 + 2 (sqrt, *r2) + 1 (+eps) + 3 (m*d) + 3 (/) + 3 (acc)


def load_package():
    path = os.path.join(ROOT, "stdpar-nbody_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location("stdpar_nbody_amd", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["stdpar_nbody_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def fast_oracle():
    """(ctypes library, build string, cores): the oracle's C restatement built with the reference's own CPU flags (-Ofast
    -march=native, ci/run:112-113), OpenMP over every core the cgroup grants.  TEST INFRASTRUCTURE used as the reported CPU
    baseline only — nothing of it is on the measured GPU path."""
    src = os.path.join(ROOT, "oracle", "nbody_oracle.c")
    tmp = tempfile.mkdtemp(prefix="nbody_cpu_")
    so = os.path.join(tmp, "liboracle_fast.so")
    flags = ["-std=c11", "-Ofast", "-march=native", "-fopenmp", "-fPIC", "-shared", "-Wno-unused-function"]
    try:
        subprocess.check_call(["gcc"] + flags + [src, "-o", so, "-lm"], stderr=subprocess.DEVNULL)
        build = "gcc -Ofast -march=native -fopenmp"
    except Exception:
        sys.path.insert(0, ROOT)
        import oracle as O
        O.lib()
        so, build = O.LIB_PATH, "gcc -O2 -fopenmp"
    cores = len(os.sched_getaffinity(0))
    try:  # respect the container's CPU quota (cgroup v2 cpu.max = "<quota> <period>")
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(quota) // int(period)))
    except Exception:
        pass
    os.environ["OMP_NUM_THREADS"] = str(cores)
    return C.CDLL(so), build, cores


def reference_serial_baseline():
    """The REAL reference (oracle/_ref/nbody_ref_d3: /root/reference compiled unmodified, g++ -O2, PSTL's serial backend because the
    image has no TBB headers) timed on ONE host core on a bounded run: all-pairs 3D double galaxy, -n 8192 -s 14 --csv-total, i.e.
    10 warm-up and 4 timed steps (src/all_pairs.h:86-97); None where the binary is absent."""
    exe = os.path.join(ROOT, "oracle", "_ref", "nbody_ref_d3")
    if not os.path.exists(exe):
        return None
    n, steps = 8192, 14
    try:
        out = subprocess.run([exe, "-n", str(n), "-s", str(steps), "--precision", "double", "--algorithm", "all-pairs", "--workload",
                              "galaxy", "--csv-total"], capture_output=True, text=True, timeout=300, check=True).stdout
        total_s = float(out.strip().splitlines()[-1].split(",")[5])
    except Exception as ex:
        return {"failed": str(ex)}
    timed = steps - 10
    return {"value": n * timed / total_s, "unit": "body-steps/s", "cores": 1, "kind": "reference",
            "interactions_per_s": n * (n - 1) * timed / total_s,
            "sample": f"oracle/_ref/nbody_ref_d3 (the unmodified reference, g++ -O2, serial PSTL) -n {n} -s {steps} --csv-total: "
                      f"{timed} timed steps in {total_s:.2f} s"}


def cpu_baseline(n, hs, target_seconds=15.0):
    """Time the oracle's all-pairs loop (reference algorithm, src/all_pairs.h:14-27) on the host cores for a
    bounded sample of target bodies against all n sources.  Built with the reference's own CPU flags
    (-Ofast -march=native, ci/run:112-113) and parallelised over targets with OpenMP on every core."""
    import numpy as np
    L, build, cores = fast_oracle()
    a = np.zeros((n, 3), np.float64)

    def run(count):
        t0 = time.perf_counter()
        rc = L.oracle_all_pairs_force(1, 3, hs.m.ctypes.data_as(C.c_void_p), hs.x.ctypes.data_as(C.c_void_p),
                                      a.ctypes.data_as(C.c_void_p), C.c_double(hs.c), C.c_uint32(n), C.c_uint32(0), C.c_uint32(count))
        assert rc == 0
        return time.perf_counter() - t0

    probe = min(n, 64 * cores)
    t = run(probe)
    count = int(min(n, max(probe, probe * target_seconds / max(t, 1e-6))))
    count = max(cores, count // cores * cores)
    t = run(count)
    return {"value": count / t, "unit": "body-steps/s", "cores": cores, "kind": "port",
            "sample": f"{count} of {n} target bodies x all {n} sources, one force pass, {t:.1f} s, {build}",
            "interactions_per_s": count * (n - 1) / t,
            "reference_serial": reference_serial_baseline()}


def config1_cpu_leg(nb):
    """BASELINE.json configs[0] on the host, WHOLE: all-pairs 2D float, -n 10000 -s 5 (the config that names the CPU path; 10
    executed steps) through the CPU port — force (src/all_pairs.h:14-27) + leapfrog (src/system.h:52-60) per step."""
    import numpy as np
    L, build, cores = fast_oracle()
    hs = nb.build_model(nb.F32, 2, "uniform", 10000)
    n = hs.n
    p = lambda arr: arr.ctypes.data_as(C.c_void_p)
    t0 = time.perf_counter()
    for _ in range(10):
        assert L.oracle_all_pairs_force(0, 2, p(hs.m), p(hs.x), p(hs.a), C.c_double(hs.c), C.c_uint32(n), C.c_uint32(0), C.c_uint32(n)) == 0
        assert L.oracle_accelerate_step(0, 2, p(hs.x), p(hs.v), p(hs.a), p(hs.ao), C.c_double(hs.dt), C.c_uint32(n)) == 0
    ms = (time.perf_counter() - t0) / 10 * 1e3
    assert np.isfinite(hs.x).all()
    return {"cpu_ms_per_step": ms, "cpu_cores": cores, "cpu_kind": "port", "cpu_build": build,
            "cpu_body_steps_per_s": n / (ms * 1e-3), "cpu_sample": "the whole config: 10 steps x 10^8 ordered pairs"}


class Telemetry:
    """Samples the GPU's shader clock and socket power with `rocm-smi` (a child process, a few times per second) while the
    timed region runs, so that the line says at which clock its numbers were measured: the kernel sits on the VALU issue
    ceiling, its rate is proportional to the clock the box sustains under FP64 load.  Best effort — never fails the run."""

    def __init__(self, device_index):
        import threading
        self.dev, self.samples, self._stop = device_index, [], threading.Event()
        self._t = threading.Thread(target=self._loop, daemon=True)
        # under rocprofv3 the tool library is preloaded into every child and initialises the GPU there before `rocm-smi`
        # (a script) re-execs its interpreter, which this pool refuses: no sampling in profiled runs, and a clean
        # environment for the child otherwise
        self.enabled = "rocprof" not in os.environ.get("LD_PRELOAD", "").lower()
        self.env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.upper().startswith(("ROCP", "ROCPROF"))}

    def _loop(self):
        import re
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showclocks", "--showpower", "--json"],
                                     capture_output=True, text=True, timeout=10, env=self.env).stdout
                rec = next(iter(json.loads(out).values()))
                sclk = next((v for k, v in rec.items() if "sclk" in k.lower()), None)
                pwr = next((v for k, v in rec.items() if "power" in k.lower() and "(w)" in k.lower()), None)
                mhz = re.search(r"(\d+)\s*mhz", str(sclk), re.I)
                if mhz:
                    self.samples.append((float(mhz.group(1)), float(pwr) if pwr not in (None, "N/A") else None))
            except Exception:
                pass
            self._stop.wait(0.25)

    def __enter__(self):
        if self.enabled:
            self._t.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self.enabled:
            self._t.join(timeout=15)

    def summary(self):
        if not self.samples:
            return None
        clk = [c for c, _ in self.samples]
        pw = [p for _, p in self.samples if p is not None]
        return {"source": "rocm-smi --showclocks --showpower, sampled during the timed steps", "samples": len(clk),
                "sclk_mhz_mean": sum(clk) / len(clk), "sclk_mhz_min": min(clk), "sclk_mhz_max": max(clk),
                "socket_power_w_mean": sum(pw) / len(pw) if pw else None}


def k1_source_sha():
    """Identity of the K1 kernel source: profiles are stamped with it, and evidence taken from another source is dropped."""
    import hashlib
    h = hashlib.sha256()
    for f in ("all_pairs.hip", "common.hpp"):
        h.update(open(os.path.join(ROOT, "stdpar-nbody_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def committed_evidence(world, n, kernel_desc):
    """(traffic bytes per launch, rocprof evidence dict) from the newest committed PMC summary under profiles/ whose stamp —
    kernel description string and source hash — matches what THIS run launches; (None, None) otherwise.  rocprofv3 counters
    cannot be collected from inside the timed run; what can be done is to refuse numbers taken with a different kernel."""
    import glob
    sha = k1_source_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_n1_pmc_all_pairs_force.json")), reverse=True):
        try:
            c = json.load(open(path))
        except Exception:
            continue
        if c.get("n") != n or c.get("gpus") != world or c.get("kernel") != kernel_desc or c.get("source_sha") != sha:
            continue
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0                        # summed over the 8 XCDs
        prof_s = c["duration_ns_pmc_sq"] * 1e-9                    # kernel duration inside the profiled pass
        traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0  # KB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
        rel = os.path.relpath(path, ROOT)
        return traffic, {
            "source": rel + " (rocprofv3 --pmc, separate passes; stamped with this kernel and source hash)",
            "valu_busy_frac": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * 1024),   # quad-cycles -> cycles, 1024 SIMDs
            "valu_insts_per_wave_pair": c["SQ_INSTS_VALU"] / (n * n / 64.0),
            "effective_clock_ghz": cycles / prof_s / 1e9,
            "hbm_gbps": traffic / prof_s / 1e9, "hbm_peak_gbps": 8000.0,
        }
    return None, None


def source_sha(*files):
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "stdpar-nbody_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def committed_config_evidence(name, sha):
    """The newest committed profiles/r*/pmc_<name>.json (tools/pmc_configs.py: rocprofv3 --pmc, separate passes, one dispatch of the
    config's dominant kernel) whose source stamp matches the library this run loaded; None otherwise."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_%s.json" % name)), reverse=True):
        try:
            c = json.load(open(path))
        except Exception:
            continue
        if c.get("source_sha") != sha:
            continue
        cycles = c["GRBM_GUI_ACTIVE"] / 8.0                                  # summed over the 8 XCDs
        busy = c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * 1024)              # quad-cycles -> cycles, 1024 SIMDs
        lanes = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0) if c.get("SQ_THREAD_CYCLES_VALU") else None
        traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None
        # rocprofv3's VALUBusy formula prices every VALU instruction at 4 cycles (SQ_ACTIVE_INST_VALU is in quad-cycles).  That is the
        # issue cost of an f64 instruction; an f32 instruction issues in 2.6 (profiles/r01_valu_rates_microbench.txt), so for a kernel
        # of f32 instructions the formula exceeds 1 and is no fraction: the raw ratio is published under its own name and the
        # fraction is withheld
        is_frac = busy <= 1.0
        return {"source": os.path.relpath(path, ROOT) + " (rocprofv3 --pmc, separate passes; stamped with this library's source hash)",
                "kernel": c.get("kernel"), "valu_busy_frac": busy if is_frac else None,
                "valu_active_quad_cycles_x4_per_simd_cycle": busy,
                "valu_busy_note": None if is_frac else "SQ_ACTIVE_INST_VALU x 4 / (cycles x SIMDs) counts 4 cycles per instruction; f32 "
                                                        "instructions issue in ~2.6, so this ratio is not a fraction for an f32 kernel",
                "valu_lanes_active_frac": lanes,
                "valu_issue_frac": busy * lanes if (lanes is not None and is_frac) else None,
                "valu_insts": c.get("SQ_INSTS_VALU"), "salu_insts": c.get("SQ_INSTS_SALU"),
                "hbm_traffic_bytes_per_launch": traffic, "duration_ms_in_profiled_pass": c.get("duration_ns", 0) / 1e6 or None}
    return None


def other_configs(nb, torch):
    """BASELINE.json configs[1..3] on this GPU, after the headline's clock has stopped (the reference's own matrix, ci/benchmark:63-98,
    runs its algorithms back to back the same way).  Each as the CLI runs it: the step recorded once and replayed (host/drivers.hpp),
    10 warm-up steps and then the timed ones; the dominant kernel's average launch time from HIP events on the context's stream
    around the call that launches it, in further steps of the same evolving system."""
    out = []

    def events_around(dev, fn, reps, between=None):
        st = torch.cuda.ExternalStream(dev.stream)
        pairs = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            fn()
            b.record(st)
            pairs.append((a, b))
            if between:
                between()
        dev.sync()
        return sum(a.elapsed_time(b) for a, b in pairs) / reps

    def replayed(dev, record, steps, warm=10):
        g = nb.StepGraph(dev, record)
        for _ in range(warm):
            g.launch()
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(steps - warm):
            g.launch()
        dev.sync()
        ms = (time.perf_counter() - t0) / (steps - warm) * 1e3
        g.close()
        return ms

    # configs[0]: all-pairs 2D float, -n 10000 -s 5 (uniform): the config that names the CPU path and the reference matrix's
    # sequential size (ci/benchmark:64,137).  -s 5 in the default mode executes max(steps, warm-up) = 10 steps: those ten,
    # replayed from the recorded step after one untimed launch.  Both legs: the GPU's, and the CPU port over the WHOLE config.
    n = 10000
    dev = nb.DeviceSystem.from_host(nb.build_model(nb.F32, 2, "uniform", n))
    dev.all_pairs_force(); dev.sync()
    g = nb.StepGraph(dev, lambda: (dev.all_pairs_force(), dev.accelerate_step()))
    g.launch(); dev.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        g.launch()
    dev.sync()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    g.close()
    k_ms = events_around(dev, dev.all_pairs_force, 20, between=dev.accelerate_step)
    tf = 14.0 * n * (n - 1) / (k_ms * 1e-3) / 1e12          # 14 flop per ordered pair in 2D
    entry = {"workload": "all-pairs 2D float, -n 10000 -s 5, uniform, 1 GPU: the 10 steps the default mode executes, recorded step",
             "ms_per_step": ms, "body_steps_per_s": n / (ms * 1e-3), "kernel": nb.describe_all_pairs(dev.state()), "avg_kernel_ms": k_ms,
             "avg_kernel_how": "HIP events around nbody_all_pairs_force (pre-pass + K1) in 20 further steps",
             "bound": "valu_fp32", "achieved": tf, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_VECTOR_PEAK_TFLOPS,
             "note": "157 target blocks x 16 chunks on 1024 block slots: a launch-shape loss, not a kernel loss"}
    dev.close()
    try:
        entry.update(config1_cpu_leg(nb))
    except Exception as ex:
        entry["cpu_failed"] = str(ex)
    out.append(entry)

    # configs[1]: all-pairs 3D double, -n 65536 -s 100 (no workload flag: the reference's default, uniform)
    n = 65536
    dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "uniform", n))
    dev.all_pairs_force(); dev.sync()
    ms = replayed(dev, lambda: (dev.all_pairs_force(), dev.accelerate_step()), 100)
    k_ms = events_around(dev, dev.all_pairs_force, 20, between=dev.accelerate_step)
    desc = nb.describe_all_pairs(dev.state())
    hand = nb.all_pairs_status(dev.stream)
    tf = FLOP_PER_INTERACTION * n * (n - 1) / (k_ms * 1e-3) / 1e12
    out.append({"workload": "all-pairs 3D double, -n 65536 -s 100, uniform (the default workload), 1 GPU: steps 11-100 of the recorded step",
                "ms_per_step": ms, "body_steps_per_s": n / (ms * 1e-3), "kernel": desc, "avg_kernel_ms": k_ms,
                "avg_kernel_how": "HIP events around nbody_all_pairs_force (the pre-pass launch, ~6 us, + K1) in steps 101-120",
                "bound": "valu_fp64", "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TFLOPS,
                "handoff": {"failed": hand["failed"], "waves_that_waited": hand["waits"], "polls": hand["polls"]}})
    dev.close()

    # configs[2]: all-pairs-collapsed 3D float, -n 262144 (uniform), -s 20
    n = 262144
    dev = nb.DeviceSystem.from_host(nb.build_model(nb.F32, 3, "uniform", n))
    dev.all_pairs_collapsed_force(); dev.sync()
    ms = replayed(dev, lambda: (dev.all_pairs_collapsed_force(), dev.accelerate_step()), 20)
    k_ms = events_around(dev, dev.all_pairs_collapsed_force, 10, between=dev.accelerate_step)
    tf = FLOP_PER_INTERACTION * n * (n - 1) / (k_ms * 1e-3) / 1e12
    ev = committed_config_evidence("k2_config3", source_sha("all_pairs.hip", "common.hpp"))
    out.append({"workload": "all-pairs-collapsed 3D float, -n 262144 -s 20, uniform, 1 GPU: steps 11-20 of the recorded step",
                "ms_per_step": ms, "body_steps_per_s": n / (ms * 1e-3), "kernel": "all_pairs_collapsed_stream_kernel<8> (float 3D: a wave owns 16 targets for its source chunk; + collapsed_reset_pack_kernel)",
                "avg_kernel_ms": k_ms, "avg_kernel_how": "HIP events around nbody_all_pairs_collapsed_force in steps 21-30",
                "bound": "valu_fp32", "achieved": tf, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_VECTOR_PEAK_TFLOPS,
                "rocprof": ev})
    dev.close()

    # configs[3]: bvh 3D double, -n 1000000 --workload galaxy (theta 0.5), -s 20
    n, theta = 1000000, 0.5
    dev = nb.DeviceSystem.from_host(nb.build_model(nb.F64, 3, "galaxy", n))
    dev.bvh_force(theta); dev.accelerate_step(); dev.sync()
    ms = replayed(dev, lambda: (dev.bvh_force(theta), dev.accelerate_step()), 20)
    st, t = dev.state(), dev.bvh
    walk = lambda: t.compute_force(st, theta, dev.stream)
    rest = lambda: (dev.accelerate_step(), t.bounding_box(st, dev.stream), t.hilbert_sort(st, dev.stream), t.build_tree(st, dev.stream))
    rest()
    k_ms = events_around(dev, walk, 10, between=rest)
    t.enable_counters(True)      # one counted traversal of the state the last timed one saw: node tests, leaf visits, accepted entries
    walk(); dev.sync()
    cnt = t.read(5, dev.stream).astype("float64").sum(axis=0)
    t.enable_counters(False)
    node_tests, leaf_visits, terms = cnt[0], cnt[1], cnt[2] + cnt[3]
    cold_bytes = node_tests * 40.0 + leaf_visits * 64.0      # 40 B per node test (monopole + width), 2 x 32 B per leaf
    ev = committed_config_evidence("k9_config4", source_sha("bvh.hip", "common.hpp"))
    out.append({"workload": "bvh 3D double, -n 1000000 -s 20 --workload galaxy --theta 0.5, 1 GPU: steps 11-20 of the recorded step",
                "ms_per_step": ms, "body_steps_per_s": n / (ms * 1e-3),
                "kernel": "bvh_force_sweep_isa_kernel<double,3> (+ bvh_items_kernel; K4-K8 and K3 are the rest of the step)",
                "avg_kernel_ms": k_ms, "avg_kernel_how": "HIP events around nbody_bvh_compute_force in steps 21-30 of the evolving system (the work-item "
                                                                "kernel 36 us + the sweep; + a 22-us threshold rewrite, which an eager traversal of a tree "
                                                                "that has been recorded always makes)",
                "bound": "valu_fp64", "node_tests_per_s": node_tests / (k_ms * 1e-3), "node_tests_per_body": node_tests / n,
                "force_terms_per_body": terms / n,
                # frac means ONE thing in this list: achieved / peak.  Algorithmic flop of a traversal: 20 per force term (K1's pair
                # term) + 11 per node test (3 sub, 3 mul + 2 add for d^2, the product theta^2 * d^2, the compare, the
                # square of the width: src/bvh.h:246-248, 297-308)
                "achieved": (20.0 * terms + 11.0 * node_tests) / (k_ms * 1e-3) / 1e12, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (20.0 * terms + 11.0 * node_tests) / (k_ms * 1e-3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "flop_model": "20 x force terms + 11 x node tests per body",
                "valu_issue_frac": ev["valu_issue_frac"] if ev else None,
                "valu_issue_frac_is": "VALU busy x lanes active (rocprofv3 PMC of this kernel): the share of the vector issue slots that do a "
                                      "body's work — what actually bounds a divergent tree walk; null unless a committed PMC summary carries "
                                      "this library's source hash",
                "rocprof": ev,
                "not_the_bound": {"hbm_cold_bytes_per_s": cold_bytes / (k_ms * 1e-3), "hbm_peak_bytes_per_s": 8.0e12,
                                  "why": "SURVEY 8(d) prices K9 as if every node test fetched its 40 B from HBM: that is %.1f TB/s here, above "
                                         "the 8 TB/s peak.  The sweep reads a record ONCE PER WAVE through the scalar cache (64 bodies share "
                                         "it) and the tree (92 MB) lives in L2 / MALL: measured HBM traffic is ~1.3 GB per traversal, 2-3 %% "
                                         "of peak.  What bounds the kernel is instruction issue: ~40 instructions per step of a wave, of which "
                                         "the lanes of a wave use about half" % (cold_bytes / (k_ms * 1e-3) / 1e12)}})
    dev.close()
    return out


def visible_gpus():
    """Number of HIP devices this process could use, without creating a HIP context (torch.cuda.device_count() reads the
    driver's device list only)."""
    import torch
    return torch.cuda.device_count()


def launch_ranks(args):
    """`python bench.py --gpus N` as typed: this parent never touches a GPU; it starts N fresh rank processes through
    torch.distributed.run (one per GPU, RCCL rendezvous on 127.0.0.1), relays rank 0's single JSON line to stdout and
    everything else to stderr, and exits with the launcher's status."""
    import socket
    have = visible_gpus()
    share = os.environ.get("NBODY_BENCH_SHARE_GPU") == "1"  # rehearsal: all ranks on device 0 (see main)
    if have < (1 if share else args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) are visible to this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env["NBODY_BENCH_CHILD"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.gpus), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--bodies", str(args.n)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.exchange != "nbody":
        cmd += ["--exchange", args.exchange]
    # The ranks run in their own process group under a wall-clock limit: a rank stuck in a collective (a peer that died
    # before ncclCommInitRank, a lost link) must end as a non-zero exit of this command, not as a hang of the caller.
    limit = float(os.environ.get("NBODY_BENCH_LAUNCH_TIMEOUT", "0")) or (600.0 + 3.0 * (args.steps + args.warmup) * (args.n / (1 << 20)) ** 2)
    import signal
    import threading
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    lines = []

    def relay():
        for out in proc.stdout:
            txt = out.strip()
            if txt.startswith("{") and '"metric"' in txt:
                lines.append(txt)
            elif txt:
                print(txt, file=sys.stderr)

    reader = threading.Thread(target=relay, daemon=True)
    reader.start()
    try:
        rc = proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the ranks did not finish within {limit:.0f} s (NBODY_BENCH_LAUNCH_TIMEOUT): killing them", file=sys.stderr)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)   # the exact process group started above
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        raise SystemExit(124)
    reader.join(timeout=10)
    if rc == 0 and lines:
        print(lines[-1], flush=True)
    elif rc == 0:
        rc = 1
        print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bodies", dest="n", type=int, default=1 << 20,
                    help="bodies (default 2^20, the BASELINE.json metric config); `--n` would collide with torchrun option prefixes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE.json configs[1..3] (the `configs` list of the line)")
    ap.add_argument("--exchange", choices=("nbody", "torch"), default="nbody",
                    help="the per-step all-gather of positions: nbody = the library's own collective (nbody_allgather_positions, "
                         "RCCL behind the C ABI; what the metric is about, and the only form that earns a normal line); torch = "
                         "torch.distributed's all_gather, for A/B runs only — the line is marked")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    force_dist = os.environ.get("NBODY_BENCH_FORCE_DIST") == "1"
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        launch_ranks(args)  # does not return

    # stdout carries exactly ONE line, the JSON: everything else written to fd 1 by this process or its libraries (RCCL
    # prints a version banner there when a communicator is created) goes to stderr until the line is printed
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus {args.gpus}` (self-launching) "
                         f"or with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # NBODY_BENCH_SHARE_GPU=1: a REHEARSAL of the multi-rank code path on a box with one GPU — every rank uses device 0,
    # the process group is gloo and the position shards are staged through the host (RCCL refuses two ranks on one device).
    # The line it prints is marked as such and is not a measurement.
    share_gpu = os.environ.get("NBODY_BENCH_SHARE_GPU") == "1" and world > 1
    if share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if share_gpu else dev  # where the small control reductions live
    # NBODY_BENCH_FORCE_DIST=1 exercises the whole multi-rank path (launcher, process group, communicator, barrier,
    # all-gather, max-reduce) even with one rank
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    nb = load_package()
    par = nb.parallel

    # Fault injection for the tests of the failure exits below (tests/test_gpu_multi.py): "<kind>:<rank>" with kind in
    # comm_create (the communicator cannot be created on that rank), stale_exchange (that rank's exchange delivers nothing from
    # the start: it takes part in the collective and throws the result away), stale_late (… from the first timed step on: only
    # the end-of-run checks can see it), corrupt_a (one acceleration row of that rank is altered before the bitwise check), nan_a (one
    # component outside the sampled windows becomes NaN: what a failed K1 chunk hand-off leaves behind).
    fault_kind, _, fault_rank = os.environ.get("NBODY_BENCH_FAULT", "").partition(":")
    fault = fault_kind if fault_kind and int(fault_rank or 0) == rank else None

    def fail_all(reason):
        """Every rank leaves with a non-zero status and the reason on stderr; no result line is printed."""
        print(f"bench.py: rank {rank}: {reason}", file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(3)   # every rank reaches the same verdict (the checks are collective), so nobody is left in a collective

    # The data-path collective is the library's own (nbody_allgather_positions, RCCL behind the C ABI); torch.distributed is the
    # launcher's side: it carries the RCCL unique id to the ranks, the barriers and the max-over-ranks of the time.  If the
    # communicator cannot be created, or its exchange fails a cross-rank check, THE RUN FAILS (non-zero exit, no line): a line
    # from another exchange would credit a collective that did not work.  `--exchange torch` selects torch.distributed's
    # all_gather explicitly, and the line says so.
    comm, comm_note = None, None
    if share_gpu:
        comm_note = "REHEARSAL: %d ranks share one GPU, gloo group, shards staged through the host" % world
    elif use_dist and args.exchange == "torch":
        comm_note = "--exchange torch: torch.distributed all_gather over padded shards (NOT the library's collective)"
    elif use_dist:
        # 1. vote BEFORE any rank enters the collective ncclCommInitRank: a rank that cannot load RCCL or has no device would
        #    otherwise leave the others blocked in it
        err = ""
        try:
            ok = 1 if nb.Comm.rccl_version() > 0 else 0
            if not ok:
                err = "RCCL cannot be loaded (nbody_comm_rccl_version() == 0): " + nb.lib().nbody_last_error().decode()
            nb.device_info(local_rank)
        except Exception as ex:
            ok, err = 0, str(ex)
        if fault == "comm_create":
            ok, err = 0, "injected fault: communicator creation refused on this rank"
        flag = torch.tensor([ok], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            fail_all("nbody_comm cannot be created on every rank (%s); no fallback — use --exchange torch to run without it"
                     % (err or "another rank reported the failure"))
        # 2. the unique id, from rank 0 to everybody
        try:
            box = [nb.Comm.unique_id() if rank == 0 else None]
        except Exception as ex:
            box, err = [None], str(ex)
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            fail_all("nbody_comm_get_unique_id failed on rank 0 (%s)" % (err or "see rank 0"))
        # 3. the collective creation, then a second vote on its outcome
        try:
            comm = nb.Comm(world, rank, box[0], local_rank)
            ok = 1
        except Exception as ex:
            ok, err = 0, str(ex)
        flag = torch.tensor([ok], dtype=torch.int32, device=red_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            fail_all("nbody_comm_create failed (%s)" % (err or "on another rank"))

    # synthetic galaxy init with the product's host generator (host/models.hpp == src/models.h:112-136)
    hs = nb.build_model(nb.F64, 3, "galaxy", args.n)
    n = hs.n
    sim = par.ShardedAllPairs(hs, rank, world, torch_device=dev, force_exchange=use_dist, comm=comm)
    kernel_desc = nb.describe_all_pairs(sim.state())

    if fault in ("stale_exchange", "stale_late"):
        real_exchange = sim.exchange_positions

        def stale_exchange():   # takes part in the collective (nobody hangs), then puts the other ranks' old rows back
            keep = sim.x.clone()
            real_exchange()
            keep[sim.first:sim.first + sim.count] = sim.x[sim.first:sim.first + sim.count]
            sim.x.copy_(keep)

        if fault == "stale_exchange":
            sim.exchange_positions = stale_exchange

    def ranks_agree():
        """Two checksums of the full x, compared across the ranks (min == max): a rank whose exchange did not deliver keeps
        stale rows, and every number below would mean nothing."""
        wts = torch.arange(1, sim.x.numel() + 1, dtype=torch.float64, device=dev).reshape(sim.x.shape)
        sums = torch.stack([sim.x.sum(dtype=torch.float64), (sim.x * wts).sum(dtype=torch.float64)]).to(red_dev)
        lo, hi = sums.clone(), sums.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return torch.equal(lo, hi), sums

    for _ in range(max(args.warmup, 1) if use_dist else args.warmup):
        sim.step()
    exchange_check = None
    if use_dist:
        same, sums = ranks_agree()   # collective: every rank takes the same branch
        if not same:
            fail_all("positions differ between ranks after the warm-up exchange (%s; this rank's checksums %s): the run is void"
                     % (sim.describe().split("exchange: ")[-1], sums.tolist()))
        dist.barrier()
    if fault == "stale_late":
        sim.exchange_positions = stale_exchange
    torch.cuda.synchronize()
    telemetry = Telemetry(local_rank) if rank == 0 else None
    if telemetry:
        telemetry.__enter__()
    t0 = time.perf_counter()
    force_events, xchg_events = [], []
    for _ in range(args.steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        sim.step(force_events=(ev[0], ev[1]), exchange_events=(ev[2], ev[3]))
        force_events.append((ev[0], ev[1]))
        xchg_events.append((ev[2], ev[3]))
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if telemetry:
        telemetry.__exit__()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant kernel: K1.  Algorithmic flops per launch = 20 x (targets of this rank) x (n - 1).  HIP events on the stream
    # the kernels are launched on (torch's current stream is handed to the C ABI).
    k1_ms = sum(a.elapsed_time(b) for a, b in force_events) / max(1, len(force_events))
    xchg_ms = sum(a.elapsed_time(b) for a, b in xchg_events) / max(1, len(xchg_events))
    flops_per_launch = FLOP_PER_INTERACTION * sim.count * (n - 1)
    achieved = flops_per_launch / (k1_ms * 1e-3) / 1e12 if k1_ms > 0 else 0.0

    # ---- after the clock has stopped: the multi-rank run verifies itself -------------------------------------------------
    per_rank, bitwise = None, None
    if use_dist:
        # (a) every rank still holds the same x after the LAST exchange
        same, sums = ranks_agree()
        if not same:
            fail_all("positions differ between ranks after the timed steps (this rank's checksums %s): the run is void" % sums.tolist())
        exchange_check = ("full x identical on all %d rank(s) after the warm-up exchange and after the last timed step "
                          "(2 checksums, min == max over ranks)" % world)
        # (b) per-rank kernel and exchange times (HIP events on the launching stream)
        mine = torch.tensor([k1_ms, xchg_ms], dtype=torch.float64, device=red_dev)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        k1s, xgs = [float(t[0]) for t in allt], [float(t[1]) for t in allt]
        per_rank = {"k1_ms": {"min": min(k1s), "max": max(k1s), "by_rank": k1s},
                    "allgather_ms": {"min": min(xgs), "max": max(xgs), "by_rank": xgs}}
        # (c) bitwise_vs_single: the design guarantees that a target's sum does not depend on the shard window or the world
        # size.  Every rank recomputes K1 on its shard from the positions it holds NOW (no K3), then rank 0 computes a window of
        # WIN targets in the middle of each rank's shard from ITS OWN x and compares with that rank's rows, bit for bit.
        WIN = 4096
        sim.force_phase()
        torch.cuda.synchronize()
        if fault == "corrupt_a":
            sim.a[sim.count // 2] += 1e-9
        if fault == "nan_a":
            sim.a[sim.count // 3, 1] = float("nan")
        # a failed chunk hand-off of K1 on ANY rank (sticky status word of its stream; its rows are NaN) voids the run, and so
        # does a NaN in any rank's accelerations whatever its origin: the windows below only sample the shards
        hand = nb.all_pairs_status(sim._stream(), check=False)
        bad = torch.tensor([1 if (hand["failed"] or bool(torch.isnan(sim.a).any().item())) else 0], dtype=torch.int32, device=red_dev)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()) != 0:
            fail_all("bitwise_vs_single: a rank holds NaN accelerations or reports a failed K1 chunk hand-off (this rank: %s): the run is void"
                     % ("hand-off failed at block %d chunk %d" % (hand["block"], hand["chunk"]) if hand["failed"] else
                        ("NaN in a" if bool(torch.isnan(sim.a).any().item()) else "clean")))
        scratch = torch.empty((WIN, 3), dtype=sim.a.dtype, device=dev)
        checked, mismatch, k1_failed, window_desc = [], [], "", None
        for p, (pf, pe) in enumerate(sim.shards):
            cnt = min(WIN, pe - pf)
            off = (pe - pf - cnt) // 2
            rows = torch.zeros((WIN, 3), dtype=sim.a.dtype, device=red_dev)
            if rank == p:
                rows[:cnt] = sim.a[off:off + cnt].to(red_dev)
            dist.broadcast(rows, src=p)
            if rank == 0 and cnt > 0:
                st = sim.state()
                st.first, st.count = pf + off, cnt
                st.a = st.v = st.ao = scratch.data_ptr()   # K1 writes a only; v/ao are not touched by it
                window_desc = nb.describe_all_pairs(st)  # a short window: one chunk per block, whatever shape the ranks' launches had
                rc = nb.lib().nbody_all_pairs_force(C.byref(st), C.c_void_p(sim._stream()))
                if rc:  # no exit here: the other ranks wait in the broadcast below; the verdict carries the failure to all of them
                    k1_failed = k1_failed or nb.lib().nbody_last_error().decode()
                    continue
                torch.cuda.synchronize()
                checked.append(p)
                if not torch.equal(scratch[:cnt].to(red_dev), rows[:cnt]):
                    mismatch.append(p)
        verdict = torch.tensor([-1 if k1_failed else len(mismatch)], dtype=torch.int32, device=red_dev)
        dist.broadcast(verdict, src=0)
        if int(verdict.item()) < 0:
            fail_all("bitwise_vs_single: rank 0's K1 on a window failed" + (": " + k1_failed if rank == 0 else " (reason on rank 0)"))
        if int(verdict.item()) != 0:
            fail_all("bitwise_vs_single failed: rank 0's K1 on a %d-target window of rank(s) %s differs from the rows those ranks "
                     "computed (stale positions on a rank, or a window-dependent sum): the run is void" % (WIN, mismatch if rank == 0 else "?"))
        bitwise = {"equal": True, "checked_ranks": checked, "targets_per_window": WIN,
                   "launch_shapes": {"rank_shard": kernel_desc, "window": window_desc},
                   "how": "after the timed steps every rank recomputed K1 on its shard; rank 0 recomputed a window in the middle of "
                          "each rank's shard from its own x and compared the rows bit for bit"}

    # K1's chunk hand-off over the whole run of this rank's stream: failed must be False; waits / polls say how often a block had to
    # wait for its turn (normally never)
    hand = nb.all_pairs_status(sim._stream(), check=False)
    if hand["failed"] and not use_dist:
        print("bench.py: K1 chunk hand-off failed (block %d, chunk %d): no result" % (hand["block"], hand["chunk"]), file=sys.stderr)
        os._exit(3)
    handoff = {"failed": hand["failed"], "waves_that_waited": hand["waits"], "polls": hand["polls"]}
    if use_dist:
        # the status is read AFTER the window recomputations above (rank 0 has launched K1 again since the `bad` vote), so the
        # failure flag travels with the counts: a sticky failure on any rank ends every rank without a result line
        mine = torch.tensor([hand["waits"], hand["polls"], 1 if hand["failed"] else 0], dtype=torch.int64, device=red_dev)
        allh = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allh, mine)
        failed = [bool(int(t[2])) for t in allh]
        if any(failed):
            fail_all("a K1 chunk hand-off failed on rank(s) %s after the timed steps (this rank: %s): the run is void"
                     % ([r for r, f in enumerate(failed) if f],
                        "block %d chunk %d" % (hand["block"], hand["chunk"]) if hand["failed"] else "clean"))
        handoff = {"failed": failed, "waves_that_waited": [int(t[0]) for t in allh], "polls": [int(t[1]) for t in allh], "by": "rank"}

    if rank == 0:
        value = n * args.steps / elapsed
        whole_job_tflops = FLOP_PER_INTERACTION * n * (n - 1) * args.steps / elapsed / 1e12
        traffic, evidence = committed_evidence(world, n, kernel_desc)
        tele = telemetry.summary() if telemetry else None
        row = 3 * 8  # bytes of one position record (3D double)
        out = {
            "metric": "body-steps/sec + %FP64 peak, 3D double all-pairs N=2^20 at 1/2/4/8 GPUs",
            **({"rehearsal": comm_note} if share_gpu else {}),
            **({"not_the_library_collective": comm_note} if (use_dist and not share_gpu and comm is None) else {}),
            "value": value, "unit": "body-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "all-pairs 3D double, galaxy init (mt19937 seed 42), n=%d" % n, "n_bodies": n,
                       "parallelism": "bodies sharded over %d GPU(s), RCCL all-gather(x) per step" % world if world > 1
                       else "single GPU", "split": sim.describe()},
            "pct_fp64_peak": 100.0 * whole_job_tflops / (FP64_VECTOR_PEAK_TFLOPS * world),
            "rccl_world": ({"world": dist.get_world_size(), "backend": dist.get_backend(),
                            "data_path": ("nbody_comm (ncclCommInitRank) world %d, RCCL %d" % (comm.world, nb.Comm.rccl_version())
                                          if comm is not None else comm_note),
                            "exchange": "nbody" if comm is not None else ("rehearsal" if share_gpu else args.exchange),
                            "exchange_check": exchange_check, "bitwise_vs_single": bitwise, "per_rank": per_rank}
                           if use_dist else None),
            "shards": [e - f for f, e in sim.shards],
            "allgather": {"sent_bytes_per_rank_per_step": sim.count * row, "gathered_bytes_per_step": n * row,
                          "avg_ms": xchg_ms if use_dist else None,
                          "how": sim.describe().split("exchange: ")[-1] if use_dist else "none (single GPU, no exchange)"},
            "gpu_telemetry": tele,
            "roofline": {"bound": "valu_fp64", "kernel": kernel_desc, "achieved": achieved,
                         "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_VECTOR_PEAK_TFLOPS,
                         "traffic": traffic, "avg_launch_ms": k1_ms, "source_sha": k1_source_sha(),
                         # the same fraction against the peak at the shader clock the box sustained (socket-power capped)
                         "frac_at_measured_clock": (achieved / (FP64_VECTOR_PEAK_TFLOPS * tele["sclk_mhz_mean"] / 2400.0)
                                                    if tele else None),
                         "rocprof": evidence,
                         "handoff": handoff,
                         "note": "north_star forbids MFMA for this path; bound is the FP64 vector pipe "
                                 "(20 algorithmic flop per ordered pair, SURVEY 8d); traffic/rocprof are null unless a "
                                 "committed PMC summary carries this run's kernel description and source hash"},
        }
        if world == 1 and not args.no_other_configs:
            try:   # BASELINE.json configs[1..3], after the headline's clock has stopped: reported extras, never lose the line over them
                out["configs"] = other_configs(nb, torch)
            except Exception as ex:
                out["configs"] = [{"workload": "configs[1..3]", "failed": str(ex)}]
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, hs)
            except Exception as ex:  # the baseline is a reported extra; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "body-steps/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
