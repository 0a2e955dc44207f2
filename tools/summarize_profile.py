#!/usr/bin/env python3
"""rocprofv3 output of tools/profile_bench.sh -> the summaries kept under profiles/<round>/:

  bench_n1_kernel_stats.csv             rocprofv3 --kernel-trace --stats: per-kernel calls / total / average duration
  bench_n1_under_rocprof.json           the bench line printed inside that traced run
  bench_n1_pmc_all_pairs_force.json     counters of the dominant kernel's dispatch (SQ pass, FETCH_SIZE pass, WRITE_SIZE pass,
                                        each with the kernel's duration in that pass), STAMPED with what they were taken on:
                                        n, gpus, the kernel description string (nbody_all_pairs_describe) and the hash of the K1
                                        sources.  bench.py reports roofline.traffic / roofline.rocprof only from a summary whose
                                        stamp matches the run's own.

    python tools/summarize_profile.py gpurun_out/prof_r02 profiles/r02
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "all_pairs_force"  # substring of the dominant kernel's name


def source_sha():
    h = hashlib.sha256()
    for f in ("all_pairs.hip", "common.hpp"):
        h.update(open(os.path.join(ROOT, "stdpar-nbody_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def counters(pass_dir):
    """{counter: value} and duration (ns) of the longest dispatch of the dominant kernel in one PMC pass."""
    files = glob.glob(os.path.join(pass_dir, "*", "*counter_collection.csv"))
    if not files:
        return {}, None, None
    per = {}
    for r in csv.DictReader(open(files[0])):
        if KERNEL in r["Kernel_Name"] and "pack" not in r["Kernel_Name"] and "combine" not in r["Kernel_Name"]:
            d = per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "c": {}, "dur": 0})
            d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                d["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not per:
        return {}, None, None
    best = max(per.values(), key=lambda d: (d["dur"], sum(d["c"].values())))
    return best["c"], best["dur"] or None, best["name"]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
    if stats:
        shutil.copy(stats[0], os.path.join(dst, "bench_n1_kernel_stats.csv"))
    # The same trace by launch SHAPE: since round 5 the command also runs configs 2-4, and config 2's K1 is the same instantiation
    # as the headline's (another grid) — rocprofv3's per-name statistics average the two together.
    trace = glob.glob(os.path.join(src, "trace", "*", "*kernel_trace.csv"))
    if trace:
        import collections
        rows = collections.OrderedDict()
        for r in csv.DictReader(open(trace[0])):
            key = (r["Kernel_Name"], "x".join(r["Grid_Size_" + a] for a in "XYZ"), "x".join(r["Workgroup_Size_" + a] for a in "XYZ"))
            rows.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        with open(os.path.join(dst, "bench_n1_kernel_stats_by_grid.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Grid", "Workgroup", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
            for (name, grid, wg), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([name, grid, wg, len(d), sum(d), "%.1f" % (sum(d) / len(d)), min(d), max(d)])
    line = None
    try:
        line = json.loads([l for l in open(os.path.join(src, "trace_bench.json")) if l.startswith("{")][-1])
        json.dump(line, open(os.path.join(dst, "bench_n1_under_rocprof.json"), "w"), indent=1)
    except Exception as ex:
        print("no bench line in the traced run:", ex)
    out = {}
    name = None
    for tag in ("sq", "fetch", "write"):
        c, dur, nm = counters(os.path.join(src, "pmc_" + tag))
        out.update(c)
        out["duration_ns_pmc_" + tag] = dur
        name = name or nm
    if line:
        out["n"] = line["config"]["n_bodies"]
        out["gpus"] = line["n_gpus"]
        out["kernel"] = line["roofline"]["kernel"]
    out["kernel_name_in_profile"] = name
    out["source_sha"] = source_sha()
    # what the profile says by itself: the clock the box sustained during the counted dispatch and the kernel's rate at that clock
    if out.get("GRBM_GUI_ACTIVE") and out.get("duration_ns_pmc_sq") and out.get("n"):
        cycles = out["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
        secs = out["duration_ns_pmc_sq"] * 1e-9
        flops = 20.0 * out["n"] * (out["n"] - 1)
        out["effective_clock_ghz"] = cycles / secs / 1e9
        out["tflops_in_pmc_pass"] = flops / secs / 1e12
        out["frac_of_fp64_peak_at_2p4ghz"] = out["tflops_in_pmc_pass"] / 78.6
        out["frac_of_fp64_peak_at_measured_clock"] = out["tflops_in_pmc_pass"] / (78.6 * out["effective_clock_ghz"] / 2.4)
    if line and line.get("gpu_telemetry"):
        out["sclk_mhz_mean_traced_run"] = line["gpu_telemetry"].get("sclk_mhz_mean")
    out["_note"] = ("per dispatch of the dominant kernel, rocprofv3 --pmc in separate passes (tools/profile_bench.sh); "
                    "duration_ns_* = that dispatch's duration in the pass; FETCH_SIZE / WRITE_SIZE in KB as reported "
                    "(bench.py doubles FETCH_SIZE for gfx950 per MI355X_MICROARCH.md)")
    json.dump(out, open(os.path.join(dst, "bench_n1_pmc_all_pairs_force.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
