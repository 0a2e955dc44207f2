#!/usr/bin/env python3
"""rocprofv3 PMC summaries of the dominant kernels of BASELINE configs 3 (K2) and 4 (K9), STAMPED with the hash of the sources they
were built from — bench.py's `configs` entries quote them (valu_busy, lanes active, HBM bytes) only when the stamp matches the
library it loaded.  Separate passes (never combined with a trace domain); the program after `--` is the CLI binary itself.

    python3 tools/pmc_configs.py <out dir>          (on the GPU box; writes pmc_k9_config4.json, pmc_k2_config3.json)
"""
import collections, csv, glob, hashlib, json, os, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "stdpar-nbody_amd", "bin", "nbody_hip_d3")
PASSES = ["SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE",
          "SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH",
          "FETCH_SIZE", "WRITE_SIZE"]
CONFIGS = {
    "k9_config4": {"kernel": "bvh_force_sweep_isa", "files": ("bvh.hip", "common.hpp"),
                   "args": ["-n", "1000000", "-s", "2", "--algorithm", "bvh", "--workload", "galaxy", "--precision", "double", "--csv-detailed"],
                   "what": "bvh 3D double n = 10^6 galaxy theta = 0.5: one traversal of the initial state"},
    "k2_config3": {"kernel": "all_pairs_collapsed_", "files": ("all_pairs.hip", "common.hpp"),
                   "args": ["-n", "262144", "-s", "2", "--algorithm", "all-pairs-collapsed", "--workload", "uniform", "--precision", "float", "--csv-detailed"],
                   "what": "all-pairs-collapsed 3D float n = 262144 uniform: one force pass"},
}


def sha(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "stdpar-nbody_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def run(extra, args, out):
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3"] + extra + ["--output-format", "csv", "-d", out, "--", BIN] + args, cwd="/tmp", env=env,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)


def derive(rec):
    """What the counters say by themselves.  rocprofv3's VALUBusy formula prices every VALU instruction at 4 cycles (SQ_ACTIVE_INST_VALU is
    in quad-cycles): the issue cost of an f64 instruction; f32 instructions issue in ~2.6, so for an f32 kernel the ratio exceeds 1, is
    no fraction and is published under its own name only."""
    cycles = rec["GRBM_GUI_ACTIVE"] / 8.0
    ratio = rec["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * 1024)
    lanes = rec["SQ_THREAD_CYCLES_VALU"] / (rec["SQ_ACTIVE_INST_VALU"] * 64.0)
    out = {"valu_active_quad_cycles_x4_per_simd_cycle": ratio, "valu_lanes_active_frac": lanes,
           "valu_insts_per_simd_cycle": rec["SQ_INSTS_VALU"] / (cycles * 1024) if "SQ_INSTS_VALU" in rec else None,
           "hbm_bytes": (2.0 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024.0,
           "note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_ACTIVE_INST_VALU counts quad-cycles over 1024 SIMDs; "
                   "FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md)"}
    if ratio <= 1.0:
        out["valu_busy_frac"], out["valu_issue_frac"] = ratio, ratio * lanes
    else:
        out["valu_busy_frac"] = out["valu_issue_frac"] = None
        out["valu_busy_note"] = "the x4 ratio exceeds 1: an f32 kernel (2.6 cycles per instruction, not 4); not a fraction"
    return out


def main():
    dst = sys.argv[1]
    os.makedirs(dst, exist_ok=True)
    for name, cfg in CONFIGS.items():
        rec = {"config": cfg["what"], "command": "nbody_hip_d3 " + " ".join(cfg["args"]), "source_sha": sha(cfg["files"]),
               "source_files": list(cfg["files"]), "passes": PASSES}
        for i, p in enumerate(PASSES):
            d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
            run(["--pmc"] + p.split(), cfg["args"], d)
            f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
            per = collections.defaultdict(dict)
            for r in csv.DictReader(open(f)):
                if cfg["kernel"] in r["Kernel_Name"]:
                    per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
                    rec["kernel"] = r["Kernel_Name"].split("(")[0].replace("void ", "")
            first = per[min(per)]          # the first dispatch: the initial state, the same in every pass
            rec.update(first)
            shutil.rmtree(d, ignore_errors=True)
        d = tempfile.mkdtemp(prefix="trace_", dir="/tmp")
        run(["--kernel-trace", "--stats"], cfg["args"], d)
        for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))[0])):
            if cfg["kernel"] in r["Name"]:
                rec["duration_ns"] = float(r["AverageNs"])
                rec["duration_calls"] = int(r["Calls"])
        shutil.copy(glob.glob(os.path.join(d, "*", "*kernel_stats.csv"))[0], os.path.join(dst, name + "_kernel_stats.csv"))
        shutil.rmtree(d, ignore_errors=True)
        rec["derived"] = derive(rec)
        json.dump(rec, open(os.path.join(dst, "pmc_%s.json" % name), "w"), indent=1)
        print(name, json.dumps(rec["derived"]))


if __name__ == "__main__":
    main()
