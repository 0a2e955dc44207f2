#!/usr/bin/env python3
"""Benchmark log -> one CSV table, with the column layout of the reference's scraper (ci/data.py:25-60):

    gpu,driver,cpu,#cores,seq,compiler,hostname,<the CLI's own CSV header>

The log is what tools/benchmark.sh / tools/benchmark_detailed.sh write (and what the reference's ci/benchmark writes): a
GPU query block (`name, driver_version` then `<gpu>, <driver>`), lscpu's `Model name:` and `Core(s) per socket:` lines,
`hostname:<h>`, then per run a `compiler:<c>` line followed by the binary's CSV header and result row.  Lines a shell
trace adds (`+ cmd`) are skipped; a `sequential` line marks the rows after it as sequential runs.  Logs that mix runs
with different headers (all-pairs vs trees under --csv-detailed) get the widest header; short rows are padded.

    python tools/scrape_bench_log.py bench.log > bench.csv
"""
import sys

ALGORITHMS = ("octree", "all-pairs", "all-pairs-collapsed", "bvh")


def scrape(lines):
    ident = {"gpu": None, "driver": None, "cpu": None, "cores": None, "compiler": None, "hostname": None}
    sequential, expect_gpu, header, rows = False, False, None, []
    for raw in lines:
        line = raw.rstrip("\n")
        if line.startswith("+") or not line.strip():
            continue
        if expect_gpu:                                   # the line after the `name, driver_version` header
            parts = [p.strip() for p in line.split(", ")]
            ident["gpu"], ident["driver"] = parts[0], (parts[1] if len(parts) > 1 else None)
            expect_gpu = False
        elif line.startswith("name"):
            expect_gpu = True
        elif line.startswith("Model name:"):
            ident["cpu"] = line.split("Model name:", 1)[1].strip()
        elif line.startswith("Core(s) per socket:"):
            ident["cores"] = line.split("Core(s) per socket:", 1)[1].strip()
        elif line.startswith("sequential"):
            sequential = True
        elif line.startswith("compiler"):
            ident["compiler"] = line.split(":", 1)[1].strip()
        elif line.startswith(("hostname", "node")):
            ident["hostname"] = line.split(":", 1)[1].strip()
        elif line.startswith("algorithm"):
            cols = line.strip().split(",")
            if header is None or len(cols) > len(header):
                header = cols
        elif line.split(",")[0] in ALGORITHMS:
            rows.append([ident["gpu"], ident["driver"], ident["cpu"], ident["cores"], sequential, ident["compiler"],
                         ident["hostname"]] + line.strip().split(","))
    return header or [], rows


def to_csv(header, rows):
    out = [",".join(["gpu", "driver", "cpu", "#cores", "seq", "compiler", "hostname"] + header)]
    width = 7 + len(header)
    for r in rows:
        out.append(",".join(str(v) for v in (r + [""] * (width - len(r)))))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    if len(sys.argv) != 2:
        sys.exit(__doc__)
    with open(sys.argv[1]) as f:
        sys.stdout.write(to_csv(*scrape(f.readlines())))
