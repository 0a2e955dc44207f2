#!/bin/bash
# Prints the host/GPU identification lines a benchmark log starts with, in the shape the reference's ci/benchmark and
# ci/benchmark_detailed emit them (ci/benchmark:44-50: the two-line `name, driver_version` CSV of the GPU query, lscpu's
# "Model name" and "Core(s) per socket" lines, "hostname:<name>") so that tools/scrape_bench_log.py — like the reference's
# ci/data.py — can attach them to every result row.  Sourced by tools/benchmark.sh and tools/benchmark_detailed.sh.
GPU=$(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 "Marketing Name:.*\(MI\|Instinct\)" | sed 's/.*Marketing Name: *//' | tr -d ',' || true)
ARCH=$(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 -o "gfx[0-9a-z]*" || true)
DRV=$(cat /sys/module/amdgpu/version 2>/dev/null || /opt/rocm/bin/rocm-smi --showdriverversion 2>/dev/null | grep -m1 -o "[0-9][0-9.]*$" || echo unknown)
echo "name, driver_version"
echo "${GPU:-AMD Instinct} (${ARCH:-unknown}), ${DRV:-unknown}"
lscpu | grep "Model name"
lscpu | grep "Core(s) per socket"
echo "hostname:$(hostname)"
