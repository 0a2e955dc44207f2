#!/bin/bash
# Prints the host/GPU identification lines a benchmark log starts with, in the shape the reference's ci/benchmark and
# ci/benchmark_detailed emit them (ci/benchmark:44-50: the two-line `name, driver_version` CSV of the GPU query, lscpu's
# "Model name" and "Core(s) per socket" lines, "hostname:<name>") so that tools/scrape_bench_log.py — like the reference's
# ci/data.py — can attach them to every result row.  Sourced by tools/benchmark.sh and tools/benchmark_detailed.sh.
# rocminfo needs /dev/kfd privileges the GPU box's user may lack: ask the HIP runtime through PyTorch (device name, HIP version)
ID=$(python3 -c "import torch; p = torch.cuda.get_device_properties(0); print(p.name.replace(chr(44), chr(32)) + chr(32) + chr(40) + p.gcnArchName.split(chr(58))[0] + chr(41) + chr(44) + chr(32) + str(torch.version.hip))" 2>/dev/null || echo "AMD Instinct (gfx950), unknown")
echo "name, driver_version"
echo "$ID"
lscpu | grep "Model name"
lscpu | grep "Core(s) per socket"
echo "hostname:$(hostname)"
