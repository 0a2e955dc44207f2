"""CPU: the reference-side binding of INTEGRATION.md meets the reference.

The stub a maintainer of UoB-HPC/stdpar-nbody would add (examples/hip_backend.h, quoted verbatim in INTEGRATION.md) is compiled
against the reference's own headers where they lie — System<T,N> and its state_t view (/root/reference/src/system.h:13-50), the
call sites it replaces (src/all_pairs.h:17, src/bvh.h:382-393, src/octree.h:321-326) and run_simulation / sim_func_t
(src/main.cpp:16-40) — with every template instantiated for float and double, and LINKED against libnbody_hip.so.  Any drift of a
signature in include/nbody_hip.h, of the stub or of the doc breaks this test.  No GPU call is made: the build container suffices.
Skipped where /root/reference is absent (the GPU box); there tests/test_gpu_cli.py RUNS the binary this test builds.
"""
import os
import re
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference/src"
needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "system.h")), reason="the reference is not on this machine")


def _cpp_blocks(md):
    return re.findall(r"```cpp\n(.*?)```", md, flags=re.S)


def test_the_doc_quotes_the_stub_verbatim():
    blocks = _cpp_blocks(open(os.path.join(ROOT, "INTEGRATION.md")).read())
    stub = open(os.path.join(ROOT, "examples", "hip_backend.h")).read()
    assert len(blocks) == 1, "INTEGRATION.md holds one cpp block: the stub (everything it shows is compiled)"
    assert blocks[0] == stub, "INTEGRATION.md and examples/hip_backend.h differ"
    for entry in ("hip_mirror", "all_pairs_force", "all_pairs_collapsed_force", "accelerate_step", "calc_energies", "bvh_force",
                  "octree_force", "hip_multi", "hip_all_pairs_callable"):
        assert re.search(r"\b%s\b" % entry, stub), entry


@needs_reference
def test_stub_from_the_doc_compiles_against_the_reference_and_links(tmp_path):
    """The cpp block of INTEGRATION.md — extracted from the doc, not read from examples/ — as a translation unit of its own over
    the reference's system.h, every template instantiated, linked against the backend."""
    block = _cpp_blocks(open(os.path.join(ROOT, "INTEGRATION.md")).read())[0]
    (tmp_path / "hip_backend.h").write_text(block)
    (tmp_path / "tu.cpp").write_text("""
#include <algorithm>
#include <chrono>
#include "all_pairs.h"   // as src/main.cpp:1-9 begins: the reference's headers lean on it for <ranges>, format.h, timer.h
#include "hip_backend.h"
template struct hip_mirror<double, 3>;
template struct hip_mirror<float, 3>;
template struct hip_mirror<double, 2>;
template struct hip_mirror<float, 2>;
template struct hip_multi<double, 3>;
template struct hip_multi<float, 2>;
#define INSTANTIATE(T, N)                                                          \\
  template void all_pairs_force<T, N>(hip_mirror<T, N>&);                         \\
  template void all_pairs_collapsed_force<T, N>(hip_mirror<T, N>&);               \\
  template void accelerate_step<T, N>(hip_mirror<T, N>&);                         \\
  template auto calc_energies<T, N>(hip_mirror<T, N>&) -> std::tuple<T, T>;      \\
  template void bvh_force<T, N>(hip_mirror<T, N>&, T);                            \\
  template void octree_force<T, N>(hip_mirror<T, N>&, T);                         \\
  template auto hip_all_pairs_callable<T, N>(hip_mirror<T, N>&);
INSTANTIATE(double, 3) INSTANTIATE(float, 3) INSTANTIATE(double, 2) INSTANTIATE(float, 2)
// the view the stub hands over is the reference's own, field for field (src/system.h:41-50)
static_assert(sizeof(vec<double, 3>) == 24 && sizeof(vec<float, 2>) == 8);
static_assert(std::is_same_v<System<double, 3>::index_t, uint32_t>);
static_assert(std::is_same_v<decltype(System<float, 3>::state_t::x), vec<float, 3>*>);
int main(int argc, char**) {
  if (argc > 1000) {   // never true: the calls below are linked, not run (no GPU here)
    System<double, 3> s(4, 0.1, 1.0);
    hip_mirror<double, 3> d(s);
    all_pairs_force(d); accelerate_step(d); bvh_force(d, 0.5); octree_force(d, 0.5);
    auto [ke, pe] = calc_energies(d); (void)ke; (void)pe;
    hip_all_pairs_callable(d)(s);
    d.sync(); d.download(s);
    hip_multi<double, 3> m(s, 2); m.step(); m.download(s);
  }
  return nbody_abi_version() == NBODY_HIP_ABI_VERSION ? 0 : 1;
}
""")
    exe = tmp_path / "tu"
    pkg = os.path.join(ROOT, "stdpar-nbody_amd")
    import torch   # the header-only {fmt} inside the PyTorch wheel: the reference's own FMT_FORMAT_WORKAROUND branch (src/format.h:3-8)
    fmt_inc = os.path.join(os.path.dirname(torch.__file__), "include")
    cmd = ["g++", "-std=c++20", "-Wall", "-Werror=return-type", "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path),
           "-I", REF, "-I", fmt_inc, "-include", "span", "-include", "numbers", str(tmp_path / "tu.cpp"), "-o", str(exe), "-L", pkg, "-lnbody_hip",
           "-Wl,-rpath," + pkg]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    # linked for real: the ABI entries are undefined symbols of the binary, resolved by the backend library
    nm = subprocess.run(["nm", "-D", "--undefined-only", str(exe)], capture_output=True, text=True, check=True).stdout
    for sym in ("nbody_create", "nbody_upload", "nbody_download", "nbody_ctx_state", "nbody_ctx_stream", "nbody_ctx_set_shard",
                "nbody_all_pairs_force", "nbody_all_pairs_collapsed_force", "nbody_accelerate_step", "nbody_calc_energies",
                "nbody_bvh_create_on", "nbody_bvh_bounding_box", "nbody_bvh_hilbert_sort", "nbody_bvh_build_tree",
                "nbody_bvh_compute_force", "nbody_octree_create_on", "nbody_octree_clear", "nbody_octree_compute_bounds",
                "nbody_octree_insert", "nbody_octree_compute_tree", "nbody_octree_compute_force", "nbody_comm_create_all",
                "nbody_comm_group_begin", "nbody_comm_group_end", "nbody_allgather_positions", "nbody_shard_range", "nbody_stream_sync"):
        assert re.search(r"\bU %s\b" % sym, nm), sym
    # and it loads and runs up to the first thing that needs no device: the ABI version of the library it was linked with
    assert subprocess.run([str(exe)], timeout=60).returncode == 0


@needs_reference
def test_reference_program_with_the_backend_dropped_in_builds():
    """examples/reference_hip_main.cpp: the reference's UNMODIFIED main.cpp (CLI, generators, run_simulation, run_all_pairs, Saver,
    print) with the stub's sim_func_t registered — `make -C oracle ref_hip` compiles and links it for D = 2 and 3.  Without a GPU
    it must fail loudly at the first backend call, after the reference's own banner."""
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_hip"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    for d in (2, 3):
        exe = os.path.join(ROOT, "oracle", "_ref", f"nbody_ref_hip_d{d}")
        assert os.path.exists(exe)
        libs = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
        assert "libnbody_hip.so" in libs and "not found" not in libs
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "-n", "10", "-s", "2", "--precision", "double", "--algorithm", "all-pairs"], capture_output=True,
                           text=True, timeout=60)
        assert r.returncode != 0 and r.stdout.startswith("Starting simulation") and "HIP error" in r.stderr
