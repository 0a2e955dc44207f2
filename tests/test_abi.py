"""CPU: the C-ABI library loads (no GPU needed to dlopen it) and exports exactly what include/nbody_hip.h declares."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "nbody_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nbody_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_list_agree(nb):
    assert _header_symbols() == sorted(nb.ABI_SYMBOLS)


def test_header_is_plain_c_and_cxx():
    """The boundary is a C ABI: the header must compile as C99 and as C++ with nothing but the standard headers."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "nbody_hip.h")
    for cmd in (["gcc", "-fsyntax-only", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", hdr],
                ["g++", "-fsyntax-only", "-x", "c++", "-std=c++17", "-Wall", "-Wextra", "-Werror", hdr]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_library_exports_every_declared_symbol(nb):
    if not os.path.exists(nb.LIB_PATH):
        nb.build()
    L = ctypes.CDLL(nb.LIB_PATH)
    for sym in _header_symbols():
        assert hasattr(L, sym), f"libnbody_hip.so does not export {sym}"


def test_code_object_is_gfx950_only(nb):
    """Write for MI355X only: the embedded code objects target gfx950 and no other GPU arch."""
    raw = open(nb.LIB_PATH, "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", raw))
    assert archs == {b"gfx950"}, archs


def test_argument_errors_do_not_need_a_gpu(nb):
    """Bad arguments are rejected before any HIP call, with a message (no exceptions cross the C boundary)."""
    L = nb.lib()
    st = nb.nbody_state()
    st.dtype, st.dim = 7, 3
    assert L.nbody_all_pairs_force(ctypes.byref(st), None) == 1
    assert b"dtype" in L.nbody_last_error()
    st.dtype, st.dim = nb.F64, 4
    assert L.nbody_accelerate_step(ctypes.byref(st), None) == 1
    assert b"dim" in L.nbody_last_error()
    assert L.nbody_all_pairs_configure(3, 0) == 1
    assert L.nbody_all_pairs_force(None, None) == 1


def test_no_cpu_fallback_in_product(nb):
    """The product must not reference the oracle, and must fail loudly when the extension is missing."""
    pkg = os.path.join(ROOT, "stdpar-nbody_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in src and "import oracle" not in src and "from oracle" not in src, f
    saved = nb.LIB_PATH
    try:
        nb.LIB_PATH = saved + ".missing"
        import importlib
        nb._lib = None
        with pytest.raises(nb.NbodyError):
            nb.lib()
    finally:
        nb.LIB_PATH = saved
        nb._lib = None


def test_bench_launcher_needs_the_gpus_it_is_asked_for():
    """`python bench.py --gpus 2` as typed: with fewer HIP devices than ranks the parent says so and exits non-zero
    before starting anything (here: a box with no GPU at all)."""
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "visible" in r.stderr and not r.stdout.strip()


def test_shard_range_matches_python(nb):
    for n in (1, 7, 301, 1 << 20, 1000003):
        for w in (1, 2, 3, 8):
            for r in range(w):
                assert nb.shard_range(n, r, w) == nb.parallel.shard_range(n, r, w)


def test_smem_pipeline_registers_untouched_in_flight(nb):
    """K1's scalar-stream loop requests the next 16-SGPR batch with inline-asm s_load_dwordx16 one compute phase before it
    waits for it (csrc/common.hpp sload16/swait).  Nothing but the register allocator's cooperation keeps other
    instructions off that range while it is in flight, so the shipped code object is disassembled and checked along
    its control-flow graph (tools/check_smem_pipeline.py)."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    spec = importlib.util.spec_from_file_location("check_smem_pipeline", os.path.join(ROOT, "tools", "check_smem_pipeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    loads, problems = mod.check(nb.LIB_PATH)
    assert loads >= 64, f"only {loads} s_load_dwordx16 found: K1's scalar-stream kernels are missing from the disassembly"
    assert not problems, "\n".join(problems[:10])


def test_k9_step_program_never_leaves_a_record_request_in_flight(nb):
    """K9's sweep requests the record behind a skip before it knows whether the walk goes on (round 4), so a request can be in
    flight when the walk ends and when the jump path asks for another record into the same block; one `s_waitcnt lgkmcnt(0)` in
    the program text closes each.  Without them a record lands in registers the compiler has handed to something else — the
    pointers of the kernel's final stores (the abort of round 4).  The checker's two rules (no instruction touches an
    in-flight range; no request of a hand-written block is in flight at s_endpgm) must hold for all eight instantiations, in double
    (s_load_dwordx16) and in float (s_load_dwordx8) — and the checker must be ABLE to see the defect: every wait that guards a record
    block is removed from the parsed program in turn, and each removal must be reported."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    spec = importlib.util.spec_from_file_location("check_smem_pipeline", os.path.join(ROOT, "tools", "check_smem_pipeline.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kernels, tried, missed = mod.self_test(nb.LIB_PATH)
    assert kernels == 8, f"{kernels} instantiations of bvh_force_sweep_isa_kernel found, expected 8 (2 precisions x 2 dimensions x counters)"
    assert tried >= 4 * kernels, f"only {tried} guarding waits found in {kernels} kernels"
    assert missed == 0, f"{missed} of {tried} removed waits went unreported"


def test_k1_handoff_loads_follow_the_poll_and_bypass_the_l2(nb):
    """K1's running total changes hands between blocks through relaxed agent-scope atomics ordered by `s_waitcnt 0` — outside the HIP
    memory model's letter, so the property is checked where it lives, in the built code object (tools/check_k1_handoff.py): every
    load and store of the total carries sc1; every load of it sits behind the poll loop (the source pins that with a wavefront-scope
    acquire fence) and nothing branches from the adding code back into the polls; the turn's compare-and-swap follows an
    `s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)` with no memory instruction in between; the collecting form loads the sums behind its
    ticket.  The checker must be ABLE to see each defect: an sc1 dropped from any access, a load of the total copied above the
    polls, the wait before the hand-over removed — every mutation of every instantiation must be reported."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    spec = importlib.util.spec_from_file_location("check_k1_handoff", os.path.join(ROOT, "tools", "check_k1_handoff.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    facts, problems = mod.check(nb.LIB_PATH)
    assert len(facts) == 32, f"{len(facts)} instantiations of all_pairs_force_sgpr_kernel found, expected 32"
    assert not problems, "\n".join(problems[:10])
    assert all(p == 2 and loads >= 2 and stores == 2 * loads and swaps == 1 for p, loads, stores, swaps in facts.values()), facts
    tried, missed = mod.self_test(nb.LIB_PATH)
    assert tried >= 10 * len(facts) and missed == 0, f"{missed} of {tried} mutations went unreported"


def test_no_unpadded_isa_hazards_in_the_code_object(nb):
    """hipcc pads the data hazards of its own instructions, not those inside or at the edge of an asm block (round 3: an asm
    v_readfirstlane_b32 directly behind the compiler's v_mov of its source read the register's previous content in three
    octree instantiations — csrc/to_sgpr.hpp).  tools/check_isa_hazards.py walks every kernel of the shipped code object along
    its control flow and applies the gfx940-class wait-state rules; built without the pads it flags exactly the three
    instantiations that were wrong on the hardware, and nothing else."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump not available")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    spec = importlib.util.spec_from_file_location("check_isa_hazards", os.path.join(ROOT, "tools", "check_isa_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    total, lanes, problems = mod.check(nb.LIB_PATH)
    assert total > 50000 and lanes > 100, (total, lanes)   # the disassembly really is the library's
    assert not problems, "\n".join(problems[:10])
    # the rules themselves: a producer directly in front of its consumer is flagged, the padded pair is not
    def run(lines):
        code = [mod.Ins(4 * k, t, None) for k, t in enumerate(lines)]
        index = {c.addr: k for k, c in enumerate(code)}
        return [r for i in range(len(code)) for r, *_ in mod.violations_from(code, index, i)]
    assert run(["v_mov_b32_e32 v9, s22", "v_readfirstlane_b32 s36, v9"]) == ["A"]
    assert run(["v_mov_b32_e32 v9, s22", "s_nop 0", "v_readfirstlane_b32 s36, v9"]) == []
    assert run(["v_cmp_gt_f32_e32 vcc, s0, v12", "s_nop 0", "v_cndmask_b32_e32 v12, v12, v13, vcc"]) == ["D"]
    assert run(["v_cmp_gt_f32_e32 vcc, s0, v12", "s_nop 1", "v_cndmask_b32_e32 v12, v12, v13, vcc"]) == []
    assert run(["v_rsq_f64_e32 v[2:3], v[4:5]", "v_mul_f64 v[6:7], v[2:3], v[2:3]"]) == ["C"]
    assert run(["v_readlane_b32 s5, v1, s2", "s_nop 0", "v_cmp_gt_u32_e32 vcc, s5, v3"]) == ["D"]
    assert run(["v_cmp_eq_u32_e64 s[4:5], v1, v2", "v_nop", "v_nop", "global_load_dword v3, v4, s[4:5]"]) == ["G"]
    assert run(["v_cmpx_eq_u32_e64 s[4:5], v1, v2", "v_nop", "v_readfirstlane_b32 s7, v9"]) == ["H"]
    assert run(["v_div_scale_f32 v15, vcc, s22, v12, s22", "v_nop", "v_nop", "v_div_fmas_f32 v13, v13, v14, v16"]) == ["F"]


def test_shipped_binaries_read_no_environment(nb):
    """the tuning switches (NBODY_K1_CHUNKS, NBODY_K2_CFG, NBODY_K9_*, NBODY_OT_FORM, NBODY_CLI_FORCE_COMM) exist
    only in the -DNBODY_EXPERIMENTS build.  The shipped library and CLI do not import getenv and do not carry the names."""
    targets = [nb.LIB_PATH] + [os.path.join(ROOT, "stdpar-nbody_amd", "bin", f"nbody_hip_d{d}") for d in (2, 3)]
    for path in targets:
        assert os.path.exists(path), path
        und = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True).stdout
        assert "getenv" not in und, f"{path} imports getenv"
        raw = open(path, "rb").read()
        for name in (b"NBODY_K1_CHUNKS", b"NBODY_K2_CFG", b"NBODY_K9_", b"NBODY_OT_FORM", b"NBODY_CLI_FORCE_COMM"):
            assert name not in raw, (path, name)


def test_hand_built_state_with_garbage_tuning_is_refused(nb):
    """nbody_state.tuning must be 0 or an NBODY_TUNING(...) value; a state that was not zero-initialised is refused
    before any launch (no GPU needed: the check precedes every HIP call)."""
    L = nb.lib()
    st = nb.nbody_state()
    st.dtype, st.dim, st.sz, st.count = nb.F64, 3, 16, 16
    st.m = st.x = st.v = st.a = st.ao = 0x1000  # never dereferenced: argument checks come first
    for bad in (0x1, 0x108 | (3 << 6), 0x10000, 0x100 | 3, 0xdeadbeef):
        st.tuning = bad
        assert L.nbody_all_pairs_force(ctypes.byref(st), None) == 1, hex(bad)
        assert L.nbody_accelerate_step(ctypes.byref(st), None) == 1, hex(bad)
    buf = ctypes.create_string_buffer(256)
    st.tuning = nb.tuning(4, 1, 1)
    assert L.nbody_all_pairs_describe(ctypes.byref(st), buf, ctypes.c_size_t(256)) == 0, L.nbody_last_error()


def test_k9_sweep_fits_eight_waves_per_simd(nb):
    """A wave that names an SGPR above s73 (sgpr_count > 80 with VCC and the reserved pairs) leaves room for seven waves per SIMD,
    not eight (measured: tools/microbench/cu_map.hip, s72 / s73 / s74 = 8 / 8 / 7 waves in profiles/r06/cu_map_microbench.txt; rounds
    3-4 ran K9 at six and seven while believing seven and eight).  The bound below (78: nothing above s71) is two registers inside.  The
    sweep's record blocks sit at s[40:71] for that reason; this holds the built kernels to the budget (metadata of the code object,
    tools/kernel_resources.py), so that a change which nudges the compiler's own scalars upward cannot quietly cost the eighth wave."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf not available")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ks = [k for k in mod.kernels(nb.LIB_PATH) if "bvh_force_sweep_isa_kernel" in k["symbol"]]
    assert len(ks) == 8, len(ks)
    for k in ks:
        assert int(k["sgpr_count"]) <= 78 and int(k["vgpr_count"]) + int(k.get("agpr_count", 0)) <= 64, (k["symbol"], k["sgpr_count"], k["vgpr_count"])
