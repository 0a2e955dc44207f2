#!/usr/bin/env python3
"""Static check of the hand-written SMEM pipeline in the shipped gfx950 code object.

K1's scalar-stream loop (and the energies' copy of it) requests the next 64-byte batch of source records with an
inline-asm `s_load_dwordx16` one compute phase before it waits for it (csrc/common.hpp: sload16 / swait).  The
compiler does not know that the asm's 16-SGPR result is still in flight, so nothing but the register allocator's
cooperation keeps other instructions off that range until the `s_waitcnt lgkmcnt(0)` that completes it.  This tool
disassembles the code object, builds the control-flow graph of every kernel, propagates the set of in-flight SGPR
ranges along it (union at joins, cleared by a wait on lgkmcnt(0)) and reports any instruction that reads or writes a
range while it is in flight.  tests/test_abi.py runs it on the built library (no GPU needed).

K9's sweep (csrc/bvh.hip, the step program) keeps two record blocks in SGPRs the same way — s_load_dwordx16 in double,
s_load_dwordx8 in float — and since round 4 it requests the record behind a skip BEFORE it knows whether the walk goes on,
so a request can be in flight (a) when the walk ends and (b) when the jump path asks for another record into the same
block.  Both are closed by an `s_waitcnt lgkmcnt(0)` in the program text; a missing one lets a record land in registers
the compiler has given to something else (after the block: the pointers of the final stores).  Rule (b) is the rule
above (the second request writes a range that is in flight).  Rule (a) is checked as: no x8 / x16 request may be in
flight at s_endpgm — some path would have left the hand-written block without waiting.  `self_test` removes each wait
that guards a block from the parsed program in turn and requires a report (tests/test_abi.py).

    python tools/check_smem_pipeline.py [path/to/libnbody_hip.so]
"""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# kernels that hold hand-written SMEM requests (rule (a) applies to these: the compiler's own wide kernel-argument loads may
# legitimately be outstanding on a path that ends the wave early)
HAND_WRITTEN = ("bvh_force_sweep_isa_kernel", "all_pairs_force_sgpr_kernel", "potential_sgpr_kernel", "ot_force_isa_kernel")
ELF_AMDGPU = b"\x7fELF\x02\x01\x01\x40"  # ELFCLASS64, little endian, ELFOSABI_AMDGPU_HSA


def disassemble(lib_path):
    raw = open(lib_path, "rb").read()
    text, pos, k = "", 0, 0
    with tempfile.TemporaryDirectory() as d:
        while True:
            pos = raw.find(ELF_AMDGPU, pos)
            if pos < 0:
                break
            path = os.path.join(d, f"co{k}.elf")
            open(path, "wb").write(raw[pos:])
            text += subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", path], capture_output=True, text=True).stdout
            pos += len(ELF_AMDGPU)
            k += 1
    return text


def functions(text):
    """{name: [(addr, instruction text, branch target addr or None)]}"""
    out, cur, start = {}, None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            start, cur = int(m.group(1), 16), m.group(2)
            out[cur] = []
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<.*?\+0x([0-9a-f]+)>|<[^+>]*>)?\s*$", line)
        if m and cur is not None:
            ins, addr = m.group(1), int(m.group(2), 16)
            target = None
            if ins.startswith(("s_branch", "s_cbranch")):
                target = start + int(m.group(3), 16) if m.group(3) else start
            out[cur].append((addr, ins, target))
    return out


def sgprs_named(ins):
    regs = []
    body = ins.split(None, 1)[1] if " " in ins else ""
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", body):
        regs.append((int(a), int(b)))
    for a in re.findall(r"\bs(\d+)\b", body):
        regs.append((int(a), int(a)))
    return regs


def check_function(name, code):
    index = {addr: i for i, (addr, _, _) in enumerate(code)}
    state = [None] * len(code)  # in-flight ranges on entry to instruction i
    state[0] = frozenset()
    work, problems, loads = [0], [], 0
    while work:
        i = work.pop()
        addr, ins, target = code[i]
        live = state[i]
        m = re.match(r"s_load_dwordx(?:8|16) s\[(\d+):(\d+)\]", ins)
        out = live
        if m:
            out = live | {(int(m.group(1)), int(m.group(2)))}
        elif ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
            out = frozenset()
        succ = []
        if not ins.startswith(("s_endpgm", "s_branch")) and i + 1 < len(code):
            succ.append(i + 1)
        if target is not None and target in index:
            succ.append(index[target])
        for j in succ:
            new = out if state[j] is None else state[j] | out
            if new != state[j]:
                state[j] = new
                work.append(j)
    for i, (addr, ins, _) in enumerate(code):
        if state[i] is None:
            continue
        if ins.startswith(("s_load_dwordx16", "s_load_dwordx8")):
            loads += 1
        if ins.startswith("s_endpgm") and any(h in name for h in HAND_WRITTEN):
            for lo, hi in state[i]:
                problems.append(f"{name} @{addr:x}: s[{lo}:{hi}] still in flight at s_endpgm (a path left the block without waiting)")
        for lo, hi in state[i]:
            for a, b in sgprs_named(ins):
                if not (b < lo or a > hi):
                    problems.append(f"{name} @{addr:x}: in-flight s[{lo}:{hi}] touched by `{ins}`")
    return loads, problems


def check(lib_path):
    """(number of s_load_dwordx16 seen, list of violations)"""
    loads, problems = 0, []
    for name, code in functions(disassemble(lib_path)).items():
        if not code or not any(ins.startswith(("s_load_dwordx16", "s_load_dwordx8")) for _, ins, _ in code):
            continue
        n, p = check_function(name, code)
        loads += n
        problems += p
    return loads, problems


def self_test(lib_path, kernel="bvh_force_sweep_isa_kernel"):
    """For every kernel whose name contains `kernel`: each `s_waitcnt lgkmcnt(0)` is replaced by a no-op in turn; returns
    (kernels, waits tried, waits whose removal was NOT reported).  In the sweep every such wait guards a record block, so the last
    number must be 0: the checker would have caught the program without its end-of-walk wait or without the jump path's."""
    kernels = tried = missed = 0
    for name, code in functions(disassemble(lib_path)).items():
        if kernel not in name or not code:
            continue
        kernels += 1
        assert not check_function(name, code)[1], "the unmodified program must be clean"
        for i, (addr, ins, target) in enumerate(code):
            if ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
                # only the waits of the hand-written block: a record block (x8 / x16) is in flight when they are reached
                mutated = code[:i] + [(addr, "s_nop 0", None)] + code[i + 1:]
                inflight_before = _inflight_at(code, i)
                if not inflight_before:
                    continue
                tried += 1
                if not check_function(name, mutated)[1]:
                    missed += 1
    return kernels, tried, missed


def _inflight_at(code, i):
    """True if some path reaches instruction i with an x8 / x16 request in flight."""
    index = {addr: k for k, (addr, _, _) in enumerate(code)}
    state = [None] * len(code)
    state[0] = False
    work = [0]
    while work:
        k = work.pop()
        addr, ins, target = code[k]
        out = state[k]
        if re.match(r"s_load_dwordx(?:8|16) ", ins):
            out = True
        elif ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
            out = False
        succ = []
        if not ins.startswith(("s_endpgm", "s_branch")) and k + 1 < len(code):
            succ.append(k + 1)
        if target is not None and target in index:
            succ.append(index[target])
        for j in succ:
            new = out if state[j] is None else (state[j] or out)
            if new != state[j]:
                state[j] = new
                work.append(j)
    return bool(state[i])


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "stdpar-nbody_amd", "libnbody_hip.so")
    n, bad = check(lib)
    for b in bad:
        print(b)
    print(f"{n} s_load_dwordx8/x16 instructions checked, {len(bad)} violation(s)")
    k, tried, missed = self_test(lib)
    print(f"self-test: {k} sweep kernels, {tried} guarding waits removed in turn, {missed} removal(s) not reported")
    sys.exit(1 if bad or n == 0 or missed or not tried else 0)
