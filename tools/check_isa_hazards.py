#!/usr/bin/env python3
"""Static check of the shipped gfx950 code object for the data hazards that need software wait states.

hipcc pads its own instructions, but it treats an `asm` statement as one opaque instruction: nothing inside the string is
padded, and nothing is padded between the compiler's last instruction and the first one of the string.  Round 3 met exactly
that: a `v_mov_b32` scheduled directly in front of an asm `v_readfirstlane_b32` of its result made the lane read return the
register's previous content (csrc/to_sgpr.hpp) — wrong forces with no fault and no message, and only in the instantiations
whose register allocation happened to put the two instructions side by side.  The library carries ~600 lines of hand-written
ISA (K1's scalar stream, K9's step program, the octree visit rounds), so the rules are checked on the disassembly of what
ships, along the control-flow graph of every kernel, compiler-scheduled code included (which doubles as a check of the rules:
hipcc's own padding must pass them).

Rules (gfx940-class parts; wait states = instructions issued in between, `s_nop N` counting N + 1):
  A  VALU writes a VGPR            -> v_readlane / v_readfirstlane / v_permlane reads it              1
  B  VALU writes a VGPR            -> a DPP instruction reads it                                     2
  C  transcendental writes a VGPR  -> a non-transcendental VALU reads it                             1
  D  VALU writes an SGPR / VCC     -> a VALU reads it (operand, select mask, carry in)               2
  E  VALU writes an SGPR / VCC     -> v_readlane / v_writelane uses it as the lane select            4
  F  VALU writes VCC               -> v_div_fmas                                                      4
  G  VALU writes an SGPR           -> a vector memory instruction reads it (base, descriptor, offset) 5
  H  VALU writes EXEC (v_cmpx)     -> v_readlane / v_readfirstlane / v_writelane                     4
  I  VALU writes EXEC (v_cmpx)     -> a DPP instruction                                              5

    python tools/check_isa_hazards.py [path/to/libnbody_hip.so]
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_smem_pipeline import disassemble, functions  # noqa: E402

TRANS = re.compile(r"^v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f16|f32|f64|legacy_f32)")
VMEM = ("global_", "buffer_", "flat_", "scratch_", "tbuffer_")
TWO_DESTS = re.compile(r"^v_(div_scale_|mad_u64_u32|mad_i64_i32|add_co_|sub_co_|subrev_co_|addc_co_|subb_co_|subbrev_co_)")
READS_DEST = re.compile(r"^v_(fmac|mac|fmaak|writelane|dot\d*c|pk_fmac)_")
MAX_STATES = 5


def regs_of(tok):
    """registers named by one operand, as a set of (file, index)"""
    tok = tok.strip()
    out = set()
    body = re.sub(r"^[-|]+|[|]+$", "", tok)          # -v1, |v1|, -|v1|
    body = re.sub(r"^(neg|abs|sext)\((.*)\)$", r"\2", body)
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", body)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"^([vsa])(\d+)$", body)
    if m:
        return {(m.group(1), int(m.group(2)))}
    if body in ("vcc", "vcc_lo", "vcc_hi"):
        return {("vcc", 0)}
    if body in ("exec", "exec_lo", "exec_hi"):
        return {("exec", 0)}
    return out


class Ins:
    def __init__(self, addr, text, target):
        self.addr, self.text, self.target = addr, text, target
        parts = text.split(None, 1)
        self.op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        # operands are comma separated; modifiers (row_shr:1, op_sel:[0,1], offset:16, sc1 ...) follow the last operand after a space
        ops, depth, cur = [], 0, ""
        for ch in rest:
            if ch == "[":
                depth += 1
            elif ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                ops.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            ops.append(cur)
        ops = [o.strip() for o in ops]
        self.mods = ""
        if ops:
            last = ops[-1].split(None, 1)
            if len(last) > 1:
                ops[-1], self.mods = last[0], last[1]
        self.ops = ops
        op = self.op
        self.valu = op.startswith("v_") and not op.startswith("v_nop")
        self.trans = bool(TRANS.match(op))
        self.vmem = op.startswith(VMEM)
        self.lane_read = op.startswith(("v_readlane_b32", "v_readfirstlane_b32", "v_permlane"))
        self.lane_sel = op.startswith(("v_readlane_b32", "v_writelane_b32"))
        self.lane_any = self.lane_read or op.startswith("v_writelane_b32")
        self.dpp = "_dpp" in op or bool(re.search(r"\b(quad_perm|row_shl|row_shr|row_ror|row_mirror|row_half_mirror|row_bcast|row_newbcast|wave_shl|wave_shr|wave_rol|wave_ror|row_share|row_xmask)", self.mods))
        self.div_fmas = op.startswith("v_div_fmas")
        ndst = 0
        if self.valu:
            ndst = 2 if TWO_DESTS.match(op) else 1
            if op.startswith("v_swap_b32"):
                ndst = 2
        self.writes, self.reads = set(), set()
        for k, o in enumerate(ops):
            r = regs_of(o)
            if self.valu and k < ndst:
                self.writes |= r
                if READS_DEST.match(op) or op.startswith("v_swap_b32"):
                    self.reads |= r
            else:
                self.reads |= r
        if self.valu and op.startswith("v_cmpx"):
            self.writes.add(("exec", 0))
        if self.div_fmas:
            self.reads.add(("vcc", 0))
        self.lane_select = set()
        if self.lane_sel and len(ops) >= 3:
            self.lane_select = regs_of(ops[2])
        if op.startswith("s_nop"):
            self.states = int(ops[0], 0) + 1 if ops else 1
        else:
            self.states = 1

    def vgprs_written(self):
        return {r for r in self.writes if r[0] in ("v", "a")}

    def sgprs_written(self):
        return {r for r in self.writes if r[0] in ("s", "vcc")}


def violations_from(code, index, i):
    """hazards whose producer is instruction i: [(rule, consumer index, wait states seen, needed)]"""
    p = code[i]
    if not p.valu:
        return []
    wv, ws, wexec = p.vgprs_written(), p.sgprs_written(), ("exec", 0) in p.writes
    if not (wv or ws or wexec):
        return []
    found, seen = [], {}
    stack = []

    def push(j, states):
        if j is None or j >= len(code):
            return
        if seen.get(j, 99) <= states:
            return
        seen[j] = states
        stack.append((j, states))

    def succ(k, states):
        ins = code[k]
        if not ins.op.startswith(("s_endpgm", "s_branch", "s_setpc", "s_swappc")) and k + 1 < len(code):
            push(k + 1, states)
        if ins.target is not None and ins.target in index:
            push(index[ins.target], states)

    succ(i, 0)
    while stack:
        j, states = stack.pop()
        c = code[j]

        def need(rule, n, hit):
            if hit and states < n:
                found.append((rule, j, states, n))

        if wv:
            rv = c.reads & wv
            need("A", 1, c.lane_read and bool(rv))
            need("B", 2, c.dpp and bool(rv))
            need("C", 1, p.trans and c.valu and not c.trans and bool(rv))
        if ws:
            rs = c.reads & ws
            need("D", 2, c.valu and bool(rs) and not (c.lane_select & ws))
            need("E", 4, bool(c.lane_select & ws))
            need("F", 4, c.div_fmas and ("vcc", 0) in ws)
            need("G", 5, c.vmem and bool({r for r in rs if r[0] == "s"}))
        if wexec:
            need("H", 4, c.lane_any)
            need("I", 5, c.dpp)
        states += c.states
        if states < MAX_STATES:
            succ(j, states)
    return found


def check(lib_path):
    """(instructions examined, lane reads seen, list of violations)"""
    total, lanes, problems = 0, 0, []
    for name, raw in functions(disassemble(lib_path)).items():
        code = [Ins(a, t, tg) for a, t, tg in raw]
        index = {ins.addr: k for k, ins in enumerate(code)}
        total += len(code)
        lanes += sum(1 for c in code if c.lane_read)
        for i in range(len(code)):
            for rule, j, states, n in violations_from(code, index, i):
                problems.append(f"{name[:60]} rule {rule}: `{code[i].text}` @{code[i].addr:x} -> `{code[j].text}` @{code[j].addr:x}: "
                                f"{states} wait state(s), {n} needed")
    return total, lanes, problems


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "stdpar-nbody_amd", "libnbody_hip.so")
    total, lanes, problems = check(lib)
    print(f"{total} instructions, {lanes} lane reads, {len(problems)} hazard(s) without their wait states")
    for p in problems[:50]:
        print("  " + p)
    sys.exit(1 if problems else 0)
