#!/usr/bin/env python3
"""Static check of K1's chunk hand-off in the built gfx950 code object (csrc/all_pairs.hip, all_pairs_force_sgpr_kernel).

The running total of a target group lives in `a` and changes hands between blocks on other CUs and XCDs.  What makes that
correct is a property of the CODE, not of the HIP memory model's letter (the accesses are relaxed agent-scope atomics):

  (1) every load and store of the total bypasses this XCD's L2 (`sc1`), so there is nothing stale to invalidate;
  (2) every load of the total is ISSUED after the poll that saw the turn — the hardware issues in order and does not
      speculate, so it is enough that the loads sit behind the poll loop in the program and that nothing branches from the
      adding code back into the polls (the source says so with a wavefront-scope acquire fence behind the loop: no
      instruction, but the compiler may not hoist the loads over it);
  (3) the turn is handed on (compare-and-swap on the turn word) only after every store of this wave has been acknowledged:
      `s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)` with no vector memory instruction between it and the swap;
  (4) the collecting form (small launches) loads the chunks' sums only after it has drawn its ticket (a returning atomic
      add) and waited for it, and no load sits between its last store and the poll loop (a load of the total moved up).

This tool disassembles the library, finds the tail of every instantiation (everything behind the last block barrier /
the last batch of the source stream) and checks (1)-(4).  `self_test` breaks each property in turn on the parsed
program — drops an `sc1`, copies a load of the total above the poll loop, removes the wait before the hand-over — and
requires a report.  tests/test_abi.py runs both (no GPU needed).

    python tools/check_k1_handoff.py [path/to/libnbody_hip.so]
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_smem_pipeline import disassemble, functions  # noqa: E402

KERNEL = "all_pairs_force_sgpr_kernel"
VMEM = ("global_", "buffer_", "flat_", "scratch_")


def _is_poll(ins):
    """32-bit sc1 load through an SGPR base: the turn word (wave-uniform address)."""
    return re.match(r"global_load_dword v\d+, v\d+, s\[\d+:\d+\].*\bsc1\b", ins) is not None


def check_function(name, code):
    """-> (facts, problems); facts = (polls, loads of the total, stores of the total, hand-overs) found."""
    problems = []
    marks = [i for i, (_, ins, _) in enumerate(code) if ins.startswith("s_barrier") or ins.startswith("s_load_dwordx16")]
    if not marks:
        return (0, 0, 0, 0), [f"{name}: no source stream found"]
    tail = marks[-1] + 1
    polls = [i for i in range(tail, len(code)) if _is_poll(code[i][1])]
    loads = [i for i in range(tail, len(code)) if code[i][1].startswith("global_load") and i not in polls]
    if not polls:
        return (0, len(loads), 0, 0), [f"{name}: no poll of the turn word found"]
    turn_base = re.search(r"s\[\d+:\d+\]", code[polls[0]][1]).group(0)
    first_poll, last_poll = polls[0], polls[-1]
    for i in polls:   # the poll's value is waited for before anything is decided on it
        nxt = [code[j][1] for j in range(i + 1, min(i + 4, len(code)))]
        if not any(t.startswith("s_waitcnt") and "vmcnt(0)" in t for t in nxt):
            problems.append(f"{name} @{code[i][0]:x}: poll not followed by s_waitcnt vmcnt(0)")
    total_loads = [i for i in loads if i > first_poll]
    collect_loads = [i for i in loads if i < first_poll]
    if not total_loads:
        problems.append(f"{name}: no load of the running total behind the poll loop")
    for i in loads:   # (1)
        if not re.search(r"\bsc1\b", code[i][1]):
            problems.append(f"{name} @{code[i][0]:x}: load in the hand-off tail without sc1: `{code[i][1]}`")
    for i in total_loads:   # (2) program order
        if i < last_poll:
            problems.append(f"{name} @{code[i][0]:x}: load of the total between two polls: `{code[i][1]}`")
    lo, hi = code[first_poll][0], code[last_poll][0]
    first_total = min(total_loads) if total_loads else len(code)
    for i in range(first_total, len(code)):   # (2) nothing re-enters the polls from the adding code
        tgt = code[i][2]
        if tgt is not None and lo <= tgt <= hi:
            problems.append(f"{name} @{code[i][0]:x}: branch from the adding code back into the poll loop")
    stores = [i for i in range(first_total, len(code)) if code[i][1].startswith("global_store")]
    for i in stores:   # (1)
        if not re.search(r"\bsc1\b", code[i][1]):
            problems.append(f"{name} @{code[i][0]:x}: store of the total without sc1: `{code[i][1]}`")
    if not stores:
        problems.append(f"{name}: no store of the running total found")
    # (3) the hand-over: compare-and-swap on the turn word behind the adding code
    swaps = [i for i in range(first_total, len(code)) if code[i][1].startswith("global_atomic_cmpswap") and turn_base in code[i][1]]
    if not swaps:
        problems.append(f"{name}: no compare-and-swap on the turn word {turn_base} behind the adding code")
    for i in swaps:
        j, ok = i - 1, False
        while j >= first_total:
            t = code[j][1]
            if t.startswith("s_waitcnt") and all(c in t for c in ("vmcnt(0)", "lgkmcnt(0)")):
                ok = True
                break
            if t.startswith(VMEM) or t.startswith("ds_"):
                break
            j -= 1
        if not ok:
            problems.append(f"{name} @{code[i][0]:x}: the turn is handed on without waiting for this wave's stores")
    # (4) the collecting form: sums are loaded behind the ticket
    if collect_loads:
        tickets = [i for i in range(tail, first_poll) if re.match(r"global_atomic_add v\d+, ", code[i][1]) and "sc0" in code[i][1]]
        if not tickets:
            problems.append(f"{name}: loads in the tail before the poll loop, but no ticket draw")
        else:
            t0 = tickets[0]
            waited = any(code[j][1].startswith("s_waitcnt") and "vmcnt(0)" in code[j][1] for j in range(t0 + 1, min(collect_loads)))
            # the collecting form ends with the (ordinary) stores of c * total into `a`: a load behind them and ahead of the poll
            # loop belongs to the turns' adding code and has been moved above the polls
            ends = [i for i in range(t0, first_poll) if code[i][1].startswith("global_store")]
            end = max(ends) if ends else t0
            for i in collect_loads:
                if i < t0 or not waited or i > end:
                    problems.append(f"{name} @{code[i][0]:x}: load in the tail ahead of the ticket / of the poll loop: `{code[i][1]}`")
    return (len(polls), len(total_loads), len(stores), len(swaps)), problems


def kernels(lib_path):
    return {n: c for n, c in functions(disassemble(lib_path)).items() if KERNEL in n and c}


def check(lib_path):
    facts, problems = {}, []
    for name, code in kernels(lib_path).items():
        f, p = check_function(name, code)
        facts[name] = f
        problems += p
    return facts, problems


def self_test(lib_path):
    """Breaks each property in turn on every instantiation: returns (mutations tried, mutations NOT reported)."""
    tried = missed = 0
    for name, code in kernels(lib_path).items():
        assert not check_function(name, code)[1], "the unmodified program must be clean"
        marks = [i for i, (_, ins, _) in enumerate(code) if ins.startswith("s_barrier") or ins.startswith("s_load_dwordx16")]
        tail = marks[-1] + 1
        polls = [i for i in range(tail, len(code)) if _is_poll(code[i][1])]
        total = [i for i in range(polls[0] + 1, len(code)) if code[i][1].startswith("global_load") and i not in polls]
        stores = [i for i in range(min(total), len(code)) if code[i][1].startswith("global_store")]
        mutants = []
        for i in total + stores:   # (1) an access of the total that goes through the L2
            mutants.append(code[:i] + [(code[i][0], re.sub(r"\s*\bsc1\b", "", code[i][1]), code[i][2])] + code[i + 1:])
        # (2) a load of the total hoisted above the poll loop
        mutants.append(code[:polls[0]] + [(code[polls[0]][0] - 1, code[total[0]][1], None)] + code[polls[0]:])
        for i, (addr, ins, tgt) in enumerate(code):   # (3) the wait before the hand-over
            if i > min(total) and ins.startswith("s_waitcnt") and "vmcnt(0)" in ins and "lgkmcnt(0)" in ins and "expcnt(0)" in ins:
                mutants.append(code[:i] + [(addr, "s_nop 0", None)] + code[i + 1:])
        for m in mutants:
            tried += 1
            if not check_function(name, m)[1]:
                missed += 1
    return tried, missed


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "stdpar-nbody_amd", "libnbody_hip.so")
    facts, bad = check(lib)
    for b in bad:
        print(b)
    print(f"{len(facts)} instantiations of {KERNEL}: polls / loads of the total / stores / hand-overs per kernel "
          f"{sorted(set(facts.values()))}; {len(bad)} violation(s)")
    tried, missed = self_test(lib)
    print(f"self-test: {tried} mutations, {missed} not reported")
    sys.exit(1 if bad or not facts or missed or not tried else 0)
