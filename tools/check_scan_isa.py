#!/usr/bin/env python3
"""Guard for the hand-issued loads of k_fwd_scan (csrc/chmc_wave.h).

The forward scan issues its `global_load_dwordx4` through inline asm and waits with counted `s_waitcnt vmcnt(N)`
statements that the compiler knows nothing about, so nothing but register allocation keeps the compiler from
touching a destination register while its load is still in flight (a spill, a copy to an AGPR or a re-materialised
move would read stale data without any diagnostic).  This script compiles the device code to assembly and replays
every k_fwd_scan kernel's instruction stream:

  * asm loads enter a FIFO (loads retire in issue order), a counted wait pops it down to N entries, vmcnt(0) empties it;
  * any other instruction that reads or writes a register of a load still in the FIFO is a violation;
  * the kernels must not use scratch memory.

The replay follows the control flow (both ways at every conditional branch, every (position, loads in flight) state once),
so peeled or rotated loops are covered whatever order the compiler lays the blocks out in.

Second audit, for every kernel that issues stores the compiler does not see (`st_async` in the reverse sweeps and the
forward grad-log-det sweep): counted waits stay valid next to an untracked store only because loads
and stores retire in issue order (MI355X_MICROARCH.md, "s_waitcnt vmcnt(N)") -- true for global_/scratch_ accesses, not
for flat_ ones -- and the store's data registers may be rewritten only after the wait states of its trailing s_nop:
  * every hand-issued store is followed by its `s_nop` inside the same asm block (nothing, in particular no spill
    code, can be scheduled between the two);
  * the kernel contains no flat_ memory instruction;
  * spill code (scratch_) is reported; the instantiations listed in NO_SPILL must have none.
usage: check_scan_isa.py [chmc.hip] [-DNAME=VALUE ...]      exit status 0 = clean"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check_kernel(name, body):
    problems = []
    if any("scratch_" in l for l in body):
        problems.append("uses scratch memory")
    # instruction stream with asm markers resolved
    stream, in_asm = [], False
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            if t.endswith(":"):
                stream.append(("label", t[:-1], False))
            continue
        stream.append(("ins", t, in_asm))
    # replay along the control flow: every path from the entry, a (position, loads in flight) pair is expanded once.  (A
    # replay in layout order -- the first version of this guard -- reported loads "in flight" at their consumers whenever
    # the compiler peeled the first trip of the ring loop and parked the peeled blocks behind the loop.)
    labels = {t: i for i, (k, t, _) in enumerate(stream) if k == "label"}
    n_loads = sum(1 for k, t, a in stream if k == "ins" and a and t.startswith("global_load_dwordx4"))
    seen, reported = set(), set()
    work = [(0, ())]
    while work:
        idx, fifo = work.pop()
        fifo = list(fifo)
        while idx < len(stream):
            k, t, in_asm = stream[idx]
            if k == "label":
                key = (idx, tuple(fifo))
                if key in seen:
                    break
                seen.add(key)
                if len(seen) > 400000:
                    problems.append("control-flow replay did not terminate (state explosion)")
                    work = []
                    break
                idx += 1
                continue
            if t.startswith("s_endpgm"):
                break
            if t.startswith("s_branch"):
                idx = labels[t.split()[-1]]
                continue
            if t.startswith("s_cbranch"):
                work.append((labels[t.split()[-1]], tuple(fifo)))
                idx += 1
                key = (idx, tuple(fifo))
                if key in seen:
                    break
                seen.add(key)
                continue
            if t.startswith(("s_setpc", "s_swappc")):
                problems.append(f"indirect jump / call in a kernel with hand-issued loads: {t}")
                break
            idx += 1
            if in_asm and t.startswith("global_load_dwordx4"):
                fifo.append(frozenset(regs_of(t.split(",")[0])))
                continue
            m = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                if in_asm:
                    del fifo[: max(0, len(fifo) - n)]
                elif n == 0:
                    fifo.clear()
                continue
            if t.startswith(("s_", "ds_")) and "v" not in t.split(None, 1)[-1]:
                continue
            if fifo:
                used = regs_of(t.split(None, 1)[1] if " " in t else "")
                for dst in fifo:
                    if used & dst:
                        if t not in reported:
                            reported.add(t)
                            problems.append(f"touches in-flight load registers {sorted(used & dst)[:4]}: {t}")
                        break
    if n_loads == 0:
        problems.append("no hand-issued loads found (kernel changed?)")
    return problems


NO_SPILL = ("k_newton_leanINS_8FhnModelELi7ELb0", "k_rev_waveINS_8FhnModelELi7E", "k_gld_fwd_waveINS_8FhnModelELi7E",
            "k_gld_bwd_waveINS_8FhnModelELi7E", "k_gld_fwd_qxINS_8FhnModelELi7E")  # the hot instantiations of the headline configuration


def audit_stores(name, body):
    """-> (problems, number of hand-issued stores, number of scratch instructions)"""
    problems, n_st = [], 0
    for j, l in enumerate(body):
        t = l.strip()
        if t.startswith("global_store") and j > 0 and "ASMSTART" in body[j - 1]:
            n_st += 1
            if not (j + 2 < len(body) and body[j + 1].strip().startswith("s_nop") and "ASMEND" in body[j + 2]):
                problems.append(f"hand-issued store without its s_nop in the same asm block: {t}")
    n_scr = sum(1 for l in body if l.strip().startswith("scratch_"))
    if n_st and any(l.strip().startswith("flat_") for l in body):
        problems.append("flat_ memory instruction in a kernel with untracked stores (vmcnt order no longer guaranteed)")
    if n_scr and any(k in name for k in NO_SPILL):
        problems.append(f"{n_scr} scratch instructions in an instantiation that must not spill")
    return problems, n_st, n_scr


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]  # (variant builds: -DCHMC_SCAN_NB=2 ...)
    args = [a for a in sys.argv[1:] if not a.startswith("-D")]
    src = args[0] if args else os.path.join(ROOT, "manifold_mcmc_for_diffusions_amd", "csrc", "chmc.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "chmc.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-S", "--cuda-device-only",
                               "-o", out, src] + defs)
        lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN4chmc10k_fwd_scan\w*:", l)]
    bad = 0
    for s in starts:
        e = next(i for i in range(s, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel"))
        name = lines[s].split(":")[0]
        probs = check_kernel(name, lines[s:e])
        print(("FAIL " if probs else "ok   ") + name)
        for p in probs[:10]:
            print("     " + p)
        bad += bool(probs)
    if not starts:
        print("no k_fwd_scan kernels found")
        return 2
    n_audit = 0
    for s in [i for i, l in enumerate(lines) if re.match(r"^_ZN4chmc\w+:", l)]:
        name = lines[s].split(":")[0]
        if "k_fwd_scan" in name:
            continue  # (its stores belong to the helper wavefront and are waited for by the compiler; audited above)
        e = next(i for i in range(s, len(lines)) if lines[i].strip().startswith((".amdhsa_kernel", ".section")))
        probs, n_st, n_scr = audit_stores(name, lines[s:e])
        if not n_st and "k_newton_lean" not in name and not any(k in name for k in NO_SPILL):
            continue
        n_audit += 1
        print(("FAIL " if probs else "ok   ") + f"{name}  [{n_st} untracked stores, {n_scr} scratch instructions]")
        for p in probs[:10]:
            print("     " + p)
        bad += bool(probs)
    if n_audit == 0:
        print("no kernel with hand-issued stores found (kernels changed?)")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
