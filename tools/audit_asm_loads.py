#!/usr/bin/env python3
"""Audit of the hand-written assembly inside k_sg_runs32 in hipcc's assembly output (-save-temps).

The tile loads are issued from inline asm; hipcc believes their destination registers are written when the statement
ends.  Between an issue and the counted wait that names the same registers, no other instruction may touch them (a copy
or spill there would read registers whose loads have not landed), and the kernel must not spill vector registers at all.
The kernel also places one VALU instruction by hand (`v_mad_i32_i16`, the first tap of every numerator).  gfx950 needs wait
states between a dot instruction (v_dot*) and a DIFFERENT vector instruction that reads its result; hipcc pads the
instructions it schedules itself but knows nothing about the operands of inline assembly (an inline
v_dot2 in that place gave nondeterministic results).  So: no inline vector instruction may read a register that a v_dot*
wrote within the last kHazardWindow instructions.
Exit status 1 with the offending lines if any of this does not hold."""
import re
import sys


def regs_of(text: str) -> set:
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


kHazardWindow = 4  # instructions (more than the wait states any dot -> VALU dependency needs)


def dot_hazards(name: str, lines: list) -> int:
    """Inline-asm vector instructions whose sources were written by a v_dot* just before them."""
    bad = 0
    recent = []  # (line number, text, registers written) of the last few instructions
    in_asm = False
    for k, raw in enumerate(lines):
        ln = raw.strip()
        if ";;#ASMSTART" in ln:
            in_asm = True
            continue
        if ";;#ASMEND" in ln:
            in_asm = False
            continue
        code = ln.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        ops = code.split(None, 1)
        operands = [o.strip() for o in ops[1].split(",")] if len(ops) > 1 else [""]
        dst = regs_of(operands[0]) if ops[0].startswith("v_") else set()
        if in_asm and ops[0].startswith("v_"):
            src = set()
            for o in operands[1:]:
                src |= regs_of(o)
            for (kk, text, written) in recent:
                if text.startswith("v_dot") and written & src:
                    print(f"{name}: line {k}: inline `{code}` reads the result of `{text}` (line {kk}) inside the dot hazard window")
                    bad += 1
        recent.append((k, code, dst))
        recent = recent[-kHazardWindow:]
    return bad


def flow_audit(name: str, lines: list) -> int:
    """Control-flow-aware form of the in-flight rule, for kernels whose tile buffers are fixed registers named in the
    assembly text (issue statements without outputs): a forward data-flow over the basic blocks -- a buffer register is
    'in flight' from a `buffer_load` inside an ASM block that writes it until an ASM `s_waitcnt vmcnt` statement whose
    comment names it; an instruction that touches a register which may be in flight on SOME path is an error."""
    # ---- instructions and blocks
    insts = []  # (line number, text, in_asm)
    labels = {}
    in_asm = False
    for k, raw in enumerate(lines):
        ln = raw.strip()
        if ";;#ASMSTART" in ln:
            in_asm = True
            continue
        if ";;#ASMEND" in ln:
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        code = ln if in_asm else ln.split(";")[0].strip()
        if not code or code.startswith(".") or code.startswith(";") or code.endswith(":"):
            continue
        insts.append((k, code, in_asm))
    n = len(insts)
    if n == 0:
        return 0

    def succ(i):
        op = insts[i][1].split()[0]
        out = []
        if op in ("s_endpgm", "s_setpc_b64"):
            return out
        if op == "s_branch" or op.startswith("s_cbranch"):
            tgt = insts[i][1].split()[-1]
            if tgt in labels and labels[tgt] < n:
                out.append(labels[tgt])
            if op == "s_branch":
                return out
        if i + 1 < n:
            out.append(i + 1)
        return out

    # ---- transfer: state = frozenset of registers that may be in flight BEFORE instruction i
    state = [None] * n
    state[0] = frozenset()
    work = [0]
    bad_lines = {}
    while work:
        i = work.pop()
        cur = set(state[i])
        k, code, asm = insts[i]
        body = code.split(";")[0]
        if asm and body.strip().startswith("buffer_load"):
            mm = re.search(r"buffer_load_dwordx4\s+(v\[\d+:\d+\])", body)
            if mm:
                cur |= regs_of(mm.group(1))
        elif asm and "s_waitcnt vmcnt" in body:
            named = regs_of(code.split(";", 1)[1]) if ";" in code else set(cur)
            cur -= named
        else:
            hit = regs_of(body) & cur
            if hit:
                bad_lines[k] = code
        new = frozenset(cur)
        for j in succ(i):
            merged = new if state[j] is None else (state[j] | new)
            if merged != state[j]:
                state[j] = merged
                work.append(j)
    for k in sorted(bad_lines):
        print(f"{name}: line {k}: `{bad_lines[k]}` touches registers that may be in flight there")
    return len(bad_lines)


def main(path: str, prefix: str = r"_ZN3wfa11k_sg_runs32") -> int:
    txt = open(path).read()
    bad = 0
    n_kernels = 0
    for fn in re.split(r"\n(?=_Z\w+:)", txt):
        if not re.match(prefix, fn):
            continue
        n_kernels += 1
        name = fn.split(":")[0]
        m = re.search(r"\.vgpr_spill_count:\s*(\d+)", fn) or re.search(r"; ScratchSize: (\d+)", fn)
        if m and int(m.group(1)) != 0:
            print(f"{name}: vector register spills ({m.group(0)})")
            bad += 1
        lines = fn.split("\n")
        bad += dot_hazards(name, lines)
        if re.search(r"s_waitcnt vmcnt\(\d+\) ; v\[\d+:\d+\]", fn):
            # the waits name their registers: the data-flow form covers every path (the linear walk below follows file order)
            bad += flow_audit(name, lines)
            continue
        i = 0
        while i < len(lines):
            if ";;#ASMSTART" in lines[i]:
                j = i + 1
                body = []
                while ";;#ASMEND" not in lines[j]:
                    body.append(lines[j])
                    j += 1
                if any("buffer_load_dwordx4" in b for b in body):
                    dst = set()
                    for b in body:
                        mm = re.search(r"buffer_load_dwordx4\s+(v\[\d+:\d+\])", b)
                        if mm:
                            dst |= regs_of(mm.group(1))
                    # walk forward (straight-line AND across labels in file order) to the next wait asm
                    k = j + 1
                    while k < len(lines):
                        ln = lines[k].strip()
                        if ";;#ASMSTART" in ln and "s_waitcnt vmcnt" in lines[k + 1]:
                            # a wait that lists its registers (as an assembly comment) only ends the walk when it names
                            # this issue's registers: with two tiles in flight the next wait is the other tile's
                            named = regs_of(lines[k + 1].split(";", 1)[1]) if ";" in lines[k + 1] else dst
                            if named & dst:
                                break
                        if ";;#ASMSTART" in ln and any("buffer_load_dwordx4" in x for x in lines[k + 1:k + 8]):
                            # the other tile's issue: its registers must be disjoint
                            other = set()
                            for x in lines[k + 1:k + 8]:
                                mm = re.search(r"buffer_load_dwordx4\s+(v\[\d+:\d+\])", x)
                                if mm:
                                    other |= regs_of(mm.group(1))
                            if other & dst:
                                break  # same tile re-issued (loop wrap in file order): stop this walk
                        if ln and not ln.startswith((";", ".")) and not ln.endswith(":"):
                            if regs_of(ln.split(";")[0]) & dst:
                                print(f"{name}: line {k}: `{ln}` touches in-flight registers of the issue at line {i}")
                                bad += 1
                        k += 1
                i = j
            i += 1
    if n_kernels == 0:
        print("no kernel matching", prefix, "found in", path)
        return 1
    print(f"audit_asm_loads: {n_kernels} kernels checked, {bad} problems")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:3]))
