#!/usr/bin/env python3
"""tools/micro/check_inflight_regs.py <file.s> [kernel-name-substring ...]

The loader waves of the round-6 persistent kernels (csrc/vit_mega.hip) carry weight stages in registers written by INLINE-ASM
global_load_dwordx4 (saddr form) and consumed by inline-asm ds_write_b128: the compiler does not know the loads are asynchronous, so a
copy, a spill or a re-use of such a register quad between the load and the store would move bytes that have not landed (the k-loop's
counted s_waitcnt in front of its first barrier is what makes the store safe).  This walks the control-flow graph of each kernel in the
compiler's assembly (hipcc -S / -save-temps) with a forward may-analysis — in-flight quads: set by the load, cleared by the ds_write
that names the quad, merged by union at joins — and reports every other instruction that mentions a register of a quad in flight.
Exit code 1 on any violation."""
import re
import sys

LD = re.compile(r"^global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\].*mg-stage")    # (the marker is a comment inside the asm text)
ST = re.compile(r"^ds_write_b128 v\d+, v\[(\d+):(\d+)\].*mg-stage")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LABEL = re.compile(r"^(\.?[A-Za-z_][\w.$]*):")
BR = re.compile(r"^s_(branch|cbranch_\w+)\s+(\.?[\w.$]+)")


def kernels(lines):
    """-> {name: [instruction / label strings]} for every .amdhsa kernel body in the file"""
    out, cur, name = {}, None, None
    for ln in lines:
        code = ln.rstrip() if "mg-stage" in ln else ln.split(";")[0].rstrip()
        m = re.match(r"^(_Z\w+):", code)
        if m and cur is None:
            name, cur = m.group(1), []
            continue
        if cur is None:
            continue
        s = code.strip()
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".section") or s.startswith(".rodata"):
            out[name] = cur
            cur = None
            continue
        if s and not s.startswith(".") or LABEL.match(s):
            cur.append(s)
    return out


def check(name, ins):
    # basic blocks
    leaders = {0}
    labels = {}
    for i, s in enumerate(ins):
        m = LABEL.match(s)
        if m:
            labels[m.group(1)] = i
            leaders.add(i)
        if BR.match(s) or s.startswith("s_endpgm") or s.startswith("s_setpc"):
            leaders.add(i + 1)
    order = sorted(x for x in leaders if x < len(ins))
    bid = {}
    for b, st in enumerate(order):
        en = order[b + 1] if b + 1 < len(order) else len(ins)
        for i in range(st, en):
            bid[i] = b
    succ = [[] for _ in order]
    for b, st in enumerate(order):
        en = (order[b + 1] if b + 1 < len(order) else len(ins)) - 1
        last = ins[en]
        m = BR.match(last)
        if m:
            tgt = labels.get(m.group(2))
            if tgt is not None:
                succ[b].append(bid[tgt])
            if m.group(1) != "branch" and en + 1 < len(ins):
                succ[b].append(bid[en + 1])
        elif last.startswith("s_endpgm") or last.startswith("s_setpc"):
            pass
        elif en + 1 < len(ins):
            succ[b].append(bid[en + 1])
    # forward may-analysis
    instate = [frozenset() for _ in order]
    work = list(range(len(order)))
    viol = {}
    nld = 0

    def transfer(b, state, report):
        nonlocal nld
        st = order[b]
        en = order[b + 1] if b + 1 < len(order) else len(ins)
        cur = set(state)
        for i in range(st, en):
            s = ins[i]
            if LABEL.match(s):
                continue
            m = LD.match(s)
            if m:
                a = int(m.group(1))
                if report:
                    nld += 1
                    if a in cur:
                        viol[(i, a)] = f"quad v[{a}:{a + 3}] loaded again while in flight: {s}"
                # the address register of the load itself may not be in flight either
                cur.add(a)
                continue
            m = ST.match(s)
            if m and int(m.group(1)) in cur:
                cur.discard(int(m.group(1)))
                continue
            if not cur:
                continue
            used = set()
            for g in REG.finditer(s):
                if g.group(1) is not None:
                    used.add(int(g.group(1)))
                else:
                    used.update(range(int(g.group(2)), int(g.group(3)) + 1))
            if report:
                for a in cur:
                    if used & set(range(a, a + 4)):
                        viol[(i, a)] = f"v[{a}:{a + 3}] in flight: {s}"
        return frozenset(cur)

    outstate = [None] * len(order)
    while work:
        b = work.pop()
        o = transfer(b, instate[b], False)
        if o != outstate[b]:
            outstate[b] = o
            for t in succ[b]:
                n = instate[t] | o
                if n != instate[t]:
                    instate[t] = n
                    if t not in work:
                        work.append(t)
                elif outstate[t] is None and t not in work:
                    work.append(t)
    for b in range(len(order)):
        transfer(b, instate[b], True)
    for (i, a), msg in sorted(viol.items()):
        print(f"{name}: instruction {i}: {msg}")
    print(f"{name}: {len(ins)} lines, {len(order)} blocks, {nld} register-stage loads, {len(viol)} violations")
    return len(viol), nld


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2:]
    bad = 0
    seen = 0
    for name, ins in kernels(lines).items():
        if want and not any(w in name for w in want):
            continue
        v, n = check(name, ins)
        bad += v
        seen += n
    if want and not seen:
        print("no register-stage load found in the named kernels")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
