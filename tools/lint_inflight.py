#!/usr/bin/env python3
"""Lint of the generated gfx950 assembly for registers with a hand-issued load in flight.

The MFMA kernels issue some loads and LDS reads by inline asm, which the compiler does not track, and wait for
them with hand-counted s_waitcnt.  Between such an instruction ("; PQI <reg>" precedes it in the asm text) and
the point from which its destination may be read ("; PQL <reg>"), the compiler must neither read nor write that
register -- it would read stale data, or have its value overwritten when the load returns -- and control must
not branch (a copy could be inserted at a join or a back edge, and this linear scan could not follow it).
A load announced by "; PQX <reg>" is meant to stay in flight across branches (a prologue that requests its
operands before it builds tables under run-time conditions): for those registers the branch rule is waived, and the
read / write rule holds for every instruction between issue and landing in text order, whatever the control flow.

  python tools/lint_inflight.py build/*.s      exit status 1 on any violation
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def lint(path):
    bad = 0
    func, inflight, pending_issue, across = None, {}, None, set()
    for ln, raw in enumerate(open(path), 1):
        line = raw.strip()
        if not line:
            continue
        if raw[0] not in " \t" and line.endswith(":") and not line.startswith("."):
            func, inflight, pending_issue, across = line[:-1], {}, None, set()
            continue
        if line.startswith("; PQI") or line.startswith("; PQX"):
            pending_issue = regs_of(line)
            if line.startswith("; PQX"):
                across |= pending_issue
            continue
        if line.startswith("; PQL"):
            for r in regs_of(line):
                inflight.pop(r, None)
                across.discard(r)
            continue
        if line.startswith(";") or line.startswith("."):
            continue
        code = line.split(";")[0]
        op = code.split()[0] if code.split() else ""
        if pending_issue is not None:           # the instruction the marker announces
            dest = regs_of(code.split(",")[0])
            if dest != pending_issue:
                print(f"{path}:{ln}: marker registers {sorted(pending_issue)} != destination of '{code.strip()}'")
                bad += 1
            srcs = regs_of(",".join(code.split(",")[1:]))
            hit = srcs & set(inflight)
            if hit:
                print(f"{path}:{ln}: [{func}] address of a hand-issued load reads in-flight v{sorted(hit)}: {code.strip()}")
                bad += 1
            for r in dest:
                if r in inflight:
                    print(f"{path}:{ln}: [{func}] v{r} loaded again while in flight since line {inflight[r]}")
                    bad += 1
                inflight[r] = ln
            pending_issue = None
            continue
        if not inflight:
            continue
        if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_endpgm"):
            held = set(inflight) - across if op != "s_endpgm" else set(inflight)
            if held:
                print(f"{path}:{ln}: [{func}] {op} with loads in flight into v{sorted(held)}")
                bad += 1
            if op == "s_endpgm":
                inflight, across = {}, set()
            continue
        hit = regs_of(code) & set(inflight)
        if hit:
            print(f"{path}:{ln}: [{func}] touches in-flight v{sorted(hit)} (issued at line "
                  f"{min(inflight[r] for r in hit)}): {code.strip()}")
            bad += 1
    return bad


def main():
    total = sum(lint(p) for p in sys.argv[1:])
    print("in-flight lint: %d violation(s) in %d file(s)" % (total, len(sys.argv) - 1))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
