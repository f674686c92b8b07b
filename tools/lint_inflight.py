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
"; PQR <reg>" is the same with a second check that does not depend on how the compiler laid the blocks out: no
instruction that control can reach from the issue before it reaches a landing marker of the register -- following
both sides of every conditional branch -- may touch it.  (For code whose branches between issue and landing do not
depend on whether the load was issued; a load issued under a condition and consumed under the same condition would
fail it on the paths that cannot happen.)

  python tools/lint_inflight.py build/*.s      exit status 1 on any violation
"""
import re
import sys

FUNC = re.compile(r"^([A-Za-z_][A-Za-z0-9_$.]*):")     # a function's label (local labels start with ".L")
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
        m = FUNC.match(raw)
        if m:
            func, inflight, pending_issue, across = m.group(1), {}, None, set()
            continue
        if line.startswith("; PQI") or line.startswith("; PQX") or line.startswith("; PQR"):
            pending_issue = regs_of(line)
            if not line.startswith("; PQI"):
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
            # (a "; PQX" / "; PQR" load still in flight at s_endpgm is harmless: the hardware waits for the wave's
            # outstanding memory operations first; a "; PQI" one that never landed is a forgotten marker)
            held = set(inflight) - across
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


LABEL = re.compile(r"^(\.L[A-Za-z0-9_$]+):")


def lint_reach(path):
    """Second pass for the "; PQR" loads: walk the control flow from each issue to the landing markers."""
    bad = 0
    funcs, cur = [], None
    for ln, raw in enumerate(open(path), 1):
        line = raw.strip()
        if not line:
            continue
        m = FUNC.match(raw)
        if m:
            cur = {"name": m.group(1), "items": [], "labels": {}}
            funcs.append(cur)
            continue
        if cur is None:
            continue
        m = LABEL.match(raw)
        if m:
            cur["labels"][m.group(1)] = len(cur["items"])
            continue
        if line.startswith("; PQR"):
            cur["items"].append((ln, "issue", regs_of(line), line))
        elif line.startswith("; PQL"):
            cur["items"].append((ln, "land", regs_of(line), line))
        elif line.startswith(";") or line.startswith("."):
            continue
        else:
            code = line.split(";")[0].strip()
            if code:
                cur["items"].append((ln, "op", regs_of(code), code))
    for f in funcs:
        items, labels = f["items"], f["labels"]
        for start, (ln0, kind, regs0, _) in enumerate(items):
            if kind != "issue":
                continue
            for r in regs0:
                seen, stack, reported = set(), [start + 2], False   # start + 1 is the load itself
                while stack and not reported:
                    i = stack.pop()
                    while i < len(items) and i not in seen:
                        seen.add(i)
                        ln, k, regs, text = items[i]
                        if k == "land":
                            if r in regs:
                                break
                            i += 1
                            continue
                        if k == "issue":
                            i += 1          # the marker; the load behind it is checked as an instruction
                            continue
                        if r in regs:
                            print(f"{path}:{ln}: [{f['name']}] reachable with v{r} in flight (issued at line {ln0}): {text}")
                            bad += 1
                            reported = True
                            break
                        op = text.split()[0]
                        if op == "s_endpgm":
                            break
                        if op == "s_setpc_b64":
                            print(f"{path}:{ln}: [{f['name']}] indirect jump with v{r} in flight (issued at line {ln0})")
                            bad += 1
                            reported = True
                            break
                        if op == "s_branch" or op.startswith("s_cbranch"):
                            t = labels.get(text.split()[-1])
                            if t is None:
                                print(f"{path}:{ln}: [{f['name']}] branch to an unknown label with v{r} in flight: {text}")
                                bad += 1
                                reported = True
                                break
                            stack.append(t)
                            if op == "s_branch":
                                break
                        i += 1
    return bad


def main():
    total = sum(lint(p) + lint_reach(p) for p in sys.argv[1:])
    print("in-flight lint: %d violation(s) in %d file(s)" % (total, len(sys.argv) - 1))
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
