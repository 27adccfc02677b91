#!/usr/bin/env python3
"""Static check of the gfx950 ISA the compiler generated for our inline-asm DPP instructions.

Rule (CDNA ISA, "VALU writes VGPR followed by a DPP read of that VGPR requires 2 wait states"): for every
v_*_dpp instruction, none of the VGPRs of its DPP source operand (src0) may be written by a VALU/VMEM-return
... instruction among the immediately preceding instructions worth < 2 wait states (every instruction counts 1,
s_nop N counts N+1).  The compiler cannot see inside asm statements, so this script is the safety net that lets
row_fnma() run without per-instruction padding.  Usage: check_dpp_hazards.py file.s [...]; exit 1 on violation.
"""
import re
import sys

REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(op):
    m = REG.search(op)
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


QUIET = False


def check(path):
    bad = 0
    ndpp = 0
    window = []   # (wait_states, written_regs, text)
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line or line.startswith(".") or line.startswith("//"):
            continue
        if line.endswith(":"):
            window = [(99, set(), "label")]   # unknown predecessor: be conservative only about fall-through
            continue
        parts = line.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        if "_dpp" in op and op.startswith("v_"):
            ndpp += 1
            src0 = regs(args[1]) if len(args) > 1 else set()
            ws = 0
            for w, wr, txt in reversed(window):
                if ws >= 2:
                    break
                if wr & src0:
                    if not QUIET or bad < 3:
                        print("%s:%d: DPP hazard: '%s' reads %s written by '%s' only %d wait state(s) earlier" % (path, ln, line, sorted(wr & src0), txt, ws))
                    bad += 1
                ws += w
        written = set()
        w = 1
        if op == "s_nop":
            w = int(args[0], 0) + 1 if args else 1
        elif op.startswith("v_") and args and not op.startswith("v_cmp"):
            written = regs(args[0])
        elif (op.startswith("ds_read") or op.startswith("global_load") or op.startswith("buffer_load")) and args:
            written = regs(args[0])
        window.append((w, written, line))
        if len(window) > 8:
            window.pop(0)
    return bad, ndpp


if __name__ == "__main__":
    total = 0
    args = [a for a in sys.argv[1:] if a != "-q"]
    QUIET = "-q" in sys.argv[1:]
    for f in args:
        b, n = check(f)
        print("%s: %d DPP instructions checked, %d hazards" % (f, n, b))
        total += b
    sys.exit(1 if total else 0)
