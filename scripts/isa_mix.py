#!/usr/bin/env python
"""Instruction mix of one kernel in a gfx950 .s file (hipcc -save-temps): counts per mnemonic and per class, straight-line body.

    python scripts/isa_mix.py <file.s> <kernel-name-substring> [--top N]
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_fmac_f64_dpp") or op.startswith("v_fma_f64_dpp"): return "fp64 DPP fmac (elimination)"
    if op in ("v_rsq_f64_e32", "v_rcp_f64_e32", "v_sqrt_f64_e32", "v_rsq_f64", "v_rcp_f64", "v_sqrt_f64", "v_log_f32_e32", "v_exp_f32_e32"): return "transcendental (quarter rate)"
    if op.startswith("v_mov_b64_dpp") or op.startswith("v_mov_b32_dpp"): return "DPP mov (broadcast)"
    if re.match(r"v_(fma|fmac|mul|add|ldexp|rndne|fract|floor|trunc|max|min|div_fixup|div_fmas|div_scale|frexp_mant)_f64", op): return "fp64 arithmetic"
    if op.startswith("v_cvt"): return "convert"
    if op.startswith("v_cndmask"): return "select (v_cndmask)"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"): return "compare"
    if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane") or op.startswith("v_swap"): return "move"
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "integer / logic VALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vector memory"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier"): return "wait / barrier"
    if op.startswith("s_"): return "scalar"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 25
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and name in l.split(":")[0]:
            start = i; break
    if start is None:
        raise SystemExit("kernel not found")
    ops = collections.Counter(); cls = collections.Counter()
    meta = {}
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".end_amdhsa_kernel") or t.startswith(".Lfunc_end"):
            break
        if t.startswith("s_endpgm"):
            ops["s_endpgm"] += 1; continue
        if not t or t.startswith(";") or t.startswith(".") or re.match(r"^[\w$.]+:", t):
            continue
        op = t.split()[0]
        ops[op] += 1; cls[classify(op)] += 1
    for l in lines[start:]:
        m = re.search(r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", l) or re.search(r"\.amdhsa_(group_segment_fixed_size) (\d+)", l)
        if m and m.group(1) not in meta: meta[m.group(1)] = int(m.group(2))
        if len(meta) >= 5: break
    print("kernel:", lines[start][:-1][:140])
    print("resources:", meta)
    total = sum(cls.values())
    valu = sum(v for k, v in cls.items() if k not in ("LDS", "vector memory", "s_nop", "wait / barrier", "scalar", "other"))
    print("instructions: %d total, %d VALU" % (total, valu))
    # issue-cost model measured with scripts/ubench/coissue.hip: every VALU op = 1 unit (4 cycles / wave64), quarter-rate ops 4.2 units
    units = sum(v * (4.2 if k.startswith("transcendental") else 1.0) for k, v in cls.items() if k not in ("LDS", "vector memory", "s_nop", "wait / barrier", "scalar", "other"))
    print("VALU issue units (transcendental = 4.2): %.0f" % units)
    for k, v in cls.most_common():
        print("  %-34s %6d  %5.1f %%" % (k, v, 100.0 * v / total))
    print("top mnemonics:")
    for k, v in ops.most_common(top):
        print("  %-34s %6d" % (k, v))


if __name__ == "__main__":
    main()
