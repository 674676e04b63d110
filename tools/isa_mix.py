#!/usr/bin/env python3
"""Instruction mix of a region of a gfx950 .s file (hipcc -S --cuda-device-only), by issue class.

    tools/isa_mix.py file.s [first_line last_line]

Classes follow profiles/r2_ubench_valu_issue_rates.txt: `fast` VALU (add/sub/and/or/xor/lshr/ashr/mov and the f32
add/mul/fma family) issue at the full rate, every other VALU form (`slow`: packed, DPP, SDWA, three-operand, min/max,
shifts left, perm/bfi ...) at half of it."""
import collections
import re
import sys

FAST = re.compile(r"^v_(add|sub|subrev)_(u32|f32|co_u32)(_e32|_e64)?$|^v_(and|or|xor)_b32(_e32|_e64)?$|"
                  r"^v_(lshrrev|ashrrev)_(b32|i32)(_e32|_e64)?$|^v_mov_b32(_e32|_e64)?$|^v_(mul|fma|fmac)_f32(_e32|_e64)?$|"
                  r"^v_not_b32(_e32)?$")


def classify(op, rest):
    if op.startswith("v_"):
        if "dpp" in op or "sdwa" in op or "row_" in rest or "wave_sh" in rest:
            return "valu_slow_dpp"
        return "valu_fast" if FAST.match(op) else "valu_slow"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    lines = open(path).read().split("\n")
    lo = int(sys.argv[2]) - 1 if len(sys.argv) > 2 else 0
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(lines)
    mix = collections.Counter()
    ops = collections.Counter()
    for ln in lines[lo:hi]:
        s = ln.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":") or s.startswith("//"):
            continue
        m = re.match(r"^([a-z_0-9]+)\s*(.*)$", s)
        if not m:
            continue
        op, rest = m.group(1), m.group(2)
        c = classify(op, rest)
        mix[c] += 1
        ops[(c, op)] += 1
    total = sum(mix.values())
    for c, n in mix.most_common():
        print(f"{c:16s} {n:6d} {100.0 * n / total:5.1f}%")
    print("total", total)
    if "-v" in sys.argv:
        for (c, op), n in sorted(ops.items(), key=lambda kv: -kv[1]):
            print(f"  {c:14s} {op:28s} {n}")


if __name__ == "__main__":
    main()
