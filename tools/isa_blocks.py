#!/usr/bin/env python3
"""Instruction counts per basic block of one function of a gfx950 .s file (hipcc -S --cuda-device-only): total, VALU, scratch,
LDS, VMEM, s_nop, s_waitcnt, s_barrier -- blocks of at least `min` instructions, or with scratch accesses.
    tools/isa_blocks.py file.s <mangled function name prefix> [min = 150]"""
import re
import sys

path, prefix = sys.argv[1], sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 150
text = open(path).read()
names = sorted({l.split(":")[0] for l in text.split("\n") if l.startswith(prefix) and ":" in l})
for name in names:
    i = text.index("\n" + name + ":")
    j = text.index(".Lfunc_end", i)
    blocks, cur = [], ["entry", 0, 0, 0, 0, 0, 0, 0, 0]
    for line in text[i:j].split("\n"):
        t = line.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append(cur)
            cur = [m.group(1) + (" (loop header)" if "Loop Header" in line else ""), 0, 0, 0, 0, 0, 0, 0, 0]
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        cur[1] += 1
        cur[2] += op.startswith("v_")
        cur[3] += op.startswith("scratch_")
        cur[4] += op.startswith("ds_")
        cur[5] += op.startswith(("buffer_", "global_", "flat_"))
        cur[6] += op.startswith("s_nop")
        cur[7] += op.startswith("s_waitcnt")
        cur[8] += op.startswith("s_barrier")
    blocks.append(cur)
    print(f"== {name}: {sum(b[1] for b in blocks)} instructions, {sum(b[2] for b in blocks)} VALU, {sum(b[3] for b in blocks)} scratch accesses")
    print(f"   {'block':28s} {'all':>6s} {'valu':>6s} {'scratch':>7s} {'lds':>5s} {'vmem':>5s} {'nop':>5s} {'wait':>5s} {'barrier':>7s}")
    for b in blocks:
        if b[1] >= least or b[3] > 0:
            print(f"   {b[0]:28s} {b[1]:6d} {b[2]:6d} {b[3]:7d} {b[4]:5d} {b[5]:5d} {b[6]:5d} {b[7]:5d} {b[8]:7d}")
