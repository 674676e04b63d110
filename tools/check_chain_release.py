#!/usr/bin/env python3
"""tools/check_chain_release.py [file.s]: the release side of the chains' hand-off between workgroups, checked in the ISA.

In every k_smooth_*_chain<true> kernel (GROUPED: several workgroups per cost buffer, sn_pool_kernels.hip) each s_barrier
of the round loop must be preceded, with no vector-memory instruction in between, by an `s_waitcnt vmcnt(0)`: every wave
drains its own sc1 stores before the barrier after which the workgroup's round counter is published
(chain_release_barrier()).  Without a file the kernel file is compiled to ISA first (hipcc cross-compiles without a GPU).
Exit code 1 and a listing when a barrier is not covered.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "avisynth_sangnom2_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-flush-denormals-to-zero",
         "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only"]


def disassemble():
    out = os.path.join(tempfile.mkdtemp(prefix="sn_isa_"), "sn_pool_kernels.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, os.path.join(CSRC, "sn_pool_kernels.hip"), "-o", out], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def kernels(text):
    """name -> list of instruction lines (labels, directives and comments dropped)"""
    found, name = {}, None
    for line in text.split("\n"):
        m = re.match(r"^(_ZN2sn\d+k_smooth_\w+_chainILb1E\w*):", line)
        if m:
            name = m.group(1)
            found[name] = []
            continue
        if name is None:
            continue
        s = line.strip()
        if s.startswith(".Lfunc_end"):
            name = None
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        found[name].append(s.split(";")[0].strip())
    return found


VMEM = ("buffer_", "global_", "flat_", "scratch_")


def uncovered(code):
    bad = []
    for i, ins in enumerate(code):
        if not ins.startswith("s_barrier"):
            continue
        ok = False
        for prev in reversed(code[max(0, i - 64):i]):
            if prev.startswith(VMEM) or prev.startswith(("s_cbranch", "s_branch", "s_barrier")):
                break
            if prev.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", prev):
                ok = True
                break
        if not ok:
            bad.append((i, code[max(0, i - 5):i + 1]))
    return bad


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else disassemble()
    ks = kernels(open(path).read())
    if len(ks) < 3:
        print("expected the three grouped chain kernels, found", sorted(ks))
        return 1
    rc = 0
    for name, code in sorted(ks.items()):
        bars = sum(1 for c in code if c.startswith("s_barrier"))
        bad = uncovered(code)
        loop_bad = bad  # (these kernels have ONE barrier, at the head of the round loop; any other must be covered too)
        print(f"{name}: {bars} s_barrier, {len(loop_bad)} without a preceding s_waitcnt vmcnt(0)")
        for i, ctx in loop_bad:
            rc = 1
            print("   at instruction", i, "::", " | ".join(ctx))
    return rc


if __name__ == "__main__":
    sys.exit(main())
