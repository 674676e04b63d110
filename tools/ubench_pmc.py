#!/usr/bin/env python3
"""Counter view of tools/ubench_valu: per measured launch, cycles per VALU wave-instruction and SIMD from PMC.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
        --output-format csv -d DIR -- tools/bin/ubench_valu --waves 2,4,8 "add ind" ...
    tools/ubench_pmc.py DIR [simds=1024]

Every test launches twice per occupancy (a 16-iteration warm-up, then the measured one): the dispatch with the larger
SQ_INSTS_VALU of each (kernel, grid) pair is the measured one.  Columns:
  W            workgroups of four waves per CU = waves per SIMD (grid / 256 threads / CUs / rounds)
  insts        SQ_INSTS_VALU, wave-instructions of the launch
  gui_cyc      GRBM_GUI_ACTIVE: shader-clock cycles the launch kept the chip busy
  cyc/inst     gui_cyc x SIMDs / insts  -- cycles per wave-instruction and SIMD, no wall clock, no s_memtime involved
  waves        SQ_WAVES (residency check: = grid / 64)
  valu_busy    SQ_ACTIVE_INST_VALU x 4 / (gui_cyc x SIMDs): the share of SIMD-cycles in which some wave was in a VALU
               instruction (quad-cycle units; can exceed 1 when waves of one SIMD overlap)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    simds = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    acc = defaultdict(lambda: defaultdict(float))
    meta = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            disp = int(row["Dispatch_Id"])
            acc[disp][row["Counter_Name"]] += float(row["Counter_Value"])
            meta[disp] = (row["Kernel_Name"].split("(")[0], int(row["Grid_Size"]))
    best = {}
    for disp, c in acc.items():
        key = meta[disp]
        if key not in best or c.get("SQ_INSTS_VALU", 0) > acc[best[key]].get("SQ_INSTS_VALU", 0):
            best[key] = disp
    print(f"{'kernel':28s} {'W':>2s} {'insts':>12s} {'gui_cyc':>11s} {'cyc/inst':>8s} {'waves':>7s} {'valu_busy':>9s} {'wave_cyc/inst/W':>15s}")
    for (k, grid), disp in sorted(best.items(), key=lambda kv: kv[1]):
        c = acc[disp]
        w = grid // 256 // (simds // 4) // rounds
        insts, gui = c.get("SQ_INSTS_VALU", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        cpi = gui * simds / insts if insts else 0.0
        busy = c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (gui * simds) if gui else 0.0
        wcpi = c.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / insts / max(w, 1) if insts else 0.0
        print(f"{k[-28:]:28s} {w:2d} {insts:12.0f} {gui:11.0f} {cpi:8.3f} {c.get('SQ_WAVES', 0):7.0f} {busy:9.3f} {wcpi:15.3f}")


if __name__ == "__main__":
    main()
