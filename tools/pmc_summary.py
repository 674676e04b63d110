#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs over the library's kernels.

    pmc_summary.py <dir> [<dir> ...]                          per dispatch and counter -> CSV on stdout
    pmc_summary.py --json OUT.json --workload NAME --frames N --steps S <dir> [<dir> ...]
        per-step ("per launch") figures of one bench.py workload, merged into OUT.json under NAME.  A step of
        bench.py may be several kernels (the three sweeps of a 4:2:0 frame): every counter is summed over all
        library kernels of a pass and divided by the S steps the pass ran (warm-up included).

Units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE
counts one half of the bytes of a wide coalesced streaming read, so HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
FETCH_SIZE and WRITE_SIZE must come from separate passes (TCC slots)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = ("k_fused", "k_smooth", "k_prepare", "k_finalize", "k_assemble", "k_turn")


def rows(d):
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if any(k in row["Kernel_Name"] for k in KERNELS):
                yield row


def per_dispatch(dirs):
    out = csv.writer(sys.stdout, quoting=csv.QUOTE_MINIMAL, lineterminator="\n")
    out.writerow(["pass", "dispatch_id", "kernel", "grid_size", "counter", "sum_over_dims"])
    for d in dirs:
        if not os.path.isdir(d):
            continue
        acc = defaultdict(float)
        meta = {}
        for row in rows(d):
            key = (int(row["Dispatch_Id"]), row["Counter_Name"])
            acc[key] += float(row["Counter_Value"])
            meta[key] = (row["Kernel_Name"].split("(")[0][-48:], row.get("Grid_Size", ""))
        for (disp, ctr), v in sorted(acc.items()):
            out.writerow([os.path.basename(d), disp, meta[(disp, ctr)][0], meta[(disp, ctr)][1], ctr, v])


def per_step(dirs, steps):
    tot = defaultdict(float)
    kernels = set()
    for d in dirs:
        for row in rows(d):
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            kernels.add(row["Kernel_Name"].split("(")[0].replace("void ", ""))
    return {k: v / steps for k, v in tot.items()}, sorted(kernels)


def main():
    a = sys.argv[1:]
    if not a or a[0] != "--json":
        return per_dispatch(a)
    opts = {a[i]: a[i + 1] for i in range(0, 8, 2)}
    dirs = a[8:]
    steps = int(opts["--steps"])
    c, kernels = per_step(dirs, steps)
    rec = {"frames_per_launch": int(opts["--frames"]), "kernels": kernels, "steps_in_pass": steps}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rec["fetch_size_kib"] = round(c["FETCH_SIZE"], 1)
        rec["write_size_kib"] = round(c["WRITE_SIZE"], 1)
        rec["hbm_bytes_per_launch"] = int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
    if "SQ_INSTS_VALU" in c:
        rec["valu_insts_per_launch"] = int(c["SQ_INSTS_VALU"])
        rec["salu_insts_per_launch"] = int(c.get("SQ_INSTS_SALU", 0))
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            rec["valu_active_frac"] = round(c.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)
            rec["stall_frac"] = round((c.get("SQ_WAIT_ANY", 0.0) + c.get("SQ_WAIT_INST_ANY", 0.0)) / wc, 4)
            rec["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / wc, 4)
            rec["wait_inst_any_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
            rec["wave_quad_cycles_per_launch"] = int(wc)
    path = opts["--json"]
    try:
        allrec = json.load(open(path))
    except (OSError, ValueError):
        allrec = {}
    allrec[opts["--workload"]] = rec
    json.dump(allrec, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps({opts["--workload"]: rec}))


if __name__ == "__main__":
    main()
