#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs per dispatch and counter (all XCDs / dimensions) for the
fused sweep kernels.  usage: pmc_summary.py <dir> [<dir> ...]  ->  CSV on stdout."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = csv.writer(sys.stdout, quoting=csv.QUOTE_MINIMAL, lineterminator="\n")
out.writerow(["pass", "dispatch_id", "kernel", "grid_size", "counter", "sum_over_dims"])
for d in sys.argv[1:]:
    if not os.path.isdir(d):
        continue
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(float)
        meta = {}
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            if "k_fused" not in name and "k_smooth" not in name:
                continue
            key = (int(row["Dispatch_Id"]), row["Counter_Name"])
            acc[key] += float(row["Counter_Value"])
            meta[key] = (name.split("(")[0][-48:], row.get("Grid_Size", ""))
        for (disp, ctr), v in sorted(acc.items()):
            out.writerow([os.path.basename(d), disp, meta[(disp, ctr)][0], meta[(disp, ctr)][1], ctr, v])
