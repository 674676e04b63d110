#!/usr/bin/env python3
"""Copies the summaries of tools/profile_round.sh (gpurun_out/prof_<tag>/) into profiles/ and prints the table of
profiles/README.md.   usage: tools/collect_profiles.py r2"""
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
d = f"gpurun_out/prof_{tag}"
lines = []
for f in sorted(glob.glob(d + "/bench_*.json")):
    t = open(f).read().strip().splitlines()
    if t:
        lines.append(t[-1])
open(f"profiles/{tag}_bench_workloads.jsonl", "w").write("\n".join(lines) + "\n")
with open(f"profiles/{tag}_kernel_stats.csv", "w") as out:
    out.write('"workload","Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n')
    for f in sorted(glob.glob(d + "/kernel_stats_*.csv")):
        wl = os.path.basename(f)[len("kernel_stats_"):-4]
        for row in open(f):
            if any(k in row for k in ("k_fused", "k_smooth", "k_prepare", "k_finalize", "k_assemble")):
                out.write(f'"{wl}",' + row)
shutil.copy(d + "/counters.json", f"profiles/{tag}_counters.json")
shutil.copy(d + "/pmc_summary_2160p-Y8.csv", f"profiles/{tag}_pmc_summary.csv")
c = json.load(open(d + "/counters.json"))
print("| workload | frames/launch | frames/s | GB/s (algorithmic) | of 8 TB/s | HBM bytes / algorithmic (PMC) |")
print("|---|---|---|---|---|---|")
for l in lines:
    j = json.loads(l)
    wl = j["config"]["workload"].split()[0]
    r = j["roofline"]
    ratio = c.get(wl, {}).get("hbm_bytes_per_launch", 0) / r["algorithmic_bytes_per_launch"]
    print(f"| `{wl}` | {j['config']['frames_per_step_per_gpu']} | {j['frames_per_s']:,.0f} | {r['achieved']:,.0f} | {100 * r['frac']:.1f} % | {ratio:.3f} |")
