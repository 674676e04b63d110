#!/bin/bash
# Bench line + rocprofv3 kernel stats for the secondary workloads (run on the GPU box, from the repo root):
#   tools/profile_workloads.sh r1  ->  gpurun_out/prof_r1_workloads/{bench.jsonl,kernel_stats.csv}
set -e -o pipefail
tag=${1:-r1}
out=gpurun_out/prof_${tag}_workloads
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
: > "$out/bench.jsonl"
echo '"workload","Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"' > "$out/kernel_stats.csv"
for wl in 1080p-Y8 4320p-Y8 2160p-YUV420P8 2160p-YUV420P8-isolated 2160p-Y16 2160p-YUV420P16 2160p-YUV420P16-dh 2160p-Y32 \
          2160p-YUV444PS-dh 480p-YUV420P8-fresh 2160p-turned-Y8-fresh; do
    python3 bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 3 2> /dev/null | tail -1 >> "$out/bench.jsonl"
    rm -rf "$out/stats"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --workload $wl --no-cpu-baseline --steps 5 --warmup 2 > "$out/stats.log" 2>&1
    f=$(find "$out/stats" -name '*kernel_stats.csv' | head -1)
    grep -E 'k_fused|k_smooth|k_prepare|k_finalize|k_assemble' "$f" | sed "s/^/\"$wl\",/" >> "$out/kernel_stats.csv" || true
    echo "$wl done"
done
rm -rf "$out/stats"
echo "done: $out"
