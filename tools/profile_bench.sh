#!/bin/bash
# Run bench.py under rocprofv3 (kernel trace + stats) on the GPU box and leave a compact per-kernel
# summary in gpurun_out/<tag>/kernel_stats_summary.csv (+ the bench JSON line measured in the same run).
#   usage (through gpurun):  bash tools/profile_bench.sh r1 [bench args...]
set -u
TAG=${1:-r1}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -o bench -- \
  python bench.py --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
STATS=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1)
if [ -n "$STATS" ]; then
  python tools/summarize_kernel_stats.py "$STATS" > "$OUT/kernel_stats_summary.csv"
  head -25 "$OUT/kernel_stats_summary.csv"
else
  echo "no kernel_stats.csv produced"; find "$OUT/raw" | head; tail -5 "$OUT/bench.err"
fi
rm -rf "$OUT/raw"
