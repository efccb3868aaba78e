#!/bin/bash
# tools/collect_profiles.sh <tag> [<tag> ...]: copy what tools/profile_bench.sh left under gpurun_out/prof_<tag>/ into the
# tracked profiles/ directory: <tag>_summary.json, <tag>_kernel_stats.csv (the --stats table of that very pass) and, for full
# passes, latest_pmc_<config>.json.
cd "$(dirname "$0")/.."
for tag in "$@"; do
  d=gpurun_out/prof_$tag
  [ -f $d/summary.json ] || { echo "no $d/summary.json"; continue; }
  cp $d/summary.json profiles/${tag}_summary.json
  [ -f $d/kernel_stats.csv ] && cp $d/kernel_stats.csv profiles/${tag}_kernel_stats.csv
  for f in $d/latest_pmc_*.json; do [ -f "$f" ] && cp $f profiles/; done
  echo "collected $tag"
done
