#!/bin/bash
# Round 5's rocprofv3 evidence: per workload one --kernel-trace --stats run and three --pmc passes (tools/profile.sh), summaries
# into gpurun_out/r5p/ (copied to profiles/r05_* afterwards; tools/refresh_traffic.py r05 then rewrites profiles/traffic_*.json).
# Run from the repo root on the GPU box.   bash tools/profile_r05.sh [workloads...]
set -u
out=gpurun_out/r5p
mkdir -p $out
for w in ${@:-c2 c3 c4 c5}; do
  bash tools/profile.sh r05_$w $out bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-reproducible-cost --lean
done
ls -la $out
