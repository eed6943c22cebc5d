#!/bin/bash
# Round 4's rocprofv3 evidence: per workload one --kernel-trace --stats run and three --pmc passes (tools/profile.sh), summaries
# into gpurun_out/r4p/ (copied to profiles/r04_* afterwards).   Run from the repo root on the GPU box.
set -u
out=gpurun_out/r4p
mkdir -p $out
for w in c2 c3 c5; do
  bash tools/profile.sh r04_$w $out bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-reproducible-cost
done
FS_REPRODUCIBLE=1 bash tools/profile.sh r04_c3_fixed_order $out bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-reproducible-cost
FS_REPRODUCIBLE=1 bash tools/profile.sh r04_c5_fixed_order $out bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline --no-reproducible-cost
ls -la $out
