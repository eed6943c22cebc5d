#!/bin/bash
# rocprofv3 kernel stats of the device-resident CG loops (fs_cg, fs_cg2) on config 2's pattern: gpurun -- bash tools/profile_cg.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/cgprof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cgprof/raw -o cg -- python3 $R/tools/microbench.py --what cg --out $R/gpurun_out/cgprof/mb.jsonl > $R/gpurun_out/cgprof/run.log 2>&1
find $R/gpurun_out/cgprof/raw -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/cgprof/kernel_stats.csv \;
rm -rf $R/gpurun_out/cgprof/raw
head -25 $R/gpurun_out/cgprof/kernel_stats.csv | cut -c1-220
cat $R/gpurun_out/cgprof/mb.jsonl
