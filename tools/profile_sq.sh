#!/bin/bash
# Where do a kernel's wave cycles go?  SQ counters of one python program of this repo in separate rocprofv3 --pmc passes (kernel
# trace only, the program directly behind `--`).   tools/profile_sq.sh <tag> <outdir> <script.py> <args...>
set -u
tag=$1; out=$2; shift 2
export TMPDIR=/tmp
mkdir -p "$out"
PROG=("$@")
run() {
  local name=$1; shift
  rm -rf "$out/raw_${tag}_$name"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/raw_${tag}_$name" -o p -- python3 "${PROG[@]}" \
    > "$out/${tag}_$name.log" 2>&1
  echo "$tag $name rc $?"
}
run a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT
run c SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH
run d SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_CYCLES_VMEM_RD
run e SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64
python3 tools/pmc_summary.py $(for n in a b c d e; do find "$out/raw_${tag}_$n" -name "*counter_collection.csv" -printf "%h\n" | head -1; done) > "$out/${tag}_sq_summary.csv"
rm -rf "$out"/raw_${tag}_*
echo "$tag done"
