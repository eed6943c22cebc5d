#!/bin/bash
# rocprofv3 evidence for one python program of this repo: kernel stats + PMC passes (separate runs, --kernel-trace only, the
# program directly behind `--`).   tools/profile.sh <tag> <outdir> <script.py> <args...>      (run from the repo root)
# Summaries: <outdir>/<tag>_kernel_stats.csv and <outdir>/<tag>_pmc_summary.csv (tools/pmc_summary.py).  FULL=1 adds the SQ / GRBM passes.
set -u
tag=$1; out=$2; shift 2
export TMPDIR=/tmp
mkdir -p "$out"
run() {  # name, rocprof args...
  local name=$1; shift
  rm -rf "$out/raw_${tag}_$name"
  timeout -k 10 420 rocprofv3 --kernel-trace "$@" --output-format csv -d "$out/raw_${tag}_$name" -o p -- python3 "${PROG[@]}" \
    > "$out/${tag}_$name.log" 2>&1
  echo "$tag $name rc $?"
}
PROG=("$@")
run stats --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run tcc --pmc TCC_HIT_sum TCC_MISS_sum
passes="fetch write tcc"
if [ "${FULL:-0}" = "1" ]; then
  run sqa --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
  run sqb --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
  run grbm --pmc GRBM_GUI_ACTIVE
  passes="$passes sqa sqb grbm"
fi
f=$(find "$out/raw_${tag}_stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv"
f=$(find "$out/raw_${tag}_stats" -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python3 tools/kernel_gaps.py "$f" 100000 > "$out/${tag}_kernel_gaps.txt"
python3 tools/pmc_summary.py $(for n in $passes; do find "$out/raw_${tag}_$n" -name "*counter_collection.csv" -printf "%h\n" | head -1; done) > "$out/${tag}_pmc_summary.csv"
grep -h '"metric"\|"name"' "$out/${tag}_stats.log" | tail -8 > "$out/${tag}_under_rocprof.jsonl"
rm -rf "$out"/raw_${tag}_*
echo "$tag done"
