#!/bin/bash
# rocprofv3 evidence for one bench.py workload: kernel stats + PMC passes (separate runs, --kernel-trace only).
#   tools/profile_r02.sh <tag> <outdir> <bench.py args...>
# Summaries: <outdir>/<tag>_kernel_stats.csv and <outdir>/<tag>_pmc_summary.csv (tools/pmc_summary.py)
set -u
tag=$1; out=$2; shift 2
export TMPDIR=/tmp
mkdir -p "$out"
run() {  # name, rocprof args...
  local name=$1; shift
  rm -rf "$out/raw_${tag}_$name"
  timeout -k 10 300 rocprofv3 --kernel-trace "$@" --output-format csv -d "$out/raw_${tag}_$name" -o p -- python3 bench.py "${BENCH_ARGS[@]}" \
    > "$out/${tag}_$name.log" 2>&1
  echo "$tag $name rc $?"
}
BENCH_ARGS=("$@" --steps 10 --no-cpu-baseline)
run stats --stats
BENCH_ARGS=("$@" --steps 5 --warmup 1 --no-cpu-baseline)
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run tcc --pmc TCC_HIT_sum TCC_MISS_sum
run sqa --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
run sqb --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
run grbm --pmc GRBM_GUI_ACTIVE
f=$(find "$out/raw_${tag}_stats" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv"
python3 tools/pmc_summary.py $(for n in fetch write tcc sqa sqb grbm; do find "$out/raw_${tag}_$n" -name "*counter_collection.csv" -printf "%h\n" | head -1; done) > "$out/${tag}_pmc_summary.csv"
grep '"metric"' "$out/${tag}_stats.log" | tail -1 > "$out/${tag}_bench_under_rocprof.json"
echo "$tag done"
