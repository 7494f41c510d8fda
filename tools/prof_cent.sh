#!/bin/bash
# rocprofv3 passes of the centroidal control step alone (cfg 2): gpurun -- 'bash tools/prof_cent.sh <tag> <commit>'
#   -> gpurun_out/<tag>_cent_kernel_stats.csv, <tag>_cent_pmc_hbm_traffic.json, <tag>_cent_pmc_sq.json
# (kernel trace and every PMC group in a run of its own, no other trace domain: MI355X_MICROARCH.md, HBM section)
set -u
TAG=${1:-rXX}
export SMPC_PROFILE_COMMIT=${2:-}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out
mkdir -p "$OUT/prof_$TAG"
export TMPDIR=/tmp
run_prof() {
  local name=$1
  shift
  (cd /tmp && rocprofv3 "$@" --output-format csv -d "$OUT/prof_$TAG/$name" -o "$name" -- python3 "$ROOT/tools/quick_bench_cent.py" 4096 3 6 > "$OUT/prof_$TAG/$name.log" 2>&1) || echo "$name failed"
}
run_prof cstats --kernel-trace --stats
run_prof cfetch --kernel-trace --pmc FETCH_SIZE
run_prof cwrite --kernel-trace --pmc WRITE_SIZE
run_prof csq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_WAIT_ANY SQ_WAIT_INST_ANY
run_prof csq2 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64
f() { find "$OUT/prof_$TAG/$1" -name "*$2" | head -1; }
cp "$(f cstats kernel_stats.csv)" "$OUT/${TAG}_cent_kernel_stats.csv" || echo "no kernel stats"
python3 tools/pmc_summary.py "$(f cfetch counter_collection.csv)" "$(f cwrite counter_collection.csv)" "$OUT/${TAG}_cent_pmc_hbm_traffic.json"
python3 tools/pmc_sq_summary.py "$OUT/${TAG}_cent_pmc_sq.json" "$(f csq1 counter_collection.csv)" "$(f csq2 counter_collection.csv)"
grep -v amdgpu "$OUT/prof_$TAG/cstats.log" | tail -2
head -8 "$OUT/${TAG}_cent_kernel_stats.csv" | cut -c1-40,140-260
find "$OUT/prof_$TAG" -name "*.csv" -size +8M -delete
