#!/bin/bash
# Everything profiles/<tag>_* is made from, in one call on the GPU box:
#   gpurun --timeout 2400 -- "bash tools/profile_round.sh r03a $(git rev-parse --short HEAD)"
# Writes gpurun_out/<tag>_{bench,bench_full,bench_side,bench_centroidal}.json, <tag>_kernel_stats.csv (all workloads), <tag>_kernel_stats_headline.csv and
# <tag>_kernel_durations.json (the headline workload alone under the profiler: full-batch launches only, checked against the HIP-event
# averages of that same run's bench line), <tag>_kernel_durations_all.json (every workload's full-batch launches), <tag>_pmc_hbm_traffic.json, <tag>_pmc_sq.json, <tag>_fp64_peak.json;
# copy them into profiles/ afterwards.  The PMC passes are separate rocprofv3 runs with --kernel-trace only (no other trace
# domain), FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md, HBM section).
set -u
TAG=${1:-rXX}
export SMPC_PROFILE_COMMIT=${2:-} # (the GPU box has no .git: pass `git rev-parse --short HEAD` of the tree that was sent)
cd "$(dirname "$0")/.."
ROOT=$PWD # (the program is named by an absolute path: rocprofv3 runs from /tmp)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT/prof_$TAG"
export TMPDIR=/tmp
BENCH="bench.py --steps 2 --warmup 1 --no-cpu-baseline"

[ -x tools/micro/fp64_peak.bin ] && ./tools/micro/fp64_peak.bin 1.0 > "$OUT/${TAG}_fp64_peak.json"
python3 bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err" || echo "bench failed"
tail -1 "$OUT/${TAG}_bench.json" | cut -c1-300
cp bench_full.json "$OUT/${TAG}_bench_full.json" 2> /dev/null # (everything the run measured; the printed line is the compact object)
cp bench_side.json "$OUT/${TAG}_bench_side.json" 2> /dev/null
python3 bench.py --workload centroidal > "$OUT/${TAG}_bench_centroidal.json" 2>> "$OUT/${TAG}_bench.err" || echo "centroidal bench failed"

run_prof() { # name, rocprofv3 options...
  local name=$1
  shift
  (cd /tmp && rocprofv3 "$@" --output-format csv -d "$OUT/prof_$TAG/$name" -o "$name" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/prof_$TAG/$name.log" 2>&1) || echo "$name failed"
}
run_prof stats --kernel-trace --stats
# the headline workload alone (its kernel symbols are shared with the side workloads -- control stack at k = 1, standing robots --, whose
# launches would shift the per-kernel means the bench line's HIP-event averages are checked against)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG/headline" -o headline -- python3 "$ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --headline-only > "$OUT/prof_$TAG/headline.log" 2>&1) || echo "headline failed"
run_prof fetch --kernel-trace --pmc FETCH_SIZE
run_prof write --kernel-trace --pmc WRITE_SIZE
run_prof sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_WAIT_ANY SQ_WAIT_INST_ANY
run_prof sq2 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64
run_prof sq3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU_TRANS_F64

f() { find "$OUT/prof_$TAG/$1" -name "*$2" | head -1; }
cp "$(f stats kernel_stats.csv)" "$OUT/${TAG}_kernel_stats.csv" || echo "no kernel stats"
# durations of the full-batch launches only, from the per-dispatch trace of the same profiled command; must agree with the HIP-event
# averages of the profiled bench line itself (the headline run above is a different process: its averages are printed beside for the eye)
python3 tools/kernel_durations.py "$(f stats kernel_trace.csv)" "$OUT/${TAG}_kernel_durations_all.json" > /dev/null
python3 tools/kernel_durations.py "$(f headline kernel_trace.csv)" "$OUT/${TAG}_kernel_durations.json" "$OUT/prof_$TAG/headline.log" || echo "DURATION CHECK FAILED"
cp "$(f headline kernel_stats.csv)" "$OUT/${TAG}_kernel_stats_headline.csv" || echo "no headline kernel stats"
python3 tools/pmc_summary.py "$(f fetch counter_collection.csv)" "$(f write counter_collection.csv)" "$OUT/${TAG}_pmc_hbm_traffic.json"
python3 tools/pmc_sq_summary.py "$OUT/${TAG}_pmc_sq.json" "$(f sq1 counter_collection.csv)" "$(f sq2 counter_collection.csv)" "$(f sq3 counter_collection.csv)"
head -6 "$OUT/${TAG}_kernel_stats.csv" | cut -c1-60,150-260
# keep the merge-back small
find "$OUT/prof_$TAG" -name "*.csv" -size +8M -delete
