#!/bin/bash
# Experiment builds of the HIP library: tools/variant_build.sh <name> [extra hipcc flags ...]
#   -> simple-mpc_amd/csrc/variants/libsmpc_hip_<name>.so, the Go2 kinodynamics engine only (-DSMPC_KINO_ONLY: a fifth of the compile time;
#      add -DSMPC_CENT_ONLY for the Go2 centroidal engine INSTEAD of the kinodynamics one);
#   run with SMPC_LIB_PATH=simple-mpc_amd/csrc/variants/libsmpc_hip_<name>.so python tools/quick_bench.py.  Never the shipped library.
set -e -o pipefail
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p simple-mpc_amd/csrc/variants
out=simple-mpc_amd/csrc/variants/libsmpc_hip_$name.so
rm -f "$out" # (a failed compile must not leave an earlier build of the same name behind)
log=$(mktemp)
if ! hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -pragma-unroll-threshold=1000000 -DSMPC_KINO_ONLY "$@" \
  -Isimple-mpc_amd/csrc -Iinclude -x hip simple-mpc_amd/csrc/smpc_capi.cpp -o "$out" > "$log" 2>&1; then
  grep -E "error|Error" "$log" | head -40
  echo "variant_build: hipcc FAILED (full log: $log)" >&2
  exit 1
fi
grep -E "warning: .*(spill|scratch)" "$log" | head -5 || true
rm -f "$log"
ls -la "$out"
