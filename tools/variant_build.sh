#!/bin/bash
# Experiment builds of the HIP library: tools/variant_build.sh <name> [extra hipcc flags ...]
#   -> simple-mpc_amd/csrc/variants/libsmpc_hip_<name>.so, the Go2 kinodynamics engine only (-DSMPC_KINO_ONLY: a fifth of the compile time);
#   run with SMPC_LIB_PATH=simple-mpc_amd/csrc/variants/libsmpc_hip_<name>.so python tools/quick_bench.py.  Never the shipped library.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p simple-mpc_amd/csrc/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -pragma-unroll-threshold=1000000 -DSMPC_KINO_ONLY "$@" \
  -Isimple-mpc_amd/csrc -Iinclude -x hip simple-mpc_amd/csrc/smpc_capi.cpp -o simple-mpc_amd/csrc/variants/libsmpc_hip_$name.so 2>&1 | grep -E "error" || true
ls -la simple-mpc_amd/csrc/variants/libsmpc_hip_$name.so
