#!/bin/bash
# Compiler-flag experiments on the GPU box: rebuild libsmpc_hip.so with extra hipcc flags and time one configuration.
#   gpurun -- 'bash tools/flag_sweep.sh'
set -e
cd "$(dirname "$0")/.."
for flags in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-O2" "-mllvm -amdgpu-use-divergent-register-indexing"; do
  if SMPC_HIPCC_FLAGS="$flags" python -c "import __graft_entry__ as g; g.build_hip(force=True)" >/dev/null 2>gpurun_out/flag_err.log; then
    echo "flags [$flags]: $(python tools/quick_bench.py 4096 3 4 2>&1 | tail -1 | cut -c1-170)"
  else
    echo "flags [$flags]: build failed: $(tail -1 gpurun_out/flag_err.log | cut -c1-150)"
  fi
done
python -c "import __graft_entry__ as g; g.build_hip(force=True)" >/dev/null 2>&1
