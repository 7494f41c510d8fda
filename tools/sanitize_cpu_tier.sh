#!/bin/bash
# The kernel bodies (sequential-lane CPU build, tests/emu) under AddressSanitizer + UndefinedBehaviorSanitizer, then the whole CPU tier:
#   bash tools/sanitize_cpu_tier.sh            (≈ 10 min on 8 cores; prints the number of sanitizer reports, expected 0)
# GPU sanitizers are not available on the pool: the same bodies compiled for gfx950 share every index expression with this build.
set -u
cd "$(dirname "$0")/.."
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -fPIC -fopenmp -shared -Itests/emu -Isimple-mpc_amd/csrc -Iinclude \
  -Wno-unknown-pragmas simple-mpc_amd/csrc/smpc_capi.cpp -o tests/emu/libsmpc_emu.so || exit 1
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 \
  UBSAN_OPTIONS=print_stacktrace=1 OMP_NUM_THREADS=4 python -m pytest tests -q -m "not gpu" > /tmp/smpc_sanitize.log 2>&1
tail -3 /tmp/smpc_sanitize.log
grep -q " passed" /tmp/smpc_sanitize.log || echo "THE TEST RUN DID NOT FINISH (an AddressSanitizer abort kills the interpreter: rerun the test after the last dot with -s)"
echo "sanitizer reports: $(grep -c 'AddressSanitizer\|runtime error' /tmp/smpc_sanitize.log)"
rm -f tests/emu/libsmpc_emu.so # (the next test run rebuilds the plain library)
