#!/bin/bash
# per-kernel averages of a short closed loop under rocprofv3 (kernel trace only):  gpurun -- 'bash tools/prof_quick.sh <tag> [B]'
# writes gpurun_out/<tag>/kernel_stats.csv and prints the top kernels
TAG=${1:-q}
B=${2:-4096}
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o q -- python3 "$ROOT/tools/quick_bench.py" $B 3 4 > "$OUT/prof.log" 2>&1)
F=$(find "$OUT/prof" -name "*kernel_stats.csv" | head -1)
cp "$F" "$OUT/kernel_stats.csv"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r["Name"]
    m = re.search(r"ZNS_(\d+)(\w+?_body)", n)
    short = (m.group(2) if m else n[:40]) + (" aux" if "ELi1EEE" in n else "")
    print("%-28s calls %6s  avg %10.1f us  total %8.2f ms  %5.1f %%" % (short, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
find "$OUT/prof" -name "*.csv" -size +4M -delete
