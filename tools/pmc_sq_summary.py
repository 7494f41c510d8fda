"""Aggregate rocprofv3 --pmc SQ counter CSVs (one or more passes) into a per-kernel summary (average per full-batch launch):
python tools/pmc_sq_summary.py out.json pass1_counter_collection.csv [pass2_counter_collection.csv ...]"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_keys import KEYS, kernel_key

res = {k: collections.defaultdict(list) for k in KEYS}
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        key = kernel_key(r["Kernel_Name"], r["Grid_Size"])
        if key:
            res[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in res.items():
    if not d:
        continue
    o = {c: sum(v) / len(v) for c, v in d.items()}
    o["launches"] = max(len(v) for v in d.values())
    wc = o.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"):
            if c in o:
                o[c + "/SQ_WAVE_CYCLES"] = o[c] / wc
    if o.get("SQ_BUSY_CYCLES") and "SQ_VALU_MFMA_BUSY_CYCLES" in o:
        o["SQ_VALU_MFMA_BUSY_CYCLES/SQ_BUSY_CYCLES"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / o["SQ_BUSY_CYCLES"]
    if o.get("SQ_WAVES"):
        for c in list(o):
            if c.startswith("SQ_INSTS_") and "/" not in c:
                o[c + "_per_wave"] = o[c] / o["SQ_WAVES"]
    out[k] = o
json.dump({"source": "rocprofv3 --kernel-trace --pmc <SQ counters> (own passes, no other trace domain) -- python3 bench.py --steps 2 --warmup 1 "
           "--no-cpu-baseline; averages over the full-batch launches (B = 4096) of the main kernel symbols; raw counter sums over all "
           "shader engines / XCDs, ratios as named", "kernels": out}, open(sys.argv[1], "w"), indent=1)
for k, o in out.items():
    print(k, {c: round(v, 4) for c, v in o.items() if "/" in c or "per_wave" in c})
