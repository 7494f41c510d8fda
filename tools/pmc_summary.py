"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (separate passes) into the per-kernel HBM traffic
summary bench.py reads: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_keys import kernel_key
res = collections.defaultdict(dict)
for name, f in (("FETCH_SIZE", sys.argv[1]), ("WRITE_SIZE", sys.argv[2])):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            key = kernel_key(r["Kernel_Name"], r["Grid_Size"])
            if key:
                acc[key].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k][name + "_KiB_avg_full_batch_launch"] = sum(v) / len(v)
        res[k][name + "_launches"] = len(v)
for k in res:
    f = res[k].get("FETCH_SIZE_KiB_avg_full_batch_launch", 0)
    w = res[k].get("WRITE_SIZE_KiB_avg_full_batch_launch", 0)
    res[k]["hbm_bytes_per_launch_corrected"] = (2 * f + w) * 1024
json.dump({
    "git_commit": os.environ.get("SMPC_PROFILE_COMMIT", ""),
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 "
              "--no-cpu-baseline; full-batch launches only (B = 4096; Talos B = 1024, H = 100), see tools/kernel_keys.py",
    "correction": "FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half the bytes of wide coalesced reads); "
                  "WRITE_SIZE as reported (matches the algorithmic knot write of deriv_body to 1%)",
    "kernels": res}, open(sys.argv[3], "w"), indent=1)
for k in res:
    print(k, "%.2f GB/launch" % (res[k]["hbm_bytes_per_launch_corrected"] / 1e9))
