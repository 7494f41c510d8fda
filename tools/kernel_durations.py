#!/usr/bin/env python3
"""Per-kernel launch durations of the FULL-BATCH launches only, from a rocprofv3 --kernel-trace CSV (one row per dispatch):

    python tools/kernel_durations.py <kernel_trace.csv> <out.json> [bench.json]

The raw `--stats` table of a bench run averages every launch of a kernel symbol, including the B = 1 latency workload, the k = 1
control stack and the cold solves, so its means say nothing about the launches the roofline is quoted on.  This summary keeps, per
kernel key of tools/kernel_keys.py, the dispatches whose grid is a full-batch launch of that workload: count, mean / min / max ms.
With a bench line (the JSON `bench.py` printed in the same command) it also checks that the line's `avg_launch_ms` (HIP events
inside bench.py) agrees with the profiler's mean to +-3 % (+-8 % for launches under 0.5 ms: event resolution) and exits non-zero otherwise."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_keys import kernel_key


def main():
    trace, out = sys.argv[1], sys.argv[2]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        key = kernel_key(r["Kernel_Name"], r["Grid_Size"] if "Grid_Size" in r else int(r["Grid_Size_X"]))
        if key:
            acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    res = {k: {"launches": len(v), "mean_ms": sum(v) / len(v), "min_ms": min(v), "max_ms": max(v)} for k, v in sorted(acc.items())}
    doc = {"source": "rocprofv3 --kernel-trace, one row per dispatch; full-batch launches only (tools/kernel_keys.py)",
           "git_commit": os.environ.get("SMPC_PROFILE_COMMIT", ""), "kernels": res}
    status = 0
    if len(sys.argv) > 3:
        # (the full object of the run: bench.py prints it to stderr as "bench.py full line: {...}"; its last stdout line is the compact one)
        lines = open(sys.argv[3]).read().splitlines()
        full = [ln[len("bench.py full line: "):] for ln in lines if ln.startswith("bench.py full line: ")]
        line = json.loads(full[-1] if full else [ln for ln in lines if ln.startswith('{"metric"')][-1])
        checks = {}
        rl = dict(line.get("roofline_other", {}))
        if "roofline" in line and "avg_launch_ms" in line["roofline"]:
            rl["dominant"] = line["roofline"]
        names = {"deriv": "deriv_body", "riccati": "riccati_kino_body", "forward": "forward_kino_body"}
        for k, e in list(rl.items()):
            if "avg_launch_ms_parts" in e: # a pass made of several kernels: every part against its own mean
                for kern, a in e["avg_launch_ms_parts"].items():
                    if kern in res and a > 0:
                        b = res[kern]["mean_ms"]
                        ok = abs(a - b) <= (0.03 if b >= 0.5 else 0.08) * b
                        checks[kern] = {"bench_avg_launch_ms": a, "profiler_mean_ms": b, "agree": ok}
                        status = status if ok else 1
                continue
            kern = next((names[n] for n in names if n in e.get("kernel", "")), None)
            if kern is None or kern not in res:
                continue
            a, b = e["avg_launch_ms"], res[kern]["mean_ms"]
            tol = 0.03 if b >= 0.5 else 0.08
            ok = abs(a - b) <= tol * b
            checks[kern] = {"bench_avg_launch_ms": a, "profiler_mean_ms": b, "agree": ok}
            if not ok:
                status = 1
        doc["agreement_with_bench_line"] = checks
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-28s %4d launches  mean %8.3f ms  min %8.3f  max %8.3f" % (k, v["launches"], v["mean_ms"], v["min_ms"], v["max_ms"]))
    if len(sys.argv) > 3:
        print("agreement with the bench line:", doc["agreement_with_bench_line"])
    sys.exit(status)


if __name__ == "__main__":
    main()
