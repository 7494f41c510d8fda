"""bench.py's full-dynamics / Talos kinodynamics lines without the CPU leg, for SMPC_FULL_PARTS experiments:
   SMPC_FULL_PARTS=n python tools/quick_bench_parts.py [talos|go2|taloskino] [batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))
import bench_side as bench  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "talos"
if what == "taloskino":
    o = bench.talos_flat_feet_line("talos_kinodynamics", int(sys.argv[2]) if len(sys.argv) > 2 else 1024, 3, 8, 2, 0, with_cpu=False)
else:
    B = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if what == "talos" else 4096)
    o = bench.fulldynamics_line(B, 3, 10, 3, 0, with_cpu=False, robot=what)
print(os.environ.get("SMPC_FULL_PARTS", "default"), json.dumps({k: o[k] for k in ("value", "ms_per_step", "kernel_ms")}))
