"""Quick timing of the flat-foot inverse-dynamics QP (bench.py's inverse_dynamics_quad_line without the CPU leg):
   [SMPC_LIB_PATH=<variant .so>] python tools/quick_bench_id6.py [batch]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))
import bench_side as bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for _ in range(2):
    o = bench.inverse_dynamics_quad_line(B, 0, with_cpu=False)
    print(json.dumps({k: o[k] for k in ("value", "ms_per_call", "max_residual")}))
