"""Quick timing of the Talos centroidal OCP with 6-D feet (bench.py's talos_flat_feet_line without the CPU leg):
   [SMPC_LIB_PATH=<variant .so>] python tools/quick_bench_talos_cent6.py [batch] [iters] [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))
import bench_side as bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
o = bench.talos_flat_feet_line("talos_centroidal", B, iters, steps, 5, 0, with_cpu=False)
print(json.dumps({k: o[k] for k in ("value", "ms_per_step", "kernel_ms")}))
