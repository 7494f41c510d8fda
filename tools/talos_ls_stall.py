"""Line searches of the Talos full-dynamics closed loop (B = 1024, k = 3) that do NOT take the full step: |dphi0| against phi0 (is the
Armijo test of these instances decided by rounding?).  python tools/talos_ls_stall.py [kind=talos|talos_kinodynamics] [steps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench_common as bench
from simple_mpc import presets as P
kind = sys.argv[1] if len(sys.argv) > 1 else "talos"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
gm, mh = bench.make_mpc(kind, 1024, 3, 0, horizon=100)
X = P.random_states(mh, 1024, scale=0.7)
rng = np.random.default_rng(0)
for step in range(steps):
    gm.iterate(X)
    info = gm.info
    idx = info[:, 11].astype(int)
    bad = idx > 0
    rel = np.abs(info[:, 1]) / np.maximum(1.0, np.abs(info[:, 0]))
    q = lambda a: np.quantile(a, [0.0, 0.5, 0.9, 1.0]) if a.size else np.zeros(4)
    print(step, "not full step:", int(bad.sum()), "failed:", int(info[:, 6].sum()),
          "| |dphi0|/max(1,|phi0|) of those  min/med/p90/max: %.1e %.1e %.1e %.1e" % tuple(q(rel[bad])),
          "| of the others: %.1e %.1e %.1e %.1e" % tuple(q(rel[~bad])), "| phi0 med %.2e dual med %.1e" % (np.median(np.abs(info[:, 0])), np.median(info[:, 5])))
    X = gm.xs[:, 1, :] + rng.normal(0, 1e-3, (1024, gm.nx)); X[:, 3:7] /= np.linalg.norm(X[:, 3:7], axis=1, keepdims=True)
