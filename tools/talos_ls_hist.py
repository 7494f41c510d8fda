"""Line-search outcomes of the Talos full-dynamics closed loop (B = 1024, k = 3): per control step, the histogram of the accepted candidate
index (alpha = 2^-index) of the last iteration, the number of failed searches and the largest proximal regularisation.
Steady state: ~94 % of the instances accept alpha = 1, the rest spread evenly over the nine smaller candidates."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from simple_mpc import presets as P
gm, mh = bench.make_mpc("talos", 1024, 3, 0, horizon=100)
X = P.random_states(mh, 1024, scale=0.7)
rng = np.random.default_rng(0)
for step in range(12):
    gm.iterate(X)
    info = gm.info
    idx = info[:, 11].astype(int)
    print(step, "ls_index hist", np.bincount(idx, minlength=10), "failed", int(info[:, 6].sum()), "preg max %.1e" % info[:, 7].max())
    X = gm.xs[:, 1, :] + rng.normal(0, 1e-3, (1024, gm.nx)); X[:, 3:7] /= np.linalg.norm(X[:, 3:7], axis=1, keepdims=True)
