"""Long closed loop of the headline workload on the GPU: python tools/long_run.py [B] [steps].  Prints, every 100 control steps, whether every
scalar is finite, the status words, the worst primal / dual residual and the share of full steps; at the end the drift of the mean base height."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import mpc_setup as S, oracle_lib as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
gm, rb, _, _ = S.make_product(B, max_iters=3)
gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.]))
X = S.random_states(rb, B)
rng = np.random.default_rng(1)
z0 = None
t0 = time.time()
for k in range(steps):
    gm.iterate(X)
    X = gm.xs[:, 1, :] + rng.normal(0.0, 1e-3, X.shape)
    X[:, 3:7] /= np.linalg.norm(X[:, 3:7], axis=1, keepdims=True)
    if z0 is None:
        z0 = X[:, 2].mean()
    if (k + 1) % 100 == 0:
        info = gm.info
        print(k + 1, "finite", bool(np.isfinite(info).all() and np.isfinite(gm.xs).all()), "status!=0:", int((gm.status != 0).sum()),
              "prim max %.1e dual max %.1e" % (info[:, 4].max(), info[:, 5].max()), "full steps %.3f" % (info[:, 2] == 1.0).mean(),
              "x advance %.2f m  z %.3f" % (X[:, 0].mean(), X[:, 2].mean()), flush=True)
print("done: %.1f s, base height drift %.2e m" % (time.time() - t0, X[:, 2].mean() - z0))
