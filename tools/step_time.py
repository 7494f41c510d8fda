"""Wall-clock time of the kinodynamics control step without profiling events: python tools/step_time.py [B] [iters] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import mpc_setup as S, oracle_lib as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
gm, rb, _, _ = S.make_product(B, max_iters=iters)
gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.]))
X = S.random_states(rb, B)
for _ in range(3):
    gm.iterate(X); X = gm.xs[:, 1, :].copy()
t0 = time.time()
for _ in range(steps):
    gm.iterate(X)
dt = (time.time() - t0) / steps
print('STREAMS=%s STAGGER=%s B=%d k=%d: %.2f ms/step (host copies included) %.0f steps/s' % (os.environ.get('SMPC_STREAMS', '1'), os.environ.get('SMPC_STAGGER', '-'), B, iters, dt * 1e3, B / dt), bool(np.isfinite(gm.xs).all()))
