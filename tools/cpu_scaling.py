"""CPU oracle scaling on the host cores (diagnostic for the cpu_baseline figure): python tools/cpu_scaling.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import mpc_setup as S, oracle_lib as O
thr = O.lib().orc_num_threads()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2 * thr
om, rb, _ = S.make_oracle(B, max_iters=3)
om.generateCycleHorizon(O.trot_cycle()); om.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
X = S.random_states(rb, B)
om.iterate(X); X = om.xs[:, 1, :].copy()
t0 = time.time(); n = 0
while time.time() - t0 < 6:
    om.iterate(X); X = om.xs[:, 1, :].copy(); n += 1
dt = time.time() - t0
print("threads %d B %d: %.0f control-steps/s (%.1f ms per instance-step per thread)" % (thr, B, B * n / dt, 1e3 * dt / n / (B / thr)))
