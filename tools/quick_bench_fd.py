"""Time the constrained forward dynamics kernel (full_fd_body) on n states: python tools/quick_bench_fd.py [n]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import mpc_setup as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
gm, rb, _, _ = S.make_product(2)
X = np.tile(S.random_states(rb, 512, seed=3), (n // 512, 1))
tau = np.zeros((n, 12))
for mask in (15, 6, 0):
    m = np.full(n, mask, np.uint32)
    for _ in range(3):
        o = gm.constraintDynamics(X, tau, m)
    print("mask %2d  n=%d  kernel %.3f ms  -> %.1f M states/s  proximal iterations %d..%d"
          % (mask, n, o["kernel_ms"], n / o["kernel_ms"] / 1e3, o["iters"].min(), o["iters"].max()))
