#!/usr/bin/env python3
"""Generate tests/golden/go2_kino_golden.npz with the CPU oracle (this container, seeded).

The reference cannot be built or imported here (Aligator / Pinocchio absent), and its own tests
hold no numerical fixture for this path, so these vectors pin the ORACLE (regression) and give the GPU box a
device-independent target: inputs and expected outputs only, no code.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mpc_setup as S  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    out = {}
    # (1) stage model: evaluation + derivatives at seeded points, three contact patterns
    rb = O.Robot("go2_like")
    K = O.Kino(rb, O.go2_kino_settings(rb))
    rng = np.random.default_rng(7)
    xs, us, masks = [], [], [15, 6, 9]
    for m in masks:
        x = rb.integrate(rb.x_ref, rng.normal(size=36) * S.SIGMA)
        u = np.concatenate([rng.normal(size=12) * 5 + np.tile([0, 0, 37.0], 4), rng.normal(size=12)])
        xs.append(x)
        us.append(u)
    out["stage_x"], out["stage_u"], out["stage_mask"] = np.array(xs), np.array(us), np.array(masks)
    u_ref = np.concatenate([np.tile([0, 0, rb.mass * 9.81 / 4], 4), np.zeros(12)])
    fr = rng.normal(size=(4, 3)) * 0.1
    out["stage_u_ref"], out["stage_foot_ref"] = u_ref, fr
    for i, m in enumerate(masks):
        e = K.eval(m, u_ref, rb.x_ref, fr, xs[i], us[i])
        d = K.deriv(m, u_ref, rb.x_ref, fr, xs[i], us[i])
        out["stage%d_xnext" % i], out["stage%d_cost" % i], out["stage%d_c" % i] = e["xnext"], e["cost"], e["c"]
        for k in ("A", "B", "lx", "lu", "Lxx", "Lxu", "Luu", "Cx"):
            out["stage%d_%s" % (i, k)] = d[k]
    # (2) closed loop: B = 4 instances, k = 1 and k = 3, 8 control steps, fed back with xs[1]
    for k in (1, 3):
        om, rb, _ = S.make_oracle(4, max_iters=k)
        om.generateCycleHorizon(O.trot_cycle())
        om.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
        X = S.random_states(rb, 4)
        out["loop%d_X0" % k] = X
        out["loop%d_cold_xs" % k] = om.xs[0]
        for _ in range(8):
            om.iterate(X)
            X = om.xs[:, 1, :].copy()
        out["loop%d_xs" % k], out["loop%d_us" % k], out["loop%d_K0" % k] = om.xs, om.us, om.K0
        out["loop%d_alpha" % k] = om.info[:, 2]
    np.savez_compressed(os.path.join(HERE, "go2_kino_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
