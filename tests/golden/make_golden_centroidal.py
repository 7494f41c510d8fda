"""Golden vectors of the centroidal path (tests/golden/go2_cent_golden.npz), produced by the CPU oracle in the build
container:  python tests/golden/make_golden_centroidal.py
(1) stage model at seeded points, (2) 6 control steps of the batched MPC, k = 1 and k = 3, default scenario and a
scenario with active friction cones (mu = 0.1, lateral / yaw velocity command)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mpc_setup as S  # noqa: E402
import oracle_lib as O  # noqa: E402

HARD = dict(settings_override=dict(mu=0.1), walk=(0.8, 0.5, 0, 0, 0, 0.5))


def run(om, rb, X, steps=6):
    for _ in range(steps):
        om.iterate(X)
        X = np.stack([rb.integrate(X[b], np.r_[np.zeros(18), 0.02, 0.01, np.zeros(16)]) for b in range(X.shape[0])])
    return X


def main():
    out = {}
    rb = O.Robot("go2_like")
    Cn = O.Cent(rb, O.go2_centroidal_settings(rb))
    rng = np.random.default_rng(11)
    masks = [15, 6, 9]
    xs = rng.normal(size=(3, 9)) * np.array([0.1, 0.1, 0.1, 1, 1, 1, 0.3, 0.3, 0.3]) + np.array([0, 0, 0.3, 0, 0, 0, 0, 0, 0])
    us = rng.normal(size=(3, 12)) * 5.0 + np.tile([0, 0, 35.0], 4)
    pos = rng.normal(size=(4, 3)) * 0.2
    u_ref, x_tgt = np.tile([0, 0, 36.0], 4), rng.normal(size=9) * 0.1
    out.update(stage_x=xs, stage_u=us, stage_mask=np.array(masks), stage_pos=pos, stage_u_ref=u_ref, stage_x_tgt=x_tgt)
    for i, m in enumerate(masks):
        e = Cn.eval(m, u_ref, x_tgt, pos, xs[i], us[i])
        d = Cn.deriv(m, u_ref, x_tgt, pos, xs[i], us[i])
        out["stage%d_xnext" % i], out["stage%d_cost" % i], out["stage%d_c" % i] = e["xnext"], e["cost"], e["c"]
        for k in ("A", "B", "lx", "lu", "Lxx", "Lxu", "Luu", "Cu"):
            out["stage%d_%s" % (i, k)] = d[k]
    for name, kw, scale in (("easy", {}, 1.0), ("hard", HARD, 2.0)):
        for k in (1, 3):
            om, rb2, _ = S.make_cent_oracle(4, k, settings_override=kw.get("settings_override"))
            om.generateCycleHorizon(O.trot_cycle())
            om.switchToWalk(np.array(kw.get("walk", (0.2, 0, 0, 0, 0, 0)), float))
            X = S.random_states(rb2, 4, scale=scale)
            tag = "%s%d" % (name, k)
            out[tag + "_X0"] = X
            out[tag + "_cold_xs"] = om.xs[0]
            run(om, rb2, X)
            out[tag + "_xs"], out[tag + "_us"], out[tag + "_K0"], out[tag + "_vs"] = om.xs, om.us, om.K0, om.vs
            out[tag + "_alpha"] = om.info[:, 2]
    np.savez_compressed(os.path.join(HERE, "go2_cent_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
