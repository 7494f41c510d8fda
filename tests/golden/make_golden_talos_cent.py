#!/usr/bin/env python3
"""Golden vectors of the centroidal OCP with 6-D feet (tests/golden/talos_cent_golden.npz), produced by the CPU oracle in the build container
(the reference cannot be built or imported here):

    python tests/golden/make_golden_talos_cent.py

(1) the stage model at seeded points (dynamics with contact torques, costs, wrench-cone rows, all derivatives), (2) closed loops of the batched
MPC on the Talos-class robot: H = 20 with the short walking cycle, k = 2, 8 control steps; the soles of record and small soles that activate
the wrench-cone rows.  Inputs and expected outputs only, no code."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mpc_setup as S  # noqa: E402
import oracle_lib as O  # noqa: E402

SHORT = dict(horizon=20, cycle=O.walk_cycle(5, 20), mpc_override=dict(T_fly=20, T_contact=5))
TIGHT = dict(mu=0.5, Lfoot=0.01, Wfoot=0.01)


def main():
    rng = np.random.default_rng(2026)
    out = {}
    rb = O.Robot("talos_like")
    cent = O.Cent(rb, O.talos_centroidal_settings(rb))
    masks = [3, 1, 2]
    xs = np.stack([np.concatenate([rng.normal(size=3) * 0.1 + [0, 0, 0.9], rng.normal(size=3) * 5, rng.normal(size=3)]) for _ in masks])
    us = np.stack([np.concatenate([np.concatenate([rng.normal(size=3) * 30 + [0, 0, 400], rng.normal(size=3) * 5]) for _ in range(2)]) for _ in masks])
    u_ref = np.array([0, 0, rb.mass * 9.81 / 2, 0, 0, 0] * 2)
    x_tgt = rng.normal(size=9) * 0.1
    pos = rb.centroidal(rb.x_ref)["feet"] + rng.normal(size=(2, 3)) * 0.02
    out.update(stage_x=xs, stage_u=us, stage_mask=np.array(masks), stage_u_ref=u_ref, stage_x_tgt=x_tgt, stage_pos=pos)
    for i, m in enumerate(masks):
        e = cent.eval(m, u_ref, x_tgt, pos, xs[i], us[i])
        d = cent.deriv(m, u_ref, x_tgt, pos, xs[i], us[i])
        out["stage%d_xnext" % i], out["stage%d_cost" % i], out["stage%d_c" % i] = e["xnext"], e["cost"], e["c"]
        for k in ("A", "B", "lx", "lu", "Lxx", "Lxu", "Luu", "Cx", "Cu"):
            out["stage%d_%s" % (i, k)] = d[k]
    for tag, over in (("loop", None), ("cone", TIGHT)):
        s = O.talos_centroidal_settings(rb)
        if over:
            s.update(over)
        ms = O.talos_mpc_settings(rb, max_iters=2)
        ms["T"] = SHORT["horizon"]
        ms.update(SHORT["mpc_override"])
        om = O.OracleCentMPC(O.Cent(rb, s), ms, 2)
        om.generateCycleHorizon(SHORT["cycle"])
        om.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
        X = S.talos_random_states(rb, 2, scale=0.7)
        out[tag + "_cold_xs"] = om.xs[0]
        Xs = []
        for _ in range(8):
            Xs.append(X.copy())
            om.iterate(X)
            X = S.talos_random_states(rb, 2, seed=len(Xs), scale=0.3)  # the measured multibody states of the loop (the OCP state is centroidal)
        out[tag + "_X"] = np.array(Xs)
        out[tag + "_xs"], out[tag + "_us"], out[tag + "_K0"], out[tag + "_alpha"], out[tag + "_vs"] = om.xs, om.us, om.K0, om.info[:, 2], om.vs
    np.savez_compressed(os.path.join(HERE, "talos_cent_golden.npz"), **out)
    print("talos centroidal: wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
