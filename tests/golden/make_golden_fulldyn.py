#!/usr/bin/env python3
"""Golden vectors of the full-dynamics path (tests/golden/go2_full_golden.npz, talos_full_golden.npz, go2_full_cone_golden.npz), produced by the CPU
oracle in the build container (the reference cannot be built or imported here):

    python tests/golden/make_golden_fulldyn.py

(1) constrained forward dynamics + derivatives and the stage model at seeded points, (2) closed loops of the batched MPC:
Go2 (3-D contacts, H = 50, k = 1 and 3, 8 control steps) and Talos (6-D contacts + wrench cones, H = 20 with the short walking
cycle so that the CPU tier can replay it, k = 2, 6 control steps; default soles and the small slippery soles that activate the
cone rows).  Inputs and expected outputs only, no code."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mpc_setup as S  # noqa: E402
import oracle_lib as O  # noqa: E402

TALOS_SHORT = dict(horizon=20, cycle=O.walk_cycle(5, 20), mpc_override=dict(T_fly=20, T_contact=5))
TIGHT = dict(mu=0.3, Lfoot=0.05, Wfoot=0.04)


def stage_vectors(out, tag, rb, full, fs, masks, rng, sigma_scale):
    nu, nf = full.nu, rb.nf
    xs = S.talos_random_states(rb, len(masks), seed=int(rng.integers(1 << 30)), scale=sigma_scale) if fs == 6 else S.random_states(rb, len(masks), seed=int(rng.integers(1 << 30)))
    us = rng.normal(size=(len(masks), nu)) * (10 if fs == 6 else 3)
    fz = np.zeros(fs)
    fz[2] = rb.mass * 9.81 / nf
    u_ref = np.concatenate([np.zeros(nu), np.tile(fz, nf)])
    feet = rb.centroidal(rb.x_ref)["feet"] + rng.normal(size=(nf, 3)) * 0.02
    out.update({tag + "_x": xs, tag + "_u": us, tag + "_mask": np.array(masks), tag + "_u_ref": u_ref, tag + "_foot_ref": feet})
    for i, m in enumerate(masks):
        e = full.eval(m, u_ref, rb.x_ref, feet, xs[i], us[i])
        d = full.deriv(m, u_ref, rb.x_ref, feet, xs[i], us[i])
        f = rb.full_forward_dynamics(xs[i], us[i], m, full.s["Kp_correction"], full.s["Kd_correction"], fs=fs)
        out["%s%d_xnext" % (tag, i)], out["%s%d_cost" % (tag, i)], out["%s%d_c" % (tag, i)] = e["xnext"], e["cost"], e["c"]
        out["%s%d_a" % (tag, i)], out["%s%d_lam" % (tag, i)] = f["a"], f["lam"]
        for k in ("A", "B", "lx", "lu", "Lxx", "Lxu", "Luu", "Cx", "Cu"):
            out["%s%d_%s" % (tag, i, k)] = d[k]


def loop(out, tag, om, rb, X, steps):
    out[tag + "_X0"] = X
    out[tag + "_cold_xs"] = om.xs[0]
    for _ in range(steps):
        om.iterate(X)
        X = om.xs[:, 1, :].copy()
    out[tag + "_xs"], out[tag + "_us"], out[tag + "_K0"], out[tag + "_alpha"] = om.xs, om.us, om.K0, om.info[:, 2]


def main():
    rng = np.random.default_rng(2024)
    out = {}
    rb = O.Robot("go2_like")
    full = O.Full(rb, O.go2_full_settings(rb))
    stage_vectors(out, "stage", rb, full, 3, [15, 6, 9], rng, 1.0)
    for k in (1, 3):
        om, _ = S.make_full_oracle(2, max_iters=k)
        loop(out, "loop%d" % k, om, rb, S.random_states(rb, 2), 8)
    np.savez_compressed(os.path.join(HERE, "go2_full_golden.npz"), **out)
    print("go2: wrote", len(out), "arrays")

    out = {}
    tb = O.Robot("talos_like")
    tfull = O.Full(tb, O.talos_full_settings(tb))
    stage_vectors(out, "stage", tb, tfull, 6, [3, 1, 2], rng, 0.5)
    for tag, over, walk in (("loop", None, (0.1, 0, 0, 0, 0, 0)), ("cone", TIGHT, (0.2, 0.1, 0, 0, 0, 0.2))):
        s = O.talos_full_settings(tb)
        if over:
            s.update(over)
        ms = O.talos_mpc_settings(tb, max_iters=2)
        ms["T"] = TALOS_SHORT["horizon"]
        ms.update(TALOS_SHORT["mpc_override"])
        om = O.OracleFullMPC(O.Full(tb, s), ms, 2)
        om.generateCycleHorizon(TALOS_SHORT["cycle"])
        om.switchToWalk(np.array(walk, float))
        loop(out, tag, om, tb, S.talos_random_states(tb, 2, scale=0.7), 6)
        out[tag + "_vs"] = om.vs
    np.savez_compressed(os.path.join(HERE, "talos_full_golden.npz"), **out)
    print("talos: wrote", len(out), "arrays")


def cone():
    """Go2 with force_cone (five friction-pyramid rows per foot in contact, round 3): H = 20, k = 2, 6 control steps, mu = 0.6"""
    out = {}
    rb = O.Robot("go2_like")
    om, _ = S.make_full_oracle(2, max_iters=2, horizon=20, walk=(0.3, 0.1, 0, 0, 0, 0.2), settings_override={"force_cone": True, "mu": 0.6})
    loop(out, "cone", om, rb, S.random_states(rb, 2), 6)
    out["cone_vs"] = om.vs
    np.savez_compressed(os.path.join(HERE, "go2_full_cone_golden.npz"), **out)
    print("go2 cone: wrote", len(out), "arrays; active cone rows", int((om.vs[:, :, 24:] != 0).sum()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cone":
        cone()
    else:
        main()
        cone()
