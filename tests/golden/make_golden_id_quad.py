#!/usr/bin/env python3
"""Golden vectors of the whole-body inverse-dynamics QPs of a robot with flat feet (tests/golden/talos_id_golden.npz; tsid Contact6d:
12 corner forces per foot, 6-D contact motion, 17 friction / force rows per foot), produced by the CPU oracle in the build container (TSID /
ProxQP cannot be built or imported here):

    python tests/golden/make_golden_id_quad.py

KinodynamicsID (contact motion as a cost and as an equality) and CentroidalID (right foot in the air, tracked) on the talos_like robot, 2 robots
each, 100 ADMM iterations per tick (fixed count), closed loops of 10 ticks: per tick the states that went in and the torques, accelerations and
contact wrenches that came out.  Inputs and expected outputs only, no code."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mpc_setup as S  # noqa: E402
import oracle_lib as O  # noqa: E402

DT, TICKS, B = 1e-3, 10, 2
KINO = dict(kp_base=1.0, kp_posture=1.0, kp_contact=10.0, w_base=1.0, w_posture=0.05, w_contact_motion=10.0, w_contact_force=1.0)  # reference tests, :192-204
CENT = dict(kp_base=7.0, kp_posture=10.0, kp_contact=10.0, w_base=100.0, w_posture=1.0, w_contact_motion=1.0, w_contact_force=1.0, kp_com=7.0,
            kp_feet_tracking=10.0, w_com=100.0, w_feet_tracking=100.0)


def step(rb, x, a, dt=DT):
    """the integration of the reference's tests (tests/inverse-dynamics/kinodynamics-id.cpp:52-56)"""
    q, v = x[: rb.nq], x[rb.nq:]
    xn = rb.integrate(np.concatenate([q, v]), np.concatenate([(v + a / 2 * dt) * dt, a * dt]))
    return np.concatenate([xn[: rb.nq], v + a * dt])


def cent_targets(rb):
    c = rb.centroidal(rb.x_ref)
    feet = c["feet"].copy()
    feet[1] += [0.03, 0.0, 0.05]
    w = np.zeros((2, 6))
    w[0, 2] = rb.mass * 9.81
    return [True, False], w, c["com"] + [0.0, 0.03, 0.0], feet


def run(rb, kind, solve):
    X = S.talos_random_states(rb, B, seed=21, scale=0.1)
    out = dict(X=[], tau=[], a=[], f=[])
    for _ in range(TICKS):
        tau, a, f = solve(X)
        for k, v in zip(("X", "tau", "a", "f"), (X, tau, a, f)):
            out[k].append(np.array(v, float).copy())
        X = np.stack([step(rb, X[b], a[b]) for b in range(B)])
    return {"%s_%s" % (kind, k): np.stack(v) for k, v in out.items()}


def main():
    rb = O.Robot("talos_like")
    out = {}
    for kind, eq in (("cost", False), ("equality", True)):
        ok = O.OracleKinoID(rb, O.talos_id_settings(rb, DT, admm_iters=100, admm_tol=-1.0, contact_motion_equality=eq, **KINO), B)
        out.update(run(rb, kind, ok.solve))
    contact, w, com, feet = cent_targets(rb)
    oc = O.OracleKinoID(rb, O.talos_id_settings(rb, DT, admm_iters=100, admm_tol=-1.0, centroidal=True, **CENT), B)
    oc.setTargetCentroidal(com, np.zeros(3), feet, np.zeros((2, 3)), contact, w)
    out.update(run(rb, "cent", oc.solve))
    path = os.path.join(HERE, "talos_id_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
