#!/usr/bin/env python3
"""Golden vectors of the whole-body inverse-dynamics QPs (tests/golden/go2_id_golden.npz), produced by the CPU oracle in the build
container (TSID / ProxQP cannot be built or imported here):

    python tests/golden/make_golden_id.py

KinodynamicsID and CentroidalID on the go2_like robot, 3 robots each, 100 ADMM iterations per tick (fixed count): a closed loop of 12 ticks
from perturbed states with one robot's foot in the air -- per tick the states that went in and the torques, accelerations and contact
forces that came out.  Inputs and expected outputs only, no code."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mpc_setup as S  # noqa: E402
import oracle_lib as O  # noqa: E402
from test_oracle_id import DT, static_forces, step  # noqa: E402

KINO = dict(kp_base=10.0, kp_posture=1.0, kp_contact=10.0, w_base=10.0, w_posture=0.1, w_contact_force=1e-3, w_contact_motion=1.0)
CENT = dict(KINO, kp_com=7.0, kp_feet_tracking=5.0, w_com=10.0, w_feet_tracking=100.0, centroidal=True)
TICKS, B = 12, 3


def targets(rb):
    """Shared by the generator and the tests: robot 1 lifts its rear-left foot 5 cm."""
    contact = [True, True, False, True]
    c = rb.centroidal(rb.x_ref)
    feet = c["feet"].copy()
    feet[2] += [0.05, -0.05, 0.05]
    return contact, static_forces(rb, contact=contact), c["com"] + [0.01, 0.0, 0.02], feet


def run(rb, kind, solver):
    """solver: object with setTarget / setTargetCentroidal-like closures and solve(X) -> (tau, a, f)."""
    X = S.random_states(rb, B, seed=11, scale=0.4)
    out = dict(X=[], tau=[], a=[], f=[])
    for _ in range(TICKS):
        tau, a, f = solver(X)
        for k, v in zip(("X", "tau", "a", "f"), (X, tau, a, f)):
            out[k].append(np.array(v, float).copy())
        X = np.stack([step(rb, X[b], a[b]) for b in range(B)])
    return {"%s_%s" % (kind, k): np.stack(v) for k, v in out.items()}


def main():
    rb = O.Robot("go2_like")
    contact, fs, com, feet = targets(rb)
    out = {}
    ok = O.OracleKinoID(rb, O.id_settings(rb, DT, admm_iters=100, admm_tol=-1.0, **KINO), B)
    ok.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), contact, fs, instance=1)
    out.update(run(rb, "kino", lambda X: ok.solve(X)))
    oc = O.OracleKinoID(rb, O.id_settings(rb, DT, admm_iters=100, admm_tol=-1.0, **CENT), B)
    oc.setTargetCentroidal(com, np.array([0.1, 0.0, -0.05]), feet, np.zeros((4, 3)), contact, fs, instance=1)
    out.update(run(rb, "cent", lambda X: oc.solve(X)))
    path = os.path.join(HERE, "go2_id_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
