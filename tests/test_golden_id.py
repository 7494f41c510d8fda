"""Committed golden vectors of the inverse-dynamics QPs (tests/golden/go2_id_golden.npz, made by tests/golden/make_golden_id.py from the
CPU oracle): the oracle still reproduces them (CPU tier), the kernel bodies compiled for the CPU and the HIP library follow them tick by
tick from the recorded states (fixed iteration count: the per-tick comparison is open loop, so rounding does not accumulate)."""
import os

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O
import simple_mpc

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "go2_id_golden.npz"))
import sys

sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden_id as M  # noqa: E402  (settings and targets of the fixture)


def test_oracle_reproduces_the_golden_vectors(built):
    rb = O.Robot("go2_like")
    contact, fs, com, feet = M.targets(rb)
    ok = O.OracleKinoID(rb, O.id_settings(rb, M.DT, admm_iters=100, admm_tol=-1.0, **M.KINO), M.B)
    ok.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), contact, fs, instance=1)
    oc = O.OracleKinoID(rb, O.id_settings(rb, M.DT, admm_iters=100, admm_tol=-1.0, **M.CENT), M.B)
    oc.setTargetCentroidal(com, np.array([0.1, 0.0, -0.05]), feet, np.zeros((4, 3)), contact, fs, instance=1)
    for kind, o in (("kino", ok), ("cent", oc)):
        for t in range(M.TICKS):
            tau, a, f = o.solve(G[kind + "_X"][t])
            assert S.rel_err(G[kind + "_tau"][t], tau) < 1e-9 and S.rel_err(G[kind + "_a"][t], a) < 1e-9 and S.rel_err(G[kind + "_f"][t], f) < 1e-9, (kind, t)


def _device_follows(lib):
    rb = O.Robot("go2_like")
    contact, fs, com, feet = M.targets(rb)
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in S.FEET:
        mh.addPointFoot(n, "root_joint")
    kk = simple_mpc.KinodynamicsID(mh, M.DT, M.KINO, O.GO2_EFFORT, O.GO2_VMAX, batch=M.B, lib=lib, admm_iters=100, admm_tol=-1.0)
    kk.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), contact, fs.reshape(4, 3), instance=1)
    kc = simple_mpc.CentroidalID(mh, M.DT, {k: v for k, v in M.CENT.items() if k != "centroidal"}, O.GO2_EFFORT, O.GO2_VMAX, batch=M.B, lib=lib,
                                 admm_iters=100, admm_tol=-1.0)
    kc.setTarget(com, np.array([0.1, 0.0, -0.05]), feet, np.zeros((4, 3)), contact, fs.reshape(4, 3), instance=1)
    for kind, k in (("kino", kk), ("cent", kc)):
        for t in range(M.TICKS):
            X = G[kind + "_X"][t]
            tau = k.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
            assert S.rel_err(G[kind + "_tau"][t], tau) < 1e-7, (kind, t, S.rel_err(G[kind + "_tau"][t], tau))
            assert S.rel_err(G[kind + "_a"][t], k.getAccelerations()) < 1e-7 and S.rel_err(G[kind + "_f"][t], k.getContactForces().reshape(M.B, -1)) < 1e-7


def test_emulated_kernels_follow_the_golden_vectors(built):
    _device_follows(S.emu_lib())


@pytest.mark.gpu
def test_hip_follows_the_golden_vectors(built):
    _device_follows(None)
