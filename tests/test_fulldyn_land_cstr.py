"""land_cstr of the full-dynamics OCP: at the stage where a foot lands (in contact there, not in the stage before of the cycle) its
LOCAL_WORLD_ALIGNED frame velocity is constrained to zero -- all 6 rows of a 6-D foot (reference src/fulldynamics.cpp:175-181), the 3 linear rows
of a 3-D foot plus the height of its contact pose, FrameTranslationResidual sliced to z (src/fulldynamics.cpp:191-210); EqualityConstraint rows
that depend on the state only.  Oracle rows against finite differences; the device kernels (instantiations FullDims<13, 4, 3, 5, 4> and
FullDims<23, 2, 6, 0, 6>) against the oracle in closed loop across a touch-down."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

LAND = {"land_cstr": True}


def _fd_rows(rb, s, x, u, mask, land, nl):
    F = O.Full(rb, s)
    nf, fs = F.nf, int(s.get("force_size", 3))
    base = F.nc - nl * nf
    m = mask | (land << 8)
    u_ref, foot = np.zeros(F.nu + fs * nf), np.zeros((nf, 3))
    c = F.eval(m, u_ref, rb.x_ref, foot, x, u)["c"][base:].reshape(nf, nl)
    d = F.deriv(m, u_ref, rb.x_ref, foot, x, u)
    eps, J = 1e-6, np.zeros((nl * nf, F.ndx))
    for i in range(F.ndx):
        dx = np.zeros(F.ndx)
        dx[i] = eps
        J[:, i] = (F.eval(m, u_ref, rb.x_ref, foot, rb.integrate(x, dx), u)["c"][base:] - F.eval(m, u_ref, rb.x_ref, foot, rb.integrate(x, -dx), u)["c"][base:]) / (2 * eps)
    assert np.abs(J - d["Cx"][base:]).max() < 1e-6 * max(1.0, np.abs(J).max())
    assert np.abs(d["Cu"][base:]).max() == 0.0
    return F, c, d["Cx"][base:].reshape(nf, nl, F.ndx)


def test_oracle_land_rows_of_point_feet():
    rb = O.Robot("go2_like")
    s = dict(O.go2_full_settings(rb), **LAND)
    x = S.random_states(rb, 1, seed=5)[0]
    u = np.random.default_rng(1).normal(0, 3, 12)
    F, c, Cx = _fd_rows(rb, s, x, u, 0b0111, 0b1010, 4)
    assert F.nc == 12 + 12 + 4 * 4
    # rows of foot 1 only: foot 3 lands in the flags but is not in contact, feet 0 and 2 do not land
    assert np.all(c[[0, 2, 3]] == 0.0) and np.all(Cx[[0, 2, 3]] == 0.0) and np.all(c[1] != 0.0)


def test_oracle_land_rows_of_planar_feet():
    rb = O.Robot("talos_like")
    s = dict(O.talos_full_settings(rb), **LAND)
    x = S.talos_random_states(rb, 1, seed=5)[0]
    u = np.random.default_rng(1).normal(0, 3, rb.nv - 6)
    F, c, Cx = _fd_rows(rb, s, x, u, 0b11, 0b01, 6)
    assert F.nc == 22 + 22 + 17 * 2 + 6 * 2
    assert np.all(c[1] == 0.0) and np.all(Cx[1] == 0.0) and np.all(c[0] != 0.0)


# a quick trot (2 stages on four feet, 6 on two): the first touch-down enters the horizon at the 8th control step
QUICK = {"T_fly": 6, "T_contact": 2}


def _go2_loop(lib, iters, steps, tol, override, horizon=20, B=2):
    om, rb = S.make_full_oracle(B, iters, horizon, settings_override=override, mpc_override=QUICK, walk=None)
    gm, _, _, _ = S.make_full_product(B, iters, lib, horizon, settings_override=override, mpc_override=QUICK)
    for m in (om, gm):
        m.generateCycleHorizon(O.trot_cycle(2, 6))
        m.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, B)
    nl, active, worst = 16, 0, 0.0
    for _ in range(steps):
        om.iterate(X)
        gm.iterate(X)
        worst = max(worst, S.rel_err(om.xs, gm.xs), 0.1 * S.rel_err(om.us, gm.us))
        assert worst < tol, worst
        assert S.alphas_agree(om, gm, rtol=1e-6)
        assert S.rel_err(om.vs[:, :, -nl:], gm.vs[:, :, -nl:]) < 1e3 * tol  # multipliers of the land rows (of size 1e3 and more)
        active = max(active, int((om.vs[:, :, -nl:] != 0).sum()))
        X = om.xs[:, 1, :].copy()
    return active, worst


def test_emulated_kernels_with_landing_feet(built):
    active, _ = _go2_loop(S.emu_lib(), 2, 16, 1e-6, LAND)
    assert active >= 8  # 4 rows per landing foot, two feet land together in a trot


def test_emulated_kernels_with_landing_feet_and_cones(built):
    active, _ = _go2_loop(S.emu_lib(), 2, 12, 1e-5, dict(LAND, force_cone=True, mu=0.6))
    assert active >= 8


def _talos_loop(lib, iters, steps, tol, B=1, horizon=30):
    om, gm, rb = S.make_talos_pair(B, max_iters=iters, lib=lib, horizon=horizon, settings_override=LAND, mpc_override={"T_fly": 8, "T_contact": 2},
                                   cycle=O.walk_cycle(2, 8))
    X = S.talos_random_states(rb, B)
    active = 0
    for _ in range(steps):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < tol and S.rel_err(om.us, gm.us) < 10 * tol
        assert S.alphas_agree(om, gm, rtol=1e-6)
        active = max(active, int((om.vs[:, :, -12:] != 0).sum()))
        X = om.xs[:, 1, :].copy()
    return active


def test_emulated_kernels_with_landing_planar_feet(built):
    # (the instantiation whose second sweep would have 129 columns: two dense rows ride in the last control panel, smpc_riccati_dense.h)
    assert _talos_loop(S.emu_lib(), 2, 14, 1e-6) >= 6


@pytest.mark.gpu
@pytest.mark.parametrize("iters", [1, 3])
def test_hip_full_dynamics_with_landing_feet(built, iters):
    active, _ = _go2_loop(None, iters, 30, 1e-6, LAND, B=3)
    assert active >= 8


@pytest.mark.gpu
def test_hip_full_dynamics_with_landing_feet_and_cones(built):
    active, _ = _go2_loop(None, 2, 16, 1e-5, dict(LAND, force_cone=True, mu=0.6), B=3)
    assert active >= 8


@pytest.mark.gpu
def test_hip_talos_with_landing_feet(built):
    assert _talos_loop(None, 2, 16, 1e-6, B=2) >= 6
