"""force_cone of the full-dynamics OCP with 3-D point feet: MultibodyFrictionConeResidual in NegativeOrthant for every foot in contact
(reference src/fulldynamics.cpp:185-190): five linear rows per foot on the contact force of the constrained dynamics (unilaterality and the
friction pyramid +-f_x, +-f_y <= mu f_z), dense in x and u through d lambda / d(x, u).  Oracle rows against finite differences; the device
kernels (instantiation FullDims<13, 4, 3, 5>) against the oracle in closed loop, with a friction coefficient small enough to activate rows."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

CONE = {"force_cone": True, "mu": 0.6}
WALK = (0.3, 0.1, 0, 0, 0, 0.2)
# (the contact forces of the Go2-class model live in the LOCAL frame of the calf -- src/fulldynamics.cpp:56-65 -- whose z axis is tilted by
#  about 40 degrees when standing: with mu = 0.6 most pyramid rows are active from the cold start on (160 of 800 row-stages); with a narrow
#  pyramid (mu = 0.25) every line search ends at 2^-7 .. 2^-9 on a merit of 1e9 and the last bits decide the step, on CPU and GPU alike)


def test_oracle_cone_rows_and_jacobians():
    rb = O.Robot("go2_like")
    s = O.go2_full_settings(rb)
    s.update(CONE)
    F = O.Full(rb, s)
    assert F.nc == 12 + 12 + 5 * 4  # torque box | joint box | 5 rows per foot
    rng = np.random.default_rng(2)
    x = S.random_states(rb, 1, seed=3)[0]
    u = rng.normal(0, 3, F.nu)
    u_ref = np.zeros(F.nu + 12)
    foot = np.zeros((4, 3))
    mask = 0b1011
    ev = F.eval(mask, u_ref, rb.x_ref, foot, x, u)
    c = ev["c"][24:].reshape(4, 5)
    assert np.all(c[2] == 0.0)  # the foot in the air has no rows
    d = F.deriv(mask, u_ref, rb.x_ref, foot, x, u)
    eps = 1e-6
    Ju = np.zeros((20, F.nu))
    for i in range(F.nu):
        du = np.zeros(F.nu)
        du[i] = eps
        Ju[:, i] = (F.eval(mask, u_ref, rb.x_ref, foot, x, u + du)["c"][24:] - F.eval(mask, u_ref, rb.x_ref, foot, x, u - du)["c"][24:]) / (2 * eps)
    assert np.abs(Ju - d["Cu"][24:]).max() < 1e-5 * max(1.0, np.abs(Ju).max())
    Jx = np.zeros((20, F.ndx))
    for i in range(F.ndx):
        dx = np.zeros(F.ndx)
        dx[i] = eps
        Jx[:, i] = (F.eval(mask, u_ref, rb.x_ref, foot, rb.integrate(x, dx), u)["c"][24:] - F.eval(mask, u_ref, rb.x_ref, foot, rb.integrate(x, -dx), u)["c"][24:]) / (2 * eps)
    assert np.abs(Jx - d["Cx"][24:]).max() < 1e-4 * max(1.0, np.abs(Jx).max())
    # the rows are the pyramid on the contact forces the dynamics returns: c = [-fz, -fx - mu fz, fx - mu fz, -fy - mu fz, fy - mu fz]
    for k in (0, 1, 3):
        fz = -c[k, 0]
        fx = (c[k, 2] - c[k, 1]) / 2
        fy = (c[k, 4] - c[k, 3]) / 2
        assert np.isclose(c[k, 1], -fx - 0.6 * fz) and np.isclose(c[k, 4], fy - 0.6 * fz)


def _loop(lib, iters, steps, tol):
    om, gm, rb = S.make_full_pair(2, max_iters=iters, lib=lib, horizon=20, walk=WALK, settings_override=CONE)
    assert gm.nc == 44
    X = S.random_states(rb, 2)
    active = 0
    for step in range(steps):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < tol, (step, S.rel_err(om.xs, gm.xs))
        assert S.rel_err(om.us, gm.us) < 10 * tol, (step, S.rel_err(om.us, gm.us))
        assert S.alphas_agree(om, gm, rtol=1e-6)
        active = max(active, int((om.vs[:, :, 24:] != 0).sum()))
        X = om.xs[:, 1, :].copy()
    return active


def test_emulated_kernels_with_friction_cones():
    assert _loop(S.emu_lib(), 2, 12, 1e-9) >= 100, "the scenario must activate cone rows"


def test_emulated_kernels_wide_cones():
    """mu = 3: no row is active along the trajectory, but rows are violated at trial points of the line search (the merit sees them): the
    accepted steps differ from the cone-free problem's and must be the oracle's"""
    om, gm, rb = S.make_full_pair(2, max_iters=2, lib=S.emu_lib(), horizon=20, settings_override={"force_cone": True, "mu": 3.0})
    X = S.random_states(rb, 2)
    for step in range(5):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < 1e-9 and S.alphas_agree(om, gm, rtol=1e-6)
        X = om.xs[:, 1, :].copy()


@pytest.mark.gpu
@pytest.mark.parametrize("iters", [1, 3])
def test_hip_full_dynamics_with_friction_cones(iters):
    assert _loop(None, iters, 12, 1e-7) >= 100
