"""Oracle checks of the centroidal OCP with 6-D feet (reference src/centroidal-dynamics.cpp:84-91, force_size == 6 branches of Aligator's
centroidal dynamics / acceleration residuals): Jacobians against central differences, the structural facts of the reference's own test
(tests/problem.cpp:231-232: 6 cost components, 1 constraint block with one foot in contact), and a closed loop of the MPC."""
import numpy as np
import pytest

import oracle_lib as O


@pytest.fixture(scope="module")
def rb():
    return O.Robot("talos_like")


@pytest.fixture(scope="module")
def cent(rb):
    return O.Cent(rb, O.talos_centroidal_settings(rb))


def _point(rng):
    x = np.concatenate([rng.normal(size=3) * 0.1 + [0, 0, 0.9], rng.normal(size=3) * 5, rng.normal(size=3)])
    u = np.concatenate([np.concatenate([rng.normal(size=3) * 30 + [0, 0, 400], rng.normal(size=3) * 5]) for _ in range(2)])
    pos = rng.normal(size=(2, 3)) * 0.2
    return x, u, pos


def test_dimensions(cent):
    assert cent.nu == 12 and cent.nc == 34  # reference src/centroidal-dynamics.cpp:31-32; 17 wrench-cone rows per foot


@pytest.mark.parametrize("mask", [0b11, 0b01, 0b10])
def test_stage_derivatives_vs_finite_differences(cent, mask):
    rng = np.random.default_rng(30 + mask)
    x, u, pos = _point(rng)
    u_ref = np.array([0, 0, 450.0, 0, 0, 0] * 2)
    x_tgt = rng.normal(size=9)
    d = cent.deriv(mask, u_ref, x_tgt, pos, x, u)
    n, m = 9, 12
    A, B, lx, lu = np.zeros((n, n)), np.zeros((n, m)), np.zeros(n), np.zeros(m)
    Cx, Cu = np.zeros((cent.nc, n)), np.zeros((cent.nc, m))
    h = 1e-5
    for k in range(n):
        dd = np.zeros(n)
        dd[k] = h
        ep, em = cent.eval(mask, u_ref, x_tgt, pos, x + dd, u), cent.eval(mask, u_ref, x_tgt, pos, x - dd, u)
        A[:, k] = (ep["xnext"] - em["xnext"]) / (2 * h)
        lx[k] = (ep["cost"] - em["cost"]) / (2 * h)
        Cx[:, k] = (ep["c"] - em["c"]) / (2 * h)
    for k in range(m):
        dd = np.zeros(m)
        dd[k] = h
        ep, em = cent.eval(mask, u_ref, x_tgt, pos, x, u + dd), cent.eval(mask, u_ref, x_tgt, pos, x, u - dd)
        B[:, k] = (ep["xnext"] - em["xnext"]) / (2 * h)
        lu[k] = (ep["cost"] - em["cost"]) / (2 * h)
        Cu[:, k] = (ep["c"] - em["c"]) / (2 * h)
    assert np.abs(A - d["A"]).max() < 1e-8
    assert np.abs(B - d["B"]).max() < 1e-8
    assert np.abs(lx - d["lx"]).max() < 1e-6 * max(1, np.abs(d["lx"]).max())
    assert np.abs(lu - d["lu"]).max() < 1e-6 * max(1, np.abs(d["lu"]).max())
    assert np.abs(Cx - d["Cx"]).max() < 1e-6 and np.all(d["Cx"] == 0)
    assert np.abs(Cu - d["Cu"]).max() < 1e-6
    # contact torques: the angular momentum rate takes them as they are (B rows 6..8, columns 6 f + 3 ..)
    for f in range(2):
        blk = d["B"][6:9, 6 * f + 3: 6 * f + 6]
        assert np.allclose(blk, 0.01 * np.eye(3) * ((mask >> f) & 1))


def test_constraint_blocks_of_the_reference_test(cent):
    """reference tests/problem.cpp:198-232: left foot in contact only -> one constraint block (its wrench cone); six cost components."""
    x, u, pos = _point(np.random.default_rng(1))
    d = cent.deriv(0b01, np.zeros(12), np.zeros(9), pos, x, u)
    present = np.abs(d["Cu"]).sum(1) > 0
    assert present[:17].all() and not present[17:].any()
    components = ["com", "control", "linear_mom", "angular_mom", "linear_acc", "angular_acc"]
    assert len(components) == 6


def test_closed_loop_of_the_mpc(rb, cent):
    ms = O.talos_mpc_settings(rb, max_iters=2)
    ms["T"] = 30
    m = O.OracleCentMPC(cent, ms, 2)
    m.generateCycleHorizon(O.walk_cycle(6, 12))
    m.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
    rng = np.random.default_rng(0)
    X = np.stack([rb.integrate(rb.x_ref, rng.normal(size=rb.ndx) * 0.01) for _ in range(2)])
    for i in range(40):
        m.iterate(X)
    us = m.us
    assert np.isfinite(m.xs).all() and np.isfinite(us).all()
    # the weight is carried by the feet in contact
    fz = us[0, 0, [2, 8]]
    assert abs(fz.sum() - rb.mass * 9.81) < 0.2 * rb.mass * 9.81
