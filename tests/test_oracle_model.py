"""Oracle self-validation (CPU): the restated Pinocchio/Aligator pieces against finite differences and
physics identities (the reference holds no numerical fixture for this path, so the oracle is
pinned by self-consistency)."""
import numpy as np
import pytest

import oracle_lib as O


@pytest.fixture(scope="module")
def rb():
    return O.Robot("go2_like")


@pytest.fixture(scope="module")
def kino(rb):
    return O.Kino(rb, O.go2_kino_settings(rb))


def _randx(rb, rng, scale=1.0):
    dx = np.concatenate([rng.normal(size=3) * 0.05, rng.normal(size=3) * 0.3, rng.normal(size=12) * 0.3,
                         rng.normal(size=6) * 0.5, rng.normal(size=12) * 1.0]) * scale
    return rb.integrate(rb.x_ref, dx)


def test_robot_dimensions(rb):
    # Go2 has the joint counts of Solo12 pinned by the reference (tests/robot_handler.cpp:60-61): nq 19, nv 18
    assert (rb.nq, rb.nv, rb.nf) == (19, 18, 4)
    assert abs(rb.mass - 15.017) < 1e-9


@pytest.mark.parametrize("scale", [1e-3, 0.04, 0.3, 1.5])
def test_se3_exp_log_jacobians(scale):
    L = O.lib()
    rng = np.random.default_rng(3)
    nu = rng.normal(size=6) * scale

    def exp6(n):
        R, p = np.zeros(9), np.zeros(3)
        L.orc_exp6(np.ascontiguousarray(n), R, p)
        M = np.eye(4)
        M[:3, :3] = R.reshape(3, 3)
        M[:3, 3] = p
        return M

    def log6(M):
        out = np.zeros(6)
        L.orc_log6(np.ascontiguousarray(M[:3, :3]).ravel(), np.ascontiguousarray(M[:3, 3]), out)
        return out

    assert np.abs(log6(exp6(nu)) - nu).max() < 1e-12
    J, Jl = np.zeros((6, 6)), np.zeros((6, 6))
    L.orc_Jexp6(nu, J)
    L.orc_Jlog6_of_exp(nu, Jl)
    assert np.abs(Jl @ J - np.eye(6)).max() < 1e-12
    h = 1e-6
    Jn = np.zeros((6, 6))
    Mi = np.linalg.inv(exp6(nu))
    for k in range(6):
        d = np.zeros(6)
        d[k] = h
        Jn[:, k] = (log6(Mi @ exp6(nu + d)) - log6(Mi @ exp6(nu - d))) / (2 * h)
    assert np.abs(J - Jn).max() < 1e-8


def test_integrate_difference_roundtrip(rb):
    rng = np.random.default_rng(0)
    x0 = _randx(rb, rng)
    dx = rng.normal(size=rb.ndx) * 0.3
    x1 = rb.integrate(x0, dx)
    assert abs(np.linalg.norm(x1[3:7]) - 1) < 1e-14
    assert np.abs(rb.difference(x0, x1) - dx).max() < 1e-12


def test_centroidal_identities(rb):
    rng = np.random.default_rng(1)
    x = _randx(rb, rng)
    v = x[rb.nq:]
    c = rb.centroidal(x)
    # hg = Ag v ; linear momentum = m * com velocity
    assert np.abs(c["Ag"] @ v - c["hg"]).max() < 1e-12
    h = 1e-6
    xp = rb.integrate(x, np.concatenate([v * h, np.zeros(rb.nv)]))
    xm = rb.integrate(x, np.concatenate([-v * h, np.zeros(rb.nv)]))
    vcom = (rb.centroidal(xp)["com"] - rb.centroidal(xm)["com"]) / (2 * h)
    assert np.abs(rb.mass * vcom - c["hg"][:3]).max() < 1e-7
    dAg = (rb.centroidal(xp)["Ag"] - rb.centroidal(xm)["Ag"]) / (2 * h)
    assert np.abs(dAg @ v - c["dAgv"]).max() < 1e-6


def test_kinodynamics_momentum_balance(rb, kino):
    """d/dt hg along xdot = [m g + sum f ; sum (p - c) x f] for the solved base acceleration."""
    rng = np.random.default_rng(2)
    x = _randx(rb, rng)
    mask = 0b1001
    u = np.concatenate([rng.normal(size=12) * 10 + np.tile([0, 0, 40], 4), rng.normal(size=12) * 2])
    e = kino.eval(mask, np.zeros(24), rb.x_ref, np.zeros((4, 3)), x, u)
    v, a = e["xdot"][:rb.nv], e["xdot"][rb.nv:]
    c = rb.centroidal(x)
    hdot = c["Ag"] @ a + c["dAgv"]
    f = u[:12].reshape(4, 3)
    act = [(mask >> i) & 1 for i in range(4)]
    lin = rb.mass * np.array([0, 0, -9.81]) + sum(f[i] for i in range(4) if act[i])
    ang = sum(np.cross(c["feet"][i] - c["com"], f[i]) for i in range(4) if act[i])
    assert np.abs(hdot[:3] - lin).max() < 1e-9
    assert np.abs(hdot[3:] - ang).max() < 1e-9
    assert np.abs(a[6:] - u[12:]).max() == 0.0
    assert np.abs(v - x[rb.nq:]).max() == 0.0


@pytest.mark.parametrize("mask", [0b1111, 0b0110, 0b1001])
def test_stage_derivatives_vs_finite_differences(rb, kino, mask):
    rng = np.random.default_rng(10 + mask)
    x = _randx(rb, rng)
    u = np.concatenate([rng.normal(size=12) * 10 + np.tile([0, 0, 40], 4), rng.normal(size=12) * 2])
    u_ref = np.concatenate([np.tile([0, 0, 36.0], 4), np.zeros(12)])
    x_tgt = _randx(rb, rng, 0.3)
    foot_ref = rng.normal(size=(4, 3)) * 0.2
    d = kino.deriv(mask, u_ref, x_tgt, foot_ref, x, u)
    e0 = kino.eval(mask, u_ref, x_tgt, foot_ref, x, u)
    n, m = kino.ndx, kino.nu
    A, B = np.zeros((n, n)), np.zeros((n, m))
    lx, lu = np.zeros(n), np.zeros(m)
    Cx, Cu = np.zeros((kino.nc, n)), np.zeros((kino.nc, m))
    h = 1e-6
    for k in range(n):
        dd = np.zeros(n)
        dd[k] = h
        ep = kino.eval(mask, u_ref, x_tgt, foot_ref, rb.integrate(x, dd), u)
        em = kino.eval(mask, u_ref, x_tgt, foot_ref, rb.integrate(x, -dd), u)
        A[:, k] = (rb.difference(e0["xnext"], ep["xnext"]) - rb.difference(e0["xnext"], em["xnext"])) / (2 * h)
        lx[k] = (ep["cost"] - em["cost"]) / (2 * h)
        Cx[:, k] = (ep["c"] - em["c"]) / (2 * h)
    for k in range(m):
        dd = np.zeros(m)
        dd[k] = h
        ep = kino.eval(mask, u_ref, x_tgt, foot_ref, x, u + dd)
        em = kino.eval(mask, u_ref, x_tgt, foot_ref, x, u - dd)
        B[:, k] = (rb.difference(e0["xnext"], ep["xnext"]) - rb.difference(e0["xnext"], em["xnext"])) / (2 * h)
        lu[k] = (ep["cost"] - em["cost"]) / (2 * h)
        Cu[:, k] = (ep["c"] - em["c"]) / (2 * h)
    assert np.abs(A - d["A"]).max() < 1e-7
    assert np.abs(B - d["B"]).max() < 1e-8
    assert np.abs(lx - d["lx"]).max() < 1e-6 * max(1, np.abs(d["lx"]).max())
    assert np.abs(lu - d["lu"]).max() < 1e-6
    assert np.abs(Cx - d["Cx"]).max() < 1e-7
    assert np.abs(Cu - d["Cu"]).max() < 1e-9
    # rows of swinging feet are absent
    for f in range(4):
        if not (mask >> f) & 1:
            assert np.all(d["Cx"][12 + 3 * f: 15 + 3 * f] == 0)
            assert np.all(e0["c"][12 + 3 * f: 15 + 3 * f] == 0)


def test_gauss_newton_hessian_is_psd_and_symmetric(rb, kino):
    rng = np.random.default_rng(5)
    x = _randx(rb, rng)
    u = rng.normal(size=24)
    d = kino.deriv(15, np.zeros(24), rb.x_ref, np.zeros((4, 3)), x, u)
    Hm = np.block([[d["Lxx"], d["Lxu"]], [d["Lxu"].T, d["Luu"]]])
    assert np.abs(Hm - Hm.T).max() < 1e-9
    assert np.linalg.eigvalsh(Hm).min() > -1e-8


def test_terminal_cost_gradient(rb, kino):
    rng = np.random.default_rng(6)
    x = _randx(rb, rng)
    c0, lx, Lxx = kino.term(rb.x_ref, x)
    g = np.zeros(kino.ndx)
    h = 1e-6
    for k in range(kino.ndx):
        dd = np.zeros(kino.ndx)
        dd[k] = h
        g[k] = (kino.term(rb.x_ref, rb.integrate(x, dd))[0] - kino.term(rb.x_ref, rb.integrate(x, -dd))[0]) / (2 * h)
    assert np.abs(g - lx).max() < 1e-6 * max(1, np.abs(lx).max())
    assert np.abs(Lxx - Lxx.T).max() < 1e-9
