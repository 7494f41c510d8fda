"""Self-consistency of the CPU oracle's centroidal OCP (oracle/orc_cent.hpp, orc_mpc_cent.hpp): SURVEY 8(c) -- the
reference holds no numerical vectors for this path (tests/problem.cpp:198-349 checks counts and setter round trips),
so the restatement is checked against finite differences, physics identities and the reference's structural KATs."""
import numpy as np
import pytest

import oracle_lib as O


@pytest.fixture(scope="module")
def rb():
    return O.Robot("go2_like")


@pytest.fixture(scope="module")
def cent(rb):
    return O.Cent(rb, O.go2_centroidal_settings(rb))


def _point(rng, nf):
    x = rng.normal(size=9) * np.array([0.1, 0.1, 0.1, 1, 1, 1, 0.3, 0.3, 0.3]) + np.array([0, 0, 0.3, 0, 0, 0, 0, 0, 0])
    u = rng.normal(size=3 * nf) * 5.0 + np.tile([0, 0, 35.0], nf)
    pos = rng.normal(size=(nf, 3)) * 0.2
    u_ref = np.tile([0, 0, 36.0], nf)
    x_tgt = rng.normal(size=9) * 0.1
    return x, u, pos, u_ref, x_tgt


def test_structure_counts_reference_kat(rb, cent):
    # tests/problem.cpp:298-300: 6 cost components, one constraint block per foot in contact (3 of 4 -> 3 blocks)
    x, u, pos, u_ref, x_tgt = _point(np.random.default_rng(0), rb.nf)
    e = cent.eval(0b0111, u_ref, x_tgt, pos, x, u)
    rows = e["c"].reshape(rb.nf, 2)
    assert np.all(rows[3] == 0.0) and np.all(rows[:3, 0] != 0.0)
    d = cent.deriv(0b0111, u_ref, x_tgt, pos, x, u)
    assert np.all(d["B"][:, 9:12] == 0.0) and np.all(d["Cu"][6:8] == 0.0) and np.all(d["Cx"] == 0.0)
    assert cent.nu == 12 and cent.nc == 8


@pytest.mark.parametrize("mask", [0b1111, 0b0110, 0b1001, 0])
def test_derivatives_vs_finite_differences(rb, cent, mask):
    rng = np.random.default_rng(3 + mask)
    x, u, pos, u_ref, x_tgt = _point(rng, rb.nf)
    d = cent.deriv(mask, u_ref, x_tgt, pos, x, u)
    h = 1e-6

    def fd(fun, z, n):
        J = np.zeros((n, z.size))
        for i in range(z.size):
            zp, zm = z.copy(), z.copy()
            zp[i] += h
            zm[i] -= h
            J[:, i] = (fun(zp) - fun(zm)) / (2 * h)
        return J

    ev = lambda xx, uu: cent.eval(mask, u_ref, x_tgt, pos, xx, uu)
    assert np.allclose(d["A"], fd(lambda z: ev(z, u)["xnext"], x, 9), atol=1e-7)
    assert np.allclose(d["B"], fd(lambda z: ev(x, z)["xnext"], u, 9), atol=1e-7)
    assert np.allclose(d["lx"], fd(lambda z: np.array([ev(z, u)["cost"]]), x, 1)[0], atol=1e-5)
    assert np.allclose(d["lu"], fd(lambda z: np.array([ev(x, z)["cost"]]), u, 1)[0], atol=1e-5)
    assert np.allclose(d["Cu"], fd(lambda z: ev(x, z)["c"], u, cent.nc), atol=1e-5)
    # Gauss-Newton Hessian: symmetric PSD; exact for the terms that are quadratic in (x, u) separately
    Hm = np.block([[d["Lxx"], d["Lxu"]], [d["Lxu"].T, d["Luu"]]])
    assert np.allclose(Hm, Hm.T) and np.linalg.eigvalsh(Hm).min() > -1e-9
    Huu = fd(lambda z: cent.deriv(mask, u_ref, x_tgt, pos, x, z)["lu"], u, 12)
    assert np.allclose(d["Luu"], Huu, atol=1e-5)  # residuals are linear in u -> GN is exact in the u block


def test_dynamics_is_newton_euler(rb, cent):
    x, u, pos, u_ref, x_tgt = _point(np.random.default_rng(5), rb.nf)
    e = cent.eval(0b1111, u_ref, x_tgt, pos, x, u)
    F = u.reshape(4, 3)
    assert np.allclose(e["xdot"][:3], x[3:6] / rb.mass)
    assert np.allclose(e["xdot"][3:6], rb.mass * np.array([0, 0, -9.81]) + F.sum(0))
    assert np.allclose(e["xdot"][6:9], sum(np.cross(pos[i] - x[:3], F[i]) for i in range(4)))
    assert np.allclose(e["xnext"], x + 0.01 * e["xdot"])
    # the same step through the stand-alone dynamics restatement (orc_centroidal_dynamics)
    xn, A, B = O.centroidal_dynamics(rb.mass, [0, 0, -9.81], 0.01, x, u, [1, 1, 1, 1], pos)
    d = cent.deriv(0b1111, u_ref, x_tgt, pos, x, u)
    assert np.allclose(xn, e["xnext"]) and np.allclose(A, d["A"]) and np.allclose(B, d["B"])


def test_terminal_cost(cent):
    x = np.random.default_rng(6).normal(size=9)
    c, lx, Lxx = cent.term(x)
    s = cent.s
    assert np.isclose(c, 0.5 * x[3:6] @ s["w_linear_mom"] @ x[3:6] + 0.5 * x[6:9] @ s["w_angular_mom"] @ x[6:9])
    assert np.allclose(lx[:3], 0) and np.allclose(lx[3:6], s["w_linear_mom"] @ x[3:6]) and np.allclose(Lxx[6:, 6:], s["w_angular_mom"])


def _mpc(rb, cent, B, k):
    m = O.OracleCentMPC(cent, O.go2_mpc_settings(rb, max_iters=k), B)
    m.generateCycleHorizon(O.trot_cycle())
    m.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    return m


def test_cold_solve_and_standing_balance(rb, cent):
    m = O.OracleCentMPC(cent, O.go2_mpc_settings(rb, max_iters=1), 1)
    tr = m.cold_trace()
    assert tr[-1, 1] < 1e-4  # primal feasible
    # standing on identity contact poses (src/ocp-handler.cpp:117): the forces carry the weight
    us = m.us[0]
    assert np.allclose(us.reshape(50, 4, 3)[:, :, 2].sum(1), rb.mass * 9.81, rtol=2e-2)


def test_mpc_state_machine_and_references(rb, cent):
    m = _mpc(rb, cent, 2, 2)
    X = np.stack([rb.x_ref, rb.integrate(rb.x_ref, np.r_[0.01, -0.02, 0, np.zeros(15), 0.1, np.zeros(17)])])
    cst = [rb.centroidal(x) for x in X]
    for _ in range(12):
        m.iterate(X)
    xs, us, info = m.xs, m.us, m.info
    for b in range(2):
        assert np.allclose(xs[b, 0, :3], cst[b]["com"]) and np.allclose(xs[b, 0, 3:], cst[b]["hg"])
    assert np.isfinite(xs).all() and np.isfinite(us).all()
    assert info[:, 4].max() < 1e-3  # primal infeasibility of the last iterate
    # swinging feet carry no force reference and their columns are out of the dynamics: u stays at the warm start / 0 cost pull
    assert m.timing(0, 1)[0] >= 0
    # friction cone holds on the stance feet up to the AL tolerance
    F = us[:, :, :].reshape(2, 50, 4, 3)
    cone = np.hypot(F[..., 0], F[..., 1]) - 0.8 * np.abs(F[..., 2])
    assert cone.max() < 1e-2


def test_oracle_is_deterministic_across_thread_counts(rb, cent):
    outs = []
    for nt in (1, 4):
        ms = O.go2_mpc_settings(rb, max_iters=2, num_threads=nt)
        m = O.OracleCentMPC(cent, ms, 3)
        m.generateCycleHorizon(O.trot_cycle())
        m.switchToWalk(np.array([0.3, 0, 0, 0, 0, 0.1]))
        X = np.stack([rb.x_ref] * 3)
        for _ in range(3):
            m.iterate(X)
        outs.append((m.xs.copy(), m.us.copy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
