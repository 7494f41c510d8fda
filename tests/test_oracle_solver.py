"""Oracle self-validation (CPU): proximal Riccati against a dense KKT solve, ProxDDP iteration
properties, and the MPC host state machine (integer known-answer tests lifted from the reference's
tests/mpc.cpp:78-90)."""
import numpy as np
import scipy.linalg as sla

import oracle_lib as O
import mpc_setup as S


def _random_lq(rng, H, ndx, nu, nc, mu):
    def spd(n, lo=0.1):
        M = rng.normal(size=(n, n))
        return M @ M.T + lo * np.eye(n)

    Q = np.stack([spd(ndx) for _ in range(H)])
    R = np.stack([spd(nu) for _ in range(H)])
    S_ = rng.normal(size=(H, ndx, nu)) * 0.1
    A = np.stack([np.eye(ndx) + 0.1 * rng.normal(size=(ndx, ndx)) for _ in range(H)])
    B = rng.normal(size=(H, ndx, nu))
    Cm = rng.normal(size=(H, nc, ndx))
    D = rng.normal(size=(H, nc, nu))
    q, r = rng.normal(size=(H, ndx)), rng.normal(size=(H, nu))
    f, d = rng.normal(size=(H, ndx)) * 0.1, rng.normal(size=(H, nc)) * 0.1
    QN, qN = spd(ndx), rng.normal(size=ndx)
    return Q, S_, R, q, r, A, B, f, Cm, D, d, QN, qN


def _dense_kkt(Q, S_, R, q, r, A, B, f, Cm, D, d, QN, qN, mu):
    H, ndx, nu = B.shape
    nc = Cm.shape[1]
    per = nu + nc + 2 * ndx
    N = H * per
    K = np.zeros((N, N))
    rhs = np.zeros(N)
    iu = lambda t: t * per
    iv = lambda t: t * per + nu
    il = lambda t: t * per + nu + nc
    ix = lambda t: t * per + nu + nc + ndx
    for t in range(H):
        r0 = iu(t)
        K[r0:r0 + nu, iu(t):iu(t) + nu] = R[t]
        K[r0:r0 + nu, il(t):il(t) + ndx] = B[t].T
        K[r0:r0 + nu, iv(t):iv(t) + nc] = D[t].T
        if t > 0:
            K[r0:r0 + nu, ix(t - 1):ix(t - 1) + ndx] = S_[t].T
        rhs[r0:r0 + nu] = -r[t]
        r0 = iv(t)
        K[r0:r0 + nc, iv(t):iv(t) + nc] = -mu * np.eye(nc)
        K[r0:r0 + nc, iu(t):iu(t) + nu] = D[t]
        if t > 0:
            K[r0:r0 + nc, ix(t - 1):ix(t - 1) + ndx] = Cm[t]
        rhs[r0:r0 + nc] = -d[t]
        r0 = il(t)
        K[r0:r0 + ndx, iu(t):iu(t) + nu] = B[t]
        K[r0:r0 + ndx, ix(t):ix(t) + ndx] = -np.eye(ndx)
        K[r0:r0 + ndx, il(t):il(t) + ndx] = -mu * np.eye(ndx)
        if t > 0:
            K[r0:r0 + ndx, ix(t - 1):ix(t - 1) + ndx] = A[t]
        rhs[r0:r0 + ndx] = -f[t]
        r0 = ix(t)
        K[r0:r0 + ndx, il(t):il(t) + ndx] = -np.eye(ndx)
        if t + 1 < H:
            K[r0:r0 + ndx, ix(t):ix(t) + ndx] = Q[t + 1]
            K[r0:r0 + ndx, iu(t + 1):iu(t + 1) + nu] = S_[t + 1]
            K[r0:r0 + ndx, il(t + 1):il(t + 1) + ndx] = A[t + 1].T
            K[r0:r0 + ndx, iv(t + 1):iv(t + 1) + nc] = Cm[t + 1].T
            rhs[r0:r0 + ndx] = -q[t + 1]
        else:
            K[r0:r0 + ndx, ix(t):ix(t) + ndx] = QN
            rhs[r0:r0 + ndx] = -qN
    assert np.abs(K - K.T).max() < 1e-12
    z = sla.solve(K, rhs)
    dx = np.stack([np.zeros(ndx)] + [z[ix(t):ix(t) + ndx] for t in range(H)])
    du = np.stack([z[iu(t):iu(t) + nu] for t in range(H)])
    dv = np.stack([z[iv(t):iv(t) + nc] for t in range(H)])
    dl = np.stack([np.zeros(ndx)] + [z[il(t):il(t) + ndx] for t in range(H)])
    return dx, du, dv, dl


def test_prox_riccati_vs_dense_kkt():
    """Riccati solution vs dense KKT solve of the whole LQ problem."""
    rng = np.random.default_rng(0)
    H, ndx, nu, nc, mu = 7, 5, 3, 2, 1e-3
    lq = _random_lq(rng, H, ndx, nu, nc, mu)
    dx, du, dv, dl, K = O.riccati(*lq, mu)
    tx, tu, tv, tl = _dense_kkt(*lq, mu)
    for a, b in ((dx, tx), (du, tu), (dv, tv), (dl, tl)):
        assert np.abs(a - b).max() < 1e-9 * max(1, np.abs(b).max())


def test_prox_riccati_small_mu_state_constraints():
    """The regime of the MPC: mu = 1e-8 and state-only equality rows (D = 0)."""
    rng = np.random.default_rng(1)
    H, ndx, nu, nc, mu = 6, 6, 4, 2, 1e-8
    Q, S_, R, q, r, A, B, f, Cm, D, d, QN, qN = _random_lq(rng, H, ndx, nu, nc, mu)
    D[:] = 0
    lq = (Q, S_, R, q, r, A, B, f, Cm, D, d, QN, qN)
    dx, du, dv, dl, K = O.riccati(*lq, mu)
    tx, tu, tv, tl = _dense_kkt(*lq, mu)
    assert np.abs(dx - tx).max() < 1e-6 * max(1, np.abs(tx).max())
    assert np.abs(du - tu).max() < 1e-6 * max(1, np.abs(tu).max())
    # constraints are (almost) closed by the step: C dx + d = mu dnu
    for t in range(1, H):
        assert np.abs(Cm[t] @ dx[t] + d[t] - mu * dv[t]).max() < 1e-9


def test_unconstrained_riccati_matches_lqr_fixed_point():
    """long-horizon time-invariant LQ -> feedback gain of the discrete ARE."""
    rng = np.random.default_rng(2)
    ndx, nu, H = 4, 2, 200
    A1 = np.eye(ndx) + 0.1 * rng.normal(size=(ndx, ndx))
    B1 = rng.normal(size=(ndx, nu))
    Q1, R1 = np.eye(ndx), 0.5 * np.eye(nu)
    P = sla.solve_discrete_are(A1, B1, Q1, R1)
    Klqr = -np.linalg.solve(R1 + B1.T @ P @ B1, B1.T @ P @ A1)
    z = lambda *s: np.zeros(s)
    rep = lambda M: np.repeat(M[None], H, 0)
    out = O.riccati(rep(Q1), z(H, ndx, nu), rep(R1), z(H, ndx), z(H, nu), rep(A1), rep(B1), z(H, ndx), z(H, 1, ndx),
                    z(H, 1, nu), z(H, 1), Q1, z(ndx), 1e-12)
    assert np.abs(out[4][0] - Klqr).max() < 1e-6


def test_cold_solve_converges_and_is_feasible():
    om, rb, K = S.make_oracle(1)
    tr = om.cold_trace()
    assert len(tr) < 30
    assert tr[-1, 1] < 1e-3  # primal infeasibility
    assert np.all(np.diff(tr[:, 0]) <= 1e-6 * np.abs(tr[:-1, 0]))  # merit never increases
    # standing solution: forces support the weight
    fz = om.us[0, 0, 2:12:3]
    assert abs(fz.sum() - rb.mass * 9.81) < 0.02 * rb.mass * 9.81


def test_iteration_is_descent_and_closes_constraints():
    om, rb, K = S.make_oracle(4, max_iters=2)
    om.generateCycleHorizon(O.trot_cycle())
    om.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, 4)
    for _ in range(5):
        om.iterate(X)
        info = om.info
        assert np.all(info[:, 1] < 0)          # dphi0 < 0
        assert np.all(info[:, 3] <= info[:, 0])  # merit decreases
        assert np.all(info[:, 2] == 1.0)       # full steps in this regime
        X = om.xs[:, 1, :].copy()
    assert np.all(info[:, 8] < 1e-3)


def test_oracle_is_deterministic_across_thread_counts():
    """SURVEY 8c item (7)."""
    outs = []
    for nt in (1, 4):
        om, rb, K = S.make_oracle(4, mpc_override={"num_threads": nt})
        om.generateCycleHorizon(O.trot_cycle())
        om.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
        X = S.random_states(rb, 4)
        om.iterate(X)
        outs.append(om.xs.copy())
    assert np.array_equal(outs[0], outs[1])


# ---- integer known-answer tests of the gait bookkeeping ----
def _biped_cycle():
    # reference tests/mpc.cpp:46-74: 10 double, 50 left-only, 10 double, 50 right-only
    return np.array([[1, 1]] * 10 + [[1, 0]] * 50 + [[1, 1]] * 10 + [[0, 1]] * 50, np.uint8)


def test_foot_timing_kat_reference_mpc_test():
    """reference tests/mpc.cpp:78-81 and :87-90 (left = foot 0, right = foot 1, H = 100)."""
    tm = O.Timer(_biped_cycle(), 100)
    assert tm.get(0, 0)[0] == 170 and tm.get(1, 0)[0] == 110
    assert tm.get(0, 1)[0] == 219 and tm.get(1, 1)[0] == 160
    for _ in range(10):
        tm.recede()
    assert tm.get(0, 0)[0] == 160 and tm.get(1, 0)[0] == 100
    assert tm.get(0, 1)[0] == 209 and tm.get(1, 1)[0] == 150


def test_foot_timing_kat_go2_trot():
    """Go2 trot, H = 50, cycle 80 = 10/30/10/30."""
    tm = O.Timer(O.trot_cycle(), 50)
    assert [tm.get(f, 0) for f in range(4)] == [[60], [100], [100], [60]]
    assert [tm.get(f, 1) for f in range(4)] == [[90], [129], [129], [90]]


def test_bezier_swing_curve_properties():
    """reference src/foot-trajectory.cpp:41-62: end points, apex at the 3/4-1/4 midpoint control point, float parameter."""
    L = O.lib()
    p0, p1 = np.array([0.1, 0.2, 0.02]), np.array([0.3, 0.25, 0.02])
    out = np.zeros(3)
    L.orc_bezier8(p0, p1, 0.15, 0.0, out)
    assert np.allclose(out, p0)
    L.orc_bezier8(p0, p1, 0.15, 1.0, out)
    assert np.allclose(out, p1)
    # Bernstein form in double with the same single-precision parameter
    from math import comb

    s = float(np.float32(7) / np.float32(30))
    mid = 0.75 * p0 + 0.25 * p1 + np.array([0, 0, 0.15])
    cps = [p0] * 4 + [mid] + [p1] * 4
    ref = sum(comb(8, i) * s**i * (1 - s) ** (8 - i) * cps[i] for i in range(9))
    L.orc_bezier8(p0, p1, 0.15, np.float32(7) / np.float32(30), out)
    assert np.abs(out - ref).max() < 1e-14


def test_mpc_reference_generation_follows_state_machine():
    om, rb, K = S.make_oracle(2)
    om.generateCycleHorizon(O.trot_cycle())
    om.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = np.stack([rb.x_ref, rb.x_ref])
    feet0 = rb.centroidal(rb.x_ref)["feet"]
    for _ in range(3):
        om.iterate(X)
    fr = om.foot_refs[0]
    # before any landing time enters the swing window nothing moves: references stay at the current feet
    assert np.abs(fr[0] - feet0).max() < 1e-12
    assert om.timing(0, 0) == [57] and om.timing(0, 1) == [87]
