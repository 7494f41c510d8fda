"""Oracle self-validation (CPU) of the first full-dynamics block (oracle/orc_full.hpp): the constrained
forward dynamics -- CRBA, RNEA, 3-D LOCAL point contacts with Baumgarte corrector, proximal iteration -- against physics
identities that go through independently validated code (the centroidal quantities of test_oracle_model.py)."""
import numpy as np
import pytest

import oracle_lib as O

G = np.array([0.0, 0.0, -9.81])


@pytest.fixture(scope="module")
def rb():
    return O.Robot("go2_like")


def _randx(rb, rng, scale=1.0):
    dx = np.concatenate([rng.normal(size=3) * 0.05, rng.normal(size=3) * 0.3, rng.normal(size=12) * 0.3,
                         rng.normal(size=6) * 0.5, rng.normal(size=12) * 1.0]) * scale
    return rb.integrate(rb.x_ref, dx)


def _quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _base_wrench_to_centroidal(rb, x, w6):
    """generalized force on the free-flyer dofs (LOCAL base frame) -> wrench about the CoM in world axes"""
    R, p = _quat_R(x[3:7]), x[:3]
    com = rb.centroidal(x)["com"]
    f = R @ w6[:3]
    return np.concatenate([f, R @ w6[3:] - np.cross(com - p, f)])


def test_joint_space_inertia_matches_the_centroidal_momentum(rb):
    rng = np.random.default_rng(0)
    for _ in range(5):
        x = _randx(rb, rng)
        r = rb.full_forward_dynamics(x, np.zeros(rb.nv - 6), 0)
        M, v = r["M"], x[rb.nq:]
        assert np.abs(M - M.T).max() == 0.0
        assert np.linalg.eigvalsh(M).min() > 1e-4
        # rows of the free-flyer = total momentum in the base frame
        hg = rb.centroidal(x)["hg"]
        assert np.abs(_base_wrench_to_centroidal(rb, x, (M @ v)[:6]) - hg).max() < 1e-11


def test_free_fall_momentum_rate_is_gravity(rb):
    rng = np.random.default_rng(1)
    for _ in range(5):
        x = _randx(rb, rng)
        tau = rng.normal(size=rb.nv - 6) * 5
        r = rb.full_forward_dynamics(x, tau, 0)
        c = rb.centroidal(x)
        hdot = c["Ag"] @ r["a"] + c["dAgv"]
        assert np.abs(hdot - np.concatenate([rb.mass * G, np.zeros(3)])).max() < 1e-9
        assert r["prox_iters"] == 0 and r["lam"].size == 0


@pytest.mark.parametrize("mask", [0b1111, 0b0110, 0b1001, 0b0001])
def test_contact_constraint_and_momentum_balance(rb, mask):
    rng = np.random.default_rng(2 + mask)
    for _ in range(4):
        x = _randx(rb, rng, 0.5)
        tau = rng.normal(size=rb.nv - 6) * 5
        r = rb.full_forward_dynamics(x, tau, mask)
        nc = 3 * bin(mask).count("1")
        assert r["J"].shape == (nc, rb.nv) and np.linalg.matrix_rank(r["J"]) == nc
        # the contact points do not accelerate (Kp = Kd = 0): the proximal iteration removes the mu-relaxation
        assert np.abs(r["J"] @ r["a"] + r["gamma"]).max() < 1e-7
        assert 1 <= r["prox_iters"] <= 10
        # Newton-Euler on the whole robot: gravity + contact forces
        c = rb.centroidal(x)
        hdot = c["Ag"] @ r["a"] + c["dAgv"]
        ext = _base_wrench_to_centroidal(rb, x, (r["J"].T @ r["lam"])[:6])
        assert np.abs(hdot - np.concatenate([rb.mass * G, np.zeros(3)]) - ext).max() < 1e-8
        # inverse dynamics of the result returns the applied generalized forces
        applied = np.concatenate([np.zeros(6), tau]) + r["J"].T @ r["lam"]
        assert np.abs(r["tau_rnea"] - applied).max() < 1e-8
        assert np.abs(r["M"] @ r["a"] + r["nle"] - applied).max() < 1e-8


def test_baumgarte_velocity_term(rb):
    rng = np.random.default_rng(7)
    x = _randx(rb, rng, 0.5)
    tau = rng.normal(size=rb.nv - 6)
    r0 = rb.full_forward_dynamics(x, tau, 0b1111)
    Kd = np.array([100.0, 50.0, 20.0])
    r1 = rb.full_forward_dynamics(x, tau, 0b1111, Kd=Kd)
    vc = (r0["J"] @ x[rb.nq:]).reshape(4, 3)  # foot velocities in their contact frames
    assert np.abs((r1["gamma"] - r0["gamma"]).reshape(4, 3) - Kd * vc).max() < 1e-11
    assert np.abs(r1["J"] @ r1["a"] + r1["gamma"]).max() < 1e-7
    # position term: pulls the foot to the origin of the universe frame (the reference's joint2 placement)
    r2 = rb.full_forward_dynamics(x, tau, 0b0001, Kp=[0, 0, 50.0])
    r3 = rb.full_forward_dynamics(x, tau, 0b0001)
    d = r2["gamma"] - r3["gamma"]
    assert abs(d[0]) < 1e-12 and abs(d[1]) < 1e-12 and abs(d[2]) > 1e-3


def test_gravity_term_is_the_gradient_of_the_potential(rb):
    rng = np.random.default_rng(3)
    x = _randx(rb, rng)
    x[rb.nq:] = 0.0
    nle = rb.full_forward_dynamics(x, np.zeros(rb.nv - 6), 0)["nle"]
    h = 1e-6
    for k in range(rb.nv):
        d = np.zeros(2 * rb.nv)
        d[k] = h
        pe = [-rb.mass * G @ rb.centroidal(rb.integrate(x, s * d))["com"] for s in (1, -1)]
        assert abs((pe[0] - pe[1]) / (2 * h) - nle[k]) < 1e-6


def test_power_balance_along_the_free_motion(rb):
    """d/dt (kinetic + potential energy) = joint torque power: pins the Coriolis part of nle (joint rows included)"""
    rng = np.random.default_rng(4)
    nv = rb.nv

    def energy(x):
        r = rb.full_forward_dynamics(x, np.zeros(nv - 6), 0)
        v = x[rb.nq:]
        return 0.5 * v @ r["M"] @ v - rb.mass * G @ rb.centroidal(x)["com"]

    for _ in range(3):
        x = _randx(rb, rng)
        tau = rng.normal(size=nv - 6) * 3
        a = rb.full_forward_dynamics(x, tau, 0)["a"]
        v = x[rb.nq:]
        h = 1e-4
        e = [energy(rb.integrate(x, np.concatenate([s * h * v, s * h * a]))) for s in (1, -1)]
        assert abs((e[0] - e[1]) / (2 * h) - tau @ v[6:]) < 1e-5 * max(1.0, abs(tau @ v[6:]))


def _fd(fun, rb, x, h=1e-6):
    """central differences along the tangent of the phase space: columns [q tangent (nv) | v (nv)]"""
    cols = []
    for k in range(2 * rb.nv):
        d = np.zeros(2 * rb.nv)
        d[k] = h
        cols.append((fun(rb.integrate(x, d)) - fun(rb.integrate(x, -d))) / (2 * h))
    return np.array(cols).T


@pytest.mark.parametrize("mask", [0b1111, 0b0110, 0])
def test_rnea_partials_against_finite_differences(rb, mask):
    rng = np.random.default_rng(11 + mask)
    x = _randx(rb, rng, 0.7)
    tau = rng.normal(size=rb.nv - 6) * 4
    r = rb.full_dynamics_derivatives(x, tau, mask)
    num = _fd(lambda xx: rb.full_rnea(xx, r["a"]), rb, x)
    scale = max(1.0, np.abs(num).max())
    assert np.abs(r["dtau_dq"] - num[:, :rb.nv]).max() < 2e-7 * scale
    assert np.abs(r["dtau_dv"] - num[:, rb.nv:]).max() < 2e-7 * scale


@pytest.mark.parametrize("mask,Kp,Kd", [(0b1111, (0, 0, 0), (0, 0, 0)), (0b1001, (0, 0, 0), (0, 0, 0)),
                                        (0b0110, (0, 0, 50.0), (100.0, 100.0, 100.0)), (0, (0, 0, 0), (0, 0, 0))])
def test_constraint_dynamics_derivatives_against_finite_differences(rb, mask, Kp, Kd):
    rng = np.random.default_rng(23 + mask)
    x = _randx(rb, rng, 0.5)
    tau = rng.normal(size=rb.nv - 6) * 4
    kw = dict(Kp=Kp, Kd=Kd, prox_accuracy=1e-14, prox_max_iter=60)
    r = rb.full_dynamics_derivatives(x, tau, mask, **kw)
    nv, nc = rb.nv, 3 * bin(mask).count("1")

    def sol(xx, tt=tau):
        o = rb.full_dynamics_derivatives(xx, tt, mask, **kw)
        return np.concatenate([o["a"], o["lam"]])

    num = _fd(sol, rb, x)
    sa, sl = max(1.0, np.abs(num[:nv]).max()), max(1.0, np.abs(num[nv:]).max() if nc else 1.0)
    assert np.abs(r["da_dq"] - num[:nv, :nv]).max() < 1e-6 * sa
    assert np.abs(r["da_dv"] - num[:nv, nv:]).max() < 1e-6 * sa
    if nc:
        assert np.abs(r["dlam_dq"] - num[nv:, :nv]).max() < 1e-6 * sl
        assert np.abs(r["dlam_dv"] - num[nv:, nv:]).max() < 1e-6 * sl
    h = 1e-5
    cols = []
    for j in range(nv - 6):
        d = np.zeros(nv - 6)
        d[j] = h
        cols.append((sol(x, tau + d) - sol(x, tau - d)) / (2 * h))
    numt = np.array(cols).T
    assert np.abs(r["da_dtau"] - numt[:nv]).max() < 1e-7 * max(1.0, np.abs(numt).max())
    if nc:
        assert np.abs(r["dlam_dtau"] - numt[nv:]).max() < 1e-7 * max(1.0, np.abs(numt).max())


# ---- the stage model of the full-dynamics OCP (oracle/orc_fulldyn.hpp) ----
@pytest.fixture(scope="module")
def full(rb):
    return O.Full(rb, O.go2_full_settings(rb))


def _refs(rb, full, rng=None):
    fref = np.tile([0.0, 0.0, rb.mass * 9.81 / 4], 4)
    u_ref = np.concatenate([np.zeros(full.nu), fref])
    feet = rb.centroidal(rb.x_ref)["feet"].copy()
    if rng is not None:
        feet = feet + rng.normal(size=feet.shape) * 0.02
    return u_ref, rb.x_ref.copy(), feet


def test_stage_structure_matches_the_reference_counts(rb, full):
    # tests/problem.cpp:48-49: the full-dynamics stage has 3 constraint blocks with torque + kinematics limits on (the third,
    # the friction cone, is off in the Go2 example): here rows = torque box (12) + joint box (12); u = 12 joint torques
    assert (full.nx, full.ndx, full.nu, full.nc) == (37, 36, 12, 24)


@pytest.mark.parametrize("mask", [0b1111, 0b0110, 0b1001])
def test_stage_derivatives_against_finite_differences(rb, full, mask):
    rng = np.random.default_rng(31 + mask)
    x = _randx(rb, rng, 0.4)
    u = rng.normal(size=full.nu) * 3
    u_ref, x_tgt, feet = _refs(rb, full, rng)
    d = full.deriv(mask, u_ref, x_tgt, feet, x, u)
    e0 = full.eval(mask, u_ref, x_tgt, feet, x, u)
    h = 1e-6
    A, B = np.zeros((36, 36)), np.zeros((36, 12))
    lx, lu = np.zeros(36), np.zeros(12)
    Cx, Cu = np.zeros((24, 36)), np.zeros((24, 12))
    for k in range(36):
        dd = np.zeros(36)
        dd[k] = h
        ep, em = (full.eval(mask, u_ref, x_tgt, feet, rb.integrate(x, s * dd), u) for s in (1, -1))
        A[:, k] = (rb.difference(e0["xnext"], ep["xnext"]) - rb.difference(e0["xnext"], em["xnext"])) / (2 * h)
        lx[k] = (ep["cost"] - em["cost"]) / (2 * h)
        Cx[:, k] = (ep["c"] - em["c"]) / (2 * h)
    for k in range(12):
        dd = np.zeros(12)
        dd[k] = h
        ep, em = (full.eval(mask, u_ref, x_tgt, feet, x, u + s * dd) for s in (1, -1))
        B[:, k] = (rb.difference(e0["xnext"], ep["xnext"]) - rb.difference(e0["xnext"], em["xnext"])) / (2 * h)
        lu[k] = (ep["cost"] - em["cost"]) / (2 * h)
        Cu[:, k] = (ep["c"] - em["c"]) / (2 * h)
    assert np.abs(d["A"] - A).max() < 2e-6 * max(1.0, np.abs(A).max())
    assert np.abs(d["B"] - B).max() < 2e-6 * max(1.0, np.abs(B).max())
    assert np.abs(d["lx"] - lx).max() < 2e-6 * max(1.0, np.abs(lx).max())
    assert np.abs(d["lu"] - lu).max() < 2e-6 * max(1.0, np.abs(lu).max())
    assert np.abs(d["Cx"] - Cx).max() < 1e-8 and np.abs(d["Cu"] - Cu).max() < 1e-8
    # Gauss-Newton Hessian: symmetric positive semi-definite
    Hm = np.block([[d["Lxx"], d["Lxu"]], [d["Lxu"].T, d["Luu"]]])
    assert np.abs(Hm - Hm.T).max() < 1e-9 * np.abs(Hm).max() and np.linalg.eigvalsh(Hm).min() > -1e-9 * np.abs(Hm).max()


def test_proxddp_converges_on_the_standing_problem(rb, full):
    """ProxDDP (orc_proxddp.hpp, templated on the stage model) on 20 standing stages from zero torques: converges to the
    static solution -- joint torques that hold the robot, contact forces m g / 4 per foot within the box limits"""
    u_ref, x_tgt, feet = _refs(rb, full)
    H = 20
    r = full.solve([0b1111] * H, u_ref, x_tgt, feet, rb.x_ref, np.zeros(full.nu), max_iter=60, tol=1e-4)
    tr = r["trace"]
    assert r["iters"] < 60 and max(tr[-1, 0], tr[-1, 1]) <= 1e-4, tr[:, :2]
    assert np.all(np.isfinite(r["xs"])) and np.all(np.isfinite(r["us"]))
    # the robot stays where it is (base position is not weighted in the example: the legs settle by a few mrad)
    D = np.array([rb.difference(rb.x_ref, x) for x in r["xs"]])
    assert np.abs(D[:, :rb.nv]).max() < 1e-2 and np.abs(D[:, rb.nv:]).max() < 0.3
    assert r["iters"] <= 10 and np.all(tr[:, 5] == 0)  # no failed line search
    eff = full.s["umax"]
    assert np.all(np.abs(r["us"]) <= eff + 1e-6)
    # the contact forces carry the weight: the momentum rate of the whole robot is small against m g
    xm, um = r["xs"][H // 2], r["us"][H // 2]
    f = rb.full_forward_dynamics(xm, um, 0b1111)
    c = rb.centroidal(xm)
    assert np.abs(c["Ag"] @ f["a"] + c["dAgv"]).max() < 0.05 * rb.mass * 9.81
    assert np.linalg.norm(f["lam"].reshape(4, 3), axis=1).min() > 0.15 * rb.mass * 9.81


def test_full_dynamics_mpc_closed_loop(rb, full):
    """The reference's MPC state machine (orc_mpc.hpp, templated on the stage model) over the full-dynamics OCP: cold solve,
    trot cycle, closed loop on its own prediction -- finite, feasible, the gait advances, torques inside their box."""
    ms = O.go2_mpc_settings(rb, max_iters=2)
    ms["T"] = 30
    B = 2
    mpc = O.OracleFullMPC(full, ms, B)
    tr = mpc.cold_trace()
    # (the default problem has identity contact poses -- src/ocp-handler.cpp:117 --, i.e. a large constant pose cost: the
    #  cold solve ends on the merit-resolution rule of DESIGN.md 4, as the kinodynamics one does)
    assert len(tr) < 100 and tr[-1][1] < 1e-3 and tr[-1][2] < 1e-2, tr[-3:]
    mpc.generateCycleHorizon(O.trot_cycle())
    mpc.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = np.tile(rb.x_ref, (B, 1))
    rng = np.random.default_rng(5)
    X[1] = rb.integrate(rb.x_ref, np.concatenate([rng.normal(size=18) * 0.01, rng.normal(size=18) * 0.05]))
    for step in range(45):
        mpc.iterate(X)
        xs, us = mpc.xs, mpc.us
        assert np.all(np.isfinite(xs)) and np.all(np.isfinite(us))
        X = xs[:, 1].copy()
    info = mpc.info
    assert np.all(info[:, 4] < 1e-2), info[:, :6]          # primal infeasibility of the last iteration
    assert np.all(np.abs(us) <= full.s["umax"] + 1e-3)
    # the swing phase has started (feet leave the ground in the predicted horizon) and the base moves forward
    assert xs[0, -1, 0] > xs[0, 0, 0] + 0.01
    assert np.abs(X[:, 2] - rb.x_ref[2]).max() < 0.05


# ---------------------------------------------------------------------------------------------------------------------------
# Talos-class robot: 6-D feet, LOCAL_WORLD_ALIGNED contacts, wrench cones (reference src/fulldynamics.cpp:56-65, 103-109,
# 163-173; settings examples/talos_fulldynamics.py:47-115; robot tests/test_utils.cpp:26-62: nq 29 / nv 28)
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tb():
    return O.Robot("talos_like")


@pytest.fixture(scope="module")
def tfull(tb):
    return O.Full(tb, O.talos_full_settings(tb))


def _trand(tb, rng, scale=1.0):
    sg = np.concatenate([np.ones(3) * 0.02, np.ones(3) * 0.05, np.ones(tb.nv - 6) * 0.1, np.ones(3) * 0.1, np.ones(3) * 0.2, np.ones(tb.nv - 6) * 0.5])
    return tb.integrate(tb.x_ref, rng.normal(size=tb.ndx) * sg * scale)


def test_talos_table_has_the_reduced_model_dimensions(tb):
    assert (tb.nq, tb.nv, tb.nf) == (29, 28, 2)  # tests/test_utils.cpp:26-62 (22 actuated joints), benchmark/talos.cpp:95-107
    feet = tb.centroidal(tb.x_ref)["feet"]
    assert np.abs(feet[:, 2]).max() < 1e-12 and abs(feet[0, 1] + feet[1, 1]) < 1e-12  # soles on the ground, symmetric


@pytest.mark.parametrize("mask", [0b11, 0b01, 0b10])
def test_6d_contact_dynamics_and_derivatives(tb, mask):
    rng = np.random.default_rng(100 + mask)
    Kp, Kd = np.array([0, 0, 50, 0, 0, 7.0]), np.array([100, 100, 100, 100, 90, 80.0])
    x, tau = _trand(tb, rng), rng.normal(size=tb.nv - 6) * 20
    r = tb.full_forward_dynamics(x, tau, mask, Kp, Kd, fs=6)
    nc = 6 * bin(mask).count("1")
    assert r["lam"].shape == (nc,)
    assert np.abs(r["J"] @ r["a"] + r["gamma"]).max() < 1e-9               # contact acceleration = corrector
    lhs = r["M"] @ r["a"] + r["nle"]
    rhs = np.concatenate([np.zeros(6), tau]) + r["J"].T @ r["lam"]
    assert np.abs(lhs - rhs).max() < 1e-9 * max(1.0, np.abs(rhs).max())     # equations of motion
    assert np.abs(r["tau_rnea"] - rhs).max() < 1e-8 * max(1.0, np.abs(rhs).max())
    d = tb.full_dynamics_derivatives(x, tau, mask, Kp, Kd, prox_accuracy=1e-14, prox_max_iter=50, fs=6)
    h = 1e-6

    def sol(xx, tt):
        rr = tb.full_dynamics_derivatives(xx, tt, mask, Kp, Kd, 1e-14, 50, fs=6)
        return np.concatenate([rr["a"], rr["lam"]])

    for name, n, off in (("q", tb.nv, 0), ("v", tb.nv, tb.nv)):
        F = np.zeros((tb.nv + nc, n))
        for k in range(n):
            dd = np.zeros(tb.ndx)
            dd[off + k] = h
            F[:, k] = (sol(tb.integrate(x, dd), tau) - sol(tb.integrate(x, -dd), tau)) / (2 * h)
        An = np.vstack([d["da_d" + name], d["dlam_d" + name]])
        assert np.abs(An - F).max() < 1e-6 * max(1.0, np.abs(F).max()), name
    F = np.zeros((tb.nv + nc, tb.nv - 6))
    for k in range(tb.nv - 6):
        dd = np.zeros(tb.nv - 6)
        dd[k] = h
        F[:, k] = (sol(x, tau + dd) - sol(x, tau - dd)) / (2 * h)
    assert np.abs(np.vstack([d["da_dtau"], d["dlam_dtau"]]) - F).max() < 1e-6 * max(1.0, np.abs(F).max())


def test_talos_stage_structure_matches_the_reference_counts(tb, tfull):
    # tests/problem.cpp:14-49: Talos stage, left foot in contact, force_cone on, land_cstr off: 6 cost components, 3 constraint
    # blocks = torque box (22) + joint box (22) + the wrench cone of the foot in contact (17)
    assert (tfull.nx, tfull.ndx, tfull.nu, tfull.nc) == (57, 56, 22, 22 + 22 + 2 * 17)
    fref = np.concatenate([[0, 0, 800.0, 0, 0, 0], np.zeros(6)])
    u_ref = np.concatenate([np.zeros(22), fref])
    feet = tb.centroidal(tb.x_ref)["feet"]
    e = tfull.eval(0b01, u_ref, tb.x_ref, feet, tb.x_ref, np.zeros(22))
    c = e["c"]
    assert np.all(c[44 + 17 :] == 0.0) and np.abs(c[44 : 44 + 17]).max() > 0  # only the left foot's cone block is present


@pytest.mark.parametrize("mask", [0b11, 0b01])
def test_talos_stage_derivatives_against_finite_differences(tb, tfull, mask):
    rng = np.random.default_rng(200 + mask)
    full, rb = tfull, tb
    x = _trand(tb, rng, 0.4)
    u = rng.normal(size=full.nu) * 10
    fref = np.tile([0.0, 0.0, rb.mass * 9.81 / 2, 0, 0, 0], 2)
    u_ref = np.concatenate([np.zeros(full.nu), fref])
    feet = rb.centroidal(rb.x_ref)["feet"] + rng.normal(size=(2, 3)) * 0.02
    d = full.deriv(mask, u_ref, rb.x_ref, feet, x, u)
    e0 = full.eval(mask, u_ref, rb.x_ref, feet, x, u)
    h = 1e-6
    n, m, k_ = full.ndx, full.nu, full.nc
    A, B, lx, lu, Cx, Cu = np.zeros((n, n)), np.zeros((n, m)), np.zeros(n), np.zeros(m), np.zeros((k_, n)), np.zeros((k_, m))
    for k in range(n):
        dd = np.zeros(n)
        dd[k] = h
        ep, em = (full.eval(mask, u_ref, rb.x_ref, feet, rb.integrate(x, s * dd), u) for s in (1, -1))
        A[:, k] = (rb.difference(e0["xnext"], ep["xnext"]) - rb.difference(e0["xnext"], em["xnext"])) / (2 * h)
        lx[k] = (ep["cost"] - em["cost"]) / (2 * h)
        Cx[:, k] = (ep["c"] - em["c"]) / (2 * h)
    for k in range(m):
        dd = np.zeros(m)
        dd[k] = h
        ep, em = (full.eval(mask, u_ref, rb.x_ref, feet, x, u + s * dd) for s in (1, -1))
        B[:, k] = (rb.difference(e0["xnext"], ep["xnext"]) - rb.difference(e0["xnext"], em["xnext"])) / (2 * h)
        lu[k] = (ep["cost"] - em["cost"]) / (2 * h)
        Cu[:, k] = (ep["c"] - em["c"]) / (2 * h)
    for nm, an, nu_ in (("A", d["A"], A), ("B", d["B"], B), ("lx", d["lx"], lx), ("lu", d["lu"], lu), ("Cx", d["Cx"], Cx), ("Cu", d["Cu"], Cu)):
        assert np.abs(an - nu_).max() < 5e-6 * max(1.0, np.abs(nu_).max()), (nm, np.abs(an - nu_).max(), np.abs(nu_).max())
    Hm = np.block([[d["Lxx"], d["Lxu"]], [d["Lxu"].T, d["Luu"]]])
    assert np.abs(Hm - Hm.T).max() < 1e-9 * np.abs(Hm).max() and np.linalg.eigvalsh(Hm).min() > -1e-9 * np.abs(Hm).max()


def test_talos_mpc_closed_loop(tb, tfull):
    """MPC state machine over the Talos full-dynamics OCP (H = 30 here): cold solve, walking cycle, closed loop on its own
    prediction: finite, feasible, torques inside their box, cone rows respected within the penalty accuracy."""
    ms = O.talos_mpc_settings(tb, max_iters=2)
    ms["T"] = 30
    ms["T_fly"], ms["T_contact"] = 20, 5
    mpc = O.OracleFullMPC(tfull, ms, 2)
    tr = mpc.cold_trace()
    assert len(tr) < 100 and np.all(np.isfinite(tr))
    mpc.generateCycleHorizon(O.walk_cycle(5, 20))
    mpc.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
    X = np.tile(tb.x_ref, (2, 1))
    rng = np.random.default_rng(5)
    X[1] = _trand(tb, rng, 0.1)
    for step in range(20):
        mpc.iterate(X)
        xs, us = mpc.xs, mpc.us
        assert np.all(np.isfinite(xs)) and np.all(np.isfinite(us))
        X = xs[:, 1].copy()
    assert np.all(np.abs(us) <= tfull.s["umax"] + 1e-2)
    assert np.abs(X[:, 2] - tb.x_ref[2]).max() < 0.08
