"""Batched whole-body inverse-dynamics QP on the device (simple-mpc_amd/csrc/smpc_id.h) against the oracle
(oracle/orc_id.hpp): the three kernels one by one -- rigid-body quantities, QP data, ADMM iterate -- and closed loops integrated like the
reference's tests (tests/inverse-dynamics/kinodynamics-id.cpp:53-60).  CPU tier: the kernel bodies compiled for the CPU; GPU tier: HIP."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O
import simple_mpc
from test_oracle_id import DT, crouch, static_forces, step

KEYS = simple_mpc.KinodynamicsID._KEYS
ALL = dict(kp_base=10.0, kp_posture=1.0, kp_contact=10.0, w_base=10.0, w_posture=0.1, w_contact_force=1e-3, w_contact_motion=1.0)


CKEYS = simple_mpc.CentroidalID._KEYS
CALL = dict(ALL, kp_com=7.0, kp_feet_tracking=5.0, w_com=10.0, w_feet_tracking=100.0)  # every task of CentroidalID on


def make(lib, B, admm_iters=100, admm_tol=-1.0, **kw):
    """(admm_tol < 0: exactly admm_iters iterations on both sides, so that the comparison does not hinge on two roundings of a residual
    falling on the same side of the stopping tolerance)"""
    rb = O.Robot("go2_like")
    s = O.id_settings(rb, DT, admm_iters=admm_iters, admm_tol=admm_tol, **kw)
    ok = O.OracleKinoID(rb, s, B)
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in S.FEET:
        mh.addPointFoot(n, "root_joint")
    if s["centroidal"]:
        gk = simple_mpc.CentroidalID(mh, DT, {k: s[k] for k in CKEYS}, s["tau_max"], s["v_max"], batch=B, lib=lib, admm_iters=admm_iters, admm_tol=admm_tol)
    else:
        gk = simple_mpc.KinodynamicsID(mh, DT, {k: s[k] for k in KEYS}, s["tau_max"], s["v_max"], batch=B, lib=lib, admm_iters=admm_iters, admm_tol=admm_tol)
    return rb, ok, gk


def _pieces(lib):
    rb, ok, gk = make(lib, 3, **ALL)
    X = S.random_states(rb, 3)
    fs = static_forces(rb)
    for k in (ok, gk):
        k.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), [True, True, False, True], fs, instance=1)  # one robot with a foot in the air
    to, ao, fo = ok.solve(X)
    tg = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
    _compare_qp(rb, ok, gk, X)
    assert S.rel_err(to, tg) < 1e-8 and S.rel_err(ao, gk.getAccelerations()) < 1e-8 and S.rel_err(fo, gk.getContactForces().reshape(3, -1)) < 1e-8
    assert np.allclose(gk.getContactForces()[1, 2], 0.0, atol=1e-6)  # the foot in the air carries nothing (to the ADMM residual)
    assert gk.resid.max() < 1e-3 and np.allclose(gk.resid, ok.resid, rtol=1e-3, atol=1e-9)


def _compare_qp(rb, ok, gk, X):
    n, m = ok.n, ok.m
    for b in range(X.shape[0]):
        Q = O.id_quantities(rb, X[b])
        for what, key in ((0, "M"), (1, "nle"), (2, "J"), (3, "Jdv"), (4, "vfoot")):
            assert S.rel_err(Q[key], gk.debug(what)[b]) < 1e-11, key
        H, g, Cm, l, u = ok.qp(b, X[b])
        assert S.rel_err(H, gk.debug(5)[b][:n, :n]) < 1e-11 and S.rel_err(g, gk.debug(6)[b][:n]) < 1e-11, (b, S.rel_err(H, gk.debug(5)[b][:n, :n]))
        assert S.rel_err(Cm, gk.debug(7)[b][:m, :n]) < 1e-11
        lg, ug = gk.debug(8)[b][:m], gk.debug(9)[b][:m]
        assert np.array_equal(np.abs(l) > 1e19, np.abs(lg) > 1e19) and np.array_equal(np.abs(u) > 1e19, np.abs(ug) > 1e19)
        fin = np.abs(l) < 1e19
        assert np.allclose(l[fin], lg[fin], rtol=1e-11, atol=1e-9) and np.allclose(u[np.abs(u) < 1e19], ug[np.abs(u) < 1e19], rtol=1e-11, atol=1e-9)


def _centroidal_pieces(lib):
    """CentroidalID: default targets, a per-robot target with a foot in the air (tracking rows), the QP data and its solution."""
    rb, ok, gk = make(lib, 3, centroidal=True, **CALL)
    X = S.random_states(rb, 3)
    # (defaults: CoM of the reference state, feet at their reference placements -- solved once before any setTarget)
    to, ao, fo = ok.solve(X)
    tg = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
    _compare_qp(rb, ok, gk, X)
    for b in range(3):
        c = rb.centroidal(X[b])
        assert S.rel_err(c["com"], gk.debug(10)[b]) < 1e-12 and S.rel_err(c["feet"].reshape(-1), gk.debug(11)[b]) < 1e-12
    assert S.rel_err(to, tg) < 1e-8 and S.rel_err(ao, gk.getAccelerations()) < 1e-8
    c = rb.centroidal(rb.x_ref)
    feet = c["feet"].copy()
    feet[2] += [0.05, -0.05, 0.05]
    contact = [True, True, False, True]
    fs = static_forces(rb, contact=contact)
    vcom, fv = np.array([0.1, 0.0, -0.05]), np.zeros((4, 3))
    fv[2] = [0.2, 0.0, 0.1]
    ok.setTargetCentroidal(c["com"] + [0.01, 0.0, 0.02], vcom, feet, fv, contact, fs, instance=1)
    gk.setTarget(c["com"] + [0.01, 0.0, 0.02], vcom, feet, np.c_[fv, np.zeros((4, 3))], contact, fs.reshape(4, 3), instance=1)  # (spatial velocities)
    to, ao, fo = ok.solve(X)
    tg = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
    _compare_qp(rb, ok, gk, X)
    assert S.rel_err(to, tg) < 1e-8 and S.rel_err(ao, gk.getAccelerations()) < 1e-8 and S.rel_err(fo, gk.getContactForces().reshape(3, -1)) < 1e-8
    assert np.allclose(gk.getContactForces()[1, 2], 0.0, atol=1e-6)
    with pytest.raises(RuntimeError, match="KinodynamicsID"):
        kk = make(lib, 1, **ALL)[2]
        kk._lib.check(kk._lib.L.smpc_id_set_target_centroidal(kk._h, -1, np.zeros(3), np.zeros(3), np.zeros(12), np.zeros(12), np.ones(4, np.uint8), np.zeros(12)))


def _centroidal_closed_loop(lib, n_steps, tol):
    """The foot-tracking scenario of tests/test_oracle_id.py (reference tests/inverse-dynamics/centroidal-id.cpp:345-405) for two robots, one
    target each through the batched setTargets."""
    kw = dict(kp_feet_tracking=5.0, kp_posture=0.1, kp_contact=1.0, w_feet_tracking=1e3, w_posture=1.0, w_contact_force=1e-3, contact_motion_equality=True,
              kp_com=7.0, w_com=10.0, kp_base=7.0, w_base=10.0)
    rb, ok, gk = make(lib, 2, centroidal=True, **kw)
    c = rb.centroidal(rb.x_ref)
    contact = [True, True, False, True]
    fs = static_forces(rb, contact=contact)
    FP = np.stack([c["feet"], c["feet"]])
    FP[0, 2] += [0.05, -0.05, 0.05]
    FP[1, 2] += [-0.03, 0.0, 0.08]
    COM = np.stack([c["com"], c["com"] + [0.0, 0.01, 0.0]])
    for b in range(2):
        ok.setTargetCentroidal(COM[b], np.zeros(3), FP[b], np.zeros((4, 3)), contact, fs, instance=b)
    gk.setTargets(COM, np.zeros((2, 3)), FP, np.zeros((2, 4, 3)), contact, np.tile(fs, (2, 1)))
    X = np.stack([rb.x_ref.copy(), rb.x_ref.copy()])
    e0 = None
    for _ in range(n_steps):
        to, ao, fo = ok.solve(X)
        tg = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
        assert S.rel_err(to, tg) < tol and S.rel_err(ao, gk.getAccelerations()) < tol
        X = np.stack([step(rb, X[b], ao[b]) for b in range(2)])
        e = [np.linalg.norm(rb.centroidal(X[b])["feet"][2] - FP[b, 2]) for b in range(2)]
        e0 = e0 or e
    assert e[0] < e0[0] and e[1] < e0[1]  # the feet in the air move towards their targets


def _closed_loop(lib, n_steps, tol, **solver):
    rb, ok, gk = make(lib, 2, **solver, **ALL)
    fs = static_forces(rb)
    for k in (ok, gk):
        k.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), [True] * 4, fs)
    X = np.stack([crouch(rb), S.random_states(rb, 1, scale=0.3)[0]])
    prev = None
    for _ in range(n_steps):
        to, ao, fo = ok.solve(X)
        tg = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
        assert S.rel_err(to, tg) < tol and S.rel_err(ao, gk.getAccelerations()) < tol
        assert np.all(np.abs(tg) <= O.GO2_EFFORT + 1e-6)
        X = np.stack([step(rb, X[b], ao[b]) for b in range(2)])
        e = np.linalg.norm(rb.difference(X[0], rb.x_ref)[: rb.nv])
        assert prev is None or e <= prev  # the whole-state error of the crouched robot decreases (reference test :293-297)
        prev = e


def test_emulated_kernels_pieces(built):
    _pieces(S.emu_lib())


def test_emulated_kernels_closed_loop(built):
    _closed_loop(S.emu_lib(), 60, 1e-7)


def test_emulated_kernels_closed_loop_default_stopping_rule(built):
    """The solver of record: at most 400 iterations, residuals checked every 20, stop below 1e-7 -- both sides agree to that tolerance (the
    weakly weighted contact forces to ~10x of it)."""
    _closed_loop(S.emu_lib(), 40, 1e-5, admm_iters=400, admm_tol=1e-7)


def test_emulated_kernels_centroidal_id_pieces(built):
    _centroidal_pieces(S.emu_lib())


def test_emulated_kernels_centroidal_id_closed_loop(built):
    _centroidal_closed_loop(S.emu_lib(), 60, 1e-7)


def _random_cases(lib, n_cases, seed):
    """Random settings (gains, weights, task switches, contact-motion variant), random contact patterns (from no foot to all four) and
    random targets, per robot: QP data and solution against the oracle."""
    rng = np.random.default_rng(seed)
    for case in range(n_cases):
        cent = bool(rng.integers(2))
        kw = dict(kp_base=rng.uniform(1, 20), kp_posture=rng.uniform(0.1, 20), kp_contact=rng.uniform(0.1, 20),
                  w_base=rng.choice([-1.0, 1.0, 50.0]), w_posture=rng.choice([0.01, 1.0]), w_contact_force=rng.choice([-1.0, 1e-3, 1.0]),
                  w_contact_motion=rng.choice([0.1, 10.0]), contact_motion_equality=bool(rng.integers(2)),
                  friction_coefficient=rng.uniform(0.3, 1.0))
        if cent:
            kw.update(centroidal=True, kp_com=rng.uniform(1, 10), kp_feet_tracking=rng.uniform(1, 100), w_com=rng.choice([-1.0, 10.0]),
                      w_feet_tracking=rng.choice([-1.0, 100.0]))
        B = 3
        rb, ok, gk = make(lib, B, **kw)
        X = S.random_states(rb, B, seed=100 + case, scale=rng.uniform(0.1, 0.6))
        for b in range(B):
            contact = [bool(c) for c in rng.integers(2, size=4)]
            f = np.zeros((4, 3))
            if any(contact):
                f[contact, 2] = rb.mass * 9.81 / sum(contact)
                f[contact, :2] = rng.normal(0, 2.0, (sum(contact), 2))
            if cent:
                c = rb.centroidal(X[b])
                com, vcom = c["com"] + rng.normal(0, 0.02, 3), rng.normal(0, 0.1, 3)
                fp, fv = c["feet"] + rng.normal(0, 0.03, (4, 3)), rng.normal(0, 0.2, (4, 3))
                ok.setTargetCentroidal(com, vcom, fp, fv, contact, f.reshape(-1), instance=b)
                gk.setTarget(com, vcom, fp, fv, contact, f, instance=b)
            else:
                xt = S.random_states(rb, 1, seed=500 + 7 * case + b, scale=0.2)[0]
                at = rng.normal(0, 1.0, rb.nv)
                ok.setTarget(xt[: rb.nq], xt[rb.nq :], at, contact, f.reshape(-1), instance=b)
                gk.setTarget(xt[: rb.nq], xt[rb.nq :], at, contact, f, instance=b)
        to, ao, fo = ok.solve(X)
        tg = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
        _compare_qp(rb, ok, gk, X)
        # (fixed 100 iterations from a cold start: not every random QP has converged, the two sides iterate identically all the same)
        assert S.rel_err(to, tg) < 1e-7 and S.rel_err(ao, gk.getAccelerations()) < 1e-7, (case, kw)
        # (residuals: equal where they are large; below 1e-5 they are the rounding of sums of multipliers of size 1e3 .. 1e4)
        assert np.all((np.abs(gk.resid - ok.resid) <= 1e-2 * ok.resid) | ((gk.resid < 1e-5) & (ok.resid < 1e-5))), (case, gk.resid, ok.resid)


def test_emulated_kernels_random_settings_contacts_targets(built):
    _random_cases(S.emu_lib(), 24, seed=7)


@pytest.mark.gpu
def test_hip_random_settings_contacts_targets(built):
    _random_cases(None, 40, seed=8)


def test_emulated_kernels_resident_solve(built):
    """smpc_id_solve_device: states and torques stay where the kernels read / write them (the emulated library's "device" memory is the
    host's); same result as the host-buffer call."""
    rb, ok, gk = make(S.emu_lib(), 3, **ALL)
    X = np.ascontiguousarray(S.random_states(rb, 3))
    tau = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :]).copy()
    rb2, _, g2 = make(S.emu_lib(), 3, **ALL)
    out = np.zeros((3, rb.nv - 6))
    g2.solve_device(X.ctypes.data, out.ctypes.data)
    g2.wait()
    assert np.array_equal(out, tau) and g2.tau_device_ptr() != 0


@pytest.mark.gpu
def test_hip_resident_solve(built):
    """Second tick from the states already in HBM (the handle's own buffer) = second tick through host buffers."""
    rb, ok, gk = make(None, 64, **ALL)
    X = np.ascontiguousarray(S.random_states(rb, 64))
    gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
    tau = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :]).copy()
    rb2, _, g2 = make(None, 64, **ALL)
    g2.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
    g2.solve_device(g2.x_device_ptr())
    g2.wait()
    g2._lib.check(g2._lib.L.smpc_id_debug_get(g2._h, 12, out := np.zeros((64, rb.nv - 6))))
    assert np.array_equal(out, tau)


def test_settings_and_errors(built):
    lib = S.emu_lib()
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in S.FEET:
        mh.addPointFoot(n, "root_joint")
    with pytest.raises(KeyError):
        simple_mpc.KinodynamicsID(mh, DT, {"kp_bas": 1.0}, O.GO2_EFFORT, O.GO2_VMAX, lib=lib)
    with pytest.raises(RuntimeError, match="nv - 6"):
        simple_mpc.KinodynamicsID(mh, DT, {}, O.GO2_EFFORT[:5], O.GO2_VMAX, lib=lib)
    with pytest.raises(RuntimeError, match="control_dt"):
        simple_mpc.KinodynamicsID(mh, 0.0, {}, O.GO2_EFFORT, O.GO2_VMAX, lib=lib)
    k = simple_mpc.KinodynamicsID(mh, DT, {"kp_posture": 20.0, "w_posture": 1.0}, O.GO2_EFFORT, O.GO2_VMAX, lib=lib)
    x = mh.getReferenceState()
    k.setTarget(x[: mh.nq], np.zeros(mh.nv), np.zeros(mh.nv), [False] * 4, [])  # no contact, no forces: the reference's posture test
    tau = np.zeros(mh.nv - 6)
    assert k.solve(0.0, x[: mh.nq], x[mh.nq :], tau) is not None and np.all(np.isfinite(tau))
    ddq = np.zeros(mh.nv)
    k.getAccelerations(ddq)
    assert abs(ddq[2] + 9.81) < 1e-6  # free fall of the base, posture held
    # the device-resident legs refuse what they cannot serve
    from simple_mpc import presets as P

    cocp = simple_mpc.CentroidalOCP(P.go2_centroidal_settings(mh), mh)
    cocp.createProblem(np.zeros(9), 10, 3, -9.81, False)
    conf = dict({kk: v for kk, v in P.go2_mpc_settings(mh, max_iters=1).items() if kk in P.MPC_KEYS}, T_fly=6, T_contact=2)
    cmpc = simple_mpc.BatchedMPC(conf, cocp, 1, lib=lib)
    with pytest.raises(RuntimeError, match="CentroidalID"):
        k.setTargetsFromMPC(cmpc, 0.0)
    with pytest.raises(RuntimeError, match="kinodynamics or a full-dynamics handle"):
        cmpc.simStepDevice(x.ctypes.data, tau.ctypes.data, [True] * 4, 1e-3)
    kocp = simple_mpc.KinodynamicsOCP(P.go2_kino_settings(mh), mh)
    kocp.createProblem(mh.getReferenceState(), 10, 3, -9.81, False)
    kmpc = simple_mpc.BatchedMPC(conf, kocp, 2, lib=lib)
    with pytest.raises(RuntimeError, match="same batch"):
        k.setTargetsFromMPC(kmpc, 0.0)
    with pytest.raises(RuntimeError, match="dt must be positive"):
        kmpc.simStepDevice(x.ctypes.data, tau.ctypes.data, [True] * 4, 0.0)


@pytest.mark.gpu
def test_hip_pieces(built):
    _pieces(None)


@pytest.mark.gpu
def test_hip_closed_loop(built):
    _closed_loop(None, 200, 1e-7)


@pytest.mark.gpu
def test_hip_closed_loop_default_stopping_rule(built):
    _closed_loop(None, 200, 1e-5, admm_iters=400, admm_tol=1e-7)


@pytest.mark.gpu
def test_hip_centroidal_id_pieces(built):
    _centroidal_pieces(None)


@pytest.mark.gpu
def test_hip_centroidal_id_closed_loop(built):
    _centroidal_closed_loop(None, 400, 1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("centroidal", [False, True])
def test_hip_batch_of_4096_robots(built, centroidal):
    """4096 robots, 16 distinct states replicated: replicas bit-identical, the distinct ones follow the oracle, limits respected."""
    B, nd = 4096, 16
    kw = dict(CALL, centroidal=True) if centroidal else ALL
    rb, ok, gk = make(None, B, **kw)
    ok = O.OracleKinoID(rb, O.id_settings(rb, DT, admm_iters=100, admm_tol=-1.0, **kw), nd)
    fs = static_forces(rb)
    if centroidal:
        c = rb.centroidal(rb.x_ref)
        ok.setTargetCentroidal(c["com"], np.zeros(3), c["feet"], np.zeros((4, 3)), [True] * 4, fs)
        gk.setTarget(c["com"], np.zeros(3), c["feet"], np.zeros((4, 3)), [True] * 4, fs.reshape(4, 3))
    else:
        for k in (ok, gk):
            k.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), [True] * 4, fs)
    Xo = S.random_states(rb, nd, seed=3, scale=0.5)
    for _ in range(3):
        X = np.tile(Xo, (B // nd, 1))
        tg = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :]).reshape(B // nd, nd, -1)
        to, ao, fo = ok.solve(Xo)
        assert np.abs(tg - tg[0:1]).max() == 0.0 and S.rel_err(to, tg[0]) < 1e-7
        assert np.all(np.abs(tg) <= O.GO2_EFFORT + 1e-6)
        Xo = np.stack([step(rb, Xo[b], ao[b]) for b in range(nd)])


def _full_stack(lib, B, mpc_steps):
    """MPC at 100 Hz -> interpolated targets -> KinodynamicsID at 1 kHz -> constrained forward dynamics as the simulator: the loop of the
    reference's examples/go2_kinodynamics.py:216-300 (examples/go2_mpc_id_batched.py is the same for a larger batch)."""
    from simple_mpc import presets as P

    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in P.GO2_FEET:
        mh.addPointFoot(n, "root_joint")
    nq, nv = mh.nq, mh.nv
    ocp = simple_mpc.KinodynamicsOCP(P.go2_kino_settings(mh), mh)
    ocp.createProblem(mh.getReferenceState(), 50, 3, -9.81, False)
    mpc = simple_mpc.BatchedMPC({k: v for k, v in P.go2_mpc_settings(mh, max_iters=1).items() if k in P.MPC_KEYS}, ocp, B, lib=lib)
    mpc.generateCycleHorizon(P.trot_cycle())
    V = np.zeros((B, 6))
    V[:, 0] = np.linspace(0.0, 0.3, B)
    mpc.switchToWalk(V[0])
    mpc.setVelocityBaseBatched(V)
    ids = dict(kp_base=7.0, kp_posture=10.0, kp_contact=10.0, w_base=100.0, w_posture=1.0, w_contact_force=1.0, w_contact_motion=1.0)
    kid = simple_mpc.KinodynamicsID(mh, 1e-3, ids, O.GO2_EFFORT, O.GO2_VMAX, batch=B, lib=lib)
    X = np.tile(mh.getReferenceState(), (B, 1))
    swing = False
    for step_i in range(mpc_steps):
        mpc.iterate(X)
        contact = mpc.ocp_handler.getContactState(0)
        swing = swing or not all(contact)
        mask = np.full(B, sum(1 << i for i, c in enumerate(contact) if c), np.uint32)
        for sub in range(10):
            x_i, a_i, f_i = mpc.interpolate(sub / 10.0 * 0.01)
            kid.setTargets(x_i[:, :nq], x_i[:, nq:], a_i, contact, f_i)
            tau = kid.solve(0.0, X[:, :nq], X[:, nq:])
            assert np.all(np.abs(tau) <= O.GO2_EFFORT + 1e-6) and kid.resid.max() < 1e-3
            a = mpc.constraintDynamics(X, tau, mask, Kp=[0.0, 0.0, 0.0], Kd=[50.0, 50.0, 50.0])["a"]
            vn = X[:, nq:] + a * 1e-3
            X = np.stack([P.integrate(np.r_[X[b, :nq], vn[b]], np.r_[vn[b] * 1e-3, np.zeros(nv)], nq) for b in range(B)])
        assert np.all(np.isfinite(X)) and np.all(np.abs(X[:, 2] - mh.getReferenceState()[2]) < 0.05) and np.abs(X[:, nq:]).max() < 5.0
    return X, swing


def _resident_stack(lib, B, mpc_steps, alloc):
    """The loop of _full_stack with nothing crossing the host between the MPC steps: targets written by the MPC's interpolation kernel
    (setTargetsFromMPC), states and torques resident (solve_device, simStepDevice) -- against the same loop through host buffers."""
    from simple_mpc import presets as P

    def setup():
        mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
        for n in P.GO2_FEET:
            mh.addPointFoot(n, "root_joint")
        ocp = simple_mpc.KinodynamicsOCP(P.go2_kino_settings(mh), mh)
        ocp.createProblem(mh.getReferenceState(), 10, 3, -9.81, False)  # (a short horizon and cycle: swing phases reach stage 0 within the test)
        conf = dict({k: v for k, v in P.go2_mpc_settings(mh, max_iters=1).items() if k in P.MPC_KEYS}, T_fly=6, T_contact=2)
        mpc = simple_mpc.BatchedMPC(conf, ocp, B, lib=lib)
        mpc.generateCycleHorizon(P.trot_cycle(2, 6))
        V = np.zeros((B, 6))
        V[:, 0] = np.linspace(0.0, 0.3, B)
        mpc.switchToWalk(V[0])
        mpc.setVelocityBaseBatched(V)
        ids = dict(kp_base=7.0, kp_posture=10.0, kp_contact=10.0, w_base=100.0, w_posture=1.0, w_contact_force=1.0, w_contact_motion=1.0)
        kid = simple_mpc.KinodynamicsID(mh, 1e-3, ids, O.GO2_EFFORT, O.GO2_VMAX, batch=B, lib=lib, admm_iters=100, admm_tol=-1.0)
        return mh, mpc, kid

    mh, mpc, kid = setup()
    nq, nv = mh.nq, mh.nv
    X = np.tile(mh.getReferenceState(), (B, 1))
    swing = False
    for _ in range(mpc_steps):  # host buffers
        mpc.iterate(X)
        contact = mpc.ocp_handler.getContactState(0)
        swing = swing or not all(contact)
        mask = np.full(B, sum(1 << i for i, c in enumerate(contact) if c), np.uint32)
        for sub in range(10):
            x_i, a_i, f_i = mpc.interpolate(sub / 10.0 * 0.01)
            kid.setTargets(x_i[:, :nq], x_i[:, nq:], a_i, contact, f_i)
            tau = kid.solve(0.0, X[:, :nq], X[:, nq:])
            a = mpc.constraintDynamics(X, tau, mask, Kp=[0.0, 0.0, 0.0], Kd=[50.0, 50.0, 50.0])["a"]
            vn = X[:, nq:] + a * 1e-3
            X = np.stack([P.integrate(np.r_[X[b, :nq], vn[b]], np.r_[vn[b] * 1e-3, np.zeros(nv)], nq) for b in range(B)])
    assert swing
    for shared in (False, True):  # the controller on its own stream (events + waits between the legs), then on the MPC's stream (one queue)
        mh, mpc, kid = setup()
        if shared:
            kid.shareStream(mpc)
        Xd = alloc(np.tile(mh.getReferenceState(), (B, 1)))  # resident
        for _ in range(mpc_steps):
            mpc.iterate_device(Xd.ptr)
            mpc.wait()
            contact = mpc.ocp_handler.getContactState(0)
            for sub in range(10):
                kid.setTargetsFromMPC(mpc, sub / 10.0 * 0.01)
                kid.solve_device(Xd.ptr)
                if not shared:
                    kid.wait()
                mpc.simStepDevice(Xd.ptr, kid.tau_device_ptr(), contact, 1e-3, Kp=[0.0, 0.0, 0.0], Kd=[50.0, 50.0, 50.0])
            mpc.wait()
        assert S.rel_err(X, Xd.get()) < 1e-8, (shared, S.rel_err(X, Xd.get()))
        if shared:
            kid.shareStream(None)


class _HostArray:
    """(the emulated library's "device" memory is the host's)"""

    def __init__(self, a):
        self.a = np.ascontiguousarray(a)
        self.ptr = self.a.ctypes.data

    def get(self):
        return self.a


def test_emulated_kernels_resident_stack(built):
    _resident_stack(S.emu_lib(), 2, 16, _HostArray)


@pytest.mark.gpu
def test_hip_resident_stack(built):
    import torch

    class Dev:
        def __init__(self, a):
            self.t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            torch.cuda.synchronize()
            self.ptr = self.t.data_ptr()

        def get(self):
            return self.t.cpu().numpy()

    _resident_stack(None, 16, 16, Dev)


@pytest.mark.gpu
def test_hip_resident_stack_three_gait_cycles(built):
    """256 robots, 2.5 s of simulated time (250 MPC steps = three trot cycles, 2500 controller ticks and simulator steps per robot) with the
    whole stack resident: every robot stays up, follows its velocity command, the torques respect the limits, and a second run from the
    same state reproduces the first bit for bit."""
    import torch
    from simple_mpc import presets as P

    B, steps = 256, 250
    runs = []
    for _ in range(2):
        mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like"), "standing", "root_joint")
        for n in P.GO2_FEET:
            mh.addPointFoot(n, "root_joint")
        ocp = simple_mpc.KinodynamicsOCP(P.go2_kino_settings(mh), mh)
        ocp.createProblem(mh.getReferenceState(), 50, 3, -9.81, False)
        mpc = simple_mpc.BatchedMPC({k: v for k, v in P.go2_mpc_settings(mh, max_iters=1).items() if k in P.MPC_KEYS}, ocp, B)
        mpc.generateCycleHorizon(P.trot_cycle())
        V = np.zeros((B, 6))
        V[:, 0] = np.linspace(0.0, 0.3, B)
        mpc.switchToWalk(V[0])
        mpc.setVelocityBaseBatched(V)
        ids = dict(kp_base=7.0, kp_posture=10.0, kp_contact=10.0, w_base=100.0, w_posture=1.0, w_contact_force=1.0, w_contact_motion=1.0)
        kid = simple_mpc.KinodynamicsID(mh, 1e-3, ids, O.GO2_EFFORT, O.GO2_VMAX, batch=B)
        kid.shareStream(mpc)
        X = torch.from_numpy(np.tile(mh.getReferenceState(), (B, 1))).cuda()
        torch.cuda.synchronize()
        tmax = 0.0
        for step_i in range(steps):
            mpc.iterate_device(X.data_ptr())
            mpc.wait()
            contact = mpc.ocp_handler.getContactState(0)
            for sub in range(10):
                kid.setTargetsFromMPC(mpc, sub * 1e-3)
                kid.solve_device(X.data_ptr())
                mpc.simStepDevice(X.data_ptr(), kid.tau_device_ptr(), contact, 1e-3, Kp=[0.0, 0.0, 0.0], Kd=[50.0, 50.0, 50.0])
            if step_i % 10 == 0:
                tmax = max(tmax, np.abs(kid.debug(12)).max())
        mpc.wait()
        runs.append((X.cpu().numpy(), tmax))
        kid.shareStream(None)  # (the MPC handle may go first)
    Xh, tmax = runs[0]
    z0 = mh.getReferenceState()[2]
    print("base x %.3f .. %.3f m after 2.5 s, height %.3f .. %.3f (reference %.3f), max |tau| %.1f" % (Xh[0, 0], Xh[-1, 0], Xh[:, 2].min(), Xh[:, 2].max(), z0, tmax))
    assert np.all(np.isfinite(Xh)) and np.all(np.abs(Xh[:, 2] - z0) < 0.05)
    assert Xh[-1, 0] > 0.25 and abs(Xh[0, 0]) < 0.1 and np.all(np.diff(Xh[:, 0]) > -0.02)  # (the first swing reaches stage 0 after 0.6 s); the commands order the robots
    assert tmax <= O.GO2_EFFORT.max() + 1e-6
    assert np.array_equal(Xh, runs[1][0])


def _centroidal_resident_targets(lib, B):
    """setTargetsFromMPC of a centroidal MPC = setTargets of its interpolated solution, foot references and contact flags."""
    from simple_mpc import presets as P

    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in P.GO2_FEET:
        mh.addPointFoot(n, "root_joint")
    conf = dict({k: v for k, v in P.go2_mpc_settings(mh, max_iters=1).items() if k in P.MPC_KEYS}, T_fly=6, T_contact=2)
    ocp = simple_mpc.CentroidalOCP(P.go2_centroidal_settings(mh), mh)
    ocp.createProblem(np.zeros(9), 10, 3, -9.81, False)
    mpc = simple_mpc.BatchedMPC(conf, ocp, B, lib=lib)
    mpc.generateCycleHorizon(P.trot_cycle(2, 6))
    V = np.zeros((B, 6))
    V[:, 0] = np.linspace(0.1, 0.3, B)
    mpc.switchToWalk(V[0])
    mpc.setVelocityBaseBatched(V)
    X = P.random_states(mh, B, scale=0.3)
    for _ in range(14):  # (swing phase at stage 0)
        mpc.iterate(X)
    contact = mpc.ocp_handler.getContactState(0)
    assert not all(contact)
    ids = dict(kp_base=7.0, kp_com=7.0, kp_posture=10.0, kp_contact=10.0, kp_feet_tracking=100.0, w_base=50.0, w_com=100.0, w_posture=1.0,
               w_contact_force=1e-3, w_contact_motion=1.0, w_feet_tracking=10.0)
    mk = lambda: simple_mpc.CentroidalID(mh, 1e-3, ids, O.GO2_EFFORT, O.GO2_VMAX, batch=B, lib=lib, admm_iters=100, admm_tol=-1.0)
    ka, kb = mk(), mk()
    d = 0.4
    x_i, _, f_i = mpc.interpolate(d * 0.01)
    refs = mpc.getReferencePoses()
    ka.setTargets(x_i[:, :3], x_i[:, 3:6] / mh.getMass(), (1 - d) * refs[:, 0] + d * refs[:, 1], (refs[:, 1] - refs[:, 0]) / 0.01, contact, f_i)
    kb.setTargetsFromMPC(mpc, d * 0.01)
    ta = ka.solve(0.0, X[:, : mh.nq], X[:, mh.nq :])
    tb = kb.solve(0.0, X[:, : mh.nq], X[:, mh.nq :])
    assert S.rel_err(ta, tb) < 1e-12 and S.rel_err(ka.debug(6), kb.debug(6)) < 1e-12
    with pytest.raises(RuntimeError, match="CentroidalID"):
        simple_mpc.KinodynamicsID(mh, 1e-3, {}, O.GO2_EFFORT, O.GO2_VMAX, batch=B, lib=lib).setTargetsFromMPC(mpc, 0.0)


def test_emulated_kernels_centroidal_resident_targets(built):
    _centroidal_resident_targets(S.emu_lib(), 3)


@pytest.mark.gpu
def test_hip_centroidal_resident_targets(built):
    _centroidal_resident_targets(None, 8)


def _centroidal_stack(lib, B, mpc_steps):
    """Centroidal MPC at 100 Hz -> interpolated CoM / force targets and foot references -> CentroidalID at 1 kHz -> constrained forward
    dynamics as the simulator (the loop of the reference's examples/talos_centroidal.py:200-246 on the point-foot quadruped;
    examples/go2_centroidal_id_batched.py is the same for a larger batch)."""
    from simple_mpc import presets as P

    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in P.GO2_FEET:
        mh.addPointFoot(n, "root_joint")
    nq, nv, mass = mh.nq, mh.nv, mh.getMass()
    conf = {k: v for k, v in P.go2_mpc_settings(mh, max_iters=1).items() if k in P.MPC_KEYS}
    ocp = simple_mpc.CentroidalOCP(P.go2_centroidal_settings(mh), mh)
    ocp.createProblem(np.zeros(9), 50, 3, -9.81, False)
    mpc = simple_mpc.BatchedMPC(conf, ocp, B, lib=lib)
    mpc.generateCycleHorizon(P.trot_cycle())
    V = np.zeros((B, 6))
    V[:, 0] = np.linspace(0.0, 0.2, B)
    mpc.switchToWalk(V[0])
    mpc.setVelocityBaseBatched(V)
    kocp = simple_mpc.KinodynamicsOCP(P.go2_kino_settings(mh), mh)  # (a kinodynamics handle lends its forward-dynamics kernel as the simulator)
    kocp.createProblem(mh.getReferenceState(), 50, 3, -9.81, False)
    sim = simple_mpc.BatchedMPC(conf, kocp, B, lib=lib)
    ids = dict(kp_base=7.0, kp_com=7.0, kp_posture=10.0, kp_contact=10.0, kp_feet_tracking=2000.0, w_base=50.0, w_com=100.0, w_posture=1.0,
               w_contact_force=1e-6, w_contact_motion=1e-3, w_feet_tracking=100.0)  # examples/talos_centroidal.py:127-136 (+ foot tracking)
    cid = simple_mpc.CentroidalID(mh, 1e-3, ids, O.GO2_EFFORT, O.GO2_VMAX, batch=B, lib=lib)
    X = np.tile(mh.getReferenceState(), (B, 1))
    swing, lift = False, 0.0
    for step_i in range(mpc_steps):
        mpc.iterate(X)
        contact = mpc.ocp_handler.getContactState(0)
        swing = swing or not all(contact)
        mask = np.full(B, sum(1 << i for i, c in enumerate(contact) if c), np.uint32)
        refs = mpc.getReferencePoses()
        for sub in range(10):
            d = sub / 10.0
            x_i, _, f_i = mpc.interpolate(d * 0.01)
            cid.setTargets(x_i[:, :3], x_i[:, 3:6] / mass, (1 - d) * refs[:, 0] + d * refs[:, 1], (refs[:, 1] - refs[:, 0]) / 0.01, contact, f_i)
            tau = cid.solve(0.0, X[:, :nq], X[:, nq:])
            assert np.all(np.abs(tau) <= O.GO2_EFFORT + 1e-6) and cid.resid.max() < 1e-3
            a = sim.constraintDynamics(X, tau, mask, Kp=[0.0, 0.0, 0.0], Kd=[50.0, 50.0, 50.0])["a"]
            vn = X[:, nq:] + a * 1e-3
            X = np.stack([P.integrate(np.r_[X[b, :nq], vn[b]], np.r_[vn[b] * 1e-3, np.zeros(nv)], nq) for b in range(B)])
        if not all(contact):
            k = contact.index(False)
            lift = max(lift, max(cid.debug(11)[b][3 * k + 2] for b in range(B)))
        assert np.all(np.isfinite(X)) and np.all(np.abs(X[:, 2] - mh.getReferenceState()[2]) < 0.05) and np.abs(X[:, nq:]).max() < 5.0
    return X, swing, lift


def test_emulated_kernels_centroidal_stack(built):
    X, swing, lift = _centroidal_stack(S.emu_lib(), 2, 92)
    assert swing and lift > 0.05, "the feet in the air must follow their swing references"
    assert X[1, 0] > 0.01 > abs(X[0, 0]) * 0  # the robot with the forward command advances


def test_emulated_kernels_full_stack(built):
    _full_stack(S.emu_lib(), 2, 5)


@pytest.mark.gpu
def test_hip_full_stack_through_a_swing_phase(built):
    X, swing = _full_stack(None, 4, 75)  # the first take-off reaches stage 0 at control step 60
    assert swing and X[-1, 0] > X[0, 0]  # the robot commanded forwards is ahead of the one commanded to stay


# ---- reference-fidelity switches, failed solves (round 3) ----
def _switches(lib, tol):
    rng = np.random.default_rng(8)
    for flags in (dict(base_reference_as_coded=True), dict(tsid_joint_bounds=True), dict(base_reference_as_coded=True, tsid_joint_bounds=True)):
        rb = O.Robot("go2_like")
        s = O.id_settings(rb, DT, admm_iters=100, admm_tol=-1.0, **ALL, **flags)
        ok = O.OracleKinoID(rb, s, 3)
        mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
        for n in S.FEET:
            mh.addPointFoot(n, "root_joint")
        gk = simple_mpc.KinodynamicsID(mh, DT, {k: s[k] for k in KEYS}, s["tau_max"], s["v_max"], batch=3, lib=lib, admm_iters=100, admm_tol=-1.0, **flags)
        X = S.random_states(rb, 3, seed=21)
        X[1, 7 + 2] = s["q_max"][2] - 1e-3  # a joint about to hit its limit
        X[1, rb.nq + 6 + 2] = 2.0
        fs = static_forces(rb)
        vt, at = rng.normal(size=rb.nv) * 0.2, rng.normal(size=rb.nv)
        for k in (ok, gk):
            k.setTarget(rb.x_ref[: rb.nq], vt, at, [True] * 4, fs)
        to, ao, fo = ok.solve(X)
        tg = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
        _compare_qp(rb, ok, gk, X)
        assert S.rel_err(to, tg) < tol and S.rel_err(ao, gk.getAccelerations()) < tol, flags


def _failed_solve_recovers(lib):
    """a NaN state for one tick must not disable that robot's controller: the warm start of a failed solve is dropped"""
    rb, ok, gk = make(lib, 3, admm_iters=60, **ALL)
    fs = static_forces(rb)
    gk.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), [True] * 4, fs)
    X = S.random_states(rb, 3, scale=0.3)
    t0 = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :]).copy()
    Xbad = X.copy()
    Xbad[1, 8] = np.nan
    tb = gk.solve(0.0, Xbad[:, : rb.nq], Xbad[:, rb.nq :])
    assert not np.isfinite(tb[1]).all() and np.isfinite(tb[0]).all() and np.isfinite(tb[2]).all()
    assert not np.isfinite(gk.getResiduals()[1])
    t1 = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
    assert np.isfinite(t1).all()
    assert S.rel_err(t0[1], t1[1]) < 1e-6  # (robot 1 restarted from scratch, exactly like its first solve)
    gk.reset()  # and the explicit reset: every robot from scratch again
    t2 = gk.solve(0.0, X[:, : rb.nq], X[:, rb.nq :])
    assert S.rel_err(t0, t2) < 1e-12


def test_emulated_kernels_reference_switches(built):
    _switches(S.emu_lib(), 1e-8)


def test_emulated_kernels_failed_solve_recovers(built):
    _failed_solve_recovers(S.emu_lib())


@pytest.mark.gpu
def test_hip_reference_switches(built):
    _switches(None, 1e-8)


@pytest.mark.gpu
def test_hip_failed_solve_recovers(built):
    _failed_solve_recovers(None)
