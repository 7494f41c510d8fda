"""createProblem(x0, T, force_size, gravity, terminal_constraint = True): the DCM equality com + tau vcom = com_ref at the terminal
node (reference src/ocp-handler.cpp:133-136, src/kinodynamics.cpp:366-388, src/fulldynamics.cpp:433-455; MPC::updateStepTrackerReferences
moves com_ref every control step, src/mpc.cpp:313-323).  The reference's own MPC tests and its Talos benchmark build their problems this
way (tests/mpc.cpp:25,110, benchmark/talos.cpp:140).  CPU tier: the oracle's residual against finite differences and its closed loop;
the kernel bodies (CPU build) against the oracle.  GPU tier: the HIP library against the oracle at the sizes of record."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

TC = {"terminal_constraint": True}


def test_oracle_dcm_residual_matches_finite_differences():
    rb = O.Robot("go2_like")
    K = O.Kino(rb, O.go2_kino_settings(rb))
    X = S.random_states(rb, 1)[0]
    ref, tau = np.array([0.1, 0.02, 0.3]), 0.18
    c, Cm = K.term_cstr(X, ref, tau)
    eps, J = 1e-6, np.zeros((3, K.ndx))
    for i in range(K.ndx):
        d = np.zeros(K.ndx)
        d[i] = eps
        J[:, i] = (K.term_cstr(rb.integrate(X, d), ref, tau)[0] - K.term_cstr(rb.integrate(X, -d), ref, tau)[0]) / (2 * eps)
    assert np.abs(J - Cm).max() < 1e-8
    # the residual is affine in the reference and the DCM of a robot at rest is its centre of mass
    assert np.allclose(K.term_cstr(X, ref + 1.0, tau)[0], c - 1.0)
    X0 = X.copy()
    X0[rb.nq:] = 0.0
    assert np.allclose(K.term_cstr(X0, np.zeros(3), tau)[0], K.term_cstr(X0, np.zeros(3), 5 * tau)[0])


def test_oracle_closed_loop_meets_the_terminal_constraint():
    om, rb, K = S.make_oracle(2, max_iters=3, horizon=50, mpc_override=TC)
    om0, _, _ = S.make_oracle(2, max_iters=3, horizon=50)
    v, ref, tau = om.terminal
    assert np.allclose(tau, np.sqrt(rb.x_ref[2] / 9.81)) and np.allclose(ref[0], rb.x_ref[:3])  # createTerminalConstraint(x0.head<3>())
    for m in (om, om0):
        m.generateCycleHorizon(O.trot_cycle())
        m.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, 2)
    for _ in range(10):
        om.iterate(X)
        om0.iterate(X)
        X = om.xs[:, 1, :].copy()
    v, ref, tau = om.terminal
    feet = om.foot_refs[:, -1]  # last foot references [B][nf][3]
    com0 = K.term_cstr(rb.x_ref, np.zeros(3), 0.0)[0]  # CoM of the reference state
    assert np.allclose(ref[:, :2], feet.mean(axis=1)[:, :2]) and np.allclose(ref[:, 2], feet.mean(axis=1)[:, 2] + com0[2])
    for b in range(2):
        c, _ = K.term_cstr(om.xs[b, -1], ref[b], tau[b])
        assert np.abs(c).max() < 1e-5
        c0, _ = K.term_cstr(om0.xs[b, -1], ref[b], tau[b])
        assert np.abs(c0).max() > 1e-3  # without the constraint the plan ends elsewhere
    assert np.abs(v).max() > 1.0 and np.all(np.isfinite(om.xs))


def _loop(om, gm, X, n, tol, tol_later=None):
    """tol_later: the gate of the steps after the first where the first step alone needs a wider one (a random state thrown onto a plan that
    ends in 1 / mu-weighted rows amplifies rounding once; a regression of the solves would show in every step)"""
    for i in range(n):
        om.iterate(X)
        gm.iterate(X)
        t = tol if (i == 0 or tol_later is None) else tol_later
        assert S.rel_err(om.xs, gm.xs) < t and S.rel_err(om.us, gm.us) < 1e3 * t, (i, S.rel_err(om.xs, gm.xs), S.rel_err(om.us, gm.us))
        assert S.alphas_agree(om, gm, rtol=1e-8)
        X = om.xs[:, 1, :].copy()


def test_emulated_kernels_kinodynamics(built):
    om, gm, rb = S.make_pair(2, max_iters=2, lib=S.emu_lib(), horizon=20, mpc_override=TC)
    _loop(om, gm, S.random_states(rb, 2), 6, 1e-8)


def test_emulated_kernels_go2_fulldynamics(built):
    om, gm, rb = S.make_full_pair(2, max_iters=2, lib=S.emu_lib(), horizon=16, mpc_override=TC)
    _loop(om, gm, S.random_states(rb, 2), 5, 1e-9)


def test_emulated_kernels_talos_fulldynamics(built):
    om, gm, rb = S.make_talos_pair(1, max_iters=2, lib=S.emu_lib(), horizon=20, mpc_override=TC)
    _loop(om, gm, S.talos_random_states(rb, 1, scale=0.7), 4, 1e-9)


def test_checkpoint_keeps_the_terminal_multipliers(built):
    gm, rb, _, _ = S.make_product(2, 2, lib=S.emu_lib(), horizon=20, mpc_override=TC)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    Xs = [S.random_states(rb, 2, seed=s) for s in range(5)]
    for k in range(2):
        gm.iterate(Xs[k])
    blob = gm.save_state()
    ref = []
    for k in range(2, 5):
        gm.iterate(Xs[k])
        ref.append(gm.xs.copy())
    g2, _, _, _ = S.make_product(2, 2, lib=S.emu_lib(), horizon=20, mpc_override=TC)
    g2.load_state(blob)
    for k in range(2, 5):
        g2.iterate(Xs[k])
        assert np.array_equal(g2.xs, ref[k - 2])
    g3, _, _, _ = S.make_product(2, 2, lib=S.emu_lib(), horizon=20)
    with pytest.raises(RuntimeError, match="does not match"):
        g3.load_state(blob)  # a problem without the constraint


def test_centroidal_problem_accepts_the_flag_like_the_reference(built):
    # CentroidalOCP::createTerminalConstraint leaves the constraint out (src/centroidal-dynamics.cpp:318-328)
    import simple_mpc

    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", S.emu_lib()), "standing", "root_joint")
    for n in S.FEET:
        mh.addPointFoot(n, "root_joint")
    rb = O.Robot("go2_like")
    ocp = simple_mpc.CentroidalOCP(O.go2_centroidal_settings(rb), mh)
    ocp.createProblem(np.zeros(9), 20, 3, -9.81, True)
    assert ocp.getSize() == 20


@pytest.mark.gpu
@pytest.mark.parametrize("iters", [1, 3])
def test_hip_kinodynamics_terminal_constraint(built, iters):
    om, gm, rb = S.make_pair(3, max_iters=iters, mpc_override=TC)
    _loop(om, gm, S.random_states(rb, 3), 15, 1e-7, tol_later=1e-8)


@pytest.mark.gpu
def test_hip_go2_fulldynamics_terminal_constraint(built):
    om, gm, rb = S.make_full_pair(2, max_iters=2, mpc_override=TC)
    # (1e-7: the first step, from a random state onto a plan that ends in the 1 / mu-weighted terminal rows, amplifies rounding to 3e-8 -- the order
    # in which the derivative solves accumulate decides the last digits; from the second step on the two sides agree to 1e-11)
    _loop(om, gm, S.random_states(rb, 2), 12, 1e-7, tol_later=1e-8)


@pytest.mark.gpu
def test_hip_talos_fulldynamics_terminal_constraint(built):
    # benchmark/talos.cpp:140 builds its problem with the terminal constraint
    om, gm, rb = S.make_talos_pair(2, max_iters=2, mpc_override=TC)
    _loop(om, gm, S.talos_random_states(rb, 2, scale=0.7), 8, 1e-8)
