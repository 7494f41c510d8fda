"""The OCPHandler per-stage setters / getters through the C ABI, written like the reference's tests/problem.cpp (kinodynamics
:108-196, centroidal :198-285, centroidal_solo :287-349): sizes, contact support / state, weight echo, set / get round trips --
and, beyond the reference, that the references that were set steer the solve exactly as they do in the oracle."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O


@pytest.fixture(scope="module")
def lib(built):
    return S.emu_lib()


def test_kinodynamics_problem_surface(lib):
    gm, rb, s, _ = S.make_product(1, lib=lib)
    ocp = gm.ocp_handler
    assert ocp.getCostNumber() == 4 + 4  # tests/problem.cpp:139 (6 for the two-footed robot there)
    assert ocp.getSize() == 50 and ocp.getContactSupport(2) == 4 and ocp.getContactState(2) == [True] * 4
    assert np.array_equal(ocp.getSettings()["w_u"], s["w_u"]) and np.array_equal(ocp.getSettings()["w_cent"], s["w_cent"])
    ocp.setReferencePose(4, "FL_foot", [0.3, -0.2, 0.7])
    assert np.array_equal(ocp.getReferencePose(4, "FL_foot"), [0.3, -0.2, 0.7])
    new = {n: np.array([(-1.0) ** i, 0.0, 2.0]) for i, n in enumerate(S.FEET)}
    ocp.setReferencePoses(3, new)
    for n in S.FEET:
        assert np.array_equal(ocp.getReferencePose(3, n), new[n])
    assert np.array_equal(gm.getReferencePose(3, "RR_foot"), new["RR_foot"])  # MPC::getReferencePose passthrough
    forces = {n: np.array([0.0, 0.0, 40.0]) for n in S.FEET}
    forces["FL_foot"][1] = 1.0
    forces["FR_foot"][0] = 1.0
    ocp.setReferenceForces(3, forces)
    for n in S.FEET:
        assert np.array_equal(ocp.getReferenceForce(3, n), forces[n])
    forces["FL_foot"][2] = 250.0
    ocp.setReferenceForce(5, "FL_foot", forces["FL_foot"])
    assert np.array_equal(ocp.getReferenceForce(5, "FL_foot"), forces["FL_foot"])
    assert np.array_equal(ocp.getReferenceForce(5, "FR_foot"), forces["FR_foot"])  # control_ref_ member semantics
    pose_base = np.array([0, 0, 2, 0, 0, 0, 1.0])
    ocp.setPoseBase(2, pose_base)
    assert np.array_equal(ocp.getPoseBase(2), pose_base)
    ocp.setVelocityBase(2, [0.1, 0, 0, 0, 0, 0.2])
    assert np.array_equal(ocp.getVelocityBase(2), [0.1, 0, 0, 0, 0, 0.2]) and np.array_equal(ocp.getPoseBase(2), pose_base)
    new_x = rb.integrate(rb.x_ref, np.r_[0.0, 0.1, 0.0, np.zeros(33)])
    ocp.setReferenceState(2, new_x)
    assert np.array_equal(ocp.getReferenceState(2), new_x)
    # errors of the reference
    with pytest.raises(RuntimeError, match="Stage index exceeds stage vector size"):
        ocp.setReferencePose(50, "FL_foot", [0, 0, 0])
    with pytest.raises(RuntimeError, match="velocity_base size should be 6"):
        ocp.setVelocityBase(2, [0.0] * 5)
    with pytest.raises(RuntimeError, match="pose_base size should be 7"):
        ocp.setPoseBase(2, [0.0] * 6)
    with pytest.raises(RuntimeError, match="pose_refs size does not match"):
        ocp.setReferencePoses(3, {"FL_foot": [0, 0, 0]})
    with pytest.raises(RuntimeError, match="force size in settings"):
        ocp.setReferenceForces(3, {n: np.zeros(6) for n in S.FEET})


def test_centroidal_problem_surface(lib):
    gm, rb, s, _ = S.make_cent_product(1, lib=lib)
    ocp = gm.ocp_handler
    assert ocp.getCostNumber() == 6 and ocp.getSize() == 50  # tests/problem.cpp:232,247
    assert ocp.getContactSupport(2) == 4 and ocp.getContactState(2) == [True] * 4
    assert np.array_equal(ocp.getSettings()["w_angular_acc"], s["w_angular_acc"])
    f = {n: np.array([0.0, 0.0, rb.mass / 3]) for n in S.FEET}
    f["FR_foot"][1] = 1.0
    ocp.setReferenceForces(3, f)
    assert np.array_equal(ocp.getReferenceForce(3, "FR_foot"), f["FR_foot"])
    ocp.setReferencePose(4, "FL_foot", [0.1, 0.2, 0.3])
    assert np.array_equal(ocp.getReferencePose(4, "FL_foot"), [0.1, 0.2, 0.3])
    ocp.setPoseBase(2, [0, 0, 2.0])
    assert np.array_equal(ocp.getPoseBase(2), [0, 0, 2.0])
    new_x = np.array([0, 0, 1, 0, 0.1, 0.1, 0.2, 0.2, 0.2])
    ocp.setReferenceState(2, new_x)
    assert np.allclose(ocp.getReferenceState(2), new_x, rtol=1e-15, atol=0)  # stored as momenta m v, returned / m
    with pytest.raises(RuntimeError, match="Stage index exceeds"):
        ocp.getReferenceState(50)


def _steer(make_pair, lib, kino):
    om, gm, rb = make_pair(2, 2, lib=lib)
    ocp = gm.ocp_handler
    X = S.random_states(rb, 2)
    om.iterate(X)
    gm.iterate(X)
    # a control target and a state target in the middle of the horizon
    u = ocp.getReferenceControl(7).copy()
    u[2] += 15.0
    u[5] -= 15.0
    ocp.setReferenceControl(7, u)
    om.set_stage_reference(7, 0, u)
    x = ocp.getReferenceState(9).copy()
    if kino:
        x[2] += 0.03
        x[rb.nq] = 0.4
    else:
        x[2] = 0.33
        x[3] = 0.4
    ocp.setReferenceState(9, x)
    om.set_stage_reference(9, 1, x)
    before = gm.us.copy()
    for _ in range(3):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < 1e-8 and S.rel_err(om.us, gm.us) < 1e-7
    assert np.abs(gm.us[:, :8] - before[:, 1:9]).max() > 1e-3, "the references that were set must steer the plan"
    # the stage that was set has moved 3 places towards the front of the horizon, with its references
    assert np.array_equal(ocp.getReferenceControl(4), u)


def test_stage_references_steer_the_solve_kinodynamics(lib):
    _steer(S.make_pair, lib, True)


def test_stage_references_steer_the_solve_centroidal(lib):
    _steer(S.make_cent_pair, lib, False)


@pytest.mark.gpu
def test_stage_references_on_the_gpu(built):
    _steer(S.make_pair, None, True)
    _steer(S.make_cent_pair, None, False)


def test_cycling_contact_state_follows_the_gait(lib):
    """MPC::getCyclingContactState / getCycleHorizon (reference include/simple-mpc/mpc.hpp:139-147): the sequence given to
    generateCycleHorizon, repeated 1 + H / n times (src/mpc.cpp:103-110) and rotated left by every walking control step (:230)."""
    gm, rb, _, _ = S.make_product(1, lib=lib)
    with pytest.raises(RuntimeError, match="generateCycleHorizon"):
        gm.getCyclingContactState(0, "FL_foot")
    cs = O.trot_cycle()  # 80 stages; H = 50 -> one copy, 80 entries
    gm.generateCycleHorizon(cs)
    cyc = gm.getCycleHorizon()
    assert len(cyc) == 80 * (1 + 50 // 80)
    for t in (0, 9, 10, 39, 40, 50, 79):
        for f, n in enumerate(S.FEET):
            assert gm.getCyclingContactState(t, n) == bool(cs[t][f]) == cyc[t][n]
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, 1)
    for k in range(1, 4):
        gm.iterate(X)
        for t in (0, 9, 38, 79):
            for f, n in enumerate(S.FEET):
                assert gm.getCyclingContactState(t, n) == bool(cs[(t + k) % 80][f])
    with pytest.raises(RuntimeError, match="Stage index"):
        gm.getCyclingContactState(80, "FL_foot")
    with pytest.raises(RuntimeError, match="pose_cost"):
        gm.setTerminalReferencePose("FL_foot", [0, 0, 0])  # the terminal cost stack has no pose cost, upstream as here


def test_problem_state_and_terminal_constraint_calls(lib):
    """getProblemState (src/kinodynamics.cpp:308-311, src/centroidal-dynamics.cpp:259-262) and the by-hand terminal-constraint calls
    (src/kinodynamics.cpp:366-388; left out upstream for the centroidal OCP, src/centroidal-dynamics.cpp:318-335)."""
    import simple_mpc

    rb = O.Robot("go2_like")
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in S.FEET:
        mh.addPointFoot(n, "root_joint")
    dh = simple_mpc.RobotDataHandler(mh)
    x = S.random_states(rb, 1)[0]
    dh.updateInternalData(x, False)
    kin = simple_mpc.KinodynamicsOCP(O.go2_kino_settings(rb), mh)
    with pytest.raises(RuntimeError, match="Create problem first"):
        kin.createTerminalConstraint(x[:3])
    kin.createProblem(mh.getReferenceState(), 20, 3, -9.81, False)
    assert np.array_equal(kin.getProblemState(dh), x)
    kin.createTerminalConstraint(mh.getReferenceState()[:3])  # what createProblem(..., True) does
    kin.updateTerminalConstraint([0.1, 0.0, 0.3])
    ms = {k: v for k, v in O.go2_mpc_settings(rb, max_iters=1).items() if k in S.MPC_KEYS}
    gm = simple_mpc.BatchedMPC(ms, kin, 1, lib=lib)
    ref, _, _ = S.make_product(1, lib=lib, horizon=20, mpc_override={"terminal_constraint": True})[0], None, None
    assert np.array_equal(gm.xs, ref.xs)  # the same problem as createProblem(..., terminal_constraint = True)
    cen = simple_mpc.CentroidalOCP(O.go2_centroidal_settings(rb), mh)
    cen.createProblem(np.zeros(9), 20, 3, -9.81, False)
    assert np.allclose(cen.getProblemState(dh), dh.getCentroidalState())
    cen.createTerminalConstraint([0, 0, 0.3])
    cen.updateTerminalConstraint([0, 0, 0.3])
