"""Kinodynamics OCP of a robot with 6-D (flat) feet on the device: the Talos configuration of the reference (KinodynamicsOCP with
force_size = 6: src/kinodynamics.cpp:40-152 -- contact torques in the momentum balance, FramePlacementResidual pose cost :66-72, 6-row LOCAL
frame velocity per foot in contact :105-123, CentroidalWrenchConeResidual :114-119; settings examples/talos_kinodynamics.py:43-106,
tests/test_utils.cpp:147-197; what tests/problem.cpp:106-160 and tests/mpc.cpp exercise).  It runs as the kinodynamics variant of the dense
stage / solver kernels (FullDims<23, 2, 6, 0, 0, 1>).  HIP path / emulated kernel bodies against the oracle, <= 1e-4 relative; golden replay
(tests/golden/talos_kino_golden.npz, make_golden_talos_kino.py)."""
import os

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

TOL = 1e-4
SHORT = dict(horizon=20, cycle=O.walk_cycle(5, 20), mpc_override=dict(T_fly=20, T_contact=5))
TIGHT = dict(mu=0.1, Lfoot=0.01, Wfoot=0.01)  # tiny, slippery soles: dozens of wrench-cone rows become active
TURN = (0.3, 0.2, 0, 0, 0, 0.3)
NA = 22
GK = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "talos_kino_golden.npz"))


def _loop(om, gm, rb, steps, scale=0.7, B=2, expect_cones=False, tol=TOL):
    assert len(om.cold_trace()) == len(gm.cold_trace())
    assert S.rel_err(om.xs, gm.xs) < tol
    X = S.talos_random_states(rb, B, scale=scale)
    worst, cones, backtracked = 0.0, 0, 0
    for step in range(steps):
        om.iterate(X)
        gm.iterate(X)
        e = S.rel_err(om.xs, gm.xs)
        worst = max(worst, e)
        assert e < tol, (step, e)
        assert S.rel_err(om.us, gm.us) < 10 * tol and S.rel_err(om.K0, gm.K0) < tol
        assert S.alphas_agree(om, gm), ("line-search step sizes differ", om.info[:, :4], gm.info[:, :4])
        assert S.rel_err(om.foot_refs, gm.getReferencePoses()) < 1e-12
        vo, vg = om.vs, gm.vs
        assert vo.shape == vg.shape
        assert S.rel_err(vo, vg) < 1e-3  # multipliers = residual / mu: rounding of the residual times 1e8
        cones = max(cones, int((np.abs(vg[:, :, NA + 12:]) > 0).sum()))
        backtracked += int((gm.info[:, 2] < 1.0).sum())
        X = om.xs[:, 1, :].copy()
    if expect_cones:
        assert cones >= 20, "the scenario must activate wrench-cone rows"
        assert backtracked > 0, "the scenario must make the line search backtrack"
    return worst


def test_dimensions_and_problem_surface(built):
    """reference tests/problem.cpp:106-160 on the batched handle: sizes, default horizon, contact state, weights echo, force references of
    6-D feet, pose references; the constraint-block count of the reference's stage (joint box + wrench cone + frame velocity = 3 with one foot
    in contact) is checked on the oracle's rows in tests/test_oracle_kino6d.py."""
    gm, rb, s, ms = S.make_talos_kino_product(2, max_iters=1, lib=S.emu_lib(), horizon=20, mpc_override=SHORT["mpc_override"])
    assert (gm.nx, gm.ndx, gm.nu) == (57, 56, 34)  # nu = nv - 6 + 6 * 2 (src/kinodynamics.cpp:34)
    ocp = gm.ocp_handler
    assert ocp.getSize() == 20 and ocp.getNu() == 34 and ocp.getCostNumber() == 6  # tests/problem.cpp:139 (6 cost components)
    assert ocp.getContactSupport(2) == 2 and ocp.getContactState(2) == [True, True]
    assert gm.vs.shape == (2, 20, NA + 12 + 34)
    u0 = ocp.getReferenceControl(0)
    fz = rb.mass * 9.81 / 2
    assert np.allclose(u0[:12], [0, 0, fz, 0, 0, 0] * 2) and np.all(u0[12:] == 0)  # default force references -m g / nf (src/ocp-handler.cpp:118-124)
    f1 = np.array([0, 1.0, 800, 0, 0, 0])
    ocp.setReferenceForce(3, "left_sole_link", f1)  # tests/problem.cpp:157-170 (6-D force references)
    assert np.array_equal(ocp.getReferenceForce(3, "left_sole_link"), f1)
    with pytest.raises(RuntimeError):
        ocp.setReferenceForces(3, {"left_sole_link": np.zeros(3), "right_sole_link": np.zeros(3)})  # force size mismatch (src/kinodynamics.cpp:235)
    p = np.array([0.1, 0.2, 0.3])
    ocp.setReferencePose(4, "left_sole_link", p)
    assert np.array_equal(ocp.getReferencePose(4, "left_sole_link"), p)
    with pytest.raises(RuntimeError):
        ocp.createProblem(rb.x_ref, 20, 3, -9.81, False)  # force size in settings does not match (src/ocp-handler.cpp:104)


def test_foot_timings_of_the_reference_test(built):
    """The integer known-answer test of reference tests/mpc.cpp:46-90 (Talos, H = 100, cycle 10 double / 50 left / 10 double / 50 right) on a
    kinodynamics handle: 170 / 110 / 219 / 160 after generateCycleHorizon, 160 / 100 / 209 / 150 after ten control steps."""
    gm, rb, _, _ = S.make_talos_kino_product(1, max_iters=1, lib=S.emu_lib(), horizon=100, settings_override=dict(force_cone=False))
    cs = np.array([[1, 1]] * 10 + [[1, 0]] * 50 + [[1, 1]] * 10 + [[0, 1]] * 50, np.uint8)
    gm.generateCycleHorizon(cs)
    assert gm.getFootTakeoffCycle("left_sole_link")[0] == 170 and gm.getFootTakeoffCycle("right_sole_link")[0] == 110
    assert gm.getFootLandCycle("left_sole_link")[0] == 219 and gm.getFootLandCycle("right_sole_link")[0] == 160
    X = np.tile(rb.x_ref, (1, 1))
    for _ in range(10):
        gm.iterate(X)
    assert gm.getFootTakeoffCycle("left_sole_link")[0] == 160 and gm.getFootTakeoffCycle("right_sole_link")[0] == 100
    assert gm.getFootLandCycle("left_sole_link")[0] == 209 and gm.getFootLandCycle("right_sole_link")[0] == 150


def test_emulated_kernels_closed_loop(built):
    om, gm, rb = S.make_talos_kino_pair(2, max_iters=2, lib=S.emu_lib(), **SHORT)
    worst = _loop(om, gm, rb, 8, tol=1e-7)
    assert worst < 1e-7


def test_emulated_kernels_active_wrench_cones(built):
    om, gm, rb = S.make_talos_kino_pair(2, max_iters=2, lib=S.emu_lib(), walk=TURN, settings_override=TIGHT, **SHORT)
    _loop(om, gm, rb, 6, expect_cones=True, tol=1e-6)


def test_emulated_kernels_without_cones_and_with_terminal_constraint(built):
    """force_cone = False (the reference's example script) and createProblem(..., terminal_constraint = True) (its tests): the DCM equality at
    the terminal node."""
    om, gm, rb = S.make_talos_kino_pair(2, max_iters=2, lib=S.emu_lib(), settings_override=dict(force_cone=False), horizon=20, cycle=SHORT["cycle"],
                                        mpc_override=dict(SHORT["mpc_override"], terminal_constraint=True))
    assert gm.vs.shape[2] == NA + 12
    _loop(om, gm, rb, 6, tol=1e-6)


def test_emulated_kernels_stage_knots(built):
    om, gm, rb = S.make_talos_kino_pair(1, max_iters=1, lib=S.emu_lib(), walk=TURN, settings_override=TIGHT, **SHORT)
    om.keep_knots()
    X = S.talos_random_states(rb, 1, scale=1.0)  # (round-6 robot table: at 0.7 no wrench-cone row wakes up at the stages looked at)
    mu = 1e-8
    for it in range(3):
        om.iterate(X)
        gm.iterate(X)
        X = om.xs[:, 1, :].copy()
        if it != 1:
            continue
        active = 0
        for t in (0, 3, 12, 19):
            ko, kg = om.knot(0, t), gm.debug_lq(0, t)
            # (r carries D^T nu: the multipliers are residuals / mu, their rounding times 1e8; f = mu (lam+ - lam) of the second iterate of a
            #  scenario with dozens of active cone rows: 0.9e-10 .. 1.1e-10 depending on the rounding of the sweep that made the iterate)
            for k in ("A", "B", "S", "R", "f", "r"):
                assert S.rel_err(ko[k], kg[k]) < (1e-7 if k == "r" else (3e-10 if k == "f" else 1e-10)), (t, k, S.rel_err(ko[k], kg[k]))
            # the frame-velocity rows: the oracle keeps their multipliers explicit, the stage kernel folds them (Q += Cv^T Cv / mu)
            Cv = kg["Cv"]
            # (Jacobian rows evaluated at the second iterate of either side, like A, B above: the same gate -- 1e-11 .. 3e-11 observed)
            assert S.rel_err(ko["C"][NA:NA + 12], Cv) < 1e-10
            Qg = np.triu(kg["Q"]) + np.triu(kg["Q"], 1).T
            assert np.abs(Qg - ko["Q"] - Cv.T @ Cv / mu).max() < 1e-12 * np.abs(Qg).max()
            # wrench-cone rows: constant rows on the wrench of the foot, present where active
            Do = ko["C"][NA + 12:, :] if ko["C"].shape[1] == 56 else None
            act = kg["act"][34 + NA: 34 + NA + 34] > 0
            active += int(act.sum())
            assert np.all(kg["Cd"] == 0)
            assert np.all((np.abs(kg["Dd"]).sum(1) > 0) == act)
            assert np.abs(ko["d"][NA:NA + 12] - kg["d"][34 + NA + 34:]).max() < 1e-9 * max(1.0, np.abs(ko["d"]).max())
            del Do
    assert active > 0, "the iterate must hold active wrench-cone rows"


def _golden(tag, lib):
    over, walk = (None, (0.1, 0, 0, 0, 0, 0)) if tag == "loop" else (TIGHT, TURN)
    gm, rb, _, _ = S.make_talos_kino_product(2, max_iters=2, lib=lib, horizon=20, settings_override=over, mpc_override=SHORT["mpc_override"])
    gm.generateCycleHorizon(SHORT["cycle"])
    gm.switchToWalk(np.array(walk, float))
    cold = gm.xs[0]
    X = GK[tag + "_X0"].copy()
    for _ in range(6):
        gm.iterate(X)
        X = gm.xs[:, 1, :].copy()
    assert S.rel_err(GK[tag + "_cold_xs"], cold) < TOL
    assert S.rel_err(GK[tag + "_xs"], gm.xs) < TOL and S.rel_err(GK[tag + "_us"], gm.us) < 10 * TOL and S.rel_err(GK[tag + "_K0"], gm.K0) < TOL
    assert np.array_equal(GK[tag + "_alpha"], gm.info[:, 2])
    assert S.rel_err(GK[tag + "_vs"], gm.vs) < 1e-3
    if tag == "cone":
        assert (np.abs(gm.vs[:, :, NA + 12:]) > 0).sum() >= 8  # (25 - 30 rows in the first steps, fewer as the loop settles)


def test_oracle_reproduces_golden_stage_vectors():
    rb = O.Robot("talos_like")
    kino = O.Kino(rb, O.talos_kino_settings(rb))
    for i, m in enumerate(GK["stage_mask"]):
        args = (int(m), GK["stage_u_ref"], rb.x_ref, GK["stage_foot_ref"], GK["stage_x"][i], GK["stage_u"][i])
        e, d = kino.eval(*args), kino.deriv(*args)
        assert S.rel_err(GK["stage%d_xnext" % i], e["xnext"]) < 1e-12 and S.rel_err(GK["stage%d_c" % i], e["c"]) < 1e-10
        assert abs(GK["stage%d_cost" % i] - e["cost"]) < 1e-9 * abs(e["cost"])
        for k in ("A", "B", "lx", "lu", "Lxx", "Lxu", "Luu", "Cx", "Cu"):
            assert S.rel_err(GK["stage%d_%s" % (i, k)], d[k]) < 1e-9, (i, k)


@pytest.mark.parametrize("tag", ["loop", "cone"])
def test_emulated_kernels_reproduce_golden_closed_loop(built, tag):
    _golden(tag, S.emu_lib())


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["loop", "cone"])
def test_hip_reproduces_golden_closed_loop(built, tag):
    _golden(tag, None)


@pytest.mark.gpu
@pytest.mark.parametrize("iters", [1, 3])
def test_hip_closed_loop_parity(built, iters):
    """H = 100, walking cycle 20 / 80 / 20 / 80 of the reference example (examples/talos_kinodynamics.py:132-150); 30 control steps (the first
    single-support stages enter the horizon at step 20), wrench-cone rows on as in the reference's tests."""
    om, gm, rb = S.make_talos_kino_pair(2, max_iters=iters)
    worst = _loop(om, gm, rb, 30)
    print("Talos kinodynamics (6-D feet), k=%d: worst relative xs error over 30 steps %.3e" % (iters, worst))


@pytest.mark.gpu
def test_hip_active_wrench_cones(built):
    om, gm, rb = S.make_talos_kino_pair(2, max_iters=2, walk=TURN, settings_override=TIGHT, **SHORT)
    worst = _loop(om, gm, rb, 12, expect_cones=True)
    print("Talos kinodynamics with active wrench cones: worst relative xs error over 12 steps %.3e" % worst)


@pytest.mark.gpu
def test_hip_full_size_properties(built):
    """B = 1024, H = 100: 8 distinct states against the oracle, replicas bit-identical, merit descent."""
    B, nd = 1024, 8
    gm, rb, _, _ = S.make_talos_kino_product(B, max_iters=3)
    gm.generateCycleHorizon(O.walk_cycle())
    gm.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
    om = O.OracleMPC(O.Kino(rb, O.talos_kino_settings(rb)), O.talos_mpc_settings(rb, max_iters=3), nd)
    om.generateCycleHorizon(O.walk_cycle())
    om.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
    Xo = S.talos_random_states(rb, nd, seed=5, scale=0.7)
    X = np.tile(Xo, (B // nd, 1))
    for _ in range(2):
        gm.iterate(X)
        om.iterate(Xo)
        xs = gm.xs
        X = xs[:, 1, :].copy()
        Xo = om.xs[:, 1, :].copy()
    xs = xs.reshape(B // nd, nd, *xs.shape[1:])
    assert np.abs(xs - xs[0:1]).max() == 0.0, "replicated instances must be bit-identical"
    assert S.rel_err(om.xs, xs[0]) < TOL
    info = gm.info
    assert np.all(np.isfinite(info)) and np.all(info[:, 1] < 0) and np.all(info[:, 3] <= info[:, 0])
