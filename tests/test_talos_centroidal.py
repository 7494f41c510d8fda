"""Centroidal OCP of a robot with 6-D (flat) feet on the device: the Talos configuration of the reference (CentroidalOCP with force_size = 6:
src/centroidal-dynamics.cpp:39-106 -- contact torques in the angular-momentum rate and the angular_acc residual, CentroidalWrenchConeResidual per
foot in contact :90-95; settings examples/talos_centroidal.py:39-96, tests/test_utils.cpp:199-218; what tests/problem.cpp:196-286 and the
`mpc_centroidal` test exercise).  (instance x stage) kernels around the dense Riccati sweep (smpc_cent6_kernels.h).  HIP path / emulated kernel
bodies against the oracle, <= 1e-4 relative; golden replay (tests/golden/talos_cent_golden.npz, make_golden_talos_cent.py)."""
import os

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

TOL = 1e-4
SHORT = dict(horizon=20, cycle=O.walk_cycle(5, 20), mpc_override=dict(T_fly=20, T_contact=5))
TIGHT = dict(mu=0.5, Lfoot=0.01, Wfoot=0.01)  # tiny soles: the centre-of-pressure and yaw rows of the wrench cones become active
GC = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "talos_cent_golden.npz"))


def _loop(om, gm, rb, steps, B=2, expect_cones=False, tol=TOL):
    assert len(om.cold_trace()) == len(gm.cold_trace())
    e0 = S.rel_err(om.xs, gm.xs)
    if e0 >= tol:
        # the cold solves may differ by a line search decided by rounding (S.alphas_agree): a step taken at an already converged point
        # (third column of the trace: dual residual below 1e-5), whose size is that of the rounding noise in the merit
        a, b = np.array(om.cold_trace()), np.array(gm.cold_trace())
        flips = a[:, 3] != b[:, 3]
        assert flips.any() and (a[flips, 2] < 1e-5).all() and e0 < 100 * tol, (e0, a, b)
    worst, cones, backtracked = 0.0, 0, 0
    for step in range(steps):
        X = S.talos_random_states(rb, B, seed=step, scale=0.5)
        om.iterate(X)
        gm.iterate(X)
        e = S.rel_err(om.xs, gm.xs)
        worst = max(worst, e)
        assert e < tol, (step, e)
        assert S.rel_err(om.us, gm.us) < 10 * tol and S.rel_err(om.K0, gm.K0) < tol
        assert S.alphas_agree(om, gm), ("line-search step sizes differ", om.info[:, :4], gm.info[:, :4])
        assert S.rel_err(om.foot_refs, gm.getReferencePoses()) < 1e-12
        assert om.vs.shape == gm.vs.shape == (B, gm.H, 34)
        assert S.rel_err(om.vs, gm.vs) < 1e-3
        xd = np.stack([gm.getStateDerivative(0), gm.getStateDerivative(1)], 1)
        assert S.rel_err(om.xdot[:, :2], xd) < tol
        cones = max(cones, int((np.abs(gm.vs) > 0).sum()))
        backtracked += int((gm.info[:, 2] < 1.0).sum())
    if expect_cones:
        assert cones >= 10, "the scenario must activate wrench-cone rows"
        assert backtracked > 0
    return worst


def test_dimensions_and_problem_surface(built):
    """reference tests/problem.cpp:196-286 on the batched handle: sizes, default horizon, contact state, 6-D force references."""
    gm, rb, s, ms = S.make_talos_cent_product(2, max_iters=1, lib=S.emu_lib(), horizon=20, mpc_override=SHORT["mpc_override"])
    assert (gm.nx, gm.ndx, gm.nu, gm.nc) == (9, 9, 12, 34)  # nu = 6 * 2 (src/centroidal-dynamics.cpp:31-32); 17 wrench-cone rows per foot
    ocp = gm.ocp_handler
    assert ocp.getSize() == 20 and ocp.getNu() == 12 and ocp.getCostNumber() == 6  # tests/problem.cpp:231 (6 cost components)
    assert ocp.getContactSupport(2) == 2 and ocp.getContactState(2) == [True, True]
    fz = rb.mass * 9.81 / 2
    assert np.allclose(ocp.getReferenceControl(0), [0, 0, fz, 0, 0, 0] * 2)
    f1 = np.array([0, 1.0, 800, 0, 0, 0])
    ocp.setReferenceForce(3, "left_sole_link", f1)  # tests/problem.cpp:250-262
    assert np.array_equal(ocp.getReferenceForce(3, "left_sole_link"), f1)
    with pytest.raises(RuntimeError):
        ocp.setReferenceForces(3, {"left_sole_link": np.zeros(3), "right_sole_link": np.zeros(3)})  # src/centroidal-dynamics.cpp:114
    with pytest.raises(RuntimeError):
        ocp.createProblem(np.zeros(9), 20, 3, -9.81, False)


def test_emulated_kernels_closed_loop(built):
    om, gm, rb = S.make_talos_cent_pair(2, max_iters=2, lib=S.emu_lib(), **SHORT)
    assert _loop(om, gm, rb, 10, tol=1e-9) < 1e-9


def test_emulated_kernels_active_wrench_cones(built):
    om, gm, rb = S.make_talos_cent_pair(2, max_iters=2, lib=S.emu_lib(), settings_override=TIGHT, **SHORT)
    _loop(om, gm, rb, 8, expect_cones=True, tol=1e-7)


def test_emulated_kernels_interpolation_feedback_checkpoint(built):
    om, gm, rb = S.make_talos_cent_pair(2, max_iters=1, lib=S.emu_lib(), **SHORT)
    X = S.talos_random_states(rb, 2, scale=0.3)
    for _ in range(3):
        om.iterate(X)
        gm.iterate(X)
    x, xd, f = gm.interpolate(0.004, knots=2)
    xs, us = gm.xs, gm.us
    assert np.allclose(x, 0.6 * xs[:, 0] + 0.4 * xs[:, 1], atol=1e-12) and f.shape == (2, 2, 6)
    assert np.allclose(f.reshape(2, 12), 0.6 * us[:, 0] + 0.4 * us[:, 1], atol=1e-12)
    u = gm.riccatiFeedback(0.0, X)
    cs = gm.updateInternalData(X)["centroidal_state"] if isinstance(gm.updateInternalData(X), dict) else None
    if cs is not None:
        assert np.allclose(u, us[:, 0] - np.einsum("bij,bj->bi", gm.K0, xs[:, 0] - cs), atol=1e-9)
    assert gm.Ks.shape == (2, 20, 12, 9) and np.array_equal(gm.Ks[:, 0], gm.K0)
    blob = gm.save_state()
    gm.iterate(X)
    a = gm.xs.copy()
    gm.load_state(blob)
    gm.iterate(X)
    assert np.array_equal(a, gm.xs)


def test_oracle_reproduces_golden_stage_vectors():
    rb = O.Robot("talos_like")
    cent = O.Cent(rb, O.talos_centroidal_settings(rb))
    for i, m in enumerate(GC["stage_mask"]):
        args = (int(m), GC["stage_u_ref"], GC["stage_x_tgt"], GC["stage_pos"], GC["stage_x"][i], GC["stage_u"][i])
        e, d = cent.eval(*args), cent.deriv(*args)
        assert S.rel_err(GC["stage%d_xnext" % i], e["xnext"]) < 1e-12 and S.rel_err(GC["stage%d_c" % i], e["c"]) < 1e-10
        assert abs(GC["stage%d_cost" % i] - e["cost"]) < 1e-9 * abs(e["cost"])
        for k in ("A", "B", "lx", "lu", "Lxx", "Lxu", "Luu", "Cx", "Cu"):
            assert S.rel_err(GC["stage%d_%s" % (i, k)], d[k]) < 1e-9, (i, k)


def _golden(tag, lib):
    gm, rb, _, _ = S.make_talos_cent_product(2, max_iters=2, lib=lib, horizon=20, settings_override=None if tag == "loop" else TIGHT,
                                             mpc_override=SHORT["mpc_override"])
    gm.generateCycleHorizon(SHORT["cycle"])
    gm.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
    assert S.rel_err(GC[tag + "_cold_xs"], gm.xs[0]) < TOL
    for X in GC[tag + "_X"]:
        gm.iterate(X)
    assert S.rel_err(GC[tag + "_xs"], gm.xs) < TOL and S.rel_err(GC[tag + "_us"], gm.us) < 10 * TOL and S.rel_err(GC[tag + "_K0"], gm.K0) < TOL
    assert np.array_equal(GC[tag + "_alpha"], gm.info[:, 2])
    assert S.rel_err(GC[tag + "_vs"], gm.vs) < 1e-3
    if tag == "cone":
        assert (np.abs(gm.vs) > 0).sum() > 0  # (wrench-cone multipliers alive at the end of the replay)


@pytest.mark.parametrize("tag", ["loop", "cone"])
def test_emulated_kernels_reproduce_golden_closed_loop(built, tag):
    _golden(tag, S.emu_lib())


def _two_pass_line_search(lib, steps):
    """SMPC_CENT6_LS_PASSES=2 (the full step first, the backtracking candidates where it failed: smpc_cent6_kernels.h) against the default (every
    candidate in one launch): the same kernels over other candidate ranges -- bit-identical trajectories, in a scenario that backtracks."""
    import os

    def run(passes):
        old = os.environ.get("SMPC_CENT6_LS_PASSES")
        os.environ["SMPC_CENT6_LS_PASSES"] = str(passes)
        try:
            gm, rb, _, _ = S.make_talos_cent_product(3, max_iters=2, lib=lib, horizon=20, settings_override=TIGHT, mpc_override=SHORT["mpc_override"])
        finally:
            os.environ.pop("SMPC_CENT6_LS_PASSES", None)
            if old is not None:
                os.environ["SMPC_CENT6_LS_PASSES"] = old
        gm.generateCycleHorizon(SHORT["cycle"])
        gm.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
        out = []
        for step in range(steps):
            gm.iterate(S.talos_random_states(rb, 3, seed=step, scale=0.5))
            out.append((gm.xs.copy(), gm.us.copy(), gm.vs.copy(), gm.info.copy(), gm.getStateDerivative(0).copy(), gm.getStateDerivative(1).copy()))
        return out

    a, b = run(1), run(2)
    for sa, sb in zip(a, b):
        for u, v in zip(sa, sb):
            assert np.array_equal(u, v)
    assert any((s[3][:, 2] < 1.0).any() for s in a), "the scenario must backtrack"


def test_emulated_kernels_two_pass_line_search(built):
    _two_pass_line_search(S.emu_lib(), 8)


@pytest.mark.gpu
def test_hip_two_pass_line_search(built):
    _two_pass_line_search(S.xcheck_lib(), 8)  # (SMPC_CENT6_LS_PASSES: a switch of the cross-check build)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["loop", "cone"])
def test_hip_reproduces_golden_closed_loop(built, tag):
    _golden(tag, None)


@pytest.mark.gpu
@pytest.mark.parametrize("iters", [1, 3])
def test_hip_closed_loop_parity(built, iters):
    """H = 100, walking cycle 20 / 80 / 20 / 80 of the reference example (examples/talos_centroidal.py:98-120); 30 control steps."""
    om, gm, rb = S.make_talos_cent_pair(2, max_iters=iters)
    worst = _loop(om, gm, rb, 30)
    print("Talos centroidal (6-D feet), k=%d: worst relative xs error over 30 steps %.3e" % (iters, worst))


@pytest.mark.gpu
def test_hip_active_wrench_cones(built):
    om, gm, rb = S.make_talos_cent_pair(2, max_iters=2, settings_override=TIGHT, **SHORT)
    worst = _loop(om, gm, rb, 12, expect_cones=True)
    print("Talos centroidal with active wrench cones: worst relative xs error over 12 steps %.3e" % worst)


@pytest.mark.gpu
def test_hip_full_size_properties(built):
    """B = 4096, H = 100: 64 distinct measured states against the oracle, replicas bit-identical."""
    B, nd = 4096, 64
    gm, rb, _, _ = S.make_talos_cent_product(B, max_iters=3)
    gm.generateCycleHorizon(O.walk_cycle())
    gm.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
    om = O.OracleCentMPC(O.Cent(rb, O.talos_centroidal_settings(rb)), O.talos_mpc_settings(rb, max_iters=3), nd)
    om.generateCycleHorizon(O.walk_cycle())
    om.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
    for step in range(2):
        Xo = S.talos_random_states(rb, nd, seed=5 + step, scale=0.5)
        gm.iterate(np.tile(Xo, (B // nd, 1)))
        om.iterate(Xo)
    xs = gm.xs.reshape(B // nd, nd, gm.H + 1, 9)
    assert np.abs(xs - xs[0:1]).max() == 0.0, "replicated instances must be bit-identical"
    assert S.rel_err(om.xs, xs[0]) < TOL
    assert np.all(np.isfinite(gm.info))
