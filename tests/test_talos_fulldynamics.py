"""Talos-class full-dynamics OCP on the device (BASELINE configs[3]: "Talos fulldynamics (state~64, friction-cone QP), H=100,
batch=1024"): 6-D feet with CONTACT_6D LOCAL_WORLD_ALIGNED constraints, FramePlacement pose costs, 17-row wrench cones whose
multipliers are pivoted explicitly in the matrix-core Riccati sweep (reference src/fulldynamics.cpp:56-65, 103-109, 163-173;
settings examples/talos_fulldynamics.py:47-115).  HIP path / emulated kernel bodies against the oracle, <= 1e-4 relative."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

TOL = 1e-4
SHORT = dict(horizon=20, cycle=O.walk_cycle(5, 20), mpc_override=dict(T_fly=20, T_contact=5))
TIGHT = dict(mu=0.3, Lfoot=0.05, Wfoot=0.04)  # small, slippery soles: dozens of wrench-cone rows become active


def _loop(om, gm, rb, steps, scale=0.7, B=2, expect_cones=False):
    assert len(om.cold_trace()) == len(gm.cold_trace())
    assert S.rel_err(om.xs, gm.xs) < TOL
    X = S.talos_random_states(rb, B, scale=scale)
    worst, cones = 0.0, 0
    for step in range(steps):
        om.iterate(X)
        gm.iterate(X)
        e = S.rel_err(om.xs, gm.xs)
        worst = max(worst, e)
        assert e < TOL, (step, e)
        assert S.rel_err(om.us, gm.us) < 10 * TOL and S.rel_err(om.K0, gm.K0) < TOL
        assert S.alphas_agree(om, gm), ("line-search step sizes differ", om.info[:, :4], gm.info[:, :4])
        assert S.rel_err(om.foot_refs, gm.getReferencePoses()) < 1e-12
        nb = 2 * gm.nu
        assert S.rel_err(om.vs, gm.vs) < 1e-3  # multipliers = residual / mu: rounding of the residual times 1e8
        cones = max(cones, int((np.abs(gm.vs[:, :, nb:]) > 0).sum()))
        X = om.xs[:, 1, :].copy()
    if expect_cones:
        assert cones >= 20, "the scenario must activate wrench-cone rows"
    return worst


def test_emulated_kernels_talos_closed_loop(built):
    om, gm, rb = S.make_talos_pair(2, max_iters=2, lib=S.emu_lib(), **SHORT)
    assert (gm.nx, gm.ndx, gm.nu, gm.nc) == (57, 56, 22, 78)  # torque box 22 + joint box 22 + 2 x 17 cone rows
    _loop(om, gm, rb, 8)


def test_emulated_kernels_talos_active_wrench_cones(built):
    om, gm, rb = S.make_talos_pair(2, max_iters=2, lib=S.emu_lib(), walk=(0.2, 0.1, 0, 0, 0, 0.2), settings_override=TIGHT, **SHORT)
    _loop(om, gm, rb, 7, expect_cones=True)
    f = gm.getContactForces()
    assert f.shape == (2, gm.H, 2, 6) and np.all(np.isfinite(f))
    for t in range(gm.H):
        on = np.array(gm.ocp_handler.getContactState(t))
        assert np.all(f[:, t, ~on, :] == 0.0)


def test_emulated_kernels_talos_stage_knots(built):
    om, gm, rb = S.make_talos_pair(1, max_iters=1, lib=S.emu_lib(), walk=(0.2, 0.1, 0, 0, 0, 0.2), settings_override=TIGHT, **SHORT)
    om.keep_knots()
    X = S.talos_random_states(rb, 1, scale=0.7)
    nb = 2 * gm.nu
    # the first iterates agree to rounding; later ones only as far as the closed loop (active-set changes, Armijo decisions) carries
    # the rounding of the earlier ones, so the comparison is tight early and loose once wrench-cone rows are active
    for it in range(6):
        om.iterate(X)
        gm.iterate(X)
        X = om.xs[:, 1, :].copy()
        if it not in (1, 5):
            continue
        tol = 1e-11 if it == 1 else 1e-6
        active = 0
        for t in (0, 3, 12, 19):
            ko, kg = om.knot(0, t), gm.debug_lq(0, t)
            for k in ("A", "B", "Q", "S", "R", "f"):
                assert S.rel_err(ko[k], kg[k]) < tol, (it, t, k, S.rel_err(ko[k], kg[k]))
            assert S.rel_err(ko["C"][nb:], kg["Cd"]) < tol, (it, t)  # active wrench-cone rows A_cone d lam / dx (zero when inactive)
            assert np.abs(ko["d"] - kg["d"]).max() < 1e3 * tol * max(1.0, np.abs(ko["d"]).max())
            active += int(np.abs(kg["Cd"]).max() > 0)
    assert active > 0, "the last iterate must hold active wrench-cone rows"


@pytest.mark.gpu
@pytest.mark.parametrize("iters", [1, 3])
def test_hip_talos_closed_loop_parity(built, iters):
    """H = 100, walking cycle 20 / 80 / 20 / 80 of the reference example; 30 control steps (the first single-support stages enter
    the horizon at step 20)."""
    om, gm, rb = S.make_talos_pair(2, max_iters=iters)
    worst = _loop(om, gm, rb, 30)
    print("Talos full dynamics, k=%d: worst relative xs error over 30 steps %.3e" % (iters, worst))


@pytest.mark.gpu
def test_hip_talos_active_wrench_cones(built):
    om, gm, rb = S.make_talos_pair(2, max_iters=2, walk=(0.2, 0.1, 0, 0, 0, 0.2), settings_override=TIGHT, **SHORT)
    _loop(om, gm, rb, 12, expect_cones=True)


@pytest.mark.gpu
def test_hip_talos_full_size_properties(built):
    """B = 1024, H = 100 (BASELINE configs[3]): 8 distinct states against the oracle, replicas bit-identical, merit descent."""
    B, nd = 1024, 8
    gm, rb, _, _ = S.make_talos_product(B, max_iters=3)
    gm.generateCycleHorizon(O.walk_cycle())
    gm.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
    s = O.talos_full_settings(rb)
    ms = O.talos_mpc_settings(rb, max_iters=3)
    om = O.OracleFullMPC(O.Full(rb, s), ms, nd)
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


@pytest.mark.gpu
@pytest.mark.parametrize("robot", ["go2", "talos"])
def test_hip_long_closed_loop_is_stable_and_deterministic(built, robot):
    """A whole gait cycle and more of the full-dynamics MPC closed on its own prediction plus noise -- Go2: 200 control steps of the trot at
    0.2 m/s (cycle 80); Talos: 260 steps of the walk at 0.1 m/s (cycle 200, both single-support phases) with wrench cones: every output stays
    finite, the base stays up and advances, the contact forces of stage 0 carry the weight, and a second engine fed the same inputs
    reproduces the trajectory bit for bit."""
    import oracle_lib as O

    talos = robot == "talos"
    B, steps = (8, 260) if talos else (16, 200)
    runs = []
    for _ in range(2):
        if talos:
            gm, rb, _, _ = S.make_talos_product(B, max_iters=2)
            gm.generateCycleHorizon(O.walk_cycle())
            gm.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
            X = S.talos_random_states(rb, B, scale=0.5)
        else:
            gm, rb, _, _ = S.make_full_product(B, max_iters=2)
            gm.generateCycleHorizon(O.trot_cycle())
            gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
            X = S.random_states(rb, B)
        rng = np.random.default_rng(3)
        fz = []
        for _ in range(steps):
            gm.iterate(X)
            assert np.all(np.isfinite(gm.info))
            fz.append(gm.getContactForces(0)[:, :, 2].sum(axis=1))
            X = gm.xs[:, 1, :] + rng.normal(0.0, 1e-3, (B, gm.nx))
            X[:, 3:7] /= np.linalg.norm(X[:, 3:7], axis=1, keepdims=True)
        runs.append((gm.xs.copy(), gm.us.copy(), np.array(fz)))
    xs, us, fz = runs[0]
    w = rb.mass * 9.81
    # (the Go2 settings of record put no weight on the base pose -- examples/go2_fulldynamics.py:46 -- so its height wanders in a loop
    #  that is closed on the plan itself; it must stay a standing robot)
    print("base height: min %.3f max %.3f (reference %.3f)" % (xs[:, 0, 2].min(), xs[:, 0, 2].max(), rb.x_ref[2]))
    assert np.all(np.abs(xs[:, 0, 2] - rb.x_ref[2]) < (0.15 if talos else 0.2)), "the base should stay up"
    if talos:  # (2.6 s of a 0.1 m/s command, most of it in the first double-support and single-support phases: little ground covered)
        assert np.all(np.abs(xs[:, 0, :2]) < 0.5), "the base must not run away"
    else:
        assert np.all(xs[:, 0, 0] > 0.0) and xs[:, 0, 0].mean() > 0.1, "the base should advance under the command"
    assert 0.8 * w < np.median(fz) < 1.2 * w, "the stance feet carry the robot"
    assert np.array_equal(xs, runs[1][0]) and np.array_equal(us, runs[1][1])
