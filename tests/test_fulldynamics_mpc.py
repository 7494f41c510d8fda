"""Full-dynamics OCP on the device (reference FullDynamicsOCP under MPC, src/fulldynamics.cpp:30-455, src/mpc.cpp:189-218;
SURVEY 8a rows a7-a9, a16): the HIP path (C ABI, libsmpc_hip.so) and -- in the CPU tier -- the same kernel bodies compiled with the
sequential-lane test backend, against the oracle (oracle/orc_full.hpp, orc_fulldyn.hpp) on the same seeded inputs.
Tolerance (north_star): <= 1e-4 relative state-trajectory error."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _closed_loop(lib, iters, steps, B=3, horizon=50, check_knots=False, **kw):
    om, gm, rb = S.make_full_pair(B, max_iters=iters, lib=lib, horizon=horizon, **kw)
    assert len(om.cold_trace()) == len(gm.cold_trace())
    assert S.rel_err(om.xs, gm.xs) < TOL and S.rel_err(om.us, gm.us) < 10 * TOL
    X = S.random_states(rb, B)
    worst = 0.0
    for step in range(steps):
        om.iterate(X)
        gm.iterate(X)
        e = S.rel_err(om.xs, gm.xs)
        worst = max(worst, e)
        assert e < TOL, (step, e)
        assert S.rel_err(om.us, gm.us) < 10 * TOL, (step, S.rel_err(om.us, gm.us))
        assert S.rel_err(om.K0, gm.K0) < TOL, (step, S.rel_err(om.K0, gm.K0))
        assert S.alphas_agree(om, gm), ("line-search step sizes differ", om.info[:, :4], gm.info[:, :4])
        assert S.rel_err(om.foot_refs, gm.getReferencePoses()) < 1e-12
        X = om.xs[:, 1, :].copy()
    return om, gm, rb, worst


@pytest.mark.parametrize("iters", [1, 3])
def test_emulated_kernels_closed_loop(built, iters):
    """CPU tier: 14 control steps (the first swing stages enter the horizon at step 10)."""
    _closed_loop(S.emu_lib(), iters, 14)


def test_emulated_kernels_stage_knots(built):
    om, gm, rb = S.make_full_pair(2, max_iters=1, lib=S.emu_lib())
    om.keep_knots()
    X = S.random_states(rb, 2)
    for _ in range(12):  # swing stages inside the horizon
        om.iterate(X)
        gm.iterate(X)
        X = om.xs[:, 1, :].copy()
    masks = set()
    for t in (0, 1, 17, 38, 48, 49):
        ko, kg = om.knot(1, t), gm.debug_lq(1, t)
        masks.add(tuple(gm.ocp_handler.getContactState(t)))
        for k in ("A", "B", "Q", "S", "R", "f", "d"):
            assert S.rel_err(ko[k], kg[k]) < 1e-8, (t, k, S.rel_err(ko[k], kg[k]))
        for k in ("q", "r"):
            assert S.rel_err(ko[k], kg[k]) < 1e-6, (t, k)
    assert len(masks) >= 2, "the compared stages must include a swing stage"


def test_emulated_kernels_are_lane_order_independent(built, tmp_path):
    """Barrier audit: the sequential-lane backend run in DESCENDING lane order must give the same trajectories."""
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, mpc_setup as S\n"
        "gm, rb, _, _ = S.make_full_product(2, max_iters=2, lib=S.emu_lib())\n"
        "import oracle_lib as O\n"
        "gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2,0,0,0,0,0.]))\n"
        "X = S.random_states(rb, 2)\n"
        "for _ in range(12):\n"
        "    gm.iterate(X); X = gm.xs[:, 1, :].copy()\n"
        "np.save(sys.argv[1], gm.xs)\n" % (ROOT, os.path.join(ROOT, "tests"))
    )
    outs = []
    for rev in ("0", "1"):
        out = str(tmp_path / ("xs%s.npy" % rev))
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, SMPC_EMU_REVERSE=rev), timeout=900)
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1])


def test_contact_forces_accessor(built):
    """MPC::getContactForces (reference src/mpc.cpp:354-380): forces of the constrained dynamics at the solution; zero for feet
    in the air; the stage-0 forces carry the robot's weight."""
    om, gm, rb = S.make_full_pair(2, max_iters=2, lib=S.emu_lib())
    X = S.random_states(rb, 2)
    for _ in range(12):
        om.iterate(X)
        gm.iterate(X)
        X = om.xs[:, 1, :].copy()
    f = gm.getContactForces()
    assert f.shape == (2, gm.H, 4, 3) and np.all(np.isfinite(f))
    for t in range(gm.H):
        on = np.array(gm.ocp_handler.getContactState(t))
        assert np.all(f[:, t, ~on, :] == 0.0)
    # LOCAL contact frames of a near-level robot: the vertical components add up to about the weight
    assert np.all(np.abs(f[:, 0, :, 2].sum(axis=1) - rb.mass * 9.81) < 0.35 * rb.mass * 9.81)
    assert np.array_equal(gm.getContactForces(3), f[:, 3])
    # the oracle's forces at stage 0 / 1 (its xdot carries none: recompute them with the constrained dynamics at the solution)
    xs, us = gm.xs, gm.us
    for b in range(2):
        for t in (0, 1, 30):
            mask = sum(1 << i for i, c in enumerate(gm.ocp_handler.getContactState(t)) if c)
            r = rb.full_forward_dynamics(xs[b, t], us[b, t], mask)
            lam = np.zeros((4, 3))
            lam[[i for i in range(4) if (mask >> i) & 1]] = r["lam"].reshape(-1, 3)
            assert np.abs(lam - f[b, t]).max() < 1e-6 * max(1.0, np.abs(lam).max())


def test_boundary_errors(built):
    import simple_mpc

    lib = S.emu_lib()
    rb = O.Robot("go2_like")
    s = O.go2_full_settings(rb)
    s.update(dict(force_size=3, mu=0.8, Lfoot=0.01, Wfoot=0.01, force_cone=False, land_cstr=False))
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in S.FEET:
        mh.addPointFoot(n, "root_joint")
    with pytest.raises(KeyError):
        simple_mpc.FullDynamicsOCP({k: v for k, v in s.items() if k != "w_forces"}, mh)
    with pytest.raises(RuntimeError, match="Kp correction"):
        simple_mpc.FullDynamicsOCP(dict(s, Kp_correction=np.zeros(6)), mh)
    ocp = simple_mpc.FullDynamicsOCP(s, mh)
    with pytest.raises(RuntimeError, match="force size"):
        ocp.createProblem(mh.getReferenceState(), 10, 6, -9.81, False)
    ocp.createProblem(mh.getReferenceState(), 10, 3, -9.81, False)
    ms = O.go2_mpc_settings(rb)
    gm = simple_mpc.BatchedMPC({k: ms[k] for k in S.MPC_KEYS}, ocp, 1, lib=lib)
    assert (gm.nx, gm.ndx, gm.nu, gm.nc) == (37, 36, 12, 24)  # torque box + joint box (tests/problem.cpp:48-49 block counts)
    assert ocp.getCostNumber() == 11
    ocp.setReferenceForce(3, "FL_foot", [1.0, 2.0, 3.0])
    assert np.array_equal(ocp.getReferenceForce(3, "FL_foot"), [1.0, 2.0, 3.0])
    with pytest.raises(RuntimeError, match="right dimension"):
        ocp.setReferenceForce(3, "FL_foot", np.zeros(6))
    with pytest.raises(RuntimeError, match="Stage index"):
        ocp.setReferenceForce(10, "FL_foot", np.zeros(3))


# ---------------------------------------------------------------------------------------------------------------------------
# GPU tier
# ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("iters", [1, 3])
def test_hip_closed_loop_parity(built, iters):
    """48 control steps: take-off, swing and touch-down inside the compared horizon (reference src/mpc.cpp:220-254)."""
    om, gm, rb, worst = _closed_loop(None, iters, 48, B=4)
    seq = [tuple(gm.ocp_handler.getContactState(t)) for t in range(gm.H)]
    sw = [i for i, m in enumerate(seq) if not all(m)]
    assert sw and sw[0] > 0 and sw[-1] < gm.H - 1
    print("full dynamics, k=%d: worst relative xs error over 48 steps %.3e" % (iters, worst))


@pytest.mark.gpu
def test_hip_stage_knots(built):
    om, gm, rb = S.make_full_pair(2, max_iters=1)
    om.keep_knots()
    X = S.random_states(rb, 2)
    for _ in range(12):
        om.iterate(X)
        gm.iterate(X)
        X = om.xs[:, 1, :].copy()
    for t in (0, 1, 17, 38, 48, 49):
        ko, kg = om.knot(1, t), gm.debug_lq(1, t)
        for k in ("A", "B", "Q", "S", "R", "f", "d"):
            assert S.rel_err(ko[k], kg[k]) < 1e-8, (t, k, S.rel_err(ko[k], kg[k]))


@pytest.mark.gpu
def test_hip_unusual_horizon_and_backtracking(built):
    om, gm, rb = S.make_full_pair(4, max_iters=2, horizon=23)
    X = S.random_states(rb, 4, seed=3, scale=3.0)
    for _ in range(4):
        om.iterate(X)
        gm.iterate(X)
        assert np.array_equal(om.info[:, 2], gm.info[:, 2])
        assert S.rel_err(om.xs, gm.xs) < TOL
        X = om.xs[:, 1, :].copy()


def _parts_run(lib, parts, B, iters, steps, horizon, scale=1.0, talos=False):
    """Closed loop with the batch as `parts` parts on streams of their own (SMPC_FULL_PARTS, read when the handle is created)."""
    old = os.environ.get("SMPC_FULL_PARTS")
    os.environ["SMPC_FULL_PARTS"] = str(parts)
    try:
        gm, rb, _, _ = (S.make_talos_product if talos else S.make_full_product)(B, max_iters=iters, lib=lib, horizon=horizon)
    finally:
        os.environ.pop("SMPC_FULL_PARTS", None)
        if old is not None:
            os.environ["SMPC_FULL_PARTS"] = old
    if talos:
        gm.generateCycleHorizon(O.walk_cycle())
        gm.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
        X = np.tile(S.talos_random_states(rb, 64, scale=scale), ((B + 63) // 64, 1))[:B]
    else:
        gm.generateCycleHorizon(O.trot_cycle())
        gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
        X = S.random_states(rb, B, scale=scale)
    out = []
    for _ in range(steps):
        gm.iterate(X)
        out.append((gm.xs.copy(), gm.us.copy(), gm.K0.copy(), gm.info.copy()))
        X = gm.xs[:, 1, :].copy()
    return out


def test_emulated_kernels_parts_are_bit_identical(built):
    """Instances are independent: the batch as parts on separate streams (per-part views of every buffer, per-part backtracking lists,
    per-part slices of the derivative kernel's device scratch) changes nothing, bit for bit.  130 instances: parts of 65."""
    a = _parts_run(S.emu_lib(), 1, 130, 2, 2, 6, scale=2.0)
    b = _parts_run(S.emu_lib(), 2, 130, 2, 2, 6, scale=2.0)
    for sa, sb in zip(a, b):
        for u, v in zip(sa, sb):
            assert np.array_equal(u, v)
    assert any((s[3][:, 2] < 1.0).any() for s in a), "the scenario must backtrack (per-part backtracking lists)"


@pytest.mark.gpu
def test_hip_parts_are_bit_identical(built):
    """The same on the device, biped (derivative blocks in device memory), B = 512."""
    a = _parts_run(None, 1, 512, 3, 2, 100, scale=0.7, talos=True)
    for n in (2, 3):
        b = _parts_run(None, n, 512, 3, 2, 100, scale=0.7, talos=True)
        for sa, sb in zip(a, b):
            for u, v in zip(sa, sb):
                assert np.array_equal(u, v)


@pytest.mark.gpu
def test_hip_full_size_properties(built):
    """B = 1024, H = 100 (the batch and horizon of BASELINE's full-dynamics configuration, on the Go2 table): 16 distinct states
    against the oracle, replicas bit-identical, merit descent, finite outputs."""
    B, nd, Hh = 1024, 16, 100
    gm, rb, _, _ = S.make_full_product(B, max_iters=3, horizon=Hh)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    om, _ = S.make_full_oracle(nd, max_iters=3, horizon=Hh)
    Xo = S.random_states(rb, nd, seed=5)
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
    assert np.all(np.isfinite(gm.getContactForces()))


def _interp_checks(lib):
    """interpolate / riccatiFeedback on a full-dynamics handle against the oracle interpolator and the outputs of the solve
    (reference examples/go2_fulldynamics.py:268-292)."""
    B = 3
    gm, rb, _, _ = S.make_full_product(B, max_iters=2, lib=lib)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, B)
    for _ in range(3):
        gm.iterate(X)
        X = gm.xs[:, 1, :].copy()
    xs, us, K0, f = gm.xs, gm.us, gm.K0, gm.getContactForces()
    xd = np.stack([gm.getStateDerivative(0), gm.getStateDerivative(1)], axis=1)
    dt = gm.settings["timestep"]
    for delay in (0.0, 0.004, 0.0099, 0.013):
        x, a, fo = gm.interpolate(delay, knots=2)
        step, s = int(delay / dt), (delay - int(delay / dt) * dt) / dt
        for b in range(B):
            xe = O.interpolate(0, rb.nv, delay, dt, [xs[b, 0], xs[b, 1]])
            assert np.abs(x[b] - xe).max() < 1e-12
            w1 = 1.0 if step >= 1 else s
            assert np.abs(a[b] - (xd[b, 1, rb.nv:] * w1 + xd[b, 0, rb.nv:] * (1 - w1))).max() < 1e-12
            assert np.abs(fo[b] - (f[b, 1] * w1 + f[b, 0] * (1 - w1))).max() < 1e-12
        Xm = np.stack([rb.integrate(xs[b, 0], np.full(rb.ndx, 0.01 * (b + 1))) for b in range(B)])
        u = gm.riccatiFeedback(delay, Xm)
        for b in range(B):
            w1 = 1.0 if step >= 1 else s
            ui = us[b, 1] * w1 + us[b, 0] * (1 - w1)
            ue = ui - K0[b] @ rb.difference(Xm[b], x[b])
            assert np.abs(u[b] - ue).max() < 1e-9 * max(1.0, np.abs(ue).max())
    Ks = gm.Ks
    assert Ks.shape == (B, gm.H, gm.nu, gm.ndx) and np.array_equal(Ks[:, 0], K0)


def test_emulated_kernels_interpolation_and_feedback(built):
    _interp_checks(S.emu_lib())


@pytest.mark.gpu
def test_hip_interpolation_and_feedback(built):
    _interp_checks(None)
