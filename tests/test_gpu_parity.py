"""GPU parity tests: the HIP path (through the C ABI, libsmpc_hip.so) against the CPU oracle on the same
seeded inputs.  Tolerance (north_star): <= 1e-4 relative state-trajectory error."""
import numpy as np
import pytest

import mpc_setup as S

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _loaded_native():
    with open("/proc/self/maps") as f:
        return "libsmpc_hip.so" in f.read()


def test_native_library_is_the_one_loaded(built):
    gm, rb, _, _ = S.make_product(1)
    assert _loaded_native(), "HIP extension not loaded"
    assert gm._lib.L.smpc_device_count() >= 1


def test_cold_solve_matches_oracle(built):
    om, gm, rb = S.make_pair(batch=3)
    assert len(om.cold_trace()) == len(gm.cold_trace())
    assert S.rel_err(om.xs, gm.xs) < TOL
    assert S.rel_err(om.us, gm.us) < TOL


@pytest.mark.parametrize("iters", [1, 3])
def test_closed_loop_parity(built, iters):
    """48 control steps: the window covers the first swing phase entering the horizon, a take-off (step 10), the swing
    and the touch-down of FL / RR (step 40) of the trot cycle (reference src/mpc.cpp:220-254)."""
    B = 8
    om, gm, rb = S.make_pair(batch=B, max_iters=iters)
    X = S.random_states(rb, B)
    worst = worst_k = 0.0
    for step in range(48):
        om.iterate(X)
        gm.iterate(X)
        worst = max(worst, S.rel_err(om.xs, gm.xs))
        assert S.rel_err(om.xs, gm.xs) < TOL, (step, S.rel_err(om.xs, gm.xs))
        assert S.rel_err(om.us, gm.us) < TOL * 10
        assert np.array_equal(om.info[:, 2], gm.info[:, 2]), "line-search step sizes differ"
        assert S.rel_err(om.foot_refs, gm.getReferencePoses()) < 1e-12
        # the first-stage feedback gain Ks_[0] (an output of MPC::iterate, reference src/mpc.cpp:216)
        ek = S.rel_err(om.K0, gm.K0)
        worst_k = max(worst_k, ek)
        assert ek < TOL, (step, ek)
        X = om.xs[:, 1, :].copy()
    # the compared horizon holds a take-off, a whole swing phase and the touch-down that follows it
    seq = [tuple(gm.ocp_handler.getContactState(t)) for t in range(gm.H)]
    sw = [i for i, m in enumerate(seq) if not all(m)]
    assert sw and sw[0] > 0 and sw[-1] < gm.H - 1 and all(seq[sw[-1] + 1]), seq
    Ks = gm.Ks
    assert S.rel_err(om.K0, Ks[:, 0]) < TOL and np.all(np.isfinite(Ks))
    for f in range(4):
        assert om.timing(f, 0) == gm.foot_takeoff_times[S.FEET[f]]
        assert om.timing(f, 1) == gm.foot_land_times[S.FEET[f]]
    print("worst relative error over the run: xs %.3e, K0 %.3e" % (worst, worst_k))


def test_line_search_backtracking_matches(built):
    om, gm, rb = S.make_pair(8, max_iters=2)
    X = S.random_states(rb, 8, seed=3, scale=4.0)
    seen = False
    for _ in range(3):
        om.iterate(X)
        gm.iterate(X)
        assert np.array_equal(om.info[:, 2], gm.info[:, 2])
        seen |= bool((om.info[:, 2] < 1.0).any())
        assert S.rel_err(om.xs, gm.xs) < TOL
        X = om.xs[:, 1, :].copy()
    assert seen


def test_dense_weight_matrices(built):
    """General (non-diagonal) w_x / w_u: the kernels' dense-weight path."""
    import oracle_lib as O

    rb = O.Robot("go2_like")
    s0 = O.go2_kino_settings(rb)
    rng = np.random.default_rng(5)

    def couple(w, eps):
        w = np.array(w, float)
        d = np.sqrt(np.abs(np.diag(w)))
        m = rng.standard_normal(w.shape)
        return w + eps * np.outer(d, d) * (m + m.T) / 2

    over = dict(w_x=couple(s0["w_x"], 0.05), w_u=couple(s0["w_u"], 0.05))
    om, gm, rb = S.make_pair(4, max_iters=2, settings_override=over)
    X = S.random_states(rb, 4)
    for _ in range(3):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < TOL
        X = om.xs[:, 1, :].copy()


@pytest.mark.parametrize("horizon", [3, 65])
def test_unusual_horizons(built, horizon):
    om, gm, rb = S.make_pair(5, max_iters=2, horizon=horizon)
    X = S.random_states(rb, 5)
    for _ in range(3):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < TOL
        assert np.array_equal(om.info[:, 2], gm.info[:, 2])
        X = om.xs[:, 1, :].copy()


def test_stage_knots_match_oracle(built):
    om, gm, rb = S.make_pair(batch=2)
    om.keep_knots()
    X = S.random_states(rb, 2)
    om.iterate(X)
    gm.iterate(X)
    for t in (0, 1, 17, 48, 49):
        ko, kg = om.knot(1, t), gm.debug_lq(1, t)
        for k in ("A", "B", "Q", "S", "R", "C", "f", "d"):
            assert S.rel_err(ko[k], kg[k]) < 1e-8, (t, k)
        for k in ("q", "r"):  # Lagrangian gradients: differences of O(|multiplier|) terms amplify the 1e-10 iterate gap
            assert S.rel_err(ko[k], kg[k]) < 1e-6, (t, k)


def test_full_size_properties(built):
    """BASELINE size (B=4096, H=50, 3 iterations): size-independent properties instead of an oracle run:
    identical instances give identical trajectories, dynamics defects and contact constraints are closed,
    the step is a descent step for every instance."""
    B = 4096
    gm, rb, _, _ = S.make_product(B, max_iters=3)
    import oracle_lib as O

    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, 64)
    X = np.tile(X, (B // 64, 1))  # 64 distinct states, each repeated 64 times
    for _ in range(3):
        gm.iterate(X)
        xs = gm.xs
        X = xs[:, 1, :].copy()
    us, K0 = gm.us, gm.K0
    xs = xs.reshape(B // 64, 64, *xs.shape[1:])
    assert np.abs(xs - xs[0:1]).max() == 0.0, "replicated instances must be bit-identical"
    # the 64 distinct instances against the oracle (same three control steps)
    om, _, _ = S.make_oracle(64, max_iters=3)
    om.generateCycleHorizon(O.trot_cycle())
    om.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    Xo = S.random_states(rb, 64)
    for _ in range(3):
        om.iterate(Xo)
        Xo = om.xs[:, 1, :].copy()
    assert S.rel_err(om.xs, xs[0]) < TOL and S.rel_err(om.us, us[:64]) < 10 * TOL and S.rel_err(om.K0, K0[:64]) < TOL
    info = gm.info
    assert np.all(np.isfinite(info))
    assert np.all(info[:, 8] < 1e-2), "primal infeasibility after the step"
    assert np.all(info[:, 1] < 0), "merit directional derivative must be negative"
    assert np.all(info[:, 3] <= info[:, 0]), "merit must not increase"


def test_long_closed_loop_walk_is_stable_and_deterministic(built):
    """300 control steps of the trot at 0.2 m/s, closed on the solver's own prediction plus noise: every
    output stays finite, the constraints stay closed, the robot advances, and a second engine fed the same inputs
    reproduces the trajectory bit for bit."""
    import oracle_lib as O

    B, steps = 32, 300
    runs = []
    for _ in range(2):
        gm, rb, _, _ = S.make_product(B, max_iters=3)
        gm.generateCycleHorizon(O.trot_cycle())
        gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
        rng = np.random.default_rng(7)
        X = S.random_states(rb, B)
        prims = []
        for _ in range(steps):
            gm.iterate(X)
            info = gm.info
            assert np.all(np.isfinite(info))
            prims.append(float(info[:, 8].max()))
            X = gm.xs[:, 1, :] + rng.normal(0.0, 1e-3, (B, gm.nx))
            X[:, 3:7] /= np.linalg.norm(X[:, 3:7], axis=1, keepdims=True)
        runs.append((gm.xs.copy(), gm.us.copy(), np.array(prims)))
    xs, us, prims = runs[0]
    # primal infeasibility after 3 iterations: small in the bulk of the gait; at touch-down the newly activated
    # foot-velocity rows start from the swing speed (< 1 m/s) and are closed over the following control steps
    print("primal infeasibility: median %.2e, 90%% %.2e, max %.2e" % (np.median(prims), np.quantile(prims, 0.9), prims.max()))
    assert np.median(prims) < 5e-2 and prims.max() < 2.0
    assert np.all(xs[:, 0, 0] > 0.0) and xs[:, 0, 0].mean() > 0.15, "the base should advance under the 0.2 m/s command"
    assert np.all(np.abs(xs[:, 0, 2] - rb.x_ref[2]) < 0.1), "the base height should stay near the reference posture"
    assert np.array_equal(xs, runs[1][0]) and np.array_equal(us, runs[1][1])


def test_per_instance_velocity_commands(built):
    B = 4
    V = np.array([[0.3, 0, 0, 0, 0, 0], [0.0, 0.2, 0, 0, 0, 0.4], [-0.2, 0.1, 0, 0, 0, -0.3], [0, 0, 0, 0, 0, 0.0]])
    om, gm, rb = S.make_pair(B, 3)
    om.setVelocityBaseBatched(V)
    gm.setVelocityBaseBatched(V)
    X = S.random_states(rb, B)
    for _ in range(6):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < TOL and S.rel_err(om.us, gm.us) < 10 * TOL
        assert S.rel_err(om.foot_refs, gm.getReferencePoses()) < 1e-12
        X = om.xs[:, 1, :].copy()
    refs = gm.getReferencePoses()
    assert np.abs(refs[0] - refs[1]).max() > 1e-3


def test_per_gpu_size_of_the_sharded_configuration(built):
    """B = 8192 on one GPU = the per-GPU share of BASELINE's 65536-instance, 8-GPU configuration: 32 distinct states against the
    oracle, replicas bit-identical."""
    import oracle_lib as O

    B, nd = 8192, 32
    gm, rb, _, _ = S.make_product(B, max_iters=3)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    om, _, _ = S.make_oracle(nd, max_iters=3)
    om.generateCycleHorizon(O.trot_cycle())
    om.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    Xo = S.random_states(rb, nd, seed=11)
    X = np.tile(Xo, (B // nd, 1))
    for _ in range(2):
        gm.iterate(X)
        om.iterate(Xo)
        xs = gm.xs
        X = xs[:, 1, :].copy()
        Xo = om.xs[:, 1, :].copy()
    xs = xs.reshape(B // nd, nd, *xs.shape[1:])
    assert np.abs(xs - xs[0:1]).max() == 0.0
    assert S.rel_err(om.xs, xs[0]) < TOL
    assert np.all(np.isfinite(gm.info))


def test_multi_stream_iterations_are_bit_identical(built, tmp_path):
    """SMPC_STREAMS=n runs the iterations of n parts of the batch on n streams: every instance's arithmetic is unchanged, so
    the trajectories must be those of the single-stream run, bit for bit -- including an instance that backtracks."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, mpc_setup as S, oracle_lib as O\n"
        "gm, rb, _, _ = S.make_product(300, max_iters=3)\n"
        "gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.1]))\n"
        "X = S.random_states(rb, 300, scale=2.0)\n"
        "for _ in range(4):\n"
        "    gm.iterate(X); X = gm.xs[:, 1, :].copy()\n"
        "np.savez(sys.argv[1], xs=gm.xs, us=gm.us, info=gm.info)\n" % (root, os.path.join(root, "tests"))
    )
    outs = []
    for n in ("1", "3"):
        out = str(tmp_path / ("s%s.npz" % n))
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, SMPC_STREAMS=n), timeout=900)
        outs.append(np.load(out))
    assert np.array_equal(outs[0]["xs"], outs[1]["xs"]) and np.array_equal(outs[0]["us"], outs[1]["us"])
    assert np.array_equal(outs[0]["info"][:, 2], outs[1]["info"][:, 2]) and outs[0]["info"][:, 2].min() < 1.0
