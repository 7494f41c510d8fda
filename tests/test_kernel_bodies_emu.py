"""CPU tier: the HIP kernel bodies of simple-mpc_amd/csrc, re-compiled with the sequential-lane test backend
(tests/emu/smpc_backend.h), against the oracle.  This checks the numerics and the phase/barrier structure of
the shipped kernels without a GPU; the same comparisons run on the real device in test_gpu_parity.py.
The emulation library is test infrastructure: the product path never loads it."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

TOL = 1e-4  # north_star: relative state-trajectory error


@pytest.fixture(scope="module")
def lib(built):
    return S.emu_lib()


def test_cold_solve_matches_oracle(lib):
    om, gm, rb = S.make_pair(2, lib=lib)
    assert len(om.cold_trace()) == len(gm.cold_trace())
    assert S.rel_err(om.cold_trace(), gm.cold_trace()) < 1e-6
    assert S.rel_err(om.xs, gm.xs) < 1e-9
    assert S.rel_err(om.us, gm.us) < 1e-9
    # both instances are copies of the single cold solve
    assert np.array_equal(gm.xs[0], gm.xs[1])


@pytest.mark.parametrize("iters", [1, 3])
def test_closed_loop_parity(lib, iters):
    B = 3
    om, gm, rb = S.make_pair(B, max_iters=iters, lib=lib)
    X = S.random_states(rb, B)
    for step in range(6):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < 1e-7, step
        assert S.rel_err(om.us, gm.us) < 1e-7
        assert S.rel_err(om.K0, gm.K0) < 1e-6
        assert np.array_equal(om.info[:, 2], gm.info[:, 2])
        assert S.rel_err(om.info[:, :2], gm.info[:, :2]) < 1e-6  # phi0, dphi0
        assert S.rel_err(om.foot_refs, gm.getReferencePoses()) < 1e-13
        xd = np.stack([gm.getStateDerivative(0), gm.getStateDerivative(1)], 1)
        assert S.rel_err(om.xdot[:, :2], xd) < 1e-6
        X = om.xs[:, 1, :].copy()
    for f in range(4):
        assert om.timing(f, 0) == gm.foot_takeoff_times[S.FEET[f]]
        assert om.timing(f, 1) == gm.foot_land_times[S.FEET[f]]


@pytest.mark.parametrize("horizon", [2, 7, 65])
def test_unusual_horizons(lib, horizon):
    """Shortest horizon, an odd one, and one with more than 64 nodes (lane = node in the apply kernel)."""
    om, gm, rb = S.make_pair(2, max_iters=2, lib=lib, horizon=horizon)
    X = S.random_states(rb, 2)
    for _ in range(4):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < 1e-7
        assert np.array_equal(om.info[:, 2], gm.info[:, 2])
        X = om.xs[:, 1, :].copy()


def test_stage_knots_match_oracle(lib):
    om, gm, rb = S.make_pair(2, lib=lib)
    om.keep_knots()
    X = S.random_states(rb, 2)
    for _ in range(2):
        om.iterate(X)
        gm.iterate(X)
        X = om.xs[:, 1, :].copy()
    for t in (0, 1, 9, 10, 25, 48, 49):  # includes contact-switch stages of the trot
        ko, kg = om.knot(1, t), gm.debug_lq(1, t)
        for k in ("A", "B", "Q", "S", "R", "C", "f", "d"):
            assert S.rel_err(ko[k], kg[k]) < 1e-8, (t, k)
        for k in ("q", "r"):  # Lagrangian gradients: differences of O(|multiplier|) terms amplify the 1e-10 iterate gap
            assert S.rel_err(ko[k], kg[k]) < 1e-6, (t, k)


def test_walk_to_stand_and_swing_references(lib):
    """Longer run through take-off / landing and switchToStand (reference src/mpc.cpp:220-254,382-392)."""
    B = 2
    om, gm, rb = S.make_pair(B, max_iters=1, lib=lib)
    X = S.random_states(rb, B, scale=0.3)
    for step in range(40):
        if step == 30:
            om.switchToStand()
            gm.switchToStand()
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.foot_refs, gm.getReferencePoses()) < 1e-12
        X = om.xs[:, 1, :].copy()
    assert S.rel_err(om.xs, gm.xs) < TOL
    # the swing references leave the ground once the landing time is inside the fly window
    assert gm.getReferencePoses()[:, :, :, 2].max() > 0.05


def test_joint_limit_rows_activate(lib):
    """Box rows (reference src/kinodynamics.cpp:91-101): tighten the limits so the AL projection is active."""
    rb = O.Robot("go2_like")
    q = rb.q_ref[7:]
    over = dict(qmin=q - 0.02, qmax=q + 0.02)
    om, gm, rb = S.make_pair(2, lib=lib, settings_override=over)
    X = S.random_states(rb, 2, scale=0.5)
    for _ in range(3):
        om.iterate(X)
        gm.iterate(X)
        X = om.xs[:, 1, :].copy()
    assert np.abs(om.vs[:, :, :12]).max() > 1.0  # multipliers of the box rows are in play
    assert S.rel_err(om.xs, gm.xs) < TOL
    assert S.rel_err(om.vs, gm.vs) < 1e-4


def test_dense_weight_matrices(lib):
    """Non-diagonal (symmetric) w_x / w_u take the general weight path of the kernels (the settings of record are
    diagonal and take the fast path)."""
    rb = O.Robot("go2_like")
    s0 = O.go2_kino_settings(rb)
    rng = np.random.default_rng(5)

    def couple(w, eps):
        w = np.array(w, float)
        d = np.sqrt(np.abs(np.diag(w)))
        m = rng.standard_normal(w.shape)
        return w + eps * np.outer(d, d) * (m + m.T) / 2

    over = dict(w_x=couple(s0["w_x"], 0.05), w_u=couple(s0["w_u"], 0.05))
    om, gm, rb = S.make_pair(2, max_iters=2, lib=lib, settings_override=over)
    X = S.random_states(rb, 2)
    for _ in range(3):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < 1e-7
        assert S.rel_err(om.K0, gm.K0) < 1e-6
        X = om.xs[:, 1, :].copy()


def test_line_search_backtracking_matches(lib):
    """Large perturbations make some instances reject alpha = 1: the speculative line search (alpha = 1 for all,
    then 2^-1..2^-9 only for the undecided ones) must pick the same step as sequential backtracking."""
    om, gm, rb = S.make_pair(4, max_iters=2, lib=lib)
    X = S.random_states(rb, 4, seed=3, scale=4.0)
    seen_backtrack = False
    for _ in range(3):
        om.iterate(X)
        gm.iterate(X)
        assert np.array_equal(om.info[:, 2], gm.info[:, 2])
        seen_backtrack |= bool((om.info[:, 2] < 1.0).any())
        assert S.rel_err(om.xs, gm.xs) < 1e-7
        X = om.xs[:, 1, :].copy()
    assert seen_backtrack, "scenario no longer exercises backtracking"


def test_dense_and_structured_riccati_agree(built):
    """The model-independent dense sweep (SMPC_RICCATI=dense) and the kinodynamics-structured sweep solve the same
    KKT system: identical trajectories up to round-off."""
    code = (
        "import sys; sys.path.insert(0, %r); import numpy as np, mpc_setup as S\n"
        "gm, rb, _, _ = S.make_product(2, max_iters=2, lib=S.emu_lib())\n"
        "import oracle_lib as O\n"
        "gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2,0,0,0,0,0.]))\n"
        "X = S.random_states(rb, 2)\n"
        "for _ in range(3):\n"
        "    gm.iterate(X); X = gm.xs[:,1,:].copy()\n"
        "np.savez(sys.argv[1], xs=gm.xs, K0=gm.K0)\n" % os.path.dirname(os.path.abspath(__file__))
    )
    outs = []
    for mode in ("dense", "kino"):
        path = "/tmp/smpc_emu_riccati_%s.npz" % mode
        env = dict(os.environ, SMPC_RICCATI=mode)
        subprocess.check_call([sys.executable, "-c", code, path], env=env)
        outs.append(np.load(path))
    assert S.rel_err(outs[0]["xs"], outs[1]["xs"]) < 1e-7
    assert S.rel_err(outs[0]["K0"], outs[1]["K0"]) < 1e-6


def test_speculative_line_search_is_the_sequential_algorithm(built):
    """Tentative full step + next derivative pass as the alpha = 1 trial (run_iterations) against explicit trial
    evaluations in every iteration (SMPC_NO_SPECULATIVE_LS): bit-identical iterates, on a scenario that rejects alpha = 1
    in early and late iterations (restore / backtrack / re-derive path)."""
    code = (
        "import sys; sys.path.insert(0, %r); import numpy as np, mpc_setup as S\n"
        "import oracle_lib as O\n"
        "gm, rb, _, _ = S.make_product(6, max_iters=3, lib=S.emu_lib())\n"
        "gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2,0,0,0,0,0.]))\n"
        "X = S.random_states(rb, 6, seed=3, scale=4.0)\n"
        "al = []\n"
        "for _ in range(3):\n"
        "    gm.iterate(X); X = gm.xs[:,1,:].copy(); al.append(gm.info[:,2].copy())\n"
        "np.savez(sys.argv[1], xs=gm.xs, us=gm.us, al=np.array(al))\n" % os.path.dirname(os.path.abspath(__file__))
    )
    outs = []
    for mode in ("spec", "seq"):
        path = "/tmp/smpc_emu_ls_%s.npz" % mode
        env = dict(os.environ)
        env.pop("SMPC_NO_SPECULATIVE_LS", None)
        if mode == "seq":
            env["SMPC_NO_SPECULATIVE_LS"] = "1"
        subprocess.check_call([sys.executable, "-c", code, path], env=env)
        outs.append(np.load(path))
    assert (outs[0]["al"] < 1.0).any(), "scenario no longer exercises backtracking"
    assert np.array_equal(outs[0]["al"], outs[1]["al"])
    assert np.array_equal(outs[0]["xs"], outs[1]["xs"])
    assert np.array_equal(outs[0]["us"], outs[1]["us"])


def test_batch_instances_are_independent(lib):
    gm4, rb, _, _ = S.make_product(4, lib=lib)
    gm1, _, _, _ = S.make_product(1, lib=lib)
    for g in (gm4, gm1):
        g.generateCycleHorizon(O.trot_cycle())
        g.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, 4)
    gm4.iterate(X)
    gm1.iterate(X[2:3])
    assert np.array_equal(gm4.xs[2], gm1.xs[0])


def test_lane_order_independence(built):
    """Run the same closed loop with lanes executed in descending order: a missing phase barrier (a lane
    reading what a later lane writes in the same phase) changes the result in one of the two orders."""
    code = (
        "import sys; sys.path.insert(0, %r); import numpy as np, mpc_setup as S\n"
        "gm, rb, _, _ = S.make_product(2, max_iters=2, lib=S.emu_lib())\n"
        "import oracle_lib as O\n"
        "gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array([0.2,0,0,0,0,0.]))\n"
        "X = S.random_states(rb, 2)\n"
        "for _ in range(3):\n"
        "    gm.iterate(X); X = gm.xs[:,1,:].copy()\n"
        "np.save(sys.argv[1], gm.xs)\n" % os.path.dirname(os.path.abspath(__file__))
    )
    outs = []
    for rev in ("0", "1"):
        path = "/tmp/smpc_emu_order_%s.npy" % rev
        env = dict(os.environ, SMPC_EMU_REVERSE=rev)
        subprocess.check_call([sys.executable, "-c", code, path], env=env)
        outs.append(np.load(path))
    assert np.array_equal(outs[0], outs[1])


def test_reference_gait_kat_through_the_c_abi(lib):
    """reference tests/mpc.cpp:78-90 on the product's own timer: feet 0/1 carry the left/right patterns of the
    reference test (H = 100); 10 iterate() calls shift every entry by 10."""
    cs = np.array([[1, 1, 1, 1]] * 10 + [[1, 0, 1, 0]] * 50 + [[1, 1, 1, 1]] * 10 + [[0, 1, 0, 1]] * 50, np.uint8)
    gm, rb, _, _ = S.make_product(1, lib=lib, horizon=100, mpc_override=dict(T_fly=80, T_contact=20))
    gm.generateCycleHorizon(cs)
    to, ld = gm.foot_takeoff_times, gm.foot_land_times
    assert (to["FL_foot"][0], to["FR_foot"][0], ld["FL_foot"][0], ld["FR_foot"][0]) == (170, 110, 219, 160)
    assert len(gm.xs[0]) == 101 and len(gm.us[0]) == 100  # reference tests/mpc.cpp:43-44
    x = rb.x_ref
    for _ in range(10):
        gm.iterate(x[None, :])
    to, ld = gm.foot_takeoff_times, gm.foot_land_times
    assert (to["FL_foot"][0], to["FR_foot"][0], ld["FL_foot"][0], ld["FR_foot"][0]) == (160, 100, 209, 150)


def test_per_instance_velocity_commands(lib):
    """smpc_set_velocity_base_batched: every instance follows its own command (Raibert footholds and the velocity part of
    the state targets) exactly as a one-instance MPC given that command would."""
    B = 3
    V = np.array([[0.3, 0, 0, 0, 0, 0], [0.0, 0.2, 0, 0, 0, 0.4], [-0.2, 0.1, 0, 0, 0, -0.3]])
    om, gm, rb = S.make_pair(B, 2, lib=lib)
    om.setVelocityBaseBatched(V)
    gm.setVelocityBaseBatched(V)
    X = S.random_states(rb, B)
    singles = []
    for b in range(B):
        g1, _, _, _ = S.make_product(1, 2, lib=lib)
        g1.generateCycleHorizon(O.trot_cycle())
        g1.switchToWalk(V[b])
        singles.append(g1)
    for _ in range(4):
        om.iterate(X)
        gm.iterate(X)
        for b in range(B):
            singles[b].iterate(X[b : b + 1])
        assert S.rel_err(om.xs, gm.xs) < 1e-8 and S.rel_err(om.us, gm.us) < 1e-7
        assert S.rel_err(om.foot_refs, gm.getReferencePoses()) < 1e-12
        X = om.xs[:, 1, :].copy()
    for b in range(B):
        assert np.array_equal(singles[b].xs[0], gm.xs[b]), "instance %d must equal the one-instance MPC with its command" % b
    refs = gm.getReferencePoses()
    assert np.abs(refs[0] - refs[1]).max() > 1e-3, "different commands must give different footholds"
    with pytest.raises(RuntimeError):
        gm.setVelocityBaseBatched(np.zeros((B, 5)))


def test_without_kinematics_limits_and_other_settings(lib):
    """The settings of the reference's own CPU benchmark (benchmark/go2.cpp:60-101: no joint box, zero centroidal weight,
    lighter frame weight, T_contact = 5, apex 0.2) on the kinodynamics OCP."""
    ov = dict(kinematics_limits=False, w_cent=np.zeros((6, 6)), w_frame=np.eye(3) * 1000.0)
    mo = dict(T_contact=5, swing_apex=0.2)
    om, rb, _ = S.make_oracle(2, 2, settings_override=ov, mpc_override=mo)
    gm, _, _, _ = S.make_product(2, 2, lib=lib, settings_override=ov, mpc_override=mo)
    cs = O.trot_cycle(T_ds=5, T_ss=30)
    for m in (om, gm):
        m.generateCycleHorizon(cs)
        m.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, 2)
    for _ in range(5):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < 1e-8 and S.rel_err(om.us, gm.us) < 1e-6
        assert np.array_equal(om.info[:, 2], gm.info[:, 2])
        X = om.xs[:, 1, :].copy()
    assert np.abs(gm.vs[:, :, :12]).max() == 0.0, "no joint-box multipliers without kinematics_limits"
