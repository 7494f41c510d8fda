"""Golden vectors of the full-dynamics path (tests/golden/go2_full_golden.npz, talos_full_golden.npz; made by
make_golden_fulldyn.py with the oracle): the oracle must reproduce them (regression pin), the emulated kernel bodies must match
them (CPU tier) and the HIP path must match them on the GPU box (-m gpu), which cannot see the container they were made in."""
import os

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
G2 = np.load(os.path.join(HERE, "golden", "go2_full_golden.npz"))
GT = np.load(os.path.join(HERE, "golden", "talos_full_golden.npz"))
GC = np.load(os.path.join(HERE, "golden", "go2_full_cone_golden.npz"))
TOL = 1e-4
TALOS_SHORT = dict(horizon=20, cycle=O.walk_cycle(5, 20), mpc_override=dict(T_fly=20, T_contact=5))
TIGHT = dict(mu=0.3, Lfoot=0.05, Wfoot=0.04)


@pytest.mark.parametrize("robot", ["go2_like", "talos_like"])
def test_oracle_reproduces_stage_vectors(robot):
    G, fs = (G2, 3) if robot == "go2_like" else (GT, 6)
    rb = O.Robot(robot)
    full = O.Full(rb, O.go2_full_settings(rb) if fs == 3 else O.talos_full_settings(rb))
    for i, m in enumerate(G["stage_mask"]):
        args = (int(m), G["stage_u_ref"], rb.x_ref, G["stage_foot_ref"], G["stage_x"][i], G["stage_u"][i])
        e, d = full.eval(*args), full.deriv(*args)
        f = rb.full_forward_dynamics(G["stage_x"][i], G["stage_u"][i], int(m), full.s["Kp_correction"], full.s["Kd_correction"], fs=fs)
        assert S.rel_err(G["stage%d_xnext" % i], e["xnext"]) < 1e-12 and S.rel_err(G["stage%d_c" % i], e["c"]) < 1e-10
        assert abs(G["stage%d_cost" % i] - e["cost"]) < 1e-9 * abs(e["cost"])
        assert S.rel_err(G["stage%d_a" % i], f["a"]) < 1e-10 and S.rel_err(G["stage%d_lam" % i], f["lam"]) < 1e-10
        for k in ("A", "B", "lx", "lu", "Lxx", "Lxu", "Luu", "Cx", "Cu"):
            assert S.rel_err(G["stage%d_%s" % (i, k)], d[k]) < 1e-9, (i, k)


def _go2(k, lib):
    gm, rb, _, _ = S.make_full_product(2, max_iters=k, lib=lib)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    cold = gm.xs[0]
    X = G2["loop%d_X0" % k].copy()
    for _ in range(8):
        gm.iterate(X)
        X = gm.xs[:, 1, :].copy()
    t = "loop%d" % k
    assert S.rel_err(G2[t + "_cold_xs"], cold) < TOL
    assert S.rel_err(G2[t + "_xs"], gm.xs) < TOL and S.rel_err(G2[t + "_us"], gm.us) < 10 * TOL and S.rel_err(G2[t + "_K0"], gm.K0) < TOL
    assert np.array_equal(G2[t + "_alpha"], gm.info[:, 2])


def _talos(tag, lib):
    over, walk = (None, (0.1, 0, 0, 0, 0, 0)) if tag == "loop" else (TIGHT, (0.2, 0.1, 0, 0, 0, 0.2))
    gm, rb, _, _ = S.make_talos_product(2, max_iters=2, lib=lib, horizon=20, settings_override=over, mpc_override=TALOS_SHORT["mpc_override"])
    gm.generateCycleHorizon(TALOS_SHORT["cycle"])
    gm.switchToWalk(np.array(walk, float))
    cold = gm.xs[0]
    X = GT[tag + "_X0"].copy()
    for _ in range(6):
        gm.iterate(X)
        X = gm.xs[:, 1, :].copy()
    assert S.rel_err(GT[tag + "_cold_xs"], cold) < TOL
    assert S.rel_err(GT[tag + "_xs"], gm.xs) < TOL and S.rel_err(GT[tag + "_us"], gm.us) < 10 * TOL and S.rel_err(GT[tag + "_K0"], gm.K0) < TOL
    assert np.array_equal(GT[tag + "_alpha"], gm.info[:, 2])
    assert S.rel_err(GT[tag + "_vs"], gm.vs) < 1e-3
    if tag == "cone":
        assert (np.abs(gm.vs[:, :, 2 * gm.nu :]) > 0).sum() >= 20


@pytest.mark.parametrize("k", [1, 3])
def test_emulated_kernels_reproduce_go2_closed_loop(built, k):
    _go2(k, S.emu_lib())


@pytest.mark.parametrize("tag", ["loop", "cone"])
def test_emulated_kernels_reproduce_talos_closed_loop(built, tag):
    _talos(tag, S.emu_lib())


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 3])
def test_hip_reproduces_go2_closed_loop(built, k):
    _go2(k, None)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["loop", "cone"])
def test_hip_reproduces_talos_closed_loop(built, tag):
    _talos(tag, None)


def _go2_cone(lib):
    """Go2 with friction-pyramid rows (force_cone, 3-D feet): replay of the committed closed loop"""
    gm, rb, _, _ = S.make_full_product(2, 2, lib, 20, settings_override={"force_cone": True, "mu": 0.6})
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.3, 0.1, 0, 0, 0, 0.2]))
    assert S.rel_err(GC["cone_cold_xs"], gm.xs[0]) < TOL
    X = GC["cone_X0"]
    for _ in range(6):
        gm.iterate(X)
        X = gm.xs[:, 1, :].copy()
    assert S.rel_err(GC["cone_xs"], gm.xs) < TOL and S.rel_err(GC["cone_us"], gm.us) < 10 * TOL
    assert np.array_equal(GC["cone_vs"][:, :, 24:] != 0, gm.vs[:, :, 24:] != 0)  # the same rows are active


def test_emulated_kernels_reproduce_go2_cone_closed_loop(built):
    _go2_cone(S.emu_lib())


@pytest.mark.gpu
def test_hip_reproduces_go2_cone_closed_loop(built):
    _go2_cone(None)
