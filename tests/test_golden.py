"""Golden vectors (tests/golden/go2_kino_golden.npz, produced by the oracle with make_golden.py):
the oracle must reproduce them (regression pin), the emulated kernel bodies must match them (CPU tier) and
the HIP path must match them on the GPU box (-m gpu), which has no access to the container they were made in."""
import os

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "go2_kino_golden.npz"))
TOL = 1e-4


def test_oracle_reproduces_stage_vectors():
    rb = O.Robot("go2_like")
    K = O.Kino(rb, O.go2_kino_settings(rb))
    for i, m in enumerate(G["stage_mask"]):
        e = K.eval(int(m), G["stage_u_ref"], rb.x_ref, G["stage_foot_ref"], G["stage_x"][i], G["stage_u"][i])
        d = K.deriv(int(m), G["stage_u_ref"], rb.x_ref, G["stage_foot_ref"], G["stage_x"][i], G["stage_u"][i])
        assert S.rel_err(G["stage%d_xnext" % i], e["xnext"]) < 1e-12
        assert abs(G["stage%d_cost" % i] - e["cost"]) < 1e-9 * abs(e["cost"])
        assert S.rel_err(G["stage%d_c" % i], e["c"]) < 1e-12
        for k in ("A", "B", "lx", "lu", "Lxx", "Lxu", "Luu", "Cx"):
            assert S.rel_err(G["stage%d_%s" % (i, k)], d[k]) < 1e-10, (i, k)


def _run_product(k, lib):
    gm, rb, _, _ = S.make_product(4, max_iters=k, lib=lib)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    cold = gm.xs[0]
    X = G["loop%d_X0" % k].copy()
    for _ in range(8):
        gm.iterate(X)
        X = gm.xs[:, 1, :].copy()
    return gm, cold


def _check(gm, cold, k):
    assert S.rel_err(G["loop%d_cold_xs" % k], cold) < TOL
    assert S.rel_err(G["loop%d_xs" % k], gm.xs) < TOL
    assert S.rel_err(G["loop%d_us" % k], gm.us) < 10 * TOL
    assert S.rel_err(G["loop%d_K0" % k], gm.K0) < TOL  # Ks_[0] is an output of MPC::iterate (reference src/mpc.cpp:216)
    assert np.array_equal(G["loop%d_alpha" % k], gm.info[:, 2])


@pytest.mark.parametrize("k", [1, 3])
def test_oracle_reproduces_closed_loop(k):
    om, rb, _ = S.make_oracle(4, max_iters=k)
    om.generateCycleHorizon(O.trot_cycle())
    om.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = G["loop%d_X0" % k].copy()
    for _ in range(8):
        om.iterate(X)
        X = om.xs[:, 1, :].copy()
    assert S.rel_err(G["loop%d_xs" % k], om.xs) < 1e-7
    assert S.rel_err(G["loop%d_K0" % k], om.K0) < 1e-5


@pytest.mark.parametrize("k", [1, 3])
def test_emulated_kernels_reproduce_closed_loop(built, k):
    gm, cold = _run_product(k, S.emu_lib())
    _check(gm, cold, k)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 3])
def test_hip_reproduces_closed_loop(built, k):
    gm, cold = _run_product(k, None)
    _check(gm, cold, k)
