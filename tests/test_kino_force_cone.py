"""force_cone of the kinodynamics OCP with 3-D feet: CentroidalFrictionConeResidual(ndx, nu, foot, mu, 1e-4) in NegativeOrthant for every
foot in contact (reference src/kinodynamics.cpp:124-129).  The kernels eliminate these rows (they act on u only): R^ += D^T D / mu,
r^ += D^T d / mu, while the oracle solves the stage KKT with explicit multipliers; with mu = 1e-8 the two agree to about 1e-7 once
rows are active (the oracle's own fold switch reproduces that figure), and to rounding while none is."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONE = {"force_cone": True, "mu": 0.1}
WALK = (0.6, 0.4, 0, 0, 0, 0.5)  # a sideways, turning trot: tangential forces reach the (narrow) cone


def test_oracle_cone_rows_and_jacobian():
    rb = O.Robot("go2_like")
    s = O.go2_kino_settings(rb)
    s.update(CONE)
    K = O.Kino(rb, s)
    assert K.nc == 24 + 8  # joint box 12 + contact velocity 12 + 2 cone rows per foot (tests/problem.cpp-style block count)
    rng = np.random.default_rng(1)
    x = S.random_states(rb, 1)[0]
    u = np.r_[rng.normal(0, 5, 12) + np.tile([0, 0, 40.0], 4), rng.normal(0, 1, 12)]
    foot = np.zeros((4, 3))
    ev = K.eval(0b0101, u, rb.x_ref, foot, x, u)
    c = ev["c"][24:].reshape(4, 2)
    f = u[:12].reshape(4, 3)
    for k in range(4):
        on = (0b0101 >> k) & 1
        want = [-f[k, 2] + 1e-4, f[k, 0] ** 2 + f[k, 1] ** 2 - 0.01 * f[k, 2] ** 2] if on else [0, 0]
        assert np.allclose(c[k], want)
    d = K.deriv(0b0101, u, rb.x_ref, foot, x, u)
    eps, J = 1e-6, np.zeros((8, K.nu))
    for i in range(K.nu):
        du = np.zeros(K.nu)
        du[i] = eps
        J[:, i] = (K.eval(0b0101, u, rb.x_ref, foot, x, u + du)["c"][24:] - K.eval(0b0101, u, rb.x_ref, foot, x, u - du)["c"][24:]) / (2 * eps)
    assert np.abs(J - d["Cu"][24:]).max() < 1e-6 and np.abs(d["Cx"][24:]).max() == 0.0


def test_oracle_fold_switch_bounds_the_elimination_error():
    outs = []
    for fold in (0, 1):
        O.lib().orc_set_fold_u_rows(fold)
        try:
            om, rb, _ = S.make_oracle(1, max_iters=2, horizon=20, settings_override=CONE)
            om.generateCycleHorizon(O.trot_cycle())
            om.switchToWalk(np.array(WALK))
            X = S.random_states(rb, 2)[1:]
            for _ in range(8):
                om.iterate(X)
                X = om.xs[:, 1, :].copy()
            outs.append((om.xs.copy(), int((om.vs[:, :, 24:] != 0).sum())))
        finally:
            O.lib().orc_set_fold_u_rows(0)
    assert outs[0][1] >= 10, "the scenario must activate cone rows"
    assert 0 < S.rel_err(outs[0][0], outs[1][0]) < 1e-5


def _loop(om, gm, X, n, tol):
    active = 0
    for _ in range(n):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < tol and S.rel_err(om.us, gm.us) < 10 * tol
        assert S.alphas_agree(om, gm, rtol=1e-6)
        active = max(active, int((om.vs[:, :, 24:] != 0).sum()))
        X = om.xs[:, 1, :].copy()
    return active


def test_emulated_kernels_with_active_cones(built):
    om, gm, rb = S.make_pair(2, max_iters=2, lib=S.emu_lib(), horizon=20, settings_override=CONE, walk=WALK)
    assert _loop(om, gm, S.random_states(rb, 2), 8, 1e-5) >= 10  # (instance 1 backtracks to alpha = 1 / 128 on the way)


def test_emulated_kernels_inactive_cones_change_nothing(built, monkeypatch):
    # a gentle forward trot never reaches a cone of mu = 0.8: the result is the cone-free one, bit for bit -- of the same kernels (problems
    # with optional constraint blocks run on the one-kernel stage path, SMPC_LANE_EVAL=0 puts the cone-free problem there too)
    monkeypatch.setenv("SMPC_LANE_EVAL", "0")
    g0, rb, _, _ = S.make_product(1, 2, lib=S.emu_lib(), horizon=20)
    monkeypatch.delenv("SMPC_LANE_EVAL")
    g1, _, _, _ = S.make_product(1, 2, lib=S.emu_lib(), horizon=20, settings_override={"force_cone": True, "mu": 0.8})
    X = S.random_states(rb, 1)
    for g in (g0, g1):
        g.generateCycleHorizon(O.trot_cycle())
        g.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    for _ in range(5):
        g0.iterate(X)
        g1.iterate(X)
        assert np.array_equal(g0.xs, g1.xs) and np.array_equal(g0.us, g1.us)
        X = g0.xs[:, 1, :].copy()


def test_emulated_kernels_are_lane_order_independent_with_cones(built, tmp_path):
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, mpc_setup as S, oracle_lib as O\n"
        "gm, rb, _, _ = S.make_product(2, max_iters=2, lib=S.emu_lib(), horizon=20, settings_override=%r)\n"
        "gm.generateCycleHorizon(O.trot_cycle()); gm.switchToWalk(np.array(%r))\n"
        "X = S.random_states(rb, 2)\n"
        "for _ in range(6):\n"
        "    gm.iterate(X); X = gm.xs[:, 1, :].copy()\n"
        "np.save(sys.argv[1], gm.xs)\n" % (ROOT, os.path.join(ROOT, "tests"), CONE, list(WALK))
    )
    outs = []
    for rev in ("0", "1"):
        out = str(tmp_path / ("xs%s.npy" % rev))
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, SMPC_EMU_REVERSE=rev), timeout=900)
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1])


def test_checkpoint_keeps_the_cone_multipliers(built):
    mk = lambda: S.make_product(2, 2, lib=S.emu_lib(), horizon=20, settings_override=CONE)[0]
    gm = mk()
    rb = O.Robot("go2_like")
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array(WALK))
    X = S.random_states(rb, 2)
    for _ in range(4):
        gm.iterate(X)
        X = gm.xs[:, 1, :].copy()
    blob = gm.save_state()
    gm.iterate(X)
    ref = gm.xs.copy()
    g2 = mk()
    g2.load_state(blob)
    g2.iterate(X)
    assert np.array_equal(g2.xs, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("iters", [1, 3])
def test_hip_kinodynamics_with_active_cones(built, iters):
    # (the scenario is hard on the solver -- every instance backtracks once rows are active -- and differences of 1e-7 grow tenfold per
    #  control step from the tenth step on, in the CPU build of the kernels as on the GPU: tools/cone_err.py; ten steps are compared)
    om, gm, rb = S.make_pair(3, max_iters=iters, settings_override=CONE, walk=WALK)
    assert _loop(om, gm, S.random_states(rb, 3), 10, 1e-5) >= 10
