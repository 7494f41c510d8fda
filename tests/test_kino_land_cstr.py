"""land_cstr of the kinodynamics OCP with 3-D feet: at the stage where a foot lands (in contact there, not in the stage before of the
cycle: reference src/mpc.cpp:167-185) its height is pinned to the contact pose the cycle was created with -- FrameTranslationResidual
sliced to z, EqualityConstraint (src/kinodynamics.cpp:134-146).  One dense state row per landing foot; the kernels eliminate it
(Q^ += c^T c / mu) like the contact-velocity rows.  Also: all optional constraint blocks switched on together."""
import os
import subprocess
import sys

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAND = {"land_cstr": True}
SHORT = {"T_fly": 8, "T_contact": 4}


def short_trot():
    cs = np.ones((24, 4), np.uint8)  # 4 all feet, 8 with FL + RR in the air, 4 all feet, 8 with FR + RL in the air
    cs[4:12, [0, 3]] = 0
    cs[16:24, [1, 2]] = 0
    return cs


def _pair(batch, iters, lib, horizon, so, mo=None, cycle=None, walk=(0.3, 0, 0, 0, 0, 0.1)):
    mo = dict(SHORT, **(mo or {}))
    om, rb, _ = S.make_oracle(batch, iters, horizon, settings_override=so, mpc_override=mo)
    gm, _, _, _ = S.make_product(batch, iters, lib, horizon, settings_override=so, mpc_override=mo)
    for m in (om, gm):
        m.generateCycleHorizon(short_trot() if cycle is None else cycle)
        m.switchToWalk(np.array(walk, float))
    return om, gm, rb


def test_oracle_land_row_and_jacobian():
    rb = O.Robot("go2_like")
    s = O.go2_kino_settings(rb)
    s.update(LAND)
    K = O.Kino(rb, s)
    assert K.nc == 24 + 4
    x = S.random_states(rb, 1)[0]
    u, foot = np.zeros(K.nu), np.zeros((4, 3))
    m = 0b0111 | (0b1101 << 8)  # feet 0, 2, 3 flagged as landing; foot 3 is not in contact: no row
    c = K.eval(m, u, rb.x_ref, foot, x, u)["c"][24:]
    assert c[0] != 0 and c[2] != 0 and c[1] == 0 and c[3] == 0
    d = K.deriv(m, u, rb.x_ref, foot, x, u)
    eps, J = 1e-6, np.zeros((4, K.ndx))
    for i in range(K.ndx):
        dx = np.zeros(K.ndx)
        dx[i] = eps
        J[:, i] = (K.eval(m, u, rb.x_ref, foot, rb.integrate(x, dx), u)["c"][24:] - K.eval(m, u, rb.x_ref, foot, rb.integrate(x, -dx), u)["c"][24:]) / (2 * eps)
    assert np.abs(J - d["Cx"][24:]).max() < 1e-8 and np.abs(d["Cu"][24:]).max() == 0.0
    # at the reference state the feet are at their contact poses
    assert np.abs(K.eval(0b1111 | (0b1111 << 8), u, rb.x_ref, foot, rb.x_ref, u)["c"][24:]).max() < 1e-12


def _loop(om, gm, X, n, tol, first_col):
    rows = 0
    for _ in range(n):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < tol and S.rel_err(om.us, gm.us) < 10 * tol
        assert S.alphas_agree(om, gm, rtol=1e-6)
        rows = max(rows, int((om.vs[:, :, first_col:] != 0).sum()))
        X = om.xs[:, 1, :].copy()
    return rows


def test_emulated_kernels_with_land_rows(built):
    om, gm, rb = _pair(2, 2, S.emu_lib(), 20, LAND)
    assert _loop(om, gm, S.random_states(rb, 2), 20, 1e-8, 24) >= 4  # the first landing stage enters the horizon at step 12


ALL = {"land_cstr": True, "force_cone": True, "mu": 0.1}
WALK = (0.6, 0.4, 0, 0, 0, 0.5)


def _loop_all(om, gm, X, n, tol):
    cone = land = 0
    for _ in range(n):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < tol and S.rel_err(om.us, gm.us) < 10 * tol
        assert S.alphas_agree(om, gm, rtol=1e-6)
        vs = om.vs
        cone, land = max(cone, int((vs[:, :, 24:32] != 0).sum())), max(land, int((vs[:, :, 32:] != 0).sum()))
        X = om.xs[:, 1, :].copy()
    return cone, land


def test_emulated_kernels_all_optional_constraints_together(built):
    """Friction cones + land rows + terminal constraint on the gait of record with a fast sideways, turning command: a strained
    problem (the terminal constraint cannot be met within the short horizon: merit ~ 1e9, line searches down to 1 / 512) in which
    the CPU build of the kernels still follows the oracle to 2e-5 over 46 control steps (the north-star bar is 1e-4).  The land rows
    enter the horizon at step 41 (cycle index 40)."""
    om, gm, rb = S.make_pair(2, max_iters=1, lib=S.emu_lib(), horizon=20, settings_override=ALL, mpc_override={"terminal_constraint": True}, walk=WALK)
    cone, land = _loop_all(om, gm, S.random_states(rb, 2), 46, 1e-4)
    assert cone >= 30 and land >= 4


def test_emulated_kernels_are_lane_order_independent_with_land_rows(built, tmp_path):
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, mpc_setup as S\n"
        "import test_kino_land_cstr as T\n"
        "gm, rb, _, _ = S.make_product(2, 2, S.emu_lib(), 20, settings_override=dict(T.LAND, force_cone=True, mu=0.1), mpc_override=T.SHORT)\n"
        "gm.generateCycleHorizon(T.short_trot()); gm.switchToWalk(np.array([0.5, 0.3, 0, 0, 0, 0.4]))\n"
        "X = S.random_states(rb, 2)\n"
        "for _ in range(16):\n"
        "    gm.iterate(X); X = gm.xs[:, 1, :].copy()\n"
        "np.save(sys.argv[1], gm.xs)\n" % (ROOT, os.path.join(ROOT, "tests"))
    )
    outs = []
    for rev in ("0", "1"):
        out = str(tmp_path / ("xs%s.npy" % rev))
        subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, SMPC_EMU_REVERSE=rev), timeout=900)
        outs.append(np.load(out))
    assert np.array_equal(outs[0], outs[1])


def test_checkpoint_keeps_the_land_multipliers(built):
    def mk():
        g = S.make_product(2, 2, S.emu_lib(), 20, settings_override=LAND, mpc_override=SHORT)[0]
        return g

    gm = mk()
    rb = O.Robot("go2_like")
    gm.generateCycleHorizon(short_trot())
    gm.switchToWalk(np.array([0.3, 0, 0, 0, 0, 0.1]))
    X = S.random_states(rb, 2)
    for _ in range(14):
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
def test_hip_kinodynamics_with_land_rows(built):
    # the gait of record at H = 50: the first landing stage (cycle index 40) enters the horizon at control step 41
    om, gm, rb = S.make_pair(2, max_iters=1, settings_override=LAND, walk=(0.3, 0, 0, 0, 0, 0.1))
    assert _loop(om, gm, S.random_states(rb, 2), 48, 1e-7, 24) >= 4


@pytest.mark.gpu
def test_hip_all_optional_constraints_together(built):
    om, gm, rb = S.make_pair(2, max_iters=1, settings_override=ALL, mpc_override={"terminal_constraint": True}, walk=WALK)
    cone, land = _loop_all(om, gm, S.random_states(rb, 2), 44, 1e-4)
    assert cone >= 10 and land >= 4


@pytest.mark.gpu
def test_hip_full_size_with_all_optional_constraints(built):
    """B = 4096 (BASELINE's batch) with cones + land rows + terminal constraint: 16 distinct states replicated 256 times must give
    bit-identical replicas (the EXT instantiations at full occupancy), follow the oracle on the distinct ones, and stay finite."""
    B, nd = 4096, 16
    so, mo = ALL, {"terminal_constraint": True}
    gm, rb, _, _ = S.make_product(B, 1, None, 50, settings_override=so, mpc_override=mo)
    om, _, _ = S.make_oracle(nd, 1, 50, settings_override=so, mpc_override=mo)
    for m in (gm, om):
        m.generateCycleHorizon(O.trot_cycle())
        m.switchToWalk(np.array(WALK, float))
    Xo = S.random_states(rb, nd, seed=5)
    X = np.tile(Xo, (B // nd, 1))
    for _ in range(3):
        gm.iterate(X)
        om.iterate(Xo)
        xs = gm.xs
        X = xs[:, 1, :].copy()
        Xo = om.xs[:, 1, :].copy()
    xs = xs.reshape(B // nd, nd, *xs.shape[1:])
    assert np.abs(xs - xs[0:1]).max() == 0.0
    assert S.rel_err(om.xs, xs[0]) < 1e-6
    assert np.all(np.isfinite(gm.info))
