"""Golden vectors of the centroidal path (tests/golden/go2_cent_golden.npz, produced by the oracle with
make_golden_centroidal.py): regression pin of the oracle, CPU-tier check of the emulated kernel body, and the check of
the HIP path on the GPU box (-m gpu), which cannot run the container's oracle build of record."""
import os

import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "go2_cent_golden.npz"))
TOL = 1e-4
HARD = dict(settings_override=dict(mu=0.1), walk=(0.8, 0.5, 0, 0, 0, 0.5))
CASES = [("easy", {}, 1), ("easy", {}, 3), ("hard", HARD, 1), ("hard", HARD, 3)]


def test_oracle_reproduces_stage_vectors():
    rb = O.Robot("go2_like")
    Cn = O.Cent(rb, O.go2_centroidal_settings(rb))
    for i, m in enumerate(G["stage_mask"]):
        a = (int(m), G["stage_u_ref"], G["stage_x_tgt"], G["stage_pos"], G["stage_x"][i], G["stage_u"][i])
        e, d = Cn.eval(*a), Cn.deriv(*a)
        assert S.rel_err(G["stage%d_xnext" % i], e["xnext"]) < 1e-13
        assert abs(G["stage%d_cost" % i] - e["cost"]) < 1e-12 * abs(e["cost"])
        assert S.rel_err(G["stage%d_c" % i], e["c"]) < 1e-13
        for k in ("A", "B", "lx", "lu", "Lxx", "Lxu", "Luu", "Cu"):
            assert S.rel_err(G["stage%d_%s" % (i, k)], d[k]) < 1e-12, (i, k)


def _drive(m, rb, tag):
    X = G[tag + "_X0"].copy()
    for _ in range(6):
        m.iterate(X)
        X = np.stack([rb.integrate(X[b], np.r_[np.zeros(18), 0.02, 0.01, np.zeros(16)]) for b in range(4)])


@pytest.mark.parametrize("name,kw,k", CASES)
def test_oracle_reproduces_closed_loop(name, kw, k):
    om, rb, _ = S.make_cent_oracle(4, k, settings_override=kw.get("settings_override"))
    om.generateCycleHorizon(O.trot_cycle())
    om.switchToWalk(np.array(kw.get("walk", (0.2, 0, 0, 0, 0, 0)), float))
    tag = "%s%d" % (name, k)
    _drive(om, rb, tag)
    assert S.rel_err(G[tag + "_xs"], om.xs) < 1e-9 and S.rel_err(G[tag + "_K0"], om.K0) < 1e-7


def _product(name, kw, k, lib):
    gm, rb, _, _ = S.make_cent_product(4, k, lib=lib, settings_override=kw.get("settings_override"))
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array(kw.get("walk", (0.2, 0, 0, 0, 0, 0)), float))
    tag = "%s%d" % (name, k)
    assert S.rel_err(G[tag + "_cold_xs"], gm.xs[0]) < TOL
    _drive(gm, rb, tag)
    assert S.rel_err(G[tag + "_xs"], gm.xs) < TOL
    assert S.rel_err(G[tag + "_us"], gm.us) < 10 * TOL
    assert S.rel_err(G[tag + "_vs"], gm.vs) < 10 * TOL
    assert S.rel_err(G[tag + "_K0"], gm.K0) < 10 * TOL
    assert np.array_equal(G[tag + "_alpha"], gm.info[:, 2])


@pytest.mark.parametrize("name,kw,k", CASES)
def test_emulated_kernel_reproduces_closed_loop(built, name, kw, k):
    _product(name, kw, k, S.emu_lib())


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,k", CASES)
def test_hip_reproduces_closed_loop(built, name, kw, k):
    _product(name, kw, k, None)
