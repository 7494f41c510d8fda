"""Lane-per-problem stage evaluation (simple-mpc_amd/csrc/smpc_kino_lane.h, smpc_kino_deriv2.h): line search and derivative pass run on it
by default; SMPC_LANE_DERIV=0 keeps the derivative pass on the one-kernel path, SMPC_LANE_EVAL=0 everything.  All three must agree and
reproduce the oracle."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O


def _loop(env, monkeypatch, lib, steps=4, batch=3, max_iters=2, horizon=12):
    for k in ("SMPC_LANE_EVAL", "SMPC_LANE_DERIV"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    gm, rb, _, _ = S.make_product(batch, max_iters=max_iters, lib=lib, horizon=horizon)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.3, 0.05, 0, 0, 0, 0.1]))
    X = S.random_states(rb, batch, seed=5)
    out = []
    for _ in range(steps):
        gm.iterate(X)
        X = gm.xs[:, 1, :].copy()
        out.append((gm.xs.copy(), gm.us.copy()))
    return out


def _check(lib, monkeypatch, tol):
    ref = _loop({"SMPC_LANE_EVAL": "0"}, monkeypatch, lib)
    for env in ({}, {"SMPC_LANE_DERIV": "0"}):
        got = _loop(env, monkeypatch, lib)
        for (xa, ua), (xb, ub) in zip(ref, got):
            assert S.rel_err(xa, xb) < tol, env
            assert S.rel_err(ua, ub) < tol, env


def test_lane_paths_match_the_one_kernel_path_emulated(monkeypatch):
    _check(S.emu_lib(), monkeypatch, 1e-9)


def test_lane_derivative_path_matches_oracle_emulated(monkeypatch):
    monkeypatch.delenv("SMPC_LANE_DERIV", raising=False)
    om, gm, rb = S.make_pair(2, max_iters=2, lib=S.emu_lib(), horizon=10)
    X = S.random_states(rb, 2, seed=11)
    for _ in range(5):
        om.iterate(X)
        gm.iterate(X)
        X = om.xs[:, 1, :].copy()
    assert S.rel_err(om.xs, gm.xs) < 1e-8
    assert S.rel_err(om.us, gm.us) < 1e-7


@pytest.mark.gpu
def test_lane_paths_match_the_one_kernel_path_gpu(monkeypatch):
    _check(None, monkeypatch, 1e-8)
