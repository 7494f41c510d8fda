"""Lane-per-problem stage evaluation (simple-mpc_amd/csrc/smpc_kino_lane.h, smpc_kino_deriv2.h): line search and derivative pass run on it
by default; SMPC_LANE_DERIV=0 keeps the derivative pass on the one-kernel path, SMPC_LANE_EVAL=0 everything.  All three must agree and
reproduce the oracle."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O


def _loop(env, monkeypatch, lib, steps=4, batch=3, max_iters=2, horizon=12):
    for k in ("SMPC_LANE_EVAL", "SMPC_LANE_DERIV", "SMPC_LANE_STREAM"):
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
    _check(S.xcheck_lib(), monkeypatch, 1e-8)  # (the one-kernel path of a cone-free problem lives in the cross-check build)


def _stream_pair(monkeypatch, lib, bitwise, **kw):
    a = _loop({"SMPC_LANE_STREAM": "0"}, monkeypatch, lib, **kw)
    b = _loop({}, monkeypatch, lib, **kw)
    for (xa, ua), (xb, ub) in zip(a, b):
        assert np.all(np.isfinite(xb))
        if bitwise:
            assert np.array_equal(xa, xb) and np.array_equal(ua, ub)
        else:
            assert S.rel_err(xa, xb) < 1e-12 and S.rel_err(ua, ub) < 1e-11


def test_stream_handover_is_the_tile_handover_emulated(monkeypatch):
    """The derivative pass takes the tree kernel's fields as one contiguous run per problem, in production order, through the order table the
    kernel records itself (Buffers::evd / ev_order; SMPC_LANE_STREAM=0: the strided tile): the same numbers by another route."""
    _stream_pair(monkeypatch, S.emu_lib(), True, steps=4, batch=3)


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [3, 70, 200])
def test_stream_handover_is_the_tile_handover_gpu(monkeypatch, batch):
    # (on the GPU the stream is flushed by a wave-wide transposition through LDS: partial wavefronts -- 3, 70 = 64 + 6 -- have idle lanes
    #  running along; the random states make some instances backtrack: list-mode launches with a few entries.  The two forms of the tree
    #  kernel are separate template instantiations: the compiler contracts a few multiply-adds differently in them, so single instances
    #  differ in the last bit after some steps -- 1e-16 .. 4e-14 observed, tools/stream_cmp.py; the CPU build of the kernels is bit-identical)
    _stream_pair(monkeypatch, S.xcheck_lib(), False, steps=6, batch=batch, max_iters=3, horizon=20)
