"""Interpolator: the oracle restatement of reference src/interpolator.cpp:5-78 against the properties
the reference's own test checks (tests/interpolator.cpp:37-182: knot hit, beyond the end -> last knot, half-way point =
integrate(q0, 0.5 difference(q0, q1)), equal knots), then the kernel bodies (sequential-lane build on CPU, HIP library on
the GPU) against the oracle, both for explicit knot lists and for the batched targets of an MPC solution."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O
import simple_mpc

DT = 0.01


def _random_configs(rb, n, rng):
    qs = []
    for _ in range(n):
        dx = np.concatenate([rng.uniform(-1, 1, rb.nv), np.zeros(rb.nv)])
        qs.append(rb.integrate(rb.x_ref, dx)[: rb.nq])
    return qs


def _random_states(rb, n, rng):
    return [np.concatenate([q, rng.uniform(-1, 1, rb.nv)]) for q in _random_configs(rb, n, rng)]


def test_oracle_matches_the_reference_test_properties():
    rb = O.Robot("go2_like")
    rng = np.random.default_rng(0)
    qs = _random_configs(rb, 4, rng)
    # reference tests/interpolator.cpp:37-66
    assert np.allclose(O.interpolate(1, rb.nv, 0.02, DT, qs), qs[2])
    assert np.allclose(O.interpolate(1, rb.nv, 0.5, DT, qs), qs[-1])
    x0 = np.concatenate([qs[0], np.zeros(rb.nv)])
    x1 = np.concatenate([qs[1], np.zeros(rb.nv)])
    half = rb.integrate(x0, 0.5 * rb.difference(x0, x1))[: rb.nq]
    assert np.allclose(O.interpolate(1, rb.nv, 0.005, DT, qs), half)
    assert np.allclose(O.interpolate(1, rb.nv, 0.005, DT, [qs[0], qs[0]]), qs[0])
    # states: reference tests/interpolator.cpp:69-110
    xs = _random_states(rb, 4, rng)
    assert np.allclose(O.interpolate(0, rb.nv, 0.02, DT, xs), xs[2])
    assert np.allclose(O.interpolate(0, rb.nv, 0.5, DT, xs), xs[-1])
    xi = O.interpolate(0, rb.nv, 0.005, DT, xs)
    d = rb.difference(xs[0], xs[1])
    d[rb.nv:] = 0
    assert np.allclose(xi[: rb.nq], rb.integrate(xs[0], 0.5 * d)[: rb.nq])
    assert np.allclose(xi[rb.nq:], 0.5 * (xs[0][rb.nq:] + xs[1][rb.nq:]))
    # linear: reference tests/interpolator.cpp:113-150
    vs = [rng.uniform(-1, 1, 7) for _ in range(4)]
    assert np.allclose(O.interpolate(2, rb.nv, 0.02, DT, vs), vs[2])
    assert np.allclose(O.interpolate(2, rb.nv, 0.5, DT, vs), vs[-1])
    assert np.allclose(O.interpolate(2, rb.nv, 0.005, DT, vs), 0.5 * (vs[0] + vs[1]))


def _check_knot_lists(lib):
    rb = O.Robot("go2_like")
    rng = np.random.default_rng(1)
    ip = simple_mpc.Interpolator(None, lib=lib)
    qs, xs = _random_configs(rb, 4, rng), _random_states(rb, 4, rng)
    vs = [rng.uniform(-1, 1, 9) for _ in range(4)]
    for delay in (0.0, 0.0031, 0.005, 0.0199, 0.02, 0.0273, 0.5):
        assert np.abs(ip.interpolateConfiguration(delay, DT, qs) - O.interpolate(1, rb.nv, delay, DT, qs)).max() < 1e-13
        assert np.abs(ip.interpolateState(delay, DT, xs) - O.interpolate(0, rb.nv, delay, DT, xs)).max() < 1e-13
        assert np.abs(ip.interpolateLinear(delay, DT, vs) - O.interpolate(2, rb.nv, delay, DT, vs)).max() < 1e-15
    cs = [[True, False], [False, True], [True, True]]
    assert ip.interpolateContacts(0.0, DT, cs) == cs[0] and ip.interpolateContacts(0.015, DT, cs) == cs[1]
    assert ip.interpolateContacts(1.0, DT, cs) == cs[2]  # reference tests/interpolator.cpp:153-182
    with pytest.raises(RuntimeError, match="State is not of the right size"):
        ip.interpolateState(0.0, DT, qs)
    # the biped (nq 29 / nv 28: the robot the reference's own interpolator test runs on, tests/interpolator.cpp:13-35)
    rt = O.Robot("talos_like")
    it = simple_mpc.Interpolator(simple_mpc.load_robot("talos_like", lib), lib=lib)
    X = S.talos_random_states(rt, 4, seed=2, scale=0.8)
    for delay in (0.0, 0.0031, 0.0199, 0.0273, 0.5):
        step = min(int(delay / DT), 3)
        sfrac = (delay - step * DT) / DT
        e = X[3] if step >= 3 else rt.integrate(X[step], sfrac * rt.difference(X[step], X[step + 1]))
        assert np.abs(it.interpolateState(delay, DT, list(X)) - e).max() < 1e-13
        assert np.abs(it.interpolateConfiguration(delay, DT, [x[: rt.nq] for x in X]) - e[: rt.nq]).max() < 1e-13


def _check_batched_targets(lib):
    B = 3
    gm, rb, _, _ = S.make_product(B, max_iters=2, lib=lib)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    gm.iterate(S.random_states(rb, B))
    xs, us = gm.xs, gm.us
    xd = np.stack([gm.getStateDerivative(0), gm.getStateDerivative(1)], 1)
    nf3 = 3 * gm.nf
    for delay in (0.0, 0.004, 0.0099, 0.01, 0.013):
        x, a, f = gm.interpolate(delay)
        for b in range(B):
            acc = [np.concatenate([xd[b, t, rb.nv : rb.nv + 6], us[b, t, nf3:]]) for t in (0, 1)]  # examples/go2_kinodynamics.py:256-260
            assert np.abs(x[b] - O.interpolate(0, rb.nv, delay, DT, [xs[b, 0], xs[b, 1]])).max() < 1e-12
            assert np.abs(a[b] - O.interpolate(2, rb.nv, delay, DT, acc)).max() < 1e-12
            assert np.abs(f[b].ravel() - O.interpolate(2, rb.nv, delay, DT, [us[b, 0, :nf3], us[b, 1, :nf3]])).max() < 1e-12
    # Riccati feedback between knots (reference examples/go2_fulldynamics.py:271-285)
    Xm = xs[:, 0, :] + 0.0
    Xm[:, 7:] += np.random.default_rng(3).normal(0, 1e-2, Xm[:, 7:].shape)
    K0 = gm.K0
    for delay in (0.0, 0.006):
        ufb = gm.riccatiFeedback(delay, Xm)
        for b in range(B):
            xi = O.interpolate(0, rb.nv, delay, DT, [xs[b, 0], xs[b, 1]])
            ui = O.interpolate(2, rb.nv, delay, DT, [us[b, 0], us[b, 1]])
            ref = ui - K0[b] @ rb.difference(Xm[b], xi)
            assert np.abs(ufb[b] - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    # more knots: anywhere along the horizon
    x, _, _ = gm.interpolate(0.237, knots=gm.H + 1)
    assert np.abs(x[1] - O.interpolate(0, rb.nv, 0.237, DT, list(xs[1]))).max() < 1e-12
    with pytest.raises(RuntimeError):
        gm.interpolate(0.0, knots=1)


def test_kernel_bodies_on_cpu(built):
    lib = S.emu_lib()
    _check_knot_lists(lib)
    _check_batched_targets(lib)


@pytest.mark.gpu
def test_hip_library(built):
    _check_knot_lists(None)
    _check_batched_targets(None)
