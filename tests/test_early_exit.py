"""SolverProxDDP's convergence test inside iterate (reference src/mpc.cpp:43 hands TOL to the solver, :212 runs it): optional
(smpc_set_early_exit_on_tol); oracle, emulated kernels and HIP library agree, and a converged instance keeps its iterate."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O


def _run(lib, tol_states):
    om, gm, rb = S.make_pair(3, max_iters=4, lib=lib, horizon=10, walk=(0, 0, 0, 0, 0, 0), mpc_override={"TOL": 1e-4})
    for m in (om, gm):
        m.switchToStand()
        m.setEarlyExitOnTol(True)
    X = np.stack([rb.x_ref] * 3)
    X[1] = S.random_states(rb, 1, seed=2, scale=0.5)[0]  # one instance away from the solution: it keeps iterating
    alphas = []
    for _ in range(3):
        om.iterate(X)
        gm.iterate(X)
        assert S.rel_err(om.xs, gm.xs) < tol_states and S.rel_err(om.us, gm.us) < 10 * tol_states
        alphas.append(gm.info[:, 2].copy())  # step length of the last iteration per instance
        X = om.xs[:, 1, :].copy()
    return np.array(alphas), om, gm


def test_early_exit_emulated():
    alphas, om, gm = _run(S.emu_lib(), 1e-9)
    # the standing instances are converged (no step: alpha 0) from some control step on, the perturbed one still steps
    assert np.any(alphas[:, 0] == 0.0) and np.any(alphas[:, 2] == 0.0)
    assert alphas[0, 1] > 0.0


def test_fixed_iterations_stay_the_default():
    """without the switch the same problem takes its max_iters steps (the metric of record is at fixed iterations)"""
    om, gm, rb = S.make_pair(2, max_iters=3, lib=S.emu_lib(), horizon=8, walk=(0, 0, 0, 0, 0, 0), mpc_override={"TOL": 1e-4})
    for m in (om, gm):
        m.switchToStand()
    X = np.stack([rb.x_ref] * 2)
    for _ in range(2):
        om.iterate(X)
        gm.iterate(X)
    assert np.all(gm.info[:, 2] > 0.0) and S.rel_err(om.xs, gm.xs) < 1e-9


@pytest.mark.gpu
def test_early_exit_gpu():
    alphas, om, gm = _run(None, 1e-8)
    assert np.any(alphas[:, 0] == 0.0)
