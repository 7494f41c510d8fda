"""Centroidal model: CentroidalFwdDynamics + IntegratorEuler with derivatives (reference
src/centroidal-dynamics.cpp:79-81).  The oracle is pinned by finite differences and by physics (free fall,
static equilibrium); the kernel body (CPU build) and the HIP library are compared with the oracle."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O
import simple_mpc

M, G, DT = 15.0, np.array([0.0, 0.0, -9.81]), 0.01


def _case(rng, nf=4):
    x = rng.uniform(-1, 1, 9)
    u = rng.uniform(-50, 150, 3 * nf)
    contact = rng.integers(0, 2, nf).astype(np.uint8)
    pos = rng.uniform(-0.4, 0.4, (nf, 3))
    return x, u, contact, pos


def test_oracle_jacobians_match_finite_differences():
    rng = np.random.default_rng(0)
    for _ in range(5):
        x, u, c, pos = _case(rng)
        xn, A, B = O.centroidal_dynamics(M, G, DT, x, u, c, pos)
        eps = 1e-6
        for j in range(9):
            d = np.zeros(9)
            d[j] = eps
            fd = (O.centroidal_dynamics(M, G, DT, x + d, u, c, pos)[0] - O.centroidal_dynamics(M, G, DT, x - d, u, c, pos)[0]) / (2 * eps)
            assert np.abs(fd - A[:, j]).max() < 1e-7
        for j in range(12):
            d = np.zeros(12)
            d[j] = eps
            fd = (O.centroidal_dynamics(M, G, DT, x, u + d, c, pos)[0] - O.centroidal_dynamics(M, G, DT, x, u - d, c, pos)[0]) / (2 * eps)
            assert np.abs(fd - B[:, j]).max() < 1e-7
        for f in range(4):
            if not c[f]:
                assert np.all(B[:, 3 * f : 3 * f + 3] == 0.0)  # feet in the air: zero columns


def test_oracle_physics():
    # free fall: momentum rate = m g, no torque, CoM moves with h / m
    x = np.array([0.1, 0.2, 0.5, 1.5, 0.0, -3.0, 0.3, 0.2, 0.1])
    xn, _, _ = O.centroidal_dynamics(M, G, DT, x, np.ones(12) * 100, np.zeros(4, np.uint8), np.zeros((4, 3)))
    assert np.allclose(xn[:3], x[:3] + DT * x[3:6] / M) and np.allclose(xn[3:6], x[3:6] + DT * M * G) and np.allclose(xn[6:], x[6:])
    # static equilibrium: four symmetric feet carrying the weight -> no momentum change
    pos = np.array([[0.2, 0.1, 0], [0.2, -0.1, 0], [-0.2, 0.1, 0], [-0.2, -0.1, 0]], float)
    x = np.concatenate([[0, 0, 0.3], np.zeros(6)])
    u = np.tile([0, 0, M * 9.81 / 4], 4)
    xn, _, _ = O.centroidal_dynamics(M, G, DT, x, u, np.ones(4, np.uint8), pos)
    assert np.abs(xn - x).max() < 1e-12


def _check(lib):
    rng = np.random.default_rng(1)
    Bn = 130  # more than two waves
    cases = [_case(rng) for _ in range(Bn)]
    X, U = np.array([c[0] for c in cases]), np.array([c[1] for c in cases])
    cs, pos = np.array([c[2] for c in cases]), np.array([c[3] for c in cases])
    Xn, A, Bm = simple_mpc.centroidal_dynamics(M, G, DT, X, U, cs, pos, lib=lib)
    for b in range(Bn):
        xo, Ao, Bo = O.centroidal_dynamics(M, G, DT, *cases[b])
        assert np.abs(Xn[b] - xo).max() < 1e-13 and np.abs(A[b] - Ao).max() < 1e-14 and np.abs(Bm[b] - Bo).max() < 1e-14
    with pytest.raises(RuntimeError):
        simple_mpc.centroidal_dynamics(M, G, DT, X[:, :8], U, cs, pos, lib=lib)


def test_kernel_body_on_cpu(built):
    _check(S.emu_lib())


@pytest.mark.gpu
def test_hip_library(built):
    _check(None)
