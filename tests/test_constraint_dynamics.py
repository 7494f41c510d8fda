"""Full-dynamics model, first device block: the constrained forward dynamics kernel (full_fd_body, behind
smpc_full_forward_dynamics) -- kernel body on the CPU and the HIP library -- against the oracle restatement of
pinocchio::constraintDynamics (oracle/orc_full.hpp, pinned in test_oracle_fulldynamics.py by momentum balance, inverse
dynamics, power balance and finite differences).  Tolerance: 1e-9 relative on accelerations and contact forces (the two
sides factor the same matrices in a different order)."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

MASKS = [0b1111, 0b0110, 0b1001, 0b0001, 0b0000, 0b1110]


def _check(lib, n=24, seed=5, Kp=None, Kd=None):
    gm, rb, _, _ = S.make_product(2, lib=lib)
    rng = np.random.default_rng(seed)
    X = S.random_states(rb, n, seed=seed, scale=2.0)
    tau = rng.normal(size=(n, rb.nv - 6)) * 5
    masks = np.array([MASKS[i % len(MASKS)] for i in range(n)], np.uint32)
    out = gm.constraintDynamics(X, tau, masks, Kp=Kp, Kd=Kd)
    for i in range(n):
        r = rb.full_forward_dynamics(X[i], tau[i], int(masks[i]), Kp if Kp is not None else (0, 0, 0), Kd if Kd is not None else (0, 0, 0))
        nc = r["lam"].size
        sa, sl = max(1.0, np.abs(r["a"]).max()), max(1.0, np.abs(r["lam"]).max() if nc else 1.0)
        assert np.abs(out["a"][i] - r["a"]).max() < 1e-9 * sa, (i, masks[i])
        if nc:
            assert np.abs(out["lam"][i, :nc] - r["lam"]).max() < 1e-9 * sl, (i, masks[i])
        assert np.all(out["lam"][i, nc:] == 0.0)
        assert out["iters"][i] == r["prox_iters"]
    return gm, rb, X, tau, masks, out


def _properties(gm, rb, X, tau, masks, out):
    """size-independent checks: Newton-Euler on the whole robot with the returned accelerations and forces"""
    G = np.array([0.0, 0.0, -9.81])
    for i in range(0, len(X), 7):
        r = rb.full_forward_dynamics(X[i], tau[i], int(masks[i]))
        nc = r["lam"].size
        c = rb.centroidal(X[i])
        # generalized forces on the free-flyer rows: M a + nle = J^T lam (no actuation on the base)
        res = r["M"] @ out["a"][i] + r["nle"] - np.concatenate([np.zeros(6), tau[i]]) - r["J"].T @ out["lam"][i, :nc]
        assert np.abs(res).max() < 1e-7
        if nc:
            assert np.abs(r["J"] @ out["a"][i] + r["gamma"]).max() < 1e-6
        else:
            assert np.abs(c["Ag"] @ out["a"][i] + c["dAgv"] - np.concatenate([rb.mass * G, np.zeros(3)])).max() < 1e-8


def _check_full_handles(lib):
    """The same entry point on full-dynamics handles (fdyn_fd_body: the stage kernel's own dynamics phases): the quadruped with 3-D LOCAL
    contacts and the biped with 6-D LOCAL_WORLD_ALIGNED contacts, with the Baumgarte gains of the reference's Talos example."""
    for talos in (False, True):
        gm, rb, _, _ = (S.make_talos_product if talos else S.make_full_product)(2, lib=lib, horizon=10)
        fs, masks_all = (6, [0b11, 0b01, 0b10, 0b00]) if talos else (3, MASKS)
        n = 10
        rng = np.random.default_rng(3)
        X = (S.talos_random_states(rb, n, seed=3, scale=0.7) if talos else S.random_states(rb, n, seed=3, scale=2.0))
        tau = rng.normal(size=(n, rb.nv - 6)) * 5
        masks = np.array([masks_all[i % len(masks_all)] for i in range(n)], np.uint32)
        Kp, Kd = ((0, 0, 50.0, 0, 0, 0), (100.0,) * 6) if talos else ((0, 0, 50.0), (100.0,) * 3)
        out = gm.constraintDynamics(X, tau, masks, Kp=Kp, Kd=Kd)
        assert out["lam"].shape == (n, fs * rb.nf)
        for i in range(n):
            r = rb.full_forward_dynamics(X[i], tau[i], int(masks[i]), Kp, Kd, fs=fs)
            nc = r["lam"].size
            sa, sl = max(1.0, np.abs(r["a"]).max()), max(1.0, np.abs(r["lam"]).max() if nc else 1.0)
            assert np.abs(out["a"][i] - r["a"]).max() < 1e-9 * sa, (talos, i, masks[i])
            if nc:
                assert np.abs(out["lam"][i, :nc] - r["lam"]).max() < 1e-9 * sl, (talos, i, masks[i])
            assert np.all(out["lam"][i, nc:] == 0.0) and out["iters"][i] == r["prox_iters"]
            # Newton-Euler on the whole robot with the returned accelerations and forces
            res = r["M"] @ out["a"][i] + r["nle"] - np.concatenate([np.zeros(6), tau[i]]) - r["J"].T @ out["lam"][i, :nc]
            assert np.abs(res).max() < 1e-6 * max(1.0, np.abs(r["nle"]).max())


def test_kernel_body_on_cpu(built):
    _properties(*_check(S.emu_lib()))
    _check(S.emu_lib(), n=6, seed=9, Kp=(0, 0, 50.0), Kd=(100.0, 100.0, 100.0))
    _check_full_handles(S.emu_lib())


def test_argument_checks(built):
    gm, rb, _, _ = S.make_product(2, lib=S.emu_lib())
    with pytest.raises(RuntimeError):
        gm.constraintDynamics(np.zeros((3, 5)), np.zeros((3, 12)), np.zeros(3))
    cm = S.make_cent_product(2, lib=S.emu_lib())[0]
    with pytest.raises(RuntimeError):
        cm.constraintDynamics(np.tile(rb.x_ref, (2, 1)), np.zeros((2, 12)), np.zeros(2))


@pytest.mark.gpu
def test_hip_library(built):
    _properties(*_check(None))
    _check(None, n=6, seed=9, Kp=(0, 0, 50.0), Kd=(100.0, 100.0, 100.0))
    _check_full_handles(None)


@pytest.mark.gpu
def test_hip_library_large_batch(built):
    """a batch far beyond the handle's own: every wave independent, replicated states bit-identical"""
    gm, rb, _, _ = S.make_product(2)
    n = 20000
    X = S.random_states(rb, 50, seed=3, scale=1.0)
    Xr = np.tile(X, (n // 50, 1))
    rng = np.random.default_rng(1)
    tau = np.tile(rng.normal(size=(50, 12)) * 5, (n // 50, 1))
    masks = np.tile(np.array([MASKS[i % len(MASKS)] for i in range(50)], np.uint32), n // 50)
    out = gm.constraintDynamics(Xr, tau, masks)
    assert np.all(np.isfinite(out["a"])) and np.all(np.isfinite(out["lam"]))
    assert np.array_equal(out["a"][:50], out["a"][-50:]) and np.array_equal(out["lam"][:50], out["lam"][-50:])
    for i in range(0, 50, 9):
        r = rb.full_forward_dynamics(Xr[i], tau[i], int(masks[i]))
        assert np.abs(out["a"][i] - r["a"]).max() < 1e-9 * max(1.0, np.abs(r["a"]).max())
