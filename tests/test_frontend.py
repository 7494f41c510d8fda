"""State feedback front-end: foot positions, centre of mass, centroidal momentum and centroidal state of
measured multibody states (reference src/robot-handler.cpp:106-149) -- kernel bodies on CPU and the HIP library against the
oracle's rigid-body routines (which test_oracle_model.py pins by momentum identities and finite differences)."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O


def _check(lib):
    B = 5
    gm, rb, _, _ = S.make_product(B, lib=lib)
    X = S.random_states(rb, B, seed=11, scale=2.0)
    out = gm.updateInternalData(X)
    for b in range(B):
        c = rb.centroidal(X[b])
        assert np.abs(out["feet"][b].ravel() - c["feet"].ravel()).max() < 1e-13
        assert np.abs(out["com"][b] - c["com"]).max() < 1e-13
        assert np.abs(out["hg"][b] - c["hg"]).max() < 1e-12
        assert np.abs(out["centroidal_state"][b] - np.concatenate([c["com"], c["hg"]])).max() < 1e-12
    # total momentum of a robot at rest is zero; its CoM height is that of the reference posture
    rest = np.tile(rb.x_ref, (B, 1))
    out = gm.updateInternalData(rest)
    assert np.abs(out["hg"]).max() < 1e-14
    with pytest.raises(RuntimeError):
        gm.updateInternalData(X[:2])


def _check_all_handles(lib):
    """The same front end behind every kind of handle and both robots ("lets callers pass raw simulator states for all three OCP types",
    SURVEY 8f f2): kinodynamics / centroidal / full dynamics of the quadruped, and the three OCPs of the biped with flat feet."""
    for mk, talos in ((S.make_cent_product, False), (S.make_full_product, False), (S.make_talos_kino_product, True), (S.make_talos_cent_product, True),
                      (S.make_talos_product, True)):
        gm, rb, _, _ = mk(3, lib=lib, horizon=10)
        X = (S.talos_random_states if talos else S.random_states)(rb, 3, seed=4, scale=1.0)
        out = gm.updateInternalData(X)
        for b in range(3):
            c = rb.centroidal(X[b])
            assert np.abs(out["feet"][b].ravel() - c["feet"].ravel()).max() < 1e-12, mk.__name__
            assert np.abs(out["com"][b] - c["com"]).max() < 1e-13 and np.abs(out["hg"][b] - c["hg"]).max() < 1e-11, mk.__name__
            assert np.abs(out["centroidal_state"][b] - np.concatenate([c["com"], c["hg"]])).max() < 1e-11


def test_kernel_body_on_cpu(built):
    _check(S.emu_lib())
    _check_all_handles(S.emu_lib())


@pytest.mark.gpu
def test_hip_library(built):
    _check(None)
    _check_all_handles(None)
