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


def test_kernel_body_on_cpu(built):
    _check(S.emu_lib())


@pytest.mark.gpu
def test_hip_library(built):
    _check(None)
