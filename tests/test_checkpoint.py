"""smpc_save_state / smpc_load_state (SURVEY 8b minimum C ABI): a handle restored from a checkpoint continues
bit-identically -- into the same handle (roll back) and into a fresh one (migration) -- for both OCP kinds."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O


def _case(make, lib):
    B = 3
    gm, rb, _, _ = make(B, 2, lib=lib)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.setVelocityBaseBatched(np.array([[0.3, 0, 0, 0, 0, 0], [0, 0.2, 0, 0, 0, 0.3], [-0.1, 0, 0, 0, 0, 0.0]]))
    Xs = [S.random_states(rb, B, seed=s) for s in range(8)]
    for k in range(3):
        gm.iterate(Xs[k])
    blob = gm.save_state()
    ref = []
    for k in range(3, 8):
        gm.iterate(Xs[k])
        ref.append((gm.xs.copy(), gm.us.copy(), gm.getReferencePoses().copy(), gm.foot_land_times))
    # roll back the same handle
    gm.load_state(blob)
    for k in range(3, 8):
        gm.iterate(Xs[k])
        assert np.array_equal(gm.xs, ref[k - 3][0]) and np.array_equal(gm.us, ref[k - 3][1])
    # a fresh handle (no gait generated, default commands) resumes from the checkpoint
    g2, _, _, _ = make(B, 2, lib=lib)
    g2.load_state(blob)
    for k in range(3, 8):
        g2.iterate(Xs[k])
        assert np.array_equal(g2.xs, ref[k - 3][0]) and np.array_equal(g2.us, ref[k - 3][1])
        assert np.array_equal(g2.getReferencePoses(), ref[k - 3][2]) and g2.foot_land_times == ref[k - 3][3]
    # shape mismatches and truncated buffers are errors, not silent corruption
    g3, _, _, _ = make(B + 1, 2, lib=lib)
    with pytest.raises(RuntimeError, match="does not match"):
        g3.load_state(blob)
    with pytest.raises(RuntimeError):
        g2.load_state(blob[: len(blob) // 2])
    return blob


def test_checkpoint_kinodynamics_emu(built):
    _case(S.make_product, S.emu_lib())


def test_checkpoint_centroidal_emu(built):
    blob = _case(S.make_cent_product, S.emu_lib())
    gk, _, _, _ = S.make_product(3, 2, lib=S.emu_lib())
    with pytest.raises(RuntimeError, match="does not match"):
        gk.load_state(blob)  # a centroidal checkpoint into a kinodynamics handle


@pytest.mark.gpu
def test_checkpoint_kinodynamics_gpu(built):
    _case(S.make_product, None)


@pytest.mark.gpu
def test_checkpoint_centroidal_gpu(built):
    _case(S.make_cent_product, None)
