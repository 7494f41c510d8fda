"""OCPHandler::setReferencePose / getReferencePose with a full SE3 (reference src/kinodynamics.cpp:154-170): what was set is what is returned
(tests/problem.cpp:157-160 sets SE3::Random() and reads it back), and MPC::iterate rewrites every stage's pose with the identity rotation before
it solves (src/mpc.cpp:303-309) -- so the rotation lives until the next control step and no solve ever evaluates it."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O
import simple_mpc


class SE3:
    def __init__(self, R, p):
        self.rotation, self.translation = np.asarray(R, float), np.asarray(p, float)


def _random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _round_trip(gm, X, names):
    rng = np.random.default_rng(3)
    ocp = gm.ocp_handler
    M = SE3(_random_rotation(rng), rng.normal(size=3))
    ocp.setReferencePose(4, names[0], M)
    got = ocp.getReferencePose(4, names[0])
    assert got == M and np.array_equal(got.rotation, M.rotation) and np.array_equal(got.translation, M.translation)
    assert np.array_equal(np.asarray(got), M.translation) and got.shape == (3,)  # (still the 3-vector earlier callers read)
    assert np.array_equal(got.homogeneous[:3, :3], M.rotation) and np.array_equal(got.homogeneous[:3, 3], M.translation)
    # the other feet and stages are untouched: identity rotations
    assert np.array_equal(ocp.getReferencePose(4, names[1]).rotation, np.eye(3)) and np.array_equal(ocp.getReferencePose(3, names[0]).rotation, np.eye(3))
    # setReferencePoses with identity placements (the second half of the reference's test), a 4 x 4 matrix, a bare translation
    new = {n: SE3(np.eye(3), [(-1.0) ** i, 0.0, 2.0]) for i, n in enumerate(names)}
    ocp.setReferencePoses(3, new)
    for n in names:
        assert ocp.getReferencePose(3, n) == new[n]
    H4 = np.eye(4)
    H4[:3, :3], H4[:3, 3] = M.rotation, [0.1, 0.2, 0.3]
    gm.setReferencePose(5, names[1], H4)
    assert np.array_equal(gm.getReferencePose(5, names[1]).homogeneous, H4)
    ocp.setReferencePose(5, names[1], np.array([0.3, 0.2, 0.1]))
    assert np.array_equal(ocp.getReferencePose(5, names[1]).rotation, np.eye(3)) and np.array_equal(np.asarray(ocp.getReferencePose(5, names[1])), [0.3, 0.2, 0.1])
    # a control step rewrites every pose: identity rotations again, and the solve is the one of a handle that never saw the rotation
    ocp.setReferencePose(4, names[0], M)
    gm.iterate(X)
    for t in (0, 4, gm.H - 1):
        for n in names:
            assert np.array_equal(ocp.getReferencePose(t, n).rotation, np.eye(3))
    with pytest.raises(RuntimeError):
        ocp.setReferencePose(gm.H, names[0], M)
    with pytest.raises(RuntimeError):
        ocp.setReferencePose(0, "no_such_foot", M)


def _kino(lib):
    gm, rb, _, _ = S.make_product(2, max_iters=1, lib=lib, horizon=12)
    ref, _, _, _ = S.make_product(2, max_iters=1, lib=lib, horizon=12)
    for g in (gm, ref):
        g.generateCycleHorizon(O.trot_cycle())
        g.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    X = S.random_states(rb, 2, seed=2)
    _round_trip(gm, X, list(gm.ocp_handler.model_handler.getFeetFrameNames()))
    ref.iterate(X)
    assert np.array_equal(gm.xs, ref.xs) and np.array_equal(gm.us, ref.us)


def test_emulated_kinodynamics_handle(built):
    _kino(S.emu_lib())


def test_emulated_flat_feet_handles(built):
    for make in (S.make_talos_kino_product, S.make_talos_cent_product):
        gm, rb, _, _ = make(1, max_iters=1, lib=S.emu_lib(), horizon=12, mpc_override=dict(T_fly=8, T_contact=2))
        gm.generateCycleHorizon(O.walk_cycle(2, 8))
        gm.switchToWalk(np.array([0.1, 0, 0, 0, 0, 0.0]))
        _round_trip(gm, S.talos_random_states(rb, 1, seed=1, scale=0.3), list(gm.ocp_handler.model_handler.getFeetFrameNames()))


@pytest.mark.gpu
def test_hip_handle(built):
    _kino(None)
