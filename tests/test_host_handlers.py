"""Host-side RobotModelHandler / RobotDataHandler of the Python mirror (reference src/robot-handler.cpp:76-149, bindings
expose-robot-handler.cpp:28-59): difference, frame placements, centroidal state -- NumPy code in the product package, checked here
against the oracle's rigid-body restatement and against the device front-end (CPU build of the kernels)."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O
import simple_mpc


@pytest.mark.parametrize("robot,feet,quad", [("go2_like", S.FEET, False), ("talos_like", S.TALOS_FEET, True)])
def test_data_handler_matches_oracle_and_device_frontend(built, robot, feet, quad):
    lib = S.emu_lib()
    rb = O.Robot(robot)
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot(robot, lib), "standing", "root_joint")
    for n in feet:
        mh.addQuadFoot(n, "root_joint", S.TALOS_QUAD) if quad else mh.addPointFoot(n, "root_joint")
    dh = simple_mpc.RobotDataHandler(mh)
    assert np.array_equal(dh.getState(), mh.getReferenceState())  # constructed at the reference state (src/robot-handler.cpp:97-103)
    X = (S.talos_random_states(rb, 3, scale=0.7) if quad else S.random_states(rb, 3))
    for x in X:
        dh.updateInternalData(x, False)
        c = rb.centroidal(x)
        assert np.allclose(dh.getCentroidalState(), np.r_[c["com"], c["hg"]], atol=1e-11)
        for i in range(len(feet)):
            assert np.allclose(dh.getFootPose(i).translation, c["feet"][i], atol=1e-12)
        bp = dh.getBaseFramePose()
        assert np.allclose(bp.translation, x[:3]) and np.allclose(bp.rotation @ bp.rotation.T, np.eye(3), atol=1e-12)
        assert np.allclose((bp.inverse() * bp).homogeneous, np.eye(4), atol=1e-12)
        assert np.allclose(dh.getFootRefPose(0).rotation, bp.rotation)
    if not quad:  # the device front-end of a kinodynamics handle (smpc_update_internal_data)
        gm, _, _, _ = S.make_product(3, lib=lib)
        fe = gm.updateInternalData(X)
        for b in range(3):
            dh.updateInternalData(X[b])
            assert np.allclose(dh.getCentroidalState(), fe["centroidal_state"][b], atol=1e-11)
            assert np.allclose([dh.getFootPose(i).translation for i in range(4)], fe["feet"][b], atol=1e-12)
        gm.generateCycleHorizon(O.trot_cycle())
        gm.iterate(X)
        assert np.allclose(gm.getDataHandler(2).getState(), X[2]) and gm.getDataHandler().getModelHandler() is gm.getModelHandler()


def test_model_handler_difference_and_frames(built):
    lib = S.emu_lib()
    rb = O.Robot("go2_like")
    mh = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in S.FEET:
        mh.addPointFoot(n, "root_joint")
    rng = np.random.default_rng(0)
    x1 = S.random_states(rb, 1)[0]
    for scale in (1e-3, 0.3, 2.5):  # small, ordinary and large rotations
        dx = rng.normal(size=rb.ndx) * scale
        dx[3:6] *= min(1.0, 3.0 / np.linalg.norm(dx[3:6]))  # keep the rotation below pi
        x2 = rb.integrate(x1, dx)
        assert np.allclose(mh.difference(x1, x2), dx, atol=1e-9)  # difference inverts integrate (pinocchio.difference semantics)
    assert np.allclose(mh.difference(x1, x1), 0.0)
    assert mh.getBaseFrameName() == "root_joint" and mh.getBaseFrameId() == 0
    ids = mh.getFeetFrameIds()
    assert len(set(ids + [mh.getFootRefFrameId(i) for i in range(4)] + [0])) == 9 and ids[2] == mh.getFootFrameId(2)
    # setFootReferencePlacement moves the foot's reference frame (private copy of the table: the built-in one is untouched)
    dh0 = simple_mpc.RobotDataHandler(mh).getFootRefPose(1).translation
    mh.setFootReferencePlacement(1, [0.3, -0.2, -0.31])
    dh1 = simple_mpc.RobotDataHandler(mh).getFootRefPose(1).translation
    assert not np.allclose(dh0, dh1) and np.allclose(dh1 - mh.getReferenceState()[:3], [0.3, -0.2, -0.31])
    mh2 = simple_mpc.RobotModelHandler(simple_mpc.load_robot("go2_like", lib), "standing", "root_joint")
    for n in S.FEET:
        mh2.addPointFoot(n, "root_joint")
    assert np.allclose(simple_mpc.RobotDataHandler(mh2).getFootRefPose(1).translation, dh0)
    # ... and the problem built from the modified handler uses it (the Raibert heuristic starts from the reference frame)
    ocp = simple_mpc.KinodynamicsOCP(O.go2_kino_settings(rb), mh)
    ocp.createProblem(mh.getReferenceState(), 20, 3, -9.81, False)
    ms = {k: v for k, v in O.go2_mpc_settings(rb, max_iters=1).items() if k in S.MPC_KEYS}
    gm = simple_mpc.BatchedMPC(ms, ocp, 1, lib=lib)
    gm.generateCycleHorizon(O.trot_cycle())
    gm.switchToWalk(np.array([0.2, 0, 0, 0, 0, 0.0]))
    for _ in range(45):
        gm.iterate(mh.getReferenceState()[None, :])
    refs = gm.getReferencePoses()[0]  # [H][nfeet][3]
    assert abs(refs[:, 1, 1].min() - (-0.2)) < 0.05  # foot 1 is sent towards y = -0.2 m
