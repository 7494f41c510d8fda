"""The whole control stack of the reference's examples/go2_kinodynamics.py -- MPC at 100 Hz (:216-262), interpolation + KinodynamicsID at
1 kHz (:264-300), the simulated robot in between -- for a batch of robots with nothing crossing the host inside the loop: the MPC's
interpolation kernel writes the controller's targets in place, states and torques stay in HBM, the "simulator" is the constrained
forward dynamics kernel + a semi-implicit Euler step.

    python examples/go2_stack_resident.py [batch] [mpc_steps]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))
from simple_mpc import BatchedMPC, KinodynamicsID, KinodynamicsOCP, RobotModelHandler, load_robot, presets  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100

mh = RobotModelHandler(load_robot("go2_like"), "standing", "root_joint")
for n in presets.GO2_FEET:
    mh.addPointFoot(n, "root_joint")
ocp = KinodynamicsOCP(presets.go2_kino_settings(mh), mh)
ocp.createProblem(mh.getReferenceState(), 50, 3, -9.81, False)
mpc = BatchedMPC({k: v for k, v in presets.go2_mpc_settings(mh, max_iters=1).items() if k in presets.MPC_KEYS}, ocp, B)
mpc.generateCycleHorizon(presets.trot_cycle())
V = np.zeros((B, 6))
V[:, 0] = np.linspace(0.0, 0.3, B)
mpc.switchToWalk(V[0])
mpc.setVelocityBaseBatched(V)

dt_mpc, N_simu = 0.01, 10
dt_simu = dt_mpc / N_simu
id_settings = dict(kp_base=7.0, kp_posture=10.0, kp_contact=10.0, w_base=100.0, w_posture=1.0, w_contact_force=1.0, w_contact_motion=1.0)  # :147-154
effort, vmax = np.array([23.7, 23.7, 45.43] * 4), np.array([30.1, 30.1, 15.7] * 4)
kino_ID = KinodynamicsID(mh, dt_simu, id_settings, effort, vmax, batch=B)
kino_ID.shareStream(mpc)  # MPC step, targets, QP solves and simulator steps in one in-order queue

X = torch.from_numpy(np.tile(mh.getReferenceState(), (B, 1))).cuda()
torch.cuda.synchronize()
t0 = time.time()
for step in range(steps):
    mpc.iterate_device(X.data_ptr())
    mpc.wait()
    contact = mpc.ocp_handler.getContactState(0)
    for sub in range(N_simu):
        kino_ID.setTargetsFromMPC(mpc, sub / float(N_simu) * dt_mpc)
        kino_ID.solve_device(X.data_ptr())
        mpc.simStepDevice(X.data_ptr(), kino_ID.tau_device_ptr(), contact, dt_simu, Kp=[0.0, 0.0, 0.0], Kd=[50.0, 50.0, 50.0])
mpc.wait()
wall = time.time() - t0
kino_ID.shareStream(None)
Xh = X.cpu().numpy()
sim_time = steps * dt_mpc
print("%d robots, %.2f s of simulated time (%d MPC steps x %d controller ticks) in %.2f s: %.1f ms per MPC period, %.2fx real time for the batch, "
      "%.0f robot-seconds per second" % (B, sim_time, steps, N_simu, wall, 1e3 * wall / steps, sim_time / wall, B * sim_time / wall))
print("base x: %.3f m (0 m/s command) ... %.3f m (0.3 m/s command); base height %.3f .. %.3f m" % (Xh[0, 0], Xh[-1, 0], Xh[:, 2].min(), Xh[:, 2].max()))
assert np.all(np.isfinite(Xh)) and np.all(np.abs(Xh[:, 2] - mh.getReferenceState()[2]) < 0.05)
