"""Batched MPC over the centroidal OCP (the reference's examples/talos_centroidal.py pattern: CentroidalOCP + MPC +
interpolation of states / forces), on the go2_like table with 3-D contact forces.

    python examples/go2_centroidal_batched.py [batch] [steps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))
from simple_mpc import BatchedMPC, CentroidalOCP, RobotModelHandler, load_robot  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
model_handler = RobotModelHandler(load_robot("go2_like"), "standing", "root_joint")
feet = ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]
for n in feet:
    model_handler.addPointFoot(n, "root_joint")
gravity = np.array([0, 0, -9.81])
problem_conf = dict(
    timestep=0.01, w_u=np.eye(12) * 1e-3, w_com=np.zeros((3, 3)), w_linear_mom=np.diag([0.01, 0.01, 100]),
    w_angular_mom=np.diag([0.1, 0.1, 1000]), w_linear_acc=0.01 * np.eye(3), w_angular_acc=0.01 * np.eye(3), gravity=gravity, mu=0.8,
    Lfoot=0.01, Wfoot=0.01, force_size=3,
)
T = 50
problem = CentroidalOCP(problem_conf, model_handler)
problem.createProblem(np.zeros(9), T, 3, gravity[2], False)
mpc_conf = dict(support_force=-model_handler.getMass() * gravity[2], TOL=1e-4, mu_init=1e-8, max_iters=1, num_threads=1, swing_apex=0.15,
                T_fly=30, T_contact=10, timestep=0.01)
mpc = BatchedMPC(mpc_conf, problem, B)
quadru = dict.fromkeys(feet, True)
mpc.generateCycleHorizon([quadru] * 10 + [dict(quadru, FL_foot=False, RR_foot=False)] * 30 + [quadru] * 10 + [dict(quadru, FR_foot=False, RL_foot=False)] * 30)
V = np.zeros((B, 6))
V[:, 0] = np.linspace(0.0, 0.4, B)
mpc.switchToWalk(V[0])
mpc.setVelocityBaseBatched(V)

X = np.tile(model_handler.getReferenceState(), (B, 1))  # measured multibody states: iterate() reduces them to [com; h; L]
t0 = time.time()
for step in range(steps):
    mpc.iterate(X)
dt = (time.time() - t0) / steps
print("%d robots, %d control steps: %.2f ms per batched step (%.0f control-steps/s incl. host copies)" % (B, steps, dt * 1e3, B / dt))
us = mpc.us.reshape(B, T, 4, 3)
print("vertical force per robot at t = 0: %.1f .. %.1f N (weight %.1f N)" % (us[:, 0, :, 2].sum(1).min(), us[:, 0, :, 2].sum(1).max(), model_handler.getMass() * 9.81))
x_i, xdot_i, f_i = mpc.interpolate(0.004)
u_fb = mpc.riccatiFeedback(0.004, X)
print("interpolated centroidal state", x_i.shape, "forces", f_i.shape, "Riccati-feedback forces", u_fb.shape)
