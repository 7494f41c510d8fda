"""Batched counterpart of the reference's examples/go2_kinodynamics.py (settings :42-106, gait :111-139, MPC loop :216-300)
without the simulator: B robots, each with its own velocity command, closed on the MPC's own prediction (x_meas = xs[1]).

    python examples/go2_kinodynamics_batched.py [batch] [steps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))
from simple_mpc import BatchedMPC, KinodynamicsOCP, RobotModelHandler, load_robot  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100

model_handler = RobotModelHandler(load_robot("go2_like"), "standing", "root_joint")
feet = ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]
for n in feet:
    model_handler.addPointFoot(n, "root_joint")
nv = model_handler.nv
nu = nv - 6 + 3 * len(feet)
gravity = np.array([0, 0, -9.81])

w_basepos, w_legpos = [0, 0, 100, 10, 10, 0], [1, 1, 1]
w_basevel, w_legvel = [10, 10, 10, 10, 10, 10], [0.1, 0.1, 0.1]
problem_conf = dict(
    timestep=0.01,
    w_x=np.diag(np.array(w_basepos + w_legpos * 4 + w_basevel + w_legvel * 4, float)),
    w_u=np.diag(np.concatenate((np.ones(12) * 0.01, np.ones(nv - 6) * 1e-5))),
    w_cent=np.diag([0, 0, 1, 0.1, 0.1, 10.0]),
    w_centder=np.diag([0, 0, 0, 0.1, 0.1, 0.1]),
    gravity=gravity,
    force_size=3,
    w_frame=np.eye(3) * 2000,
    qmin=model_handler.lowerPositionLimit[7:],
    qmax=model_handler.upperPositionLimit[7:],
    mu=0.8,
    Lfoot=0.01,
    Wfoot=0.01,
    kinematics_limits=True,
    force_cone=False,
    land_cstr=False,
)
T = 50
dynproblem = KinodynamicsOCP(problem_conf, model_handler)
dynproblem.createProblem(model_handler.getReferenceState(), T, 3, gravity[2], False)

T_ds, T_ss = 10, 30
mpc_conf = dict(
    support_force=-model_handler.getMass() * gravity[2], TOL=1e-4, mu_init=1e-8, max_iters=1, num_threads=1, swing_apex=0.15,
    T_fly=T_ss, T_contact=T_ds, timestep=0.01,
)
mpc = BatchedMPC(mpc_conf, dynproblem, B)

quadru = dict.fromkeys(feet, True)
lift_fl_rr = dict(quadru, FL_foot=False, RR_foot=False)
lift_fr_rl = dict(quadru, FR_foot=False, RL_foot=False)
mpc.generateCycleHorizon([quadru] * T_ds + [lift_fl_rr] * T_ss + [quadru] * T_ds + [lift_fr_rl] * T_ss)

# every robot gets its own command: forward speed 0 .. 0.4 m/s, yaw rate -0.3 .. 0.3 rad/s
V = np.zeros((B, 6))
V[:, 0] = np.linspace(0.0, 0.4, B)
V[:, 5] = np.linspace(-0.3, 0.3, B)
mpc.switchToWalk(V[0])
mpc.setVelocityBaseBatched(V)

X = np.tile(model_handler.getReferenceState(), (B, 1))
t0 = time.time()
for step in range(steps):
    mpc.iterate(X)
    X = mpc.xs[:, 1, :]  # closed on the plan; a simulator would integrate us[:, 0] and the interpolated targets here
dt = (time.time() - t0) / steps
xs = mpc.xs
print("%d robots, %d control steps: %.2f ms per batched step (%.0f control-steps/s incl. host copies)" % (B, steps, dt * 1e3, B / dt))
print("base x after %.1f s: slowest %.3f m, fastest %.3f m ; yaw-rate command spread gives base y in [%.3f, %.3f] m" % (
    steps * 0.01, xs[0, 0, 0], xs[-1, 0, 0], xs[:, 0, 1].min(), xs[:, 0, 1].max()))
x_i, acc_i, f_i = mpc.interpolate(0.004)  # targets for a 1 kHz whole-body controller, 4 ms after the solve
print("interpolated targets: x", x_i.shape, "acc", acc_i.shape, "forces", f_i.shape)
