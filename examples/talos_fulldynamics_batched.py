"""Batched counterpart of the reference's examples/talos_fulldynamics.py (settings :47-99, MPC :101-137, loop :168-210) without the
simulator: B bipeds (the built-in talos_like table: Talos' joint tree, two flat feet), 6-D contacts with wrench cones, H = 100, each robot
with its own forward-speed command, closed on the MPC's own prediction (x_meas = xs[1]); the 1 kHz torque of the reference's inner loop,
u = us[0] - K_0 (x_meas (-) xs[0]), is produced for every robot by riccatiFeedback.

    python examples/talos_fulldynamics_batched.py [batch] [steps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))
from simple_mpc import BatchedMPC, FullDynamicsOCP, RobotModelHandler, load_robot  # noqa: E402
from simple_mpc import presets  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40

model_handler = RobotModelHandler(load_robot("talos_like"), "half_sitting", "root_joint")
for n in presets.TALOS_FEET:
    model_handler.addQuadFoot(n, "root_joint", presets.TALOS_QUAD)
gravity = np.array([0, 0, -9.81])

problem_conf = presets.talos_full_settings(model_handler)  # the dict of the reference script: w_x, w_u, w_cent, w_forces, w_frame,
T = 100                                                    # Kp / Kd_correction, limits, mu, Lfoot, Wfoot, force_cone = True
dynproblem = FullDynamicsOCP(problem_conf, model_handler)
dynproblem.createProblem(model_handler.getReferenceState(), T, 6, gravity[2], False)

T_ss, T_ds = 80, 20
mpc_conf = dict(
    support_force=-model_handler.getMass() * gravity[2], TOL=1e-4, mu_init=1e-8, max_iters=1, num_threads=8, swing_apex=0.15,
    T_fly=T_ss, T_contact=T_ds, timestep=problem_conf["timestep"],
)
mpc = BatchedMPC(mpc_conf, dynproblem, B)

double = {"left_sole_link": True, "right_sole_link": True}
left = dict(double, right_sole_link=False)
right = dict(double, left_sole_link=False)
mpc.generateCycleHorizon([double] * T_ds + [left] * T_ss + [double] * T_ds + [right] * T_ss)

V = np.zeros((B, 6))
V[:, 0] = np.linspace(0.05, 0.15, B)  # the reference walks at 0.1 m/s
mpc.switchToWalk(V[0])
mpc.setVelocityBaseBatched(V)

X = np.tile(model_handler.getReferenceState(), (B, 1))
t0 = time.time()
for step in range(steps):
    if step == steps - 10:
        mpc.switchToStand()  # (t == 600 in the reference)
    mpc.iterate(X)
    X = mpc.xs[:, 1, :]
    tau = mpc.riccatiFeedback(0.0, X)  # us[0] - K_0 (x_meas (-) xs[0]) for every robot
dt = (time.time() - t0) / steps
forces = mpc.getContactForces(0)  # [B][nfeet][6]
print("%d bipeds, %d control steps: %.1f ms per batched step (%.0f control-steps/s incl. host copies)" % (B, steps, dt * 1e3, B / dt))
print("takeoff / landing times of the left foot: %s / %s" % (mpc.getFootTakeoffCycle("left_sole_link"), mpc.getFootLandCycle("left_sole_link")))
print("vertical contact forces at stage 0 (left, right), robot 0: %.1f N, %.1f N of %.1f N weight" % (
    forces[0, 0, 2], forces[0, 1, 2], model_handler.getMass() * 9.81))
print("feedback torque range: [%.1f, %.1f] N m" % (tau.min(), tau.max()))
