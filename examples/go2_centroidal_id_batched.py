"""The control stack of the reference's examples/talos_centroidal.py (centroidal MPC at 100 Hz :200-216, interpolated CoM / force targets
and foot references + CentroidalID at 1 kHz :218-246) for a batch of point-foot quadrupeds, with the simulator replaced by the constrained
forward dynamics kernel: every robot is integrated under the torques its own inverse-dynamics QP returned.

    python examples/go2_centroidal_id_batched.py [batch] [mpc_steps]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-mpc_amd", "python"))
from simple_mpc import BatchedMPC, CentroidalID, CentroidalOCP, KinodynamicsOCP, RobotModelHandler, load_robot, presets  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100

mh = RobotModelHandler(load_robot("go2_like"), "standing", "root_joint")
for n in presets.GO2_FEET:
    mh.addPointFoot(n, "root_joint")
nq, nv, mass = mh.nq, mh.nv, mh.getMass()
mpc_conf = {k: v for k, v in presets.go2_mpc_settings(mh, max_iters=1).items() if k in presets.MPC_KEYS}

ocp = CentroidalOCP(presets.go2_centroidal_settings(mh), mh)
ocp.createProblem(np.zeros(9), 50, 3, -9.81, False)
mpc = BatchedMPC(mpc_conf, ocp, B)
mpc.generateCycleHorizon(presets.trot_cycle())
V = np.zeros((B, 6))
V[:, 0] = np.linspace(0.0, 0.2, B)
mpc.switchToWalk(V[0])
mpc.setVelocityBaseBatched(V)

# "device.execute(tau)": a kinodynamics handle lends its constrained forward dynamics kernel as the simulator
kocp = KinodynamicsOCP(presets.go2_kino_settings(mh), mh)
kocp.createProblem(mh.getReferenceState(), 50, 3, -9.81, False)
sim = BatchedMPC(mpc_conf, kocp, B)

dt_mpc, N_simu = 0.01, 10
dt_simu = dt_mpc / N_simu
id_settings = dict(kp_base=7.0, kp_com=7.0, kp_posture=10.0, kp_contact=10.0, kp_feet_tracking=2000.0, w_base=50.0, w_com=100.0, w_posture=1.0,
                   w_contact_force=1e-6, w_contact_motion=1e-3, w_feet_tracking=100.0)  # :127-136 (+ tracking of the feet in the air)
effort, vmax = np.array([23.7, 23.7, 45.43] * 4), np.array([30.1, 30.1, 15.7] * 4)
centroidal_ID = CentroidalID(mh, dt_simu, id_settings, effort, vmax, batch=B)

X = np.tile(mh.getReferenceState(), (B, 1))
t_mpc = t_id = 0.0
for step in range(steps):
    t0 = time.time()
    mpc.iterate(X)
    t_mpc += time.time() - t0
    contact = mpc.ocp_handler.getContactState(0)
    mask = np.full(B, sum(1 << i for i, c in enumerate(contact) if c), np.uint32)
    refs = mpc.getReferencePoses()  # [B][H][nf][3]: the foot references of the horizon
    for sub in range(N_simu):
        d = sub / float(N_simu)
        x_i, _, f_i = mpc.interpolate(d * dt_mpc)  # [com; linear momentum; angular momentum], forces
        t0 = time.time()
        centroidal_ID.setTargets(x_i[:, :3], x_i[:, 3:6] / mass, (1 - d) * refs[:, 0] + d * refs[:, 1], (refs[:, 1] - refs[:, 0]) / dt_mpc, contact, f_i)
        tau = centroidal_ID.solve(step * dt_mpc + sub * dt_simu, X[:, :nq], X[:, nq:])
        t_id += time.time() - t0
        a = sim.constraintDynamics(X, tau, mask, Kp=[0.0, 0.0, 0.0], Kd=[50.0, 50.0, 50.0])["a"]
        vn = X[:, nq:] + a * dt_simu
        X = np.stack([presets.integrate(np.r_[X[b, :nq], vn[b]], np.r_[vn[b] * dt_simu, np.zeros(nv)], nq) for b in range(B)])
print("%d robots, %d MPC steps x %d controller ticks: MPC %.2f ms / step, inverse dynamics %.2f ms / tick (host copies included)" % (
    B, steps, N_simu, 1e3 * t_mpc / steps, 1e3 * t_id / (steps * N_simu)))
print("base x after %.2f s: %.3f m (0 m/s command) ... %.3f m (0.2 m/s command); base height %.3f .. %.3f m; max |tau| %.1f N m; QP residual %.1e" % (
    steps * dt_mpc, X[0, 0], X[-1, 0], X[:, 2].min(), X[:, 2].max(), np.abs(tau).max(), centroidal_ID.resid.max()))
assert np.all(np.isfinite(X)) and np.all(np.abs(tau) <= effort + 1e-6)
