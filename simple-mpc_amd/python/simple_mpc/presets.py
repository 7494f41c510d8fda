"""Settings of record and synthetic inputs of the benchmark configurations (BASELINE.json), owned by the product
package so that examples and bench.py need nothing from tests/.

Sources (reference): examples/go2_kinodynamics.py:30-139 (kinodynamics OCP, MPC settings, trot cycle),
examples/go2_fulldynamics.py:42-98 / benchmark/go2.cpp:56-116 (full dynamics), examples/talos_centroidal.py:50-77 (centroidal
weights), examples/talos_fulldynamics.py:47-115 (Talos full dynamics).  SURVEY.md 8(d) defines the synthetic initial states.
"""
import numpy as np

GO2_FEET = ["FL_foot", "FR_foot", "RL_foot", "RR_foot"]
TALOS_FEET = ["left_sole_link", "right_sole_link"]
MPC_KEYS = ["support_force", "TOL", "mu_init", "max_iters", "num_threads", "swing_apex", "T_fly", "T_contact", "timestep"]
# sigma of the synthetic initial-state perturbation: base position / orientation, joints, base linear / angular
# velocity, joint velocities
SIGMA_BASE = dict(pos=0.02, ori=0.05, joints=0.1, vlin=0.1, vang=0.2, vjoints=0.5)


def sigma(nv):
    s = SIGMA_BASE
    na = nv - 6
    return np.concatenate([np.ones(3) * s["pos"], np.ones(3) * s["ori"], np.ones(na) * s["joints"], np.ones(3) * s["vlin"],
                           np.ones(3) * s["vang"], np.ones(na) * s["vjoints"]])


def go2_kino_settings(model_handler):
    """KinodynamicsSettings of record: reference examples/go2_kinodynamics.py:42-85."""
    nv = model_handler.nv
    w_x = np.diag(np.array([0, 0, 100, 10, 10, 0] + [1, 1, 1] * 4 + [10] * 6 + [0.1, 0.1, 0.1] * 4, float))
    w_u = np.diag(np.concatenate([np.ones(12) * 0.01, np.ones(nv - 6) * 1e-5]))
    return dict(
        timestep=0.01, w_x=w_x, w_u=w_u, w_cent=np.diag([0.0, 0.0, 1.0, 0.1, 0.1, 10.0]), w_centder=np.diag([0.0, 0.0, 0.0, 0.1, 0.1, 0.1]),
        gravity=np.array([0.0, 0.0, -9.81]), force_size=3, w_frame=np.eye(3) * 2000.0, qmin=model_handler.lowerPositionLimit[7:].copy(),
        qmax=model_handler.upperPositionLimit[7:].copy(), mu=0.8, Lfoot=0.01, Wfoot=0.01, kinematics_limits=True, force_cone=False,
        land_cstr=False)


def go2_mpc_settings(model_handler, max_iters=1, num_threads=0):
    """MPC settings of record: reference examples/go2_kinodynamics.py:96-106."""
    return dict(support_force=model_handler.getMass() * 9.81, TOL=1e-4, mu_init=1e-8, max_iters=max_iters, num_threads=num_threads,
                swing_apex=0.15, T_fly=30, T_contact=10, timestep=0.01)


def go2_centroidal_settings(model_handler):
    """CentroidalSettings for the "Go2 centroidal, H=50" configuration: the reference ships no Go2 centroidal script, the weights
    are those of its centroidal example (examples/talos_centroidal.py:50-76) with 3-D contact forces."""
    nf = model_handler.getFeetNb()
    return dict(timestep=0.01, w_u=np.diag(np.ones(3 * nf) * 0.001), w_com=np.zeros((3, 3)), w_linear_mom=np.diag([0.01, 0.01, 100.0]),
                w_angular_mom=np.diag([0.1, 0.1, 1000.0]), w_linear_acc=0.01 * np.eye(3), w_angular_acc=0.01 * np.eye(3),
                gravity=np.array([0.0, 0.0, -9.81]), mu=0.8, Lfoot=0.01, Wfoot=0.01, force_size=3)


def go2_full_settings(model_handler):
    """FullDynamicsSettings of record: reference examples/go2_fulldynamics.py:42-77 (3-D feet).  The robot table holds no effort
    limits: Go2's actuator limits (hip / thigh 23.7 N m, calf 45.43 N m) are used."""
    nv = model_handler.nv
    w_x = np.diag(np.array([0] * 6 + [1, 1, 1] * 4 + [10] * 6 + [0.1, 0.1, 0.1] * 4, float))
    eff = np.array([23.7, 23.7, 45.43] * 4)
    return dict(timestep=0.01, w_x=w_x, w_u=np.eye(nv - 6) * 1e-4, w_cent=np.diag([0.04, 0.04, 0, 0, 0, 0.0]), w_forces=np.eye(3) * 1e-4,
                w_frame=np.eye(3) * 1000.0, gravity=np.array([0, 0, -9.81]), force_size=3, Kp_correction=np.zeros(3),
                Kd_correction=np.zeros(3), umin=-eff, umax=eff, qmin=model_handler.lowerPositionLimit[7:].copy(),
                qmax=model_handler.upperPositionLimit[7:].copy(), mu=0.8, Lfoot=0.01, Wfoot=0.01, torque_limits=True,
                kinematics_limits=True, force_cone=False, land_cstr=False)


def go2_full_mpc_settings(model_handler, max_iters=1, num_threads=0):
    """MPC settings of the Go2 full-dynamics example: reference examples/go2_fulldynamics.py:88-98 (the same as kinodynamics)."""
    return go2_mpc_settings(model_handler, max_iters, num_threads)


TALOS_EFFORT = np.array([100, 160, 160, 300, 160, 100] * 2 + [200, 200] + [44, 44, 22, 22] * 2, float)


def talos_full_settings(model_handler):
    """FullDynamicsSettings of the Talos example: reference examples/talos_fulldynamics.py:47-115 (6-D feet, wrench cones).  The
    robot table holds no effort limits: Talos-like actuator limits are used (TALOS_EFFORT)."""
    w_x = np.diag(np.array([0, 0, 0, 10, 10, 10] + [0.1] * 6 * 2 + [1, 100] + [1, 1, 10, 10] * 2 + [10] * 6 + [1] * 6 * 2 + [1, 100]
                           + [10] * 4 * 2, float))
    nu = model_handler.nv - 6
    return dict(timestep=0.01, w_x=w_x, w_u=np.eye(nu) * 1e-4, w_cent=np.diag([0.1, 0.1, 10, 0.1, 0.1, 10.0]), w_forces=np.eye(6) * 1e-3,
                w_frame=np.eye(6) * 2000.0, gravity=np.array([0, 0, -9.81]), force_size=6, Kp_correction=np.array([0, 0, 50, 0, 0, 0.0]),
                Kd_correction=np.ones(6) * 100.0, umin=-TALOS_EFFORT, umax=TALOS_EFFORT.copy(), qmin=model_handler.lowerPositionLimit[7:].copy(),
                qmax=model_handler.upperPositionLimit[7:].copy(), mu=0.8, Lfoot=0.1, Wfoot=0.075, torque_limits=True, kinematics_limits=True,
                force_cone=True, land_cstr=False)


def talos_kino_settings(model_handler, force_cone=True):
    """KinodynamicsSettings of the reference's Talos configuration with 6-D feet: weights of examples/talos_kinodynamics.py:50-106
    (force_cone as in tests/test_utils.cpp:147-197, which the reference's own problem / MPC tests use)."""
    nv = model_handler.nv
    w_x = 10.0 * np.diag(np.array([0, 0, 1000, 1000, 1000, 1000] + [0.1] * 6 * 2 + [1, 1000] + [1, 1, 10, 10] * 2 + [10] * 6 + [1] * 6 * 2
                                  + [0.1, 100] + [10] * 4 * 2, float))
    w_u = np.diag(np.concatenate([[0.001, 0.001, 0.01], np.ones(3) * 0.1] * 2 + [np.ones(nv - 6) * 1e-4]))
    return dict(timestep=0.01, w_x=w_x, w_u=w_u, w_cent=np.diag([0.0, 0.0, 1.0, 0.1, 0.1, 10.0]), w_centder=np.diag([0.0, 0.0, 0.0, 0.1, 0.1, 0.1]),
                gravity=np.array([0.0, 0.0, -9.81]), force_size=6, w_frame=np.eye(6) * 100000.0, qmin=model_handler.lowerPositionLimit[7:].copy(),
                qmax=model_handler.upperPositionLimit[7:].copy(), mu=0.8, Lfoot=0.1, Wfoot=0.075, kinematics_limits=True,
                force_cone=bool(force_cone), land_cstr=False)


def talos_centroidal_settings(model_handler):
    """CentroidalSettings of the reference's Talos configuration with 6-D feet: examples/talos_centroidal.py:50-76."""
    nf = model_handler.getFeetNb()
    return dict(timestep=0.01, w_u=np.diag(([0.001] * 3 + [0.1] * 3) * nf), w_com=np.zeros((3, 3)), w_linear_mom=np.diag([0.01, 0.01, 100.0]),
                w_angular_mom=np.diag([0.1, 0.1, 1000.0]), w_linear_acc=0.01 * np.eye(3), w_angular_acc=0.01 * np.eye(3),
                gravity=np.array([0.0, 0.0, -9.81]), mu=0.8, Lfoot=0.1, Wfoot=0.075, force_size=6)


def talos_mpc_settings(model_handler, max_iters=1, num_threads=0):
    """MPC settings of the Talos example: reference examples/talos_fulldynamics.py:101-113 (T = 100, T_fly 80, T_contact 20)."""
    return dict(support_force=model_handler.getMass() * 9.81, TOL=1e-4, mu_init=1e-8, max_iters=max_iters, num_threads=num_threads,
                swing_apex=0.15, T_fly=80, T_contact=20, timestep=0.01)


TALOS_VMAX = np.array([3.87, 5.86, 5.86, 7.0, 5.86, 4.8] * 2 + [5.4, 5.4] + [2.7, 3.66, 4.58, 4.58] * 2, float)  # Talos-like joint velocity limits
TALOS_QUAD = np.array([[0.1, 0.075, 0], [-0.1, 0.075, 0], [-0.1, -0.075, 0], [0.1, -0.075, 0]])  # examples/talos_fulldynamics.py:22-33


def trot_cycle(T_ds=10, T_ss=30):
    """Contact cycle of reference examples/go2_kinodynamics.py:111-139, foot order FL FR RL RR."""
    quad, lift_fl, lift_fr = [1, 1, 1, 1], [0, 1, 1, 0], [1, 0, 0, 1]
    return np.array([quad] * T_ds + [lift_fl] * T_ss + [quad] * T_ds + [lift_fr] * T_ss, np.uint8)


def walk_cycle(T_ds=20, T_ss=80):
    """Biped contact cycle of reference examples/talos_fulldynamics.py:117-136, foot order left, right."""
    both, left, right = [1, 1], [1, 0], [0, 1]
    return np.array([both] * T_ds + [left] * T_ss + [both] * T_ds + [right] * T_ss, np.uint8)


def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def _quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def integrate(x, dx, nq):
    """x (+) dx on SE(3) x R^(nq-7) x R^nv (Pinocchio's free-flyer convention: dx[0:6] is a body-frame twist)."""
    x, dx = np.asarray(x, float), np.asarray(dx, float)
    nv = nq - 1
    v, w = dx[0:3], dx[3:6]
    t = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if t < 1e-8:
        Bc, Cc = 0.5, 1.0 / 6.0
        qd = np.concatenate([0.5 * w, [1.0]])
    else:
        Bc, Cc = (1 - np.cos(t)) / t**2, (t - np.sin(t)) / t**3
        qd = np.concatenate([np.sin(t / 2) / t * w, [np.cos(t / 2)]])
    p = (np.eye(3) + Bc * W + Cc * W @ W) @ v
    out = x.copy()
    out[0:3] = x[0:3] + _quat_R(x[3:7]) @ p
    q = _quat_mul(x[3:7], qd)
    out[3:7] = q / np.linalg.norm(q)
    out[7:nq] = x[7:nq] + dx[6:nv]
    out[nq:] = x[nq:] + dx[nv:]
    return out


def random_states(model_handler, batch, seed=20240529, scale=1.0):  # scale: 1 for Go2; 0.7 keeps a biped near its balanced posture
    """Synthetic initial states of SURVEY 8(d): x_ref (+) N(0, diag(sigma^2)), numpy default_rng(seed)."""
    rng = np.random.default_rng(seed)
    x_ref = model_handler.getReferenceState()
    sg = sigma(model_handler.nv)
    return np.stack([integrate(x_ref, rng.normal(size=2 * model_handler.nv) * sg * scale, model_handler.nq) for _ in range(batch)])
