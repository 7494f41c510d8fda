"""ctypes binding of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORC_DIR, "liborc.so")

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_bp = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    srcs = [os.path.join(ORC_DIR, f) for f in os.listdir(ORC_DIR) if f.endswith((".hpp", ".cpp"))]
    srcs += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", ORC_DIR, "-s"])
    return LIB_PATH


class IdSettingsC(C.Structure):
    _fields_ = [
        ("friction_coefficient", C.c_double), ("contact_weight_ratio_max", C.c_double), ("contact_weight_ratio_min", C.c_double),
        ("kp_base", C.c_double), ("kp_posture", C.c_double), ("kp_contact", C.c_double),
        ("w_base", C.c_double), ("w_posture", C.c_double), ("w_contact_motion", C.c_double), ("w_contact_force", C.c_double),
        ("contact_motion_equality", C.c_int), ("control_dt", C.c_double),
        ("tau_max", C.c_void_p), ("v_max", C.c_void_p), ("q_min", C.c_void_p), ("q_max", C.c_void_p),
        ("admm_iters", C.c_int), ("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double), ("admm_tol", C.c_double),
        ("centroidal", C.c_int), ("kp_com", C.c_double), ("kp_feet_tracking", C.c_double), ("w_com", C.c_double), ("w_feet_tracking", C.c_double),
        ("base_reference_as_coded", C.c_int), ("tsid_joint_bounds", C.c_int),
        ("force_size", C.c_int), ("quad_points", C.c_void_p),
    ]


class MpcSettingsC(C.Structure):
    _fields_ = [
        ("swing_apex", C.c_double),
        ("support_force", C.c_double),
        ("TOL", C.c_double),
        ("mu_init", C.c_double),
        ("timestep", C.c_double),
        ("max_iters", C.c_int),
        ("num_threads", C.c_int),
        ("T_fly", C.c_int),
        ("T_contact", C.c_int),
        ("T", C.c_int),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.orc_builtin_robot.restype = vp
    L.orc_builtin_robot.argtypes = [C.c_char_p]
    L.orc_robot_dims.argtypes = [vp, _ip]
    L.orc_robot_info.argtypes = [vp, _dp, _dp, _dp, _dp]
    L.orc_x_integrate.argtypes = [vp, _dp, _dp, _dp]
    L.orc_x_difference.argtypes = [vp, _dp, _dp, _dp]
    L.orc_exp6.argtypes = [_dp, _dp, _dp]
    L.orc_log6.argtypes = [_dp, _dp, _dp]
    L.orc_Jexp6.argtypes = [_dp, _dp]
    L.orc_Jlog6_of_exp.argtypes = [_dp, _dp]
    L.orc_kino_create.restype = vp
    L.orc_kino_create.argtypes = [vp, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int]
    L.orc_kino_destroy.argtypes = [vp]
    L.orc_kino_dims.argtypes = [vp, _ip]
    L.orc_kino_eval.argtypes = [vp, C.c_uint, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.orc_kino_deriv.argtypes = [vp, C.c_uint] + [_dp] * 14
    L.orc_kino_set_force_cone.argtypes = [vp, C.c_int, C.c_double]
    L.orc_set_fold_u_rows.argtypes = [C.c_int]
    L.orc_kino_set_land_cstr.argtypes = [vp, C.c_int]
    L.orc_kino_set_force_size.argtypes = [vp, C.c_int, _dp, _dp, C.c_double, C.c_double, C.c_double]
    L.orc_full_set_land_cstr.argtypes = [vp, C.c_int]
    L.orc_kino_term.argtypes = [vp, _dp, _dp, _dp, _dp, _dp]
    L.orc_kino_term_cstr.argtypes = [vp, _dp, _dp, C.c_double, _dp, _dp]
    L.orc_full_term.argtypes = [vp, _dp, _dp]
    L.orc_full_term.restype = C.c_double
    L.orc_centroidal.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.orc_full_forward_dynamics.argtypes = [vp, _dp, _dp, C.c_uint, C.c_int, _dp, _dp] + [_dp] * 7
    L.orc_full_forward_dynamics.restype = C.c_int
    L.orc_full_dynamics_derivatives.argtypes = [vp, _dp, _dp, C.c_uint, C.c_int, _dp, _dp, C.c_double, C.c_int] + [_dp] * 10
    L.orc_full_dynamics_derivatives.restype = C.c_int
    L.orc_full_rnea.argtypes = [vp, _dp, _dp, _dp]
    L.orc_full_create.argtypes = [vp, C.c_double] + [_dp] * 12 + [C.c_int] * 4 + [C.c_double] * 3
    L.orc_full_create.restype = vp
    L.orc_full_destroy.argtypes = [vp]
    L.orc_full_dims.argtypes = [vp, _ip]
    L.orc_full_eval.argtypes = [vp, C.c_uint, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.orc_full_deriv.argtypes = [vp, C.c_uint] + [_dp] * 14
    L.orc_full_solve.argtypes = [vp, C.c_int, C.POINTER(C.c_uint), _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_double, _dp, _dp, _dp]
    L.orc_full_solve.restype = C.c_int
    L.orc_kino_solve.argtypes = [vp, C.c_int, C.POINTER(C.c_uint)] + [_dp] * 6 + [C.c_int, C.c_double, C.c_double] + [_dp] * 5
    L.orc_kino_solve.restype = C.c_int
    L.orc_cent_solve.argtypes = [vp, C.c_int, C.POINTER(C.c_uint)] + [_dp] * 5 + [C.c_int, C.c_double, C.c_double] + [_dp] * 5
    L.orc_cent_solve.restype = C.c_int
    for _f in (L.orc_kino_row_kinds, L.orc_cent_row_kinds, L.orc_full_row_kinds):
        _f.argtypes = [vp, C.c_uint, _ip, _dp, _dp]
    L.orc_riccati.argtypes = [C.c_int] * 4 + [C.c_double] + [_dp] * 18
    L.orc_timer_create.restype = vp
    L.orc_timer_create.argtypes = [_bp, C.c_int, C.c_int, C.c_int]
    L.orc_timer_destroy.argtypes = [vp]
    L.orc_timer_recede.argtypes = [vp]
    L.orc_timer_get.argtypes = [vp, C.c_int, C.c_int, _ip, C.c_int]
    L.orc_bezier8.argtypes = [_dp, _dp, C.c_double, C.c_float, _dp]
    L.orc_mpc_create.restype = vp
    L.orc_mpc_create.argtypes = [vp, C.POINTER(MpcSettingsC), C.c_int, C.c_double]
    L.orc_mpc_destroy.argtypes = [vp]
    L.orc_mpc_generate_cycle.argtypes = [vp, _bp, C.c_int]
    L.orc_mpc_switch_to_walk.argtypes = [vp, _dp]
    L.orc_mpc_switch_to_stand.argtypes = [vp]
    L.orc_mpc_set_velocity_batched.argtypes = [vp, _dp]
    L.orc_mpc_set_stage_reference.argtypes = [vp, C.c_int, C.c_int, _dp]
    L.orc_cmpc_set_stage_reference.argtypes = [vp, C.c_int, C.c_int, _dp]
    L.orc_cmpc_set_velocity_batched.argtypes = [vp, _dp]
    L.orc_mpc_set_x_reference.argtypes = [vp, _dp]
    L.orc_mpc_iterate.restype = C.c_double
    L.orc_mpc_iterate.argtypes = [vp, _dp]
    L.orc_mpc_get.argtypes = [vp, C.c_int, _dp]
    L.orc_mpc_cold_iters.argtypes = [vp]
    L.orc_mpc_cold_trace.argtypes = [vp, _dp]
    L.orc_mpc_timing.argtypes = [vp, C.c_int, C.c_int, _ip, C.c_int]
    L.orc_set_terminal_constraint.argtypes = [C.c_int]
    L.orc_id_create.restype = vp
    L.orc_id_create.argtypes = [vp, C.POINTER(IdSettingsC), C.c_int]
    L.orc_id_destroy.argtypes = [vp]
    L.orc_id_set_target.argtypes = [vp, C.c_int, _dp, _dp, _dp, C.c_uint, _dp]
    L.orc_id_solve.argtypes = [vp, _dp, _dp, _dp, _dp, _dp]
    L.orc_id_set_target_centroidal.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp, C.c_uint, _dp]
    L.orc_id_quantities.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.orc_id_quantities6.argtypes = [vp, _dp, _dp, _dp, _dp, _dp, _dp]
    L.orc_id_qp.restype = C.c_int
    L.orc_id_qp.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
    L.orc_fmpc_create.restype = vp
    L.orc_fmpc_create.argtypes = [vp, C.POINTER(MpcSettingsC), C.c_int, C.c_double]
    L.orc_fmpc_destroy.argtypes = [vp]
    L.orc_fmpc_generate_cycle.argtypes = [vp, _bp, C.c_int]
    L.orc_fmpc_switch_to_walk.argtypes = [vp, _dp]
    L.orc_fmpc_switch_to_stand.argtypes = [vp]
    L.orc_fmpc_set_velocity_batched.argtypes = [vp, _dp]
    L.orc_fmpc_set_x_reference.argtypes = [vp, _dp]
    L.orc_fmpc_iterate.restype = C.c_double
    L.orc_fmpc_iterate.argtypes = [vp, _dp]
    L.orc_fmpc_get.argtypes = [vp, C.c_int, _dp]
    L.orc_fmpc_cold_iters.argtypes = [vp]
    L.orc_fmpc_cold_trace.argtypes = [vp, _dp]
    L.orc_fmpc_timing.argtypes = [vp, C.c_int, C.c_int, _ip, C.c_int]
    L.orc_fmpc_keep_knots.argtypes = [vp, C.c_int]
    L.orc_fmpc_get_knot.argtypes = [vp, C.c_int, C.c_int, _dp]
    L.orc_num_threads.restype = C.c_int
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_cent_create.restype = vp
    L.orc_cent_create.argtypes = [vp, C.c_double] + [_dp] * 7 + [C.c_double]
    L.orc_cent_create6.restype = vp
    L.orc_cent_create6.argtypes = [vp, C.c_double] + [_dp] * 7 + [C.c_double] * 3
    L.orc_cent_destroy.argtypes = [vp]
    L.orc_cent_eval.argtypes = [vp, C.c_uint] + [_dp] * 9
    L.orc_cent_deriv.argtypes = [vp, C.c_uint] + [_dp] * 14
    L.orc_cent_term.argtypes = [vp, _dp, _dp, _dp, _dp]
    L.orc_cmpc_create.restype = vp
    L.orc_cmpc_create.argtypes = [vp, vp, C.POINTER(MpcSettingsC), C.c_int, C.c_double]
    L.orc_cmpc_destroy.argtypes = [vp]
    L.orc_cmpc_generate_cycle.argtypes = [vp, _bp, C.c_int]
    L.orc_cmpc_switch_to_walk.argtypes = [vp, _dp]
    L.orc_cmpc_switch_to_stand.argtypes = [vp]
    L.orc_cmpc_set_x_reference.argtypes = [vp, _dp]
    L.orc_cmpc_iterate.restype = C.c_double
    L.orc_cmpc_iterate.argtypes = [vp, _dp]
    L.orc_cmpc_get.argtypes = [vp, C.c_int, _dp]
    L.orc_cmpc_cold_iters.argtypes = [vp]
    L.orc_cmpc_cold_trace.argtypes = [vp, _dp]
    L.orc_cmpc_timing.argtypes = [vp, C.c_int, C.c_int, _ip, C.c_int]
    L.orc_cmpc_keep_knots.argtypes = [vp, C.c_int]
    L.orc_cmpc_get_knot.argtypes = [vp, C.c_int, C.c_int, _dp]
    L.orc_mpc_keep_knots.argtypes = [vp, C.c_int]
    L.orc_mpc_get_knot.argtypes = [vp, C.c_int, C.c_int, _dp]
    L.orc_centroidal_dynamics.argtypes = [C.c_double, _dp, C.c_double, C.c_int, _dp, _dp, _bp, _dp, _dp, _dp, _dp]
    L.orc_friction.argtypes = [_dp, _dp, C.c_int, _dp, _dp, C.c_int]
    L.orc_interpolate.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _dp, C.c_int, C.c_int, _dp]
    _lib = L
    return L


def effective_cpus():
    """CPUs this process may really use: hardware threads capped by the cgroup CPU quota (a container with a 16-CPU quota on a
    256-thread host is throttled, not sped up, by 128 OpenMP threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()
            if q != "max":
                n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, p = int(f.read()), int(g.read())
                if q > 0:
                    n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def use_effective_cpus():
    """Size the oracle's OpenMP team to the CPUs the process may use; returns the thread count."""
    n = effective_cpus()
    lib().orc_set_num_threads(n)
    return n


class Robot:
    def __init__(self, name="go2_like"):
        self.ptr = lib().orc_builtin_robot(name.encode())
        assert self.ptr, name
        self._init_from_ptr()

    def _init_from_ptr(self):
        L = lib()
        d = np.zeros(4, np.int32)
        L.orc_robot_dims(self.ptr, d)
        self.nq, self.nv, self.nf, self.nj = (int(v) for v in d)
        self.nx = self.nq + self.nv
        self.ndx = 2 * self.nv
        self.q_ref = np.zeros(self.nq)
        self.q_lo = np.zeros(self.nv - 6)
        self.q_hi = np.zeros(self.nv - 6)
        m = np.zeros(1)
        L.orc_robot_info(self.ptr, self.q_ref, self.q_lo, self.q_hi, m)
        self.mass = float(m[0])
        self.x_ref = np.concatenate([self.q_ref, np.zeros(self.nv)])

    def integrate(self, x, dx):
        out = np.zeros(self.nx)
        lib().orc_x_integrate(self.ptr, np.ascontiguousarray(x, float), np.ascontiguousarray(dx, float), out)
        return out

    def difference(self, x0, x1):
        out = np.zeros(self.ndx)
        lib().orc_x_difference(self.ptr, np.ascontiguousarray(x0, float), np.ascontiguousarray(x1, float), out)
        return out

    def centroidal(self, x):
        hg, Ag, dAgv, com, feet = np.zeros(6), np.zeros((6, self.nv)), np.zeros(6), np.zeros(3), np.zeros((self.nf, 3))
        lib().orc_centroidal(self.ptr, np.ascontiguousarray(x, float), hg, Ag, dAgv, com, feet)
        return dict(hg=hg, Ag=Ag, dAgv=dAgv, com=com, feet=feet)

    def full_forward_dynamics(self, x, tau, mask, Kp=None, Kd=None, fs=3):
        """Constrained forward dynamics of the full-dynamics model (oracle/orc_full.hpp); fs = contact size (3: LOCAL point
        contacts, 6: LOCAL_WORLD_ALIGNED 6-D contacts)."""
        Kp = np.zeros(fs) if Kp is None else Kp
        Kd = np.zeros(fs) if Kd is None else Kd
        nv, nc = self.nv, fs * bin(int(mask)).count("1")
        a, lam, M, nle = np.zeros(nv), np.zeros(max(nc, 1)), np.zeros((nv, nv)), np.zeros(nv)
        J, gamma, tau_rnea = np.zeros((max(nc, 1), nv)), np.zeros(max(nc, 1)), np.zeros(nv)
        it = lib().orc_full_forward_dynamics(self.ptr, np.ascontiguousarray(x, float), np.ascontiguousarray(tau, float), int(mask), int(fs),
                                             np.ascontiguousarray(Kp, float), np.ascontiguousarray(Kd, float), a, lam, M, nle,
                                             J, gamma, tau_rnea)
        return dict(a=a, lam=lam[:nc], M=M, nle=nle, J=J[:nc], gamma=gamma[:nc], tau_rnea=tau_rnea, prox_iters=it)

    def full_dynamics_derivatives(self, x, tau, mask, Kp=None, Kd=None, prox_accuracy=0.0, prox_max_iter=0, fs=3):
        Kp = np.zeros(fs) if Kp is None else Kp
        Kd = np.zeros(fs) if Kd is None else Kd
        nv, nu, nc = self.nv, self.nv - 6, fs * bin(int(mask)).count("1")
        n1 = max(nc, 1)
        o = dict(a=np.zeros(nv), lam=np.zeros(n1), da_dq=np.zeros((nv, nv)), da_dv=np.zeros((nv, nv)), da_dtau=np.zeros((nv, nu)),
                 dlam_dq=np.zeros((n1, nv)), dlam_dv=np.zeros((n1, nv)), dlam_dtau=np.zeros((n1, nu)),
                 dtau_dq=np.zeros((nv, nv)), dtau_dv=np.zeros((nv, nv)))
        it = lib().orc_full_dynamics_derivatives(self.ptr, np.ascontiguousarray(x, float), np.ascontiguousarray(tau, float), int(mask), int(fs),
                                                 np.ascontiguousarray(Kp, float), np.ascontiguousarray(Kd, float),
                                                 float(prox_accuracy), int(prox_max_iter), *o.values())
        for k in ("lam", "dlam_dq", "dlam_dv", "dlam_dtau"):
            o[k] = o[k][:nc]
        o["prox_iters"] = it
        return o

    def full_rnea(self, x, a):
        tau = np.zeros(self.nv)
        lib().orc_full_rnea(self.ptr, np.ascontiguousarray(x, float), np.ascontiguousarray(a, float), tau)
        return tau


def go2_kino_settings(robot):
    """KinodynamicsSettings of record: reference examples/go2_kinodynamics.py:42-85."""
    nv = robot.nv
    w_basepos = [0, 0, 100, 10, 10, 0]
    w_legpos = [1, 1, 1]
    w_basevel = [10, 10, 10, 10, 10, 10]
    w_legvel = [0.1, 0.1, 0.1]
    w_x = np.diag(np.array(w_basepos + w_legpos * 4 + w_basevel + w_legvel * 4, float))
    w_u = np.diag(np.concatenate([np.ones(12) * 0.01, np.ones(nv - 6) * 1e-5]))
    w_cent = np.diag([0.0, 0.0, 1.0, 0.1, 0.1, 10.0])
    w_centder = np.diag([0.0, 0.0, 0.0, 0.1, 0.1, 0.1])
    return dict(
        timestep=0.01,
        w_x=w_x,
        w_u=w_u,
        w_cent=w_cent,
        w_centder=w_centder,
        gravity=np.array([0.0, 0.0, -9.81]),
        force_size=3,
        w_frame=np.eye(3) * 2000.0,
        qmin=robot.q_lo.copy(),
        qmax=robot.q_hi.copy(),
        mu=0.8,
        Lfoot=0.01,
        Wfoot=0.01,
        kinematics_limits=True,
        force_cone=False,
        land_cstr=False,
    )


def go2_mpc_settings(robot, max_iters=1, num_threads=0):
    """MPC settings of record: reference examples/go2_kinodynamics.py:96-106."""
    return dict(
        support_force=robot.mass * 9.81,
        TOL=1e-4,
        mu_init=1e-8,
        max_iters=max_iters,
        num_threads=num_threads,
        swing_apex=0.15,
        T_fly=30,
        T_contact=10,
        timestep=0.01,
        T=50,
    )


def trot_cycle(T_ds=10, T_ss=30):
    """Contact cycle of examples/go2_kinodynamics.py:111-139, order FL FR RL RR."""
    quad = [1, 1, 1, 1]
    lift_fl = [0, 1, 1, 0]
    lift_fr = [1, 0, 0, 1]
    cs = [quad] * T_ds + [lift_fl] * T_ss + [quad] * T_ds + [lift_fr] * T_ss
    return np.array(cs, np.uint8)


class Kino:
    def __init__(self, robot, s):
        L = lib()
        self.robot = robot
        self.s = s
        c = lambda a: np.ascontiguousarray(a, float)
        self.h = L.orc_kino_create(
            robot.ptr, s["timestep"], c(s["w_x"]), c(s["w_u"]), c(s["w_frame"]), c(s["w_cent"]), c(s["w_centder"]),
            c(s["qmin"]), c(s["qmax"]), c(s["gravity"]), int(s["kinematics_limits"]),
        )
        if int(s.get("force_size", 3)) == 6:  # (orc_kino_create read the 3-D prefixes of w_u / w_frame; this call takes the whole matrices)
            L.orc_kino_set_force_size(self.h, 6, c(s["w_u"]), c(s["w_frame"]), float(s.get("mu", 0.8)), float(s.get("Lfoot", 0.1)),
                                      float(s.get("Wfoot", 0.075)))
        if s.get("force_cone", False):
            L.orc_kino_set_force_cone(self.h, 1, float(s.get("mu", 0.8)))
        if s.get("land_cstr", False):
            L.orc_kino_set_land_cstr(self.h, 1)
        d = np.zeros(5, np.int32)
        L.orc_kino_dims(self.h, d)
        self.nx, self.ndx, self.nu, self.nc, self.nf = (int(v) for v in d)
        self.nv = robot.nv

    def eval(self, mask, u_ref, x_tgt, foot_ref, x, u):
        c = lambda a: np.ascontiguousarray(a, float)
        xnext, xdot, cost, cc = np.zeros(self.nx), np.zeros(2 * self.nv), np.zeros(1), np.zeros(self.nc)
        lib().orc_kino_eval(self.h, mask, c(u_ref), c(x_tgt), c(foot_ref), c(x), c(u), xnext, xdot, cost, cc)
        return dict(xnext=xnext, xdot=xdot, cost=float(cost[0]), c=cc)

    def deriv(self, mask, u_ref, x_tgt, foot_ref, x, u):
        c = lambda a: np.ascontiguousarray(a, float)
        n, m, k = self.ndx, self.nu, self.nc
        o = dict(
            A=np.zeros((n, n)), B=np.zeros((n, m)), lx=np.zeros(n), lu=np.zeros(m), Lxx=np.zeros((n, n)),
            Lxu=np.zeros((n, m)), Luu=np.zeros((m, m)), Cx=np.zeros((k, n)), Cu=np.zeros((k, m)),
        )
        lib().orc_kino_deriv(
            self.h, mask, c(u_ref), c(x_tgt), c(foot_ref), c(x), c(u), o["A"], o["B"], o["lx"], o["lu"], o["Lxx"],
            o["Lxu"], o["Luu"], o["Cx"], o["Cu"],
        )
        return o

    def term_cstr(self, x, ref, tau):
        """DCMPositionResidual at the terminal node: value (3) and Jacobian (3 x ndx)."""
        c, Cm = np.zeros(3), np.zeros((3, self.ndx))
        lib().orc_kino_term_cstr(self.h, np.ascontiguousarray(x, float), np.ascontiguousarray(ref, float), float(tau), c, Cm)
        return c, Cm

    def term(self, x_tgt, x):
        c = lambda a: np.ascontiguousarray(a, float)
        cost, lx, Lxx = np.zeros(1), np.zeros(self.ndx), np.zeros((self.ndx, self.ndx))
        lib().orc_kino_term(self.h, c(x_tgt), c(x), cost, lx, Lxx)
        return float(cost[0]), lx, Lxx

    def row_kinds(self, mask):
        """Per constraint row of a stage with this contact mask: kind (0 absent, 1 equality, 2 box [lo, hi], 3 <= 0), lo, hi."""
        k, lo, hi = np.zeros(self.nc, np.int32), np.zeros(self.nc), np.zeros(self.nc)
        lib().orc_kino_row_kinds(self.h, int(mask), k, lo, hi)
        return k, lo, hi

    def solve(self, masks, u_ref, x_tgt, foot_ref, x_tgt_term, x0, u0, max_iter=200, tol=1e-9, mu=1e-8):
        """The oracle's ProxDDP run to convergence on an H-stage problem with per-stage references (u_ref [H][nu], x_tgt [H][nx],
        foot_ref [H][nf*3]) from the constant guess (x0, u0)."""
        c = lambda a: np.ascontiguousarray(a, float)
        H = len(masks)
        mk = (C.c_uint * H)(*[int(m) for m in masks])
        trace, xs, us = np.zeros((max_iter, 6)), np.zeros((H + 1, self.nx)), np.zeros((H, self.nu))
        vs, lams = np.zeros((H, self.nc)), np.zeros((H + 1, self.ndx))
        it = lib().orc_kino_solve(self.h, H, mk, c(u_ref), c(x_tgt), c(foot_ref), c(x_tgt_term), c(x0), c(u0), max_iter, tol, mu, trace, xs, us, vs, lams)
        return dict(iters=it, trace=trace[:it], xs=xs, us=us, vs=vs, lams=lams)


def go2_full_settings(robot):
    """FullDynamicsSettings of record: reference examples/go2_fulldynamics.py:42-77 (3-D feet).  The robot table holds no
    effort limits: Go2's actuator limits (hip / thigh 23.7 N m, calf 45.43 N m) are used."""
    nv = robot.nv
    w_x = np.diag(np.array([0] * 6 + [1, 1, 1] * 4 + [10] * 6 + [0.1, 0.1, 0.1] * 4, float))
    eff = np.array([23.7, 23.7, 45.43] * 4)
    return dict(timestep=0.01, w_x=w_x, w_u=np.eye(nv - 6) * 1e-4, w_cent=np.diag([0.04, 0.04, 0, 0, 0, 0.0]),
                w_forces=np.eye(3) * 1e-4, w_frame=np.eye(3) * 1000.0, gravity=np.array([0, 0, -9.81]),
                Kp_correction=np.zeros(3), Kd_correction=np.zeros(3), umin=-eff, umax=eff, qmin=robot.q_lo.copy(),
                qmax=robot.q_hi.copy(), torque_limits=True, kinematics_limits=True)


class Full:
    """Stage model of the full-dynamics OCP (oracle/orc_fulldyn.hpp); u_ref = [control reference ; force reference per foot]."""

    def __init__(self, robot, s):
        L = lib()
        self.robot, self.s = robot, s
        c = lambda a: np.ascontiguousarray(a, float)
        self.h = L.orc_full_create(robot.ptr, s["timestep"], c(s["w_x"]), c(s["w_u"]), c(s["w_cent"]), c(s["w_forces"]),
                                   c(s["w_frame"]), c(s["gravity"]), c(s["Kp_correction"]), c(s["Kd_correction"]), c(s["umin"]),
                                   c(s["umax"]), c(s["qmin"]), c(s["qmax"]), int(s["torque_limits"]), int(s["kinematics_limits"]),
                                   int(s.get("force_size", 3)), int(s.get("force_cone", False)), float(s.get("mu", 0.8)),
                                   float(s.get("Lfoot", 0.1)), float(s.get("Wfoot", 0.075)))
        if s.get("land_cstr", False):
            L.orc_full_set_land_cstr(self.h, 1)
        d = np.zeros(5, np.int32)
        L.orc_full_dims(self.h, d)
        self.nx, self.ndx, self.nu, self.nc, self.nf = (int(v) for v in d)
        self.nv = robot.nv

    def eval(self, mask, u_ref, x_tgt, foot_ref, x, u):
        c = lambda a: np.ascontiguousarray(a, float)
        xnext, xdot, cost, cc = np.zeros(self.nx), np.zeros(2 * self.nv), np.zeros(1), np.zeros(self.nc)
        lib().orc_full_eval(self.h, mask, c(u_ref), c(x_tgt), c(foot_ref), c(x), c(u), xnext, xdot, cost, cc)
        return dict(xnext=xnext, xdot=xdot, cost=float(cost[0]), c=cc)

    def deriv(self, mask, u_ref, x_tgt, foot_ref, x, u):
        c = lambda a: np.ascontiguousarray(a, float)
        n, m, k = self.ndx, self.nu, self.nc
        o = dict(A=np.zeros((n, n)), B=np.zeros((n, m)), lx=np.zeros(n), lu=np.zeros(m), Lxx=np.zeros((n, n)),
                 Lxu=np.zeros((n, m)), Luu=np.zeros((m, m)), Cx=np.zeros((k, n)), Cu=np.zeros((k, m)))
        lib().orc_full_deriv(self.h, mask, c(u_ref), c(x_tgt), c(foot_ref), c(x), c(u), *o.values())
        return o

    def row_kinds(self, mask):
        k, lo, hi = np.zeros(self.nc, np.int32), np.zeros(self.nc), np.zeros(self.nc)
        lib().orc_full_row_kinds(self.h, int(mask), k, lo, hi)
        return k, lo, hi

    def term(self, x_tgt, x):
        """Terminal cost (state + 10 x centroidal)."""
        return float(lib().orc_full_term(self.h, np.ascontiguousarray(x_tgt, float), np.ascontiguousarray(x, float)))

    def solve(self, masks, u_ref, x_tgt, foot_ref, x0, u0, max_iter=50, tol=1e-4, mu=1e-8):
        c = lambda a: np.ascontiguousarray(a, float)
        H = len(masks)
        mk = (C.c_uint * H)(*[int(m) for m in masks])
        trace, xs, us = np.zeros((max_iter, 6)), np.zeros((H + 1, self.nx)), np.zeros((H, self.nu))
        it = lib().orc_full_solve(self.h, H, mk, c(u_ref), c(x_tgt), c(foot_ref), c(x0), c(u0), max_iter, tol, mu, trace, xs, us)
        return dict(iters=it, trace=trace[:it], xs=xs, us=us)


def talos_full_settings(robot):
    """FullDynamicsSettings of the Talos example: reference examples/talos_fulldynamics.py:47-115 (6-D feet, wrench cones).  The
    robot table holds no effort limits: Talos-like actuator limits are used (legs 100/160/160/300/160/100, torso 200, arms
    44/44/22/22 N m)."""
    w_x = np.diag(np.array([0, 0, 0, 10, 10, 10] + [0.1] * 6 * 2 + [1, 100] + [1, 1, 10, 10] * 2 + [10] * 6 + [1] * 6 * 2 + [1, 100]
                           + [10] * 4 * 2, float))
    eff = np.array([100, 160, 160, 300, 160, 100] * 2 + [200, 200] + [44, 44, 22, 22] * 2, float)
    nu = robot.nv - 6
    return dict(timestep=0.01, w_x=w_x, w_u=np.eye(nu) * 1e-4, w_cent=np.diag([0.1, 0.1, 10, 0.1, 0.1, 10.0]), w_forces=np.eye(6) * 1e-3,
                w_frame=np.eye(6) * 2000.0, gravity=np.array([0, 0, -9.81]), force_size=6, Kp_correction=np.array([0, 0, 50, 0, 0, 0.0]),
                Kd_correction=np.ones(6) * 100.0, umin=-eff, umax=eff, qmin=robot.q_lo.copy(), qmax=robot.q_hi.copy(), mu=0.8, Lfoot=0.1,
                Wfoot=0.075, torque_limits=True, kinematics_limits=True, force_cone=True, land_cstr=False)


def talos_kino_settings(robot, force_cone=True):
    """KinodynamicsSettings of the reference's Talos configuration with 6-D feet: weights of examples/talos_kinodynamics.py:50-106
    (force_cone as in tests/test_utils.cpp:147-197, which the reference's own problem / MPC tests use)."""
    nv = robot.nv
    w_x = 10.0 * np.diag(np.array([0, 0, 1000, 1000, 1000, 1000] + [0.1] * 6 * 2 + [1, 1000] + [1, 1, 10, 10] * 2 + [10] * 6 + [1] * 6 * 2
                                  + [0.1, 100] + [10] * 4 * 2, float))
    w_u = np.diag(np.concatenate([[0.001, 0.001, 0.01], np.ones(3) * 0.1] * 2 + [np.ones(nv - 6) * 1e-4]))
    return dict(timestep=0.01, w_x=w_x, w_u=w_u, w_cent=np.diag([0.0, 0.0, 1.0, 0.1, 0.1, 10.0]), w_centder=np.diag([0.0, 0.0, 0.0, 0.1, 0.1, 0.1]),
                gravity=np.array([0.0, 0.0, -9.81]), force_size=6, w_frame=np.eye(6) * 100000.0, qmin=robot.q_lo.copy(), qmax=robot.q_hi.copy(),
                mu=0.8, Lfoot=0.1, Wfoot=0.075, kinematics_limits=True, force_cone=bool(force_cone), land_cstr=False)


def talos_centroidal_settings(robot):
    """CentroidalSettings of the reference's Talos configuration with 6-D feet: examples/talos_centroidal.py:50-76 =
    tests/test_utils.cpp:199-218 but for w_u (identity there; the example's force / torque weights are used)."""
    nf = robot.nf
    return dict(timestep=0.01, w_u=np.diag([0.001] * 3 + [0.1] * 3) if nf == 1 else np.diag(([0.001] * 3 + [0.1] * 3) * nf), w_com=np.zeros((3, 3)), w_linear_mom=np.diag([0.01, 0.01, 100.0]),
                w_angular_mom=np.diag([0.1, 0.1, 1000.0]), w_linear_acc=0.01 * np.eye(3), w_angular_acc=0.01 * np.eye(3),
                gravity=np.array([0.0, 0.0, -9.81]), mu=0.8, Lfoot=0.1, Wfoot=0.075, force_size=6)


def talos_mpc_settings(robot, max_iters=1, num_threads=0):
    """MPC settings of the Talos example: reference examples/talos_fulldynamics.py:101-113 (T = 100, T_fly 80, T_contact 20)."""
    return dict(support_force=robot.mass * 9.81, TOL=1e-4, mu_init=1e-8, max_iters=max_iters, num_threads=num_threads, swing_apex=0.15,
                T_fly=80, T_contact=20, timestep=0.01, T=100)


def walk_cycle(T_ds=20, T_ss=80):
    """Biped contact cycle of reference examples/talos_fulldynamics.py:117-136, order left, right."""
    return np.array([[1, 1]] * T_ds + [[1, 0]] * T_ss + [[1, 1]] * T_ds + [[0, 1]] * T_ss, np.uint8)


def riccati(Q, S, R, q, r, A, B, f, Cm, D, d, QN, qN, mu):
    H, ndx, nu = B.shape
    nc = Cm.shape[1]
    c = lambda a: np.ascontiguousarray(a, float)
    dxs, dus, dvs, dlams = np.zeros((H + 1, ndx)), np.zeros((H, nu)), np.zeros((H, nc)), np.zeros((H + 1, ndx))
    Ks = np.zeros((H, nu, ndx))
    lib().orc_riccati(
        H, ndx, nu, nc, mu, c(Q), c(S), c(R), c(q), c(r), c(A), c(B), c(f), c(Cm), c(D), c(d), c(QN), c(qN), dxs, dus,
        dvs, dlams, Ks,
    )
    return dxs, dus, dvs, dlams, Ks


class Timer:
    def __init__(self, cs, H):
        cs = np.ascontiguousarray(cs, np.uint8)
        self.nf = cs.shape[1]
        self.h = lib().orc_timer_create(cs, cs.shape[0], cs.shape[1], H)

    def get(self, foot, which):
        out = np.zeros(64, np.int32)
        n = lib().orc_timer_get(self.h, foot, which, out, 64)
        return [int(v) for v in out[:n]]

    def recede(self):
        lib().orc_timer_recede(self.h)


class OracleMPC:
    """Batched restatement of simple_mpc.MPC (reference: include/simple-mpc/mpc.hpp:55-197)."""

    _p = "orc_mpc_"

    def _f(self, name):
        return getattr(lib(), self._p + name)

    def _create(self, s, B, gravity_arg):
        return lib().orc_mpc_create(self.kino.h, C.byref(s), B, gravity_arg)

    def __init__(self, kino, mpc_settings, B, gravity_arg=-9.81):
        self.kino = kino
        self.nx_in = kino.nx
        self.B = B
        self.H = mpc_settings["T"]
        s = MpcSettingsC(
            mpc_settings["swing_apex"], mpc_settings["support_force"], mpc_settings["TOL"], mpc_settings["mu_init"],
            mpc_settings["timestep"], mpc_settings["max_iters"], mpc_settings.get("num_threads", 0),
            mpc_settings["T_fly"], mpc_settings["T_contact"], mpc_settings["T"],
        )
        lib().orc_set_terminal_constraint(int(bool(mpc_settings.get("terminal_constraint", False))))  # createProblem's last argument
        try:
            self.h = self._create(s, B, gravity_arg)
        finally:
            lib().orc_set_terminal_constraint(0)

    def generateCycleHorizon(self, cs):
        cs = np.ascontiguousarray(cs, np.uint8)
        self._f("generate_cycle")(self.h, cs, cs.shape[0])

    def switchToWalk(self, v6):
        self._f("switch_to_walk")(self.h, np.ascontiguousarray(v6, float))

    def switchToStand(self):
        self._f("switch_to_stand")(self.h)

    def set_stage_reference(self, t, what, v):
        """OCPHandler setReferenceControl (what = 0) / setReferenceState (what = 1) on stage t, all instances."""
        self._f("set_stage_reference")(self.h, int(t), int(what), np.ascontiguousarray(v, float))

    def setVelocityBaseBatched(self, V):
        V = np.ascontiguousarray(V, float)
        assert V.shape == (self.B, 6)
        self._f("set_velocity_batched")(self.h, V)

    def iterate(self, X):
        X = np.ascontiguousarray(X, float)
        assert X.shape == (self.B, self.nx_in)
        return self._f("iterate")(self.h, X)

    def setEarlyExitOnTol(self, on=True):
        lib().orc_mpc_set_early_exit.argtypes = [C.c_void_p, C.c_int]
        lib().orc_mpc_set_early_exit.restype = None
        lib().orc_mpc_set_early_exit(self.h, int(on))

    def _get(self, what, shape):
        out = np.zeros(shape)
        self._f("get")(self.h, what, out)
        return out

    @property
    def xs(self):
        return self._get(0, (self.B, self.H + 1, self.kino.nx))

    @property
    def us(self):
        return self._get(1, (self.B, self.H, self.kino.nu))

    @property
    def K0(self):
        return self._get(2, (self.B, self.kino.nu, self.kino.ndx))

    @property
    def vs(self):
        return self._get(3, (self.B, self.H, self.kino.nc))

    @property
    def lams(self):
        return self._get(4, (self.B, self.H + 1, self.kino.ndx))

    @property
    def foot_refs(self):
        return self._get(5, (self.B, self.H, self.kino.nf, 3))

    @property
    def info(self):
        return self._get(6, (self.B, 12))

    @property
    def xdot(self):
        return self._get(7, (self.B, self.H, 2 * self.kino.nv))

    @property
    def terminal(self):
        """Terminal constraint state: multipliers [B][3], DCM reference [B][3], tau [B]."""
        o = self._get(9, (self.B, 7))
        return o[:, :3], o[:, 3:6], o[:, 6]

    def keep_knots(self, on=True):
        self._f("keep_knots")(self.h, int(on))

    def knot(self, b, t):
        k = self.kino
        ndx, nu, nc = k.ndx, k.nu, k.nc
        out = np.zeros(2 * ndx * ndx + 2 * ndx * nu + nu * nu + nc * ndx + 2 * ndx + nu + nc)
        n = self._f("get_knot")(self.h, b, t, out)
        assert n == out.size, n
        res, o = {}, 0
        for name, shape in (
            ("A", (ndx, ndx)), ("B", (ndx, nu)), ("Q", (ndx, ndx)), ("S", (ndx, nu)), ("R", (nu, nu)), ("C", (nc, ndx)),
            ("q", (ndx,)), ("r", (nu,)), ("f", (ndx,)), ("d", (nc,)),
        ):
            sz = int(np.prod(shape))
            res[name] = out[o : o + sz].reshape(shape).copy()
            o += sz
        return res

    def cold_trace(self):
        n = self._f("cold_iters")(self.h)
        out = np.zeros((n, 4))
        self._f("cold_trace")(self.h, out)
        return out

    def timing(self, foot, which):
        out = np.zeros(64, np.int32)
        n = self._f("timing")(self.h, foot, which, out, 64)
        return [int(v) for v in out[:n]]



def go2_centroidal_settings(robot):
    """CentroidalSettings for the BASELINE "Go2 centroidal, H=50" configuration.  The reference ships no Go2 centroidal
    script: the weights are those of its centroidal example (examples/talos_centroidal.py:50-76) with 3-D contact forces."""
    nf = robot.nf
    return dict(
        timestep=0.01,
        w_u=np.diag(np.ones(3 * nf) * 0.001),
        w_com=np.zeros((3, 3)),
        w_linear_mom=np.diag([0.01, 0.01, 100.0]),
        w_angular_mom=np.diag([0.1, 0.1, 1000.0]),
        w_linear_acc=0.01 * np.eye(3),
        w_angular_acc=0.01 * np.eye(3),
        gravity=np.array([0.0, 0.0, -9.81]),
        mu=0.8,
        Lfoot=0.01,
        Wfoot=0.01,
        force_size=3,
    )


class Cent:
    """Oracle centroidal stage model (oracle/orc_cent.hpp)."""

    def __init__(self, robot, s):
        c = lambda a: np.ascontiguousarray(a, float)
        self.robot, self.s = robot, s
        self.fs = int(s.get("force_size", 3))
        if self.fs == 6:
            self.h = lib().orc_cent_create6(
                robot.ptr, s["timestep"], c(s["w_u"]), c(s["w_com"]), c(s["w_linear_mom"]), c(s["w_angular_mom"]),
                c(s["w_linear_acc"]), c(s["w_angular_acc"]), c(s["gravity"]), float(s["mu"]), float(s["Lfoot"]), float(s["Wfoot"]),
            )
        else:
            self.h = lib().orc_cent_create(
                robot.ptr, s["timestep"], c(s["w_u"]), c(s["w_com"]), c(s["w_linear_mom"]), c(s["w_angular_mom"]),
                c(s["w_linear_acc"]), c(s["w_angular_acc"]), c(s["gravity"]), float(s["mu"]),
            )
        self.nf = robot.nf
        self.nx = self.ndx = self.nv = 9
        self.nu, self.nc = self.fs * self.nf, (17 if self.fs == 6 else 2) * self.nf

    def eval(self, mask, u_ref, x_tgt, pos, x, u):
        c = lambda a: np.ascontiguousarray(a, float)
        xnext, xdot, cost, cc = np.zeros(9), np.zeros(9), np.zeros(1), np.zeros(self.nc)
        lib().orc_cent_eval(self.h, mask, c(u_ref), c(x_tgt), c(pos), c(x), c(u), xnext, xdot, cost, cc)
        return dict(xnext=xnext, xdot=xdot, cost=float(cost[0]), c=cc)

    def deriv(self, mask, u_ref, x_tgt, pos, x, u):
        c = lambda a: np.ascontiguousarray(a, float)
        n, m, k = 9, self.nu, self.nc
        o = dict(
            A=np.zeros((n, n)), B=np.zeros((n, m)), lx=np.zeros(n), lu=np.zeros(m), Lxx=np.zeros((n, n)),
            Lxu=np.zeros((n, m)), Luu=np.zeros((m, m)), Cx=np.zeros((k, n)), Cu=np.zeros((k, m)),
        )
        lib().orc_cent_deriv(
            self.h, mask, c(u_ref), c(x_tgt), c(pos), c(x), c(u), o["A"], o["B"], o["lx"], o["lu"], o["Lxx"], o["Lxu"],
            o["Luu"], o["Cx"], o["Cu"],
        )
        return o

    def term(self, x):
        cost, lx, Lxx = np.zeros(1), np.zeros(9), np.zeros((9, 9))
        lib().orc_cent_term(self.h, np.ascontiguousarray(x, float), cost, lx, Lxx)
        return float(cost[0]), lx, Lxx

    def row_kinds(self, mask):
        k, lo, hi = np.zeros(self.nc, np.int32), np.zeros(self.nc), np.zeros(self.nc)
        lib().orc_cent_row_kinds(self.h, int(mask), k, lo, hi)
        return k, lo, hi

    def solve(self, masks, u_ref, x_tgt, pos, x0, u0, max_iter=200, tol=1e-9, mu=1e-8):
        """The oracle's ProxDDP run to convergence on an H-stage centroidal problem with per-stage references (u_ref [H][nu], x_tgt [H][9],
        contact positions pos [H][nf*3]) from the constant guess (x0, u0)."""
        c = lambda a: np.ascontiguousarray(a, float)
        H = len(masks)
        mk = (C.c_uint * H)(*[int(m) for m in masks])
        trace, xs, us = np.zeros((max_iter, 6)), np.zeros((H + 1, 9)), np.zeros((H, self.nu))
        vs, lams = np.zeros((H, self.nc)), np.zeros((H + 1, 9))
        it = lib().orc_cent_solve(self.h, H, mk, c(u_ref), c(x_tgt), c(pos), c(x0), c(u0), max_iter, tol, mu, trace, xs, us, vs, lams)
        return dict(iters=it, trace=trace[:it], xs=xs, us=us, vs=vs, lams=lams)


class OracleCentMPC(OracleMPC):
    """simple_mpc.MPC over a CentroidalOCP, batched (oracle/orc_mpc_cent.hpp).  iterate() takes the measured MULTIBODY
    states [B][nq + nv] like the reference; xs are centroidal states [B][H+1][9]."""

    _p = "orc_cmpc_"

    def _create(self, s, B, gravity_arg):
        return lib().orc_cmpc_create(self.kino.h, self.kino.robot.ptr, C.byref(s), B, gravity_arg)

    def __init__(self, cent, mpc_settings, B, gravity_arg=-9.81):
        OracleMPC.__init__(self, cent, mpc_settings, B, gravity_arg)
        self.nx_in = cent.robot.nq + cent.robot.nv

    def set_x_reference(self, x9):
        lib().orc_cmpc_set_x_reference(self.h, np.ascontiguousarray(x9, float))

    @property
    def xdot(self):
        return self._get(7, (self.B, self.H, 18))[:, :, :9]


class OracleFullMPC(OracleMPC):
    """simple_mpc.MPC over a FullDynamicsOCP (3-D feet), batched: the host state machine of orc_mpc.hpp on the stage model of
    oracle/orc_fulldyn.hpp.  us are joint torques [B][H][nv - 6]."""

    _p = "orc_fmpc_"

    def _create(self, s, B, gravity_arg):
        return lib().orc_fmpc_create(self.kino.h, C.byref(s), B, gravity_arg)


def interpolate(kind, nv, delay, timestep, knots):
    """Oracle restatement of the reference Interpolator (src/interpolator.cpp:5-78): kind 0 state, 1 configuration, 2 linear."""
    k = np.ascontiguousarray(np.array(knots, dtype=np.float64))
    out = np.zeros(k.shape[1])
    lib().orc_interpolate(kind, nv, float(delay), float(timestep), k, k.shape[0], k.shape[1], out)
    return out


def centroidal_dynamics(mass, gravity, dt, x, u, contact, pos):
    """Oracle: CentroidalFwdDynamics + IntegratorEuler for one instance -> (xnext[9], A[9,9], B[9,3nf])."""
    nf = len(contact)
    xn, A, B = np.zeros(9), np.zeros((9, 9)), np.zeros((9, 3 * nf))
    lib().orc_centroidal_dynamics(float(mass), np.ascontiguousarray(gravity, float), float(dt), nf, np.ascontiguousarray(x, float),
                                  np.ascontiguousarray(u, float), np.ascontiguousarray(contact, np.uint8),
                                  np.ascontiguousarray(pos, float), xn, A, B)
    return xn, A, B


ID_DEFAULTS = dict(friction_coefficient=0.6, contact_weight_ratio_max=10.0, contact_weight_ratio_min=0.01, kp_base=0.0, kp_posture=0.0,
                   kp_contact=0.0, w_base=-1.0, w_posture=-1.0, w_contact_motion=-1.0, w_contact_force=-1.0, contact_motion_equality=False,
                   admm_iters=400, rho=0.1, sigma=1e-6, alpha=1.6, admm_tol=1e-7,  # reference include/simple-mpc/inverse-dynamics/kinodynamics-id.hpp:24-50
                   centroidal=False, kp_com=0.0, kp_feet_tracking=0.0, w_com=-1.0, w_feet_tracking=-1.0,  # centroidal-id.hpp:17-26
                   base_reference_as_coded=False, tsid_joint_bounds=False,
                   force_size=3, quad_points=None)  # 6-D feet (tsid Contact6d): force_size 6 and the corners of the soles [nf][4][3]
TALOS_EFFORT = np.array([100, 160, 160, 300, 160, 100] * 2 + [200, 200] + [44, 44, 22, 22] * 2, float)
TALOS_VMAX = np.array([3.87, 5.86, 5.86, 7.0, 5.86, 4.8] * 2 + [5.4, 5.4] + [2.7, 3.66, 4.58, 4.58] * 2, float)
TALOS_QUAD = np.array([[0.1, 0.075, 0], [-0.1, 0.075, 0], [-0.1, -0.075, 0], [0.1, -0.075, 0]])


def talos_id_settings(robot, control_dt=1e-3, **kw):
    """KinodynamicsID::Settings for the Talos-class robot with flat feet (reference tests/inverse-dynamics/kinodynamics-id.cpp:192-236 use
    getTalosModelHandler(): soles of 0.2 x 0.15 m; the robot table holds no effort / velocity limits: Talos-like values)."""
    return id_settings(robot, control_dt, tau_max=TALOS_EFFORT, v_max=TALOS_VMAX, force_size=6, quad_points=np.tile(TALOS_QUAD, (robot.nf, 1, 1)), **kw)
GO2_EFFORT = np.array([23.7, 23.7, 45.43] * 4)
GO2_VMAX = np.array([30.1, 30.1, 15.7] * 4)


def id_settings(robot, control_dt=1e-3, tau_max=None, v_max=None, **kw):
    """KinodynamicsID::Settings + what the reference reads from the pinocchio model (effort / velocity / position limits)."""
    s = dict(ID_DEFAULTS, control_dt=control_dt, tau_max=GO2_EFFORT if tau_max is None else tau_max, v_max=GO2_VMAX if v_max is None else v_max,
             q_min=robot.q_lo.copy(), q_max=robot.q_hi.copy())
    s.update(kw)
    return s


class OracleKinoID:
    """Batched restatement of simple_mpc.KinodynamicsID (oracle/orc_id.hpp)."""

    def __init__(self, robot, s, B):
        self.robot, self.B = robot, B
        self._keep = [np.ascontiguousarray(s[k], float) for k in ("tau_max", "v_max", "q_min", "q_max")]
        c = IdSettingsC(s["friction_coefficient"], s["contact_weight_ratio_max"], s["contact_weight_ratio_min"], s["kp_base"], s["kp_posture"],
                        s["kp_contact"], s["w_base"], s["w_posture"], s["w_contact_motion"], s["w_contact_force"],
                        int(s["contact_motion_equality"]), s["control_dt"], *[a.ctypes.data for a in self._keep], int(s["admm_iters"]),
                        s["rho"], s["sigma"], s["alpha"], s["admm_tol"], int(s["centroidal"]), s["kp_com"], s["kp_feet_tracking"], s["w_com"], s["w_feet_tracking"],
                        int(s["base_reference_as_coded"]), int(s["tsid_joint_bounds"]), int(s.get("force_size", 3)), None)
        self.fs = int(s.get("force_size", 3))
        if self.fs == 6:
            self._quad = np.ascontiguousarray(s["quad_points"], float).reshape(robot.nf, 4, 3)
            c.quad_points = self._quad.ctypes.data
        self.h = lib().orc_id_create(robot.ptr, C.byref(c), B)
        nfv, nmot, nfr = (12, 6, 17) if self.fs == 6 else (3, 3, 4)
        self.nfw = 6 if self.fs == 6 else 3
        self.n = robot.nv + nfv * robot.nf
        self.m = self.n + 6 + nmot * robot.nf + nfr * robot.nf + robot.nv - 6

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_id_destroy(self.h)
            self.h = None

    def setTarget(self, q, v, a, contact_state, f, instance=-1):
        mask = sum(1 << i for i, on in enumerate(contact_state) if on)
        c = lambda x: np.ascontiguousarray(x, float)
        lib().orc_id_set_target(self.h, instance, c(q), c(v), c(a), mask, c(np.asarray(f, float).reshape(-1)))

    def setTargetCentroidal(self, com, vcom, feet_p, feet_v, contact_state, f, instance=-1):
        mask = sum(1 << i for i, on in enumerate(contact_state) if on)
        c = lambda x: np.ascontiguousarray(np.asarray(x, float).reshape(-1))
        lib().orc_id_set_target_centroidal(self.h, instance, c(com), c(vcom), c(feet_p), c(feet_v), mask, c(f))

    def solve(self, X):
        rb = self.robot
        X = np.ascontiguousarray(X, float)
        tau, a, f, res = np.zeros((self.B, rb.nv - 6)), np.zeros((self.B, rb.nv)), np.zeros((self.B, self.nfw * rb.nf)), np.zeros(self.B)
        lib().orc_id_solve(self.h, X, tau, a, f, res)
        self.resid = res
        return tau, a, f

    def qp(self, b, x):
        H, g = np.zeros((self.n, self.n)), np.zeros(self.n)
        Cm, l, u = np.zeros((self.m, self.n)), np.zeros(self.m), np.zeros(self.m)
        m = lib().orc_id_qp(self.h, b, np.ascontiguousarray(x, float), H, g, Cm, l, u)
        assert m == self.m
        return H, g, Cm, l, u


def id_quantities6(robot, x):
    """LOCAL 6-D rows of flat feet: J (6 nf x nv), their drift and the frame velocities (6 nf)."""
    nv, nf = robot.nv, robot.nf
    M, nle, J, Jdv, vf = np.zeros((nv, nv)), np.zeros(nv), np.zeros((6 * nf, nv)), np.zeros(6 * nf), np.zeros(6 * nf)
    lib().orc_id_quantities6(robot.ptr, np.ascontiguousarray(x, float), M, nle, J, Jdv, vf)
    return dict(M=M, nle=nle, J=J, Jdv=Jdv, vfoot=vf)


def id_quantities(robot, x):
    nv, nf = robot.nv, robot.nf
    M, nle, J, Jdv, vf = np.zeros((nv, nv)), np.zeros(nv), np.zeros((3 * nf, nv)), np.zeros(3 * nf), np.zeros(3 * nf)
    lib().orc_id_quantities(robot.ptr, np.ascontiguousarray(x, float), M, nle, J, Jdv, vf)
    return dict(M=M, nle=nle, J=J, Jdv=Jdv, vfoot=vf)
