"""ctypes binding of include/smpc.h (libsmpc_hip.so).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).  There is
no CPU implementation behind it: loading fails loudly if the library is missing, and every compute
entry point fails if no HIP device is visible.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.normpath(os.path.join(_HERE, "..", "..", "csrc", "libsmpc_hip.so"))

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_lp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_bp = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class KinodynamicsSettingsC(C.Structure):
    _fields_ = [
        ("timestep", C.c_double),
        ("w_x", C.c_void_p),
        ("w_u", C.c_void_p),
        ("w_frame", C.c_void_p),
        ("w_cent", C.c_void_p),
        ("w_centder", C.c_void_p),
        ("qmin", C.c_void_p),
        ("qmax", C.c_void_p),
        ("gravity", C.c_double * 3),
        ("mu", C.c_double),
        ("Lfoot", C.c_double),
        ("Wfoot", C.c_double),
        ("force_size", C.c_int),
        ("kinematics_limits", C.c_int),
        ("force_cone", C.c_int),
        ("land_cstr", C.c_int),
        ("terminal_constraint", C.c_int),
    ]


class CentroidalSettingsC(C.Structure):
    _fields_ = [
        ("timestep", C.c_double),
        ("w_u", C.c_void_p),
        ("w_com", C.c_void_p),
        ("w_linear_mom", C.c_void_p),
        ("w_angular_mom", C.c_void_p),
        ("w_linear_acc", C.c_void_p),
        ("w_angular_acc", C.c_void_p),
        ("gravity", C.c_double * 3),
        ("mu", C.c_double),
        ("Lfoot", C.c_double),
        ("Wfoot", C.c_double),
        ("force_size", C.c_int),
    ]


class FullDynamicsSettingsC(C.Structure):
    _fields_ = [
        ("timestep", C.c_double),
        ("w_x", C.c_void_p),
        ("w_u", C.c_void_p),
        ("w_cent", C.c_void_p),
        ("w_forces", C.c_void_p),
        ("w_frame", C.c_void_p),
        ("umin", C.c_void_p),
        ("umax", C.c_void_p),
        ("qmin", C.c_void_p),
        ("qmax", C.c_void_p),
        ("Kp_correction", C.c_void_p),
        ("Kd_correction", C.c_void_p),
        ("gravity", C.c_double * 3),
        ("mu", C.c_double),
        ("Lfoot", C.c_double),
        ("Wfoot", C.c_double),
        ("force_size", C.c_int),
        ("torque_limits", C.c_int),
        ("kinematics_limits", C.c_int),
        ("force_cone", C.c_int),
        ("land_cstr", C.c_int),
        ("terminal_constraint", C.c_int),
    ]


class MpcSettingsC(C.Structure):
    _fields_ = [
        ("swing_apex", C.c_double),
        ("support_force", C.c_double),
        ("TOL", C.c_double),
        ("mu_init", C.c_double),
        ("max_iters", C.c_int),
        ("num_threads", C.c_int),
        ("T_fly", C.c_int),
        ("T_contact", C.c_int),
        ("T", C.c_int),
        ("timestep", C.c_double),
    ]


MAXJ, MAXF, NAME = 32, 4, 32


class IdSettingsC(C.Structure):
    """Mirror of smpc_id_settings (include/smpc.h)."""

    _fields_ = [
        ("friction_coefficient", C.c_double), ("contact_weight_ratio_max", C.c_double), ("contact_weight_ratio_min", C.c_double),
        ("kp_base", C.c_double), ("kp_posture", C.c_double), ("kp_contact", C.c_double),
        ("w_base", C.c_double), ("w_posture", C.c_double), ("w_contact_motion", C.c_double), ("w_contact_force", C.c_double),
        ("contact_motion_equality", C.c_int), ("control_dt", C.c_double),
        ("effort_limit", C.c_void_p), ("velocity_limit", C.c_void_p), ("q_min", C.c_void_p), ("q_max", C.c_void_p),
        ("admm_iters", C.c_int), ("admm_rho", C.c_double), ("admm_sigma", C.c_double), ("admm_alpha", C.c_double), ("admm_tol", C.c_double),
        ("centroidal", C.c_int), ("kp_com", C.c_double), ("kp_feet_tracking", C.c_double), ("w_com", C.c_double), ("w_feet_tracking", C.c_double),
        ("base_reference_as_coded", C.c_int), ("tsid_joint_bounds", C.c_int),
        ("force_size", C.c_int), ("quad_contact_points", C.c_void_p),
    ]


class RobotModelC(C.Structure):
    """Mirror of smpc_robot_model (include/smpc_robot.h)."""

    _fields_ = [
        ("name", C.c_char * NAME),
        ("njoints", C.c_int),
        ("nq", C.c_int),
        ("nv", C.c_int),
        ("parent", C.c_int * MAXJ),
        ("jtype", C.c_int * MAXJ),
        ("jp_R", (C.c_double * 9) * MAXJ),
        ("jp_p", (C.c_double * 3) * MAXJ),
        ("mass", C.c_double * MAXJ),
        ("com", (C.c_double * 3) * MAXJ),
        ("inertia", (C.c_double * 6) * MAXJ),
        ("nfeet", C.c_int),
        ("foot_name", (C.c_char * NAME) * MAXF),
        ("foot_joint", C.c_int * MAXF),
        ("foot_p", (C.c_double * 3) * MAXF),
        ("foot_ref_p", (C.c_double * 3) * MAXF),
        ("q_ref", C.c_double * (MAXJ + 6)),
        ("q_lo", C.c_double * MAXJ),
        ("q_hi", C.c_double * MAXJ),
        ("total_mass", C.c_double),
    ]


# every symbol declared in include/smpc.h
SYMBOLS = [
    "smpc_builtin_robot", "smpc_last_error", "smpc_device_count", "smpc_create", "smpc_create_centroidal", "smpc_create_fulldynamics", "smpc_get_contact_forces", "smpc_destroy", "smpc_get_dims",
    "smpc_generate_cycle_horizon", "smpc_switch_to_walk", "smpc_switch_to_stand", "smpc_set_velocity_base_batched", "smpc_set_stage_reference", "smpc_get_stage_reference", "smpc_set_reference_pose",
    "smpc_get_reference_pose", "smpc_set_reference_pose_se3", "smpc_get_reference_pose_se3", "smpc_get_contact_state", "smpc_get_cycling_contact_state", "smpc_debug_get_extra_multipliers", "smpc_set_x_reference",
    "smpc_state_size", "smpc_save_state", "smpc_load_state", "smpc_iterate", "smpc_iterate_device", "smpc_wait", "smpc_get_stream", "smpc_get_x_device", "smpc_get_xs", "smpc_get_us", "smpc_get_K0", "smpc_get_Ks",
    "smpc_get_vs", "smpc_get_lams", "smpc_get_state_derivative01", "smpc_get_reference_poses",
    "smpc_set_early_exit_on_tol", "smpc_iterate_async", "smpc_gather_outputs", "smpc_gather_outputs_device", "smpc_gather_outputs_peer", "smpc_get_foot_timing", "smpc_get_info", "smpc_get_status", "smpc_get_cold_trace", "smpc_lq_size", "smpc_debug_get_lq",
    "smpc_debug_get_steps", "smpc_debug_get_terminal", "smpc_debug_get_phase_cycles", "smpc_set_profiling", "smpc_get_kernel_times", "smpc_get_kernel_times_n", "smpc_kernel_time_slots", "smpc_reset_kernel_times",
    "smpc_interpolate", "smpc_interpolate_knots", "smpc_friction_compensation", "smpc_update_internal_data", "smpc_full_forward_dynamics", "smpc_centroidal_dynamics", "smpc_riccati_feedback",
    "smpc_id_create", "smpc_id_destroy", "smpc_id_set_target", "smpc_id_set_targets", "smpc_id_set_target_centroidal", "smpc_id_set_targets_centroidal", "smpc_id_solve", "smpc_id_solve_device", "smpc_id_wait", "smpc_id_get_resid", "smpc_id_reset", "smpc_id_get_tau_device", "smpc_id_get_x_device", "smpc_id_set_targets_from_mpc", "smpc_id_share_stream", "smpc_sim_step_device", "smpc_id_debug_get",
]


class SmpcLib:
    def __init__(self, path=None):
        if path is None and os.environ.get("SMPC_LIB_PATH"):
            path = os.environ["SMPC_LIB_PATH"]  # experiment builds (tools/variant_build.sh): say so, a stale variant must not pass for the product
            print("simple_mpc: SMPC_LIB_PATH overrides the shipped library: %s" % path, file=sys.stderr)
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise RuntimeError(
                "simple_mpc: native library %s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950).  There is no CPU fallback." % path
            )
        self.path = path
        # PyTorch-ROCm wheels carry their own copy of the HIP runtime; if this library initialises the system runtime first, torch's copy
        # then reports "No HIP GPUs are available".  Loading torch first (when it is installed) makes either import order work.
        if os.path.basename(path).startswith("libsmpc_hip") and "torch" not in sys.modules and not os.environ.get("SMPC_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(path)
        self.L = L
        vp = C.c_void_p
        L.smpc_builtin_robot.restype = C.POINTER(RobotModelC)
        L.smpc_builtin_robot.argtypes = [C.c_char_p]
        L.smpc_last_error.restype = C.c_char_p
        L.smpc_device_count.restype = C.c_int
        L.smpc_create.argtypes = [
            C.POINTER(RobotModelC), C.POINTER(KinodynamicsSettingsC), C.POINTER(MpcSettingsC), C.c_int, C.c_double,
            C.c_int, C.POINTER(vp),
        ]
        L.smpc_create_centroidal.argtypes = [
            C.POINTER(RobotModelC), C.POINTER(CentroidalSettingsC), C.POINTER(MpcSettingsC), C.c_int, C.c_double,
            C.c_int, C.POINTER(vp),
        ]
        L.smpc_create_fulldynamics.argtypes = [
            C.POINTER(RobotModelC), C.POINTER(FullDynamicsSettingsC), C.POINTER(MpcSettingsC), C.c_int, C.c_double,
            C.c_int, C.POINTER(vp),
        ]
        L.smpc_get_contact_forces.argtypes = [vp, _dp]
        L.smpc_set_early_exit_on_tol.argtypes = [vp, C.c_int]
        L.smpc_iterate_async.argtypes = [vp, _dp]
        L.smpc_gather_outputs.argtypes = [vp, C.c_void_p, C.c_size_t]
        L.smpc_gather_outputs_device.argtypes = [vp, C.c_void_p, C.c_size_t]
        L.smpc_gather_outputs_peer.argtypes = [vp, C.c_void_p, C.c_int]
        L.smpc_destroy.argtypes = [vp]
        L.smpc_get_dims.argtypes = [vp, _ip]
        L.smpc_generate_cycle_horizon.argtypes = [vp, _bp, C.c_int]
        L.smpc_switch_to_walk.argtypes = [vp, _dp]
        L.smpc_switch_to_stand.argtypes = [vp]
        L.smpc_set_velocity_base_batched.argtypes = [vp, _dp]
        L.smpc_set_stage_reference.argtypes = [vp, C.c_int, C.c_int, _dp, C.c_int]
        L.smpc_get_stage_reference.argtypes = [vp, C.c_int, C.c_int, _dp, C.c_int]
        L.smpc_set_reference_pose.argtypes = [vp, C.c_int, C.c_int, _dp]
        L.smpc_get_reference_pose.argtypes = [vp, C.c_int, C.c_int, C.c_int, _dp]
        L.smpc_set_reference_pose_se3.argtypes = [vp, C.c_int, C.c_int, _dp, _dp]
        L.smpc_get_reference_pose_se3.argtypes = [vp, C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.smpc_get_contact_state.argtypes = [vp, C.c_int, _bp]
        L.smpc_get_cycling_contact_state.argtypes = [vp, C.c_int, C.c_void_p]
        L.smpc_id_create.argtypes = [vp, C.POINTER(IdSettingsC), C.c_int, C.c_int, C.POINTER(vp)]
        L.smpc_id_destroy.argtypes = [vp]
        L.smpc_id_destroy.restype = None
        L.smpc_id_set_target.argtypes = [vp, C.c_int, _dp, _dp, _dp, _bp, _dp]
        L.smpc_id_set_targets.argtypes = [vp, _dp, _dp, _dp, _bp, _dp]
        L.smpc_id_set_target_centroidal.argtypes = [vp, C.c_int, _dp, _dp, _dp, _dp, _bp, _dp]
        L.smpc_id_set_targets_centroidal.argtypes = [vp, _dp, _dp, _dp, _dp, _bp, _dp]
        L.smpc_id_solve.argtypes = [vp, _dp, _dp, C.c_void_p, C.c_void_p, C.c_void_p]
        L.smpc_id_debug_get.argtypes = [vp, C.c_int, _dp]
        L.smpc_id_get_resid.argtypes = [vp, _dp]
        L.smpc_id_reset.argtypes = [vp, C.c_int]
        L.smpc_id_solve_device.argtypes = [vp, vp, vp]
        L.smpc_id_wait.argtypes = [vp]
        L.smpc_id_get_tau_device.argtypes = [vp]
        L.smpc_id_get_tau_device.restype = C.c_void_p
        L.smpc_id_get_x_device.argtypes = [vp]
        L.smpc_id_set_targets_from_mpc.argtypes = [vp, vp, C.c_double, C.c_int]
        L.smpc_id_share_stream.argtypes = [vp, vp]
        L.smpc_sim_step_device.argtypes = [vp, vp, vp, _bp, C.c_void_p, C.c_void_p, C.c_double]
        L.smpc_id_get_x_device.restype = C.c_void_p
        L.smpc_get_status.argtypes = [vp, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")]
        L.smpc_debug_get_extra_multipliers.argtypes = [vp, C.c_int, _dp]
        L.smpc_set_x_reference.argtypes = [vp, _dp]
        L.smpc_state_size.argtypes = [vp, C.POINTER(C.c_size_t)]
        L.smpc_save_state.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.smpc_load_state.argtypes = [vp, vp, C.c_size_t]
        L.smpc_iterate.argtypes = [vp, _dp]
        L.smpc_iterate_device.argtypes = [vp, vp]
        L.smpc_wait.argtypes = [vp]
        L.smpc_get_stream.argtypes = [vp]
        L.smpc_get_stream.restype = C.c_void_p
        L.smpc_get_x_device.argtypes = [vp, C.c_int, vp]
        for nm in ("xs", "us", "K0", "Ks", "vs", "lams", "state_derivative01", "reference_poses", "info"):
            getattr(L, "smpc_get_" + nm).argtypes = [vp, _dp]
        L.smpc_get_foot_timing.argtypes = [vp, C.c_int, C.c_int, _ip, C.c_int]
        L.smpc_get_cold_trace.argtypes = [vp, _dp, C.c_int]
        L.smpc_lq_size.argtypes = [vp]
        L.smpc_debug_get_lq.argtypes = [vp, C.c_int, C.c_int, _dp]
        L.smpc_debug_get_steps.argtypes = [vp, _dp, _dp]
        L.smpc_debug_get_terminal.argtypes = [vp, C.c_int, _dp, _dp]
        L.smpc_debug_get_phase_cycles.argtypes = [vp, _dp]
        L.smpc_set_profiling.argtypes = [vp, C.c_int]
        L.smpc_get_kernel_times.argtypes = [vp, _dp, _lp]
        L.smpc_get_kernel_times_n.argtypes = [vp, _dp, _lp, C.c_int]
        L.smpc_kernel_time_slots.argtypes = []
        L.smpc_kernel_time_slots.restype = C.c_int
        L.smpc_reset_kernel_times.argtypes = [vp]
        L.smpc_interpolate.argtypes = [vp, C.c_double, C.c_int, vp, vp, vp]
        L.smpc_friction_compensation.argtypes = [_dp, _dp, C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int, C.c_int]
        L.smpc_centroidal_dynamics.argtypes = [C.c_double, _dp, C.c_double, C.c_int, _dp, _dp, _bp, _dp, C.c_int, _dp, _dp, _dp, C.c_int]
        L.smpc_riccati_feedback.argtypes = [vp, C.c_double, _dp, _dp]
        L.smpc_update_internal_data.argtypes = [vp, _dp, vp, vp, vp, vp]
        L.smpc_full_forward_dynamics.argtypes = [vp, C.c_int, _dp, _dp, vp, vp, vp, C.c_double, C.c_double, C.c_int, _dp, _dp, vp, vp]
        L.smpc_interpolate_knots.argtypes = [C.c_int, C.c_double, C.c_double, _dp, C.c_int, C.c_int, _dp, C.c_int]

    def check(self, code):
        if code < 0:
            msg = self.L.smpc_last_error().decode()
            raise RuntimeError(msg)  # the reference raises std::runtime_error -> Python RuntimeError
        return code


_default = None


def default_lib():
    global _default
    if _default is None:
        _default = SmpcLib()
    return _default
