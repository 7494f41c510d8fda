"""simple_mpc -- drop-in Python surface for the MPC hot path of Simple-Robotics/simple-mpc, backed by
the MI355X HIP engine through the C ABI of include/smpc.h.

Mirrors the reference's Python module (reference bindings/module.cpp:23-40, bindings/expose-mpc.cpp:28-107,
bindings/expose-kinodynamics.cpp:9-129, bindings/expose-robot-handler.cpp:25-60): same class names, method
names and dict keys for the classes on the `MPC.iterate` path.  Differences, all forced by the missing
Pinocchio/Aligator Python modules:
  * `RobotModelHandler` takes a robot table (`load_robot("go2_like")`) instead of a `pinocchio.Model`;
  * `MPC.solver` / `getTrajOptProblem()` (Aligator objects) are not reproduced;
  * `BatchedMPC` is an addition: same verbs, `iterate(X[B, nx])`, outputs `[B, ...]`.
There is no CPU implementation: constructing an MPC without the HIP library or without a GPU raises.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import IdSettingsC, CentroidalSettingsC, FullDynamicsSettingsC, KinodynamicsSettingsC, MpcSettingsC, SmpcLib, default_lib

__all__ = ["load_robot", "RobotModelHandler", "RobotDataHandler", "KinodynamicsOCP", "CentroidalOCP", "FullDynamicsOCP", "MPC", "BatchedMPC", "Interpolator", "FrictionCompensation", "KinodynamicsID", "CentroidalID", "centroidal_dynamics"]


def load_robot(name, lib=None):
    """Built-in robot table (stands in for example_robot_data.load(name).model)."""
    lib = lib or default_lib()
    ptr = lib.L.smpc_builtin_robot(name.encode())
    if not ptr:
        raise RuntimeError("unknown robot %r" % name)
    return ptr


class RobotModelHandler:
    """reference: include/simple-mpc/robot-handler.hpp:28-225, src/robot-handler.cpp:12-96."""

    def __init__(self, model, reference_configuration_name="standing", base_frame_name="root_joint"):
        self._ptr = model
        self._m = model.contents
        if reference_configuration_name not in ("standing", "half_sitting", "straight_standing"):  # the table holds one reference posture
            raise RuntimeError("unknown reference configuration %r" % reference_configuration_name)
        self._base = base_frame_name
        self._feet = []

    def _table_foot(self, name):
        for f in range(self._m.nfeet):
            if self._m.foot_name[f].value.decode() == name:
                return f
        raise RuntimeError("frame %r is not a foot of robot %r" % (name, self._m.name.decode()))

    def addPointFoot(self, foot_name, reference_parent_frame_name):
        f = self._table_foot(foot_name)
        if f != len(self._feet):
            raise RuntimeError("feet must be added in the order of the robot table")
        self._feet.append(foot_name)
        return f

    def addQuadFoot(self, foot_name, reference_parent_frame_name, contact_points):
        """6-D foot (reference src/robot-handler.cpp:62-79).  The four sole corners only matter to the whole-body ID controllers;
        the MPC problems use the sole half-sizes of their settings (Lfoot, Wfoot)."""
        f = self._table_foot(foot_name)
        if f != len(self._feet):
            raise RuntimeError("feet must be added in the order of the robot table")
        q = np.asarray(contact_points, float)
        if q.shape != (4, 3):
            raise RuntimeError("contact_points must be a 4 x 3 matrix")
        self._feet.append(foot_name)
        self._quads = getattr(self, "_quads", {})
        self._quads[foot_name] = q
        return f

    def getFeetNb(self):
        return len(self._feet)

    def getFootFrameName(self, i):
        return self._feet[i]

    def getFeetFrameNames(self):
        return list(self._feet)

    def getFootNb(self, name):
        return self._feet.index(name)

    def getMass(self):
        return float(self._m.total_mass)

    # reference src/robot-handler.cpp:81-95
    def difference(self, x1, x2):
        from . import _hostmath as hm

        return hm.difference(self.nq, self.nv, x1, x2)

    def getBaseFrameName(self):
        return self._base

    # frame ids: the table has no frame list; the ids are stable integers with the meaning the reference gives them (0 = base frame,
    # 1 + 2 i / 2 + 2 i = foot i / its reference frame), good for comparisons and dictionary keys
    def getBaseFrameId(self):
        return 0

    def getFootFrameId(self, i):
        return 1 + 2 * int(i)

    def getFootRefFrameId(self, i):
        return 2 + 2 * int(i)

    def getFeetFrameIds(self):
        return [self.getFootFrameId(i) for i in range(len(self._feet))]

    def setFootReferencePlacement(self, foot_nb, parentframeMfootref):
        """reference src/robot-handler.cpp:76-79: the placement of the foot's reference frame in the base frame (the Raibert heuristic of
        MPC::updateStepTrackerReferences starts from it).  Takes an SE3-like object (.translation) or 3 numbers; the handler then owns a
        private copy of the robot table."""
        if not getattr(self, "_owned", None):
            self._owned = _capi.RobotModelC.from_buffer_copy(self._m)
            self._m = self._owned
            self._ptr = C.pointer(self._owned)
        t = np.asarray(getattr(parentframeMfootref, "translation", parentframeMfootref), float).reshape(3)
        for k in range(3):
            self._m.foot_ref_p[int(foot_nb)][k] = float(t[k])

    def getReferenceState(self):
        return np.concatenate([np.array(self._m.q_ref[: self._m.nq]), np.zeros(self._m.nv)])

    def getModel(self):
        return self

    # minimal pinocchio.Model-like attributes used by the reference's example scripts
    @property
    def nq(self):
        return int(self._m.nq)

    @property
    def nv(self):
        return int(self._m.nv)

    @property
    def lowerPositionLimit(self):
        return np.concatenate([np.full(7, -np.inf), np.array(self._m.q_lo[: self._m.nv - 6])])

    @property
    def upperPositionLimit(self):
        return np.concatenate([np.full(7, np.inf), np.array(self._m.q_hi[: self._m.nv - 6])])


class RobotDataHandler:
    """Host-side data of ONE state (reference include/simple-mpc/robot-handler.hpp:152-225, src/robot-handler.cpp:97-149): frame
    placements, centre of mass, centroidal momentum, computed with NumPy from the robot table.  The batch's counterpart runs on the device
    (BatchedMPC.updateInternalData / inside iterate); this class serves the scripts written for the reference, which read foot poses and
    the centroidal state of the measured state between control steps."""

    def __init__(self, model_handler):
        self.model_handler = model_handler
        self.updateInternalData(model_handler.getReferenceState(), True)

    def updateInternalData(self, x, updateJacobians=False):
        from . import _hostmath as hm

        self._x = np.array(x, float).copy()
        self._k = hm.kinematics(self.model_handler._m, self._x)

    def updateJacobiansMassMatrix(self, x):
        raise RuntimeError("joint Jacobians / mass matrix are not kept on the host; the kernels form them per stage on the device")

    def getData(self):
        raise RuntimeError("there is no pinocchio.Data behind this handler; use getFootPose / getBaseFramePose / getCentroidalState")

    def getModelHandler(self):
        return self.model_handler

    def getState(self):
        return self._x.copy()

    def getBaseFramePose(self):
        from . import _hostmath as hm

        return hm.SE3(self._k["R"][0], self._k["p"][0])

    def getFootPose(self, i):
        from . import _hostmath as hm

        m = self.model_handler._m
        f = self.model_handler._table_foot(self.model_handler.getFootFrameName(i))
        j = m.foot_joint[f]
        return hm.SE3(self._k["R"][j], self._k["p"][j] + self._k["R"][j] @ np.array(m.foot_p[f]))

    def getFootRefPose(self, i):
        from . import _hostmath as hm

        m = self.model_handler._m
        f = self.model_handler._table_foot(self.model_handler.getFootFrameName(i))
        return hm.SE3(self._k["R"][0], self._k["p"][0] + self._k["R"][0] @ np.array(m.foot_ref_p[f]))

    def getCentroidalState(self):
        return np.concatenate([self._k["com"], self._k["hg"]])




class _StageReferences:
    """Per-stage setters / getters of OCPHandler (reference include/simple-mpc/ocp-handler.hpp:66-127), broadcast over the
    batch.  The batched problem lives in the MPC handle: they work once BatchedMPC / MPC has been constructed on this OCP.
    Poses are translations (3-vectors; an object with a `.translation` attribute is accepted); getters return instance 0."""

    _mpc = None

    def _handle(self):
        if self._problem is None:
            raise RuntimeError("Create problem first!")
        if self._mpc is None:
            raise RuntimeError("the batched problem lives in the MPC handle: construct BatchedMPC(settings, ocp, batch) first")
        return self._mpc

    def _foot(self, ee_name):
        names = self.model_handler.getFeetFrameNames()
        if ee_name not in names:
            raise RuntimeError("unknown end effector %r" % ee_name)
        return names.index(ee_name)

    def _set(self, t, what, v):
        m = self._handle()
        v = np.ascontiguousarray(v, float)
        m._lib.check(m._lib.L.smpc_set_stage_reference(m._h, int(t), what, v, v.size))

    def _get(self, t, what, n):
        m = self._handle()
        out = np.zeros(n)
        m._lib.check(m._lib.L.smpc_get_stage_reference(m._h, int(t), what, out, n))
        return out

    # control target and its force segments (reference src/ocp-handler.cpp:58-70, src/kinodynamics.cpp:229-265)
    def setReferenceControl(self, t, u_ref):
        u = np.ascontiguousarray(u_ref, float)
        if u.shape != (self.nu,):
            raise RuntimeError("u_ref not of the right size")
        self._set(t, 0, u)

    def getReferenceControl(self, t):
        return self._get(t, 0, self.nu)

    def _control_from_forces(self, force_refs):
        fs = int(self.settings["force_size"])
        for i, name in enumerate(self.model_handler.getFeetFrameNames()):
            f = np.asarray(force_refs[name], float)
            if f.size != fs:
                raise RuntimeError("force size in settings does not match reference force size")
            self._control_ref[i * fs : (i + 1) * fs] = f

    def setReferenceForces(self, t, force_refs):
        self._control_from_forces(force_refs)
        self.setReferenceControl(t, self._control_ref)

    def setReferenceForce(self, t, ee_name, force_ref):
        # like the reference, the other segments come from the handler's control_ref_ member, not from stage t
        fs, i = int(self.settings["force_size"]), self._foot(ee_name)
        self._control_ref[i * fs : (i + 1) * fs] = np.asarray(force_ref, float)
        self.setReferenceControl(t, self._control_ref)

    def getReferenceForce(self, t, ee_name):
        fs, i = int(self.settings["force_size"]), self._foot(ee_name)
        return self.getReferenceControl(t)[i * fs : (i + 1) * fs]

    # foot references (reference src/kinodynamics.cpp:154-228, src/centroidal-dynamics.cpp:120-188)
    def setReferencePose(self, t, ee_name, pose_ref):
        """pose_ref: an SE3-like object (`.translation`, `.rotation`), a 4 x 4 homogeneous matrix or a translation.  The rotation is kept and
        returned by getReferencePose (reference tests/problem.cpp:157-160); like the reference's, it lives until the next MPC.iterate, which
        rewrites every stage's pose with the identity rotation before it solves (src/mpc.cpp:303-309)."""
        m = self._handle()
        p, R = _pose_parts(pose_ref)
        m._lib.check(m._lib.L.smpc_set_reference_pose_se3(m._h, int(t), self._foot(ee_name), p, R))

    def setReferencePoses(self, t, pose_refs):
        if len(pose_refs) != self.model_handler.getFeetNb():
            raise RuntimeError("pose_refs size does not match number of end effectors")
        for name in self.model_handler.getFeetFrameNames():
            self.setReferencePose(t, name, pose_refs[name])

    def getReferencePose(self, t, ee_name):
        """The reference placement of the foot at stage t: a `Pose` -- a 3-vector (the translation, as before) that also carries `.translation`,
        `.rotation` and `.homogeneous` (the SE3 of the reference)."""
        m = self._handle()
        p, R = np.zeros(3), np.zeros(9)
        m._lib.check(m._lib.L.smpc_get_reference_pose_se3(m._h, int(t), self._foot(ee_name), 0, p, R))
        return Pose(p, R.reshape(3, 3))

    # state target and its segments
    def setReferenceState(self, t, x_ref):
        x = np.ascontiguousarray(x_ref, float)
        if x.shape != (self._nx_ref,):
            raise RuntimeError("x_ref not of the right size")
        self._set(t, 1, x)

    def getReferenceState(self, t):
        return self._get(t, 1, self._nx_ref)

    def getContactState(self, t):
        m = self._handle()
        out = np.zeros(self.model_handler.getFeetNb(), np.uint8)
        m._lib.check(m._lib.L.smpc_get_contact_state(m._h, int(t), out))
        return [bool(v) for v in out]

    def getContactSupport(self, t):
        return int(sum(self.getContactState(t)))

_KINO_KEYS = [
    "timestep", "w_x", "w_u", "w_frame", "w_cent", "w_centder", "qmin", "qmax", "gravity", "mu", "Lfoot", "Wfoot",
    "force_size", "kinematics_limits", "force_cone", "land_cstr",
]


class KinodynamicsOCP(_StageReferences):
    """reference: src/kinodynamics.cpp:29-38 (ctor), bindings/expose-kinodynamics.cpp:9-34 (dict keys)."""

    def __init__(self, settings, model_handler):
        for k in _KINO_KEYS:
            if k not in settings:
                raise KeyError(k)  # boost.python raises KeyError on a missing key
        self.settings = dict(settings)
        self.model_handler = model_handler
        self._problem = None
        nv = model_handler.nv
        self.nu = nv - 6 + int(settings["force_size"]) * model_handler.getFeetNb()

    def getSettings(self):
        return dict(self.settings)

    def getModelHandler(self):
        return self.model_handler

    def getNu(self):
        return self.nu

    def createProblem(self, x0, horizon, force_size, gravity, terminal_constraint=False):
        """reference src/ocp-handler.cpp:96-137"""
        if force_size != self.settings["force_size"]:
            raise RuntimeError("force size in settings does not match reference force size")
        self._problem = dict(x0=np.array(x0, float), horizon=int(horizon), gravity=float(gravity),
                             terminal_constraint=bool(terminal_constraint))

    def getSize(self):
        if self._problem is None:
            raise RuntimeError("Create problem first!")
        return self._problem["horizon"]

    # reference src/kinodynamics.cpp:308-311, src/fulldynamics.cpp:365-368
    def getProblemState(self, data_handler):
        return data_handler.getState()

    # reference src/kinodynamics.cpp:366-388: createProblem(..., terminal_constraint = True) calls createTerminalConstraint(x0.head<3>());
    # called by hand it adds the same DCM constraint.  Its reference is the base position of x0 until the first control step, after which
    # MPC::updateStepTrackerReferences overwrites it every step (src/mpc.cpp:313-323) -- which is also all updateTerminalConstraint does.
    def createTerminalConstraint(self, com_ref):
        if self._problem is None:
            raise RuntimeError("Create problem first!")
        if not np.allclose(np.asarray(com_ref, float).reshape(3), self._problem["x0"][:3]):
            raise RuntimeError("the terminal constraint is created with the base position of x0 as its reference (createProblem's choice)")
        self._problem["terminal_constraint"] = True

    def updateTerminalConstraint(self, com_ref):
        if self._problem is None:
            raise RuntimeError("Create problem first!")
        self._problem["terminal_com_ref"] = np.asarray(com_ref, float).reshape(3).copy()  # (every iterate() recomputes it on the device)

    def _default_x_reference(self):
        return self.model_handler.getReferenceState()

    @property
    def _nx_ref(self):
        return self.model_handler.nq + self.model_handler.nv

    def getCostNumber(self):
        return 4 + self.model_handler.getFeetNb()  # state, control, centroidal, centroidal derivative, one pose cost per foot

    # reference src/kinodynamics.cpp:267-306
    def setVelocityBase(self, t, velocity_base):
        v = np.asarray(velocity_base, float)
        if v.size != 6:
            raise RuntimeError("velocity_base size should be 6")
        x = self.getReferenceState(t)
        x[self.model_handler.nq : self.model_handler.nq + 6] = v
        self.setReferenceState(t, x)

    def getVelocityBase(self, t):
        return self.getReferenceState(t)[self.model_handler.nq : self.model_handler.nq + 6]

    def setPoseBase(self, t, pose_base):
        p = np.asarray(pose_base, float)
        if p.size != 7:
            raise RuntimeError("pose_base size should be 7")
        x = self.getReferenceState(t)
        x[:7] = p
        self.setReferenceState(t, x)

    def getPoseBase(self, t):
        return self.getReferenceState(t)[:7]

    def _xdot_size(self):
        return 2 * self.model_handler.nv

    def _kernel_names(self):
        return ["recede", "deriv", "riccati", "forward", "trial", "select", "apply", "tree", "tree_ls"]

    def _create_handle(self, lib, ms, batch, device_id):
        ocp = self
        s = ocp.settings
        c = lambda a: np.ascontiguousarray(np.asarray(a, float))
        self._keep = [c(s[k]) for k in ("w_x", "w_u", "w_frame", "w_cent", "w_centder", "qmin", "qmax")]
        mh = ocp.model_handler
        ndx, nu = 2 * mh.nv, ocp.nu
        fs = int(s["force_size"])  # w_frame: 3 x 3 on the translation of a point foot, 6 x 6 on the placement of a flat foot
        shapes = [(ndx, ndx), (nu, nu), (fs, fs), (6, 6), (6, 6), (mh.nv - 6,), (mh.nv - 6,)]
        for a, sh, k in zip(self._keep, shapes, ("w_x", "w_u", "w_frame", "w_cent", "w_centder", "qmin", "qmax")):
            if a.shape != sh:
                raise RuntimeError("%s has shape %s, expected %s" % (k, a.shape, sh))
        ks = KinodynamicsSettingsC()
        ks.timestep = s["timestep"]
        for name, arr in zip(("w_x", "w_u", "w_frame", "w_cent", "w_centder", "qmin", "qmax"), self._keep):
            setattr(ks, name, arr.ctypes.data)
        for i in range(3):
            ks.gravity[i] = float(s["gravity"][i])
        ks.mu, ks.Lfoot, ks.Wfoot = s["mu"], s["Lfoot"], s["Wfoot"]
        ks.force_size = int(s["force_size"])
        ks.kinematics_limits = int(bool(s["kinematics_limits"]))
        ks.force_cone = int(bool(s["force_cone"]))
        ks.land_cstr = int(bool(s["land_cstr"]))
        ks.terminal_constraint = int(ocp._problem.get("terminal_constraint", False))
        h = C.c_void_p()
        lib.check(lib.L.smpc_create(mh._ptr, C.byref(ks), C.byref(ms), batch, ocp._problem["gravity"], device_id, C.byref(h)))
        return h


_CENT_KEYS = [
    "timestep", "w_u", "w_com", "w_linear_mom", "w_angular_mom", "w_linear_acc", "w_angular_acc", "gravity", "mu", "Lfoot",
    "Wfoot", "force_size",
]


class CentroidalOCP(_StageReferences):
    """reference: src/centroidal-dynamics.cpp:27-37 (ctor), bindings/expose-centroidal.cpp (dict keys =
    CentroidalSettings fields, include/simple-mpc/centroidal-dynamics.hpp:27-43).  State [com; h_lin; h_ang], control = the
    stacked 3-D contact forces."""

    def __init__(self, settings, model_handler):
        for k in _CENT_KEYS:
            if k not in settings:
                raise KeyError(k)
        self.settings = dict(settings)
        self.model_handler = model_handler
        self._problem = None
        self.nu = int(settings["force_size"]) * model_handler.getFeetNb()

    def getSettings(self):
        return dict(self.settings)

    def getModelHandler(self):
        return self.model_handler

    def getNu(self):
        return self.nu

    def createProblem(self, x0, horizon, force_size, gravity, terminal_constraint=False):
        """reference src/ocp-handler.cpp:96-137; the terminal constraint of this OCP is disabled upstream
        (src/centroidal-dynamics.cpp:318-328), so the flag is accepted and has no effect, like there."""
        if force_size != self.settings["force_size"]:
            raise RuntimeError("force size in settings does not match reference force size")
        self._problem = dict(x0=np.array(x0, float), horizon=int(horizon), gravity=float(gravity))

    def getSize(self):
        if self._problem is None:
            raise RuntimeError("Create problem first!")
        return self._problem["horizon"]

    _nx_ref = 9

    # reference src/centroidal-dynamics.cpp:259-262
    def getProblemState(self, data_handler):
        return data_handler.getCentroidalState()

    # reference src/centroidal-dynamics.cpp:318-335: the constraint is left out upstream; both calls succeed and change nothing
    def createTerminalConstraint(self, com_ref):
        if self._problem is None:
            raise RuntimeError("Create problem first!")

    def updateTerminalConstraint(self, com_ref):
        pass

    def getCostNumber(self):
        return 6  # com, control, linear / angular momentum, linear / angular acceleration (tests/problem.cpp:232)

    # reference src/centroidal-dynamics.cpp:212-257: velocities are stored as momenta m v, the CoM reference is 3-D
    def setVelocityBase(self, t, velocity_base):
        v = np.asarray(velocity_base, float)
        if v.size != 6:
            raise RuntimeError("velocity_base not of the right size")
        x = self.getReferenceState(t)
        x[3:] = v
        self.setReferenceState(t, x)

    def getVelocityBase(self, t):
        return self.getReferenceState(t)[3:]

    def setPoseBase(self, t, pose_base):
        p = np.asarray(pose_base, float)
        if p.size != 3:
            raise RuntimeError("pose_base not of the right size")
        x = self.getReferenceState(t)
        x[:3] = p
        self.setReferenceState(t, x)

    def getPoseBase(self, t):
        return self.getReferenceState(t)[:3]

    def _default_x_reference(self):
        return np.zeros(9)  # getReferenceState(0) of the default problem (src/centroidal-dynamics.cpp:293-298, com_ref_ = 0)

    def _xdot_size(self):
        return 9

    def _kernel_names(self):
        if int(self.settings.get("force_size", 3)) == 6:  # (6-D feet: smpc_cent6_kernels.h)
            return ["frontend", "recede", "deriv", "riccati", "forward", "line_search", "trial", "-7", "-8"]
        # point feet (smpc_cent_split.h): recede | pre-pass | backward | forward | line search; with SMPC_CENT_FUSED=1 "step" is the one-kernel
        # control step and the others are never launched
        return ["frontend", "step", "pre", "backward", "forward", "line_search", "-6", "-7", "-8"]

    def _create_handle(self, lib, ms, batch, device_id):
        s = self.settings
        c = lambda a: np.ascontiguousarray(np.asarray(a, float))
        names = ("w_u", "w_com", "w_linear_mom", "w_angular_mom", "w_linear_acc", "w_angular_acc")
        self._keep = [c(s[k]) for k in names]
        shapes = [(self.nu, self.nu)] + [(3, 3)] * 5
        for a, sh, k in zip(self._keep, shapes, names):
            if a.shape != sh:
                raise RuntimeError("%s has shape %s, expected %s" % (k, a.shape, sh))
        cs = CentroidalSettingsC()
        cs.timestep = s["timestep"]
        for name, arr in zip(names, self._keep):
            setattr(cs, name, arr.ctypes.data)
        for i in range(3):
            cs.gravity[i] = float(s["gravity"][i])
        cs.mu, cs.Lfoot, cs.Wfoot = s["mu"], s["Lfoot"], s["Wfoot"]
        cs.force_size = int(s["force_size"])
        h = C.c_void_p()
        lib.check(lib.L.smpc_create_centroidal(
            self.model_handler._ptr, C.byref(cs), C.byref(ms), batch, self._problem["gravity"], device_id, C.byref(h)))
        return h


_FULL_KEYS = [
    "timestep", "w_x", "w_u", "w_cent", "gravity", "force_size", "w_forces", "w_frame", "umin", "umax", "qmin", "qmax",
    "Kp_correction", "Kd_correction", "mu", "Lfoot", "Wfoot", "torque_limits", "kinematics_limits", "force_cone", "land_cstr",
]


class FullDynamicsOCP(KinodynamicsOCP):
    """reference: src/fulldynamics.cpp:30-76 (ctor), bindings/expose-fulldynamics.cpp:12-42 (dict keys = FullDynamicsSettings
    fields, include/simple-mpc/fulldynamics.hpp:28-65).  State (q, v), control = the nv - 6 joint torques; the contact forces
    are outputs of the constrained dynamics (MPC.getContactForces), their references live beside the control reference."""

    def __init__(self, settings, model_handler):
        for k in _FULL_KEYS:
            if k not in settings:
                raise KeyError(k)
        self.settings = dict(settings)
        self.model_handler = model_handler
        self._problem = None
        self.nu = model_handler.nv - 6
        fs = int(settings["force_size"])
        if np.asarray(settings["Kp_correction"]).size != fs:
            raise RuntimeError("Force must be of same size as Kp correction")  # src/fulldynamics.cpp:41-44
        if np.asarray(settings["Kd_correction"]).size != fs:
            raise RuntimeError("Force must be of same size as Kd correction")

    def createProblem(self, x0, horizon, force_size, gravity, terminal_constraint=False):
        if force_size != self.settings["force_size"]:
            raise RuntimeError("force size in settings does not match reference force size")
        self._problem = dict(x0=np.array(x0, float), horizon=int(horizon), gravity=float(gravity),
                             terminal_constraint=bool(terminal_constraint))

    def getCostNumber(self):
        # state, control, centroidal + one pose cost per foot + one force cost per foot in contact (tests/problem.cpp:48)
        return 3 + 2 * self.model_handler.getFeetNb()

    # force references: their own vector beside the control reference (src/fulldynamics.cpp:258-334)
    def _force_refs(self, t):
        n = int(self.settings["force_size"]) * self.model_handler.getFeetNb()
        return self._get(t, 2, n)

    def setReferenceForces(self, t, force_refs):
        fs = int(self.settings["force_size"])
        names = self.model_handler.getFeetFrameNames()
        if len(force_refs) != len(names):
            raise RuntimeError("force_refs size does not match number of end effectors")
        v = self._force_refs(t)
        for i, name in enumerate(names):
            f = np.asarray(force_refs[name], float)
            if f.size != fs:
                raise RuntimeError("Reference forces do not have the right dimension")
            v[i * fs : (i + 1) * fs] = f
        self._set(t, 2, v)

    def setReferenceForce(self, t, ee_name, force_ref):
        fs, i = int(self.settings["force_size"]), self._foot(ee_name)
        f = np.asarray(force_ref, float)
        if f.size != fs:
            raise RuntimeError("Reference forces do not have the right dimension")
        v = self._force_refs(t)
        v[i * fs : (i + 1) * fs] = f
        self._set(t, 2, v)

    def getReferenceForce(self, t, ee_name):
        fs, i = int(self.settings["force_size"]), self._foot(ee_name)
        return self._force_refs(t)[i * fs : (i + 1) * fs]

    def _create_handle(self, lib, ms, batch, device_id):
        s = self.settings
        c = lambda a: np.ascontiguousarray(np.asarray(a, float))
        mh = self.model_handler
        ndx, nu, fs = 2 * mh.nv, self.nu, int(s["force_size"])
        names = ("w_x", "w_u", "w_cent", "w_forces", "w_frame", "umin", "umax", "qmin", "qmax", "Kp_correction", "Kd_correction")
        shapes = [(ndx, ndx), (nu, nu), (6, 6), (fs, fs), (fs, fs), (nu,), (nu,), (nu,), (nu,), (fs,), (fs,)]
        self._keep = [c(s[k]) for k in names]
        for a, sh, k in zip(self._keep, shapes, names):
            if a.shape != sh:
                raise RuntimeError("%s has shape %s, expected %s" % (k, a.shape, sh))
        fsx = FullDynamicsSettingsC()
        fsx.timestep = s["timestep"]
        for name, arr in zip(names, self._keep):
            setattr(fsx, name, arr.ctypes.data)
        for i in range(3):
            fsx.gravity[i] = float(s["gravity"][i])
        fsx.mu, fsx.Lfoot, fsx.Wfoot = s["mu"], s["Lfoot"], s["Wfoot"]
        fsx.force_size = fs
        for k in ("torque_limits", "kinematics_limits", "force_cone", "land_cstr"):
            setattr(fsx, k, int(bool(s[k])))
        fsx.terminal_constraint = int(self._problem.get("terminal_constraint", False))
        h = C.c_void_p()
        lib.check(lib.L.smpc_create_fulldynamics(mh._ptr, C.byref(fsx), C.byref(ms), batch, self._problem["gravity"], device_id, C.byref(h)))
        return h


_MPC_KEYS = ["support_force", "TOL", "mu_init", "max_iters", "num_threads", "swing_apex", "T_fly", "T_contact", "timestep"]


class BatchedMPC:
    """B phase-aligned instances of the reference's MPC (include/simple-mpc/mpc.hpp:55-197) solved together."""

    def __init__(self, settings, ocp, batch, device_id=0, lib=None):
        for k in _MPC_KEYS:
            if k not in settings:
                raise KeyError(k)
        if ocp._problem is None:
            raise RuntimeError("Create problem first!")
        self._lib = lib or default_lib()
        L = self._lib.L
        self.settings = dict(settings)
        self.ocp_handler = ocp
        ms = MpcSettingsC()
        ms.swing_apex = settings["swing_apex"]
        ms.support_force = settings["support_force"]
        ms.TOL = settings["TOL"]
        ms.mu_init = settings["mu_init"]
        ms.max_iters = int(settings["max_iters"])
        ms.num_threads = int(settings["num_threads"])
        ms.T_fly = int(settings["T_fly"])
        ms.T_contact = int(settings["T_contact"])
        ms.T = int(ocp._problem["horizon"])
        ms.timestep = settings["timestep"]
        mh = ocp.model_handler
        self._h = ocp._create_handle(self._lib, ms, int(batch), int(device_id))
        d = np.zeros(8, np.int32)
        L.smpc_get_dims(self._h, d)
        self.nq, self.nv, self.nx, self.ndx, self.nu, self.nc, self.nf, self.H = (int(v) for v in d)
        self.B = int(batch)
        ocp._mpc = self
        # OCPHandler::control_ref_ as the constructors leave it (createProblem's default forces, src/ocp-handler.cpp:107-109)
        fs = int(ocp.settings["force_size"])
        ocp._control_ref = np.zeros(ocp.nu)
        if not isinstance(ocp, FullDynamicsOCP):
            ocp._control_ref[2 : fs * mh.getFeetNb() : fs] = -mh.getMass() * ocp._problem["gravity"] / mh.getFeetNb()
        self.nx_in = self.nq + self.nv  # iterate takes measured multibody states (reference src/mpc.cpp:189-192)
        self._x_reference = ocp._default_x_reference()
        self._velocity_base = np.zeros(6)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.L.smpc_destroy(h)
            self._h = None

    # ---- reference verbs ----
    def generateCycleHorizon(self, contact_states):
        """contact_states: list of {foot_name: bool} (reference bindings/expose-mpc.cpp:78) or uint8 array [n][nfeet]."""
        if len(contact_states) and isinstance(contact_states[0], dict):
            names = self.ocp_handler.model_handler.getFeetFrameNames()
            cs = np.array([[1 if st[n] else 0 for n in names] for st in contact_states], np.uint8)
        else:
            cs = np.ascontiguousarray(contact_states, np.uint8)
        self._lib.check(self._lib.L.smpc_generate_cycle_horizon(self._h, cs, cs.shape[0]))
        # control_ref_ ends as the force distribution of the last cycle stage created (src/mpc.cpp:144-163)
        ocp, last = self.ocp_handler, cs[-1]
        fs = int(ocp.settings["force_size"])
        ocp._control_ref[:] = 0.0
        if not isinstance(ocp, FullDynamicsOCP):
            for f in range(len(last)):
                if last[f]:
                    ocp._control_ref[f * fs + 2] = self.settings["support_force"] / max(1, int(last.sum()))

    # MPC::setReferencePose / getReferencePose (reference src/mpc.cpp:326-339)
    def setReferencePose(self, t, ee_name, pose_ref):
        self.ocp_handler.setReferencePose(t, ee_name, pose_ref)

    def getReferencePose(self, t, ee_name):
        return self.ocp_handler.getReferencePose(t, ee_name)

    def switchToWalk(self, velocity_base):
        v = np.ascontiguousarray(velocity_base, float)
        if v.shape != (6,):
            raise RuntimeError("velocity_base size should be 6")
        self._velocity_base = v.copy()
        self._lib.check(self._lib.L.smpc_switch_to_walk(self._h, v))

    def switchToStand(self):
        self._velocity_base = np.zeros(6)
        self._lib.check(self._lib.L.smpc_switch_to_stand(self._h))

    @property
    def velocity_base(self):
        return self._velocity_base

    @velocity_base.setter
    def velocity_base(self, v):
        # reference: public member MPC::velocity_base_ (bindings/expose-mpc.cpp:86); walking state is unchanged
        v = np.ascontiguousarray(v, float)
        self._velocity_base = v.copy()
        self._lib.check(self._lib.L.smpc_switch_to_walk(self._h, v))

    def setVelocityBaseBatched(self, V):
        """One velocity command per instance, V[B, 6] (the reference has one `velocity_base` per MPC object)."""
        V = np.ascontiguousarray(V, float)
        if V.shape != (self.B, 6):
            raise RuntimeError("velocity_base size should be (batch, 6)")
        self._velocity_base = V.copy()
        self._lib.check(self._lib.L.smpc_set_velocity_base_batched(self._h, V))

    @property
    def x_reference(self):
        return self._x_reference

    @x_reference.setter
    def x_reference(self, x):
        x = np.ascontiguousarray(x, float)
        if x.shape != (self.nx,):
            raise RuntimeError("x_ref not of the right size")
        self._x_reference = x.copy()
        self._lib.check(self._lib.L.smpc_set_x_reference(self._h, x))

    def iterate(self, X):
        X = np.ascontiguousarray(X, float)
        if X.shape != (self.B, self.nx_in):
            raise RuntimeError("X must have shape (batch, nq+nv)")
        self._lib.check(self._lib.L.smpc_iterate(self._h, X))
        self._last_X = X  # (getDataHandler)

    def iterate_device(self, device_ptr):
        self._lib.check(self._lib.L.smpc_iterate_device(self._h, C.c_void_p(int(device_ptr))))

    def wait(self):
        self._lib.check(self._lib.L.smpc_wait(self._h))
        self._out_async = None

    def stream(self):
        """The handle's hipStream_t as an integer (0 in the CPU test build): torch.cuda.ExternalStream(mpc.stream()) puts a caller's device work
        in the same in-order queue as the control steps."""
        return int(self._lib.L.smpc_get_stream(self._h) or 0)

    def iterateAsync(self, X):
        """iterate() without the final synchronisation and without fetching the solution: X (kept alive until wait()) -> launches only.
        With gatherOutputs / wait, one host thread drives several handles, one per device."""
        self._x_async = np.ascontiguousarray(X, dtype=np.float64)
        if self._x_async.shape != (self.B, self.nx_in):
            raise RuntimeError("X must be [batch, nq + nv]")
        self._lib.check(self._lib.L.smpc_iterate_async(self._h, self._x_async))

    def gatherOutputs(self, out, first_row=0):
        """Rows [x1 | u0 | K0] of this handle's instances into rows first_row .. first_row + batch of `out` (a C-contiguous float64 array
        [n, nx + nu + nu ndx] -- one buffer for the handles of all devices; pinned memory for full PCIe rate).  Asynchronous: wait()."""
        row = self.nx + self.nu + self.nu * self.ndx
        if out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"] or out.ndim != 2 or out.shape[1] < row or out.shape[0] < first_row + self.B:
            raise RuntimeError("out must be a C-contiguous float64 array [>= first_row + batch, >= nx + nu + nu * ndx]")
        assert out.strides == (out.shape[1] * 8, 8)
        self._out_async = out  # the copies land after this call returns: the buffer lives until the next gather / wait()
        self._lib.check(self._lib.L.smpc_gather_outputs(self._h, C.c_void_p(out.ctypes.data + first_row * out.shape[1] * 8), out.shape[1]))

    def gather_outputs_device(self, device_ptr, row_doubles=None):
        """The same rows packed into a device buffer [batch][row_doubles] (one kernel on the handle's stream)."""
        row = self.nx + self.nu + self.nu * self.ndx
        self._lib.check(self._lib.L.smpc_gather_outputs_device(self._h, C.c_void_p(int(device_ptr)), int(row_doubles or row)))

    def gather_outputs_peer(self, device_ptr, dst_device):
        """The packed rows [batch][nx + nu + nu ndx] into a buffer on another device of the node (peer copy on the handle's stream)."""
        self._lib.check(self._lib.L.smpc_gather_outputs_peer(self._h, C.c_void_p(int(device_ptr)), int(dst_device)))

    def setEarlyExitOnTol(self, on=True):
        """SolverProxDDP's convergence test inside iterate (reference src/mpc.cpp:43,212): an instance converged to settings TOL at the
        start of an iteration takes no further step in that control step.  Off by default (the metric is at fixed iterations)."""
        self._lib.check(self._lib.L.smpc_set_early_exit_on_tol(self._h, int(bool(on))))

    def get_x_device(self, t, device_ptr):
        """xs[t] of every instance into a device buffer [B][nx] (asynchronous on the engine's stream)."""
        self._lib.check(self._lib.L.smpc_get_x_device(self._h, int(t), C.c_void_p(int(device_ptr))))

    def save_state(self):
        """Checkpoint of the whole batch (bytes): everything a later iterate depends on."""
        n = C.c_size_t()
        self._lib.check(self._lib.L.smpc_state_size(self._h, C.byref(n)))
        buf = np.zeros(n.value, np.uint8)
        w = C.c_size_t()
        self._lib.check(self._lib.L.smpc_save_state(self._h, buf.ctypes.data_as(C.c_void_p), n.value, C.byref(w)))
        return buf[: w.value].tobytes()

    def load_state(self, blob):
        """Resume from save_state() of a handle of the same kind, batch, horizon and robot."""
        buf = np.frombuffer(blob, np.uint8).copy()
        self._lib.check(self._lib.L.smpc_load_state(self._h, buf.ctypes.data_as(C.c_void_p), buf.size))

    def _get(self, fn, shape):
        out = np.zeros(shape)
        self._lib.check(getattr(self._lib.L, fn)(self._h, out))
        return out

    @property
    def xs(self):
        return self._get("smpc_get_xs", (self.B, self.H + 1, self.nx))

    @property
    def us(self):
        return self._get("smpc_get_us", (self.B, self.H, self.nu))

    @property
    def K0(self):
        return self._get("smpc_get_K0", (self.B, self.nu, self.ndx))

    @property
    def Ks(self):
        return self._get("smpc_get_Ks", (self.B, self.H, self.nu, self.ndx))

    @property
    def vs(self):
        v = self._get("smpc_get_vs", (self.B, self.H, self.nc))
        st = getattr(self.ocp_handler, "settings", {})
        if isinstance(self.ocp_handler, KinodynamicsOCP) and not isinstance(self.ocp_handler, FullDynamicsOCP) and int(st.get("force_size", 3)) == 6:
            # 6-D feet: the device keeps [control box (absent) | joint box | wrench-cone rows | frame-velocity rows] (smpc_full_model.h); returned as
            # joint box | 6 velocity rows per foot (all feet) | (force_cone) 17 cone rows per foot (all feet) -- the order of THIS repository's
            # oracle.  NOT the reference's constraint stack: src/kinodynamics.cpp:105-123 adds, foot by foot, the wrench cone and then the frame
            # velocity (interleaved); a consumer comparing with Aligator's stack has to permute the per-foot blocks.
            nf = self.ocp_handler.model_handler.getFeetNb()
            na = self.nu - 6 * nf
            parts = [v[:, :, self.nu : self.nu + na], v[:, :, self.nu + na + 17 * nf :]]
            if st.get("force_cone", False):
                parts.append(v[:, :, self.nu + na : self.nu + na + 17 * nf])
            return np.concatenate(parts, axis=2)
        if isinstance(self.ocp_handler, KinodynamicsOCP) and not isinstance(self.ocp_handler, FullDynamicsOCP):
            nf = self.ocp_handler.model_handler.getFeetNb()
            for which, key, n in ((0, "force_cone", 2 * nf), (1, "land_cstr", nf)):  # optional rows, in the oracle's order
                if st.get(key, False):
                    e = np.zeros((self.B, self.H, n))
                    self._lib.check(self._lib.L.smpc_debug_get_extra_multipliers(self._h, which, e))
                    v = np.concatenate([v, e], axis=2)
        return v

    @property
    def lams(self):
        return self._get("smpc_get_lams", (self.B, self.H + 1, self.ndx))

    @property
    def info(self):
        return self._get("smpc_get_info", (self.B, 16))

    @property
    def status(self):
        """Per-instance status word of the last control step (smpc_get_status): bit 0 non-finite scalars, bit 1 the last line search
        failed, bit 2 primal regularisation at its upper limit.  0 = healthy."""
        out = np.zeros(self.B, np.int32)
        rc = self._lib.L.smpc_get_status(self._h, out)
        if rc < 0:
            self._lib.check(rc)
        return out

    def getStateDerivative(self, t):
        if t not in (0, 1):
            raise RuntimeError("state derivative is retained for t = 0, 1 only")
        return self._get("smpc_get_state_derivative01", (self.B, 2, self.ocp_handler._xdot_size()))[:, t, :]

    def getReferencePoses(self):
        return self._get("smpc_get_reference_poses", (self.B, self.H, self.nf, 3))

    def getContactForces(self, t=None):
        """MPC::getContactForces (reference src/mpc.cpp:354-380): contact forces of the constrained dynamics at the solution,
        [B, nf, force_size] for stage t or [B, H, nf, force_size] for every stage.  Full-dynamics problems only."""
        fs = int(self.ocp_handler.settings["force_size"])
        f = self._get("smpc_get_contact_forces", (self.B, self.H, self.nf, fs))
        return f if t is None else f[:, t]

    def updateInternalData(self, X):
        """State feedback front-end, batched on the device (reference src/robot-handler.cpp:106-149): for measured
        states X[B, nx] returns dict(feet[B, nf, 3], com[B, 3], hg[B, 6], centroidal_state[B, 9])."""
        X = np.ascontiguousarray(np.array(X, dtype=np.float64))
        if X.shape != (self.B, self.nx_in):
            raise RuntimeError("X must have shape (batch, nq + nv)")
        feet, com = np.zeros((self.B, self.nf, 3)), np.zeros((self.B, 3))
        hg, cs = np.zeros((self.B, 6)), np.zeros((self.B, 9))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._lib.check(self._lib.L.smpc_update_internal_data(self._h, X, p(feet), p(com), p(hg), p(cs)))
        return dict(feet=feet, com=com, hg=hg, centroidal_state=cs)

    def constraintDynamics(self, X, tau, contact_mask, Kp=None, Kd=None, prox_accuracy=0.0, prox_mu=0.0, prox_max_iter=0):
        """Constrained forward dynamics of the full-dynamics model, batched on the device (what the reference's
        FullDynamicsOCP gets from MultibodyConstraintFwdDynamics, src/fulldynamics.cpp:39,50-75,139): for states X[n, nx], joint
        torques tau[n, nv - 6] and contact masks [n] returns dict(a[n, nv], lam[n, fs nf] (contact frames, feet in contact
        first; fs = 3 for point feet, 6 = wrenches of flat feet), iters[n], kernel_ms).  Kinodynamics handles of the quadruped and
        full-dynamics handles of either robot (Kp / Kd: fs entries)."""
        X = np.ascontiguousarray(np.array(X, dtype=np.float64))
        tau = np.ascontiguousarray(np.array(tau, dtype=np.float64))
        mk = np.ascontiguousarray(np.array(contact_mask, dtype=np.uint32))
        n = X.shape[0]
        if X.ndim != 2 or X.shape[1] != self.nx_in or tau.shape != (n, self.nv - 6) or mk.shape != (n,):
            raise RuntimeError("X [n, nq + nv], tau [n, nv - 6], contact_mask [n] expected")
        fs = int(self.ocp_handler.settings.get("force_size", 3)) if isinstance(self.ocp_handler, FullDynamicsOCP) else 3
        a, lam, it, ms = np.zeros((n, self.nv)), np.zeros((n, fs * self.nf)), np.zeros(n, np.int32), np.zeros(1)
        p = lambda v: v.ctypes.data_as(C.c_void_p)
        kp = None if Kp is None else np.ascontiguousarray(np.array(Kp, dtype=np.float64))
        kd = None if Kd is None else np.ascontiguousarray(np.array(Kd, dtype=np.float64))
        self._lib.check(self._lib.L.smpc_full_forward_dynamics(
            self._h, n, X, tau, p(mk), None if kp is None else p(kp), None if kd is None else p(kd), float(prox_accuracy),
            float(prox_mu), int(prox_max_iter), a, lam, p(it), p(ms)))
        return dict(a=a, lam=lam, iters=it, kernel_ms=float(ms[0]))

    def riccatiFeedback(self, delay, X_meas):
        """u = interpolateLinear(us) - Ks[0] @ difference(x_meas, interpolateState(xs)) for every instance (reference
        examples/go2_fulldynamics.py:271-285)."""
        X = np.ascontiguousarray(np.array(X_meas, dtype=np.float64))
        if X.shape != (self.B, self.nx_in):
            raise RuntimeError("X_meas must have shape (batch, nq + nv)")
        u = np.zeros((self.B, self.nu))
        self._lib.check(self._lib.L.smpc_riccati_feedback(self._h, float(delay), X, u))
        return u

    def simStepDevice(self, x_device_ptr, tau_device_ptr, contact_state, dt, Kp=None, Kd=None):
        """One step of a simulated batch with states [B][nq + nv] and torques [B][nv - 6] resident in HBM: constrained forward dynamics of
        the feet in contact (Baumgarte gains Kp, Kd), then semi-implicit Euler over dt; the states are updated in place.  Asynchronous on
        this handle's stream (wait() joins); the torques must be complete (KinodynamicsID.wait()).  Kinodynamics handles of the quadruped
        and full-dynamics handles of either robot (Kp / Kd: force_size entries)."""
        c = np.ascontiguousarray(np.array([1 if b else 0 for b in contact_state], dtype=np.uint8))
        kp = np.ascontiguousarray(np.array(Kp, dtype=np.float64)) if Kp is not None else None
        kd = np.ascontiguousarray(np.array(Kd, dtype=np.float64)) if Kd is not None else None
        fs = int(self.ocp_handler.settings.get("force_size", 3)) if isinstance(self.ocp_handler, FullDynamicsOCP) else 3
        if c.shape != (self.nf,) or any(g is not None and g.shape != (fs,) for g in (kp, kd)):
            raise RuntimeError("simStepDevice: one contact flag per foot, force_size Baumgarte gains each for Kp and Kd")
        self._lib.check(self._lib.L.smpc_sim_step_device(self._h, C.c_void_p(int(x_device_ptr)), C.c_void_p(int(tau_device_ptr)), c,
                                                         kp.ctypes.data if kp is not None else None, kd.ctypes.data if kd is not None else None, float(dt)))

    def interpolate(self, delay, knots=2):
        """Targets between MPC knots for the whole-body controller, batched on the device (reference
        examples/go2_kinodynamics.py:276-284 with src/interpolator.cpp:5-78): returns (x[B, nx], acc[B, nv],
        forces[B, nf, 3]) at `delay` seconds after the last iterate."""
        fs = int(self.ocp_handler.settings["force_size"])
        x = np.zeros((self.B, self.nx))
        a = np.zeros((self.B, self.nv if self.nx != 9 else 9))  # centroidal handle: the interpolated state derivative
        f = np.zeros((self.B, self.nf * fs))
        self._lib.check(self._lib.L.smpc_interpolate(
            self._h, float(delay), int(knots), x.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p)))
        return x, a, f.reshape(self.B, self.nf, fs)

    def _timing(self, which):
        names = self.ocp_handler.model_handler.getFeetFrameNames()
        out = {}
        for f, n in enumerate(names):
            buf = np.zeros(256, np.int32)
            cnt = self._lib.check(self._lib.L.smpc_get_foot_timing(self._h, f, which, buf, 256))
            out[n] = [int(v) for v in buf[:cnt]]
        return out

    @property
    def foot_takeoff_times(self):
        return self._timing(0)

    @property
    def foot_land_times(self):
        return self._timing(1)

    # MPC::getCyclingContactState (reference include/simple-mpc/mpc.hpp:144-147): the contact sequence of generateCycleHorizon,
    # repeated to cover the horizon and rotated with every walking control step (src/mpc.cpp:103-110,230)
    def getCyclingContactState(self, t, ee_name):
        names = self.ocp_handler.model_handler.getFeetFrameNames()
        out = np.zeros(len(names), np.uint8)
        self._lib.check(self._lib.L.smpc_get_cycling_contact_state(self._h, int(t), out.ctypes.data))
        return bool(out[names.index(ee_name)])

    def getCycleHorizon(self):
        """The cycle horizon as contact states per stage ({foot: bool}); the reference returns its StageModel objects
        (include/simple-mpc/mpc.hpp:139-142), which exist only as device data here."""
        names = self.ocp_handler.model_handler.getFeetFrameNames()
        n = self._lib.L.smpc_get_cycling_contact_state(self._h, 0, None)
        out = np.zeros(len(names), np.uint8)
        res = []
        for t in range(n):
            self._lib.check(self._lib.L.smpc_get_cycling_contact_state(self._h, t, out.ctypes.data))
            res.append({nm: bool(out[i]) for i, nm in enumerate(names)})
        return res

    # MPC::setTerminalReferencePose (reference src/mpc.cpp:331-334) asks the terminal cost stack for "<foot>_pose_cost"; neither
    # OCP puts one there (src/kinodynamics.cpp:350-363, src/fulldynamics.cpp:418-431), so upstream the call fails in the lookup
    def setTerminalReferencePose(self, ee_name, pose_ref):
        raise RuntimeError("the terminal cost has no %s_pose_cost component" % ee_name)

    def getFootTakeoffCycle(self, name):
        return self._timing(0)[name]

    def getFootLandCycle(self, name):
        return self._timing(1)[name]

    def cold_trace(self):
        out = np.zeros((100, 4))
        n = self._lib.L.smpc_get_cold_trace(self._h, out, 100)
        return out[:n]

    def getSettings(self):
        return dict(self.settings)

    def getModelHandler(self):
        return self.ocp_handler.model_handler

    def getDataHandler(self, instance=0):
        """MPC::getDataHandler (reference include/simple-mpc/mpc.hpp:131-134): host-side data of the measured state the last iterate(X)
        received for `instance` (the reference state before the first control step)."""
        dh = RobotDataHandler(self.ocp_handler.model_handler)
        X = getattr(self, "_last_X", None)
        if X is not None:
            dh.updateInternalData(X[instance], False)
        return dh

    # ---- debug / profiling ----
    def debug_lq(self, inst, t):
        n = self._lib.L.smpc_lq_size(self._h)
        out = np.zeros(n)
        self._lib.check(self._lib.L.smpc_debug_get_lq(self._h, inst, t, out))
        ndx, nu, nc = self.ndx, self.nu, self.nc
        o = 0
        res = {}
        layout = (
            ("A", (ndx, ndx)), ("B", (ndx, nu)), ("Q", (ndx, ndx)), ("S", (ndx, nu)), ("R", (nu, nu)), ("C", (nc, ndx)),
            ("q", (ndx,)), ("r", (nu,)), ("f", (ndx,)), ("d", (nc,)), ("lx", (ndx,)), ("lu", (nu,)), ("lpd", (ndx,)),
            ("vpd", (nc,)),
        )
        if isinstance(self.ocp_handler, FullDynamicsOCP):
            # full-dynamics knot (smpc_full_model.h): the box rows are unit selectors kept as activity flags; Cd / Dd hold the
            # dense cone rows of 6-D feet only
            nbox = 2 * nu
            ncone = nc - nbox
            layout = (
                ("A", (ndx, ndx)), ("B", (ndx, nu)), ("Q", (ndx, ndx)), ("S", (ndx, nu)), ("R", (nu, nu)), ("Cd", (ncone, ndx)),
                ("Dd", (ncone, nu)), ("q", (ndx,)), ("r", (nu,)), ("f", (ndx,)), ("d", (nc,)), ("lx", (ndx,)), ("lu", (nu,)),
                ("lpd", (ndx,)), ("vpd", (nc,)), ("act", (nc,)),
            )
        elif int(self.ocp_handler.settings.get("force_size", 3)) == 6 and isinstance(self.ocp_handler, KinodynamicsOCP):
            # kinodynamics OCP with 6-D feet on the dense stage kernels (FullDims<..., KIN = 1>): rows [u box (absent) | joint box | 17 wrench-cone
            # rows per foot (Cd = 0, Dd constant) | 6 frame-velocity rows per foot (Cv: folded into Q / q, kept for the multiplier step)]
            nf = self.ocp_handler.model_handler.getFeetNb()
            ncone, nvel = 17 * nf, 6 * nf
            layout = (
                ("A", (ndx, ndx)), ("B", (ndx, nu)), ("Q", (ndx, ndx)), ("S", (ndx, nu)), ("R", (nu, nu)), ("Cd", (ncone, ndx)),
                ("Dd", (ncone, nu)), ("Cv", (nvel, ndx)), ("q", (ndx,)), ("r", (nu,)), ("f", (ndx,)), ("d", (nc,)), ("lx", (ndx,)),
                ("lu", (nu,)), ("lpd", (ndx,)), ("vpd", (nc,)), ("act", (nc,)),
            )
        for name, shape in layout:
            sz = int(np.prod(shape))
            res[name] = out[o : o + sz].reshape(shape).copy()
            o += sz
        return res

    def debug_steps(self):
        dxs = np.zeros((self.B, self.H + 1, self.ndx))
        dus = np.zeros((self.B, self.H, self.nu))
        self._lib.check(self._lib.L.smpc_debug_get_steps(self._h, dxs, dus))
        return dxs, dus

    def debug_terminal(self, inst):
        QN, qN = np.zeros((self.ndx, self.ndx)), np.zeros(self.ndx)
        self._lib.check(self._lib.L.smpc_debug_get_terminal(self._h, inst, QN, qN))
        return QN, qN

    def set_profiling(self, on):
        self._lib.L.smpc_set_profiling(self._h, int(on))

    def kernel_times(self):
        n = int(self._lib.L.smpc_kernel_time_slots())  # (the library says how many slots it reports; the call carries the capacity)
        ms = np.zeros(n)
        calls = np.zeros(n, np.int64)
        self._lib.check(self._lib.L.smpc_get_kernel_times_n(self._h, ms, calls, n))
        names = list(self.ocp_handler._kernel_names())
        names += ["slot%d" % i for i in range(len(names), n)]
        return {nm: (float(m), int(c)) for nm, m, c in zip(names, ms, calls) if not nm.startswith("-")}

    def reset_kernel_times(self):
        self._lib.check(self._lib.L.smpc_reset_kernel_times(self._h))


class MPC(BatchedMPC):
    """Single-instance view with the reference's exact verbs: iterate(x), xs/us/Ks as lists of vectors
    (reference bindings/expose-mpc.cpp:73-106)."""

    def __init__(self, settings, ocp, device_id=0, lib=None):
        super().__init__(settings, ocp, 1, device_id, lib)

    def iterate(self, x):
        x = np.ascontiguousarray(x, float)
        if x.shape != (self.nx,):
            raise RuntimeError("x must have size nq+nv")
        super().iterate(x[None, :])

    @property
    def xs(self):
        return list(super().xs[0])

    @property
    def us(self):
        return list(super().us[0])

    @property
    def Ks(self):
        return list(super().Ks[0])

    def getStateDerivative(self, t):
        return super().getStateDerivative(t)[0]


class Interpolator:
    """reference include/simple-mpc/interpolator.hpp / bindings/expose-interpolate.cpp: same four methods, on lists of
    numpy vectors.  The arithmetic runs in the HIP library (smpc_interpolate_knots); contacts are a table look-up."""

    def __init__(self, model, lib=None, device_id=0):
        self._lib = lib or default_lib()
        self._dev = device_id
        self.model = model

    def _run(self, kind, delay, timestep, knots):
        k = np.ascontiguousarray(np.array(knots, dtype=np.float64))
        if k.ndim != 2:
            raise RuntimeError("knots must be a list of equally sized vectors")
        out = np.zeros(k.shape[1])
        self._lib.check(self._lib.L.smpc_interpolate_knots(kind, float(delay), float(timestep), k, k.shape[0], k.shape[1], out, self._dev))
        return out

    def interpolateConfiguration(self, delay, timestep, qs):
        return self._run(1, delay, timestep, qs)

    def interpolateState(self, delay, timestep, xs):
        return self._run(0, delay, timestep, xs)

    def interpolateLinear(self, delay, timestep, vs):
        return self._run(2, delay, timestep, vs)

    def interpolateContacts(self, delay, timestep, cs):
        step = int(delay / timestep)
        return list(cs[min(max(step, 0), len(cs) - 1)])


class KinodynamicsID:
    """Whole-body inverse-dynamics controller of the reference (include/simple-mpc/inverse-dynamics/kinodynamics-id.hpp:17-92,
    src/inverse-dynamics/kinodynamics-id.cpp:7-237; bindings/expose-kinodynamics-id.cpp), batched: `batch` robots, one QP each per
    control tick, solved on the device (simple-mpc_amd/csrc/smpc_id.h).  `settings`: the reference's KinodynamicsIDSettings keys
    (friction_coefficient, contact_weight_ratio_max / min, kp_base, kp_posture, kp_contact, w_base, w_posture, w_contact_motion,
    w_contact_force, contact_motion_equality).  The reference reads effort and velocity limits from its pinocchio model; the robot table
    holds position limits only, so `effort_limit` and `velocity_limit` (nv - 6 each) are arguments.  Point feet (addPointFoot: tsid ContactPoint)
    or flat feet (addQuadFoot: tsid Contact6d -- force targets and contact forces are then 6-D wrenches per foot, in the foot frames)."""

    _KEYS = ["friction_coefficient", "contact_weight_ratio_max", "contact_weight_ratio_min", "kp_base", "kp_posture", "kp_contact", "w_base",
             "w_posture", "w_contact_motion", "w_contact_force", "contact_motion_equality"]
    _DEFAULTS = dict(friction_coefficient=0.6, contact_weight_ratio_max=10.0, contact_weight_ratio_min=0.01, kp_base=0.0, kp_posture=0.0,
                     kp_contact=0.0, w_base=-1.0, w_posture=-1.0, w_contact_motion=-1.0, w_contact_force=-1.0, contact_motion_equality=False)
    _CENTROIDAL = False

    def __init__(self, model_handler, control_dt, settings, effort_limit, velocity_limit, batch=1, device_id=0, lib=None, admm_iters=0, admm_tol=0.0,
                 base_reference_as_coded=False, tsid_joint_bounds=False):
        """base_reference_as_coded: the base task exactly as the reference codes it (kinodynamics-id.cpp:222-223, the acceleration target
        as velocity reference); tsid_joint_bounds: TSID's TaskJointPosVelAccBounds in full -- see include/smpc.h."""
        unknown = [k for k in settings if k not in self._KEYS]
        if unknown:
            raise KeyError("unknown %s settings: %s" % (type(self).__name__, unknown))
        self.settings = dict(self._DEFAULTS, **settings)
        self.model_handler = model_handler
        self._lib = lib or default_lib()
        self.B = int(batch)
        na = model_handler.nv - 6
        self._keep = [np.ascontiguousarray(np.array(a, dtype=np.float64)) for a in
                      (effort_limit, velocity_limit, model_handler.lowerPositionLimit[7:], model_handler.upperPositionLimit[7:])]
        if any(a.shape != (na,) for a in self._keep):
            raise RuntimeError("effort_limit and velocity_limit must have nv - 6 entries")
        s = self.settings
        c = IdSettingsC(s["friction_coefficient"], s["contact_weight_ratio_max"], s["contact_weight_ratio_min"], s["kp_base"], s["kp_posture"],
                        s["kp_contact"], s["w_base"], s["w_posture"], s["w_contact_motion"], s["w_contact_force"],
                        int(bool(s["contact_motion_equality"])), float(control_dt), *[a.ctypes.data for a in self._keep], int(admm_iters), 0.0, 0.0, 0.0, float(admm_tol),
                        int(self._CENTROIDAL), float(s.get("kp_com", 0.0)), float(s.get("kp_feet_tracking", 0.0)), float(s.get("w_com", -1.0)),
                        float(s.get("w_feet_tracking", -1.0)), int(bool(base_reference_as_coded)), int(bool(tsid_joint_bounds)), 3, None)
        quads = getattr(model_handler, "_quads", {})
        self._fs = 3
        if quads:  # flat feet (RobotModelHandler.addQuadFoot): every foot must be one
            names = model_handler.getFeetFrameNames()
            if any(n not in quads for n in names):
                raise RuntimeError("point and flat feet cannot be mixed")
            self._quad = np.ascontiguousarray(np.stack([quads[n] for n in names]), float)
            c.force_size, c.quad_contact_points, self._fs = 6, self._quad.ctypes.data, 6
        h = C.c_void_p()
        self._lib.check(self._lib.L.smpc_id_create(model_handler._ptr, C.byref(c), self.B, device_id, C.byref(h)))
        self._h = h
        self._nq, self._nv, self._nf = model_handler.nq, model_handler.nv, model_handler.getFeetNb()
        self._a = np.zeros((self.B, self._nv))
        self._f = np.zeros((self.B, self._fs * self._nf))
        self.resid = np.zeros(self.B)

    def __del__(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.L.smpc_id_destroy(self._h)
            self._h = None

    def reset(self, instance=-1):
        """Forget the warm start (ADMM iterate, step-size parameter) of one robot, or of every robot."""
        self._lib.check(self._lib.L.smpc_id_reset(self._h, int(instance)))

    def getResiduals(self):
        """Residuals [B] of the last solve (also after solveDevice); not finite: that robot's solve failed and its warm start was dropped."""
        self._lib.check(self._lib.L.smpc_id_get_resid(self._h, self.resid))
        return self.resid.copy()

    def setTarget(self, q_target, v_target, a_target, contact_state_target, f_target, instance=-1):
        """reference kinodynamics-id.cpp:120-183.  f_target: one 3-vector per foot (a list, or an array [nf][3]); instance = -1: every robot."""
        c = lambda x, n: np.ascontiguousarray(np.array(x, dtype=np.float64).reshape(n))
        contact = np.ascontiguousarray(np.array([1 if b else 0 for b in contact_state_target], dtype=np.uint8))
        if contact.size != self._nf:
            raise RuntimeError("contact_state_target must have one entry per foot")
        f = np.zeros(self._fs * self._nf) if len(f_target) == 0 else c(f_target, self._fs * self._nf)  # (the reference's tests pass {} with no contact)
        self._lib.check(self._lib.L.smpc_id_set_target(self._h, int(instance), c(q_target, self._nq), c(v_target, self._nv), c(a_target, self._nv), contact, f))

    def setTargets(self, Q, V, A, contact_states, F):
        """One target per robot: Q [B][nq], V [B][nv], A [B][nv], contact_states [B][nf] (or one list for all), F [B][nf][3]."""
        c = lambda x, n: np.ascontiguousarray(np.array(x, dtype=np.float64).reshape(self.B, n))
        cs = np.array(contact_states)
        cs = np.ascontiguousarray(np.broadcast_to(cs.reshape(-1, self._nf), (self.B, self._nf)).astype(np.uint8))
        self._lib.check(self._lib.L.smpc_id_set_targets(self._h, c(Q, self._nq), c(V, self._nv), c(A, self._nv), cs, c(F, self._fs * self._nf)))

    def solve(self, t, q_meas, v_meas, tau_res=None):
        """reference kinodynamics-id.cpp:185-237: one robot (vectors) or the batch (q_meas [B][nq], v_meas [B][nv]); returns tau (and fills
        tau_res when given, as the reference does)."""
        q, v = np.array(q_meas, dtype=np.float64), np.array(v_meas, dtype=np.float64)
        single = q.ndim == 1
        X = np.ascontiguousarray(np.concatenate([q.reshape(self.B, self._nq), v.reshape(self.B, self._nv)], axis=1))
        tau = np.zeros((self.B, self._nv - 6))
        self._lib.check(self._lib.L.smpc_id_solve(self._h, X, tau, self._a.ctypes.data, self._f.ctypes.data, self.resid.ctypes.data))
        out = tau[0] if single else tau
        if tau_res is not None:
            tau_res[...] = out
        return out

    def solve_device(self, x_device_ptr, tau_device_ptr=None):
        """The batch with states (and torques) resident in HBM: x [B][nq + nv], tau [B][nv - 6] device pointers; asynchronous, pair with wait().
        Without tau_device_ptr the torques stay in the handle's buffer (tau_device_ptr())."""
        self._lib.check(self._lib.L.smpc_id_solve_device(self._h, C.c_void_p(int(x_device_ptr)), C.c_void_p(int(tau_device_ptr)) if tau_device_ptr else None))

    def wait(self):
        self._lib.check(self._lib.L.smpc_id_wait(self._h))

    def tau_device_ptr(self):
        return int(self._lib.L.smpc_id_get_tau_device(self._h))

    def setTargetsFromMPC(self, mpc, delay, knots=2):
        """Targets of every robot from the solution a BatchedMPC holds (kinodynamics MPC -> KinodynamicsID, centroidal MPC -> CentroidalID),
        interpolated on the device at `delay` seconds after its last iterate and written straight into this controller's target buffers
        (the contact flags are those of the MPC's stage 0): the device-resident form of `setTargets(*mpc.interpolate(delay), ...)`."""
        self._lib.check(self._lib.L.smpc_id_set_targets_from_mpc(self._h, mpc._h, float(delay), int(knots)))

    def shareStream(self, mpc):
        """Issue this controller's work on the BatchedMPC's stream from now on (None: back to its own): MPC step, targets, QP solves and
        simulator steps form one in-order queue, and wait() is needed only before the host reads a result."""
        self._lib.check(self._lib.L.smpc_id_share_stream(self._h, mpc._h if mpc is not None else None))
        self._shared_mpc = mpc  # (keeps the owner of the stream alive for as long as this controller issues work on it)

    def x_device_ptr(self):
        """The handle's own state buffer [B][nq + nv] in HBM (solve() copies the host states there)."""
        return int(self._lib.L.smpc_id_get_x_device(self._h))

    def getAccelerations(self, ddq=None):
        out = self._a[0] if self.B == 1 else self._a
        if ddq is not None:
            ddq[...] = out
        return out.copy()

    def getContactForces(self):
        """Contact forces of the last solution: [B][nf][3] (point feet, world frame) or the wrenches T f [B][nf][6] of flat feet (foot frames)."""
        return self._f.reshape(self.B, self._nf, self._fs).copy()

    def debug(self, what):
        nm = self._fs * self._nf  # contact-motion rows: 3 per point foot, 6 per flat foot
        n = self._nv + (12 if self._fs == 6 else 3) * self._nf
        m = n + 6 + nm + (17 if self._fs == 6 else 4) * self._nf + self._nv - 6
        npad, mpad = (n + 15) // 16 * 16, (m + 15) // 16 * 16
        per = {0: (self._nv, self._nv), 1: (self._nv,), 2: (nm, self._nv), 3: (nm,), 4: (nm,), 5: (npad, npad), 6: (npad,),
               7: (mpad, npad), 8: (mpad,), 9: (mpad,), 10: (3,), 11: (3 * self._nf,), 12: (self._nv - 6,)}[what]
        out = np.zeros((self.B,) + per)
        self._lib.check(self._lib.L.smpc_id_debug_get(self._h, what, out))
        return out


class Pose(np.ndarray):
    """Reference placement of a foot: the translation as a 3-vector (what earlier versions of getReferencePose returned), with the SE3's parts as
    attributes: `.translation`, `.rotation` (3 x 3), `.homogeneous` (4 x 4)."""

    def __new__(cls, translation, rotation=None):
        obj = np.asarray(translation, dtype=np.float64).reshape(3).copy().view(cls)
        obj.rotation = np.eye(3) if rotation is None else np.asarray(rotation, dtype=np.float64).reshape(3, 3).copy()
        return obj

    def __array_finalize__(self, obj):
        if obj is not None:
            self.rotation = getattr(obj, "rotation", np.eye(3))

    @property
    def translation(self):
        return np.asarray(self)

    @property
    def homogeneous(self):
        M = np.eye(4)
        M[:3, :3] = self.rotation
        M[:3, 3] = np.asarray(self)
        return M

    def __eq__(self, other):
        if hasattr(other, "rotation") and hasattr(other, "translation"):
            return bool(np.array_equal(np.asarray(self), np.asarray(other.translation).reshape(3)) and np.array_equal(self.rotation, np.asarray(other.rotation).reshape(3, 3)))
        return np.ndarray.__eq__(self, other)

    __hash__ = None


def _pose_parts(pose_ref):
    """(translation [3], rotation [9]) of an SE3-like object, a 4 x 4 homogeneous matrix or a translation."""
    if hasattr(pose_ref, "translation"):
        p = np.ascontiguousarray(pose_ref.translation, float).reshape(3)
        R = np.ascontiguousarray(getattr(pose_ref, "rotation", np.eye(3)), float).reshape(9)
        return p, R
    a = np.asarray(pose_ref, float)
    if a.shape == (4, 4):
        return np.ascontiguousarray(a[:3, 3]), np.ascontiguousarray(a[:3, :3]).reshape(9)
    return np.ascontiguousarray(a, float).reshape(3), np.eye(3).reshape(9)


def _require_identity_rotation(pose_ref, ocp=None):
    """Targets of the inverse-dynamics controllers (CentroidalID's TaskSE3Equality, src/inverse-dynamics/centroidal-id.cpp:109-113) are tracked as
    (identity rotation, p) -- level soles: a target with another rotation is refused, not dropped.  (The OCP handlers KEEP the rotation of a
    reference pose and return it -- setReferencePose / getReferencePose -- although no solve uses it: MPC.iterate rewrites every pose with the
    identity rotation first, as the reference's does.)"""
    rot = getattr(pose_ref, "rotation", None)
    if rot is None:
        return
    if ocp is not None and not (int(ocp.settings.get("force_size", 3)) == 6 and isinstance(ocp, KinodynamicsOCP)):
        return
    if not np.allclose(np.asarray(rot, float).reshape(3, 3), np.eye(3), atol=1e-12):
        raise RuntimeError("foot reference with a non-identity rotation: this build tracks level soles only (translation of the reference placement); "
                           "rotated foot references are not implemented")


def _positions(poses, nf, flat_feet=False):
    """[nf][3] from an array of positions or a list of placements (objects with `.translation`)."""
    if flat_feet:
        for p in poses:
            _require_identity_rotation(p)
    return np.array([np.asarray(p.translation if hasattr(p, "translation") else p, dtype=np.float64).reshape(-1)[:3] for p in poses]).reshape(nf, 3)


def _linear_velocities(vels, nf, flat_feet=False):
    """[nf][3] from an array [nf][3], spatial velocities [nf][6] (linear part first) or objects with `.linear`.  flat_feet: the angular part of
    a 6-D target (TaskSE3Equality of flat feet) is not tracked by this build -- a non-zero one is refused instead of dropped."""
    if flat_feet:
        for m in vels:
            w = np.asarray(m.angular, float) if hasattr(m, "angular") else np.asarray(m, float).reshape(-1)[3:6]
            if w.size and np.abs(w).max() > 1e-12:
                raise RuntimeError("foot velocity target with a non-zero angular part: this build tracks the linear velocity of level soles only")
    return np.array([np.asarray(m.linear if hasattr(m, "linear") else m, dtype=np.float64).reshape(-1)[:3] for m in vels]).reshape(nf, 3)


class CentroidalID(KinodynamicsID):
    """reference include/simple-mpc/inverse-dynamics/centroidal-id.hpp, src/inverse-dynamics/centroidal-id.cpp:6-147 (CentroidalIDSettings
    = the KinodynamicsID keys + kp_com, kp_feet_tracking, w_com, w_feet_tracking): the base task keeps the orientation rows, a
    centre-of-mass task and a position-tracking task for every foot out of contact are added; the posture and base-orientation targets are
    the reference state.  Same device kernels as KinodynamicsID, same batch semantics."""

    _KEYS = KinodynamicsID._KEYS + ["kp_com", "kp_feet_tracking", "w_com", "w_feet_tracking"]
    _DEFAULTS = dict(KinodynamicsID._DEFAULTS, kp_com=0.0, kp_feet_tracking=0.0, w_com=-1.0, w_feet_tracking=-1.0)
    _CENTROIDAL = True

    def setTarget(self, com_position, com_velocity, feet_pose, feet_velocity, contact_state_target, f_target, instance=-1):
        """reference centroidal-id.cpp:86-147.  feet_pose: one placement per foot (objects with `.translation`, or positions [nf][3]);
        feet_velocity: one spatial velocity per foot (objects with `.linear`, [nf][6] linear part first, or [nf][3]); instance = -1: every
        robot."""
        c = lambda x, n: np.ascontiguousarray(np.array(x, dtype=np.float64).reshape(n))
        contact = np.ascontiguousarray(np.array([1 if b else 0 for b in contact_state_target], dtype=np.uint8))
        if contact.size != self._nf:
            raise RuntimeError("contact_state_target must have one entry per foot")
        f = np.zeros(self._fs * self._nf) if len(f_target) == 0 else c(f_target, self._fs * self._nf)
        self._lib.check(self._lib.L.smpc_id_set_target_centroidal(
            self._h, int(instance), c(com_position, 3), c(com_velocity, 3), c(_positions(feet_pose, self._nf, self._fs == 6), 3 * self._nf),
            c(_linear_velocities(feet_velocity, self._nf, self._fs == 6), 3 * self._nf), contact, f))

    def setTargets(self, COM, VCOM, FEET_P, FEET_V, contact_states, F):
        """One target per robot: COM, VCOM [B][3], FEET_P, FEET_V [B][nf][3], contact_states [B][nf] (or one list for all), F [B][nf][3]."""
        c = lambda x, n: np.ascontiguousarray(np.array(x, dtype=np.float64).reshape(self.B, n))
        cs = np.array(contact_states)
        cs = np.ascontiguousarray(np.broadcast_to(cs.reshape(-1, self._nf), (self.B, self._nf)).astype(np.uint8))
        self._lib.check(self._lib.L.smpc_id_set_targets_centroidal(self._h, c(COM, 3), c(VCOM, 3), c(FEET_P, 3 * self._nf), c(FEET_V, 3 * self._nf), cs,
                                                                   c(F, self._fs * self._nf)))


class FrictionCompensation:
    """reference include/simple-mpc/friction-compensation.hpp: torque += viscuous * v + dry * sign(v).  The reference reads
    the coefficients from the tail of `model.friction` / `model.damping`; here they are passed as arrays (there is no
    pinocchio.Model).  `computeFriction` accepts one vector pair or a batch [B, nu]; the torque is updated in place."""

    def __init__(self, dry_friction, viscuous_friction, lib=None, device_id=0):
        self.dry_friction_ = np.ascontiguousarray(np.array(dry_friction, dtype=np.float64))
        self.viscuous_friction_ = np.ascontiguousarray(np.array(viscuous_friction, dtype=np.float64))
        if self.dry_friction_.shape != self.viscuous_friction_.shape or self.dry_friction_.ndim != 1:
            raise RuntimeError("friction coefficient vectors must have the same size")
        self.nu_ = int(self.dry_friction_.size)
        self._lib = lib or default_lib()
        self._dev = device_id

    # the reference's read-only properties (bindings/expose-friction-compensation.cpp:33-34)
    @property
    def dry_friction(self):
        return self.dry_friction_.copy()

    @property
    def viscuous_friction(self):
        return self.viscuous_friction_.copy()

    def computeFriction(self, velocity, torque):
        v = np.ascontiguousarray(np.array(velocity, dtype=np.float64))
        if not (isinstance(torque, np.ndarray) and torque.dtype == np.float64 and torque.flags["C_CONTIGUOUS"]):
            raise RuntimeError("torque must be a C-contiguous float64 array (updated in place)")
        batch = 1 if v.ndim == 1 else v.shape[0]
        self._lib.check(self._lib.L.smpc_friction_compensation(
            self.dry_friction_, self.viscuous_friction_, self.nu_, v, v.shape[-1], torque, torque.shape[-1], batch, self._dev))
        return torque


def centroidal_dynamics(mass, gravity, timestep, X, U, contact, contact_pos, lib=None, device_id=0):
    """CentroidalFwdDynamics + IntegratorEuler with derivatives for a batch (reference src/centroidal-dynamics.cpp:79-81):
    X[B, 9] = [com; linear momentum; angular momentum], U[B, 3 nf] contact forces, contact[B, nf] flags,
    contact_pos[B, nf, 3] -> (Xnext[B, 9], A[B, 9, 9], B[B, 9, 3 nf])."""
    lib = lib or default_lib()
    X = np.ascontiguousarray(np.array(X, dtype=np.float64))
    U = np.ascontiguousarray(np.array(U, dtype=np.float64))
    cs = np.ascontiguousarray(np.array(contact, dtype=np.uint8))
    pos = np.ascontiguousarray(np.array(contact_pos, dtype=np.float64))
    Bn, nf = cs.shape
    if X.shape != (Bn, 9) or U.shape != (Bn, 3 * nf) or pos.shape != (Bn, nf, 3):
        raise RuntimeError("force size in settings does not match reference force size")
    Xn, A, Bm = np.zeros((Bn, 9)), np.zeros((Bn, 9, 9)), np.zeros((Bn, 9, 3 * nf))
    lib.check(lib.L.smpc_centroidal_dynamics(float(mass), np.ascontiguousarray(gravity, dtype=np.float64), float(timestep), nf,
                                             X, U, cs, pos, Bn, Xn, A, Bm, device_id))
    return Xn, A, Bm
