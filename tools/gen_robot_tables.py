#!/usr/bin/env python3
"""Generate include/smpc_robots_builtin.h: the built-in robot tables ("go2_like", "biped_like", "talos_like").

The reference loads Go2/Talos URDFs from example-robot-data (reference:
examples/go2_kinodynamics.py:17-27, tests/test_utils.cpp:14-97), which is not available here
(no network, no URDF on disk).  The tables below are therefore the *specification* of the
robots used by this repository; link parameters are Go2-like values entered by hand
(SURVEY.md H2), fixed child links (feet) are merged into their parent body here.

Run:  python tools/gen_robot_tables.py
"""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "include", "smpc_robots_builtin.h")


def inertia_mat(ixx, ixy, ixz, iyy, iyz, izz):
    return np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]], dtype=float)


def merge(bodies):
    """Merge rigidly attached bodies [(m, com, I_com)] into one (m, com, I_com)."""
    m = sum(b[0] for b in bodies)
    c = sum(b[0] * np.asarray(b[1], float) for b in bodies) / m
    I = np.zeros((3, 3))
    for mb, cb, Ib in bodies:
        d = np.asarray(cb, float) - c
        I += Ib + mb * (d @ d * np.eye(3) - np.outer(d, d))
    return m, c, I


def rot(axis, a):
    c, s = np.cos(a), np.sin(a)
    if axis == 1:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 2:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def quat_to_R(q):
    x, y, z, w = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ]
    )


class Robot:
    def __init__(self, name):
        self.name = name
        self.parent, self.jtype, self.jR, self.jp = [], [], [], []
        self.mass, self.com, self.I = [], [], []
        self.feet = []  # (name, joint, p)
        self.q_ref = None
        self.q_lo, self.q_hi = [], []

    def add_joint(self, parent, jtype, p, body, lo=0.0, hi=0.0, R=None):
        self.parent.append(parent)
        self.jtype.append(jtype)
        self.jR.append(np.eye(3) if R is None else R)
        self.jp.append(np.asarray(p, float))
        m, c, I = body
        self.mass.append(m)
        self.com.append(np.asarray(c, float))
        self.I.append(I)
        if jtype != 0:
            self.q_lo.append(lo)
            self.q_hi.append(hi)
        return len(self.parent) - 1

    def fk(self, q):
        nj = len(self.parent)
        R = [None] * nj
        p = [None] * nj
        R[0] = quat_to_R(q[3:7])
        p[0] = np.asarray(q[0:3], float)
        for j in range(1, nj):
            Rj = self.jR[j] @ rot(self.jtype[j], q[7 + j - 1])
            R[j] = R[self.parent[j]] @ Rj
            p[j] = p[self.parent[j]] + R[self.parent[j]] @ self.jp[j]
        return R, p


def go2_like():
    r = Robot("go2_like")
    base = (6.921, [0.021112, 0.0, -0.005366], inertia_mat(0.02448, 0.00012166, 0.0014849, 0.098077, -3.12e-05, 0.107))
    r.add_joint(-1, 0, [0, 0, 0], base)
    legs = [("FL", 1, 1), ("FR", 1, -1), ("RL", -1, 1), ("RR", -1, -1)]
    for nm, sx, sy in legs:
        hip = (
            0.678,
            [-0.0054 * sx, 0.00194 * sy, -0.000105],
            inertia_mat(0.00048, -3.01e-06 * sx * sy, 1.11e-06 * sx, 0.000884, -1.42e-06 * sy, 0.000596),
        )
        thigh = (
            1.152,
            [-0.00374, -0.0223 * sy, -0.0327],
            inertia_mat(0.00584, 8.72e-05 * sy, -0.000289, 0.0058, 0.000808 * sy, 0.00103),
        )
        calf = (0.154, [0.00548, -0.000975 * sy, -0.115], inertia_mat(0.00108, 3.4e-07 * sy, 1.72e-05, 0.0011, 8.28e-06 * sy, 3.29e-05))
        foot = (0.04, [0.0, 0.0, -0.213], 9.6e-06 * np.eye(3))
        calf_foot = merge([calf, foot])
        jh = r.add_joint(0, 1, [0.1934 * sx, 0.0465 * sy, 0.0], hip, -1.0472, 1.0472)
        if sx > 0:
            tlo, thi = -1.5708, 3.4907
        else:
            tlo, thi = -0.5236, 4.5379
        jt = r.add_joint(jh, 2, [0.0, 0.0955 * sy, 0.0], thigh, tlo, thi)
        jc = r.add_joint(jt, 2, [0.0, 0.0, -0.213], calf_foot, -2.7227, -0.83776)
        r.feet.append((nm + "_foot", jc, [0.0, 0.0, -0.213]))
    # "standing" reference configuration
    leg_q = lambda sy: [0.068 * sy, 0.785, -1.44]
    r.q_ref = [0, 0, 0.335, 0, 0, 0, 1] + leg_q(1) + leg_q(-1) + leg_q(1) + leg_q(-1)
    return r


def biped_like():
    """Small two-legged robot (2 x 6 leg joints) used for the two-feet gait-timing
    known-answer test of the reference (tests/mpc.cpp:46-90).  Talos-like proportions."""
    r = Robot("biped_like")
    torso = (45.0, [-0.05, 0.0, 0.15], inertia_mat(1.8, 0.0, 0.05, 1.5, 0.0, 0.6))
    r.add_joint(-1, 0, [0, 0, 0], torso)
    for nm, sy in (("left", 1), ("right", -1)):
        b = lambda m, c, d: (m, c, inertia_mat(d[0], 0, 0, d[1], 0, d[2]))
        j1 = r.add_joint(0, 3, [-0.02, 0.085 * sy, -0.27], b(1.9, [0.02, 0.0, 0.03], (0.004, 0.006, 0.004)), -0.35, 1.57)
        j2 = r.add_joint(j1, 1, [0, 0, 0], b(2.0, [-0.01, 0.0, 0.0], (0.004, 0.004, 0.003)), -0.52, 0.52)
        j3 = r.add_joint(j2, 2, [0, 0, 0], b(6.2, [0.0, 0.02 * sy, -0.15], (0.12, 0.11, 0.02)), -2.1, 0.7)
        j4 = r.add_joint(j3, 2, [0, 0, -0.38], b(3.8, [0.01, 0.0, -0.14], (0.06, 0.06, 0.008)), 0.0, 2.62)
        j5 = r.add_joint(j4, 2, [0, 0, -0.325], b(1.3, [-0.01, 0.0, 0.01], (0.003, 0.004, 0.003)), -1.27, 0.68)
        j6 = r.add_joint(j5, 1, [0, 0, 0], b(1.6, [-0.0, 0.0, -0.08], (0.004, 0.008, 0.009)), -0.52, 0.52)
        r.feet.append((nm + "_sole_link", j6, [0.0, 0.0, -0.107]))
    leg = [0.0, 0.0, -0.45, 0.9, -0.45, 0.0]
    r.q_ref = [0, 0, 1.0, 0, 0, 0, 1] + leg + leg
    # base height so that soles are at z = 0
    R, p = r.fk(r.q_ref)
    zf = (p[r.feet[0][1]] + R[r.feet[0][1]] @ np.asarray(r.feet[0][2]))[2]
    r.q_ref[2] -= zf
    return r


def talos_like():
    """Talos-like reduced humanoid: free-flyer + 2 x 6 leg joints + 2 torso joints + 2 x 4 arm joints (nq 29, nv 28), two 6-D
    feet -- the joint list the reference keeps when it builds its reduced Talos (tests/test_utils.cpp:26-62, examples/utils.py:
    14-25): legs 1-6, torso 1-2, arms 1-4, everything else locked at half_sitting and merged into the parent body.  Link
    parameters are Talos-like values entered by hand (the URDF is not available here); total mass about 91 kg."""
    r = Robot("talos_like")
    b = lambda m, c, d: (m, c, inertia_mat(d[0], 0, 0, d[1], 0, d[2]))
    r.add_joint(-1, 0, [0, 0, 0], (15.0, [-0.00978, 0.0, -0.02], inertia_mat(0.2, 0.0, 0.01, 0.1, 0.0, 0.2)))  # (mass centre placed so that the robot's CoM at half_sitting is over the centre of its soles)
    for nm, sy in (("left", 1), ("right", -1)):
        j1 = r.add_joint(0, 3, [-0.02, 0.085 * sy, -0.27], b(1.9, [0.02, 0.0, 0.03], (0.004, 0.006, 0.004)), -0.35, 1.57)
        j2 = r.add_joint(j1, 1, [0, 0, 0], b(2.0, [-0.01, 0.0, 0.0], (0.004, 0.004, 0.003)), -0.52, 0.52)
        j3 = r.add_joint(j2, 2, [0, 0, 0], b(6.2, [0.0, 0.02 * sy, -0.15], (0.12, 0.11, 0.02)), -2.1, 0.7)
        j4 = r.add_joint(j3, 2, [0, 0, -0.38], b(3.8, [0.01, 0.0, -0.14], (0.06, 0.06, 0.008)), 0.0, 2.62)
        j5 = r.add_joint(j4, 2, [0, 0, -0.325], b(1.3, [-0.01, 0.0, 0.01], (0.003, 0.004, 0.003)), -1.27, 0.68)
        j6 = r.add_joint(j5, 1, [0, 0, 0], b(1.6, [-0.0, 0.0, -0.08], (0.004, 0.008, 0.009)), -0.52, 0.52)
        r.feet.append((nm + "_sole_link", j6, [0.0, 0.0, -0.107]))
    t1 = r.add_joint(0, 3, [0.0, 0.0, 0.0722], b(3.0, [0.0, 0.0, 0.05], (0.008, 0.008, 0.006)), -1.26, 1.26)
    # torso_2 carries the chest, the head and nothing else (the arms hang below it)
    t2 = r.add_joint(t1, 2, [0.0, 0.0, 0.0], (18.0, [-0.04, 0.0, 0.2], inertia_mat(0.45, 0.0, 0.02, 0.35, 0.0, 0.25)), -0.23, 0.73)
    for nm, sy in (("left", 1), ("right", -1)):
        a1 = r.add_joint(t2, 3, [0.0, 0.1575 * sy, 0.232], b(2.7, [-0.01, 0.04 * sy, -0.02], (0.012, 0.004, 0.011)), -1.57 if sy > 0 else -0.79, 0.79 if sy > 0 else 1.57)
        a2 = r.add_joint(a1, 1, [0.00493, 0.1365 * sy, 0.04673], b(2.4, [0.02, 0.0, -0.03], (0.004, 0.006, 0.004)), 0.0 if sy > 0 else -2.87, 2.87 if sy > 0 else 0.0)
        a3 = r.add_joint(a2, 3, [0.0, 0.0, 0.0], b(2.2, [0.0, 0.0, -0.12], (0.02, 0.02, 0.003)), -2.43, 2.43)
        # arm_4 carries the forearm, the wrist and the hand (locked joints merged)
        a4 = r.add_joint(a3, 2, [0.02, 0.0, -0.273], b(3.5, [0.0, 0.0, -0.15], (0.04, 0.04, 0.004)), -2.23, 0.0)
    leg = [0.0, 0.0, -0.4, 0.8, -0.4, 0.0]
    arm_l, arm_r = [0.25, 0.17, 0.0, -0.5], [-0.25, -0.17, 0.0, -0.5]
    r.q_ref = [0, 0, 1.0, 0, 0, 0, 1] + leg + leg + [0.0, 0.006] + arm_l + arm_r  # "half_sitting"
    R, p = r.fk(r.q_ref)
    zf = (p[r.feet[0][1]] + R[r.feet[0][1]] @ np.asarray(r.feet[0][2]))[2]
    r.q_ref[2] -= zf
    return r


def fmt(v):
    return ", ".join(repr(float(x)) for x in np.asarray(v, float).ravel())


def emit(r, sym):
    nj = len(r.parent)
    R, p = r.fk(r.q_ref)
    ref_p = []
    for (_, j, fp) in r.feet:
        pw = p[j] + R[j] @ np.asarray(fp, float)
        ref_p.append(R[0].T @ (pw - p[0]))
    L = []
    L.append("static const smpc_robot_model %s = {" % sym)
    L.append('  "%s", %d, %d, %d,' % (r.name, nj, 7 + nj - 1, 6 + nj - 1))
    L.append("  {%s}," % ", ".join(str(x) for x in r.parent))
    L.append("  {%s}," % ", ".join(str(x) for x in r.jtype))
    L.append("  {%s}," % ", ".join("{%s}" % fmt(x) for x in r.jR))
    L.append("  {%s}," % ", ".join("{%s}" % fmt(x) for x in r.jp))
    L.append("  {%s}," % fmt(r.mass))
    L.append("  {%s}," % ", ".join("{%s}" % fmt(x) for x in r.com))
    L.append(
        "  {%s},"
        % ", ".join("{%s}" % fmt([I[0, 0], I[0, 1], I[1, 1], I[0, 2], I[1, 2], I[2, 2]]) for I in r.I)
    )
    L.append("  %d," % len(r.feet))
    L.append("  {%s}," % ", ".join('"%s"' % f[0] for f in r.feet))
    L.append("  {%s}," % ", ".join(str(f[1]) for f in r.feet))
    L.append("  {%s}," % ", ".join("{%s}" % fmt(f[2]) for f in r.feet))
    L.append("  {%s}," % ", ".join("{%s}" % fmt(x) for x in ref_p))
    L.append("  {%s}," % fmt(r.q_ref))
    L.append("  {%s}," % fmt(r.q_lo))
    L.append("  {%s}," % fmt(r.q_hi))
    L.append("  %r" % float(sum(r.mass)))
    L.append("};")
    return "\n".join(L)


def main():
    out = []
    out.append("/* GENERATED by tools/gen_robot_tables.py -- do not edit by hand.")
    out.append(" * Built-in robot tables (data only).  See smpc_robot.h for the layout and the")
    out.append(" * generator for provenance of every number. */")
    out.append("#ifndef SMPC_ROBOTS_BUILTIN_H")
    out.append("#define SMPC_ROBOTS_BUILTIN_H")
    out.append('#include "smpc_robot.h"')
    out.append("")
    out.append(emit(go2_like(), "SMPC_ROBOT_GO2_LIKE"))
    out.append("")
    out.append(emit(biped_like(), "SMPC_ROBOT_BIPED_LIKE"))
    out.append("")
    out.append(emit(talos_like(), "SMPC_ROBOT_TALOS_LIKE"))
    out.append("")
    out.append("#endif")
    with open(OUT, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
