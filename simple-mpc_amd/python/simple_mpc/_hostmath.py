"""Host-side (NumPy) rigid-body helpers of the Python mirror: what the reference's RobotModelHandler / RobotDataHandler answer on the
host for ONE state (src/robot-handler.cpp:76-149) -- difference, frame placements, centre of mass, centroidal momentum.  The batched
counterparts run on the device (BatchedMPC.updateInternalData); nothing here is on the iterate() path."""
import numpy as np


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def quat_to_R(q):
    """(x, y, z, w) -> rotation matrix (the free-flyer convention of the state vector)."""
    x, y, z, w = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def log3(R):
    c = min(max(0.5 * (np.trace(R) - 1.0), -1.0), 1.0)
    th = np.arccos(c)
    w = 0.5 * np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-8:
        return w
    if np.pi - th < 1e-6:  # near pi: the axis from the symmetric part
        A = 0.5 * (R + np.eye(3))
        k = int(np.argmax(np.diag(A)))
        a = A[:, k] / np.sqrt(A[k, k])
        if np.dot(a, w) < 0:
            a = -a
        return th * a
    return th / np.sin(th) * w


def log6(R, p):
    """se(3) logarithm of the placement (R, p): (v, w) with exp6(v, w) = (R, p)."""
    w = log3(R)
    th = np.linalg.norm(w)
    W = skew(w)
    if th < 1e-8:
        Vinv = np.eye(3) - 0.5 * W + W @ W / 12.0
    else:
        Vinv = np.eye(3) - 0.5 * W + (1.0 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * (W @ W)
    return np.concatenate([Vinv @ p, w])


class SE3:
    """Minimal stand-in for pinocchio.SE3: .rotation, .translation, .homogeneous, inverse(), act on points / placements."""

    def __init__(self, R=None, p=None):
        self.rotation = np.eye(3) if R is None else np.array(R, float)
        self.translation = np.zeros(3) if p is None else np.array(p, float)

    @property
    def homogeneous(self):
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = self.rotation, self.translation
        return T

    def inverse(self):
        return SE3(self.rotation.T, -self.rotation.T @ self.translation)

    def __mul__(self, o):
        if isinstance(o, SE3):
            return SE3(self.rotation @ o.rotation, self.rotation @ o.translation + self.translation)
        return self.rotation @ np.asarray(o, float) + self.translation

    def act(self, o):
        return self * o


def _joint_R(jt, ang):
    s, c = np.sin(ang), np.cos(ang)
    if jt == 1:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if jt == 2:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def _axis(jt):
    return np.eye(3)[{1: 0, 2: 1}.get(jt, 2)]


def kinematics(m, x):
    """Joint placements (world), body spatial velocities (world-frame angular velocity and velocity of the joint origin), centre of
    mass, centroidal momentum of the robot table m at state x = (q, v).  Joint 0 is the free flyer (velocity in its local frame)."""
    nj, nq, nv = m.njoints, m.nq, m.nv
    q, v = np.asarray(x[:nq], float), np.asarray(x[nq : nq + nv], float)
    R, p, w, vo = [None] * nj, [None] * nj, [None] * nj, [None] * nj
    R[0], p[0] = quat_to_R(q[3:7]), q[:3].copy()
    w[0], vo[0] = R[0] @ v[3:6], R[0] @ v[:3]
    for j in range(1, nj):
        par = m.parent[j]
        Rp = np.array(m.jp_R[j]).reshape(3, 3)
        R[j] = R[par] @ Rp @ _joint_R(m.jtype[j], q[6 + j])
        p[j] = p[par] + R[par] @ np.array(m.jp_p[j])
        w[j] = w[par] + R[j] @ _axis(m.jtype[j]) * v[5 + j]
        vo[j] = vo[par] + np.cross(w[par], p[j] - p[par])
    mass = np.array(m.mass[:nj])
    cw = [p[j] + R[j] @ np.array(m.com[j]) for j in range(nj)]
    com = sum(mass[j] * cw[j] for j in range(nj)) / mass.sum()
    h_lin, h_ang = np.zeros(3), np.zeros(3)
    for j in range(nj):
        vc = vo[j] + np.cross(w[j], cw[j] - p[j])
        i6 = m.inertia[j]  # xx xy yy xz yz zz about the body's centre of mass, joint axes (include/smpc_robot.h)
        I = np.array([[i6[0], i6[1], i6[3]], [i6[1], i6[2], i6[4]], [i6[3], i6[4], i6[5]]])
        h_lin += mass[j] * vc
        h_ang += R[j] @ I @ R[j].T @ w[j] + mass[j] * np.cross(cw[j] - com, vc)
    return dict(R=R, p=p, com=com, hg=np.concatenate([h_lin, h_ang]))


def difference(nq, nv, x1, x2):
    """RobotModelHandler::difference (reference src/robot-handler.cpp:81-95): pinocchio.difference on q (free flyer: log6 of the relative
    placement, joints: plain difference), plain difference on v."""
    x1, x2 = np.asarray(x1, float), np.asarray(x2, float)
    R1, R2 = quat_to_R(x1[3:7]), quat_to_R(x2[3:7])
    dq = np.concatenate([log6(R1.T @ R2, R1.T @ (x2[:3] - x1[:3])), x2[7:nq] - x1[7:nq]])
    return np.concatenate([dq, x2[nq : nq + nv] - x1[nq : nq + nv]])
