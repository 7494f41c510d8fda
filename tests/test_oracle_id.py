"""Oracle of the whole-body inverse-dynamics QP (oracle/orc_id.hpp): the acceptance properties of the reference's own
tests (tests/inverse-dynamics/kinodynamics-id.cpp:46-91 joint / torque limits at every step, :110-143 posture error decreasing, :145-175
contact velocity below 1e-2, :231-266 base error decreasing then below 2e-2, :268-300 whole-state error decreasing) on the go2_like robot,
integrated the way that test integrates (:53-60), plus the KKT conditions of the QP the ADMM iterations return."""
import numpy as np
import pytest

import mpc_setup as S
import oracle_lib as O

DT = 1e-3


@pytest.fixture(scope="module")
def rb():
    return O.Robot("go2_like")


def crouch(rb):
    """The reference test's start (solo_q_start, :14-27) for this robot: legs folded, base lowered so that the feet stay where they are."""
    x = rb.x_ref.copy()
    for leg in range(4):
        x[7 + 3 * leg + 1] += 0.25
        x[7 + 3 * leg + 2] -= 0.5
    x[2] += (rb.centroidal(rb.x_ref)["feet"][:, 2] - rb.centroidal(x)["feet"][:, 2]).mean()
    return x


def step(rb, x, a):
    """q <- integrate(q, (v + a dt / 2) dt) ; v <- v + a dt (reference test :53-60)."""
    nq, nv = rb.nq, rb.nv
    q = rb.integrate(np.r_[x[:nq], np.zeros(nv)], np.r_[(x[nq:] + 0.5 * a * DT) * DT, np.zeros(nv)])[:nq]
    return np.r_[q, x[nq:] + a * DT]


def qerr(rb, x, sl):
    d = rb.difference(np.r_[x[: rb.nq], np.zeros(rb.nv)], np.r_[rb.x_ref[: rb.nq], np.zeros(rb.nv)]) if hasattr(rb, "difference") else None
    return np.linalg.norm(d[sl])


class Sim:
    def __init__(self, rb, **kw):
        self.rb = rb
        self.s = O.id_settings(rb, DT, **kw)
        self.id = O.OracleKinoID(rb, self.s, 1)
        self.x = rb.x_ref.copy()
        self.tau = np.zeros(rb.nv - 6)

    def step(self):
        tau, a, f = self.id.solve(self.x[None, :])
        self.tau, self.a, self.f = tau[0], a[0], f[0]
        self.x = step(self.rb, self.x, self.a)
        q, v = self.x[7 : self.rb.nq], self.x[self.rb.nq + 6 :]
        assert np.all(q <= self.s["q_max"] + 1e-6) and np.all(q >= self.s["q_min"] - 1e-6)  # check_joint_limits, :74-88 (to the ADMM residual)
        assert np.all(np.abs(v) <= self.s["v_max"] + 1e-6)
        assert np.all(np.abs(self.tau) <= self.s["tau_max"] + 1e-6)


def static_forces(rb, x=None, contact=(True,) * 4):
    """Vertical forces on the feet in contact that hold the posture still: sum = weight, no moment about the centre of mass (least norm
    when four feet share it).  The reference's tests share the weight equally, which is static for their symmetric robot; this robot's
    centre of mass sits ahead of the feet's centroid."""
    c = rb.centroidal(rb.x_ref if x is None else x)
    on = [i for i in range(4) if contact[i]]
    r = c["feet"][on, :2] - c["com"][:2]
    A = np.vstack([np.ones(len(on)), r.T])
    fz = np.linalg.lstsq(A, np.array([rb.mass * 9.81, 0.0, 0.0]), rcond=None)[0]
    assert np.all(fz > 0), "the centre of mass must lie inside the support polygon"
    f = np.zeros((4, 3))
    f[on, 2] = fz
    return f.reshape(-1)


def test_qp_solution_satisfies_kkt(rb):
    s = O.id_settings(rb, DT, kp_base=10.0, kp_posture=1.0, kp_contact=10.0, w_base=10.0, w_posture=0.1, w_contact_force=1.0, w_contact_motion=1.0,
                      admm_iters=400)
    kid = O.OracleKinoID(rb, s, 1)
    x = S.random_states(rb, 1)[0]
    tau, a, f = kid.solve(x[None, :])
    assert kid.resid[0] < 1e-8
    H, g, Cm, l, u = kid.qp(0, x)
    y = np.r_[a[0], f[0]]
    z = Cm @ y
    assert np.all(z >= l - 1e-7 * (1 + np.abs(l) * (l > -1e19))) and np.all(z <= u + 1e-7 * (1 + np.abs(u) * (u < 1e19)))
    # stationarity with multipliers of the right sign: solve the equality-constrained QP on the active set and compare
    act = (np.abs(z - l) < 1e-6) | (np.abs(z - u) < 1e-6)
    A = Cm[act]
    KKT = np.block([[H + 1e-12 * np.eye(len(y)), A.T], [A, np.zeros((A.shape[0], A.shape[0]))]])
    sol = np.linalg.lstsq(KKT, np.r_[-g, z[act]], rcond=None)[0]
    assert np.abs(sol[: len(y)] - y).max() < 1e-5 * (1 + np.abs(y).max())
    # the dynamics hold: M a + h = S^T tau + J^T f
    Q = O.id_quantities(rb, x)
    res = Q["M"] @ a[0] + Q["nle"] - Q["J"].T @ f[0] - np.r_[np.zeros(6), tau[0]]
    assert np.abs(res).max() < 1e-6 * (1 + np.abs(Q["nle"]).max())


def test_posture_task(rb):
    sim = Sim(rb, kp_posture=20.0, w_posture=1.0)
    sim.id.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), [False] * 4, np.zeros(12))
    sim.x = crouch(rb)
    prev = None
    for _ in range(400):
        sim.step()
        sim.x[:7] = rb.x_ref[:7]  # compensate the free fall: only the posture matters (:131-133)
        sim.x[rb.nq : rb.nq + 6] = 0.0
        e = np.linalg.norm(sim.x[7 : rb.nq] - rb.x_ref[7 : rb.nq])
        assert prev is None or e <= prev
        prev = e
    assert prev < 1.0


@pytest.mark.parametrize("equality", [False, True])
def test_contacts_hold(rb, equality):
    sim = Sim(rb, kp_base=1.0, kp_contact=10.0, w_base=1.0, w_contact_motion=10.0, w_contact_force=1.0, contact_motion_equality=equality)
    sim.id.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), [True] * 4, static_forces(rb))
    sim.x = crouch(rb)
    for _ in range(300):
        sim.step()
        vf = O.id_quantities(rb, sim.x)["vfoot"].reshape(4, 3)
        assert np.linalg.norm(vf, axis=1).max() <= 1e-2  # :158-160
    assert np.all(sim.f[2::3] > 0.0)


def test_base_task(rb):
    # (force weight: the reference uses 1.0 with a robot whose static forces ARE its default force target; here the crouched posture
    #  shifts the static distribution, and a strong pull towards the standing one tilts the base before it rises)
    sim = Sim(rb, kp_base=7.0, kp_contact=0.1, w_base=100.0, w_contact_force=1e-3, w_contact_motion=1.0)
    sim.id.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), [True] * 4, static_forces(rb))
    sim.x = crouch(rb)
    prev, n = None, 2500
    for i in range(n):
        sim.step()
        e = np.linalg.norm(rb.difference(sim.x, rb.x_ref)[:6])
        if e > 2e-2:
            assert prev is None or e <= prev  # strictly decreasing until converged (:253-260)
        if i > 9 * n // 10:
            assert e < 2e-2
        prev = e


def test_all_tasks(rb):
    sim = Sim(rb, kp_base=10.0, kp_posture=1.0, kp_contact=10.0, w_base=10.0, w_posture=0.1, w_contact_force=1e-3, w_contact_motion=1.0)
    sim.id.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), [True] * 4, static_forces(rb))
    sim.x = crouch(rb)
    prev = None
    for _ in range(600):
        sim.step()
        e = np.linalg.norm(rb.difference(sim.x, rb.x_ref)[: rb.nv])
        assert prev is None or e <= prev  # :293-297
        prev = e


# ---- CentroidalID (reference src/inverse-dynamics/centroidal-id.cpp; tests/inverse-dynamics/centroidal-id.cpp:249-400) ----
class SimC(Sim):
    def __init__(self, rb, **kw):
        super().__init__(rb, centroidal=True, **kw)
        c = rb.centroidal(rb.x_ref)
        self.com0, self.feet0 = c["com"].copy(), c["feet"].copy()


def test_centroidal_id_com_task(rb):
    # (a light posture task on top of the reference test's settings: with the CoM and the contact tasks alone the QP leaves the joints'
    #  null-space motion undetermined -- upstream ProxQP's proximal term keeps it near its warm start; here it would drift)
    sim = SimC(rb, kp_com=7.0, kp_contact=0.1, w_com=100.0, w_contact_force=1e-3, w_contact_motion=1.0, kp_posture=1.0, w_posture=0.01)
    sim.x = crouch(rb)
    c = rb.centroidal(sim.x)
    target = c["com"] + np.array([-0.01, -0.01, 0.03])  # a reachable point beside the crouched CoM (:308)
    sim.id.setTargetCentroidal(target, np.zeros(3), c["feet"], np.zeros((4, 3)), [True] * 4, static_forces(rb))
    prev, n = None, 2500
    for i in range(n):
        sim.step()
        e = np.linalg.norm(rb.centroidal(sim.x)["com"] - target)
        if e > 2e-3:  # (the reference tests use 1e-3 for both thresholds; the error hovers there before it settles)
            assert prev is None or e <= prev  # :337-338
        if i > 9 * n // 10:
            assert e < 1e-3
        prev = e


def test_centroidal_id_base_orientation_task(rb):
    # (plus CoM and light posture tasks and a small force weight: this ID's base task is orientation only; with the reference test's
    #  settings alone nothing holds the height of this robot, it sinks until the QP runs out of joint velocity)
    sim = SimC(rb, kp_base=7.0, kp_contact=0.1, w_base=100.0, w_contact_force=1e-3, w_contact_motion=1.0, kp_posture=1.0, w_posture=0.01,
               kp_com=7.0, w_com=10.0)
    q = np.array([0.0025, 0.05, 0.05, 0.998])  # ~6 degrees of pitch and yaw (:267-268)
    sim.x[3:7] = q / np.linalg.norm(q)
    c = rb.centroidal(sim.x)
    sim.id.setTargetCentroidal(c["com"], np.zeros(3), c["feet"], np.zeros((4, 3)), [True] * 4, static_forces(rb))
    prev, n = None, 5000
    for i in range(n):
        sim.step()
        e = np.linalg.norm(rb.difference(sim.x, rb.x_ref)[3:6])
        if e > 1e-3:
            assert prev is None or e <= prev  # :283-284
        if i > 9 * n // 10:
            assert e < 1e-3
        prev = e
    assert abs(sim.x[2] - rb.x_ref[2]) < 5e-3


def test_centroidal_id_foot_tracking_task(rb):
    # (rear foot, force targets that balance the robot on the other three, CoM and base tasks on: this robot's centre of mass sits ahead
    #  of the feet's centroid and there is no ground in this simulation, with a front foot up and equal force shares it tips over; the
    #  reference pitches its lighter, symmetric robot back for the same reason, :357-359)
    sim = SimC(rb, kp_feet_tracking=5.0, kp_posture=0.1, kp_contact=1.0, w_feet_tracking=1e3, w_posture=1.0, w_contact_force=1e-3,
               contact_motion_equality=True, kp_com=7.0, w_com=10.0, kp_base=7.0, w_base=10.0)
    c = rb.centroidal(sim.x)
    feet = c["feet"].copy()
    k = 2
    contact = [i != k for i in range(4)]
    feet[k] += [0.05, -0.05, 0.05]  # lift the foot: 5 cm forwards, 5 cm inwards, 5 cm up (:377-380)
    sim.id.setTargetCentroidal(c["com"], np.zeros(3), feet, np.zeros((4, 3)), contact, static_forces(rb, contact=contact))
    prev, n = None, 5000
    for i in range(n):
        sim.step()
        e = np.linalg.norm(rb.centroidal(sim.x)["feet"][k] - feet[k])
        if e > 1e-3:
            assert prev is None or e <= prev  # :399-400
        if i > 9 * n // 10:
            assert e < 1e-3
        prev = e
    assert abs(sim.x[2] - rb.x_ref[2]) < 2e-2


# ---- the two reference-fidelity switches (include/smpc.h: base_reference_as_coded, tsid_joint_bounds) ----
def test_base_reference_as_coded_is_the_reference_literally(rb):
    """kinodynamics-id.cpp:222-223 sets the base ACCELERATION target as the velocity reference and no acceleration reference: the QP of the
    as-coded variant with targets (v, a) is the QP of the default variant with targets (a, 0); with zero base targets -- the reference's
    own tests -- the two variants coincide."""
    kw = dict(kp_base=10.0, kp_posture=1.0, kp_contact=10.0, w_base=10.0, w_posture=0.1, w_contact_force=1e-3, w_contact_motion=1.0)
    coded = O.OracleKinoID(rb, O.id_settings(rb, DT, base_reference_as_coded=True, **kw), 1)
    plain = O.OracleKinoID(rb, O.id_settings(rb, DT, **kw), 1)
    rng = np.random.default_rng(3)
    x = S.random_states(rb, 1, seed=4)[0]
    fs = static_forces(rb)
    vt, at = rng.normal(size=rb.nv) * 0.3, rng.normal(size=rb.nv)
    # as coded with (v, a)  ==  default with base velocity target a[:6] and base acceleration target 0 (joint parts untouched)
    vt2, at2 = vt.copy(), at.copy()
    vt2[:6], at2[:6] = at[:6], 0.0
    coded.setTarget(rb.x_ref[: rb.nq], vt, at, [True] * 4, fs)
    plain.setTarget(rb.x_ref[: rb.nq], vt2, at2, [True] * 4, fs)
    Hc, gc, Cc, lc, uc = coded.qp(0, x)
    Hp, gp, Cp, lp, up = plain.qp(0, x)
    assert np.array_equal(Hc, Hp) and np.allclose(gc, gp, rtol=0, atol=1e-12) and np.array_equal(Cc, Cp)
    # the variants differ for non-zero base targets ...
    plain.setTarget(rb.x_ref[: rb.nq], vt, at, [True] * 4, fs)
    assert np.abs(plain.qp(0, x)[1] - gc).max() > 1e-2
    # ... and coincide for zero ones
    z = np.zeros(rb.nv)
    coded.setTarget(rb.x_ref[: rb.nq], z, z, [True] * 4, fs)
    plain.setTarget(rb.x_ref[: rb.nq], z, z, [True] * 4, fs)
    assert np.array_equal(coded.qp(0, x)[1], plain.qp(0, x)[1])


def _acc_rows(rb, x, **kw):
    k = O.OracleKinoID(rb, O.id_settings(rb, DT, kp_posture=1.0, w_posture=1.0, **kw), 1)
    _, _, _, l, u = k.qp(0, x)
    return l[6 : rb.nv], u[6 : rb.nv]


def test_tsid_joint_bounds_properties(rb):
    """TaskJointPosVelAccBounds restated ([UPSTREAM-RECALL], oracle/orc_id.hpp tsid_acc_limits): time step 2 dt; away from the limits the
    velocity bound binds; the viability bound evaluated with TSID's default acceleration limit 1e10 equals the position bound to rounding
    (i.e. it is inactive on any trajectory, which is what the default variant assumes); close to a limit and moving towards it the
    braking-distance form takes over; a lower bound never exceeds the upper one."""
    s = O.id_settings(rb, DT)
    x = rb.x_ref.copy()
    x[rb.nq + 6 :] = 0.5
    l, u = _acc_rows(rb, x, tsid_joint_bounds=True)
    dt2 = 2 * DT
    q, v = x[7 : rb.nq], x[rb.nq + 6 :]
    assert np.all(l <= u)
    pos_u = 2.0 * (s["q_max"] - q - dt2 * v) / dt2**2
    vel_u = (s["v_max"] - v) / dt2
    assert np.allclose(u, np.minimum(pos_u, vel_u), rtol=1e-5)  # (the viability bound coincides with pos_u to ~1e-6: 4e4 against O(1) in fp64)
    # moving towards the upper limit from 1 mm below it, faster than one step can stop: the braking form
    j = 2
    x2 = rb.x_ref.copy()
    x2[7 + j] = s["q_max"][j] - 1e-3
    x2[rb.nq + 6 + j] = 2.0
    l2, u2 = _acc_rows(rb, x2, tsid_joint_bounds=True)
    assert np.isclose(u2[j], min(-(2.0**2) / (2 * 1e-3), -2.0 / dt2), rtol=1e-9) and l2[j] <= u2[j]
    # the default variant on the same state (one control period, plain position / velocity bounds)
    l0, u0 = _acc_rows(rb, x)
    assert np.allclose(u0, np.minimum(2.0 * (s["q_max"] - q - DT * v) / DT**2, (s["v_max"] - v) / DT), rtol=1e-12)


def test_posture_task_with_tsid_joint_bounds(rb):
    """the reference's posture test (tests/inverse-dynamics/kinodynamics-id.cpp:110-143) with the full TSID bounds: limits kept, error down"""
    sim = Sim(rb, kp_posture=20.0, w_posture=1.0, tsid_joint_bounds=True)
    sim.id.setTarget(rb.x_ref[: rb.nq], np.zeros(rb.nv), np.zeros(rb.nv), [False] * 4, np.zeros(12))
    sim.x = crouch(rb)
    prev = None
    for _ in range(300):
        sim.step()
        sim.x[:7] = rb.x_ref[:7]
        sim.x[rb.nq : rb.nq + 6] = 0.0
        e = np.linalg.norm(sim.x[7 : rb.nq] - rb.x_ref[7 : rb.nq])
        assert prev is None or e <= prev
        prev = e
