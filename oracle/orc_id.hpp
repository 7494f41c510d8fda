// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_id.hpp: CPU restatement of the whole-body inverse-dynamics QP of KinodynamicsID (reference
// src/inverse-dynamics/kinodynamics-id.cpp:7-237, include/simple-mpc/inverse-dynamics/kinodynamics-id.hpp:24-50), for robots with
// 3-D point feet.  The reference delegates the formulation to TSID 1.9 (InverseDynamicsFormulationAccForce, TaskJointPosture,
// TaskSE3Equality, ContactPoint, TaskJointPosVelAccBounds, TaskActuationBounds) and the solve to proxsuite's ProxQP, neither of which is
// in the image; what follows restates their published formulation ([UPSTREAM-RECALL] where the library's source would decide):
//
//   variables        y = [a (nv) ; f (3 per foot, world frame)]                    (inactive feet keep their variables, pinned to 0)
//   equality         M_b a + h_b = J_b^T f                                         (the six unactuated rows of the dynamics)
//   posture task     w_posture |a_j - (a_t + Kp (q_t - q) + Kd (v_t - v))|^2      (Kd = 2 sqrt(Kp) everywhere, as the reference sets it)
//   base task        w_base |a_b + drift - (Kp log6(M_b^-1 M_t) + Kd (v_t - v_b) + a_t)|^2, local frame, drift = [w x v ; 0]
//                    (as coded, kinodynamics-id.cpp:222-223 calls setDerivative twice: the base ACCELERATION target becomes TSID's velocity
//                     reference and no acceleration reference is set.  Restated literally, that closes a positive feedback from the MPC's
//                     planned base acceleration into the commanded base velocity: the MPC + ID loop of examples/go2_mpc_id_batched.py blows up
//                     within 20 control steps (factor 2.4 per step).  The evident intent -- velocity and acceleration references from the
//                     velocity and acceleration targets -- is what is built; with zero targets, as in the reference's tests, both coincide)
//   contact motion   J_f a + dJ_f v = -Kd v_f  (the contact reference is reset to the measured foot pose at every solve, :196-213:
//                    no position error) -- a cost with w_contact_motion, or an equality with contact_motion_equality
//   contact force    w_contact_force |f_f - f_t|^2 ; friction pyramid |f_x|, |f_y| <= mu f_z ; f_min <= f_z <= f_max
//   joint bounds     position / velocity limits as acceleration bounds over one control period (TaskJointPosVelAccBounds with the
//                    acceleration bound off; its viability refinement is not restated):  a in [max((-vmax - v)/dt, 2 (qmin - q - v dt)/dt^2),
//                    min((vmax - v)/dt, 2 (qmax - q - v dt)/dt^2)]
//   actuation        |M_a a + h_a - J_a^T f| <= tau_max  ;  output tau = M_a a + h_a - J_a^T f
//
// Robots with 6-D (flat) feet -- tsid::contacts::Contact6d, kinodynamics-id.cpp:41-49, 163-167, 204-208 (IDSettings::force_size == 6)
// [UPSTREAM-RECALL tsid 1.9 src/contacts/contact-6d.cpp]:
//   variables        12 per foot: the forces at the four corners of the sole (getQuadFootContactPoints), in the foot's LOCAL frame; the contact
//                    wrench is T f, T = [I I I I ; [p_1]x .. [p_4]x] (force generator matrix, 6 x 12); dynamics M a + h = S^T tau + J^T T f with the
//                    LOCAL 6-D frame Jacobian J
//   contact motion   the whole 6-D LOCAL frame acceleration: J a + dJ v = -Kd v_frame (reference pose = measured pose at every solve)
//   contact force    w_contact_force |diag(1, 1, 1e-3, 2, 2, 2) (T f - wrench_target)|^2  (Contact6d's force regularisation task weights)
//   friction         per corner the pyramid |f_x|, |f_y| <= mu f_z (16 rows) + f_min <= sum of the normal forces <= f_max (1 row): 17 rows per foot
//   foot tracking    (CentroidalID, feet in the air; position_mask all ones for QUAD feet, centroidal-id.cpp:38-41) the 6-D LOCAL task
//                    J a + dJ v = Kp log6(M_foot^-1 M_ref) + Kd (v_ref - v_frame), M_ref = (identity rotation, target position) -- the pose
//                    MPC::getReferencePose hands over (src/mpc.cpp:304-308), angular velocity target zero
//
// Solver: ADMM on  min 1/2 y^T H y + g^T y  s.t.  l <= C y <= u  (the operator splitting of OSQP: one factorisation of
// H + sigma I + C^T diag(rho) C per solve, then matrix-vector iterations; equality rows carry 1e3 rho, free rows 1e-6 rho), at most
// admm_iters iterations (residuals checked every 20, stop below 1e-7), warm-started from the previous control tick.  ProxQP is a proximal augmented-Lagrangian method; both converge to the
// same (unique, H + constraints) solution, which the tests check through KKT residuals.
#pragma once
#include "orc_full.hpp"
#include <stdexcept>

namespace orc
{
  struct IDSettings
  {
    double friction_coefficient = 0.6, contact_weight_ratio_max = 10.0, contact_weight_ratio_min = 0.01;
    double kp_base = 0, kp_posture = 0, kp_contact = 0;
    double w_base = -1, w_posture = -1, w_contact_motion = -1, w_contact_force = -1;
    bool contact_motion_equality = false;
    // CentroidalID (reference src/inverse-dynamics/centroidal-id.cpp:6-147): the base task keeps its orientation rows only, a centre-of-mass
    // task and a position-tracking task per foot out of contact are added; the posture / base targets are the reference state
    bool centroidal = false;
    double kp_com = 0, kp_feet_tracking = 0, w_com = -1, w_feet_tracking = -1;
    double control_dt = 1e-3;
    Vec tau_max, v_max, q_min, q_max; // nv - 6 each (the robot table holds position limits only)
    int admm_iters = 400; // cap; the loop stops earlier once the residuals are below ADMM_TOL
    double rho = 0.1, sigma = 1e-6, alpha = 1.6;
    double admm_tol = 1e-7; // < 0: never stop early
    // the reference AS CODED: kinodynamics-id.cpp:222-223 calls sampleBase_.setDerivative twice, so the base ACCELERATION target is TSID's
    // velocity reference and the acceleration reference stays zero (TaskSE3Equality turns the world-aligned reference back into the
    // local frame with the measured rotation, i.e. the local target itself); off: the evident intent (see the header)
    bool base_reference_as_coded = false;
    // TSID's TaskJointPosVelAccBounds restated in full ([UPSTREAM-RECALL] tsid 1.9 src/tasks/task-joint-posVelAcc-bounds.cpp): its time
    // step is TWICE the control period, the position bounds switch to the braking-distance form when the joint moves towards a
    // limit it cannot reach within a step, and the viability bounds (setImposeBounds(true, true, true, false)) are evaluated with the
    // default acceleration limit 1e10; off: position / velocity limits over one control period
    bool tsid_joint_bounds = false;
    // 6-D feet: force_size 6 and the four corners of every sole in its foot frame, [nf][4][3] (RobotModelHandler::addQuadFoot)
    int force_size = 3;
    Vec quad_points;
    int nmot() const { return force_size == 6 ? 6 : 3; }   // contact-motion rows per foot
    int nfv() const { return force_size == 6 ? 12 : 3; }   // force variables per foot
    int nfric() const { return force_size == 6 ? 17 : 4; } // friction / force-bound rows per foot
    int nfw() const { return force_size == 6 ? 6 : 3; }    // size of a force target / of the reported contact force (wrench)
  };
  // force generator matrix of foot f: wrench (local frame, at the frame origin) = T [f_1; f_2; f_3; f_4]
  inline Mat id_force_generator(const IDSettings & s, int f)
  {
    Mat T(6, 12);
    for (int c = 0; c < 4; c++)
    {
      const V3 p = v3(s.quad_points[(f * 4 + c) * 3], s.quad_points[(f * 4 + c) * 3 + 1], s.quad_points[(f * 4 + c) * 3 + 2]);
      const M3 X = skew(p);
      for (int i = 0; i < 3; i++)
      {
        T(i, 3 * c + i) = 1.0;
        for (int j = 0; j < 3; j++)
          T(3 + i, 3 * c + j) = X(i, j);
      }
    }
    return T;
  }
  constexpr double ID_WRENCH_W[6] = {1.0, 1.0, 1e-3, 2.0, 2.0, 2.0}; // Contact6d::m_weightForceRegTask [UPSTREAM-RECALL]
  // acceleration bounds [lb, ub] of one joint as TaskJointPosVelAccBounds::computeAccLimits forms them ([UPSTREAM-RECALL])
  inline void tsid_acc_limits(double q, double dq, double qmin, double qmax, double dqmax, double control_dt, double & lb, double & ub)
  {
    const double dt = 2.0 * control_dt, ddqmax = 1e10, big = 1e10;
    // position
    const double two_dt_sq = 2.0 / (dt * dt);
    const double max_q3 = two_dt_sq * (qmax - q - dt * dq), min_q3 = two_dt_sq * (qmin - q - dt * dq), mdq = -dq / dt;
    double lbp = -big, ubp = big;
    if (dq <= 0.0)
    {
      ubp = max_q3;
      if (min_q3 < mdq)
        lbp = min_q3;
      else if (q != qmin)
        lbp = std::fmax(dq * dq / (2.0 * (q - qmin)), mdq);
      else
        lbp = 1e6;
    }
    else
    {
      lbp = min_q3;
      if (max_q3 > mdq)
        ubp = max_q3;
      else if (q != qmax)
        ubp = std::fmin(-dq * dq / (2.0 * (qmax - q)), mdq);
      else
        ubp = -1e6;
    }
    // velocity
    const double lbv = (-dqmax - dq) / dt, ubv = (dqmax - dq) / dt;
    // viability
    const double dt_sq = dt * dt, dt_dq = dt * dq, two_a = 2.0 * dt_sq, q_plus = q + dt_dq;
    const double b1 = 2.0 * dt_dq + ddqmax * dt_sq, b2 = 2.0 * dt_dq - ddqmax * dt_sq;
    const double c1 = dq * dq - 2.0 * ddqmax * (qmax - q_plus), c2 = dq * dq - 2.0 * ddqmax * (q_plus - qmin);
    double ddq1 = mdq, ddq2 = mdq;
    const double d1 = b1 * b1 - 2.0 * two_a * c1, d2 = b2 * b2 - 2.0 * two_a * c2;
    if (d1 >= 0.0)
      ddq1 = (-b1 + std::sqrt(d1)) / two_a;
    if (d2 >= 0.0)
      ddq2 = (-b2 - std::sqrt(d2)) / two_a;
    const double ubvia = std::fmax(ddq1, mdq), lbvia = std::fmin(ddq2, mdq);
    // the most conservative of position / viability / velocity (the acceleration bound itself is not imposed)
    ub = std::fmin(ubp, std::fmin(ubvia, ubv));
    lb = std::fmax(lbp, std::fmax(lbvia, lbv));
    if (ub < lb)
    { // conflict: priority to the position bounds
      if (ub == ubp)
        lb = lbp;
      else
        ub = ubp;
      if (ub < lb) // (still inverted: a joint outside its range) the tighter one wins, as in the plain variant
        lb = ub = std::fmin(lb, ub);
    }
  }
  struct IDTarget
  {
    Vec q, v, a; // nq, nv, nv
    unsigned mask = 0;
    Vec f;       // 3 nf (6 nf for 6-D feet: wrench targets in the foot frames)
    Vec com, vcom, feet_p, feet_v; // CentroidalID: 3, 3, 3 nf, 3 nf (world frame)
  };
  struct IDQuantities
  {
    Mat M;          // nv x nv
    Vec nle;        // nv
    Mat J;          // 3 nf x nv: world-frame linear Jacobians of the foot points (6-D feet: 6 nf x nv, LOCAL 6-D frame Jacobians)
    Vec Jdv, vfoot; // 3 nf: classical acceleration of the points at zero joint accelerations ; their velocity (6-D feet: 6 nf, LOCAL frame)
    Vec com, footp; // 3 ; 3 nf (world frame)
    std::vector<M3> footR; // foot rotations (6-D feet)
  };
  struct QP
  {
    int n = 0, m = 0;
    Mat H, C;
    Vec g, l, u;
  };
  constexpr double ID_INF = 1e20;
  constexpr int ADMM_CHECK = 20;      // residual check period of the ADMM loop
  constexpr double ADMM_ADAPT_FLOOR = 1e-7; // rho is adapted only while the residuals are above

  inline void id_quantities(const smpc_robot_model * m, const double * x, IDQuantities & o, int force_size = 3)
  {
    ConstraintDynamics cd(m);
    cd.fs = 6; // LOCAL_WORLD_ALIGNED rows: the first three of each foot are the world-frame point Jacobian and its drift
    Vec tau(m->nv - 6, 0.0);
    cd.compute(x, x + m->nq, tau.data(), (1u << m->nfeet) - 1u);
    const int nv = m->nv, nf = m->nfeet;
    o.M = cd.Mq;
    o.nle = cd.nle;
    o.J = Mat(3 * nf, nv);
    o.Jdv.assign(3 * nf, 0.0);
    o.vfoot.assign(3 * nf, 0.0);
    for (int f = 0; f < nf; f++)
      for (int i = 0; i < 3; i++)
      {
        for (int k = 0; k < nv; k++)
        {
          o.J(3 * f + i, k) = cd.Jc(6 * f + i, k);
          o.vfoot[3 * f + i] += cd.Jc(6 * f + i, k) * x[m->nq + k];
        }
        o.Jdv[3 * f + i] = cd.gamma[6 * f + i];
      }
    if (force_size == 6)
    { // LOCAL 6-D rows: the world-aligned rows at the frame origin turned into the foot frame, [R^T lin ; R^T ang]
      o.J = Mat(6 * nf, nv);
      o.Jdv.assign(6 * nf, 0.0);
      o.vfoot.assign(6 * nf, 0.0);
      o.footR.resize(nf);
      for (int f = 0; f < nf; f++)
      {
        const M3 R = cd.R.oMi[m->foot_joint[f]].R;
        o.footR[f] = R;
        for (int blk = 0; blk < 2; blk++)
          for (int i = 0; i < 3; i++)
          {
            const int r = 6 * f + 3 * blk + i;
            for (int k = 0; k < nv; k++)
            {
              double acc = 0.0;
              for (int j = 0; j < 3; j++)
                acc += R(j, i) * cd.Jc(6 * f + 3 * blk + j, k);
              o.J(r, k) = acc;
              o.vfoot[r] += acc * x[m->nq + k];
            }
            for (int j = 0; j < 3; j++)
              o.Jdv[r] += R(j, i) * cd.gamma[6 * f + 3 * blk + j];
          }
      }
    }
    o.com = {cd.R.com[0], cd.R.com[1], cd.R.com[2]};
    o.footp.assign(3 * nf, 0.0);
    for (int f = 0; f < nf; f++)
      for (int i = 0; i < 3; i++)
        o.footp[3 * f + i] = cd.R.foot_p[f][i];
  }

  // row layout of C: [0, n) box on y ; 6 dynamics rows ; 3 nf contact-motion rows ; 4 nf friction rows ; nv - 6 actuation rows
  inline void id_assemble(const smpc_robot_model * m, const IDSettings & s, const IDTarget & t, const double * x, const IDQuantities & Q, QP & qp)
  {
    const int nq = m->nq, nv = m->nv, nf = m->nfeet, na = nv - 6;
    const int NM = s.nmot(), NFV = s.nfv(), NFR = s.nfric();
    const bool quad = s.force_size == 6;
    const int n = nv + NFV * nf;
    const int mrows = n + 6 + NM * nf + NFR * nf + na;
    // J^T G of foot f as an (nv x NFV) block: the generalised force of its force variables (G = I for a point foot, T for a flat one)
    std::vector<Mat> JG(nf);
    std::vector<Mat> Tg(nf);
    for (int f = 0; f < nf; f++)
    {
      JG[f] = Mat(nv, NFV);
      if (quad)
        Tg[f] = id_force_generator(s, f);
      for (int k = 0; k < nv; k++)
        for (int c = 0; c < NFV; c++)
        {
          double acc = 0.0;
          if (quad)
            for (int r = 0; r < 6; r++)
              acc += Q.J(6 * f + r, k) * Tg[f](r, c);
          else
            acc = Q.J(3 * f + c, k);
          JG[f](k, c) = acc;
        }
    }
    qp.n = n;
    qp.m = mrows;
    qp.H = Mat(n, n);
    qp.g.assign(n, 0.0);
    qp.C = Mat(mrows, n);
    qp.l.assign(mrows, -ID_INF);
    qp.u.assign(mrows, ID_INF);
    const double * q = x;
    const double * v = x + nq;
    auto kd = [](double kp) { return 2.0 * std::sqrt(kp); };
    // ---- costs ----
    if (s.w_posture > 0)
      for (int j = 0; j < na; j++)
      {
        const double b = t.a[6 + j] + s.kp_posture * (t.q[7 + j] - q[7 + j]) + kd(s.kp_posture) * (t.v[6 + j] - v[6 + j]);
        qp.H(6 + j, 6 + j) += s.w_posture;
        qp.g[6 + j] -= s.w_posture * b;
      }
    if (s.w_base > 0)
    {
      const SE3 Mb{quat_to_R(q + 3), v3(q[0], q[1], q[2])}, Mt{quat_to_R(t.q.data() + 3), v3(t.q[0], t.q[1], t.q[2])};
      double e[6];
      log6(inv(Mb) * Mt, e);
      const V3 wl = v3(v[3], v[4], v[5]), vl = v3(v[0], v[1], v[2]);
      const V3 dr = cross(wl, vl);
      for (int i = (s.centroidal ? 3 : 0); i < 6; i++) // (CentroidalID: orientation rows only, centroidal-id.cpp:10-20)
      {
        const double ades = s.base_reference_as_coded ? s.kp_base * e[i] + kd(s.kp_base) * (t.a[i] - v[i])
                                                      : s.kp_base * e[i] + kd(s.kp_base) * (t.v[i] - v[i]) + t.a[i]; // (the evident intent: see the header)
        const double b = ades - (i < 3 ? dr[i] : 0.0);
        qp.H(i, i) += s.w_base;
        qp.g[i] -= s.w_base * b;
      }
    }
    auto add_rows3 = [&](double w, const double * A, const double * b, int rows = 3) { // w |A y_a - b|^2, A: rows x nv row-major
      for (int i = 0; i < rows; i++)
        for (int a = 0; a < nv; a++)
        {
          qp.g[a] -= w * A[i * nv + a] * b[i];
          for (int c = 0; c < nv; c++)
            qp.H(a, c) += w * A[i * nv + a] * A[i * nv + c];
        }
    };
    if (s.centroidal && s.w_com > 0)
    {
      // a_com = J_com a + drift:  J_com = R_b M_lin / m (the base's linear rows of M are the total linear momentum map in the base
      // frame), drift = R_b nle_lin / m + g  (TaskComEquality, centroidal-id.cpp:22-27)
      const M3 Rb = quat_to_R(q + 3);
      const double im = 1.0 / m->total_mass;
      std::vector<double> Jc(3 * nv);
      double vc[3] = {0, 0, 0}, dr[3], b3[3];
      for (int i = 0; i < 3; i++)
      {
        for (int k = 0; k < nv; k++)
        {
          Jc[i * nv + k] = im * (Rb(i, 0) * Q.M(0, k) + Rb(i, 1) * Q.M(1, k) + Rb(i, 2) * Q.M(2, k));
          vc[i] += Jc[i * nv + k] * v[k];
        }
        dr[i] = im * (Rb(i, 0) * Q.nle[0] + Rb(i, 1) * Q.nle[1] + Rb(i, 2) * Q.nle[2]) + (i == 2 ? -9.81 : 0.0);
      }
      for (int i = 0; i < 3; i++)
        b3[i] = s.kp_com * (t.com[i] - Q.com[i]) + kd(s.kp_com) * (t.vcom[i] - vc[i]) - dr[i];
      add_rows3(s.w_com, Jc.data(), b3);
    }
    if (s.centroidal && s.w_feet_tracking > 0)
      for (int f = 0; f < nf; f++)
        if (!((t.mask >> f) & 1u))
        { // position tracking of the feet out of contact (centroidal-id.cpp:101-129; point feet: linear part)
          if (quad)
          { // flat feet: the 6-D LOCAL task towards (identity rotation, target position), zero angular velocity target
            const SE3 Mf{Q.footR[f], v3(Q.footp[3 * f], Q.footp[3 * f + 1], Q.footp[3 * f + 2])};
            const SE3 Mr{m3_id(), v3(t.feet_p[3 * f], t.feet_p[3 * f + 1], t.feet_p[3 * f + 2])};
            double e[6], b6[6];
            log6(inv(Mf) * Mr, e);
            const V3 vr = tr(Q.footR[f]) * v3(t.feet_v[3 * f], t.feet_v[3 * f + 1], t.feet_v[3 * f + 2]);
            for (int i = 0; i < 6; i++)
              b6[i] = s.kp_feet_tracking * e[i] + kd(s.kp_feet_tracking) * ((i < 3 ? vr[i] : 0.0) - Q.vfoot[6 * f + i]) - Q.Jdv[6 * f + i];
            add_rows3(s.w_feet_tracking, &Q.J.a[(size_t)(6 * f) * nv], b6, 6);
            continue;
          }
          double b3[3];
          for (int i = 0; i < 3; i++)
            b3[i] = s.kp_feet_tracking * (t.feet_p[3 * f + i] - Q.footp[3 * f + i]) + kd(s.kp_feet_tracking) * (t.feet_v[3 * f + i] - Q.vfoot[3 * f + i]) - Q.Jdv[3 * f + i];
          add_rows3(s.w_feet_tracking, &Q.J.a[(size_t)(3 * f) * nv], b3);
        }
    const double kdc = kd(s.kp_contact);
    for (int f = 0; f < nf; f++)
    {
      if (!((t.mask >> f) & 1u))
        continue;
      if (!s.contact_motion_equality && s.w_contact_motion > 0)
        for (int i = 0; i < NM; i++)
        {
          const int r = NM * f + i;
          const double b = -Q.Jdv[r] - kdc * Q.vfoot[r];
          for (int a = 0; a < nv; a++)
          {
            qp.g[a] -= s.w_contact_motion * Q.J(r, a) * b;
            for (int c = 0; c < nv; c++)
              qp.H(a, c) += s.w_contact_motion * Q.J(r, a) * Q.J(r, c);
          }
        }
      if (s.w_contact_force > 0 && quad)
      { // |diag(w6) (T f - wrench_target)|^2
        for (int a = 0; a < 12; a++)
        {
          for (int c = 0; c < 12; c++)
          {
            double acc = 0.0;
            for (int r = 0; r < 6; r++)
              acc += Tg[f](r, a) * ID_WRENCH_W[r] * ID_WRENCH_W[r] * Tg[f](r, c);
            qp.H(nv + 12 * f + a, nv + 12 * f + c) += s.w_contact_force * acc;
          }
          double gb = 0.0;
          for (int r = 0; r < 6; r++)
            gb += Tg[f](r, a) * ID_WRENCH_W[r] * ID_WRENCH_W[r] * t.f[6 * f + r];
          qp.g[nv + 12 * f + a] -= s.w_contact_force * gb;
        }
      }
      else if (s.w_contact_force > 0)
        for (int i = 0; i < 3; i++)
        {
          qp.H(nv + 3 * f + i, nv + 3 * f + i) += s.w_contact_force;
          qp.g[nv + 3 * f + i] -= s.w_contact_force * t.f[3 * f + i];
        }
    }
    // ---- constraints ----
    const double W = m->total_mass * 9.81, fmax = s.contact_weight_ratio_max * W, fmin = s.contact_weight_ratio_min * W, dt = s.control_dt;
    for (int i = 0; i < n; i++)
      qp.C(i, i) = 1.0;
    for (int j = 0; j < na; j++)
    {
      const double qa = q[7 + j], va = v[6 + j];
      double lb = std::fmax((-s.v_max[j] - va) / dt, 2.0 * (s.q_min[j] - qa - va * dt) / (dt * dt));
      double ub = std::fmin((s.v_max[j] - va) / dt, 2.0 * (s.q_max[j] - qa - va * dt) / (dt * dt));
      if (lb > ub) // (a joint beyond both: the tighter one wins)
        lb = ub = std::fmin(lb, ub);
      if (s.tsid_joint_bounds)
        tsid_acc_limits(qa, va, s.q_min[j], s.q_max[j], s.v_max[j], dt, lb, ub);
      qp.l[6 + j] = lb;
      qp.u[6 + j] = ub;
    }
    for (int f = 0; f < nf; f++)
    {
      const bool on = (t.mask >> f) & 1u;
      for (int i = 0; i < NFV; i++)
        if (!on)
          qp.l[nv + NFV * f + i] = qp.u[nv + NFV * f + i] = 0.0;
      if (on && !quad)
      {
        qp.l[nv + 3 * f + 2] = fmin;
        qp.u[nv + 3 * f + 2] = fmax;
      }
    }
    int r0 = n;
    for (int i = 0; i < 6; i++)
    {
      for (int k = 0; k < nv; k++)
        qp.C(r0 + i, k) = Q.M(i, k);
      for (int f = 0; f < nf; f++)
        for (int c = 0; c < NFV; c++)
          qp.C(r0 + i, nv + NFV * f + c) = -JG[f](i, c);
      qp.l[r0 + i] = qp.u[r0 + i] = -Q.nle[i];
    }
    r0 += 6;
    for (int f = 0; f < nf; f++)
      if (((t.mask >> f) & 1u) && s.contact_motion_equality)
        for (int i = 0; i < NM; i++)
        {
          const int r = NM * f + i;
          for (int k = 0; k < nv; k++)
            qp.C(r0 + r, k) = Q.J(r, k);
          qp.l[r0 + r] = qp.u[r0 + r] = -Q.Jdv[r] - kdc * Q.vfoot[r];
        }
    r0 += NM * nf;
    for (int f = 0; f < nf; f++)
      if ((t.mask >> f) & 1u)
      {
        if (quad)
        { // per corner c: rows 4 c + k: +-f_x - mu f_z, +-f_y - mu f_z <= 0 ; row 16: f_min <= sum of the normal forces <= f_max
          for (int c = 0; c < 4; c++)
            for (int k = 0; k < 4; k++)
            {
              qp.C(r0 + 17 * f + 4 * c + k, nv + 12 * f + 3 * c + k / 2) = (k % 2 == 0) ? 1.0 : -1.0;
              qp.C(r0 + 17 * f + 4 * c + k, nv + 12 * f + 3 * c + 2) = -s.friction_coefficient;
              qp.u[r0 + 17 * f + 4 * c + k] = 0.0;
            }
          for (int c = 0; c < 4; c++)
            qp.C(r0 + 17 * f + 16, nv + 12 * f + 3 * c + 2) = 1.0;
          qp.l[r0 + 17 * f + 16] = fmin;
          qp.u[r0 + 17 * f + 16] = fmax;
          continue;
        }
        for (int k = 0; k < 4; k++)
        {
          qp.C(r0 + 4 * f + k, nv + 3 * f + k / 2) = (k % 2 == 0) ? 1.0 : -1.0;
          qp.C(r0 + 4 * f + k, nv + 3 * f + 2) = -s.friction_coefficient;
          qp.u[r0 + 4 * f + k] = 0.0;
        }
      }
    r0 += NFR * nf;
    for (int j = 0; j < na; j++)
    {
      for (int k = 0; k < nv; k++)
        qp.C(r0 + j, k) = Q.M(6 + j, k);
      for (int f = 0; f < nf; f++)
        for (int c = 0; c < NFV; c++)
          qp.C(r0 + j, nv + NFV * f + c) = -JG[f](6 + j, c);
      qp.l[r0 + j] = -s.tau_max[j] - Q.nle[6 + j];
      qp.u[r0 + j] = s.tau_max[j] - Q.nle[6 + j];
    }
  }

  // ADMM (x = y of the QP); x, z, lam are the warm start on entry and the iterate on exit.  Returns max(primal, dual) residual.
  // `rho` is the step-size parameter of this instance: kept across solves (warm start) and adapted as OSQP does (Stellato et al. 2020,
  // section 5.2): at a residual check, rho <- rho sqrt((r_prim / max(|Cx|, |z|)) / (r_dual / max(|Hx|, |C^T lam|, |g|))), applied -- with a
  // new factorisation of K -- when it moves by more than a factor 5.
  inline double qp_admm(const QP & qp, double & rho, double sigma, double alpha, int iters, double tol, Vec & x, Vec & z, Vec & lam)
  {
    const int n = qp.n, m = qp.m;
    for (int i = 0; i < n; i++) // the first n rows are the box on y (id_assemble): the iteration below uses it
      for (int j = 0; j < n; j++)
        if (qp.C(i, j) != (i == j ? 1.0 : 0.0))
          throw std::runtime_error("qp_admm: the first n rows of C must be the identity");
    Vec r(m);
    Mat K;
    auto factor = [&]() {
      for (int i = 0; i < m; i++)
        r[i] = (qp.u[i] - qp.l[i] < 1e-12) ? 1e3 * rho : ((qp.l[i] <= -ID_INF && qp.u[i] >= ID_INF) ? 1e-6 * rho : rho);
      K = qp.H;
      for (int i = 0; i < n; i++)
        K(i, i) += sigma;
      for (int k = 0; k < m; k++)
        for (int i = 0; i < n; i++)
        {
          const double ci = qp.C(k, i);
          if (ci == 0.0)
            continue;
          for (int j = 0; j < n; j++)
            K(i, j) += r[k] * ci * qp.C(k, j);
        }
      bool ok = cholesky(K);
      assert(ok);
      (void)ok;
    };
    factor();
    if ((int)x.size() != n)
    {
      x.assign(n, 0.0);
      z.assign(m, 0.0);
      lam.assign(m, 0.0);
      for (int k = 0; k < m; k++)
        z[k] = std::fmin(std::fmax(0.0, qp.l[k]), qp.u[k]);
    }
    Vec rhs(n), zt(m);
    double pr, du, np_, nd_; // residuals and the norms they are measured against
    auto residual = [&]() {
      pr = du = np_ = nd_ = 0.0;
      for (int k = 0; k < m; k++)
      {
        double acc = 0.0;
        for (int i = 0; i < n; i++)
          acc += qp.C(k, i) * x[i];
        pr = std::fmax(pr, std::fabs(acc - z[k]));
        np_ = std::fmax(np_, std::fmax(std::fabs(acc), std::fabs(z[k])));
      }
      for (int i = 0; i < n; i++)
      {
        double hx = 0.0, cl = 0.0;
        for (int j = 0; j < n; j++)
          hx += qp.H(i, j) * x[j];
        for (int k = 0; k < m; k++)
          cl += qp.C(k, i) * lam[k];
        du = std::fmax(du, std::fabs((qp.g[i] + hx) + cl));
        nd_ = std::fmax(nd_, std::fmax(std::fabs(hx), std::fmax(std::fabs(cl), std::fabs(qp.g[i]))));
      }
      return std::fmax(pr, du);
    };
    double res = 0.0;
    for (int it = 0; it < iters; it++)
    {
      // every ADMM_CHECK iterations: stop once both residuals are below the tolerance (`iters` is the cap) ; adapt rho
      if (it > 0 && it % ADMM_CHECK == 0)
      {
        res = residual();
        if (tol >= 0.0 && res <= tol)
          return res;
        const double est = std::fmin(std::fmax(rho * std::sqrt((pr / (np_ + 1e-10)) / (du / (nd_ + 1e-10) + 1e-10)), 1e-6), 1e6);
        if (res > ADMM_ADAPT_FLOOR && (est > 5.0 * rho || est < 0.2 * rho)) // (below the floor the ratio of the residuals is rounding noise)
        {
          rho = est;
          factor();
        }
      }
      for (int i = 0; i < n; i++)
      {
        double acc = sigma * x[i] - qp.g[i];
        acc += r[i] * z[i] - lam[i]; // box row i: C(i, i) = 1
        for (int k = n; k < m; k++)
          acc += qp.C(k, i) * (r[k] * z[k] - lam[k]);
        rhs[i] = acc;
      }
      chol_solve_inplace(K, rhs);
      for (int k = 0; k < m; k++)
      {
        double acc = 0.0;
        for (int i = 0; i < n; i++)
          acc += qp.C(k, i) * rhs[i];
        zt[k] = k < n ? rhs[k] : acc;
      }
      for (int i = 0; i < n; i++)
        x[i] = alpha * rhs[i] + (1.0 - alpha) * x[i];
      for (int k = 0; k < m; k++)
      {
        const double zh = alpha * zt[k] + (1.0 - alpha) * z[k];
        const double zn = std::fmin(std::fmax(zh + lam[k] / r[k], qp.l[k]), qp.u[k]);
        lam[k] += r[k] * (zh - zn);
        z[k] = zn;
      }
    }
    return residual();
  }

  // KinodynamicsID for a batch of robots: setTarget (shared or per instance) + solve
  struct BatchKinoID
  {
    const smpc_robot_model * M;
    IDSettings s;
    int B;
    std::vector<IDTarget> tgt;
    std::vector<Vec> x, z, lam; // ADMM state per instance (warm start)
    std::vector<double> resid, rho; // rho: the ADMM step-size parameter each instance has adapted to
    BatchKinoID(const smpc_robot_model * m, const IDSettings & st, int B_) : M(m), s(st), B(B_), tgt(B_), x(B_), z(B_), lam(B_), resid(B_, 0.0), rho(B_, st.rho)
    {
      // default target: the reference state, every foot in contact with an equal share of the weight (kinodynamics-id.cpp:96-112)
      IDTarget t;
      t.q.assign(m->q_ref, m->q_ref + m->nq);
      t.v.assign(m->nv, 0.0);
      t.a.assign(m->nv, 0.0);
      t.mask = (1u << m->nfeet) - 1u;
      t.f.assign(s.nfw() * m->nfeet, 0.0);
      for (int f = 0; f < m->nfeet; f++)
        t.f[s.nfw() * f + 2] = m->total_mass * 9.81 / m->nfeet;
      {
        // CentroidalID defaults: the CoM of the reference state, the feet at their reference placements (centroidal-id.cpp:60-84)
        Rigid R(m);
        R.fk(t.q.data());
        t.com = {R.com[0], R.com[1], R.com[2]};
        t.vcom.assign(3, 0.0);
        t.feet_p.assign(3 * m->nfeet, 0.0);
        t.feet_v.assign(3 * m->nfeet, 0.0);
        for (int f = 0; f < m->nfeet; f++)
          for (int i = 0; i < 3; i++)
            t.feet_p[3 * f + i] = R.oMi[0].p[i] + (R.oMi[0].R * v3(m->foot_ref_p[f][0], m->foot_ref_p[f][1], m->foot_ref_p[f][2]))[i];
      }
      for (auto & e : tgt)
        e = t;
    }
    // X [B][nq + nv] -> tau [B][nv - 6], a [B][nv], f [B][3 nf] (6-D feet: the contact wrenches T f, [B][6 nf], foot frames)
    void solve(const double * X, double * tau, double * a, double * f)
    {
      const int nq = M->nq, nv = M->nv, nf = M->nfeet, na = nv - 6;
      const int NFW = s.nfw();
#pragma omp parallel for schedule(dynamic)
      for (int b = 0; b < B; b++)
      {
        const double * xb = X + (size_t)b * (nq + nv);
        IDQuantities Q;
        id_quantities(M, xb, Q, s.force_size);
        QP qp;
        id_assemble(M, s, tgt[b], xb, Q, qp);
        resid[b] = qp_admm(qp, rho[b], s.sigma, s.alpha, s.admm_iters, s.admm_tol, x[b], z[b], lam[b]);
        for (int k = 0; k < nv; k++)
          a[(size_t)b * nv + k] = x[b][k];
        // contact forces: the force variables of a point foot, the wrench T f of a flat one
        Vec w(NFW * nf, 0.0);
        for (int ff = 0; ff < nf; ff++)
        {
          if (s.force_size == 6)
          {
            const Mat T = id_force_generator(s, ff);
            for (int r = 0; r < 6; r++)
              for (int c = 0; c < 12; c++)
                w[6 * ff + r] += T(r, c) * x[b][nv + 12 * ff + c];
          }
          else
            for (int i = 0; i < 3; i++)
              w[3 * ff + i] = x[b][nv + 3 * ff + i];
        }
        for (int k = 0; k < NFW * nf; k++)
          f[(size_t)b * NFW * nf + k] = w[k];
        for (int j = 0; j < na; j++)
        {
          double acc = Q.nle[6 + j];
          for (int k = 0; k < nv; k++)
            acc += Q.M(6 + j, k) * x[b][k];
          for (int r = 0; r < NFW * nf; r++) // tau = M_a a + h_a - J_a^T (contact force / wrench)
            acc -= Q.J(r, 6 + j) * w[r];
          tau[(size_t)b * na + j] = acc;
        }

      }
    }
  };
} // namespace orc
