// smpc_id.h -- batched whole-body inverse-dynamics QP: the KinodynamicsID and CentroidalID controllers of the reference
// (src/inverse-dynamics/kinodynamics-id.cpp:7-237, centroidal-id.cpp:6-147; SURVEY 8f row f3, the "proxsuite contact-force QP"
// downstream of MPC::iterate) for robots with 3-D point feet, one wavefront per robot.
//
// The reference builds the problem with TSID 1.9 (InverseDynamicsFormulationAccForce + tasks) and solves it with proxsuite's ProxQP;
// neither library is available, so the formulation is restated (the CPU checker holds the same statement, term by term, with the
// places where upstream source would decide marked [UPSTREAM-RECALL]):
//     y = [a ; f],   M_b a + h_b = J_b^T f,   posture / base / contact-motion / contact-force least-squares tasks,
//     friction pyramids, force bounds, joint position / velocity limits as acceleration bounds, |tau| <= tau_max,   tau = M_a a + h_a - J_a^T f
// Three kernels, each with its own parity test against the oracle:
//   id_quant_body     joint-space inertia, bias forces, world-frame foot Jacobians, their drift and the foot velocities: the phases of the
//                     full-dynamics stage kernel (smpc_full_stage.h) up to the contact rows
//   id_assemble_body  H, g, C, l, u of the QP (n = nv + 3 nf = 30 variables, m = 76 rows for a quadruped), padded to 32 x 80
//   qp_admm_body      min 1/2 y^T H y + g^T y, l <= C y <= u by ADMM (the operator splitting of OSQP: K = H + sigma I + C^T diag(rho) C is
//                     assembled and inverted once per solve on the matrix cores, the iterations are matrix-vector products out of
//                     registers with v_readlane broadcasts; residual-based stop, OSQP's rho adaptation), warm-started from the previous
//                     control tick; then tau
#pragma once
#include "smpc_full_stage.h"

namespace smpc
{
  struct IdSettingsDev
  {
    double friction_coefficient, ratio_max, ratio_min;
    double kp_base, kp_posture, kp_contact;
    double w_base, w_posture, w_contact_motion, w_contact_force;
    int contact_motion_equality, admm_iters;
    double control_dt, rho, sigma, alpha, admm_tol;
    // CentroidalID (reference src/inverse-dynamics/centroidal-id.cpp:6-147): orientation-only base task, CoM task, tracking of the feet in the air
    int centroidal, base_as_coded;
    double kp_com, kp_feet_tracking, w_com, w_feet_tracking;
    int tsid_bounds, pad_;
  };
  template <class D>
  struct IdDims
  {
    static constexpr int NV = D::NV, NQ = D::NQ, NX = D::NX, NF = D::NF, NA = D::NV - 6, FS = D::FS;
    // point feet (tsid ContactPoint): 3 force variables, 3 motion rows, 4 friction rows per foot; flat feet (tsid Contact6d, FS == 6): the forces at
    // the four corners of the sole (12 variables, foot frame), the 6-D LOCAL frame motion (6 rows), 16 pyramid rows + the bound on the total
    // normal force (17 rows); NFW: size of a force target / of the reported contact force (the wrench T f of a flat foot)
    static constexpr int NFV = FS == 6 ? 12 : 3, NM = FS == 6 ? 6 : 3, NFR = FS == 6 ? 17 : 4, NFW = FS == 6 ? 6 : 3;
    static constexpr int N = NV + NFV * NF;                     // variables [a ; f]
    static constexpr int M = N + 6 + NM * NF + NFR * NF + NA;   // rows: box | dynamics | contact motion | friction | actuation
    static constexpr int NP = ((N + 15) / 16) * 16, MP = ((M + 15) / 16) * 16, LDC = NP + 1; // padded sizes ; row stride of C in LDS
    static constexpr int R_DYN = N, R_MOT = N + 6, R_FRI = R_MOT + NM * NF, R_ACT = R_FRI + NFR * NF;
    static constexpr int GR = M - N;                            // general rows (the first N rows of C are the identity: the box on y)
    static_assert(FS == 6 || NP == 32, "point feet: the K inverse is instantiated for 32 x 32");
  };
  constexpr double ID_INF = 1e20;

  template <class D>
  struct IdBuffers
  {
    int B = 0;
    DevModel<D> * model = nullptr;
    const double * X = nullptr;                                   // [B][NX] measured states
    double *Mq = nullptr, *nle = nullptr, *J = nullptr, *Jdv = nullptr, *vfoot = nullptr; // [B][NV NV], [NV], [3 NF][NV], [3 NF], [3 NF]
    double *com = nullptr, *footp = nullptr;                                               // [B][3], [3 NF] world frame
    // flat feet (FS == 6): J / Jdv / vfoot hold the LOCAL 6-D rows (6 NF); footR [B][NF][9] foot rotations; quad [NF][4][3] corners of the soles;
    // tf / f are wrenches (6 NF)
    double *footR = nullptr, *quad = nullptr;
    double *H = nullptr, *g = nullptr, *C = nullptr, *l = nullptr, *u = nullptr;           // [B][NP NP], [NP], [MP][NP], [MP], [MP]
    double *x = nullptr, *z = nullptr, *lam = nullptr, *rho = nullptr;                     // ADMM iterate [B][NP], [MP], [MP] and step-size parameter [B]
    int * warm = nullptr;                                                                   // [B] 0 = start from scratch
    double *tx = nullptr, *ta = nullptr, *tf = nullptr;                                    // targets [B][NQ + NV] (q | v, the layout of an MPC state), [NV], [3 NF]
    unsigned * tmask = nullptr;                                                             // [B]
    double *tcom = nullptr, *tvcom = nullptr, *tfp = nullptr, *tfv = nullptr;              // CentroidalID targets [B][3], [3], [3 NF], [3 NF]
    double *tau = nullptr, *a = nullptr, *f = nullptr, *resid = nullptr;                   // [B][NA], [NV], [3 NF], [B]
    double *tau_max = nullptr, *v_max = nullptr, *q_min = nullptr, *q_max = nullptr;       // [NA] each
    IdSettingsDev s;
  };

  // ---- kernel 1: rigid-body quantities ----
  template <class D>
  SMPC_DEV void id_quant_body(const IdBuffers<D> & b, int block)
  {
    typedef FullScratch<D, false> SC;
    constexpr int NT = 64, NV = D::NV, NX = D::NX, NF = D::NF, NQ = D::NQ, NR = SC::NR;
    const int inst = block;
    const DevModel<D> & mg = *b.model;
    SMPC_LDS(SC, scs, 1);
    SC & sc = scs[0];
    SMPC_LANES(NT)
    {
      full_load_head<D, NT>(sc.h, &mg, lane);
      for (int i = lane; i < NX; i += NT)
        sc.x[i] = b.X[(size_t)inst * NX + i];
      for (int i = lane; i < D::NU; i += NT)
        sc.u[i] = 0.0;
    }
    SMPC_LANES_END_WAVE
    FullProf fp;
    fp.prof = nullptr;
    fp.tprev = 0;
    const unsigned mask = (1u << NF) - 1u; // every foot's rows (the QP decides which are contacts)
    full_dynamics_phases<D, false>(sc, (FullScratchDeriv<D> *)nullptr, mg, mask, true, fp, true);
    if constexpr (D::FS == 6)
    {
      // flat feet: the stage kernel's rows are LOCAL_WORLD_ALIGNED at the frame origin; tsid's Contact6d works in the foot's LOCAL frame:
      // [R^T lin ; R^T ang] of the Jacobian, of the drift (classical acceleration at zero joint accelerations) and of the frame velocity
      SMPC_LANES(NT)
      {
        for (int idx = lane; idx < NV * NV; idx += NT)
          b.Mq[(size_t)inst * NV * NV + idx] = sc.M[idx];
        for (int i = lane; i < NV; i += NT)
          b.nle[(size_t)inst * NV + i] = -sc.W[i * NR];
        for (int idx = lane; idx < 6 * NF * NV; idx += NT)
        {
          const int r = idx / NV, k = idx % NV, f = r / 6, blk = (r % 6) / 3, i = r % 3;
          const double * Rf = &sc.oR[sc.h.foot_joint[f] * 9];
          const int r0 = 6 * f + 3 * blk;
          b.J[(size_t)inst * 6 * NF * NV + idx] = Rf[i] * sc.J[r0 * NV + k] + Rf[3 + i] * sc.J[(r0 + 1) * NV + k] + Rf[6 + i] * sc.J[(r0 + 2) * NV + k];
        }
        if (lane < 6 * NF)
        {
          const int f = lane / 6, blk = (lane % 6) / 3, i = lane % 3;
          const int jf = sc.h.foot_joint[f];
          const double * Rf = &sc.oR[jf * 9];
          const int r0 = 6 * f + 3 * blk;
          b.Jdv[(size_t)inst * 6 * NF + lane] = Rf[i] * sc.gam[r0] + Rf[3 + i] * sc.gam[r0 + 1] + Rf[6 + i] * sc.gam[r0 + 2];
          const SV v = ldsv(&sc.vel[jf * 6]);
          const V3 w = blk == 0 ? v.l + cross(v.a, ld3(&sc.footp[f * 3])) : v.a;
          b.vfoot[(size_t)inst * 6 * NF + lane] = Rf[i] * w.x + Rf[3 + i] * w.y + Rf[6 + i] * w.z;
        }
        if (lane >= 32 && lane < 32 + 3 * NF)
          b.footp[(size_t)inst * 3 * NF + lane - 32] = sc.footp[lane - 32];
        for (int i = lane; i < 9 * NF; i += NT)
          b.footR[(size_t)inst * 9 * NF + i] = sc.oR[sc.h.foot_joint[i / 9] * 9 + i % 9];
        if (lane < 3)
          b.com[(size_t)inst * 3 + lane] = sc.com[lane];
      }
      SMPC_LANES_END_WAVE
      return;
    }
    // M, nle = -(S tau - nle) at tau = 0 ; LOCAL rows -> world frame: J_w = R_f J_loc, drift likewise (Kp = Kd = 0 in this model)
    SMPC_LANES(NT)
    {
      for (int idx = lane; idx < NV * NV; idx += NT)
        b.Mq[(size_t)inst * NV * NV + idx] = sc.M[idx];
      for (int i = lane; i < NV; i += NT)
        b.nle[(size_t)inst * NV + i] = -sc.W[i * NR];
      for (int idx = lane; idx < 3 * NF * NV; idx += NT)
      {
        const int r = idx / NV, k = idx % NV, f = r / 3, i = r % 3;
        const double * Rf = &sc.oR[sc.h.foot_joint[f] * 9];
        b.J[(size_t)inst * 3 * NF * NV + idx] =
          Rf[i * 3] * sc.J[(3 * f) * NV + k] + Rf[i * 3 + 1] * sc.J[(3 * f + 1) * NV + k] + Rf[i * 3 + 2] * sc.J[(3 * f + 2) * NV + k];
      }
      if (lane < 3 * NF)
      {
        const int f = lane / 3, i = lane % 3;
        const double * Rf = &sc.oR[sc.h.foot_joint[f] * 9];
        b.Jdv[(size_t)inst * 3 * NF + lane] = Rf[i * 3] * sc.gam[3 * f] + Rf[i * 3 + 1] * sc.gam[3 * f + 1] + Rf[i * 3 + 2] * sc.gam[3 * f + 2];
        // velocity of the foot point: v_joint.lin + w x p (world frame)
        const int jf = sc.h.foot_joint[f];
        const SV v = ldsv(&sc.vel[jf * 6]);
        const V3 vp = v.l + cross(v.a, ld3(&sc.footp[f * 3]));
        b.vfoot[(size_t)inst * 3 * NF + lane] = i == 0 ? vp.x : (i == 1 ? vp.y : vp.z);
        b.footp[(size_t)inst * 3 * NF + lane] = sc.footp[lane];
      }
      if (lane < 3)
        b.com[(size_t)inst * 3 + lane] = sc.com[lane];
    }
    SMPC_LANES_END_WAVE
  }

  // acceleration bounds of one joint as TSID's TaskJointPosVelAccBounds::computeAccLimits forms them with position, velocity and
  // viability bounds imposed and the default acceleration limit ([UPSTREAM-RECALL] tsid 1.9; the CPU checker holds the same statement)
  SMPC_HD void id_tsid_acc_limits(double q, double dq, double qmin, double qmax, double dqmax, double control_dt, double & lb, double & ub)
  {
    const double dt = 2.0 * control_dt, ddqmax = 1e10, big = 1e10;
    const double two_dt_sq = 2.0 / (dt * dt);
    const double max_q3 = two_dt_sq * (qmax - q - dt * dq), min_q3 = two_dt_sq * (qmin - q - dt * dq), mdq = -dq / dt;
    double lbp = -big, ubp = big;
    if (dq <= 0.0)
    {
      ubp = max_q3;
      if (min_q3 < mdq)
        lbp = min_q3;
      else if (q != qmin)
        lbp = fmax(dq * dq / (2.0 * (q - qmin)), mdq);
      else
        lbp = 1e6;
    }
    else
    {
      lbp = min_q3;
      if (max_q3 > mdq)
        ubp = max_q3;
      else if (q != qmax)
        ubp = fmin(-dq * dq / (2.0 * (qmax - q)), mdq);
      else
        ubp = -1e6;
    }
    const double lbv = (-dqmax - dq) / dt, ubv = (dqmax - dq) / dt;
    const double dt_sq = dt * dt, dt_dq = dt * dq, two_a = 2.0 * dt_sq, q_plus = q + dt_dq;
    const double b1 = 2.0 * dt_dq + ddqmax * dt_sq, b2 = 2.0 * dt_dq - ddqmax * dt_sq;
    const double c1 = dq * dq - 2.0 * ddqmax * (qmax - q_plus), c2 = dq * dq - 2.0 * ddqmax * (q_plus - qmin);
    double ddq1 = mdq, ddq2 = mdq;
    const double d1 = b1 * b1 - 2.0 * two_a * c1, d2 = b2 * b2 - 2.0 * two_a * c2;
    if (d1 >= 0.0)
      ddq1 = (-b1 + sqrt(d1)) / two_a;
    if (d2 >= 0.0)
      ddq2 = (-b2 - sqrt(d2)) / two_a;
    const double ubvia = fmax(ddq1, mdq), lbvia = fmin(ddq2, mdq);
    ub = fmin(ubp, fmin(ubvia, ubv));
    lb = fmax(lbp, fmax(lbvia, lbv));
    if (ub < lb)
    {
      if (ub == ubp)
        lb = lbp;
      else
        ub = ubp;
      if (ub < lb)
        lb = ub = fmin(lb, ub);
    }
  }

  // ---- kernel 2: QP data ----
  template <class D>
  SMPC_DEV void id_assemble_body(const IdBuffers<D> & b, int block)
  {
    typedef IdDims<D> G;
    constexpr int NT = 64, NV = G::NV, NQ = G::NQ, NF = G::NF, NA = G::NA, N = G::N, NP = G::NP, MP = G::MP;
    const int inst = block;
    const IdSettingsDev & s = b.s;
    const double * x = b.X + (size_t)inst * G::NX;
    const double * q = x;
    const double * v = x + NQ;
    // (the inertia matrix and the foot Jacobians are read tens of times per entry of H: staged in LDS)
    SMPC_LDS(double, sM, NV * NV);
    SMPC_LDS(double, sJ, 3 * NF * NV);
    SMPC_LANES(NT)
    {
      for (int idx = lane; idx < NV * NV; idx += NT)
        sM[idx] = b.Mq[(size_t)inst * NV * NV + idx];
      for (int idx = lane; idx < 3 * NF * NV; idx += NT)
        sJ[idx] = b.J[(size_t)inst * 3 * NF * NV + idx];
    }
    SMPC_LANES_END_WAVE
    const double * Mq = sM;
    const double * nle = b.nle + (size_t)inst * NV;
    const double * J = sJ;
    const double * Jdv = b.Jdv + (size_t)inst * 3 * NF;
    const double * vf = b.vfoot + (size_t)inst * 3 * NF;
    const double *tq = b.tx + (size_t)inst * G::NX, *tv = tq + NQ, *ta = b.ta + (size_t)inst * NV, *tf = b.tf + (size_t)inst * 3 * NF;
    const unsigned mask = b.tmask[inst];
    double * H = b.H + (size_t)inst * NP * NP;
    double * g = b.g + (size_t)inst * NP;
    double * C = b.C + (size_t)inst * MP * NP;
    double * l = b.l + (size_t)inst * MP;
    double * u = b.u + (size_t)inst * MP;
    const double kdp = 2.0 * sqrt(s.kp_posture), kdb = 2.0 * sqrt(s.kp_base), kdc = 2.0 * sqrt(s.kp_contact);
    const double kdm = 2.0 * sqrt(s.kp_com), kdt = 2.0 * sqrt(s.kp_feet_tracking);
    const bool com_task = s.centroidal && s.w_com > 0, track_task = s.centroidal && s.w_feet_tracking > 0;
    const int base0 = s.centroidal ? 3 : 0; // (CentroidalID: orientation rows only, centroidal-id.cpp:10-20)
    SMPC_LDS(double, e6, 6);
    SMPC_LDS(double, Jc, 3 * NV); // CoM Jacobian R_b M_lin / m
    SMPC_LDS(double, bc, 3);      // right-hand side of the CoM task
    SMPC_LDS(double, bt, 3 * NF); // right-hand sides of the foot-tracking tasks
    SMPC_LANES(NT)
    if (com_task)
      for (int idx = lane; idx < 3 * NV; idx += NT)
      {
        const int i = idx / NV, k = idx % NV;
        const M3 Rb = quat_to_R(Quat{q[3], q[4], q[5], q[6]});
        const double im = 1.0 / b.model->total_mass;
        const double r0 = i == 0 ? Rb.a00 : (i == 1 ? Rb.a10 : Rb.a20), r1 = i == 0 ? Rb.a01 : (i == 1 ? Rb.a11 : Rb.a21),
                     r2 = i == 0 ? Rb.a02 : (i == 1 ? Rb.a12 : Rb.a22);
        Jc[idx] = im * (r0 * Mq[k] + r1 * Mq[NV + k] + r2 * Mq[2 * NV + k]);
      }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      if (com_task && lane < 3)
      { // a_com = J_com a + drift, drift = R_b nle_lin / m + g (TaskComEquality, centroidal-id.cpp:22-27)
        const int i = lane;
        const M3 Rb = quat_to_R(Quat{q[3], q[4], q[5], q[6]});
        const double im = 1.0 / b.model->total_mass;
        const double r0 = i == 0 ? Rb.a00 : (i == 1 ? Rb.a10 : Rb.a20), r1 = i == 0 ? Rb.a01 : (i == 1 ? Rb.a11 : Rb.a21),
                     r2 = i == 0 ? Rb.a02 : (i == 1 ? Rb.a12 : Rb.a22);
        double vc = 0.0;
        for (int k = 0; k < NV; k++)
          vc += Jc[i * NV + k] * v[k];
        const double dr = im * (r0 * nle[0] + r1 * nle[1] + r2 * nle[2]) + (i == 2 ? -9.81 : 0.0);
        bc[i] = s.kp_com * (b.tcom[(size_t)inst * 3 + i] - b.com[(size_t)inst * 3 + i]) + kdm * (b.tvcom[(size_t)inst * 3 + i] - vc) - dr;
      }
      if (track_task && lane >= 32 && lane < 32 + 3 * NF)
      { // position tracking of the feet out of contact (centroidal-id.cpp:101-129; point feet: linear part)
        const int r = lane - 32;
        const size_t o = (size_t)inst * 3 * NF + r;
        bt[r] = s.kp_feet_tracking * (b.tfp[o] - b.footp[o]) + kdt * (b.tfv[o] - vf[r]) - Jdv[r];
      }
    }
    SMPC_LANES_END_WAVE
    static_assert(3 * NF <= 32, "lane map of the task right-hand sides");
    SMPC_LANES(NT)
    if (lane == 0)
    { // base error log6(M_b^-1 M_t), local frame
      const SE3 Mb{quat_to_R(Quat{q[3], q[4], q[5], q[6]}), mk3(q[0], q[1], q[2])};
      const SE3 Mt{quat_to_R(Quat{tq[3], tq[4], tq[5], tq[6]}), mk3(tq[0], tq[1], tq[2])};
      V3 ev, ew;
      log6(se3_mul(se3_inv(Mb), Mt), ev, ew);
      st3(e6, ev);
      st3(e6 + 3, ew);
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      // ---- H (N x N, padded with unit diagonal) and g ----
      for (int idx = lane; idx < NP * NP; idx += NT)
      {
        const int i = idx / NP, j = idx % NP;
        double h = 0.0;
        if (i >= N || j >= N)
          h = (i == j) ? 1.0 : 0.0; // padding variables: pinned by their own unit curvature and zero gradient
        else
        {
          if (i == j && i >= 6 && i < NV && s.w_posture > 0)
            h += s.w_posture;
          if (i == j && i >= base0 && i < 6 && s.w_base > 0)
            h += s.w_base;
          if (i < NV && j < NV && com_task)
            for (int r = 0; r < 3; r++)
              h += s.w_com * Jc[r * NV + i] * Jc[r * NV + j];
          if (i < NV && j < NV && track_task)
            for (int r = 0; r < 3 * NF; r++)
              if (!((mask >> (r / 3)) & 1u))
                h += s.w_feet_tracking * J[r * NV + i] * J[r * NV + j];
          if (i < NV && j < NV && !s.contact_motion_equality && s.w_contact_motion > 0)
            for (int r = 0; r < 3 * NF; r++)
              if ((mask >> (r / 3)) & 1u)
                h += s.w_contact_motion * J[r * NV + i] * J[r * NV + j];
          if (i == j && i >= NV && s.w_contact_force > 0 && ((mask >> ((i - NV) / 3)) & 1u))
            h += s.w_contact_force;
        }
        H[idx] = h;
      }
      for (int i = lane; i < NP; i += NT)
      {
        double gi = 0.0;
        if (i < N)
        {
          if (i >= 6 && i < NV && s.w_posture > 0)
            gi -= s.w_posture * (ta[i] + s.kp_posture * (tq[i + 1] - q[i + 1]) + kdp * (tv[i] - v[i]));
          if (i >= base0 && i < 6 && s.w_base > 0)
          {
            const V3 dr = cross(mk3(v[3], v[4], v[5]), mk3(v[0], v[1], v[2]));
            // (velocity / acceleration references: DESIGN 3.12; base_as_coded: the reference literally, kinodynamics-id.cpp:222-223)
            const double ades = s.base_as_coded ? s.kp_base * e6[i] + kdb * (ta[i] - v[i]) : s.kp_base * e6[i] + kdb * (tv[i] - v[i]) + ta[i];
            gi -= s.w_base * (ades - (i == 0 ? dr.x : (i == 1 ? dr.y : (i == 2 ? dr.z : 0.0))));
          }
          if (i < NV && com_task)
            for (int r = 0; r < 3; r++)
              gi -= s.w_com * Jc[r * NV + i] * bc[r];
          if (i < NV && track_task)
            for (int r = 0; r < 3 * NF; r++)
              if (!((mask >> (r / 3)) & 1u))
                gi -= s.w_feet_tracking * J[r * NV + i] * bt[r];
          if (i < NV && !s.contact_motion_equality && s.w_contact_motion > 0)
            for (int r = 0; r < 3 * NF; r++)
              if ((mask >> (r / 3)) & 1u)
                gi -= s.w_contact_motion * J[r * NV + i] * (-Jdv[r] - kdc * vf[r]);
          if (i >= NV && s.w_contact_force > 0 && ((mask >> ((i - NV) / 3)) & 1u))
            gi -= s.w_contact_force * tf[i - NV];
        }
        g[i] = gi;
      }
      // ---- C, l, u ----  (rows 0 .. N-1, the box on y, are the identity and the padding rows are zero: written once when the engine
      //                     is created; only the general rows change with the state)
      for (int idx = N * NP + lane; idx < G::M * NP; idx += NT)
      {
        const int r = idx / NP, c = idx % NP;
        double val = 0.0;
        if (r < N)
          val = (r == c) ? 1.0 : 0.0;
        else if (c < N && r < G::R_MOT)
        { // dynamics rows: [M_b | -J_b^T]
          const int i = r - G::R_DYN;
          val = c < NV ? Mq[i * NV + c] : -J[(c - NV) * NV + i];
        }
        else if (c < N && r < G::R_FRI)
        { // contact motion rows (equality variant, feet in contact)
          const int rr = r - G::R_MOT;
          if (s.contact_motion_equality && ((mask >> (rr / 3)) & 1u) && c < NV)
            val = J[rr * NV + c];
        }
        else if (c < N && r < G::R_ACT)
        { // friction pyramid: +-f_x - mu f_z, +-f_y - mu f_z
          const int rr = r - G::R_FRI, f = rr / 4, k = rr % 4;
          if ((mask >> f) & 1u)
          {
            if (c == NV + 3 * f + k / 2)
              val = (k % 2 == 0) ? 1.0 : -1.0;
            else if (c == NV + 3 * f + 2)
              val = -s.friction_coefficient;
          }
        }
        else if (c < N && r < G::M)
        { // actuation rows: [M_a | -J_a^T]
          const int j = r - G::R_ACT;
          val = c < NV ? Mq[(6 + j) * NV + c] : -J[(c - NV) * NV + 6 + j];
        }
        C[idx] = val;
      }
      for (int r = lane; r < MP; r += NT)
      {
        double lo = -ID_INF, hi = ID_INF;
        const double W = b.model->total_mass * 9.81, dt = s.control_dt;
        if (r >= 6 && r < NV)
        { // joint limits as acceleration bounds over one control period
          const int j = r - 6;
          const double qa = q[7 + j], va = v[6 + j];
          lo = fmax((-b.v_max[j] - va) / dt, 2.0 * (b.q_min[j] - qa - va * dt) / (dt * dt));
          hi = fmin((b.v_max[j] - va) / dt, 2.0 * (b.q_max[j] - qa - va * dt) / (dt * dt));
          if (lo > hi)
            lo = hi = fmin(lo, hi);
          if (s.tsid_bounds)
            id_tsid_acc_limits(qa, va, b.q_min[j], b.q_max[j], b.v_max[j], dt, lo, hi);
        }
        else if (r >= NV && r < N)
        {
          const int f = (r - NV) / 3, i = (r - NV) % 3;
          if (!((mask >> f) & 1u))
            lo = hi = 0.0;
          else if (i == 2)
          {
            lo = s.ratio_min * W;
            hi = s.ratio_max * W;
          }
        }
        else if (r >= N && r < G::R_MOT)
          lo = hi = -nle[r - G::R_DYN];
        else if (r >= G::R_MOT && r < G::R_FRI)
        {
          const int rr = r - G::R_MOT;
          if (s.contact_motion_equality && ((mask >> (rr / 3)) & 1u))
            lo = hi = -Jdv[rr] - kdc * vf[rr];
        }
        else if (r >= G::R_FRI && r < G::R_ACT)
        {
          if ((mask >> ((r - G::R_FRI) / 4)) & 1u)
            hi = 0.0;
        }
        else if (r >= G::R_ACT && r < G::M)
        {
          const int j = r - G::R_ACT;
          lo = -b.tau_max[j] - nle[6 + j];
          hi = b.tau_max[j] - nle[6 + j];
        }
        l[r] = lo;
        u[r] = hi;
      }
    }
    SMPC_LANES_END_WAVE
  }

  // ---- kernel 3: ADMM ----
  // One wavefront per robot; the iteration runs out of registers.  The first N rows of C are the identity (box on y), so only the
  // GR = M - N general rows are stored: lane k < GR holds row k (NP doubles), lane i < NP holds column i (GR doubles) and row i of
  // K^-1 (NP doubles); the vectors live one element per lane and are broadcast with v_readlane (SMPC_XLANE): no LDS traffic in the
  // loop, so the four SIMDs of a CU iterate independently instead of queueing on the CU's one LDS pipe.
  template <class D>
  struct QpLds
  {
    typedef IdDims<D> G;
    static constexpr int GR = G::M - G::N;
    double K[G::NP * G::NP];   // H + sigma I + C^T diag(r) C -> its inverse
    double C[GR * G::LDC];     // general rows, row stride NP + 1
    double rg[GR];
    double swp[2 * 4 * 16 * ((2 * G::NP + 15) / 16)];
    double red[256], red4[4];
  };
  constexpr int ADMM_CHECK = 20; // residual check period of the ADMM loop
  constexpr double ADMM_ADAPT_FLOOR = 1e-7; // rho is adapted only while the residuals are above

  // The step-size parameter rho of a robot is kept across solves with the iterate and adapted as OSQP does (Stellato et al. 2020, section
  // 5.2): at a residual check, rho <- rho sqrt((r_prim / max(|Cx|, |z|)) / (r_dual / max(|Hx|, |C^T lam|, |g|))), applied -- with a new
  // inverse of K -- when it moves by more than a factor 5.
  template <class D>
  SMPC_DEV void qp_admm_body(const IdBuffers<D> & b, int block)
  {
    typedef IdDims<D> G;
    constexpr int NT = 64, NV = G::NV, NF = G::NF, NA = G::NA, N = G::N, NP = G::NP, MP = G::MP, LDC = G::LDC, GR = QpLds<D>::GR;
    static_assert(GR <= NT && NP <= NT && N <= NP, "one general row / one variable per lane");
    const int inst = block;
    const IdSettingsDev & st = b.s;
    const double sigma = st.sigma, alpha = st.alpha;
    SMPC_LDS(QpLds<D>, ls, 1);
    QpLds<D> & s = ls[0];
    const double * Hg = b.H + (size_t)inst * NP * NP;
    const double * Cg = b.C + (size_t)inst * MP * NP + (size_t)N * NP; // general rows
    const bool warm = b.warm[inst] != 0;
    double rho = warm ? b.rho[inst] : st.rho;
    // per-lane state: variable `lane` (< NP), box row `lane` (< N), general row `lane` (< GR)
    SMPC_PL(double, x, NT);
    SMPC_PL(double, g, NT);
    SMPC_PL(double, rhs, NT);
    SMPC_PL(double, xt, NT);
    SMPC_PL(double, zb, NT);
    SMPC_PL(double, lamb, NT);
    SMPC_PL(double, lb, NT);
    SMPC_PL(double, ub, NT);
    SMPC_PL(double, rb, NT);
    SMPC_PL(double, zg, NT);
    SMPC_PL(double, lamg, NT);
    SMPC_PL(double, lg, NT);
    SMPC_PL(double, ug, NT);
    SMPC_PL(double, rg, NT);
    SMPC_PL(double, wg, NT);
    SMPC_PL(double, ztg, NT);
    SMPC_PLA(double, Ccol, NT, GR);
    SMPC_PLA(double, Crow, NT, NP);
    SMPC_PLA(double, Krow, NT, NP);
    SMPC_LANES(NT)
    {
      // rows and columns of C straight from HBM into the registers they stay in (all loads independent: in flight together); the
      // LDS copy the matrix cores read K's operands from is written from the column registers
      const int i = lane < NP ? lane : 0, kb = lane < N ? lane : 0, kg = N + (lane < GR ? lane : 0);
#pragma unroll
      for (int kk = 0; kk < GR; kk++)
        SMPC_PLV(Ccol)[kk] = Cg[kk * NP + i];
#pragma unroll
      for (int ii = 0; ii < NP; ii++)
        SMPC_PLV(Crow)[ii] = Cg[(kg - N) * NP + ii];
      if (lane < NP)
      {
#pragma unroll
        for (int kk = 0; kk < GR; kk++)
          s.C[kk * LDC + lane] = SMPC_PLV(Ccol)[kk];
      }
      SMPC_PLV(g) = b.g[(size_t)inst * NP + i];
      SMPC_PLV(x) = warm ? b.x[(size_t)inst * NP + i] : 0.0;
      SMPC_PLV(rhs) = SMPC_PLV(xt) = 0.0;
      {
        const double lo = b.l[(size_t)inst * MP + kb], hi = b.u[(size_t)inst * MP + kb];
        SMPC_PLV(lb) = lo;
        SMPC_PLV(ub) = hi;
        SMPC_PLV(zb) = warm ? b.z[(size_t)inst * MP + kb] : fmin(fmax(0.0, lo), hi);
        SMPC_PLV(lamb) = warm ? b.lam[(size_t)inst * MP + kb] : 0.0;
      }
      {
        const double lo = b.l[(size_t)inst * MP + kg], hi = b.u[(size_t)inst * MP + kg];
        SMPC_PLV(lg) = lo;
        SMPC_PLV(ug) = hi;
        SMPC_PLV(zg) = warm ? b.z[(size_t)inst * MP + kg] : fmin(fmax(0.0, lo), hi);
        SMPC_PLV(lamg) = warm ? b.lam[(size_t)inst * MP + kg] : 0.0;
      }
      SMPC_PLV(wg) = SMPC_PLV(ztg) = 0.0;
    }
    SMPC_LANES_END_WAVE
    // row weights r = rho (1e3 rho on equality rows, 1e-6 rho on free rows) ; K = H + sigma I + C^T diag(r) C -> its inverse -> rows in registers
    auto factor = [&]() {
      SMPC_LANES(NT)
      {
        auto weight = [&](double lo, double hi) { return (hi - lo < 1e-12) ? 1e3 * rho : ((lo <= -ID_INF && hi >= ID_INF) ? 1e-6 * rho : rho); };
        SMPC_PLV(rb) = weight(SMPC_PLV(lb), SMPC_PLV(ub));
        SMPC_PLV(rg) = weight(SMPC_PLV(lg), SMPC_PLV(ug));
        if (lane < GR)
          s.rg[lane] = SMPC_PLV(rg);
        if (lane < N)
          s.red[lane] = SMPC_PLV(rb);
      }
      SMPC_LANES_END_WAVE
      fwave_gemm<NP, NP, GR>(
        [&](int i, int k) { return s.rg[k] * s.C[k * LDC + i]; }, [&](int k, int j) { return s.C[k * LDC + j]; },
        [&](int i, int j, double v) { s.K[i * NP + j] = (Hg[i * NP + j] + (i == j ? sigma + (i < N ? s.red[i] : 0.0) : 0.0)) + v; });
      fwave_spd_inverse<NP>(s.K, s.swp);
      SMPC_LANES(NT)
      {
        const int i = lane < NP ? lane : 0;
#pragma unroll
        for (int j = 0; j < NP; j++)
          SMPC_PLV(Krow)[j] = s.K[j * NP + i]; // (K^-1 is symmetric: read along the row of j, conflict-free)
      }
      SMPC_LANES_END_WAVE
    };
    // residuals of the iterate and the norms they are measured against (the same values in every lane):
    //   rs[0] = |C x - z|_inf, rs[1] = |H x + g + C^T lam|_inf, rs[2] = max(|C x|, |z|)_inf, rs[3] = max(|H x|, |C^T lam|, |g|)_inf
    double rs[4] = {0.0, 0.0, 0.0, 0.0};
    auto residual = [&]() {
      SMPC_LANES(NT)
      {
        double pr = 0.0, np = 0.0, du = 0.0, nd = 0.0;
        double cx = 0.0, hx = 0.0, cl = SMPC_PLV(lamb);
#pragma unroll
        for (int i = 0; i < N; i++)
        {
          cx += SMPC_PLV(Crow)[i] * SMPC_XLANE(x, i);
          if (i % 8 == 7)
            SMPC_SCHED_FENCE();
        }
        for (int j = 0; j < N; j++)
          hx += Hg[j * NP + (lane < NP ? lane : 0)] * SMPC_XLANE(x, j); // (H is symmetric: coalesced along the row of j)
#pragma unroll
        for (int k = 0; k < GR; k++)
        {
          cl += SMPC_PLV(Ccol)[k] * SMPC_XLANE(lamg, k);
          if (k % 8 == 7)
            SMPC_SCHED_FENCE();
        }
        if (lane < GR)
        {
          pr = fabs(cx - SMPC_PLV(zg));
          np = fmax(fabs(cx), fabs(SMPC_PLV(zg)));
        }
        if (lane < N)
        { // box rows: C x = x
          pr = fmax(pr, fabs(SMPC_PLV(x) - SMPC_PLV(zb)));
          np = fmax(np, fmax(fabs(SMPC_PLV(x)), fabs(SMPC_PLV(zb))));
          du = fabs((SMPC_PLV(g) + hx) + cl);
          nd = fmax(fabs(hx), fmax(fabs(cl), fabs(SMPC_PLV(g))));
        }
        s.red[lane] = pr;
        s.red[64 + lane] = du;
        s.red[128 + lane] = np;
        s.red[192 + lane] = nd;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane < 4)
      {
        double m = 0.0; // (a NaN entry must survive the reduction: fmax would drop it and a failed solve would look converged)
        for (int i = 0; i < NT; i++)
        {
          const double v = s.red[64 * lane + i];
          m = (v != v) ? v : ((m != m) ? m : fmax(m, v));
        }
        s.red4[lane] = m;
      }
      SMPC_LANES_END_WAVE
      for (int i = 0; i < 4; i++)
        rs[i] = s.red4[i];
    };
    factor();
    bool done = false;
    for (int it = 0; it < st.admm_iters; it++)
    {
      if (it > 0 && it % ADMM_CHECK == 0)
      {
        residual();
        if (st.admm_tol >= 0.0 && fmax(rs[0], rs[1]) <= st.admm_tol)
        {
          done = true;
          break;
        }
        const double est = fmin(fmax(rho * sqrt((rs[0] / (rs[2] + 1e-10)) / (rs[1] / (rs[3] + 1e-10) + 1e-10)), 1e-6), 1e6);
        if (fmax(rs[0], rs[1]) > ADMM_ADAPT_FLOOR && (est > 5.0 * rho || est < 0.2 * rho)) // (below the floor the ratio is rounding noise)
        {
          rho = est;
          factor();
        }
      }
      SMPC_LANES(NT)
      SMPC_PLV(wg) = SMPC_PLV(rg) * SMPC_PLV(zg) - SMPC_PLV(lamg);
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      { // rhs = sigma x - g + C^T (r z - lam): the box rows contribute their own entry
        double acc = sigma * SMPC_PLV(x) - SMPC_PLV(g);
        acc += lane < N ? SMPC_PLV(rb) * SMPC_PLV(zb) - SMPC_PLV(lamb) : 0.0; // (a select, not a branch: the cross-lane reads below stay in this block)
#pragma unroll
        for (int k = 0; k < GR; k++)
        {
          acc += SMPC_PLV(Ccol)[k] * SMPC_XLANE(wg, k);
          if (k % 8 == 7)
            SMPC_SCHED_FENCE();
        }
        SMPC_PLV(rhs) = acc;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < N; j++) // (the padding variables are decoupled: their rows and columns of K^-1 are the identity's, their rhs is 0)
        {
          acc += SMPC_PLV(Krow)[j] * SMPC_XLANE(rhs, j);
          if (j % 8 == 7)
            SMPC_SCHED_FENCE();
        }
        SMPC_PLV(xt) = acc;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < N; i++) // (columns N .. NP - 1 of C are zero)
        {
          acc += SMPC_PLV(Crow)[i] * SMPC_XLANE(xt, i);
          if (i % 8 == 7)
            SMPC_SCHED_FENCE();
        }
        SMPC_PLV(ztg) = acc;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        { // box rows: z~ = x~
          const double zh = alpha * SMPC_PLV(xt) + (1.0 - alpha) * SMPC_PLV(zb);
          const double zn = fmin(fmax(zh + SMPC_PLV(lamb) / SMPC_PLV(rb), SMPC_PLV(lb)), SMPC_PLV(ub));
          SMPC_PLV(lamb) += SMPC_PLV(rb) * (zh - zn);
          SMPC_PLV(zb) = zn;
        }
        {
          const double zh = alpha * SMPC_PLV(ztg) + (1.0 - alpha) * SMPC_PLV(zg);
          const double zn = fmin(fmax(zh + SMPC_PLV(lamg) / SMPC_PLV(rg), SMPC_PLV(lg)), SMPC_PLV(ug));
          SMPC_PLV(lamg) += SMPC_PLV(rg) * (zh - zn);
          SMPC_PLV(zg) = zn;
        }
        SMPC_PLV(x) = alpha * SMPC_PLV(xt) + (1.0 - alpha) * SMPC_PLV(x);
      }
      SMPC_LANES_END_WAVE
    }
    if (!done)
      residual();
    const double res = (rs[0] != rs[0] || rs[1] != rs[1]) ? rs[0] + rs[1] : fmax(rs[0], rs[1]);
    // the solution through LDS for the torque rows (red is free now)
    SMPC_LANES(NT)
    if (lane < NP)
      s.red[lane] = SMPC_PLV(x);
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      // the iterate is kept as the next tick's warm start only when the solve ended finite: a NaN state or target (a diverged simulator,
      // an MPC instance with a non-zero status) must not disable this robot's controller for good
      const bool ok = res == res && res < 1e300;
      if (ok && lane < NP)
        b.x[(size_t)inst * NP + lane] = SMPC_PLV(x);
      if (ok && lane < N)
      {
        b.z[(size_t)inst * MP + lane] = SMPC_PLV(zb);
        b.lam[(size_t)inst * MP + lane] = SMPC_PLV(lamb);
      }
      if (ok && lane < GR)
      {
        b.z[(size_t)inst * MP + N + lane] = SMPC_PLV(zg);
        b.lam[(size_t)inst * MP + N + lane] = SMPC_PLV(lamg);
      }
      for (int i = lane; i < NV; i += NT)
        b.a[(size_t)inst * NV + i] = s.red[i];
      for (int i = lane; i < 3 * NF; i += NT)
        b.f[(size_t)inst * 3 * NF + i] = s.red[NV + i];
      if (lane < NA)
      { // tau = M_a a + h_a - J_a^T f
        const double * Mq = b.Mq + (size_t)inst * NV * NV;
        const double * J = b.J + (size_t)inst * 3 * NF * NV;
        double acc = b.nle[(size_t)inst * NV + 6 + lane];
        for (int k = 0; k < NV; k++)
          acc += Mq[(6 + lane) * NV + k] * s.red[k];
        for (int r = 0; r < 3 * NF; r++)
          acc -= J[r * NV + 6 + lane] * s.red[NV + r];
        b.tau[(size_t)inst * NA + lane] = acc;
      }
      if (lane == 0)
      {
        b.resid[inst] = res;
        b.rho[inst] = ok ? rho : st.rho;
        b.warm[inst] = ok ? 1 : 0;
      }
    }
    SMPC_LANES_END_WAVE
  }

  // =====================================================================================================================
  // Flat feet (tsid::contacts::Contact6d; reference kinodynamics-id.cpp:41-49, 163-167, 204-208; centroidal-id.cpp:38-41): QP data and ADMM of the
  // 52-variable / 74-general-row problem of a Talos-class biped.  Formulation: DESIGN.md, "Robots with 6-D feet".
  // =====================================================================================================================
  SMPC_HD double id6_tgen(const double * quad, int f, int r, int c)
  { // force generator matrix T (6 x 12) of foot f: [I I I I ; [p_1]x .. [p_4]x]
    const int k = c / 3, j = c % 3;
    if (r < 3)
      return r == j ? 1.0 : 0.0;
    const double * p = quad + (f * 4 + k) * 3;
    const int i = r - 3;
    if (i == j)
      return 0.0;
    const int o = 3 - i - j; // the remaining axis: [p]x (i, j) = +-p_o
    const bool pos = (i == 0 && j == 2) || (i == 1 && j == 0) || (i == 2 && j == 1);
    return pos ? p[o] : -p[o];
  }
  SMPC_HD double id6_wrench_w(int r) { return r < 2 ? 1.0 : (r == 2 ? 1e-3 : 2.0); } // Contact6d::m_weightForceRegTask [UPSTREAM-RECALL]

  template <class D>
  SMPC_DEV void id6_assemble_body(const IdBuffers<D> & b, int block)
  {
    typedef IdDims<D> G;
    constexpr int NT = 64, NV = G::NV, NQ = G::NQ, NF = G::NF, NA = G::NA, N = G::N, NP = G::NP, MP = G::MP, NFV = G::NFV;
    static_assert(G::FS == 6, "flat feet");
    const int inst = block;
    const IdSettingsDev & s = b.s;
    const double * x = b.X + (size_t)inst * G::NX;
    const double * q = x;
    const double * v = x + NQ;
    SMPC_LDS(double, sM, NV * NV);
    SMPC_LDS(double, sJ, 6 * NF * NV);
    SMPC_LDS(double, JG, NV * NFV * NF); // J^T T per foot: generalised force of the corner forces
    SMPC_LDS(double, Jc, 3 * NV);
    SMPC_LDS(double, bc, 3);
    SMPC_LDS(double, bt, 6 * NF);
    SMPC_LDS(double, e6, 6);
    SMPC_LANES(NT)
    {
      for (int idx = lane; idx < NV * NV; idx += NT)
        sM[idx] = b.Mq[(size_t)inst * NV * NV + idx];
      for (int idx = lane; idx < 6 * NF * NV; idx += NT)
        sJ[idx] = b.J[(size_t)inst * 6 * NF * NV + idx];
    }
    SMPC_LANES_END_WAVE
    const double * Mq = sM;
    const double * nle = b.nle + (size_t)inst * NV;
    const double * J = sJ;
    const double * Jdv = b.Jdv + (size_t)inst * 6 * NF;
    const double * vf = b.vfoot + (size_t)inst * 6 * NF;
    const double *tq = b.tx + (size_t)inst * G::NX, *tv = tq + NQ, *ta = b.ta + (size_t)inst * NV, *tf = b.tf + (size_t)inst * 6 * NF;
    const unsigned mask = b.tmask[inst];
    double * H = b.H + (size_t)inst * NP * NP;
    double * g = b.g + (size_t)inst * NP;
    double * C = b.C + (size_t)inst * MP * NP;
    double * l = b.l + (size_t)inst * MP;
    double * u = b.u + (size_t)inst * MP;
    const double kdp = 2.0 * sqrt(s.kp_posture), kdb = 2.0 * sqrt(s.kp_base), kdc = 2.0 * sqrt(s.kp_contact);
    const double kdm = 2.0 * sqrt(s.kp_com), kdt = 2.0 * sqrt(s.kp_feet_tracking);
    const bool com_task = s.centroidal && s.w_com > 0, track_task = s.centroidal && s.w_feet_tracking > 0;
    const bool mot_cost = !s.contact_motion_equality && s.w_contact_motion > 0;
    const int base0 = s.centroidal ? 3 : 0;
    SMPC_LANES(NT)
    {
      for (int idx = lane; idx < NV * NFV * NF; idx += NT)
      {
        const int k = idx / (NFV * NF), c = idx % (NFV * NF), f = c / NFV, cc = c % NFV;
        double acc = 0.0;
        for (int r = 0; r < 6; r++)
          acc += J[(6 * f + r) * NV + k] * id6_tgen(b.quad, f, r, cc);
        JG[idx] = acc;
      }
      if (com_task)
        for (int idx = lane; idx < 3 * NV; idx += NT)
        {
          const int i = idx / NV, k = idx % NV;
          const M3 Rb = quat_to_R(Quat{q[3], q[4], q[5], q[6]});
          const double im = 1.0 / b.model->total_mass;
          const double r0 = i == 0 ? Rb.a00 : (i == 1 ? Rb.a10 : Rb.a20), r1 = i == 0 ? Rb.a01 : (i == 1 ? Rb.a11 : Rb.a21),
                       r2 = i == 0 ? Rb.a02 : (i == 1 ? Rb.a12 : Rb.a22);
          Jc[idx] = im * (r0 * Mq[k] + r1 * Mq[NV + k] + r2 * Mq[2 * NV + k]);
        }
      if (lane == 63)
      { // base error log6(M_b^-1 M_t), local frame
        const SE3 Mb{quat_to_R(Quat{q[3], q[4], q[5], q[6]}), mk3(q[0], q[1], q[2])};
        const SE3 Mt{quat_to_R(Quat{tq[3], tq[4], tq[5], tq[6]}), mk3(tq[0], tq[1], tq[2])};
        V3 ev, ew;
        log6(se3_mul(se3_inv(Mb), Mt), ev, ew);
        st3(e6, ev);
        st3(e6 + 3, ew);
      }
      if (track_task && lane >= 32 && lane < 32 + NF)
      { // feet in the air: 6-D LOCAL task towards (identity rotation, target position), zero angular velocity target
        const int f = lane - 32;
        const size_t o = (size_t)inst * 3 * NF + 3 * f;
        const M3 Rf = ldm3(b.footR + ((size_t)inst * NF + f) * 9);
        const SE3 Mf{Rf, ld3(b.footp + o)}, Mr{m3_id(), ld3(b.tfp + o)};
        V3 ev, ew;
        log6(se3_mul(se3_inv(Mf), Mr), ev, ew);
        const V3 vr = tmul(Rf, ld3(b.tfv + o));
        const double e[6] = {ev.x, ev.y, ev.z, ew.x, ew.y, ew.z}, vrr[6] = {vr.x, vr.y, vr.z, 0.0, 0.0, 0.0};
        for (int i = 0; i < 6; i++)
          bt[6 * f + i] = s.kp_feet_tracking * e[i] + kdt * (vrr[i] - vf[6 * f + i]) - Jdv[6 * f + i];
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (com_task && lane < 3)
    {
      const int i = lane;
      const M3 Rb = quat_to_R(Quat{q[3], q[4], q[5], q[6]});
      const double im = 1.0 / b.model->total_mass;
      const double r0 = i == 0 ? Rb.a00 : (i == 1 ? Rb.a10 : Rb.a20), r1 = i == 0 ? Rb.a01 : (i == 1 ? Rb.a11 : Rb.a21),
                   r2 = i == 0 ? Rb.a02 : (i == 1 ? Rb.a12 : Rb.a22);
      double vc = 0.0;
      for (int k = 0; k < NV; k++)
        vc += Jc[i * NV + k] * v[k];
      const double dr = im * (r0 * nle[0] + r1 * nle[1] + r2 * nle[2]) + (i == 2 ? -9.81 : 0.0);
      bc[i] = s.kp_com * (b.tcom[(size_t)inst * 3 + i] - b.com[(size_t)inst * 3 + i]) + kdm * (b.tvcom[(size_t)inst * 3 + i] - vc) - dr;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      // ---- H and g ----
      for (int idx = lane; idx < NP * NP; idx += NT)
      {
        const int i = idx / NP, j = idx % NP;
        double h = 0.0;
        if (i >= N || j >= N)
          h = (i == j) ? 1.0 : 0.0;
        else
        {
          if (i == j && i >= 6 && i < NV && s.w_posture > 0)
            h += s.w_posture;
          if (i == j && i >= base0 && i < 6 && s.w_base > 0)
            h += s.w_base;
          if (i < NV && j < NV)
          {
            if (com_task)
              for (int r = 0; r < 3; r++)
                h += s.w_com * Jc[r * NV + i] * Jc[r * NV + j];
            for (int f = 0; f < NF; f++)
            {
              const bool on = (mask >> f) & 1u;
              const double w = on ? (mot_cost ? s.w_contact_motion : 0.0) : (track_task ? s.w_feet_tracking : 0.0);
              if (w != 0.0)
                for (int r = 6 * f; r < 6 * f + 6; r++)
                  h += w * J[r * NV + i] * J[r * NV + j];
            }
          }
          if (i >= NV && j >= NV && (i - NV) / NFV == (j - NV) / NFV && s.w_contact_force > 0 && ((mask >> ((i - NV) / NFV)) & 1u))
          { // force regularisation: T^T diag(w^2) T of the foot
            const int f = (i - NV) / NFV, a = (i - NV) % NFV, c = (j - NV) % NFV;
            double acc = 0.0;
            for (int r = 0; r < 6; r++)
              acc += id6_tgen(b.quad, f, r, a) * id6_wrench_w(r) * id6_wrench_w(r) * id6_tgen(b.quad, f, r, c);
            h += s.w_contact_force * acc;
          }
        }
        H[idx] = h;
      }
      for (int i = lane; i < NP; i += NT)
      {
        double gi = 0.0;
        if (i < N)
        {
          if (i >= 6 && i < NV && s.w_posture > 0)
            gi -= s.w_posture * (ta[i] + s.kp_posture * (tq[i + 1] - q[i + 1]) + kdp * (tv[i] - v[i]));
          if (i >= base0 && i < 6 && s.w_base > 0)
          {
            const V3 dr = cross(mk3(v[3], v[4], v[5]), mk3(v[0], v[1], v[2]));
            const double ades = s.base_as_coded ? s.kp_base * e6[i] + kdb * (ta[i] - v[i]) : s.kp_base * e6[i] + kdb * (tv[i] - v[i]) + ta[i];
            gi -= s.w_base * (ades - (i == 0 ? dr.x : (i == 1 ? dr.y : (i == 2 ? dr.z : 0.0))));
          }
          if (i < NV)
          {
            if (com_task)
              for (int r = 0; r < 3; r++)
                gi -= s.w_com * Jc[r * NV + i] * bc[r];
            for (int f = 0; f < NF; f++)
            {
              const bool on = (mask >> f) & 1u;
              if (on && mot_cost)
                for (int r = 6 * f; r < 6 * f + 6; r++)
                  gi -= s.w_contact_motion * J[r * NV + i] * (-Jdv[r] - kdc * vf[r]);
              if (!on && track_task)
                for (int r = 6 * f; r < 6 * f + 6; r++)
                  gi -= s.w_feet_tracking * J[r * NV + i] * bt[r];
            }
          }
          if (i >= NV && s.w_contact_force > 0 && ((mask >> ((i - NV) / NFV)) & 1u))
          {
            const int f = (i - NV) / NFV, a = (i - NV) % NFV;
            double acc = 0.0;
            for (int r = 0; r < 6; r++)
              acc += id6_tgen(b.quad, f, r, a) * id6_wrench_w(r) * id6_wrench_w(r) * tf[6 * f + r];
            gi -= s.w_contact_force * acc;
          }
        }
        g[i] = gi;
      }
      // ---- general rows of C, l, u ----
      for (int idx = N * NP + lane; idx < G::M * NP; idx += NT)
      {
        const int r = idx / NP, c = idx % NP;
        double val = 0.0;
        if (c < N && r < G::R_MOT)
        { // dynamics rows: [M_b | -(J^T T)_b]
          const int i = r - G::R_DYN;
          val = c < NV ? Mq[i * NV + c] : -JG[i * (NFV * NF) + c - NV];
        }
        else if (c < N && r < G::R_FRI)
        { // contact motion rows (equality variant, feet in contact)
          const int rr = r - G::R_MOT;
          if (s.contact_motion_equality && ((mask >> (rr / 6)) & 1u) && c < NV)
            val = J[rr * NV + c];
        }
        else if (c < N && r < G::R_ACT)
        { // per corner k of foot f: rows 4 k + m: +-f_x - mu f_z, +-f_y - mu f_z ; row 16: sum of the normal forces
          const int rr = r - G::R_FRI, f = rr / 17, m = rr % 17;
          if (((mask >> f) & 1u) && c >= NV + NFV * f && c < NV + NFV * (f + 1))
          {
            const int cc = c - NV - NFV * f, k = cc / 3, j = cc % 3;
            if (m == 16)
              val = j == 2 ? 1.0 : 0.0;
            else if (m / 4 == k)
              val = j == 2 ? -s.friction_coefficient : (j == (m % 4) / 2 ? ((m % 2 == 0) ? 1.0 : -1.0) : 0.0);
          }
        }
        else if (c < N && r < G::M)
        { // actuation rows: [M_a | -(J^T T)_a]
          const int j = r - G::R_ACT;
          val = c < NV ? Mq[(6 + j) * NV + c] : -JG[(6 + j) * (NFV * NF) + c - NV];
        }
        C[idx] = val;
      }
      for (int r = lane; r < MP; r += NT)
      {
        double lo = -ID_INF, hi = ID_INF;
        const double W = b.model->total_mass * 9.81, dt = s.control_dt;
        if (r >= 6 && r < NV)
        {
          const int j = r - 6;
          const double qa = q[7 + j], va = v[6 + j];
          lo = fmax((-b.v_max[j] - va) / dt, 2.0 * (b.q_min[j] - qa - va * dt) / (dt * dt));
          hi = fmin((b.v_max[j] - va) / dt, 2.0 * (b.q_max[j] - qa - va * dt) / (dt * dt));
          if (lo > hi)
            lo = hi = fmin(lo, hi);
          if (s.tsid_bounds)
            id_tsid_acc_limits(qa, va, b.q_min[j], b.q_max[j], b.v_max[j], dt, lo, hi);
        }
        else if (r >= NV && r < N)
        {
          if (!((mask >> ((r - NV) / NFV)) & 1u))
            lo = hi = 0.0;
        }
        else if (r >= N && r < G::R_MOT)
          lo = hi = -nle[r - G::R_DYN];
        else if (r >= G::R_MOT && r < G::R_FRI)
        {
          const int rr = r - G::R_MOT;
          if (s.contact_motion_equality && ((mask >> (rr / 6)) & 1u))
            lo = hi = -Jdv[rr] - kdc * vf[rr];
        }
        else if (r >= G::R_FRI && r < G::R_ACT)
        {
          const int rr = r - G::R_FRI;
          if ((mask >> (rr / 17)) & 1u)
          {
            if (rr % 17 == 16)
            {
              lo = s.ratio_min * W;
              hi = s.ratio_max * W;
            }
            else
              hi = 0.0;
          }
        }
        else if (r >= G::R_ACT && r < G::M)
        {
          const int j = r - G::R_ACT;
          lo = -b.tau_max[j] - nle[6 + j];
          hi = b.tau_max[j] - nle[6 + j];
        }
        l[r] = lo;
        u[r] = hi;
      }
    }
    SMPC_LANES_END_WAVE
  }

  // ADMM of the flat-foot QP (round 5): the algorithm of qp_admm_body (and of the CPU checker's qp_admm) in the same register layout -- one
  // variable, one box row, one DENSE general row and one FRICTION row per lane, the matrix-vector products fed by v_readlane -- made to fit by
  // what the rows are: of the 74 general rows only the 6 dynamics, 12 contact-motion and 22 actuation rows are dense (40 <= 64 lanes, 52
  // entries each); the 34 friction rows of tsid::Contact6d are pyramids on single corner forces (+-f_t - mu f_n: two non-zeros) and the bound
  // on a sole's total normal force (four non-zeros).  Their products, and their 12 x 12 blocks of C^T diag(r) C, are written out in closed form.
  // LDS holds K (for the inverse on the matrix cores), the sweep operands and three small vectors: 31 KB instead of 79 KB -- five resident
  // wavefronts per CU instead of two, and no LDS round trip inside a product (the first form kept C and K^-1 in LDS: 1.09 M QPs/s).
  // The operands of C^T diag(r) C are read from the assembled QP in HBM / L2 (once per factorisation).
  template <class D>
  struct Qp6Lds
  {
    typedef IdDims<D> G;
    double K[G::N * G::N];
    double swp[2 * 4 * 16 * ((2 * G::N + 15) / 16)];
    double rd[64], wfl[64], xl[64]; // weights of the dense rows (operands of the gemm) ; friction-row vector / variable vector handed across lanes
    double red[256], red4[4];
  };
  template <class D>
  SMPC_DEV void qp6_admm_body(const IdBuffers<D> & b, int block)
  {
    typedef IdDims<D> G;
    constexpr int NT = 64, NV = G::NV, NF = G::NF, NA = G::NA, N = G::N, NP = G::NP, MP = G::MP, M = G::M, NFV = G::NFV;
    constexpr int DR = 6 + G::NM * NF + NA; // dense general rows: dynamics | contact motion | actuation
    constexpr int FR = G::NFR * NF;        // friction rows
    static_assert(NP == NT && DR <= NT && FR <= NT && G::NFR == 17 && NFV == 12, "one variable / dense row / friction row per lane; Contact6d rows");
    const int inst = block;
    const IdSettingsDev & st = b.s;
    const double sigma = st.sigma, alpha = st.alpha, fmu = st.friction_coefficient;
    SMPC_LDS(Qp6Lds<D>, ls, 1);
    Qp6Lds<D> & s = ls[0];
    const double * Hg = b.H + (size_t)inst * NP * NP;
    const double * Cg = b.C + (size_t)inst * MP * NP + (size_t)N * NP; // general rows
    const unsigned mask = b.tmask[inst];
    const bool warm = b.warm[inst] != 0;
    double rho = warm ? b.rho[inst] : st.rho;
    // general row of dense row d / of friction row rr
    auto drow = [](int d) { return d < 6 + G::NM * NF ? d : d + FR; };
    SMPC_PL(double, x, NT);
    SMPC_PL(double, g, NT);
    SMPC_PL(double, rhs, NT);
    SMPC_PL(double, xt, NT);
    SMPC_PL(double, zb, NT);
    SMPC_PL(double, lamb, NT);
    SMPC_PL(double, lb, NT);
    SMPC_PL(double, ub, NT);
    SMPC_PL(double, rb, NT);
    SMPC_PL(double, zd, NT);
    SMPC_PL(double, lamd, NT);
    SMPC_PL(double, ld, NT);
    SMPC_PL(double, ud, NT);
    SMPC_PL(double, rdv, NT);
    SMPC_PL(double, wd, NT);
    SMPC_PL(double, zf, NT);
    SMPC_PL(double, lamf, NT);
    SMPC_PL(double, lf, NT);
    SMPC_PL(double, uf, NT);
    SMPC_PL(double, rf, NT);
    SMPC_PLA(double, Ccol, NT, DR);
    SMPC_PLA(double, Crow, NT, N);
    SMPC_PLA(double, Krow, NT, N);
    // ---- lane roles of the friction structure ----
    //   variable lane i = NV + 12 f + 3 k + j (corner k, component j): its friction rows are 17 f + 4 k + {2 j, 2 j + 1} (j < 2) or
    //   17 f + 4 k + {0..3} and 17 f + 16 (j == 2);  friction lane rr = 17 f + m: variables NV + 12 f + 3 (m / 4) + {(m % 4) / 2, 2} (m < 16)
    // C^T w of the friction rows for variable lane `lane` (0 for the accelerations), w read from s.wfl
    auto fric_t = [&](int lane) {
      const int c = lane - NV;
      if (c < 0 || c >= NFV * NF)
        return 0.0;
      const int f = c / NFV, k = (c % NFV) / 3, j = c % 3;
      if (!((mask >> f) & 1u))
        return 0.0;
      const double * w = s.wfl + 17 * f;
      if (j < 2)
        return w[4 * k + 2 * j] - w[4 * k + 2 * j + 1];
      return w[16] - fmu * (w[4 * k] + w[4 * k + 1] + w[4 * k + 2] + w[4 * k + 3]);
    };
    // C y of friction row `lane` (y read from s.xl)
    auto fric_r = [&](int lane) {
      if (lane >= FR)
        return 0.0;
      const int f = lane / 17, m = lane % 17;
      if (!((mask >> f) & 1u))
        return 0.0;
      const double * y = s.xl + NV + NFV * f;
      if (m == 16)
        return y[2] + y[5] + y[8] + y[11];
      const int k = m / 4, j = (m % 4) / 2;
      return ((m % 2 == 0) ? y[3 * k + j] : -y[3 * k + j]) - fmu * y[3 * k + 2];
    };
    SMPC_LANES(NT)
    {
      const int kb = lane < N ? lane : 0;
      const int kd = N + drow(lane < DR ? lane : 0), kf = N + 6 + G::NM * NF + (lane < FR ? lane : 0);
      SMPC_PLV(g) = b.g[(size_t)inst * NP + lane];
      SMPC_PLV(x) = warm ? b.x[(size_t)inst * NP + lane] : 0.0;
      SMPC_PLV(rhs) = SMPC_PLV(xt) = SMPC_PLV(wd) = 0.0;
      auto row = [&](int k, double & lo_, double & hi_, double & z_, double & lam_) {
        const double lo = k < M ? b.l[(size_t)inst * MP + k] : -ID_INF, hi = k < M ? b.u[(size_t)inst * MP + k] : ID_INF;
        lo_ = lo;
        hi_ = hi;
        z_ = warm ? b.z[(size_t)inst * MP + k] : fmin(fmax(0.0, lo), hi);
        lam_ = warm ? b.lam[(size_t)inst * MP + k] : 0.0;
      };
      row(kb, SMPC_PLV(lb), SMPC_PLV(ub), SMPC_PLV(zb), SMPC_PLV(lamb));
      row(kd, SMPC_PLV(ld), SMPC_PLV(ud), SMPC_PLV(zd), SMPC_PLV(lamd));
      row(kf, SMPC_PLV(lf), SMPC_PLV(uf), SMPC_PLV(zf), SMPC_PLV(lamf));
    }
    SMPC_LANES_END_WAVE
    // row weights r = rho (1e3 rho on equality rows, 1e-6 rho on free rows) ; K = H + sigma I + C^T diag(r) C -> its inverse -> rows in registers
    auto factor = [&]() {
      SMPC_LANES(NT)
      {
        auto weight = [&](double lo, double hi) { return (hi - lo < 1e-12) ? 1e3 * rho : ((lo <= -ID_INF && hi >= ID_INF) ? 1e-6 * rho : rho); };
        SMPC_PLV(rb) = weight(SMPC_PLV(lb), SMPC_PLV(ub));
        SMPC_PLV(rdv) = weight(SMPC_PLV(ld), SMPC_PLV(ud));
        SMPC_PLV(rf) = weight(SMPC_PLV(lf), SMPC_PLV(uf));
        s.rd[lane] = lane < DR ? SMPC_PLV(rdv) : 0.0;
        s.wfl[lane] = lane < FR ? SMPC_PLV(rf) : 0.0;
        s.xl[lane] = lane < N ? SMPC_PLV(rb) : 0.0;
      }
      SMPC_LANES_END_WAVE
      fwave_gemm<N, N, DR, 2>( // (operands from the assembled QP in device memory: fetched two K-steps ahead)
        [&](int i, int k) { return s.rd[k] * Cg[drow(k) * NP + i]; }, [&](int k, int j) { return Cg[drow(k) * NP + j]; },
        [&](int i, int j, double v) {
          double kk = Hg[i * NP + j] + (i == j ? sigma + s.xl[i] : 0.0);
          // friction rows: 12 x 12 block per foot in contact
          const int ci = i - NV, cj = j - NV;
          if (ci >= 0 && cj >= 0 && ci / NFV == cj / NFV && ((mask >> (ci / NFV)) & 1u))
          {
            const int f = ci / NFV, ki = (ci % NFV) / 3, ji = ci % 3, kj = (cj % NFV) / 3, jj = cj % 3;
            const double * r = s.wfl + 17 * f;
            if (ki == kj)
            {
              const double * q = r + 4 * ki;
              if (ji == jj)
                kk += ji == 0 ? q[0] + q[1] : (ji == 1 ? q[2] + q[3] : fmu * fmu * (q[0] + q[1] + q[2] + q[3]) + r[16]);
              else if (ji == 2 || jj == 2)
              {
                const int t = ji == 2 ? jj : ji; // the tangential component of the pair
                kk -= fmu * (q[2 * t] - q[2 * t + 1]);
              }
            }
            else if (ji == 2 && jj == 2)
              kk += r[16];
          }
          s.K[i * N + j] = kk + v;
        });
      fwave_spd_inverse<N>(s.K, s.swp);
      SMPC_LANES(NT)
      {
        const int ln = lane < N ? lane : 0; // (lanes N .. 63 carry no variable: their x~ is zero)
#pragma unroll
        for (int j = 0; j < N; j++)
          SMPC_PLV(Krow)[j] = s.K[j * N + ln]; // (K^-1 is symmetric: read along the row of j, conflict-free)
        // rows and columns of the dense part of C straight from HBM / L2 into the registers they stay in -- (re)loaded AFTER the inverse, so that
        // they are not live across it (the accumulators of the product and of the sweeps need the registers)
        const int kd = drow(lane < DR ? lane : 0);
#pragma unroll
        for (int d = 0; d < DR; d++)
          SMPC_PLV(Ccol)[d] = Cg[drow(d) * NP + ln];
#pragma unroll
        for (int ii = 0; ii < N; ii++)
          SMPC_PLV(Crow)[ii] = Cg[kd * NP + ii];
      }
      SMPC_LANES_END_WAVE
    };
    double rs[4] = {0.0, 0.0, 0.0, 0.0};
    auto residual = [&]() {
      SMPC_LANES(NT)
      {
        s.xl[lane] = lane < N ? SMPC_PLV(x) : 0.0;
        s.wfl[lane] = lane < FR ? SMPC_PLV(lamf) : 0.0;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        double pr = 0.0, np = 0.0, du = 0.0, nd = 0.0;
        double cx = 0.0, hx = 0.0, cl = SMPC_PLV(lamb);
#pragma unroll
        for (int i = 0; i < N; i++)
        {
          cx += SMPC_PLV(Crow)[i] * SMPC_XLANE(x, i);
          if (i % 8 == 7)
            SMPC_SCHED_FENCE();
        }
        for (int j = 0; j < N; j++)
          hx += Hg[j * NP + lane] * SMPC_XLANE(x, j); // (H is symmetric: coalesced along the row of j)
#pragma unroll
        for (int k = 0; k < DR; k++)
        {
          cl += SMPC_PLV(Ccol)[k] * SMPC_XLANE(lamd, k);
          if (k % 8 == 7)
            SMPC_SCHED_FENCE();
        }
        cl += fric_t(lane);
        const double cf = fric_r(lane);
        if (lane < DR)
        {
          pr = fabs(cx - SMPC_PLV(zd));
          np = fmax(fabs(cx), fabs(SMPC_PLV(zd)));
        }
        if (lane < FR)
        {
          pr = fmax(pr, fabs(cf - SMPC_PLV(zf)));
          np = fmax(np, fmax(fabs(cf), fabs(SMPC_PLV(zf))));
        }
        if (lane < N)
        { // box rows: C x = x
          pr = fmax(pr, fabs(SMPC_PLV(x) - SMPC_PLV(zb)));
          np = fmax(np, fmax(fabs(SMPC_PLV(x)), fabs(SMPC_PLV(zb))));
          du = fabs((SMPC_PLV(g) + hx) + cl);
          nd = fmax(fabs(hx), fmax(fabs(cl), fabs(SMPC_PLV(g))));
        }
        s.red[lane] = pr;
        s.red[64 + lane] = du;
        s.red[128 + lane] = np;
        s.red[192 + lane] = nd;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane < 4)
      {
        double m = 0.0; // (a NaN entry must survive the reduction)
        for (int i = 0; i < NT; i++)
        {
          const double v = s.red[64 * lane + i];
          m = (v != v) ? v : ((m != m) ? m : fmax(m, v));
        }
        s.red4[lane] = m;
      }
      SMPC_LANES_END_WAVE
      for (int i = 0; i < 4; i++)
        rs[i] = s.red4[i];
    };
    factor();
    bool done = false;
    for (int it = 0; it < st.admm_iters; it++)
    {
      if (it > 0 && it % ADMM_CHECK == 0)
      {
        residual();
        if (st.admm_tol >= 0.0 && fmax(rs[0], rs[1]) <= st.admm_tol)
        {
          done = true;
          break;
        }
        const double est = fmin(fmax(rho * sqrt((rs[0] / (rs[2] + 1e-10)) / (rs[1] / (rs[3] + 1e-10) + 1e-10)), 1e-6), 1e6);
        if (fmax(rs[0], rs[1]) > ADMM_ADAPT_FLOOR && (est > 5.0 * rho || est < 0.2 * rho))
        {
          rho = est;
          factor();
        }
      }
      SMPC_LANES(NT)
      {
        SMPC_PLV(wd) = SMPC_PLV(rdv) * SMPC_PLV(zd) - SMPC_PLV(lamd);
        s.wfl[lane] = lane < FR ? SMPC_PLV(rf) * SMPC_PLV(zf) - SMPC_PLV(lamf) : 0.0;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      { // rhs = sigma x - g + C^T (r z - lam): the box rows contribute their own entry, the friction rows their closed form
        double acc = sigma * SMPC_PLV(x) - SMPC_PLV(g);
        acc += lane < N ? SMPC_PLV(rb) * SMPC_PLV(zb) - SMPC_PLV(lamb) : 0.0;
#pragma unroll
        for (int k = 0; k < DR; k++)
        {
          acc += SMPC_PLV(Ccol)[k] * SMPC_XLANE(wd, k);
          if (k % 8 == 7)
            SMPC_SCHED_FENCE();
        }
        acc += fric_t(lane);
        SMPC_PLV(rhs) = lane < N ? acc : 0.0;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < N; j++)
        {
          acc += SMPC_PLV(Krow)[j] * SMPC_XLANE(rhs, j);
          if (j % 8 == 7)
            SMPC_SCHED_FENCE();
        }
        SMPC_PLV(xt) = lane < N ? acc : 0.0;
        s.xl[lane] = SMPC_PLV(xt);
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < N; i++)
        {
          acc += SMPC_PLV(Crow)[i] * SMPC_XLANE(xt, i);
          if (i % 8 == 7)
            SMPC_SCHED_FENCE();
        }
        const double ztd = acc, ztf = fric_r(lane);
        auto upd = [&](double zt, double & z, double & lam, double r, double lo, double hi) {
          const double zh = alpha * zt + (1.0 - alpha) * z;
          const double zn = fmin(fmax(zh + lam / r, lo), hi);
          lam += r * (zh - zn);
          z = zn;
        };
        upd(SMPC_PLV(xt), SMPC_PLV(zb), SMPC_PLV(lamb), SMPC_PLV(rb), SMPC_PLV(lb), SMPC_PLV(ub)); // box rows: z~ = x~
        upd(ztd, SMPC_PLV(zd), SMPC_PLV(lamd), SMPC_PLV(rdv), SMPC_PLV(ld), SMPC_PLV(ud));
        upd(ztf, SMPC_PLV(zf), SMPC_PLV(lamf), SMPC_PLV(rf), SMPC_PLV(lf), SMPC_PLV(uf));
        SMPC_PLV(x) = alpha * SMPC_PLV(xt) + (1.0 - alpha) * SMPC_PLV(x);
      }
      SMPC_LANES_END_WAVE
    }
    if (!done)
      residual();
    const double res = (rs[0] != rs[0] || rs[1] != rs[1]) ? rs[0] + rs[1] : fmax(rs[0], rs[1]);
    SMPC_LANES(NT)
    s.xl[lane] = SMPC_PLV(x);
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      const bool ok = res == res && res < 1e300;
      if (ok)
      {
        b.x[(size_t)inst * NP + lane] = SMPC_PLV(x);
        if (lane < N)
        {
          b.z[(size_t)inst * MP + lane] = SMPC_PLV(zb);
          b.lam[(size_t)inst * MP + lane] = SMPC_PLV(lamb);
        }
        if (lane < DR)
        {
          b.z[(size_t)inst * MP + N + drow(lane)] = SMPC_PLV(zd);
          b.lam[(size_t)inst * MP + N + drow(lane)] = SMPC_PLV(lamd);
        }
        if (lane < FR)
        {
          b.z[(size_t)inst * MP + N + 6 + G::NM * NF + lane] = SMPC_PLV(zf);
          b.lam[(size_t)inst * MP + N + 6 + G::NM * NF + lane] = SMPC_PLV(lamf);
        }
      }
      for (int i = lane; i < NV; i += NT)
        b.a[(size_t)inst * NV + i] = s.xl[i];
      // contact wrenches T f (foot frames) -> red[0 .. 6 NF)
      if (lane < 6 * NF)
      {
        const int f = lane / 6, r = lane % 6;
        double acc = 0.0;
        for (int c = 0; c < NFV; c++)
          acc += id6_tgen(b.quad, f, r, c) * s.xl[NV + NFV * f + c];
        s.red[lane] = acc;
        b.f[(size_t)inst * 6 * NF + lane] = acc;
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      if (lane < NA)
      { // tau = M_a a + h_a - J_a^T (T f)
        const double * Mq = b.Mq + (size_t)inst * NV * NV;
        const double * J = b.J + (size_t)inst * 6 * NF * NV;
        double acc = b.nle[(size_t)inst * NV + 6 + lane];
        for (int k = 0; k < NV; k++)
          acc += Mq[(6 + lane) * NV + k] * s.xl[k];
        for (int r = 0; r < 6 * NF; r++)
          acc -= J[r * NV + 6 + lane] * s.red[r];
        b.tau[(size_t)inst * NA + lane] = acc;
      }
      if (lane == 0)
      {
        const bool ok = res == res && res < 1e300;
        b.resid[inst] = res;
        b.rho[inst] = ok ? rho : st.rho;
        b.warm[inst] = ok ? 1 : 0;
      }
    }
    SMPC_LANES_END_WAVE
  }

  // ---- host engine ----
  struct HostIdSettings
  {
    IdSettingsDev dev;
    std::vector<double> tau_max, v_max, q_min, q_max;
    std::vector<double> quad_points; // flat feet: [nf][4][3] corners of the soles in their foot frames
  };
  struct IdEngineBase
  {
    int B = 0, nq = 0, nv = 0, nf = 0, n = 0, m = 0, np = 0, mp = 0;
    int nfw = 3, nmot = 3; // size of a force target / reported contact force per foot (3, or the 6-D wrench of a flat foot) ; contact-motion rows per foot
    virtual ~IdEngineBase() {}
    virtual void set_target(int inst, const double * q, const double * v, const double * a, unsigned mask, const double * f) = 0;
    virtual void set_targets(const double * Q, const double * V, const double * A, const unsigned char * contact, const double * F) = 0;
    // CentroidalID::setTarget (centroidal-id.cpp:86-147): com, vcom (3), feet positions / velocities (3 nf, world frame), contacts, forces
    virtual void set_target_centroidal(int inst, const double * com, const double * vcom, const double * fp, const double * fv, unsigned mask, const double * f) = 0;
    virtual void set_targets_centroidal(const double * COM, const double * VCOM, const double * FP, const double * FV, const unsigned char * contact, const double * F) = 0;
    virtual void solve(const double * X, double * tau, double * a, double * f, double * resid) = 0;
    // states and results resident in HBM; asynchronous on the engine's stream (wait() joins).  tau_dev may be null: results stay in
    // the engine's own buffers (tau_device() ...)
    virtual void solve_device(const double * X_dev, double * tau_dev) = 0;
    // device buffers the targets live in (an MPC engine's interpolation kernel writes them in place): states [B][nq + nv], accelerations
    // [B][nv], forces [B][3 nf]; set_mask_all: the contact flags of every robot; stream(): where the solve is issued
    virtual void target_buffers(double ** x, double ** a, double ** f) = 0;
    virtual void centroidal_target_buffers(double ** com, double ** vcom, double ** fp, double ** fv) = 0; // CentroidalID; null otherwise
    virtual void set_mask_all(unsigned mask) = 0;
    virtual stream_t solve_stream() = 0;
    // issue this controller's work on another engine's stream from now on (one in-order queue for MPC, controller and simulator: no
    // events, no host-side waits between them); back_to_own: the controller's own stream again
    virtual void adopt_stream(stream_t s, bool back_to_own) = 0;
    virtual void wait() = 0;
    virtual int device() const = 0;
    virtual const double * tau_device() const = 0;
    virtual double * x_device() = 0; // the engine's own state buffer [B][nq + nv] (what solve() copies the host states into)
    virtual void reset(int inst) = 0;          // forget the warm start of one robot (inst < 0: all)
    virtual void get_resid(double * out) = 0;  // residuals of the last solve [B] (joins the stream)
    virtual void debug_get(int what, double * out) = 0; // 0 M, 1 nle, 2 J, 3 Jdv, 4 vfoot, 5 H, 6 g, 7 C, 8 l, 9 u (padded layouts), 10 com, 11 footp, 12 tau
  };
  template <class D>
  struct IdEngine : IdEngineBase
  {
    typedef IdDims<D> G;
    IdBuffers<D> buf;
    stream_t stream, own_stream;
    int device_id = 0;
    double * Xd = nullptr;
    std::vector<void *> allocs;
    unsigned mask_all = 0;
    bool mask_all_valid = false; // (the per-robot setters invalidate it)
    IdEngine(const smpc_robot_model * rm, const HostIdSettings & hs, int batch, int device)
    {
      if (rm->njoints != D::NJ || rm->nfeet != D::NF)
        throw std::runtime_error("robot shape (njoints, nfeet) does not match this kernel instantiation");
      if (batch <= 0)
        throw std::runtime_error("batch must be positive");
      if ((int)hs.tau_max.size() != G::NA || (int)hs.v_max.size() != G::NA || (int)hs.q_min.size() != G::NA || (int)hs.q_max.size() != G::NA)
        throw std::runtime_error("inverse-dynamics settings: limit vectors must have nv - 6 entries");
      if (!(hs.dev.control_dt > 0.0) || hs.dev.admm_iters <= 0)
        throw std::runtime_error("inverse-dynamics settings: control_dt and the iteration count must be positive");
      device_id = device;
      set_device(device);
      stream = own_stream = stream_create();
      try
      {
        construct(rm, hs, batch);
      }
      catch (...)
      { // (the destructor does not run for a partially constructed engine)
        for (void * p : allocs)
          dev_free(p);
        stream_destroy(own_stream);
        throw;
      }
    }
    void construct(const smpc_robot_model * rm, const HostIdSettings & hs, int batch)
    {
      B = batch;
      nq = D::NQ;
      nv = D::NV;
      nf = D::NF;
      nfw = G::NFW;
      nmot = G::NM;
      n = G::N;
      m = G::M;
      np = G::NP;
      mp = G::MP;
      std::vector<DevModel<D>> hm(1);
      std::memset(&hm[0], 0, sizeof(DevModel<D>));
      fill_tree_model<D>(rm, hm[0]);
      hm[0].gravity[2] = -9.81; // (Kp = Kd = 0: plain contact rows)
      auto dalloc = [&](size_t cnt) {
        void * p = dev_alloc(cnt * sizeof(double));
        dev_zero(p, cnt * sizeof(double), stream);
        allocs.push_back(p);
        return (double *)p;
      };
      buf.B = B;
      buf.model = (DevModel<D> *)dev_alloc(sizeof(DevModel<D>));
      allocs.push_back(buf.model);
      h2d(buf.model, hm.data(), sizeof(DevModel<D>), stream);
      const size_t Bs = (size_t)B;
      Xd = dalloc(Bs * D::NX);
      buf.X = Xd;
      buf.Mq = dalloc(Bs * nv * nv);
      buf.nle = dalloc(Bs * nv);
      buf.J = dalloc(Bs * nmot * nf * nv);
      buf.Jdv = dalloc(Bs * nmot * nf);
      buf.vfoot = dalloc(Bs * nmot * nf);
      buf.com = dalloc(Bs * 3);
      buf.footp = dalloc(Bs * 3 * nf);
      buf.tcom = dalloc(Bs * 3);
      buf.tvcom = dalloc(Bs * 3);
      buf.tfp = dalloc(Bs * 3 * nf);
      buf.tfv = dalloc(Bs * 3 * nf);
      buf.H = dalloc(Bs * np * np);
      buf.g = dalloc(Bs * np);
      buf.C = dalloc(Bs * mp * np);
      {
        std::vector<double> c0(Bs * mp * np, 0.0); // constant rows of C: the identity of the box on y
        for (size_t b = 0; b < Bs; b++)
          for (int i = 0; i < n; i++)
            c0[(b * mp + i) * np + i] = 1.0;
        h2d(buf.C, c0.data(), c0.size() * sizeof(double), stream);
        stream_sync(stream);
      }
      buf.l = dalloc(Bs * mp);
      buf.u = dalloc(Bs * mp);
      buf.x = dalloc(Bs * np);
      buf.z = dalloc(Bs * mp);
      buf.lam = dalloc(Bs * mp);
      buf.rho = dalloc(Bs);
      buf.warm = (int *)dev_alloc(Bs * sizeof(int));
      allocs.push_back(buf.warm);
      dev_zero(buf.warm, Bs * sizeof(int), stream);
      buf.tx = dalloc(Bs * (nq + nv));
      buf.ta = dalloc(Bs * nv);
      buf.tf = dalloc(Bs * nfw * nf);
      buf.tmask = (unsigned *)dev_alloc(Bs * sizeof(unsigned));
      allocs.push_back(buf.tmask);
      buf.tau = dalloc(Bs * G::NA);
      buf.a = dalloc(Bs * nv);
      buf.f = dalloc(Bs * nfw * nf);
      if constexpr (D::FS == 6)
      {
        if ((int)hs.quad_points.size() != nf * 12)
          throw std::runtime_error("inverse-dynamics settings: flat feet need the four corners of every sole (quad_points, [nfeet][4][3])");
        buf.footR = dalloc(Bs * 9 * nf);
        buf.quad = dalloc((size_t)nf * 12);
        h2d(buf.quad, hs.quad_points.data(), (size_t)nf * 12 * sizeof(double), stream);
      }
      buf.resid = dalloc(Bs);
      buf.tau_max = dalloc(G::NA);
      buf.v_max = dalloc(G::NA);
      buf.q_min = dalloc(G::NA);
      buf.q_max = dalloc(G::NA);
      h2d(buf.tau_max, hs.tau_max.data(), G::NA * sizeof(double), stream);
      h2d(buf.v_max, hs.v_max.data(), G::NA * sizeof(double), stream);
      h2d(buf.q_min, hs.q_min.data(), G::NA * sizeof(double), stream);
      h2d(buf.q_max, hs.q_max.data(), G::NA * sizeof(double), stream);
      buf.s = hs.dev;
      stream_sync(stream);
      // default target: the reference state, every foot in contact with an equal share of the weight (kinodynamics-id.cpp:96-112)
      std::vector<double> q(rm->q_ref, rm->q_ref + nq), z(nv, 0.0), f((size_t)nfw * nf, 0.0);
      for (int k = 0; k < nf; k++)
        f[nfw * k + 2] = rm->total_mass * 9.81 / nf;
      set_target(-1, q.data(), z.data(), z.data(), (1u << nf) - 1u, f.data());
      if (buf.s.centroidal)
      { // CoM of the reference state (one pass of the first kernel), feet at their reference placements (centroidal-id.cpp:60-84)
        std::vector<double> X((size_t)B * D::NX, 0.0);
        for (int b = 0; b < B; b++)
          std::copy(q.begin(), q.end(), X.begin() + (size_t)b * D::NX);
        h2d(Xd, X.data(), X.size() * sizeof(double), stream);
        launch<IdBuffers<D>, id_quant_body<D>, 64, 1, 0>(B, stream, buf);
        double com[3];
        d2h(com, buf.com, sizeof(com), stream);
        stream_sync(stream);
        const M3 R0 = quat_to_R(Quat{q[3], q[4], q[5], q[6]});
        std::vector<double> fp(3 * nf), zf(3 * nf, 0.0);
        for (int k = 0; k < nf; k++)
          st3(&fp[3 * k], ld3(q.data()) + R0 * ld3(rm->foot_ref_p[k]));
        set_target_centroidal(-1, com, z.data(), fp.data(), zf.data(), (1u << nf) - 1u, f.data());
      }
    }
    ~IdEngine()
    {
      for (void * p : allocs)
        dev_free(p);
      stream_destroy(own_stream);
    }
    void set_target(int inst, const double * q, const double * v, const double * a, unsigned mask, const double * f) override
    {
      set_device(device_id);
      if (inst >= B)
        throw std::runtime_error("instance index exceeds the batch");
      const int i0 = inst < 0 ? 0 : inst, i1 = inst < 0 ? B : inst + 1;
      const int nx = nq + nv;
      std::vector<double> tx((size_t)(i1 - i0) * nx), ta((size_t)(i1 - i0) * nv), tf((size_t)(i1 - i0) * nfw * nf);
      std::vector<unsigned> tm(i1 - i0, mask);
      for (int i = 0; i < i1 - i0; i++)
      {
        std::copy(q, q + nq, tx.begin() + (size_t)i * nx);
        std::copy(v, v + nv, tx.begin() + (size_t)i * nx + nq);
        std::copy(a, a + nv, ta.begin() + (size_t)i * nv);
        std::copy(f, f + nfw * nf, tf.begin() + (size_t)i * nfw * nf);
      }
      h2d(buf.tx + (size_t)i0 * nx, tx.data(), tx.size() * sizeof(double), stream);
      h2d(buf.ta + (size_t)i0 * nv, ta.data(), ta.size() * sizeof(double), stream);
      h2d(buf.tf + (size_t)i0 * nfw * nf, tf.data(), tf.size() * sizeof(double), stream);
      h2d(buf.tmask + i0, tm.data(), tm.size() * sizeof(unsigned), stream);
      mask_all_valid = false;
      stream_sync(stream);
    }
    // one target per robot: Q [B][nq], V [B][nv], A [B][nv], contact [B][nf], F [B][3 nf]
    void set_targets(const double * Q, const double * V, const double * A, const unsigned char * contact, const double * F) override
    {
      set_device(device_id);
      std::vector<unsigned> tm(B, 0u);
      for (int b = 0; b < B; b++)
        for (int k = 0; k < nf; k++)
          tm[b] |= contact[(size_t)b * nf + k] ? (1u << k) : 0u;
      const int nx = nq + nv;
      std::vector<double> tx((size_t)B * nx);
      for (int b = 0; b < B; b++)
      {
        std::copy(Q + (size_t)b * nq, Q + (size_t)(b + 1) * nq, tx.begin() + (size_t)b * nx);
        std::copy(V + (size_t)b * nv, V + (size_t)(b + 1) * nv, tx.begin() + (size_t)b * nx + nq);
      }
      h2d(buf.tx, tx.data(), tx.size() * sizeof(double), stream);
      h2d(buf.ta, A, (size_t)B * nv * sizeof(double), stream);
      h2d(buf.tf, F, (size_t)B * nfw * nf * sizeof(double), stream);
      h2d(buf.tmask, tm.data(), (size_t)B * sizeof(unsigned), stream);
      mask_all_valid = false;
      stream_sync(stream);
    }
    void set_target_centroidal(int inst, const double * com, const double * vcom, const double * fp, const double * fv, unsigned mask, const double * f) override
    {
      set_device(device_id);
      if (!buf.s.centroidal)
        throw std::runtime_error("this inverse-dynamics engine was created as KinodynamicsID");
      if (inst >= B)
        throw std::runtime_error("instance index exceeds the batch");
      const int i0 = inst < 0 ? 0 : inst, i1 = inst < 0 ? B : inst + 1, cnt = i1 - i0;
      std::vector<double> c3((size_t)cnt * 3), v3((size_t)cnt * 3), tp((size_t)cnt * 3 * nf), tv((size_t)cnt * 3 * nf), tf((size_t)cnt * nfw * nf);
      std::vector<unsigned> tm(cnt, mask);
      for (int i = 0; i < cnt; i++)
      {
        std::copy(com, com + 3, c3.begin() + (size_t)i * 3);
        std::copy(vcom, vcom + 3, v3.begin() + (size_t)i * 3);
        std::copy(fp, fp + 3 * nf, tp.begin() + (size_t)i * 3 * nf);
        std::copy(fv, fv + 3 * nf, tv.begin() + (size_t)i * 3 * nf);
        std::copy(f, f + nfw * nf, tf.begin() + (size_t)i * nfw * nf);
      }
      h2d(buf.tcom + (size_t)i0 * 3, c3.data(), c3.size() * sizeof(double), stream);
      h2d(buf.tvcom + (size_t)i0 * 3, v3.data(), v3.size() * sizeof(double), stream);
      h2d(buf.tfp + (size_t)i0 * 3 * nf, tp.data(), tp.size() * sizeof(double), stream);
      h2d(buf.tfv + (size_t)i0 * 3 * nf, tv.data(), tv.size() * sizeof(double), stream);
      h2d(buf.tf + (size_t)i0 * nfw * nf, tf.data(), tf.size() * sizeof(double), stream);
      h2d(buf.tmask + i0, tm.data(), tm.size() * sizeof(unsigned), stream);
      mask_all_valid = false;
      stream_sync(stream);
    }
    void set_targets_centroidal(const double * COM, const double * VCOM, const double * FP, const double * FV, const unsigned char * contact, const double * F) override
    {
      set_device(device_id);
      if (!buf.s.centroidal)
        throw std::runtime_error("this inverse-dynamics engine was created as KinodynamicsID");
      std::vector<unsigned> tm(B, 0u);
      for (int b = 0; b < B; b++)
        for (int k = 0; k < nf; k++)
          tm[b] |= contact[(size_t)b * nf + k] ? (1u << k) : 0u;
      h2d(buf.tcom, COM, (size_t)B * 3 * sizeof(double), stream);
      h2d(buf.tvcom, VCOM, (size_t)B * 3 * sizeof(double), stream);
      h2d(buf.tfp, FP, (size_t)B * 3 * nf * sizeof(double), stream);
      h2d(buf.tfv, FV, (size_t)B * 3 * nf * sizeof(double), stream);
      h2d(buf.tf, F, (size_t)B * nfw * nf * sizeof(double), stream);
      h2d(buf.tmask, tm.data(), (size_t)B * sizeof(unsigned), stream);
      mask_all_valid = false;
      stream_sync(stream);
    }
    void launch_all()
    {
      launch<IdBuffers<D>, id_quant_body<D>, 64, 1, 0>(B, stream, buf);
      if constexpr (D::FS == 6)
      { // flat feet (tsid Contact6d): 52 variables / 74 general rows
        launch<IdBuffers<D>, id6_assemble_body<D>, 64, 1, 0>(B, stream, buf);
        launch<IdBuffers<D>, qp6_admm_body<D>, 64, 1, 0>(B, stream, buf);
      }
      else
      {
        launch<IdBuffers<D>, id_assemble_body<D>, 64, 1, 0>(B, stream, buf);
        launch<IdBuffers<D>, qp_admm_body<D>, 64, 1, 0>(B, stream, buf);
      }
    }
    void solve_device(const double * X_dev, double * tau_dev) override
    {
      set_device(device_id);
      buf.X = X_dev;
      launch_all();
      buf.X = Xd;
      if (tau_dev)
        d2d(tau_dev, buf.tau, (size_t)B * G::NA * sizeof(double), stream);
    }
    void wait() override
    {
      set_device(device_id);
      stream_sync(stream);
    }
    int device() const override { return device_id; }
    const double * tau_device() const override { return buf.tau; }
    double * x_device() override { return Xd; }
    void target_buffers(double ** x, double ** a, double ** f) override
    {
      *x = buf.tx;
      *a = buf.ta;
      *f = buf.tf;
    }
    void centroidal_target_buffers(double ** com, double ** vcom, double ** fp, double ** fv) override
    {
      const bool c = buf.s.centroidal != 0;
      *com = c ? buf.tcom : nullptr;
      *vcom = c ? buf.tvcom : nullptr;
      *fp = c ? buf.tfp : nullptr;
      *fv = c ? buf.tfv : nullptr;
    }
    void set_mask_all(unsigned mask) override
    {
      set_device(device_id);
      if (mask != mask_all || !mask_all_valid)
      {
        std::vector<unsigned> tm(B, mask);
        h2d(buf.tmask, tm.data(), tm.size() * sizeof(unsigned), stream);
        stream_sync(stream);
        mask_all = mask;
        mask_all_valid = true;
      }
    }
    stream_t solve_stream() override { return stream; }
    void adopt_stream(stream_t s, bool back_to_own) override
    {
      set_device(device_id);
      stream_sync(stream);
      stream = back_to_own ? own_stream : s;
    }
    void solve(const double * X, double * tau, double * a, double * f, double * resid) override
    {
      set_device(device_id);
      h2d(Xd, X, (size_t)B * D::NX * sizeof(double), stream);
      launch_all();
      d2h(tau, buf.tau, (size_t)B * G::NA * sizeof(double), stream);
      d2h(a, buf.a, (size_t)B * nv * sizeof(double), stream);
      d2h(f, buf.f, (size_t)B * nfw * nf * sizeof(double), stream);
      if (resid)
        d2h(resid, buf.resid, (size_t)B * sizeof(double), stream);
      stream_sync(stream);
    }
    void reset(int inst) override
    {
      set_device(device_id);
      if (inst >= B)
        throw std::runtime_error("instance index exceeds the batch");
      const int i0 = inst < 0 ? 0 : inst, cnt = inst < 0 ? B : 1;
      dev_zero(buf.warm + i0, (size_t)cnt * sizeof(int), stream);
      stream_sync(stream);
    }
    void get_resid(double * out) override
    {
      set_device(device_id);
      d2h(out, buf.resid, (size_t)B * sizeof(double), stream);
      stream_sync(stream);
    }
    void debug_get(int what, double * out) override
    {
      set_device(device_id);
      const double * src[13] = {buf.Mq, buf.nle, buf.J, buf.Jdv, buf.vfoot, buf.H, buf.g, buf.C, buf.l, buf.u, buf.com, buf.footp, buf.tau};
      const size_t per[13] = {(size_t)nv * nv, (size_t)nv, (size_t)nmot * nf * nv, (size_t)nmot * nf, (size_t)nmot * nf, (size_t)np * np, (size_t)np, (size_t)mp * np, (size_t)mp, (size_t)mp, 3, (size_t)3 * nf, (size_t)G::NA};
      if (what < 0 || what > 12)
        throw std::runtime_error("unknown quantity");
      d2h(out, src[what], (size_t)B * per[what] * sizeof(double), stream);
      stream_sync(stream);
    }
  };
} // namespace smpc
