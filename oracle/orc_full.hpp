// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_full.hpp: first block of the full-dynamics OCP: the constrained forward dynamics the
// reference obtains from pinocchio::constraintDynamics (Pinocchio 3.8, un-vendored, pixi.lock:154) through Aligator's
// MultibodyConstraintFwdDynamics [REF src/fulldynamics.cpp:139], restated on the smpc_robot_model tree:
//   - one RigidConstraintModel per foot in contact, CONTACT_3D, reference frame LOCAL, between the foot frame on its
//     parent joint (joint1_placement = frame placement) and the universe joint with identity placement
//     [REF src/fulldynamics.cpp:50-75], Baumgarte corrector (Kp, Kd) [REF :64-65]
//   - actuation matrix = [0; I]: the nv-6 joint torques [REF src/fulldynamics.cpp:35-37]
//   - ProximalSettings(1e-9, 1e-10, 10) [REF src/fulldynamics.cpp:39]: absolute accuracy, mu, max iterations
//
//     [ M  J^T ] [  a  ]   [ S tau - nle           ]        J a + drift = -Kd v_c + Kp (p_2 - p_c)   (contact frame)
//     [ J -mu I] [-lam ] = [ -(drift - a_desired)  ]
//
// solved by the proximal iteration on the contact forces (Schur complement on the Cholesky factor of M):
//     (J M^-1 J^T + mu I) lam_{k+1} = mu lam_k - gamma - J M^-1 (S tau - nle),   gamma = drift - a_desired
// until |lam_{k+1} - lam_k|_inf <= accuracy.
//   M   : composite-rigid-body joint-space inertia,  M_kl = S_k . (Ic_j S_l)
//   nle : recursive Newton-Euler with zero joint accelerations and the gravity field
//   J   : LOCAL linear Jacobian of the foot point;  drift : its classical acceleration at zero joint accelerations
// The anchor of the corrector is the ORIGIN of the universe frame (joint2 = 0, placement identity), as the reference
// builds it: with Kp = diag(0, 0, kz) it pulls the foot height to z = 0 (examples/talos_fulldynamics.py:86).
// derivatives() restates pinocchio::computeConstraintDynamicsDerivatives: implicit differentiation of
//     r1 = RNEA(q, v, a) - S tau - J^T lam = 0,      r2 = (contact acceleration)(q, v, a) + Kd v_c - Kp p_err = 0
// at the solution, with the partial derivatives of RNEA and of the contact-frame acceleration written in the world-frame
// formulation of orc_rigid.hpp (d_k = v_lam x S_k, A_k = (a_lam - g) x S_k + v_lam x d_k):
//     d tau_m / d q_k = S_m . (Ic_s A_k + Bc_s d_k)  [+ S_m . (S_k x* Fc_i) if joint(m) is a strict ancestor of joint(k)]
//     d tau_m / d v_k = S_m . (Bc_s S_k + Ic_s (v_i x S_k + d_k))        s = the lower of joint(m), joint(k) (same branch)
// 6-D feet (fs = 6): RigidConstraintModel(CONTACT_6D, LOCAL_WORLD_ALIGNED) [REF src/fulldynamics.cpp:56-65]: the contact frame has
// its origin at the foot frame and the axes of the world; rows = [linear ; angular]:
//   J_k   = [S_k.l + S_k.a x p ; S_k.a],   drift = [classical acceleration of the point ; angular acceleration] at zero joint
//   accelerations,  corrector  Kd o [v_p ; w]  +  Kp o [p ; log3(R)]   (anchor = universe frame, identity placement),
//   lam   = the wrench [f ; tau] ON the robot at the foot point, world axes.
// The world-aligned components of a body-fixed vector u turn with the body when q_k moves: d u / d q_k = (co-moving part, the
// LOCAL formulas) + S_k.a x u; a world-aligned force does not turn, only its point of application moves (dp = J_k lin).
// The OCP on top (costs, constraints, integrator): orc_fulldyn.hpp.
#pragma once
#include "orc_rigid.hpp"

namespace orc
{
  inline double sv_dot(const SV & motion, const SV & force) { return dot(motion.l, force.l) + dot(motion.a, force.a); }

  struct ConstraintDynamics
  {
    const smpc_robot_model * M;
    Rigid R;
    int nv, nu;
    double prox_accuracy = 1e-9, prox_mu = 1e-10;
    int prox_max_iter = 10;
    double gravity[3] = {0, 0, -9.81};
    int fs = 3;          // contact size: 3 (CONTACT_3D, LOCAL) or 6 (CONTACT_6D, LOCAL_WORLD_ALIGNED)
    double Kp[6] = {0, 0, 0, 0, 0, 0}, Kd[6] = {0, 0, 0, 0, 0, 0};
    // results of the last call
    Mat Mq, Jc;          // nv x nv ; fs n_c x nv
    Vec nle, gamma;      // nv ; fs n_c
    Vec a, lam;          // nv ; fs n_c (force ON the robot at the foot, contact frame)
    int prox_iters = 0;
    std::vector<int> feet; // feet in contact, in order
    Mat Lm, MJ, Gc;      // Cholesky factor of M ; M^-1 J^T ; Cholesky factor of the damped Delassus matrix
    // derivatives() results
    Mat da_dq, da_dv, da_dtau, dlam_dq, dlam_dv, dlam_dtau;
    Mat dtau_dq, dtau_dv; // partial derivatives of RNEA(q, v, a) at the solution

    explicit ConstraintDynamics(const smpc_robot_model * m) : M(m), R(m), nv(m->nv), nu(m->nv - 6) {}

    // joint-space inertia (needs R.fk)
    void crba()
    {
      Mq = Mat(nv, nv);
      for (int l = 0; l < nv; l++)
      {
        const int j = R.dof2j[l];
        const SV f = R.Ic[j] * R.S[l];
        for (int k = 0; k < nv; k++)
          if (R.is_ancestor_dof(k, j))
          {
            Mq(k, l) = sv_dot(R.S[k], f);
            Mq(l, k) = Mq(k, l);
          }
      }
    }
    // generalized force of the body forces F_j (summed over the subtree of each dof)
    Vec project(const std::vector<SV> & Fsub) const
    {
      Vec t(nv);
      for (int k = 0; k < nv; k++)
        t[k] = sv_dot(R.S[k], Fsub[R.dof2j[k]]);
      return t;
    }
    // recursive Newton-Euler: tau = M a + nle (a may be null)
    Vec rnea(const double * v, const double * acc)
    {
      R.forces(v, acc);
      const SV g{v3(gravity[0], gravity[1], gravity[2]), v3(0, 0, 0)};
      std::vector<SV> Fs(R.nj);
      for (int j = 0; j < R.nj; j++)
        Fs[j] = R.F[j] - R.I[j] * g; // uniform field: every body accelerates with -g relative to free fall
      for (int j = R.nj - 1; j > 0; j--)
        Fs[M->parent[j]] = Fs[M->parent[j]] + Fs[j];
      return project(Fs);
    }
    M3 foot_R(int f) const { return R.oMi[M->foot_joint[f]].R; } // identity frame rotation (smpc_robot.h:14)

    // q (nq), v (nv), tau (nv - 6), contact mask
    void compute(const double * q, const double * v, const double * tau, unsigned mask)
    {
      R.fk(q);
      R.velocities(v);
      crba();
      nle = rnea(v, nullptr); // leaves R.acc = bias accelerations
      feet.clear();
      for (int f = 0; f < M->nfeet; f++)
        if ((mask >> f) & 1u)
          feet.push_back(f);
      const int nc = fs * (int)feet.size();
      Jc = Mat(nc, nv);
      gamma.assign(nc, 0.0);
      for (size_t c = 0; c < feet.size() && fs == 6; c++)
      {
        const int f = feet[c], j = M->foot_joint[f];
        for (int k = 0; k < nv; k++)
          if (R.is_ancestor_dof(k, j))
          {
            const V3 col = R.Jfoot_col(f, k);
            for (int i = 0; i < 3; i++)
            {
              Jc(6 * (int)c + i, k) = col[i];
              Jc(6 * (int)c + 3 + i, k) = R.S[k].a[i];
            }
          }
        const V3 p = R.foot_p[f];
        const V3 w = R.vel[j].a;
        const V3 vp = R.vel[j].l + cross(w, p);
        const V3 ap = R.acc[j].l + cross(R.acc[j].a, p) + cross(w, vp);
        const V3 rot = log3(foot_R(f));
        for (int i = 0; i < 3; i++)
        {
          gamma[6 * c + i] = ap[i] + Kd[i] * vp[i] + Kp[i] * p[i];
          gamma[6 * c + 3 + i] = R.acc[j].a[i] + Kd[3 + i] * w[i] + Kp[3 + i] * rot[i];
        }
      }
      for (size_t c = 0; c < feet.size() && fs == 3; c++)
      {
        const int f = feet[c], j = M->foot_joint[f];
        const M3 Rt = tr(foot_R(f));
        for (int k = 0; k < nv; k++)
        {
          const V3 col = Rt * R.Jfoot_col(f, k);
          for (int i = 0; i < 3; i++)
            Jc(3 * (int)c + i, k) = col[i];
        }
        const V3 p = R.foot_p[f];
        const V3 w = R.vel[j].a;
        const V3 vp = R.vel[j].l + cross(w, p);
        // classical acceleration of the body-fixed point at zero joint accelerations
        const V3 ap = R.acc[j].l + cross(R.acc[j].a, p) + cross(w, vp);
        const V3 drift = Rt * ap, verr = Rt * vp, perr = Rt * ((-1.0) * p);
        for (int i = 0; i < 3; i++)
          gamma[3 * c + i] = drift[i] + Kd[i] * verr[i] - Kp[i] * perr[i];
      }
      // M = L L^T ;  Minv_b = M^-1 (S tau - nle) ;  MJ = M^-1 J^T
      Mat L = Mq;
      cholesky(L);
      Lm = L;
      Vec b(nv);
      for (int k = 0; k < nv; k++)
        b[k] = (k >= 6 ? tau[k - 6] : 0.0) - nle[k];
      Vec Mb = b;
      chol_solve_inplace(L, Mb);
      lam.assign(nc, 0.0);
      prox_iters = 0;
      if (nc > 0)
      {
        Mat MJ(nv, nc);
        for (int k = 0; k < nv; k++)
          for (int c = 0; c < nc; c++)
            MJ(k, c) = Jc(c, k);
        chol_solve_inplace(L, MJ);
        Mat G = mul(Jc, MJ); // Delassus matrix
        for (int c = 0; c < nc; c++)
          G(c, c) += prox_mu;
        cholesky(G);
        Gc = G;
        this->MJ = MJ;
        const Vec JMb = mul(Jc, Mb);
        for (int it = 0; it < prox_max_iter; it++)
        {
          Vec rhs(nc);
          for (int c = 0; c < nc; c++)
            rhs[c] = prox_mu * lam[c] - gamma[c] - JMb[c];
          chol_solve_inplace(G, rhs);
          double diff = 0;
          for (int c = 0; c < nc; c++)
            diff = std::fmax(diff, std::fabs(rhs[c] - lam[c]));
          lam = rhs;
          prox_iters = it + 1;
          if (diff <= prox_accuracy)
            break;
        }
        a = Mb;
        for (int k = 0; k < nv; k++)
          for (int c = 0; c < nc; c++)
            a[k] += MJ(k, c) * lam[c];
      }
      else
        a = Mb;
    }

    bool anc_or_eq(int ja, int jb) const // joint ja is jb or one of its ancestors
    {
      for (int j = jb; j >= 0; j = M->parent[j])
        if (j == ja)
          return true;
      return false;
    }
    static SV mat_sv(const Mat & B, const SV & x) { return vec_sv(mul(B, sv_vec(x))); }

    // needs compute(); v = the same velocity
    void derivatives(const double * v)
    {
      const int nc = fs * (int)feet.size();
      const SV g{v3(gravity[0], gravity[1], gravity[2]), v3(0, 0, 0)};
      // body accelerations / forces at the solution, gravity included in the composite forces
      R.forces(v, a.data());
      R.compute_Bc();
      std::vector<SV> Fg(R.nj);
      for (int j = 0; j < R.nj; j++)
        Fg[j] = R.F[j] - R.I[j] * g;
      for (int j = R.nj - 1; j > 0; j--)
        Fg[M->parent[j]] = Fg[M->parent[j]] + Fg[j];
      std::vector<SV> dk(nv), Ak(nv);
      for (int k = 0; k < nv; k++)
      {
        const int lam_ = M->parent[R.dof2j[k]];
        dk[k] = lam_ >= 0 ? crm(R.vel[lam_], R.S[k]) : sv_zero();
        Ak[k] = (lam_ >= 0 ? crm(R.acc[lam_] - g, R.S[k]) + crm(R.vel[lam_], dk[k]) : crm(sv_zero() - g, R.S[k]));
      }
      dtau_dq = Mat(nv, nv);
      dtau_dv = Mat(nv, nv);
      for (int m = 0; m < nv; m++)
        for (int k = 0; k < nv; k++)
        {
          const int jm = R.dof2j[m], i = R.dof2j[k];
          int s;
          bool below = false; // joint(k) strictly below joint(m)
          if (anc_or_eq(i, jm))
            s = jm;
          else if (anc_or_eq(jm, i))
          {
            s = i;
            below = true;
          }
          else
            continue;
          SV Xq = R.Ic[s] * Ak[k] + mat_sv(R.Bc[s], dk[k]);
          if (below)
            Xq = Xq + crf(R.S[k], Fg[i]);
          const SV Xv = mat_sv(R.Bc[s], R.S[k]) + R.Ic[s] * (crm(R.vel[i], R.S[k]) + dk[k]);
          dtau_dq(m, k) = sv_dot(R.S[m], Xq);
          dtau_dv(m, k) = sv_dot(R.S[m], Xv);
        }
      // residual partials
      Mat r1q = dtau_dq, r1v = dtau_dv, r2q(nc, nv), r2v(nc, nv);
      for (size_t c = 0; c < feet.size() && fs == 6; c++)
      {
        const int f = feet[c], l = M->foot_joint[f];
        const V3 p = R.foot_p[f];
        const V3 fl = v3(lam[6 * c], lam[6 * c + 1], lam[6 * c + 2]), ft = v3(lam[6 * c + 3], lam[6 * c + 4], lam[6 * c + 5]);
        const SV W{fl, cross(p, fl) + ft}; // the contact wrench as a spatial force at the world origin
        const V3 w = R.vel[l].a;
        const V3 vp = R.vel[l].l + cross(w, p);
        const V3 ap = R.acc[l].l + cross(R.acc[l].a, p) + cross(w, vp); // at the solution
        const V3 al = R.acc[l].a;
        const V3 rotv = log3(foot_R(f));
        // d log3(R) for a rotation increment expressed in the world frame: inverse left Jacobian = Jlog3(-phi)
        const M3 Jl = Jlog3((-1.0) * rotv);
        for (int k = 0; k < nv; k++)
        {
          const int i = R.dof2j[k];
          if (!anc_or_eq(i, l))
            continue;
          const V3 vv = R.S[k].l + cross(R.S[k].a, p); // d(point position)/dq_k = d(point velocity)/dv_k
          // - d(J^T lam)/dq_k: the wrench keeps its world axes, its point of application moves; the columns S_m below joint(k)
          // (and the other base columns, for a base dof) move with S_k
          const SV U{v3(0, 0, 0), cross(vv, fl)};
          for (int m = 0; m < nv; m++)
          {
            const int jm = R.dof2j[m];
            if (!anc_or_eq(jm, l))
              continue;
            double t = sv_dot(R.S[m], U);
            const bool moves = (jm != i && anc_or_eq(i, jm)) || (jm == 0 && i == 0);
            if (moves)
              t -= sv_dot(R.S[m], crf(R.S[k], W));
            r1q(m, k) -= t;
          }
          const SV & d = dk[k];
          const int lam_ = M->parent[i];
          const SV A = lam_ >= 0 ? crm(R.acc[lam_], R.S[k]) + crm(R.vel[lam_], d) + crm(d, R.vel[l]) : sv_zero();
          const V3 aq = A.l + cross(A.a, p) + cross(d.a, vp) + cross(w, d.l + cross(d.a, p));
          const SV Av = d + crm(R.S[k], R.vel[l] - R.vel[i]);
          const V3 av = Av.l + cross(Av.a, p) + cross(R.S[k].a, vp) + cross(w, vv);
          const V3 vq = d.l + cross(d.a, p);
          const V3 sa = R.S[k].a;
          const V3 lq = aq + cross(sa, ap), lvq = vq + cross(sa, vp);
          const V3 aqq = A.a + cross(sa, al), wq = d.a + cross(sa, w), rq = Jl * sa;
          for (int r = 0; r < 3; r++)
          {
            r2q(6 * (int)c + r, k) = lq[r] + Kd[r] * lvq[r] + Kp[r] * vv[r];
            r2q(6 * (int)c + 3 + r, k) = aqq[r] + Kd[3 + r] * wq[r] + Kp[3 + r] * rq[r];
            r2v(6 * (int)c + r, k) = av[r] + Kd[r] * vv[r];
            r2v(6 * (int)c + 3 + r, k) = Av.a[r] + Kd[3 + r] * sa[r];
          }
        }
      }
      for (size_t c = 0; c < feet.size() && fs == 3; c++)
      {
        const int f = feet[c], l = M->foot_joint[f];
        const M3 Rf = foot_R(f), Rt = tr(Rf);
        const V3 p = R.foot_p[f];
        const V3 lc = v3(lam[3 * c], lam[3 * c + 1], lam[3 * c + 2]);
        const V3 fw = Rf * lc;
        const SV W{fw, cross(p, fw)}; // the contact force as a spatial force at the world origin
        const V3 w = R.vel[l].a;
        const V3 vp = R.vel[l].l + cross(w, p);
        for (int k = 0; k < nv; k++)
        {
          const int i = R.dof2j[k];
          if (!anc_or_eq(i, l))
            continue;
          // - d(J^T lam)/dq_k: the force moves with the foot; only dofs above joint(k) see a change
          for (int m = 0; m < nv; m++)
          {
            const int jm = R.dof2j[m];
            if (jm != i && anc_or_eq(jm, i))
              r1q(m, k) -= sv_dot(R.S[m], crf(R.S[k], W));
          }
          // contact acceleration (classical, contact frame): rigid parts of the variation cancel in the local frame
          const SV & d = dk[k];
          const int lam_ = M->parent[i];
          // non-rigid part of the variation of the body acceleration: A_k + d_k x v_l (kinematic: no gravity here)
          const SV A = lam_ >= 0 ? crm(R.acc[lam_], R.S[k]) + crm(R.vel[lam_], d) + crm(d, R.vel[l]) : sv_zero();
          const V3 aq = A.l + cross(A.a, p) + cross(d.a, vp) + cross(w, d.l + cross(d.a, p));
          const SV Av = d + crm(R.S[k], R.vel[l] - R.vel[i]);
          const V3 av = Av.l + cross(Av.a, p) + cross(R.S[k].a, vp) + cross(w, R.S[k].l + cross(R.S[k].a, p));
          const V3 vq = d.l + cross(d.a, p);                      // d(point velocity)/dq_k, non-rigid part
          const V3 vv = R.S[k].l + cross(R.S[k].a, p);            // d(point velocity)/dv_k
          const V3 cq = Rt * aq, cv = Rt * av, eq = Rt * vq, ev = Rt * vv, pq = (-1.0) * (Rt * R.S[k].l);
          for (int r = 0; r < 3; r++)
          {
            r2q(3 * (int)c + r, k) = cq[r] + Kd[r] * eq[r] - Kp[r] * pq[r];
            r2v(3 * (int)c + r, k) = cv[r] + Kd[r] * ev[r];
          }
        }
      }
      // [M -J^T; J mu] [da; dlam] = -[r1; r2]   (the damped factorisation of the forward solve, as the reference's backend)
      auto solve = [&](const Mat & r1, const Mat & r2, Mat & da, Mat & dl) {
        const int n = r1.c;
        Mat Mr = r1; // M^-1 r1
        chol_solve_inplace(Lm, Mr);
        da = Mat(nv, n);
        dl = Mat(nc > 0 ? nc : 0, n);
        if (nc > 0)
        {
          Mat rhs = mul(Jc, Mr);
          for (int c = 0; c < nc; c++)
            for (int j = 0; j < n; j++)
              rhs(c, j) -= r2(c, j);
          chol_solve_inplace(Gc, rhs);
          dl = rhs;
        }
        for (int k = 0; k < nv; k++)
          for (int j = 0; j < n; j++)
          {
            double acc_ = -Mr(k, j);
            for (int c = 0; c < nc; c++)
              acc_ += MJ(k, c) * dl(c, j);
            da(k, j) = acc_;
          }
      };
      solve(r1q, r2q, da_dq, dlam_dq);
      solve(r1v, r2v, da_dv, dlam_dv);
      Mat r1t(nv, nu), r2t(nc, nu);
      for (int j = 0; j < nu; j++)
        r1t(6 + j, j) = -1.0;
      solve(r1t, r2t, da_dtau, dlam_dtau);
    }
  };
} // namespace orc
