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
// Derivatives (d a / d(q, v, tau), d lam / d(q, v, tau)) and the OCP on top are the next block.
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
    double Kp[3] = {0, 0, 0}, Kd[3] = {0, 0, 0};
    // results of the last call
    Mat Mq, Jc;          // nv x nv ; 3 n_c x nv
    Vec nle, gamma;      // nv ; 3 n_c
    Vec a, lam;          // nv ; 3 n_c (force ON the robot at the foot, contact frame)
    int prox_iters = 0;
    std::vector<int> feet; // feet in contact, in order

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
      const int nc = 3 * (int)feet.size();
      Jc = Mat(nc, nv);
      gamma.assign(nc, 0.0);
      for (size_t c = 0; c < feet.size(); c++)
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
  };
} // namespace orc
