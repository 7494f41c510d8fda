// smpc_full_kernels.h -- first device block of the full-dynamics model:
//   full_fd_body   constrained forward dynamics of a batch of states: what the reference obtains from
//                  pinocchio::constraintDynamics through Aligator's MultibodyConstraintFwdDynamics
//                  [REF src/fulldynamics.cpp:139], contacts as built in [REF src/fulldynamics.cpp:50-75]
//                  (CONTACT_3D, LOCAL, joint2 = universe with identity placement, Baumgarte corrector Kp / Kd),
//                  ProximalSettings(1e-9, 1e-10, 10) [REF :39], actuation [0; I] [REF :35-37].
// One wavefront per state.  On top of the kinematics / composite phases of the kinodynamics kernels (kino_tree_phases,
// KIN_ONLY): joint-space inertia M_kl = S_k . (Ic_j S_l), bias forces S_k . (Fc_j - Ic_j g), LOCAL contact Jacobian and
// drift, M = L L^T in LDS, M^-1 [S tau - nle | J^T], the damped Delassus matrix and its inverse, then the proximal
// iteration on the contact forces as a 12 x 12 matrix-vector product per sweep.  Derivatives and the OCP stage on top
// are next.
#pragma once
#include "smpc_kino_kernels.h"

namespace smpc
{
  template <class D>
  struct FullFdArgs
  {
    Buffers<D> b;          // only b.model is read
    const double * X;      // [n][NX] states (device)
    const double * tau;    // [n][NV - 6] joint torques (device)
    const unsigned * mask; // [n] contact bit per foot (device)
    double Kp[3], Kd[3];
    double prox_accuracy, prox_mu;
    int prox_max_iter;
    double * a_out;   // [n][NV]
    double * lam_out; // [n][3 NF]: contact forces ON the robot, contact frame, feet in contact first (in order), rest 0
    int * iters_out;  // [n] proximal iterations taken (may be null)
  };

  template <class D>
  struct FullFdLds
  {
    static constexpr int NV = D::NV, NCM = 3 * D::NF, NR = NCM + 1;
    // two blocks reused along the kernel (8 resident waves per CU need <= 20 KB with the evaluation scratch):
    //   M : joint-space inertia -> its Cholesky factor -> (dead after the W solve) the damped Delassus matrix / its factor
    //   J : Ic_j S_l (dead once M is formed) -> contact Jacobian (dead once G and J M^-1 b are formed) -> inverse of G
    double M[NV * NV];
    double J[NCM * NV];     // contact Jacobian, rows of absent contacts zero
    double W[NV * NR];      // [M^-1 (S tau - nle) | M^-1 J^T], row-major [NV][NR]
    SMPC_HD double * FS_() { return J; } // [NV * 6]
    SMPC_HD double * G_() { return M; }  // [NCM * NCM]
    SMPC_HD double * Gi_() { return J; } // [NCM * NCM]
    static_assert(NV * 6 <= NCM * NV && NCM * NCM <= NCM * NV && NCM * NCM <= NV * NV, "overlays");
    double gam[NCM], JMb[NCM], lam[NCM], rhs[NCM], dl[NCM];
    double tmp[NV];
    unsigned anc[D::NJ]; // bit a of anc[j]: joint a is j or one of its ancestors
  };

  template <class D>
  SMPC_DEV bool full_anc_or_eq(const DevModelSmall<D> & md, int ja, int jb) // joint ja is jb or one of its ancestors
  {
    for (int j = jb; j >= 0; j = md.parent[j])
      if (j == ja)
        return true;
    return false;
  }
  SMPC_HD double sv_dot(const SV & m, const SV & f) { return dot(m.l, f.l) + dot(m.a, f.a); }

  // in-place Cholesky of the N x N matrix A (row-major, stride N), lower factor; lane = row.  The lane keeps its row in
  // registers; the finished entries of row j are broadcast reads from LDS (all loads of a column step are independent)
  template <int NT, int N>
  SMPC_DEV void wave_cholesky(double * A, double * tmp)
  {
    SMPC_PLA(double, row, NT, N);
    SMPC_LANES(NT)
    {
      const int r = lane < N ? lane : 0;
#pragma unroll
      for (int k = 0; k < N; k++)
        SMPC_PLV(row)[k] = A[r * N + k];
    }
    SMPC_LANES_END_WAVE
#pragma unroll
    for (int j = 0; j < N; j++)
    {
      SMPC_LANES(NT)
      {
        double s = SMPC_PLV(row)[j];
#pragma unroll
        for (int k = 0; k < j; k++)
          s -= SMPC_PLV(row)[k] * A[j * N + k];
        if (lane >= j && lane < N)
          tmp[lane] = s;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        const double d = sqrt(tmp[j]);
        const double v = lane == j ? d : tmp[lane < N ? lane : 0] / d;
        SMPC_PLV(row)[j] = v;
        if (lane >= j && lane < N)
          A[lane * N + j] = v;
      }
      SMPC_LANES_END_WAVE
    }
  }
  // columns of X (N x nrhs, row stride ldx) <- (L L^T)^-1 X ; lane = column, held in registers
  template <int NT, int N>
  SMPC_DEV void wave_chol_solve(const double * L, double * X, int nrhs, int ldx)
  {
    SMPC_LANES(NT)
    {
      const int c = lane < nrhs ? lane : 0;
      double y[N];
#pragma unroll
      for (int i = 0; i < N; i++)
        y[i] = X[i * ldx + c];
#pragma unroll
      for (int i = 0; i < N; i++)
      {
        double s = y[i];
#pragma unroll
        for (int k = 0; k < i; k++)
          s -= L[i * N + k] * y[k];
        y[i] = s / L[i * N + i];
      }
#pragma unroll
      for (int i = N - 1; i >= 0; i--)
      {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < N; k++)
          s -= L[k * N + i] * y[k];
        y[i] = s / L[i * N + i];
      }
      if (lane < nrhs)
      {
#pragma unroll
        for (int i = 0; i < N; i++)
          X[i * ldx + lane] = y[i];
      }
    }
    SMPC_LANES_END_WAVE
  }

  template <class D>
  SMPC_DEV void full_fd_body(const FullFdArgs<D> & ka, int block)
  {
    typedef KinoScratch<D, false> KinoScratchT;
    typedef FullFdLds<D> FL;
    constexpr int NT = 64;
    constexpr int NX = D::NX, NU = D::NU, NV = D::NV, NF = D::NF, NCM = FL::NCM, NR = FL::NR;
    static_assert(NV <= NT && NR <= NT, "one row / column per lane");
    const int inst = block;
    const DevModel<D> & mg = *ka.b.model;
    SMPC_LDS(KinoScratchT, scs, 1);
    SMPC_LDS(FL, fls, 1);
    KinoScratchT & sc = scs[0];
    FL & s = fls[0];
    const unsigned mask = ka.mask[inst] & ((1u << NF) - 1u);
    StageIn<D> in;
    in.md = &mg;
    in.terminal = true;
    in.mask = 0u;
    in.u_ref = nullptr;
    in.x_tgt = mg.x_term;
    in.foot_ref = nullptr;
    SMPC_LANES(NT)
    {
      lanes_load_model<D, NT>(sc, &mg, lane);
      if (lane < NX)
        sc.x[lane] = ka.X[(size_t)inst * NX + lane];
      if (lane < NU)
        sc.u[lane] = 0.0;
    }
    SMPC_LANES_END_WAVE
    kino_tree_phases<D, false, true>(sc, in); // kinematics, S, velocities, bias accelerations, Ic / Fc (composites)
    const DevModelSmall<D> & md = sc.ml;

    // ---- Ic_j S_l ; bias forces ; right-hand side S tau - nle ----
    SMPC_LANES(NT)
    if (lane < NV)
    {
      const int l = lane, j = l < 6 ? 0 : l - 5;
      const SI Ic = ldsi(&sc.Ic[j * 10]);
      const SV Sl = ldsv(&sc.S[l * 6]);
      stsv(&s.FS_()[l * 6], Ic * Sl);
      const SV g{ld3(md.gravity), mk3(0, 0, 0)};
      const SV Fg = ldsv(&sc.Fc[j * 6]) - Ic * g; // uniform field: every body accelerates with -g relative to free fall
      const double nle = sv_dot(Sl, Fg);
      s.W[l * NR] = (l >= 6 ? ka.tau[(size_t)inst * (NV - 6) + l - 6] : 0.0) - nle;
    }
    else if (lane >= 32 && lane < 32 + D::NJ)
    {
      // ancestor sets, once: the entry loops below test a bit instead of walking the tree
      unsigned bits = 0u;
      for (int j = lane - 32; j >= 0; j = md.parent[j])
        bits |= 1u << j;
      s.anc[lane - 32] = bits;
    }
    SMPC_LANES_END_WAVE
    static_assert(D::NJ <= 32, "ancestor bit sets");
    // ---- joint-space inertia ----
    SMPC_LANES(NT)
    for (int idx = lane; idx < NV * NV; idx += NT)
    {
      const int k = idx / NV, l = idx % NV;
      const int jk = k < 6 ? 0 : k - 5, jl = l < 6 ? 0 : l - 5;
      double v = 0.0;
      if ((s.anc[jl] >> jk) & 1u)
        v = sv_dot(ldsv(&sc.S[k * 6]), ldsv(&s.FS_()[l * 6]));
      else if ((s.anc[jk] >> jl) & 1u)
        v = sv_dot(ldsv(&sc.S[l * 6]), ldsv(&s.FS_()[k * 6]));
      s.M[idx] = v;
    }
    SMPC_LANES_END_WAVE
    // ---- contact rows: feet in contact first, in order ----
    SMPC_LANES(NT)
    {
      for (int idx = lane; idx < NCM * NV; idx += NT)
        s.J[idx] = 0.0;
      if (lane < NCM)
      {
        s.gam[lane] = 0.0;
        s.lam[lane] = 0.0;
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    for (int idx = lane; idx < NF * NV; idx += NT)
    {
      const int f = idx / NV, k = idx % NV;
      const int jf = md.foot_joint[f], jk = k < 6 ? 0 : k - 5;
      if (((mask >> f) & 1u) && ((s.anc[jf] >> jk) & 1u))
      {
        const int c = __builtin_popcount(mask & ((1u << f) - 1u));
        const M3 Rt = transpose(ldm3(&sc.oR[jf * 9])); // foot frame rotation = joint rotation
        const SV Sk = ldsv(&sc.S[k * 6]);
        const V3 col = Rt * (Sk.l + cross(Sk.a, ld3(&sc.footp[f * 3])));
        s.J[(3 * c + 0) * NV + k] = col.x;
        s.J[(3 * c + 1) * NV + k] = col.y;
        s.J[(3 * c + 2) * NV + k] = col.z;
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < NF && ((mask >> lane) & 1u))
    {
      const int f = lane, jf = md.foot_joint[f];
      const int c = __builtin_popcount(mask & ((1u << f) - 1u));
      const M3 Rt = transpose(ldm3(&sc.oR[jf * 9]));
      const V3 p = ld3(&sc.footp[f * 3]);
      const SV v = ldsv(&sc.vel[jf * 6]), ab = ldsv(&sc.acc[jf * 6]);
      const V3 vp = v.l + cross(v.a, p);
      // classical acceleration of the body-fixed point at zero joint accelerations
      const V3 ap = ab.l + cross(ab.a, p) + cross(v.a, vp);
      const V3 drift = Rt * ap, verr = Rt * vp, perr = Rt * ((-1.0) * p);
      s.gam[3 * c + 0] = drift.x + ka.Kd[0] * verr.x - ka.Kp[0] * perr.x;
      s.gam[3 * c + 1] = drift.y + ka.Kd[1] * verr.y - ka.Kp[1] * perr.y;
      s.gam[3 * c + 2] = drift.z + ka.Kd[2] * verr.z - ka.Kp[2] * perr.z;
    }
    SMPC_LANES_END_WAVE
    // ---- M = L L^T ; W = M^-1 [S tau - nle | J^T] ----
    wave_cholesky<NT, NV>(s.M, s.tmp);
    SMPC_LANES(NT)
    for (int idx = lane; idx < NV * NCM; idx += NT)
    {
      const int k = idx / NCM, c = idx % NCM;
      s.W[k * NR + 1 + c] = s.J[c * NV + k];
    }
    SMPC_LANES_END_WAVE
    wave_chol_solve<NT, NV>(s.M, s.W, NR, NR);
    // ---- damped Delassus matrix (unit diagonal on the rows of absent contacts), its inverse, J M^-1 b ----
    const int nc = 3 * __builtin_popcount(mask);
    SMPC_LANES(NT)
    {
      for (int idx = lane; idx < NCM * NCM; idx += NT)
      {
        const int c = idx / NCM, d = idx % NCM;
        double acc = 0.0;
        for (int k = 0; k < NV; k++)
          acc += s.J[c * NV + k] * s.W[k * NR + 1 + d];
        if (c == d)
          acc += c < nc ? ka.prox_mu : 1.0;
        s.G_()[idx] = acc;
      }
      if (lane < NCM)
      {
        double acc = 0.0;
        for (int k = 0; k < NV; k++)
          acc += s.J[lane * NV + k] * s.W[k * NR];
        s.JMb[lane] = acc;
      }
    }
    SMPC_LANES_END_WAVE
    // (the Jacobian is dead now: its block takes the inverse)
    SMPC_LANES(NT)
    for (int idx = lane; idx < NCM * NCM; idx += NT)
      s.Gi_()[idx] = idx / NCM == idx % NCM ? 1.0 : 0.0;
    SMPC_LANES_END_WAVE
    wave_cholesky<NT, NCM>(s.G_(), s.tmp);
    wave_chol_solve<NT, NCM>(s.G_(), s.Gi_(), NCM, NCM);
    // ---- proximal iteration:  lam <- G^-1 (mu lam - gamma - J M^-1 b)  until |d lam|_inf <= accuracy ----
    int iters = 0;
    if (nc > 0)
      for (int it = 0; it < ka.prox_max_iter; it++)
      {
        SMPC_LANES(NT)
        if (lane < NCM)
          s.rhs[lane] = lane < nc ? ka.prox_mu * s.lam[lane] - s.gam[lane] - s.JMb[lane] : 0.0;
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        if (lane < NCM)
        {
          double acc = 0.0;
          for (int d = 0; d < NCM; d++)
            acc += s.Gi_()[lane * NCM + d] * s.rhs[d];
          s.dl[lane] = fabs(acc - s.lam[lane]);
          s.lam[lane] = acc;
        }
        SMPC_LANES_END_WAVE
        iters = it + 1;
        double diff = 0.0; // wave-uniform: every lane reads the same values
        for (int c = 0; c < NCM; c++)
          diff = fmax(diff, s.dl[c]);
        if (diff <= ka.prox_accuracy)
          break;
      }
    // ---- a = M^-1 (S tau - nle) + M^-1 J^T lam ----
    SMPC_LANES(NT)
    {
      if (lane < NV)
      {
        double acc = s.W[lane * NR];
        for (int c = 0; c < NCM; c++)
          acc += s.W[lane * NR + 1 + c] * s.lam[c];
        ka.a_out[(size_t)inst * NV + lane] = acc;
      }
      if (lane < NCM)
        ka.lam_out[(size_t)inst * NCM + lane] = s.lam[lane];
      if (lane == 0 && ka.iters_out != nullptr)
        ka.iters_out[inst] = iters;
    }
    SMPC_LANES_END_WAVE
  }
} // namespace smpc
