// smpc_full_stage.h -- the per-(instance, stage) body of the FULL-DYNAMICS OCP: one 64-lane wavefront evaluates the
// stage the reference builds in FullDynamicsOCP::createStage (src/fulldynamics.cpp:78-214):
//   HOT(1) evaluate    constrained forward dynamics (MultibodyConstraintFwdDynamics, contacts of src/fulldynamics.cpp:50-75,
//                      ProximalSettings(1e-9, 1e-10, 10) :39) + IntegratorSemiImplEuler (:139-140), the cost stack (:88-137:
//                      state, control, centroidal momentum, foot pose per foot, contact force per foot in contact), the
//                      torque / joint boxes (:144-162)
//   HOT(2) derivatives d a / d(q, v, tau), d lambda / d(q, v, tau) by implicit differentiation of the contact KKT system
//                      (what pinocchio::computeConstraintDynamicsDerivatives provides), A, B, Gauss-Newton Hessians
//   HOT(3) LQ assembly knot (A, B, Q, S, R, q, r, f, d, ...) straight to HBM
// World-frame spatial formulation (the one of smpc_kino_stage.h, DESIGN.md "Rigid-body derivatives"):
//   d_k = v_lam x S_k,  A_k = (a_lam - g) x S_k + v_lam x d_k
//   d tau_m / d q_k = S_m . (Ic_s A_k + Bc_s d_k) [+ S_m . (S_k x* (Fc_i - contact wrenches below i)) if joint(m) is a strict
//                     ancestor of joint(k) = i],   s = the lower of joint(m), joint(k) (same branch)
//   d tau_m / d v_k = S_m . (Bc_s S_k + Ic_s (v_i x S_k + d_k))
// Generic in the robot shape (FullDims<NJ, NF, FS>): every lane map is a strided loop, nothing assumes NV <= 20.
#pragma once
#include "smpc_full_model.h"
#include "smpc_riccati_kino.h" // wave_block_sweep, tix; lanes_integrate / lanes_difference, StageKernelArgs
#include <cstddef>

namespace smpc
{
  SMPC_HD int jof(int k) { return k < 6 ? 0 : k - 5; } // joint of dof k

  // optional in-kernel phase timers (SMPC_PHASE_PROFILE=1): cycles since the previous tick accumulate into prof[slot]
  struct FullProf
  {
    double * prof = nullptr;
    long long tprev = 0;
  };
  SMPC_DEV void ftick(FullProf & fp, int slot)
  {
    if (fp.prof)
      prof_tick(fp.prof, slot, fp.tprev);
  }

  // The two widest derivative blocks of a stage -- R1 (NV x NCOL) and the force rows JT (NCM x NCOL) of the Gauss-Newton Jacobian.  They are
  // operands / results of the column-wise solve chain.  Where D::WIDE_DEV says so they live in a per-block slice of DEVICE MEMORY instead of LDS
  // (round 5): LDS, not registers, decides the occupancy of this kernel (one wavefront per block), and for the biped these 24.4 KB are the
  // difference between two and three resident blocks per CU.  A block writes and re-reads its slice within its lifetime (L2 of its XCD); the
  // accesses are kept latency-tolerant: operand fetches of the matrix-core products run one K-step ahead, read-modify-write passes are
  // products that START from the old value (fwave_gemm's `init`), columns are read into registers before anything is stored.
  template <class D>
  struct FullDerivWide
  {
    static constexpr int NV = D::NV, NCM = D::NCM, NU = D::NU;
    static constexpr int NCOL = 2 * NV + NU;
    // kinodynamics variant: only the six base rows are solved for (rows 6 .. of [da_dq | da_dv | da_du] are the unit block of the joint
    // accelerations in u: r1())
    static constexpr int R1ROWS = D::KINO ? 6 : NV;
    double R1[R1ROWS * NCOL];   // [r1q | r1v | r1t] -> M^-1 R1 -> [da_dq | da_dv | da_dtau]
    SMPC_HD double r1(int i, int cc) const
    {
      if constexpr (D::KINO)
        return i < 6 ? R1[i * NCOL + cc] : (cc == 2 * NV + NCM + i - 6 ? 1.0 : 0.0);
      else
        return R1[i * NCOL + cc];
    }
    // force rows of the stacked Gauss-Newton Jacobian: [r2q | r2v | 0] -> [dlam_dq | dlam_dv | dlam_dtau]
    // kinodynamics variant: rows 0 .. 5 = Jacobian of the centroidal_derivative residual hdot(u, q), the rest zero
    // In LDS (WIDE_DEV = false) the momentum / foot-pose rows live in the dead dynamics block of the evaluation scratch (FullScratch::jt2_());
    // in the device slice they follow the force rows (round 6: the dynamics block of such a block is no longer its own LDS, see FullScratch).
    static constexpr int NGN = 6 + D::PF * D::NF + NCM;
    static constexpr int JTROWS = D::WIDE_DEV ? NGN : NCM;
    double JT[JTROWS * NCOL];
    // Round 6, WIDE_DEV: the factorised dynamics block [M^-1 | J | W | G^-1] of the stage, copied out of LDS once the accelerations and forces
    // are solved -- its only readers afterwards are operand fetches of the derivative solve chain (latency-tolerant, two K-steps ahead), and
    // its LDS then holds the derivative scratch: 53.0 -> 40.0 KB per block, four resident blocks per CU instead of three.
    // Full dynamics: what goes out is the TRANSPOSE of the (NV + NCM)^2 inverse of the contact KKT matrix (full_kkt_inv; the operand fetch of the
    // solve then reads 16 consecutive entries per K index); kinodynamics variant: the block as it lies ([M | J | W | Gi], its six base rows are used).
    static constexpr int DYN4 = NV * NV + NCM * NV + NV * (NCM + 1) + NCM * NCM, NK = NV + NCM;
    static constexpr int DYN_OUT_DOUBLES = D::KINO ? DYN4 : NK * NK;
    double dyn[D::WIDE_DEV ? DYN_OUT_DOUBLES : 2];
    double Cv[D::KINO ? D::NVEL * D::NDX : 2]; // kinodynamics variant: the frame-velocity rows (NVEL x NDX) of the feet in contact
    SMPC_HD const double * Mi_() const { return dyn; }
    SMPC_HD const double * J_() const { return dyn + NV * NV; }
    SMPC_HD const double * W_() const { return dyn + NV * NV + NCM * NV; }
    SMPC_HD const double * Gi_() const { return dyn + NV * NV + NCM * NV + NV * (NCM + 1); }
  };
  // LDS part of the wide blocks: the blocks themselves (WIDE_DEV = false), or the per-dof vectors of the R1 fill, which otherwise borrow the
  // force rows of JT (strided per-lane accesses: not for device memory)
  template <class D, bool DEV = D::WIDE_DEV>
  struct FullDerivWideLds
  {
    FullDerivWide<D> w;
    SMPC_HD double * tmp_(FullDerivWide<D> & sw) { return sw.JT; }
  };
  template <class D>
  struct FullDerivWideLds<D, true>
  {
    static constexpr int NV = D::NV, NF = D::NF, FS = D::FS;
    static constexpr int MR = FullDerivWide<D>::R1ROWS, N1 = (MR + 2 * NV) * 6, N2 = FS == 6 ? (6 * MR + 3 * NV) * NF : 0;
    double tmp[N1 > N2 ? N1 : N2];
    SMPC_HD double * tmp_(FullDerivWide<D> &) { return tmp; }
  };
  template <class D>
  struct FullScratchDeriv
  {
    static constexpr int NV = D::NV, NJ = D::NJ, NF = D::NF, NCM = D::NCM, NU = D::NU, NDX = D::NDX;
    static constexpr int NCOL = 2 * NV + NU, NGN = 6 + D::PF * NF + NCM;
    double Bc[NJ * 36], dk[NV * 6], Ak[NV * 6], Wc[NF * 6];
    double Je3[9], JeQ[9], Jq[36], Jl[36];
    double Jlf[D::FS == 6 ? NF * 36 : 2]; // Jlog6 of the foot-placement residuals (6-D feet)
    // tables of the assembly phases: they live in the composite velocity-product matrices, dead once R1 is formed
    //   WJl (NDX x 6), JWJ (36): state-cost tables ; gx, gu: cost gradients ; cq: (C_x^T nu ; C_u^T nu) of the cone rows ;
    //   yc: A_cone^T nu per contact ; dual: stacked weighted residual
    SMPC_HD double * WJl_() const { return const_cast<double *>(Bc); }
    SMPC_HD double * JWJ_() const { return WJl_() + NDX * 6; }
    SMPC_HD double * gx_() const { return JWJ_() + 36; }
    SMPC_HD double * gu_() const { return gx_() + NDX; }
    SMPC_HD double * cq_() const { return gu_() + NU; }
    SMPC_HD double * yc_() const { return cq_() + NDX + NU; }
    SMPC_HD double * dual_() const { return yc_() + NCM + 1; }
    static_assert(NDX * 6 + 36 + NDX + NU + NDX + NU + NCM + 1 + NGN <= NJ * 36, "assembly tables fit the dead Bc block");
  };

  template <class D, bool DERIV>
  struct FullScratch
  {
    static constexpr int NV = D::NV, NJ = D::NJ, NF = D::NF, NCM = D::NCM, NU = D::NU, NDX = D::NDX, NX = D::NX, NC = D::NC;
    static constexpr int NR = NCM + 1;           // columns of W = [M^-1 b | M^-1 J^T]
    static constexpr int NCOL = 2 * NV + NU;     // derivative columns (q | v | tau)
    static constexpr int NGN = 6 + D::PF * NF + NCM; // rows of the stacked Gauss-Newton Jacobian: forces | momentum | foot poses
    FullHead<D> h;
    // block inputs
    double x[NX], u[NU], xn1[NX], x_tgt[NX], u_ref[NU], f_ref[NCM], foot_ref[NF * 3];
    double lam_next[NDX], nu[NC];
    // kinematics
    double oR[NJ * 9], op[NJ * 3], S[NV * 6], vel[NJ * 6], acc[NJ * 6], I[NJ * 10], Ic[NJ * 10], hc[NJ * 6], Fc[NJ * 6];
    double footp[NF * 3], com[3], hg[6];
    double IcS[NV * 6];
    double gam[NCM], JMb[NCM], lam[NCM], rhs[NCM], dl[NCM];
    double rotl[D::FS == 6 ? NF * 3 : 1]; // log3 of the foot rotations (6-D contacts: angular part of the corrector)
    static constexpr int NVP = ((NV + 3) / 4) * 4, NCP = ((NCM + 3) / 4) * 4;
    static constexpr int NTM = (2 * NVP + 15) / 16, NTG = (2 * NCP + 15) / 16; // tile grids of the two bordered inverses
    static constexpr int SWP_DOUBLES = 2 * 4 * 16 * (NTM > NTG ? NTM : NTG);     // sweep operands
    double a[NV];
    // ---- "late block": written only after the dynamics phases (integration, residuals, multipliers, reductions); until then its
    //      head serves as the operand scratch of the two bordered inverses, see swp_() ----
    double xnext[NX], e[NDX];
    // costs / constraints / multipliers
    double rx[NDX], Wrx[NDX], ru[NU], Wru[NU], Whg[6], rf[NF * D::PF], Wrf[NF * D::PF], rl[NCM], Wrl[NCM];
    double cval[NC], vplus[NC], lamp[NDX];
    int act[NC + NC % 2];
    double part[64], part8[16], red[4];
    int iters_[2];
    SMPC_HD double * swp_() { return xnext; } // (size checked where the inverses are called)
    // ---- constrained dynamics: one contiguous block at the END of the scratch ----
    // WIDE_DEV = false: dead after the derivative solves, it then takes the momentum / foot-pose rows of the stacked Gauss-Newton Jacobian
    // (jt2_()).  Derivative kernel with WIDE_DEV (round 6, DYN_OUT): once a and lam are solved the block is copied to the device slice
    // (FullDerivWide::dyn) and its LDS becomes the derivative scratch (FullScratchDeriv + the per-dof vectors of the R1 fill: dyn_overlay_()),
    // which is why the composite velocity-product matrices Bc of such a block are formed after the dynamics phases, not with the other composites.
    static constexpr bool DYN_OUT = DERIV && D::WIDE_DEV;
    static constexpr int DYN_DOUBLES = NV * NV + NCM * NV + NV * NR + NCM * NCM;
    static constexpr int OVL_DOUBLES = (int)((sizeof(FullScratchDeriv<D>) + sizeof(FullDerivWideLds<D, true>)) / sizeof(double));
    static constexpr int DYN_PAD = (DYN_OUT && OVL_DOUBLES > DYN_DOUBLES) ? OVL_DOUBLES - DYN_DOUBLES : 0;
    static_assert(DYN_OUT || DYN_DOUBLES >= (NGN - NCM) * NCOL, "the momentum / pose rows of the Gauss-Newton Jacobian fit the dead dynamics block");
    double M[NV * NV];    // joint-space inertia -> its inverse
    double J[NCM * NV];   // contact Jacobian (rows of absent contacts zero)
    double W[NV * NR];    // [M^-1 (S tau - nle) | M^-1 J^T]
    double Gi[NCM * NCM + DYN_PAD]; // damped Delassus matrix -> its inverse (+ what the overlay needs beyond the block)
    SMPC_HD double * jt2_() { return M; } // (not DYN_OUT) rows NCM .. NGN-1 of the stacked Jacobian, [NGN - NCM][NCOL]
    SMPC_HD double * dyn_overlay_() { return M; }
  };

  // copy the head of the device model into LDS (all loads in flight before the first store)
  template <class D, int NT>
  SMPC_DEV void full_load_head(FullHead<D> & dst, const DevModel<D> * gm, int lane)
  {
    constexpr int N = (int)(sizeof(FullHead<D>) / sizeof(double)), PER = (N + NT - 1) / NT;
    static_assert(sizeof(FullHead<D>) % sizeof(double) == 0, "LDS copy is done in doubles");
    const alias_double * src = reinterpret_cast<const alias_double *>(static_cast<const FullHead<D> *>(gm));
    alias_double * d = reinterpret_cast<alias_double *>(&dst);
    double r[PER];
#pragma unroll
    for (int n = 0; n < PER; n++)
      r[n] = src[lane + n * NT < N ? lane + n * NT : 0];
#pragma unroll
    for (int n = 0; n < PER; n++)
      if (lane + n * NT < N)
        d[lane + n * NT] = r[n];
  }

  SMPC_HD double sv_dot6(const SV & m, const SV & f) { return dot(m.l, f.l) + dot(m.a, f.a); }

  // in-place lower Cholesky of the N x N matrix A (row-major), lane = row held in registers, finished rows broadcast from LDS
  template <int NT, int N>
  SMPC_DEV void fwave_cholesky(double * A, double * tmp)
  {
    static_assert(N <= NT, "one row per lane");
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
  // columns c0 .. c0 + NT - 1 (< ncols) of X (N rows, row stride ldx) <- (L L^T)^-1 X ; lane = column, held in registers
  template <int NT, int N>
  SMPC_DEV void fwave_chol_solve(const double * L, double * X, int ncols, int ldx)
  {
    for (int c0 = 0; c0 < ncols; c0 += NT)
    {
      SMPC_LANES(NT)
      {
        const int c = c0 + lane < ncols ? c0 + lane : c0;
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
        if (c0 + lane < ncols)
        {
#pragma unroll
          for (int i = 0; i < N; i++)
            X[i * ldx + c] = y[i];
        }
      }
      SMPC_LANES_END_WAVE
    }
  }

  // C (M x N) = init(i, j) + sum_k a(i, k) b(k, j) on the FP64 matrix cores, one wave; a / b read their operand entries (LDS, or device memory: the
  // fetch of K-step k + 1 is in flight during step k), store(i, j, v) receives every entry once after the last K-step (so C may overwrite an
  // operand).  K-steps of 4, 16 x 16 tiles.  `init` makes a read-modify-write pass one product: all its reads are issued before the first K-step.
  // PF = K-steps the operand fetch runs ahead (1: an LDS round trip is shorter than a K-step's matrix instructions; 2: operands in device memory --
  // a K-step of 10 instructions is 640 cycles, an L2 round trip more).
  template <int M, int N, int K, int PF = 1, class FA, class FB, class FI, class FS_>
  SMPC_DEV void fwave_gemm(FA a, FB b, FI init, FS_ store);
  template <int M, int N, int K, int PF = 1, class FA, class FB, class FS_>
  SMPC_DEV void fwave_gemm(FA a, FB b, FS_ store)
  {
    fwave_gemm<M, N, K, PF>(a, b, [](int, int) { return 0.0; }, store);
  }
  template <int M, int N, int K, int PF, class FA, class FB, class FI, class FS_>
  SMPC_DEV void fwave_gemm(FA a, FB b, FI init, FS_ store)
  {
    static_assert(PF >= 1 && PF <= (K + 3) / 4, "operand sets");
    constexpr int NS = PF + 1;
    constexpr int NT = 64, TI = (M + 15) / 16, TJ = (N + 15) / 16, KS = (K + 3) / 4;
    SMPC_ACC(acc, NT, TI * TJ);
    // two operand sets: the LDS reads of K-step k + 1 are issued before the matrix instructions of step k, so that with one
    // resident wave per SIMD the matrix pipe does not idle through every LDS round trip
    SMPC_PLA(double, av, NT, NS * TI);
    SMPC_PLA(double, bv, NT, NS * TJ);
    auto fetch = [&](int ks) {
      const int ob = ks % NS;
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
        const int k = 4 * ks + lr;
#pragma unroll
        for (int I = 0; I < TI; I++)
        {
          const int i = 16 * I + lc;
          const bool ok = i < M && k < K;
          const double x = a(ok ? i : 0, ok ? k : 0);
          SMPC_PLV(av)[ob * TI + I] = (PF > 1 || ok) ? x : 0.0; // (PF > 1: masked at use, see mask())
        }
#pragma unroll
        for (int J = 0; J < TJ; J++)
        {
          const int j = 16 * J + lc;
          const bool ok = j < N && k < K;
          const double x = b(ok ? k : 0, ok ? j : 0);
          SMPC_PLV(bv)[ob * TJ + J] = (PF > 1 || ok) ? x : 0.0;
        }
      }
      SMPC_LANES_END_WAVE
    };
    // operands in device memory: the entries outside the matrices are zeroed when the K-step USES them, not when it fetches them -- a select on
    // a loaded value is a wait for the load
    auto mask = [&](int ks) {
      const int ob = ks % NS;
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
        const int k = 4 * ks + lr;
#pragma unroll
        for (int I = 0; I < TI; I++)
          if (16 * I + 16 > M || 4 * ks + 4 > K)
            SMPC_PLV(av)[ob * TI + I] = (16 * I + lc < M && k < K) ? SMPC_PLV(av)[ob * TI + I] : 0.0;
#pragma unroll
        for (int J = 0; J < TJ; J++)
          if (16 * J + 16 > N || 4 * ks + 4 > K)
            SMPC_PLV(bv)[ob * TJ + J] = (16 * J + lc < N && k < K) ? SMPC_PLV(bv)[ob * TJ + J] : 0.0;
      }
      SMPC_LANES_END_WAVE
    };
    SMPC_LANES(NT)
    {
      const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
      for (int I = 0; I < TI; I++)
#pragma unroll
        for (int J = 0; J < TJ; J++)
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            const int i = 16 * I + lr + 4 * v, j = 16 * J + lc;
            const bool ok = i < M && j < N;
            const double x = init(ok ? i : 0, ok ? j : 0);
            SMPC_ACCV(acc, I * TJ + J, v) = ok ? x : 0.0;
          }
    }
    SMPC_LANES_END_WAVE
#pragma unroll
    for (int p = 0; p < PF && p < KS; p++)
      fetch(p);
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
    {
      if (ks + PF < KS)
        fetch(ks + PF);
      // (the scheduler otherwise sinks the loads of the fetch below the matrix instructions, next to their use: the operands of the biped's
      //  blocks lie in device memory and every K-step then waited for its own loads)
      if constexpr (PF > 1)
      {
        SMPC_SCHED_FENCE();
        mask(ks);
      }
      const int ob = ks % NS;
#pragma unroll
      for (int I = 0; I < TI; I++)
#pragma unroll
        for (int J = 0; J < TJ; J++)
          SMPC_MFMA(acc, I * TJ + J, av, ob * TI + I, bv, ob * TJ + J);
      if constexpr (PF > 1)
        SMPC_SCHED_FENCE();
    }
    SMPC_LANES(NT)
    {
      const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
      for (int I = 0; I < TI; I++)
#pragma unroll
        for (int J = 0; J < TJ; J++)
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            const int i = 16 * I + lr + 4 * v, j = 16 * J + lc;
            if (i < M && j < N)
              store(i, j, SMPC_ACCV(acc, I * TJ + J, v));
          }
    }
    SMPC_LANES_END_WAVE
  }

  // (fwave_spd_inverse: smpc_riccati_kino.h, beside the block sweep it is made of)

  // -------------------------------------------------------------------------------------------------------------
  // Kinematics, composites, joint-space inertia, contact rows, factorisations, proximal iteration, accelerations.
  // dyn = false: kinematics / momentum only (terminal node).
  // -------------------------------------------------------------------------------------------------------------
  // per-body velocity-product matrices  B_l y = v_l x* (I_l y) - I_l (v_l x y)  (lane = (body, column)); in: world inertias I (NOT yet the
  // composite forces the derivative phases put in their block), body velocities
  template <class D, class SC, class SD>
  SMPC_DEV void full_bc_bodies(SC & sc, SD & sd)
  {
    constexpr int NT = 64, NJ = D::NJ;
    SMPC_LANES(NT)
    for (int idx = lane; idx < NJ * 6; idx += NT)
    {
      const int l = idx / 6, m = idx % 6;
      const SI Il = ldsi(&sc.I[l * 10]);
      const SV vl = ldsv(&sc.vel[l * 6]);
      const V3 e = mk3(m % 3 == 0, m % 3 == 1, m % 3 == 2), z = mk3(0, 0, 0);
      const SV y = m < 3 ? SV{e, z} : SV{z, e};
      const SV col = crf(vl, Il * y) - Il * crm(vl, y);
      double * dst = &sd.Bc[l * 36 + m];
      dst[0] = col.l.x;
      dst[6] = col.l.y;
      dst[12] = col.l.z;
      dst[18] = col.a.x;
      dst[24] = col.a.y;
      dst[30] = col.a.z;
    }
    SMPC_LANES_END_WAVE
  }
  // the same and their composites (leaf -> root), as a pass of its own: blocks whose dynamics block is staged out (FullScratch::DYN_OUT)
  template <class D, class SC, class SD>
  SMPC_DEV void full_bc_composites(SC & sc, SD & sd)
  {
    constexpr int NT = 64, NJ = D::NJ;
    full_bc_bodies<D>(sc, sd);
    SMPC_LANES(NT)
    if (lane < 36)
    {
      int par[NJ];
#pragma unroll
      for (int j = 1; j < NJ; j++)
        par[j] = sc.h.parent[j];
#pragma unroll
      for (int j = NJ - 1; j >= 1; j--)
        sd.Bc[par[j] * 36 + lane] += sd.Bc[j * 36 + lane];
    }
    SMPC_LANES_END_WAVE
  }

  template <class D, bool DERIV, class SC, class SD>
  SMPC_DEV void full_dynamics_phases(SC & sc, SD * sd, const DevModel<D> & mg, unsigned mask, bool dyn, FullProf & fp, bool rows_only = false)
  {
    constexpr int NT = 64;
    constexpr int NJ = D::NJ, NV = D::NV, NQ = D::NQ, NF = D::NF, NCM = D::NCM, NR = SC::NR, FS = D::FS;
    static_assert(NJ <= NT && NV <= NT && NR <= NT, "one joint / dof / right-hand side per lane");
    const FullHead<D> & h = sc.h;
    const int nlev = h.nlevels;
    const double * vq = &sc.x[NQ];
    SMPC_PLA(double, rl, NT, 9);
    // what a joint's lane needs in its pass of the level loop, read once (the loop then waits for its parent's entries only)
    SMPC_PLA(double, jc, NT, 4); // offset from the parent (3), joint velocity
    SMPC_PLA(int, ji, NT, 3);    // level, parent, axis
    // ---- joint-local transforms (all joints at once) ----
    SMPC_LANES(NT)
    if (lane < NJ)
    {
      const int j = lane;
      SMPC_PLV(ji)[0] = j > 0 ? h.level[j] : 0;
      SMPC_PLV(ji)[1] = j > 0 ? h.parent[j] : 0;
      SMPC_PLV(ji)[2] = j > 0 ? h.jtype[j] - 1 : 0;
      SMPC_PLV(jc)[0] = mg.jpp[j][0];
      SMPC_PLV(jc)[1] = mg.jpp[j][1];
      SMPC_PLV(jc)[2] = mg.jpp[j][2];
      SMPC_PLV(jc)[3] = j > 0 ? vq[j + 5] : 0.0;
      if (j == 0)
      {
        const M3 R = quat_to_R(Quat{sc.x[3], sc.x[4], sc.x[5], sc.x[6]});
        const V3 p = ld3(sc.x);
        SV v = sv0();
        for (int k = 0; k < 6; k++)
        {
          const int col = k % 3;
          const V3 ax = col == 0 ? mk3(R.a00, R.a10, R.a20) : (col == 1 ? mk3(R.a01, R.a11, R.a21) : mk3(R.a02, R.a12, R.a22));
          const SV sk = k < 3 ? SV{ax, mk3(0, 0, 0)} : SV{cross(p, ax), ax};
          stsv(&sc.S[k * 6], sk);
          v = v + vq[k] * sk;
        }
        stm3(&sc.oR[0], R);
        st3(&sc.op[0], p);
        stsv(&sc.vel[0], v);
        stsv(&sc.acc[0], sv0());
      }
      else
      {
        double s, c;
        sincos(sc.x[6 + j], &s, &c);
        const int jt = h.jtype[j];
        const M3 Rq = jt == 1 ? M3{1, 0, 0, 0, c, -s, 0, s, c} : (jt == 2 ? M3{c, 0, s, 0, 1, 0, -s, 0, c} : M3{c, -s, 0, s, c, 0, 0, 0, 1});
        stm3(SMPC_PLV(rl), ldm3(mg.jpR[j]) * Rq);
      }
    }
    SMPC_LANES_END_WAVE
    // ---- root -> leaf: placement, motion column, velocity, bias acceleration ----
    for (int lvl = 1; lvl < nlev; lvl++)
    {
      SMPC_LANES(NT)
      if (lane > 0 && lane < NJ && SMPC_PLV(ji)[0] == lvl)
      {
        const int j = lane, par = SMPC_PLV(ji)[1];
        const M3 Rp = ldm3(&sc.oR[par * 9]);
        const M3 R = Rp * ldm3(SMPC_PLV(rl));
        const V3 p = ld3(&sc.op[par * 3]) + Rp * mk3(SMPC_PLV(jc)[0], SMPC_PLV(jc)[1], SMPC_PLV(jc)[2]);
        const int col = SMPC_PLV(ji)[2];
        const V3 ax = col == 0 ? mk3(R.a00, R.a10, R.a20) : (col == 1 ? mk3(R.a01, R.a11, R.a21) : mk3(R.a02, R.a12, R.a22));
        const SV sk = SV{cross(p, ax), ax};
        const SV vp = ldsv(&sc.vel[par * 6]);
        const double qd = SMPC_PLV(jc)[3];
        stm3(&sc.oR[j * 9], R);
        st3(&sc.op[j * 3], p);
        stsv(&sc.S[(j + 5) * 6], sk);
        stsv(&sc.vel[j * 6], vp + qd * sk);
        stsv(&sc.acc[j * 6], ldsv(&sc.acc[par * 6]) + qd * crm(vp, sk));
      }
      SMPC_LANES_END_WAVE
    }
    // ---- world inertias, momenta, bias forces (all joints at once); foot positions ----
    SMPC_LANES(NT)
    if (lane < NJ)
    {
      const int j = lane;
      const M3 R = ldm3(&sc.oR[j * 9]);
      const V3 p = ld3(&sc.op[j * 3]);
      const SV v = ldsv(&sc.vel[j * 6]), a = ldsv(&sc.acc[j * 6]);
      const double m = mg.mass[j];
      const V3 c = R * ld3(mg.com[j]) + p;
      const double * il = mg.inertia[j];
      const M3 Il = M3{il[0], il[1], il[3], il[1], il[2], il[4], il[3], il[4], il[5]};
      const M3 Iw = R * Il * transpose(R);
      const double cc = dot(c, c);
      SI I;
      I.m = m;
      I.mc = m * c;
      I.jxx = Iw.a00 + m * (cc - c.x * c.x);
      I.jxy = Iw.a01 - m * c.x * c.y;
      I.jxz = Iw.a02 - m * c.x * c.z;
      I.jyy = Iw.a11 + m * (cc - c.y * c.y);
      I.jyz = Iw.a12 - m * c.y * c.z;
      I.jzz = Iw.a22 + m * (cc - c.z * c.z);
      stsi(&sc.I[j * 10], I);
      stsi(&sc.Ic[j * 10], I);
      const SV hh = I * v;
      stsv(&sc.hc[j * 6], hh);
      stsv(&sc.Fc[j * 6], I * a + crf(v, hh));
    }
    else if (lane >= 32 && lane < 32 + NF)
    {
      const int f = lane - 32, j = h.foot_joint[f];
      st3(&sc.footp[f * 3], ldm3(&sc.oR[j * 9]) * ld3(mg.foot_p[f]) + ld3(&sc.op[j * 3]));
    }
    SMPC_LANES_END_WAVE
    static_assert(NF <= 32 && NJ <= 32, "lane map of the inertia / foot phase");
    ftick(fp, 1);
    // (a block whose dynamics block is staged out forms Bc afterwards, in the LDS the block leaves: full_bc_composites)
    constexpr bool BC_HERE = DERIV && !SC::DYN_OUT;
    if constexpr (BC_HERE)
      full_bc_bodies<D>(sc, *sd);
    // ---- composites, leaf -> root: lane = one scalar of (Ic | hc | Fc | Bc) ----
    SMPC_LANES(NT)
    if (lane < 22 + (BC_HERE ? 36 : 0))
    {
      double * base = lane < 10 ? sc.Ic : (lane < 16 ? sc.hc : (lane < 22 ? sc.Fc : (BC_HERE ? sd->Bc : sc.Fc)));
      const int stride = lane < 10 ? 10 : (lane < 22 ? 6 : 36);
      const int e = lane < 10 ? lane : (lane < 16 ? lane - 10 : (lane < 22 ? lane - 16 : lane - 22));
      // (the parents first: each step of the chain then waits for one LDS round trip, not two)
      int par[NJ];
#pragma unroll
      for (int j = 1; j < NJ; j++)
        par[j] = h.parent[j];
#pragma unroll
      for (int j = NJ - 1; j >= 1; j--)
        base[par[j] * stride + e] += base[j * stride + e];
    }
    SMPC_LANES_END_WAVE
    // ---- CoM, centroidal momentum; centroidal map columns (derivative pass) ----
    SMPC_LANES(NT)
    {
      const SI I0 = ldsi(&sc.Ic[0]);
      const V3 com = (1.0 / I0.m) * I0.mc;
      if (lane == 32)
      {
        st3(sc.com, com);
        const SV h0 = ldsv(&sc.hc[0]);
        st3(&sc.hg[0], h0.l);
        st3(&sc.hg[3], h0.a - cross(com, h0.l));
      }
    }
    SMPC_LANES_END_WAVE
    ftick(fp, 2);
    if (!dyn)
      return;
    // ---- Ic_j S_l ; bias forces ; right-hand side S tau - nle ----
    SMPC_LANES(NT)
    if (lane < NV)
    {
      const int l = lane, j = jof(l);
      const SI Ic = ldsi(&sc.Ic[j * 10]);
      const SV Sl = ldsv(&sc.S[l * 6]);
      stsv(&sc.IcS[l * 6], Ic * Sl);
      const SV g{ld3(h.gravity), mk3(0, 0, 0)};
      const SV Fg = ldsv(&sc.Fc[j * 6]) - Ic * g; // uniform field: every body accelerates with -g relative to free fall
      sc.W[l * NR] = ((l >= 6 && !D::KINO) ? sc.u[l - 6] : 0.0) - sv_dot6(Sl, Fg); // (kinodynamics variant: no joint torques)
    }
    SMPC_LANES_END_WAVE
    // ---- joint-space inertia M_kl = S_k . (Ic_j S_l) on the branch ; contact rows zeroed ----
    SMPC_LANES(NT)
    {
      for (int idx = lane; idx < (D::KINO ? 6 : NV) * NV; idx += NT) // (kinodynamics variant: the six base rows are all it needs)
      {
        const int k = idx / NV, l = idx % NV;
        const int jk = jof(k), jl = jof(l);
        double v = 0.0;
        if ((h.anc[jl] >> jk) & 1u)
          v = sv_dot6(ldsv(&sc.S[k * 6]), ldsv(&sc.IcS[l * 6]));
        else if ((h.anc[jk] >> jl) & 1u)
          v = sv_dot6(ldsv(&sc.S[l * 6]), ldsv(&sc.IcS[k * 6]));
        sc.M[idx] = v;
      }
      for (int idx = lane; idx < NCM * NV; idx += NT)
        sc.J[idx] = 0.0;
      if (lane < NCM)
      {
        sc.gam[lane] = 0.0;
        sc.lam[lane] = 0.0;
      }
    }
    SMPC_LANES_END_WAVE
    // ---- contact rows (feet in contact first, in order) ----
    if constexpr (FS == 6)
    {
      // CONTACT_6D, LOCAL_WORLD_ALIGNED (src/fulldynamics.cpp:56-65): rows [linear ; angular] in world axes at the foot point
      SMPC_LANES(NT)
      {
        for (int idx = lane; idx < NF * NV; idx += NT)
        {
          const int f = idx / NV, k = idx % NV;
          const int jf = h.foot_joint[f], jk = jof(k);
          if (((mask >> f) & 1u) && ((h.anc[jf] >> jk) & 1u))
          {
            const int c = __builtin_popcount(mask & ((1u << f) - 1u));
            const SV Sk = ldsv(&sc.S[k * 6]);
            const V3 col = Sk.l + cross(Sk.a, ld3(&sc.footp[f * 3]));
            sc.J[(6 * c + 0) * NV + k] = col.x;
            sc.J[(6 * c + 1) * NV + k] = col.y;
            sc.J[(6 * c + 2) * NV + k] = col.z;
            sc.J[(6 * c + 3) * NV + k] = Sk.a.x;
            sc.J[(6 * c + 4) * NV + k] = Sk.a.y;
            sc.J[(6 * c + 5) * NV + k] = Sk.a.z;
          }
        }
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane < NF && ((mask >> lane) & 1u))
      {
        const int f = lane, jf = h.foot_joint[f];
        const int c = __builtin_popcount(mask & ((1u << f) - 1u));
        const V3 p = ld3(&sc.footp[f * 3]);
        const SV v = ldsv(&sc.vel[jf * 6]), ab = ldsv(&sc.acc[jf * 6]);
        const V3 vp = v.l + cross(v.a, p);
        const V3 ap = ab.l + cross(ab.a, p) + cross(v.a, vp);
        const V3 rot = log3(ldm3(&sc.oR[jf * 9]));
        st3(&sc.rotl[f * 3], rot);
        sc.gam[6 * c + 0] = ap.x + h.Kd[0] * vp.x + h.Kp[0] * p.x;
        sc.gam[6 * c + 1] = ap.y + h.Kd[1] * vp.y + h.Kp[1] * p.y;
        sc.gam[6 * c + 2] = ap.z + h.Kd[2] * vp.z + h.Kp[2] * p.z;
        sc.gam[6 * c + 3] = ab.a.x + h.Kd[3] * v.a.x + h.Kp[3] * rot.x;
        sc.gam[6 * c + 4] = ab.a.y + h.Kd[4] * v.a.y + h.Kp[4] * rot.y;
        sc.gam[6 * c + 5] = ab.a.z + h.Kd[5] * v.a.z + h.Kp[5] * rot.z;
      }
      SMPC_LANES_END_WAVE
    }
    else
    {
    // CONTACT_3D, LOCAL (src/fulldynamics.cpp:66-74): LOCAL linear Jacobian of the foot point, drift + corrector
    SMPC_LANES(NT)
    {
      for (int idx = lane; idx < NF * NV; idx += NT)
      {
        const int f = idx / NV, k = idx % NV;
        const int jf = h.foot_joint[f], jk = jof(k);
        if (((mask >> f) & 1u) && ((h.anc[jf] >> jk) & 1u))
        {
          const int c = __builtin_popcount(mask & ((1u << f) - 1u));
          const M3 Rf = ldm3(&sc.oR[jf * 9]); // foot frame rotation = joint rotation
          const SV Sk = ldsv(&sc.S[k * 6]);
          const V3 col = tmul(Rf, Sk.l + cross(Sk.a, ld3(&sc.footp[f * 3])));
          sc.J[(3 * c + 0) * NV + k] = col.x;
          sc.J[(3 * c + 1) * NV + k] = col.y;
          sc.J[(3 * c + 2) * NV + k] = col.z;
        }
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < NF && ((mask >> lane) & 1u))
    {
      const int f = lane, jf = h.foot_joint[f];
      const int c = __builtin_popcount(mask & ((1u << f) - 1u));
      const M3 Rf = ldm3(&sc.oR[jf * 9]);
      const V3 p = ld3(&sc.footp[f * 3]);
      const SV v = ldsv(&sc.vel[jf * 6]), ab = ldsv(&sc.acc[jf * 6]);
      const V3 vp = v.l + cross(v.a, p);
      const V3 ap = ab.l + cross(ab.a, p) + cross(v.a, vp); // classical acceleration of the point at zero joint accelerations
      const V3 drift = tmul(Rf, ap), verr = tmul(Rf, vp), perr = tmul(Rf, (-1.0) * p);
      sc.gam[3 * c + 0] = drift.x + h.Kd[0] * verr.x - h.Kp[0] * perr.x;
      sc.gam[3 * c + 1] = drift.y + h.Kd[1] * verr.y - h.Kp[1] * perr.y;
      sc.gam[3 * c + 2] = drift.z + h.Kd[2] * verr.z - h.Kp[2] * perr.z;
    }
    SMPC_LANES_END_WAVE
    }
    ftick(fp, 3);
    if (rows_only) // (inverse-dynamics front end, smpc_id.h: M, S tau - nle, the contact rows and their drift are what it needs)
      return;
    if constexpr (D::KINO)
    {
      // ---- kinodynamics variant (KinodynamicsFwdDynamics, constructed at reference src/kinodynamics.cpp:85-87): the contact wrenches and the
      //      joint accelerations are controls; the base acceleration solves the six unactuated rows of the equations of motion,
      //      M_bb a_b = -nle_b + (J^T lam)_b - M_bj a_j  (J: the world-aligned 6-D rows at the foot points = the wrench [f ; (p - 0) x f + tau]
      //      at the world origin; the same balance as Ag a + dAg v = [m g + sum f ; sum (p - c) x f + tau] taken at the CoM) ----
      SMPC_LANES(NT)
      {
        if (lane < NCM)
        {
          const int c = lane / FS;
          int f = -1, cnt = 0;
          for (int ff = 0; ff < NF; ff++)
            if ((mask >> ff) & 1u)
            {
              if (cnt == c)
                f = ff;
              cnt++;
            }
          sc.lam[lane] = f >= 0 ? sc.u[FS * f + lane % FS] : 0.0; // compact: wrenches of the feet in contact, in order
        }
        for (int idx = lane; idx < 36; idx += NT)
          sc.Gi[idx] = sc.M[(idx / 6) * NV + idx % 6];
      }
      SMPC_LANES_END_WAVE
      static_assert(offsetof(SC, M) - offsetof(SC, xnext) >= 2 * 4 * 16 * sizeof(double), "the sweep scratch fits the late block");
      fwave_spd_inverse<6>(sc.Gi, sc.swp_());
      SMPC_LANES(NT)
      if (lane < 6)
      {
        double r = sc.W[lane * NR];
        for (int i = 0; i < NCM; i++)
          r += sc.J[i * NV + lane] * sc.lam[i];
        for (int l = 6; l < NV; l++)
          r -= sc.M[lane * NV + l] * sc.u[NCM + l - 6];
        sc.rhs[lane] = r;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        if (lane < 6)
        {
          double acc = 0.0;
          for (int d = 0; d < 6; d++)
            acc += sc.Gi[lane * 6 + d] * sc.rhs[d];
          sc.a[lane] = acc;
        }
        else if (lane < NV)
          sc.a[lane] = sc.u[NCM + lane - 6];
        if (lane == 0)
          sc.iters_[0] = 0;
      }
      SMPC_LANES_END_WAVE
      ftick(fp, 6);
      return;
    }
    // ---- M <- M^-1 (bordered symmetric sweep) ; W = M^-1 [S tau - nle | J^T] on the matrix cores ----
    static_assert(offsetof(SC, M) - offsetof(SC, xnext) >= SC::SWP_DOUBLES * sizeof(double), "the sweep scratch fits the late block");
    fwave_spd_inverse<NV>(sc.M, sc.swp_());
    fwave_gemm<NV, NR, NV>(
      [&](int i, int k) { return sc.M[k * NV + i]; },                                     // symmetric: read along the row of k
      [&](int k, int j) { return j == 0 ? sc.W[k * NR] : sc.J[(j - 1) * NV + k]; },       // [b | J^T]
      [&](int i, int j, double v) { sc.W[i * NR + j] = v; });
    ftick(fp, 4);
    // ---- damped Delassus matrix (unit diagonal on the rows of absent contacts), its inverse, J M^-1 b ----
    const int nc = FS * __builtin_popcount(mask & ((1u << NF) - 1u));
    {
      const double pmu = h.prox_mu;
      fwave_gemm<NCM, NR, NV>(
        [&](int i, int k) { return sc.J[i * NV + k]; }, [&](int k, int j) { return sc.W[k * NR + j]; },
        [&](int i, int j, double v) {
          if (j == 0)
            sc.JMb[i] = v;
          else
          {
            sc.Gi[i * NCM + j - 1] = v + (i == j - 1 ? (i < nc ? pmu : 1.0) : 0.0);
          }
        });
    }
    fwave_spd_inverse<NCM>(sc.Gi, sc.swp_());
    ftick(fp, 5);
    // ---- proximal iteration:  lam <- G^-1 (mu lam - gamma - J M^-1 b)  until |d lam|_inf <= accuracy ----
    int iters = 0;
    if (nc > 0)
      for (int it = 0; it < h.prox_max_iter; it++)
      {
        SMPC_LANES(NT)
        if (lane < NCM)
          sc.rhs[lane] = lane < nc ? h.prox_mu * sc.lam[lane] - sc.gam[lane] - sc.JMb[lane] : 0.0;
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        if (lane < NCM)
        {
          double acc = 0.0;
          for (int d = 0; d < NCM; d++)
            acc += sc.Gi[lane * NCM + d] * sc.rhs[d];
          sc.dl[lane] = fabs(acc - sc.lam[lane]);
          sc.lam[lane] = acc;
        }
        SMPC_LANES_END_WAVE
        iters = it + 1;
        double diff = 0.0; // wave-uniform: every lane reads the same values
        for (int c = 0; c < NCM; c++)
          diff = fmax(diff, sc.dl[c]);
        if (diff <= h.prox_accuracy)
          break;
      }
    // ---- a = M^-1 (S tau - nle) + M^-1 J^T lam ----
    SMPC_LANES(NT)
    {
      if (lane < NV)
      {
        double acc = sc.W[lane * NR];
        for (int c = 0; c < NCM; c++)
          acc += sc.W[lane * NR + 1 + c] * sc.lam[c];
        sc.a[lane] = acc;
      }
      if (lane == 0)
        sc.iters_[0] = iters;
    }
    SMPC_LANES_END_WAVE
    if constexpr (DERIV && D::WIDE_DEV)
    {
      // ---- blocks of the inverse of the contact KKT matrix [M -J^T ; J mu] for the derivative solve (full_kkt_inv): the solve is then ONE
      //      product over the (NV + NCM) x NCOL right-hand sides instead of a chain of four whose results go through memory.
      //      T = G^-1 J M^-1 (into J, dead from here) ;  A = M^-1 - M^-1 J^T T (into M) ----
      fwave_gemm<NCM, NV, NCM>(
        [&](int i, int k) { return sc.Gi[k * NCM + i]; }, [&](int k, int j) { return sc.W[j * NR + 1 + k]; },
        [&](int i, int j, double v) { sc.J[i * NV + j] = v; });
      fwave_gemm<NV, NV, NCM>(
        [&](int i, int k) { return sc.W[i * NR + 1 + k]; }, [&](int k, int j) { return -sc.J[k * NV + j]; },
        [&](int i, int j) { return sc.M[i * NV + j]; }, [&](int i, int j, double v) { sc.M[i * NV + j] = v; });
    }
    ftick(fp, 6);
  }
  // entry (i, k) of  [da ; dlam] = Kinv [r1 ; r2]:  Kinv = [-A  -T^T ; T  -G^-1]  from the blocks full_dynamics_phases<D, true> leaves in the
  // dynamics block (A in M, T in J, G^-1 in Gi):  [M -J^T ; J mu] [da ; dlam] = -[r1 ; r2]
  template <class D, class SC>
  SMPC_DEV double full_kkt_inv(const SC & sc, int i, int k)
  {
    constexpr int NV = D::NV, NCM = D::NCM;
    const bool iu = i < NV, ku = k < NV;
    const double * src = iu ? (ku ? &sc.M[k * NV + i] : &sc.J[(k - NV) * NV + i]) : (ku ? &sc.J[(i - NV) * NV + k] : &sc.Gi[(k - NV) * NCM + (i - NV)]);
    return ((!iu && ku) ? 1.0 : -1.0) * *src;
  }

  // entry (i, j) of the 17 x 6 wrench-cone matrix of a rectangular sole (half length L, half width W, friction mu) acting on the
  // contact wrench [f ; tau]:  A lam <= 0  (MultibodyWrenchConeResidual, reference src/fulldynamics.cpp:167-172; rows: unilateral
  // force, friction pyramid (4), centre of pressure inside the sole (4), yaw-torque bounds (8))
  SMPC_HD double wrench_cone_entry(int i, int j, double mu, double L, double W)
  {
    if (i == 0)
      return j == 2 ? -1.0 : 0.0;
    if (i < 5)
    {
      const int a = (i - 1) / 2;                 // 0: x, 1: y
      const double sgn = ((i - 1) % 2) ? 1.0 : -1.0;
      return j == a ? sgn : (j == 2 ? -mu : 0.0);
    }
    if (i < 9)
    {
      const int a = (i - 5) / 2;                 // 0: tau_x with W, 1: tau_y with L
      const double sgn = ((i - 5) % 2) ? 1.0 : -1.0;
      return j == 3 + a ? sgn : (j == 2 ? (a == 0 ? -W : -L) : 0.0);
    }
    const int q = i - 9, hi = q / 4, r = q % 4; // r: signs of (W, L) = (+,+) (+,-) (-,+) (-,-)
    const double sw = (r < 2) ? 1.0 : -1.0, sl = (r % 2 == 0) ? 1.0 : -1.0;
    const double st = hi ? 1.0 : -1.0;           // sign of the tau_z entry; the tau_x / tau_y entries are st * mu * (sw, sl)
    switch (j)
    {
    case 0:
      return sw * W;
    case 1:
      return sl * L;
    case 2:
      return -mu * (L + W);
    case 3:
      return st * sw * mu;
    case 4:
      return st * sl * mu;
    default:
      return st;
    }
  }

  // SE(3) work of a stage on two lanes: lane 0 integrates the base (x+ and, with derivatives, Jexp6(nu) and the action matrix
  // of exp6(nu)^-1), lane 1 forms the base block of the state residual log6(M_tgt^-1 M) (and Jlog6).  One instruction stream.
  template <class D, bool DERIV, class SC, class SD>
  SMPC_DEV void full_se3_pair(SC & sc, SD * sd, double dt, bool zero_acc, int lane)
  {
    constexpr int NQ = D::NQ;
    const double * vq = &sc.x[NQ];
    const bool ex = lane == 0;
    V3 vec, w;
    if (ex)
    {
      const double a0 = zero_acc ? 0.0 : sc.a[0], a1 = zero_acc ? 0.0 : sc.a[1], a2 = zero_acc ? 0.0 : sc.a[2];
      const double a3 = zero_acc ? 0.0 : sc.a[3], a4 = zero_acc ? 0.0 : sc.a[4], a5 = zero_acc ? 0.0 : sc.a[5];
      vec = mk3(dt * (vq[0] + dt * a0), dt * (vq[1] + dt * a1), dt * (vq[2] + dt * a2));
      w = mk3(dt * (vq[3] + dt * a3), dt * (vq[4] + dt * a4), dt * (vq[5] + dt * a5));
    }
    else
    {
      const double * xt = sc.x_tgt;
      const SE3 Mt{quat_to_R(Quat{xt[3], xt[4], xt[5], xt[6]}), ld3(xt)};
      const SE3 M = se3_mul(se3_inv(Mt), SE3{ldm3(&sc.oR[0]), ld3(&sc.op[0])});
      w = log3(M.R);
      vec = M.p;
    }
    const double t = sqrt(dot(w, w));
    const M3 W = skew(w);
    const M3 W2 = W * W;
    const SE3Coef kf = se3_coef(t);
    const double cB = kf.B, cC = kf.C, cD = kf.D;
    const double c1 = ex ? cB : -0.5, c2 = ex ? cC : cD;
    const V3 out = vec + c1 * (W * vec) + c2 * (W2 * vec);
    const V3 v = ex ? vec : out;
    M3 J = m3_id(), Q = m3_id();
    if constexpr (DERIV)
    {
      Q = se3_Q(-1.0 * v, -1.0 * w, kf);
      J = m3_id() + (ex ? -cB : 0.5) * W + c2 * W2; // Jexp3(w) | Jlog3(w)
    }
    if (ex)
    {
      const M3 R0 = ldm3(&sc.oR[0]);
      st3(&sc.xnext[0], ld3(&sc.op[0]) + R0 * out);
      const double qs = 0.5 * kf.sinch;
      Quat qn = quat_mul(Quat{sc.x[3], sc.x[4], sc.x[5], sc.x[6]}, Quat{qs * w.x, qs * w.y, qs * w.z, kf.ch});
      const double n = 1.0 / sqrt(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
      sc.xnext[3] = qn.x * n;
      sc.xnext[4] = qn.y * n;
      sc.xnext[5] = qn.z * n;
      sc.xnext[6] = qn.w * n;
      if constexpr (DERIV)
      {
        stm3(sd->Je3, J);
        stm3(sd->JeQ, Q);
        const M3 Rt = transpose(m3_id() + kf.sinc * W + cB * W2);
        const M3 X = (-1.0) * (Rt * skew(out));
        const double rt[9] = {Rt.a00, Rt.a01, Rt.a02, Rt.a10, Rt.a11, Rt.a12, Rt.a20, Rt.a21, Rt.a22};
        const double xx[9] = {X.a00, X.a01, X.a02, X.a10, X.a11, X.a12, X.a20, X.a21, X.a22};
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
          {
            sd->Jq[i * 6 + j] = rt[i * 3 + j];
            sd->Jq[(i + 3) * 6 + j + 3] = rt[i * 3 + j];
            sd->Jq[i * 6 + j + 3] = xx[i * 3 + j];
            sd->Jq[(i + 3) * 6 + j] = 0.0;
          }
      }
    }
    else
    {
      st3(&sc.rx[0], out);
      st3(&sc.rx[3], w);
      if constexpr (DERIV)
      {
        const M3 X = (-1.0) * (J * Q * J);
        const double ji[9] = {J.a00, J.a01, J.a02, J.a10, J.a11, J.a12, J.a20, J.a21, J.a22};
        const double xx[9] = {X.a00, X.a01, X.a02, X.a10, X.a11, X.a12, X.a20, X.a21, X.a22};
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
          {
            sd->Jl[i * 6 + j] = ji[i * 3 + j];
            sd->Jl[(i + 3) * 6 + j + 3] = ji[i * 3 + j];
            sd->Jl[i * 6 + j + 3] = xx[i * 3 + j];
            sd->Jl[(i + 3) * 6 + j] = 0.0;
          }
      }
    }
  }

  // land_cstr rows of a landing foot f (reference src/fulldynamics.cpp:175-181, 191-210): its LOCAL_WORLD_ALIGNED frame velocity (the point
  // velocity in world axes, rows 0 .. 2; 6-D feet: the angular velocity of the body, rows 3 .. 5), 3-D feet: row 3 = height above the contact pose
  template <class D, class SC>
  SMPC_DEV double full_land_value(const SC & sc, int f, int r)
  {
    const int jf = sc.h.foot_joint[f];
    const V3 p = ld3(&sc.footp[f * 3]);
    const SV v = ldsv(&sc.vel[jf * 6]);
    if (r < 3)
    {
      const V3 vp = v.l + cross(v.a, p);
      return r == 0 ? vp.x : (r == 1 ? vp.y : vp.z);
    }
    if (D::FS == 6)
      return r == 3 ? v.a.x : (r == 4 ? v.a.y : v.a.z);
    return p.z - sc.h.land_z[f];
  }
  // entry (r, k) of their Jacobian, k a tangent index of the state (q | v): with d_k = v_parent(k) x S_k (the part of d v_l / d q_k that is
  // not the rigid motion of the subtree by S_k) d(point velocity)/dq_k = d_k.l + d_k.a x p + S_k.a x v_p, d(omega)/dq_k = d_k.a + S_k.a x omega,
  // d(.)/dv_k = the point / angular Jacobian column, which is d(position)/dq_k too
  template <class D, class SC, class SD>
  SMPC_DEV double full_land_entry(const SC & sc, const SD & sd, int f, int r, int k)
  {
    constexpr int NV = D::NV;
    const int jf = sc.h.foot_joint[f], kk = k < NV ? k : k - NV;
    if (!((sc.h.anc[jf] >> jof(kk)) & 1u))
      return 0.0;
    const V3 p = ld3(&sc.footp[f * 3]);
    const SV Sk = ldsv(&sc.S[kk * 6]);
    V3 o;
    if (k >= NV) // velocity columns
      o = r < 3 ? Sk.l + cross(Sk.a, p) : (D::FS == 6 ? Sk.a : V3{0.0, 0.0, 0.0});
    else
    {
      const SV v = ldsv(&sc.vel[jf * 6]), d = ldsv(&sd.dk[kk * 6]);
      if (r < 3)
        o = d.l + cross(d.a, p) + cross(Sk.a, v.l + cross(v.a, p));
      else if (D::FS == 6)
        o = d.a + cross(Sk.a, v.a);
      else
        return (Sk.l + cross(Sk.a, p)).z;
    }
    const int c = r % 3;
    return c == 0 ? o.x : (c == 1 ? o.y : o.z);
  }

  // kinodynamics variant: entry (r, k) of the Jacobian of the LOCAL 6-D frame velocity of foot f (rows: R^T v_point, R^T omega), k a tangent
  // index of the state (q | v).  The rigid turn of the subtree by S_k cancels in the LOCAL frame: what is left of d/dq_k is d_k = v_parent x S_k;
  // d/dv_k is the LOCAL frame Jacobian column (Pinocchio getFrameVelocityDerivatives, LOCAL)
  template <class D, class SC, class SD>
  SMPC_DEV double kino_vel_entry(const SC & sc, const SD & sd, int f, int r, int k)
  {
    constexpr int NV = D::NV;
    const int jf = sc.h.foot_joint[f], kk = k < NV ? k : k - NV;
    if (!((sc.h.anc[jf] >> jof(kk)) & 1u))
      return 0.0;
    const V3 p = ld3(&sc.footp[f * 3]);
    const SV m = k >= NV ? ldsv(&sc.S[kk * 6]) : ldsv(&sd.dk[kk * 6]);
    const V3 o = tmul(ldm3(&sc.oR[jf * 9]), r < 3 ? m.l + cross(m.a, p) : m.a);
    const int c = r % 3;
    return c == 0 ? o.x : (c == 1 ? o.y : o.z);
  }

  // x+ (semi-implicit Euler), defect, residuals, weighted residuals, cost, constraint values, AL multipliers, merit pieces.
  // Results: sc.red[0] cost, sc.red[1] penalty part of the merit, sc.red[2] primal infeasibility.
  template <class D, bool DERIV, class SC, class SD>
  SMPC_DEV void full_eval_tail(SC & sc, SD * sd, const DevModel<D> & mg, unsigned mask, unsigned land, bool term, const double * lam_e, const double * nu_e, FullProf * fpp = nullptr)
  {
    constexpr int NT = 64;
    constexpr int NV = D::NV, NQ = D::NQ, NF = D::NF, NCM = D::NCM, NU = D::NU, NDX = D::NDX, NC = D::NC, NA = D::NA, FS = D::FS, NX = D::NX;
    const FullHead<D> & h = sc.h;
    const double dt = h.dt, mu = h.mu;
    const double * vq = &sc.x[NQ];
    // the centres of the augmented Lagrangian (device memory) are what the multiplier phase below waits for: their loads are issued here, a
    // whole SE(3) pair ahead of their use
    constexpr int PDE = (NDX + NT - 1) / NT, PCE = (NC + NT - 1) / NT;
    SMPC_PLA(double, lame_r, NT, PDE);
    SMPC_PLA(double, nue_r, NT, PCE);
    SMPC_LANES(NT)
    {
#pragma unroll
      for (int n = 0; n < PDE; n++)
        SMPC_PLV(lame_r)[n] = lam_e[lane + n * NT < NDX ? lane + n * NT : NDX - 1];
#pragma unroll
      for (int n = 0; n < PCE; n++)
        SMPC_PLV(nue_r)[n] = nu_e[lane + n * NT < NC ? lane + n * NT : NC - 1];
    }
    SMPC_LANES_END_WAVE
    // ---- x+ = x (+) [dt (v + dt a); dt a] ; base block of the state residual ----
    SMPC_LANES(NT)
    {
      if (lane < 2)
        full_se3_pair<D, DERIV>(sc, sd, dt, term, lane);
      for (int i = 6 + lane; i < NV; i += NT)
        sc.xnext[i + 1] = sc.x[i + 1] + dt * (vq[i] + dt * (term ? 0.0 : sc.a[i]));
      for (int i = lane; i < NV; i += NT)
        sc.xnext[NQ + i] = vq[i] + dt * (term ? 0.0 : sc.a[i]);
    }
    SMPC_LANES_END_WAVE
    if (fpp) ftick(*fpp, 20);
    if (!term)
    {
      SMPC_LANES(NT)
      lanes_difference<D>(sc.xn1, sc.xnext, sc.e, lane, 61);
      SMPC_LANES_END_WAVE
    if (fpp) ftick(*fpp, 21);
    }
    // ---- residuals and constraint values ----
    SMPC_LANES(NT)
    {
      for (int i = 6 + lane; i < NV; i += NT)
        sc.rx[i] = sc.x[i + 1] - sc.x_tgt[i + 1];
      for (int i = lane; i < NV; i += NT)
        sc.rx[NV + i] = sc.x[NQ + i] - sc.x_tgt[NQ + i];
      if (!term)
      {
        for (int i = lane; i < NU; i += NT)
        {
          sc.ru[i] = sc.u[i] - sc.u_ref[i];
          sc.cval[i] = h.torque_limits ? sc.u[i] : 0.0;
        }
        for (int i = lane; i < NA; i += NT)
          sc.cval[NU + i] = h.kinematics_limits ? sc.x[7 + i] : 0.0;
        if constexpr (FS == 3)
        {
          for (int i = lane; i < NF * 3; i += NT)
            sc.rf[i] = sc.footp[i] - sc.foot_ref[i];
        }
        else
        {
          // FramePlacementResidual: log6(M_ref^-1 oMf), M_ref = (identity rotation, reference translation) (src/mpc.cpp:304-308)
          if (lane >= 48 && lane < 48 + NF)
          {
            const int f = lane - 48;
            const SE3 M{ldm3(&sc.oR[h.foot_joint[f] * 9]), ld3(&sc.footp[f * 3]) - ld3(&sc.foot_ref[f * 3])};
            V3 v, w;
            log6(M, v, w);
            st3(&sc.rf[6 * f], v);
            st3(&sc.rf[6 * f + 3], w);
            if constexpr (DERIV)
            {
              M3 Ji, X;
              Jlog6(v, w, Ji, X);
              const double ji[9] = {Ji.a00, Ji.a01, Ji.a02, Ji.a10, Ji.a11, Ji.a12, Ji.a20, Ji.a21, Ji.a22};
              const double xx[9] = {X.a00, X.a01, X.a02, X.a10, X.a11, X.a12, X.a20, X.a21, X.a22};
              double * Jl = &sd->Jlf[f * 36];
              for (int a = 0; a < 3; a++)
                for (int bb = 0; bb < 3; bb++)
                {
                  Jl[a * 6 + bb] = ji[a * 3 + bb];
                  Jl[(a + 3) * 6 + bb + 3] = ji[a * 3 + bb];
                  Jl[a * 6 + bb + 3] = xx[a * 3 + bb];
                  Jl[(a + 3) * 6 + bb] = 0.0;
                }
            }
          }
        }
        // wrench-cone rows of the feet in contact: A_cone lam
        for (int i = lane; i < D::NCONE; i += NT)
        {
          const int f = i / (D::NCONE1 > 0 ? D::NCONE1 : 1), r = i % (D::NCONE1 > 0 ? D::NCONE1 : 1);
          double acc = 0.0;
          if (h.force_cone && ((mask >> f) & 1u))
          {
            const int c = __builtin_popcount(mask & ((1u << f) - 1u));
            for (int j = 0; j < FS; j++) // (3-D feet: rows 0 .. 4 and columns 0 .. 2 of the wrench cone = the friction pyramid on the force)
              acc += wrench_cone_entry(r, j, h.fric_mu, h.Lfoot, h.Wfoot) * sc.lam[c * FS + j];
          }
          sc.cval[NU + NA + i] = acc;
        }
        // land_cstr rows of the feet that land at this stage (`land`: already masked by the contacts and the switch)
        for (int i = lane; i < D::NLAND; i += NT)
        {
          const int f = i / (D::NLAND1 > 0 ? D::NLAND1 : 1), r = i % (D::NLAND1 > 0 ? D::NLAND1 : 1);
          sc.cval[NU + NA + D::NCONE + i] = ((land >> f) & 1u) ? full_land_value<D>(sc, f, r) : 0.0;
        }
        if constexpr (D::KINO)
        {
          // frame-velocity rows of the feet in contact: the LOCAL 6-D frame velocity [R^T v_point ; R^T omega] (FrameVelocityResidual(..., LOCAL),
          // reference src/kinodynamics.cpp:110-123)
          for (int i = lane; i < D::NVEL; i += NT)
          {
            const int f = i / FS, r = i % FS;
            double v = 0.0;
            if ((mask >> f) & 1u)
            {
              const int jf = h.foot_joint[f];
              const SV vl = ldsv(&sc.vel[jf * 6]);
              const V3 w = r < 3 ? vl.l + cross(vl.a, ld3(&sc.footp[f * 3])) : vl.a;
              const V3 o = tmul(ldm3(&sc.oR[jf * 9]), w);
              v = r % 3 == 0 ? o.x : (r % 3 == 1 ? o.y : o.z);
            }
            sc.cval[NU + NA + D::NCD + i] = v;
          }
          // centroidal_derivative residual hdot(u, q) = [m g + sum f ; sum (p - c) x f + tau] over the feet in contact (rows 0 .. 5 of the
          // "force" block of the stacked residual; src/kinodynamics.cpp:57-59, 63-64)
          if (lane >= 56 && lane < 62)
          {
            const int r = lane - 56;
            V3 lin = h.total_mass * ld3(h.gravity), ang = mk3(0, 0, 0);
            const V3 com = ld3(sc.com);
            for (int f = 0; f < NF; f++)
              if ((mask >> f) & 1u)
              {
                const V3 F = ld3(&sc.u[FS * f]);
                lin = lin + F;
                ang = ang + cross(ld3(&sc.footp[f * 3]) - com, F) + ld3(&sc.u[FS * f + 3]);
              }
            sc.rl[r] = r < 3 ? (r == 0 ? lin.x : (r == 1 ? lin.y : lin.z)) : (r == 3 ? ang.x : (r == 4 ? ang.y : ang.z));
          }
          for (int i = 6 + lane; i < NCM; i += NT)
            sc.rl[i] = 0.0;
        }
        // contact-force residual of the feet in contact (compact row c of foot f)
        for (int i = lane; i < (D::KINO ? 0 : NCM); i += NT)
        {
          int f = -1, cnt = 0;
          for (int ff = 0; ff < NF; ff++)
            if ((mask >> ff) & 1u)
            {
              if (cnt == i / FS)
                f = ff;
              cnt++;
            }
          sc.rl[i] = f >= 0 ? sc.lam[i] - sc.f_ref[f * FS + i % FS] : 0.0;
        }
      }
    }
    SMPC_LANES_END_WAVE
    if (fpp) ftick(*fpp, 22);
    // ---- weighted residuals ----
    SMPC_LANES(NT)
    {
      for (int i = lane; i < NDX; i += NT)
      {
        double s = 0.0;
        if (h.w_diag)
          s = h.wxd[i] * sc.rx[i];
        else
          for (int j = 0; j < NDX; j++)
            s += mg.w_x[i * NDX + j] * sc.rx[j];
        sc.Wrx[i] = s;
      }
      if (lane >= 40 && lane < 46)
      {
        const int i = lane - 40;
        double s = 0.0;
        const double sc10 = term ? 10.0 : 1.0; // terminal: 10 * w_cent (src/fulldynamics.cpp:427)
        for (int j = 0; j < 6; j++)
          s += sc10 * h.w_cent[i * 6 + j] * sc.hg[j];
        sc.Whg[i] = s;
      }
      if (!term)
      {
        for (int i = lane; i < NU; i += NT)
        {
          double s = 0.0;
          if (h.w_diag)
            s = h.wud[i] * sc.ru[i];
          else
            for (int j = 0; j < NU; j++)
              s += mg.w_u[i * NU + j] * sc.ru[j];
          sc.Wru[i] = s;
        }
        for (int i = lane; i < NF * D::PF; i += NT)
        {
          const int f = i / D::PF, r = i % D::PF;
          double s = 0.0;
          for (int j = 0; j < D::PF; j++)
            s += h.w_frame[r * FS + j] * sc.rf[f * D::PF + j];
          sc.Wrf[i] = s;
        }
        for (int i = lane; i < NCM; i += NT)
        {
          const int c = i / FS, r = i % FS;
          double s = 0.0;
          if constexpr (D::KINO)
          {
            if (i < 6)
              for (int j = 0; j < 6; j++)
                s += h.w_centder[i * 6 + j] * sc.rl[j];
          }
          else
            for (int j = 0; j < FS; j++)
              s += h.w_forces[r * FS + j] * sc.rl[c * FS + j];
          sc.Wrl[i] = s;
        }
      }
    }
    SMPC_LANES_END_WAVE
    if (fpp) ftick(*fpp, 23);
    // ---- cost: lane-strided partial sums, fixed-order reduction ----
    SMPC_LANES(NT)
    {
      double c = 0.0;
      for (int i = lane; i < NDX; i += NT)
        c += sc.rx[i] * sc.Wrx[i];
      if (lane < 6)
        c += sc.hg[lane] * sc.Whg[lane];
      if (!term)
      {
        for (int i = lane; i < NU; i += NT)
          c += sc.ru[i] * sc.Wru[i];
        for (int i = lane; i < NF * D::PF; i += NT)
          c += sc.rf[i] * sc.Wrf[i];
        for (int i = lane; i < NCM; i += NT)
          c += sc.rl[i] * sc.Wrl[i];
      }
      sc.part[lane] = c;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < 8)
    {
      double c = 0.0;
      for (int i = 0; i < 8; i++)
        c += sc.part[lane * 8 + i];
      sc.part8[lane] = c;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double c = 0.0;
      for (int i = 0; i < 8; i++)
        c += sc.part8[i];
      sc.red[0] = 0.5 * c;
    }
    SMPC_LANES_END_WAVE
    if (fpp) ftick(*fpp, 24);
    if (term)
      return;
    // ---- AL multipliers (SolverProxDDP computeMultipliers), merit penalty, primal infeasibility ----
    SMPC_PL(double, prim_l, NT); // (the two reductions share `part`, one after the other: 512 B of LDS decide between 3 and 4 resident blocks)
    SMPC_PL(double, pen8_l, NT);
    SMPC_LANES(NT)
    {
      double pen = 0.0, prim = 0.0;
#pragma unroll
      for (int n = 0; n < PDE; n++)
      {
        const int i = lane + n * NT;
        if (i >= NDX)
          break;
        const double lp = SMPC_PLV(lame_r)[n] + sc.e[i] / mu;
        sc.lamp[i] = lp;
        const double dl = lp - sc.lam_next[i];
        pen += 0.5 * mu * (lp * lp + dl * dl);
        prim = fmax(prim, fabs(sc.e[i]));
      }
#pragma unroll
      for (int n = 0; n < PCE; n++)
      {
        const int i = lane + n * NT;
        if (i >= NC)
          break;
        // rows: torque box | joint box | wrench-cone rows of the feet in contact (negative orthant) | rows of the landing feet (equality)
        // (kinodynamics variant: | frame-velocity rows of the feet in contact (equality), behind the dense rows)
        const bool box = i < NU + NA, eq = i >= NU + NA + D::NCONE, vel = i >= NU + NA + D::NCD;
        const bool present = i < NU ? h.torque_limits != 0
                                    : (box ? h.kinematics_limits != 0
                                           : (vel ? ((mask >> ((i - NU - NA - D::NCD) / FS)) & 1u) != 0u
                                            : (eq ? ((land >> ((i - NU - NA - D::NCONE) / (D::NLAND1 > 0 ? D::NLAND1 : 1))) & 1u) != 0u
                                                 : (h.force_cone != 0 && ((mask >> ((i - NU - NA) / (D::NCONE1 > 0 ? D::NCONE1 : 1))) & 1u)))));
        double vp = 0.0;
        int act = 0;
        if (present)
        {
          const double lo = i < NU ? h.umin[i] : (box ? h.qmin[i - NU] : -1e300), hi = i < NU ? h.umax[i] : (box ? h.qmax[i - NU] : 0.0);
          const double z = sc.cval[i] + mu * SMPC_PLV(nue_r)[n];
          const double proj = box ? fmin(fmax(z, lo), hi) : (eq ? 0.0 : fmin(z, 0.0));
          vp = (z - proj) / mu;
          act = eq || z != proj;
          prim = fmax(prim, box ? fmax(fmax(sc.cval[i] - hi, lo - sc.cval[i]), 0.0) : (eq ? fabs(sc.cval[i]) : fmax(sc.cval[i], 0.0)));
        }
        sc.vplus[i] = vp;
        sc.act[i] = act;
        const double dv = vp - sc.nu[i];
        pen += 0.5 * mu * (vp * vp + dv * dv);
      }
      sc.part[lane] = pen;
      SMPC_PLV(prim_l) = prim;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      double pen = 0.0;
      if (lane < 8)
        for (int i = 0; i < 8; i++)
          pen += sc.part[lane * 8 + i];
      SMPC_PLV(pen8_l) = pen;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      sc.part[lane] = SMPC_PLV(prim_l);
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < 8)
    {
      double prim = 0.0;
      for (int i = 0; i < 8; i++)
        prim = fmax(prim, sc.part[lane * 8 + i]);
      sc.part8[lane] = SMPC_PLV(pen8_l);
      sc.part8[8 + lane] = prim;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double pen = 0.0, prim = 0.0;
      for (int i = 0; i < 8; i++)
      {
        pen += sc.part8[i];
        prim = fmax(prim, sc.part8[8 + i]);
      }
      sc.red[1] = pen;
      sc.red[2] = prim;
    }
    SMPC_LANES_END_WAVE
    (void)NX;
  }
} // namespace smpc

namespace smpc
{
  // Momentum and foot-pose rows of the stacked Gauss-Newton Jacobian (rows NCM .. NGN-1): written into the dynamics block of the
  // evaluation scratch (M | J | W | Gi | IcS), which is dead once the derivative solves are done (terminal node: never used).
  // row r >= NCM (momentum, foot poses) of the stacked Gauss-Newton Jacobian: behind the force rows in the device slice, or in the dead
  // dynamics block of the LDS scratch
  template <class D, class SC>
  SMPC_DEV double * full_gn_row(SC & sc, FullDerivWide<D> & sw, int r)
  {
    if constexpr (D::WIDE_DEV)
      return &sw.JT[r * SC::NCOL];
    else
      return sc.jt2_() + (r - D::NCM) * SC::NCOL;
  }
  template <class D, class SC, class SD>
  SMPC_DEV void full_gn_rows(SC & sc, SD & sd, FullDerivWide<D> & sw, bool term)
  {
    constexpr int NT = 64;
    constexpr int NV = D::NV, NF = D::NF, NU = D::NU, NCOL = SC::NCOL, FS = D::FS;
    const FullHead<D> & h = sc.h;
    SMPC_LANES(NT)
    if (lane < NV)
    {
      const int k = lane, i = jof(k);
      const SV s = ldsv(&sc.S[k * 6]);
      const SV d = ldsv(&sd.dk[k * 6]);
      const SI Ici = ldsi(&sc.Ic[i * 10]);
      const SV hci = ldsv(&sc.hc[i * 6]);
      const SV dh = crf(s, hci) + Ici * d;
      const V3 com = ld3(sc.com);
      const V3 jc = (1.0 / h.total_mass) * (Ici * s).l;
      const SV h0 = ldsv(&sc.hc[0]);
      const V3 dha = dh.a - cross(com, dh.l) - cross(jc, h0.l);
      double * jc6 = full_gn_row<D>(sc, sw, D::NCM);
      jc6[0 * NCOL + k] = dh.l.x;
      jc6[1 * NCOL + k] = dh.l.y;
      jc6[2 * NCOL + k] = dh.l.z;
      jc6[3 * NCOL + k] = dha.x;
      jc6[4 * NCOL + k] = dha.y;
      jc6[5 * NCOL + k] = dha.z;
      {
        // centroidal map column: (Ic_i S_k) translated to the CoM
        const SV c = Ici * s;
        const V3 ang = c.a - cross(com, c.l);
        const double agc[6] = {c.l.x, c.l.y, c.l.z, ang.x, ang.y, ang.z};
        for (int r = 0; r < 6; r++)
        {
          jc6[r * NCOL + NV + k] = agc[r];
          for (int c = k; c < NU; c += NV)
            jc6[r * NCOL + 2 * NV + c] = 0.0;
        }
      }
      double * jf = full_gn_row<D>(sc, sw, D::NCM + 6);
      constexpr int PF = D::PF;
      for (int f = 0; f < NF; f++)
      {
        const bool on = (h.anc[h.foot_joint[f]] >> i) & 1u;
        V3 c = mk3(0, 0, 0);
        if (on)
          c = s.l + cross(s.a, ld3(&sc.footp[f * 3]));
        if constexpr (FS == 3)
        {
          jf[(3 * f + 0) * NCOL + k] = c.x;
          jf[(3 * f + 1) * NCOL + k] = c.y;
          jf[(3 * f + 2) * NCOL + k] = c.z;
        }
        else
        {
          // FramePlacementResidual: Jlog6(M_ref^-1 oMf) * (LOCAL 6-D frame Jacobian column)
          double col[6] = {0, 0, 0, 0, 0, 0};
          if (on && !term)
          {
            const M3 Rf = ldm3(&sc.oR[h.foot_joint[f] * 9]);
            const V3 lin = tmul(Rf, c), ang = tmul(Rf, s.a);
            const double lv[6] = {lin.x, lin.y, lin.z, ang.x, ang.y, ang.z};
            const double * Jl = &sd.Jlf[f * 36];
            for (int r = 0; r < 6; r++)
              for (int m = 0; m < 6; m++)
                col[r] += Jl[r * 6 + m] * lv[m];
          }
          for (int r = 0; r < 6; r++)
            jf[(6 * f + r) * NCOL + k] = col[r];
        }
        for (int r = 0; r < PF; r++)
        {
          jf[(PF * f + r) * NCOL + NV + k] = 0.0;
          for (int c = k; c < NU; c += NV)
            jf[(PF * f + r) * NCOL + 2 * NV + c] = 0.0;
        }
      }
    }
    SMPC_LANES_END_WAVE
  }

  // -------------------------------------------------------------------------------------------------------------
  // Derivative phases.  In: everything full_dynamics_phases / full_eval_tail left in the scratch.  Out (sd):
  //   R1 = [da_dq | da_dv | da_dtau] (NV x NCOL),  JT rows: [dlam_* (NCM) ; centroidal momentum (6) ; foot positions (3 NF)],
  //   WJ = block-diagonal weight x JT,  Je3 / JeQ / Jq / Jl from the SE(3) pair.
  // -------------------------------------------------------------------------------------------------------------
  template <class D, class SC, class SD>
  SMPC_DEV void full_deriv_phases(SC & sc, SD & sd, FullDerivWide<D> & sw, double * wtmp, const DevModel<D> & mg, unsigned mask, bool term, FullProf & fp)
  {
    constexpr int NT = 64;
    constexpr int NJ = D::NJ, NV = D::NV, NF = D::NF, NCM = D::NCM, NU = D::NU, NR = SC::NR, NCOL = SC::NCOL, FS = D::FS, NGN = SC::NGN;
    const FullHead<D> & h = sc.h;
    (void)mg;
    const SV g6{ld3(h.gravity), mk3(0, 0, 0)};
    if (!term)
    {
      // ---- body accelerations at the solved a ; body forces with the gravity field ; contact wrenches at the world origin ----
      SMPC_LANES(NT)
      if (lane < NJ)
      {
        const int i = lane;
        const unsigned anci = h.anc[i];
        SV dacc = sv0();
        for (int k = 0; k < NV; k++)
          if ((anci >> jof(k)) & 1u)
            dacc = dacc + sc.a[k] * ldsv(&sc.S[k * 6]);
        const SV acc = ldsv(&sc.acc[i * 6]) + dacc;
        stsv(&sc.acc[i * 6], acc);
        const SI I = ldsi(&sc.I[i * 10]);
        const SV v = ldsv(&sc.vel[i * 6]);
        stsv(&sc.Fc[i * 6], I * (acc - g6) + crf(v, I * v));
      }
      else if (lane >= 32 && lane < 32 + NF)
      {
        const int f = lane - 32;
        SV W = sv0();
        if ((mask >> f) & 1u)
        {
          const int c = __builtin_popcount(mask & ((1u << f) - 1u));
          if constexpr (FS == 3)
          {
            const V3 fw = ldm3(&sc.oR[h.foot_joint[f] * 9]) * ld3(&sc.lam[3 * c]); // LOCAL contact force in world axes
            W = SV{fw, cross(ld3(&sc.footp[f * 3]), fw)};
          }
          else
          {
            const V3 fl = ld3(&sc.lam[6 * c]);                                       // world-aligned wrench at the foot point
            W = SV{fl, cross(ld3(&sc.footp[f * 3]), fl) + ld3(&sc.lam[6 * c + 3])};
          }
        }
        stsv(&sd.Wc[f * 6], W);
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane < 6)
      {
        int par[NJ];
#pragma unroll
        for (int j = 1; j < NJ; j++)
          par[j] = h.parent[j];
#pragma unroll
        for (int j = NJ - 1; j >= 1; j--)
          sc.Fc[par[j] * 6 + lane] += sc.Fc[j * 6 + lane];
      }
      SMPC_LANES_END_WAVE
      // composite force below joint i minus the contact wrenches applied below it
      SMPC_LANES(NT)
      if (lane < NJ)
      {
        const int i = lane;
        SV F = ldsv(&sc.Fc[i * 6]);
        if constexpr (FS == 3) // a LOCAL contact force turns with the foot: it joins the composite force below joint i
          for (int f = 0; f < NF; f++)
            if ((h.anc[h.foot_joint[f]] >> i) & 1u)
              F = F - ldsv(&sd.Wc[f * 6]);
        stsv(&sc.I[i * 6], F); // (the body inertias are dead: their block takes the composite forces)
      }
      SMPC_LANES_END_WAVE
    }
    ftick(fp, 8);
    // ---- per dof: d_k, A_k ; ----
    SMPC_LANES(NT)
    if (lane < NV)
    {
      const int k = lane, i = jof(k), lam = h.parent[i];
      const SV s = ldsv(&sc.S[k * 6]);
      SV d = sv0(), A = crm(sv0() - g6, s);
      if (lam >= 0)
      {
        const SV vl = ldsv(&sc.vel[lam * 6]);
        d = crm(vl, s);
        A = crm(ldsv(&sc.acc[lam * 6]) - g6, s) + crm(vl, d);
      }
      stsv(&sd.dk[k * 6], d);
      stsv(&sd.Ak[k * 6], A);
    }
    SMPC_LANES_END_WAVE
    ftick(fp, 9);
    if (term)
    {
      full_gn_rows<D>(sc, sd, sw, true);
      return;
    }
    // ---- partial derivatives of RNEA(q, v, a) - J^T lam at the solution: R1 = [r1q | r1v | -S], as masked small GEMMs ----
    //   joint(k) at or above joint(m) (s = joint(m)):  r1q(m,k) = (Ic_s S_m) . A_k + D_m . d_k ,  r1v(m,k) = (Ic_s S_m) . E_k + D_m . S_k
    //       D_m = Bc_s^T S_m - S_m x* hc_s ,  E_k = v_i x S_k + d_k          (S_m . (d x* h) = -d . (S_m x* h))
    //   joint(k) strictly below joint(m) (s = i = joint(k)):  r1q(m,k) = S_m . Yq_k ,  r1v(m,k) = S_m . Yv_k
    //       Yq_k = Ic_i A_k + Bc_i d_k + d_k x* hc_i + S_k x* Fgc_i ,  Yv_k = Bc_i S_k + S_k x* hc_i + Ic_i E_k
    // The per-dof vectors live in the force rows of JT (written only afterwards, by the contact partials).
    {
      // (E_k is not stored: the product that needs it forms its entries from vel, S and d_k as it fetches them)
      constexpr int MR = FullDerivWide<D>::R1ROWS; // (kinodynamics variant: the six base rows)
      double * Dm_ = wtmp, * Yq_ = Dm_ + MR * 6, * Yv_ = Yq_ + NV * 6; // (D_m: one per ROW of R1)
      static_assert((MR + 2 * NV) * 6 <= NCM * NCOL && (FS != 6 || (6 * MR + 3 * NV) * NF <= NCM * NCOL), "per-dof vectors fit the force rows of JT");
      SMPC_LANES(NT)
      {
        if (lane < NV)
        {
          const int k = lane, i = jof(k);
          const SV Sk = ldsv(&sc.S[k * 6]), d = ldsv(&sd.dk[k * 6]), A = ldsv(&sd.Ak[k * 6]);
          const SI Ici = ldsi(&sc.Ic[i * 10]);
          const SV hci = ldsv(&sc.hc[i * 6]);
          const double * Bc = &sd.Bc[i * 36];
          const double dv[6] = {d.l.x, d.l.y, d.l.z, d.a.x, d.a.y, d.a.z};
          const double sv[6] = {Sk.l.x, Sk.l.y, Sk.l.z, Sk.a.x, Sk.a.y, Sk.a.z};
          double o1[6], o2[6], o3[6] = {0, 0, 0, 0, 0, 0};
          for (int r = 0; r < 6; r++)
          {
            double a1 = 0.0, a2 = 0.0;
            for (int c = 0; c < 6; c++)
            {
              const double bv = Bc[r * 6 + c];
              a1 += bv * dv[c];
              a2 += bv * sv[c];
              o3[c] += bv * sv[r]; // Bc^T S_k
            }
            o1[r] = a1;
            o2[r] = a2;
          }
          const SV E = crm(ldsv(&sc.vel[i * 6]), Sk) + d;
          const SV Cm = crf(Sk, hci);
          const SV Dm = SV{mk3(o3[0], o3[1], o3[2]), mk3(o3[3], o3[4], o3[5])} - Cm;
          const SV Yq = Ici * A + SV{mk3(o1[0], o1[1], o1[2]), mk3(o1[3], o1[4], o1[5])} + crf(d, hci) + crf(Sk, ldsv(&sc.I[i * 6]));
          const SV Yv = SV{mk3(o2[0], o2[1], o2[2]), mk3(o2[3], o2[4], o2[5])} + Cm + Ici * E;
          if (k < MR)
            stsv(&Dm_[k * 6], Dm);
          stsv(&Yq_[k * 6], Yq);
          stsv(&Yv_[k * 6], Yv);
        }
        if constexpr (!D::KINO)
          for (int idx = lane; idx < NV * NU; idx += NT)
          {
            const int m = idx / NU, j = idx % NU;
            sw.R1[m * NCOL + 2 * NV + j] = m == 6 + j ? -1.0 : 0.0;
          }
      }
      SMPC_LANES_END_WAVE
      fwave_gemm<MR, 2 * NV, 12>(
        [&](int m, int kk) { return kk < 6 ? sc.IcS[m * 6 + kk] : Dm_[m * 6 + kk - 6]; },
        [&](int kk, int j) {
          const int jj = j < NV ? j : j - NV;
          if (j >= NV && kk < 6)
          { // E_k = v_i x S_k + d_k
            const SV E = crm(ldsv(&sc.vel[jof(jj) * 6]), ldsv(&sc.S[jj * 6])) + ldsv(&sd.dk[jj * 6]);
            return kk < 3 ? v3c(E.l, kk) : v3c(E.a, kk - 3);
          }
          const double * src = j < NV ? (kk < 6 ? &sd.Ak[jj * 6 + kk] : &sd.dk[jj * 6 + kk - 6]) : &sc.S[jj * 6 + kk - 6];
          return *src;
        },
        [&](int m, int j, double v) {
          const int jm = jof(m), i = jof(j < NV ? j : j - NV);
          sw.R1[m * NCOL + j] = ((h.anc[jm] >> i) & 1u) ? v : 0.0;
        });
      fwave_gemm<MR, 2 * NV, 6>(
        [&](int m, int kk) { return sc.S[m * 6 + kk]; },
        [&](int kk, int j) { return j < NV ? Yq_[j * 6 + kk] : Yv_[(j - NV) * 6 + kk]; },
        [&](int m, int j, double v) {
          const int jm = jof(m), i = jof(j < NV ? j : j - NV);
          if (!((h.anc[jm] >> i) & 1u) && ((h.anc[i] >> jm) & 1u))
            sw.R1[m * NCOL + j] = v;
        });
      if constexpr (FS == 6)
      {
        // - d(J^T lam)/dq_k of world-aligned wrenches: the wrench keeps its axes, its point of application moves (first product);
        // the columns S_m below joint(k) -- and the other base columns, for a base dof -- move with S_k (second product):
        //   r1q(m,k) -= sum_f [m, k above foot f] ( (f_f x S_m.ang) . dp_kf + [S_m moves with S_k] (S_m x* W_f) . S_k )
        double * Hm_ = wtmp, * Pv_ = Hm_ + MR * 6 * NF; // (G_m = f_f x S_m.ang is formed by the product that needs it ; H_m: one per ROW of R1)
        SMPC_LANES(NT)
        for (int idx = lane; idx < NV * NF; idx += NT)
        {
          const int m = idx / NF, f = idx % NF;
          const int l = h.foot_joint[f];
          const bool on = ((mask >> f) & 1u) && ((h.anc[l] >> jof(m)) & 1u);
          const SV Sm = ldsv(&sc.S[m * 6]), W = ldsv(&sd.Wc[f * 6]);
          const V3 z = mk3(0, 0, 0);
          if (m < MR)
            stsv(&Hm_[m * 6 * NF + 6 * f], on ? crf(Sm, W) : SV{z, z});
          st3(&Pv_[m * 3 * NF + 3 * f], on ? Sm.l + cross(Sm.a, ld3(&sc.footp[f * 3])) : z);
        }
        SMPC_LANES_END_WAVE
        fwave_gemm<MR, NV, 3 * NF>(
          [&](int m, int kk) {
            const int f = kk / 3;
            const bool on = ((mask >> f) & 1u) && ((h.anc[h.foot_joint[f]] >> jof(m)) & 1u);
            return on ? -v3c(cross(ld3(&sd.Wc[f * 6]), ld3(&sc.S[m * 6 + 3])), kk % 3) : 0.0;
          },
          [&](int kk, int j) { return Pv_[j * 3 * NF + kk]; },
          [&](int m, int j) { return sw.R1[m * NCOL + j]; }, [&](int m, int j, double v) { sw.R1[m * NCOL + j] = v; });
        fwave_gemm<MR, NV, 6 * NF>(
          [&](int m, int kk) { return -Hm_[m * 6 * NF + kk]; },
          [&](int kk, int j) {
            const int f = kk / 6;
            const double x = sc.S[j * 6 + kk % 6];
            return (((mask >> f) & 1u) && ((h.anc[h.foot_joint[f]] >> jof(j)) & 1u)) ? x : 0.0;
          },
          [&](int m, int j) { return sw.R1[m * NCOL + j]; },
          [&](int m, int j, double v) {
            const int jm = jof(m), i = jof(j);
            const bool moves = (jm != i && ((h.anc[jm] >> i) & 1u)) || (jm == 0 && i == 0);
            if (moves)
              sw.R1[m * NCOL + j] = v;
          });
      }
      SMPC_LANES(NT)
      for (int idx = lane; idx < NCM * NCOL; idx += NT)
        sw.JT[idx] = 0.0;
      SMPC_LANES_END_WAVE
    }
    ftick(fp, 10);
    if constexpr (D::KINO)
    {
      // ---- kinodynamics variant: the base rows  M_bb da_b = -(r1q | r1v) dq,dv + (J^T)_b d lam - M_bj d a_j ; the joint accelerations are
      //      controls (r1()).  Right-hand sides in place on the six rows of R1, then R1 <- M_bb^-1 R1 (lane = column) ----
      const double * const kM = SC::DYN_OUT ? sw.Mi_() : sc.M, * const kJ = SC::DYN_OUT ? sw.J_() : sc.J, * const kGi = SC::DYN_OUT ? sw.Gi_() : sc.Gi;
      // (lane = column: the six right-hand-side entries of a column come from one source -- a column of R1, a row of J, a row of M_bj -- through
      //  one pointer and stride per lane, read together; the product with M_bb^-1 follows in the same lane)
      SMPC_LANES(NT)
      for (int c = lane; c < NCOL; c += NT)
      {
        const bool isx = c < 2 * NV, isf = !isx && c < 2 * NV + NCM;
        const int cf = isf ? c - 2 * NV : 0, f = cf / FS, j = cf % FS;
        const bool on = (mask >> f) & 1u;
        const int cc = on ? __builtin_popcount(mask & ((1u << f) - 1u)) : 0;
        const int ca = (!isx && !isf) ? c - 2 * NV - NCM : 0;
        const double * const src = isx ? &sw.R1[c] : (isf ? &kJ[(FS * cc + j) * NV] : &kM[6 + ca]);
        const int stride = isx ? NCOL : (isf ? 1 : NV);
        const double sg = isf ? (on ? 1.0 : 0.0) : -1.0;
        double t[6], o[6];
#pragma unroll
        for (int b = 0; b < 6; b++)
          t[b] = src[b * stride];
#pragma unroll
        for (int b = 0; b < 6; b++)
          t[b] *= sg;
#pragma unroll
        for (int b = 0; b < 6; b++)
        {
          double acc = 0.0;
#pragma unroll
          for (int d = 0; d < 6; d++)
            acc += kGi[b * 6 + d] * t[d];
          o[b] = acc;
        }
#pragma unroll
        for (int b = 0; b < 6; b++)
          sw.R1[b * NCOL + c] = o[b];
      }
      SMPC_LANES_END_WAVE
      // ---- Jacobian of the centroidal_derivative residual hdot(u, q) (rows 0 .. 5 of JT):  d/dq_k = sum_f (dp_f/dq_k - dc/dq_k) x f_f on the
      //      angular rows;  d/du_f = [I 0 ; [p_f - c]x I] for a foot in contact ; frame-velocity rows Cv of the feet in contact ----
      SMPC_LANES(NT)
      {
        const V3 com = ld3(sc.com);
        if (lane < NV)
        {
          const int k = lane, i = jof(k);
          const SV sk = ldsv(&sc.S[k * 6]);
          const V3 jc = (1.0 / h.total_mass) * (ldsi(&sc.Ic[i * 10]) * sk).l;
          V3 acc = mk3(0, 0, 0);
          for (int f = 0; f < NF; f++)
            if ((mask >> f) & 1u)
            {
              const bool on = (h.anc[h.foot_joint[f]] >> i) & 1u;
              const V3 pf = on ? sk.l + cross(sk.a, ld3(&sc.footp[f * 3])) : mk3(0, 0, 0);
              acc = acc + cross(pf - jc, ld3(&sc.u[FS * f]));
            }
          sw.JT[3 * NCOL + k] = acc.x;
          sw.JT[4 * NCOL + k] = acc.y;
          sw.JT[5 * NCOL + k] = acc.z;
        }
        for (int idx = lane; idx < 6 * NCM; idx += NT)
        {
          const int r = idx / NCM, c = idx % NCM, f = c / FS, j = c % FS;
          double v = 0.0;
          if ((mask >> f) & 1u)
          {
            if (j < 3)
            {
              const V3 e = mk3(j == 0, j == 1, j == 2);
              const V3 x = cross(ld3(&sc.footp[f * 3]) - com, e);
              v = r < 3 ? (r == j ? 1.0 : 0.0) : (r == 3 ? x.x : (r == 4 ? x.y : x.z));
            }
            else
              v = r == j ? 1.0 : 0.0;
          }
          sw.JT[r * NCOL + 2 * NV + c] = v;
        }
        if constexpr (D::NVEL > 0)
        {
          // (one task = the three entries of a (foot, linear | angular) row triple in column k: they share everything but the component)
          static_assert(D::NVEL == 0 || FS == 6, "frame-velocity rows of 6-D feet");
          for (int idx = lane; idx < NF * 2 * D::NDX; idx += NT)
          {
            const int f = idx / (2 * D::NDX), half = (idx / D::NDX) % 2, k = idx % D::NDX;
            const int jf = h.foot_joint[f], kk = k < NV ? k : k - NV;
            V3 o = mk3(0, 0, 0);
            if (((mask >> f) & 1u) && ((h.anc[jf] >> jof(kk)) & 1u))
            {
              const V3 p = ld3(&sc.footp[f * 3]);
              const SV m = k >= NV ? ldsv(&sc.S[kk * 6]) : ldsv(&sd.dk[kk * 6]);
              o = tmul(ldm3(&sc.oR[jf * 9]), half == 0 ? m.l + cross(m.a, p) : m.a);
            }
            double * dst = &sw.Cv[(FS * f + 3 * half) * D::NDX + k];
            dst[0] = o.x;
            dst[D::NDX] = o.y;
            dst[2 * D::NDX] = o.z;
          }
        }
      }
      SMPC_LANES_END_WAVE
      full_gn_rows<D>(sc, sd, sw, false);
      ftick(fp, 12);
      return;
    }
    // ---- partial derivatives of the contact acceleration residual (classical acceleration, contact frame, corrector) ----
    if constexpr (FS == 6)
    {
      SMPC_LANES(NT)
      for (int idx = lane; idx < NF * NV; idx += NT)
      {
        const int f = idx / NV, k = idx % NV;
        const int l = h.foot_joint[f], i = jof(k);
        if (((mask >> f) & 1u) && ((h.anc[l] >> i) & 1u))
        {
          const int c = __builtin_popcount(mask & ((1u << f) - 1u));
          const V3 p = ld3(&sc.footp[f * 3]);
          const SV vl = ldsv(&sc.vel[l * 6]), al = ldsv(&sc.acc[l * 6]); // (acc: at the solution)
          const V3 w = vl.a, vp = vl.l + cross(w, p);
          const V3 ap = al.l + cross(al.a, p) + cross(w, vp);
          const SV Sk = ldsv(&sc.S[k * 6]), d = ldsv(&sd.dk[k * 6]);
          const int lam = h.parent[i];
          SV A = sv0();
          if (lam >= 0)
            A = crm(ldsv(&sc.acc[lam * 6]), Sk) + crm(ldsv(&sc.vel[lam * 6]), d) + crm(d, vl);
          const V3 pv = Sk.l + cross(Sk.a, p); // d(point position)/dq_k = d(point velocity)/dv_k
          const V3 aq = A.l + cross(A.a, p) + cross(d.a, vp) + cross(w, d.l + cross(d.a, p));
          const SV Av = d + crm(Sk, vl - ldsv(&sc.vel[i * 6]));
          const V3 av = Av.l + cross(Av.a, p) + cross(Sk.a, vp) + cross(w, pv);
          const V3 vq = d.l + cross(d.a, p);
          // world-aligned components of body-fixed vectors turn with the body: + S_k.a x u
          const V3 lq = aq + cross(Sk.a, ap), lvq = vq + cross(Sk.a, vp);
          const V3 aqq = A.a + cross(Sk.a, al.a), wq = d.a + cross(Sk.a, w);
          // d log3(R) for a rotation increment expressed in the world frame: inverse left Jacobian = Jlog3(-phi)
          const V3 rq = Jlog3((-1.0) * ld3(&sc.rotl[f * 3])) * Sk.a;
          double * r2 = &sw.JT[(6 * c) * NCOL];
          const double lqv[3] = {lq.x, lq.y, lq.z}, lvqv[3] = {lvq.x, lvq.y, lvq.z}, pvv[3] = {pv.x, pv.y, pv.z};
          const double aqv[3] = {aqq.x, aqq.y, aqq.z}, wqv[3] = {wq.x, wq.y, wq.z}, rqv[3] = {rq.x, rq.y, rq.z};
          const double avv[3] = {av.x, av.y, av.z}, ava[3] = {Av.a.x, Av.a.y, Av.a.z}, sav[3] = {Sk.a.x, Sk.a.y, Sk.a.z};
          for (int r = 0; r < 3; r++)
          {
            r2[r * NCOL + k] = lqv[r] + h.Kd[r] * lvqv[r] + h.Kp[r] * pvv[r];
            r2[(3 + r) * NCOL + k] = aqv[r] + h.Kd[3 + r] * wqv[r] + h.Kp[3 + r] * rqv[r];
            r2[r * NCOL + NV + k] = avv[r] + h.Kd[r] * pvv[r];
            r2[(3 + r) * NCOL + NV + k] = ava[r] + h.Kd[3 + r] * sav[r];
          }
        }
      }
      SMPC_LANES_END_WAVE
    }
    else
    {
    SMPC_LANES(NT)
    for (int idx = lane; idx < NF * NV; idx += NT)
    {
      const int f = idx / NV, k = idx % NV;
      const int l = h.foot_joint[f], i = jof(k);
      if (((mask >> f) & 1u) && ((h.anc[l] >> i) & 1u))
      {
        const int c = __builtin_popcount(mask & ((1u << f) - 1u));
        const M3 Rf = ldm3(&sc.oR[l * 9]);
        const V3 p = ld3(&sc.footp[f * 3]);
        const SV vl = ldsv(&sc.vel[l * 6]);
        const V3 w = vl.a, vp = vl.l + cross(w, p);
        const SV Sk = ldsv(&sc.S[k * 6]), d = ldsv(&sd.dk[k * 6]);
        const int lam = h.parent[i];
        // non-rigid part of the variation of the body acceleration (kinematic: no gravity here)
        SV A = sv0();
        if (lam >= 0)
          A = crm(ldsv(&sc.acc[lam * 6]), Sk) + crm(ldsv(&sc.vel[lam * 6]), d) + crm(d, vl);
        const V3 aq = A.l + cross(A.a, p) + cross(d.a, vp) + cross(w, d.l + cross(d.a, p));
        const SV Av = d + crm(Sk, vl - ldsv(&sc.vel[i * 6]));
        const V3 av = Av.l + cross(Av.a, p) + cross(Sk.a, vp) + cross(w, Sk.l + cross(Sk.a, p));
        const V3 vq = d.l + cross(d.a, p), vv = Sk.l + cross(Sk.a, p);
        const V3 cq = tmul(Rf, aq), cv = tmul(Rf, av), eq = tmul(Rf, vq), ev = tmul(Rf, vv), pq = (-1.0) * tmul(Rf, Sk.l);
        double * r2 = &sw.JT[(3 * c) * NCOL];
        r2[0 * NCOL + k] = cq.x + h.Kd[0] * eq.x - h.Kp[0] * pq.x;
        r2[1 * NCOL + k] = cq.y + h.Kd[1] * eq.y - h.Kp[1] * pq.y;
        r2[2 * NCOL + k] = cq.z + h.Kd[2] * eq.z - h.Kp[2] * pq.z;
        r2[0 * NCOL + NV + k] = cv.x + h.Kd[0] * ev.x;
        r2[1 * NCOL + NV + k] = cv.y + h.Kd[1] * ev.y;
        r2[2 * NCOL + NV + k] = cv.z + h.Kd[2] * ev.z;
      }
    }
    SMPC_LANES_END_WAVE
    }
    ftick(fp, 11);
    // ---- [M -J^T; J mu] [da; dlam] = -[r1; r2] ----
    if constexpr (!D::WIDE_DEV)
    {
      // everything in LDS (the quadruped): Mr = M^-1 R1 ; rhs = J Mr - r2 ; dlam = G^-1 rhs ; da = -Mr + M^-1 J^T dlam, each in place -- four short
      // products whose operands are an LDS round trip away (the one-product form below measured 3 % slower here: 11.7 -> 12.0 ms per launch)
      fwave_gemm<NV, NCOL, NV>(
        [&](int i, int k) { return sc.M[k * NV + i]; }, [&](int k, int j) { return sw.R1[k * NCOL + j]; },
        [&](int i, int j, double v) { sw.R1[i * NCOL + j] = v; });
      fwave_gemm<NCM, NCOL, NV>(
        [&](int i, int k) { return sc.J[i * NV + k]; }, [&](int k, int j) { return sw.R1[k * NCOL + j]; },
        [&](int i, int j) { return -sw.JT[i * NCOL + j]; }, [&](int i, int j, double v) { sw.JT[i * NCOL + j] = v; });
      fwave_gemm<NCM, NCOL, NCM>(
        [&](int i, int k) { return sc.Gi[k * NCM + i]; }, [&](int k, int j) { return sw.JT[k * NCOL + j]; },
        [&](int i, int j, double v) { sw.JT[i * NCOL + j] = v; });
      fwave_gemm<NV, NCOL, NCM>(
        [&](int i, int k) { return sc.W[i * NR + 1 + k]; }, [&](int k, int j) { return sw.JT[k * NCOL + j]; },
        [&](int i, int j) { return -sw.R1[i * NCOL + j]; }, [&](int i, int j, double v) { sw.R1[i * NCOL + j] = v; });
    }
    else
    {
    // One product with the inverse of the contact KKT matrix (blocks formed by full_dynamics_phases<D, true>; its transpose lies in the
    // device slice, the LDS of the dynamics block is the derivative scratch by now), in place on [R1 ; force rows of JT]: every operand entry is
    // fetched before the first result is stored.
      constexpr int NK = NV + NCM;
      fwave_gemm<NK, NCOL, NK, 2>(
        [&](int i, int k) { return sw.dyn[k * NK + i]; },
        [&](int k, int j) { return *(k < NV ? &sw.R1[k * NCOL + j] : &sw.JT[(k - NV) * NCOL + j]); },
        [&](int i, int j, double v) { *(i < NV ? &sw.R1[i * NCOL + j] : &sw.JT[(i - NV) * NCOL + j]) = v; });
    }
    ftick(fp, 12);
    full_gn_rows<D>(sc, sd, sw, false);
    ftick(fp, 19);
    (void)NGN;
  }

  // entry (r, k) of the block-diagonal Gauss-Newton weight W~ = blockdiag(w_forces per contact, w_cent (x 10 at the terminal
  // node), w_frame per foot) over the rows of JT
  template <class D, class SC>
  SMPC_DEV double full_wtilde(const FullHead<D> & h, int r, int k, bool term, bool tcs = false)
  {
    constexpr int NCM = D::NCM, FS = D::FS, NGN = SC::NGN;
    if constexpr (D::KINO) // frame-velocity rows behind the cost rows: folded equality rows, weight 1 / mu (Cv^T Cv / mu)
      if (r >= NGN && r < NGN + D::NVEL)
        return (!term && r == k) ? 1.0 / h.mu : 0.0;
    if (r >= NGN || k >= NGN)
      return 0.0;
    if (r < NCM && term) // terminal node: the first three rows hold the terminal constraint's Jacobian, weight 1 / mu (C^T C / mu)
      return (tcs && r < 3 && r == k) ? 1.0 / h.mu : 0.0;
    if constexpr (D::KINO) // rows 0 .. 5: centroidal_derivative residual, weight w_centder
      if (r < NCM)
        return (r < 6 && k < 6) ? h.w_centder[r * 6 + k] : 0.0;
    if (r < NCM)
      return (term || k >= NCM || r / FS != k / FS) ? 0.0 : h.w_forces[(r % FS) * FS + k % FS];
    if (r < NCM + 6)
      return (k < NCM || k >= NCM + 6) ? 0.0 : (term ? 10.0 : 1.0) * h.w_cent[(r - NCM) * 6 + k - NCM];
    constexpr int PF = D::PF;
    const int a = r - NCM - 6, bb = k - NCM - 6;
    return (term || bb < 0 || a / PF != bb / PF) ? 0.0 : h.w_frame[(a % PF) * FS + bb % PF];
  }

  // entry (i, j) of the state-cost Hessian Jx^T w_x Jx, Jx = blockdiag(Jlog6, I)
  template <class D, class SC, class SD>
  SMPC_DEV double full_state_hessian(const SC & sc, const SD & sd, const DevModel<D> & mg, int i, int j)
  {
    constexpr int NDX = D::NDX;
    const FullHead<D> & h = sc.h;
    if (i >= 6 && j >= 6)
      return h.w_diag ? (i == j ? h.wxd[i] : 0.0) : mg.w_x[i * NDX + j];
    if (i < 6 && j < 6)
      return sd.JWJ_()[i * 6 + j];
    return i < 6 ? sd.WJl_()[j * 6 + i] : sd.WJl_()[i * 6 + j];
  }

  // tables of the state cost: WJl = w_x[:, 0:6] Jl (NDX x 6), JWJ = Jl^T w_x[0:6, 0:6] Jl
  template <class D, class SC, class SD>
  SMPC_DEV void full_state_tables(SC & sc, SD & sd, const DevModel<D> & mg)
  {
    constexpr int NT = 64;
    constexpr int NDX = D::NDX;
    const FullHead<D> & h = sc.h;
    SMPC_LANES(NT)
    for (int idx = lane; idx < NDX * 6; idx += NT)
    {
      const int a = idx / 6, k = idx % 6;
      double s = 0.0;
      if (h.w_diag)
        s = a < 6 ? h.wxd[a] * sd.Jl[a * 6 + k] : 0.0;
      else
        for (int b = 0; b < 6; b++)
          s += mg.w_x[a * NDX + b] * sd.Jl[b * 6 + k];
      sd.WJl_()[idx] = s;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < 36)
    {
      const int i = lane / 6, j = lane % 6;
      double s = 0.0;
      for (int a = 0; a < 6; a++)
        s += sd.Jl[a * 6 + i] * sd.WJl_()[a * 6 + j];
      sd.JWJ_()[lane] = s;
    }
    SMPC_LANES_END_WAVE
  }

  // Gauss-Newton Hessian  [Q S; S^T R] = H_0 + JT^T (W~ JT) + preg I  on the FP64 matrix cores:  W~ JT in accumulator tiles
  // (tile row R of the NGN rows, tile column J of the NXU columns); the accumulator layout (rows lr + 4 v of tile row R) is the
  // B-operand layout of K-step 4 R + v of the second product, so the weighted Jacobian never leaves the registers.  Upper 16 x 16
  // tiles of the (x, u) grid are written: Q (upper tiles; mirrored when `mirror`), S, R (readers take (min, max) indices).
  template <class D, class SC, class SD>
  SMPC_DEV void full_hessian_mfma(SC & sc, SD & sd, FullDerivWide<D> & sw, const DevModel<D> & mg, bool term, double preg, double * Qd, double * Sd, double * Rd, bool mirror, bool tcs = false, FullProf * fpp = nullptr)
  {
    constexpr int NT = 64;
    constexpr int NDX = D::NDX, NU = D::NU, NXU = D::NXU, NCOL = SC::NCOL, NGN0 = SC::NGN;
    constexpr int NGN = NGN0 + D::NVEL; // (kinodynamics variant: the folded frame-velocity rows ride behind the cost rows, from sw.Cv)
    constexpr int NTC = (NXU + 15) / 16, NTR = (NGN + 15) / 16, KS = (NGN + 3) / 4;
    static_assert(NCOL == NXU, "the derivative columns (q, v, tau) are the (x, u) columns");
    const FullHead<D> & h = sc.h;
    SMPC_ACC(wj, NT, NTR * NTC);
    SMPC_ACC(qa, NT, NTC * (NTC + 1) / 2);
    SMPC_PLA(double, jtv, NT, NTC);
    SMPC_PLA(double, wop, NT, NTR);
    SMPC_LANES(NT)
    {
      const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
      for (int tt = 0; tt < NTR * NTC; tt++)
#pragma unroll
        for (int v = 0; v < 4; v++)
          SMPC_ACCV(wj, tt, v) = 0.0;
      // H_0: state-cost block (with the Jlog6 base block), control weight, primal regularisation
#pragma unroll
      for (int I = 0; I < NTC; I++)
#pragma unroll
        for (int J = I; J < NTC; J++)
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            const int row = 16 * I + lr + 4 * v, col = 16 * J + lc;
            double val = 0.0;
            if (row < NDX && col < NDX)
              val = full_state_hessian<D>(sc, sd, mg, row, col);
            else if (!term && row >= NDX && row < NXU && col >= NDX && col < NXU)
              val = h.w_diag ? (row == col ? h.wud[row - NDX] : 0.0) : mg.w_u[(row - NDX) * NU + col - NDX];
            if (row == col && row < (term ? NDX : NXU))
              val += preg;
            SMPC_ACCV(qa, tix<NTC>(I, J), v) = val;
          }
    }
    SMPC_LANES_END_WAVE
    // Blocks with the rows of the Jacobian in the device slice: a lane's entries of WIN K-steps are in flight ahead of the matrix instructions
    // (the biped's full dynamics: all of them, fetched once for the two products; its kinodynamics variant: four K-steps, each product fetches);
    // entries outside the rows are zeroed where a K-step uses them (a select on a loaded value is a wait for the load).
    if (fpp)
      ftick(*fpp, 26);
    constexpr int WIN = !D::WIDE_DEV ? 0 : (D::KINO ? (KS < 4 ? KS : 4) : KS);
    constexpr bool PRE = WIN > 0;
    SMPC_PLA(double, jta, NT, PRE ? WIN * NTC : 1);
    // raw entries of K-step ks (rows 4 ks .. 4 ks + 3) into their slot of the window
    auto prefetch = [&](int ks) {
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
        const int r = 4 * ks + lr;
        const bool velr = D::KINO && r >= NGN0;
        const double * row = velr ? &sw.Cv[((r < NGN ? r : NGN0) - NGN0) * (D::KINO ? NDX : 0)] : &sw.JT[(r < NGN0 ? r : 0) * NCOL];
        const int cmax = velr ? NDX : NCOL;
#pragma unroll
        for (int J = 0; J < NTC; J++)
        {
          const int c = 16 * J + lc;
          SMPC_PLV(jta)[(ks % (PRE ? WIN : 1)) * NTC + J] = row[c < cmax ? c : 0];
        }
      }
      SMPC_LANES_END_WAVE
    };
    if constexpr (PRE)
    {
#pragma unroll
      for (int ks = 0; ks < WIN; ks++)
        prefetch(ks);
      SMPC_SCHED_FENCE();
    }
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
    {
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
        const int r = 4 * ks + lr;
#pragma unroll
        for (int J = 0; J < NTC; J++)
        {
          const int c = 16 * J + lc;
          const bool velr = D::KINO && r >= NGN0;
          const bool ok = r < NGN && (velr ? (c < NDX && !term) : c < NCOL) && (!term || (r >= D::NCM && r < D::NCM + 6) || (tcs && r < 3)); // terminal node: momentum (+ constraint) rows only
          if constexpr (PRE)
            SMPC_PLV(jtv)[J] = ok ? SMPC_PLV(jta)[(ks % WIN) * NTC + J] : 0.0;
          else
          {
          const double * row = (ok && velr) ? &sw.Cv[(r - NGN0) * (D::KINO ? NDX : 0)]
                                            : ((ok && r < D::NCM) ? &sw.JT[r * NCOL] : full_gn_row<D>(sc, sw, ok ? r : D::NCM));
          const double v = row[(c < NCOL && ok) ? c : 0];
          SMPC_PLV(jtv)[J] = ok ? v : 0.0;
          }
        }
#pragma unroll
        for (int R = 0; R < NTR; R++)
          SMPC_PLV(wop)[R] = full_wtilde<D, SC>(h, 16 * R + lc, r, term, tcs);
      }
      SMPC_LANES_END_WAVE
      if constexpr (PRE && WIN < KS)
      {
        // the slot is free: the next K-step of the window
        if (ks + WIN < KS)
          prefetch(ks + WIN);
        SMPC_SCHED_FENCE();
      }
#pragma unroll
      for (int R = 0; R < NTR; R++)
#pragma unroll
        for (int J = 0; J < NTC; J++)
          SMPC_MFMA(wj, R * NTC + J, wop, R, jtv, J);
      if constexpr (PRE && WIN < KS)
        SMPC_SCHED_FENCE();
    }
    if (fpp)
      ftick(*fpp, 27);
    if constexpr (PRE && WIN < KS)
    {
#pragma unroll
      for (int ks = 0; ks < WIN; ks++)
        prefetch(ks);
      SMPC_SCHED_FENCE();
    }
    SMPC_PLA(double, bv, NT, NTC);
#pragma unroll
    for (int R = 0; R < NTR; R++)
#pragma unroll
      for (int v = 0; v < 4; v++)
      {
        if (16 * R + 4 * v >= NGN)
          continue;
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
          const int r = 4 * (4 * R + v) + lr;
#pragma unroll
          for (int J = 0; J < NTC; J++)
          {
            const int c = 16 * J + lc;
            const bool velr = D::KINO && r >= NGN0;
            const bool ok = r < NGN && (velr ? (c < NDX && !term) : c < NCOL) && (!term || (r >= D::NCM && r < D::NCM + 6) || (tcs && r < 3));
            if constexpr (PRE)
              SMPC_PLV(jtv)[J] = ok ? SMPC_PLV(jta)[((4 * R + v) % WIN) * NTC + J] : 0.0;
            else
            {
            const double * row = (ok && velr) ? &sw.Cv[(r - NGN0) * (D::KINO ? NDX : 0)]
                                              : ((ok && r < D::NCM) ? &sw.JT[r * NCOL] : full_gn_row<D>(sc, sw, ok ? r : D::NCM));
            const double x = row[(c < NCOL && ok) ? c : 0];
            SMPC_PLV(jtv)[J] = ok ? x : 0.0;
            }
            SMPC_PLV(bv)[J] = SMPC_ACCV(wj, R * NTC + J, v);
          }
        }
        SMPC_LANES_END_WAVE
        if constexpr (PRE && WIN < KS)
        {
          if (4 * R + v + WIN < KS)
            prefetch(4 * R + v + WIN);
          SMPC_SCHED_FENCE();
        }
#pragma unroll
        for (int I = 0; I < NTC; I++)
#pragma unroll
          for (int J = I; J < NTC; J++)
            SMPC_MFMA(qa, tix<NTC>(I, J), jtv, I, bv, J);
        if constexpr (PRE && WIN < KS)
          SMPC_SCHED_FENCE();
      }
    if (fpp)
      ftick(*fpp, 28);
    SMPC_LANES(NT)
    {
      const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
      for (int I = 0; I < NTC; I++)
#pragma unroll
        for (int J = I; J < NTC; J++)
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            const int row = 16 * I + lr + 4 * v, col = 16 * J + lc;
            const double val = SMPC_ACCV(qa, tix<NTC>(I, J), v);
            // (diagonal tiles hold both triangles: only the upper one is written, so that no two lanes target one address)
            if (col < NDX)
            {
              if (row <= col)
              {
                Qd[row * NDX + col] = val;
                if (mirror && row < col)
                  Qd[col * NDX + row] = val;
              }
            }
            else if (col < NXU && Sd != nullptr)
            {
              if (row < NDX)
                Sd[row * NU + col - NDX] = val;
              else if (row <= col)
                Rd[(row - NDX) * NU + col - NDX] = val;
            }
          }
    }
    SMPC_LANES_END_WAVE
  }

  // =============================================================================================
  // fdyn_deriv_body: grid = B * (H+1) (or slots * (H+1) walking the list of undecided instances); block (inst, t);
  // t == H is the terminal node (state cost + 10 x centroidal cost, src/fulldynamics.cpp:418-430).
  // =============================================================================================
  template <class D>
  SMPC_DEV void fdyn_deriv_one(const StageKernelArgs<D> & ka, int inst, int t, int block)
  {
    typedef FullScratch<D, true> SC;
    typedef FullScratchDeriv<D> SD;
    constexpr int NT = 64;
    constexpr int NV = D::NV, NX = D::NX, NDX = D::NDX, NU = D::NU, NC = D::NC, NF = D::NF, NA = D::NA, NCM = D::NCM, NCOL = SC::NCOL, NGN = SC::NGN,
                  NXU = D::NXU, NQ = D::NQ;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const bool term = t == H;
    const DevModel<D> & mg = *b.model;
    SMPC_LDS(SC, scs, 1);
    // LDS decides the resident blocks per CU here (160 KB / block size, one wave per SIMD at most): the quadruped with neither cone nor
    // land rows must stay under 40 KB -- four blocks; at 41.2 KB it ran three and the launch took 15.2 ms instead of 11.9
    static_assert(!(D::NJ == 13 && D::FS == 3 && D::NCONE == 0 && D::NLAND == 0 && !D::KINO) || sizeof(SC) + sizeof(SD) + sizeof(FullDerivWideLds<D>) <= 40960,
                  "fdyn_deriv_body of the point-foot quadruped: 4 resident blocks per CU");
#ifdef SMPC_FDYN_PAD
    SMPC_LDS(double, padlds, SMPC_FDYN_PAD); // (occupancy experiment)
    if (ka.b.B < 0)
      padlds[ka.b.H] = 1.0;
#endif
    SC & sc = scs[0];
    // derivative scratch, per-dof vectors of the R1 fill, the wide blocks: LDS objects of their own, or (WIDE_DEV) an overlay of the
    // dynamics block at the end of the evaluation scratch + a slice of device memory
    SD * sdp;
    double * wtmp;
    FullDerivWide<D> * swp;
    if constexpr (D::WIDE_DEV)
    {
      static_assert(SC::DYN_OUT && sizeof(SC) - offsetof(SC, M) >= sizeof(SD) + sizeof(FullDerivWideLds<D>), "derivative scratch fits the dynamics block");
      sdp = reinterpret_cast<SD *>(sc.dyn_overlay_());
      wtmp = sc.dyn_overlay_() + sizeof(SD) / sizeof(double);
      swp = reinterpret_cast<FullDerivWide<D> *>(ka.wide) + block; // (one slice per block of the persistent GRID: fdyn_deriv_body, smpc_full_engine.h)
      static_assert(sizeof(SC) <= 40960, "fdyn_deriv_body with its wide blocks in device memory: 4 resident blocks per CU");
    }
    else
    {
      SMPC_LDS(SD, sds, 1);
      SMPC_LDS(FullDerivWideLds<D>, swls, 1);
      sdp = &sds[0];
      swp = &swls[0].w;
      wtmp = swls[0].tmp_(*swp);
    }
    SD & sd = *sdp;
    FullDerivWide<D> & sw = *swp;
    const FullHead<D> & h = sc.h;
    const int st = ring_slot(ka.head, t, R);
    const size_t ib = (size_t)inst * R;
    const int sprev = ring_slot(ka.head, t > 0 ? t - 1 : 0, R);
    const int snext = ring_slot(ka.head, term ? t : t + 1, R);
    const double preg = b.scal[(size_t)inst * SC_N + SC_PREG];
    // (flag of the knot's dense cone rows, asked for here -- a scalar load -- and used at the end of the block)
    const bool cone_block_dirty = (D::NCONE > 0 && !term) ? b.lq[((size_t)inst * H + t) * D::LQ_STRIDE + D::O_cdirty] != 0.0 : false;
    const unsigned mask = term ? 0u : (b.stages[t].mask & ((1u << NF) - 1u));
    const unsigned land = (D::NLAND > 0 && !term && mg.land_cstr) ? (b.stages[t].land & mask) : 0u; // feet with land_cstr rows at this stage
    // ---- block inputs ----
    // Every global load of the block's inputs is issued before the first value is committed to LDS (indices clamped, sources chosen by address;
    // written as one loop per array, each a load followed by its LDS store, this phase was a chain of ten memory round trips per block).
    // The stage-shared inputs of the terminal node are read from stage 0 and not used.
    SMPC_LANES(NT)
    {
      constexpr int PX = (NX + NT - 1) / NT, PU = (NU + NT - 1) / NT, PD = (NDX + NT - 1) / NT, PC = (NC + NT - 1) / NT;
      static_assert(NCM <= NT && NF * 3 <= NT, "one entry per lane");
      const int ts = term ? 0 : t;
      double vx[PX], vxn[PX], vxt[PX], vu[PU], vur[PU], vl[PD], vn[PC];
#pragma unroll
      for (int n = 0; n < PX; n++)
      {
        const int i = lane + n * NT < NX ? lane + n * NT : NX - 1;
        vx[n] = b.xs[(ib + st) * NX + i];
        vxn[n] = b.xs[(ib + snext) * NX + i];
        // state_cost target: shared pose part, per-instance base-velocity part
        const double * pt = term ? &mg.x_term[i] : ((i >= NQ && i < NQ + 6) ? &b.vref[(ib + st) * 6 + (i - NQ)] : &b.stages[ts].x_tgt[i]);
        vxt[n] = *pt;
      }
#pragma unroll
      for (int n = 0; n < PU; n++)
      {
        const int i = lane + n * NT < NU ? lane + n * NT : NU - 1;
        vu[n] = b.us[(ib + st) * NU + i];
        vur[n] = b.stages[ts].u_ref[i];
      }
      const double vfr = b.stages[ts].f_ref[lane < NCM ? lane : NCM - 1];
      const double vfo = b.foot_ref[((size_t)inst * H + ts) * NF * 3 + (lane < NF * 3 ? lane : NF * 3 - 1)];
#pragma unroll
      for (int n = 0; n < PD; n++)
        vl[n] = b.lams[(ib + st) * NDX + (lane + n * NT < NDX ? lane + n * NT : NDX - 1)];
#pragma unroll
      for (int n = 0; n < PC; n++)
        vn[n] = b.vs[(ib + st) * NC + (lane + n * NT < NC ? lane + n * NT : NC - 1)];
      SMPC_SCHED_FENCE();
      full_load_head<D, NT>(sc.h, &mg, lane);
#pragma unroll
      for (int n = 0; n < PX; n++)
        if (lane + n * NT < NX)
        {
          sc.x[lane + n * NT] = vx[n];
          sc.xn1[lane + n * NT] = vxn[n];
          sc.x_tgt[lane + n * NT] = vxt[n];
        }
#pragma unroll
      for (int n = 0; n < PU; n++)
        if (lane + n * NT < NU)
        {
          sc.u[lane + n * NT] = term ? 0.0 : vu[n];
          sc.u_ref[lane + n * NT] = term ? 0.0 : vur[n];
        }
      if (lane < NCM)
        sc.f_ref[lane] = term ? 0.0 : vfr;
      if (lane < NF * 3)
        sc.foot_ref[lane] = term ? 0.0 : vfo;
#pragma unroll
      for (int n = 0; n < PD; n++)
        if (lane + n * NT < NDX)
          sc.lam_next[lane + n * NT] = term ? 0.0 : vl[n];
#pragma unroll
      for (int n = 0; n < PC; n++)
        if (lane + n * NT < NC)
          sc.nu[lane + n * NT] = term ? 0.0 : vn[n];
    }
    SMPC_LANES_END_WAVE
    FullProf fp;
    fp.prof = (b.dbg != nullptr && inst == 0 && t == 17 && ka.slots == 0) ? b.dbg : nullptr; // one mid-horizon block
    fp.tprev = SMPC_CLOCK();
    ftick(fp, 0);
    full_dynamics_phases<D, true>(sc, &sd, mg, mask, !term, fp);
    if constexpr (SC::DYN_OUT)
    {
      // a and lam are solved: the factorised block goes to the device slice (every load before the first store), the derivative scratch
      // takes its place, starting with the composite velocity-product matrices
      if (!term)
      {
        if constexpr (!D::KINO)
        {
          constexpr int NK = FullDerivWide<D>::NK, PERK = (NK * NK + NT - 1) / NT;
          SMPC_LANES(NT)
          {
            double v[PERK];
#pragma unroll
            for (int i = 0; i < PERK; i++)
            {
              const int idx = lane + i * NT < NK * NK ? lane + i * NT : NK * NK - 1;
              v[i] = full_kkt_inv<D>(sc, idx % NK, idx / NK);
            }
#pragma unroll
            for (int i = 0; i < PERK; i++)
            {
              const int idx = lane + i * NT < NK * NK ? lane + i * NT : NK * NK - 1;
              sw.dyn[idx] = v[i];
            }
          }
          SMPC_LANES_END_WAVE
        }
        else
        {
        constexpr int N = FullDerivWide<D>::DYN4, PER = (N / 2 + NT - 1) / NT;
        static_assert(N == SC::DYN_DOUBLES && N % 2 == 0 && sizeof(FullDerivWide<D>) % 16 == 0 && offsetof(FullDerivWide<D>, dyn) % 16 == 0 && offsetof(SC, M) % 16 == 0, "16-byte copies");
        SMPC_LANES(NT)
        {
          struct alignas(16) Pair
          {
            double a, b;
          };
          const Pair * src = reinterpret_cast<const Pair *>(sc.M);
          Pair * dst = reinterpret_cast<Pair *>(sw.dyn);
          Pair v[PER];
          // (indices clamped, not predicated: the last pass stores a few pairs twice)
#pragma unroll
          for (int i = 0; i < PER; i++)
          {
            const int idx = lane + i * NT < N / 2 ? lane + i * NT : N / 2 - 1;
            v[i] = src[idx];
          }
#pragma unroll
          for (int i = 0; i < PER; i++)
          {
            const int idx = lane + i * NT < N / 2 ? lane + i * NT : N / 2 - 1;
            dst[idx] = v[i];
          }
        }
        SMPC_LANES_END_WAVE
        }
        ftick(fp, 18);
        full_bc_composites<D>(sc, sd);
      }
      ftick(fp, 17);
    }
    full_eval_tail<D, true>(sc, &sd, mg, mask, land, term, b.lams_e + (ib + st) * NDX, b.vs_e + (ib + st) * NC, &fp);
    ftick(fp, 7);
    full_deriv_phases<D>(sc, sd, sw, wtmp, mg, mask, term, fp);
    ftick(fp, 13);
    full_state_tables<D>(sc, sd, mg);
    double * parts = b.parts0 + ((size_t)inst * (H + 1) + t) * 4;
    // stacked weighted residual [Wrl | Whg | Wrf] for the gradients
    SMPC_LANES(NT)
    {
      for (int r = lane; r < NGN; r += NT)
        sd.dual_()[r] = r < NCM ? (term ? 0.0 : sc.Wrl[r]) : (r < NCM + 6 ? sc.Whg[r - NCM] : (term ? 0.0 : sc.Wrf[r - NCM - 6]));
      // wrench-cone rows: A_cone^T nu per contact (the multipliers of the unmasked rows enter the Lagrangian gradient)
      for (int r = lane; r < NCM; r += NT)
      {
        double acc = 0.0;
        if constexpr (D::NCONE > 0)
        {
          if (!term && h.force_cone)
          {
            const int c = r / D::FS, j = r % D::FS;
            int f = -1, cnt = 0;
            for (int ff = 0; ff < NF; ff++)
              if ((mask >> ff) & 1u)
              {
                if (cnt == c)
                  f = ff;
                cnt++;
              }
            if (f >= 0)
              for (int i = 0; i < D::NCONE1; i++)
                acc += wrench_cone_entry(i, j, h.fric_mu, h.Lfoot, h.Wfoot) * sc.nu[NU + NA + D::NCONE1 * f + i];
          }
        }
        sd.yc_()[r] = acc;
      }
    }
    SMPC_LANES_END_WAVE
    // ---- cost gradients ----
    SMPC_LANES(NT)
    for (int k = lane; k < NXU; k += NT)
    {
      double g = 0.0;
      if (k < NDX)
      {
        if (k < 6)
          for (int a = 0; a < 6; a++)
            g += sd.Jl[a * 6 + k] * sc.Wrx[a];
        else
          g = sc.Wrx[k];
      }
      else if (!term)
        g = sc.Wru[k - NDX];
      // (terminal node: only the momentum rows exist)
      // (the force rows may lie in device memory: the column's entries are read together, not one per pass of the sum)
      double jtc[NCM];
#pragma unroll
      for (int r = 0; r < NCM; r++)
        jtc[r] = term ? 0.0 : sw.JT[r * NCOL + k];
      if (!term)
      {
#pragma unroll
        for (int r = 0; r < NCM; r++)
          g += jtc[r] * sd.dual_()[r];
      }
      if constexpr (D::WIDE_DEV)
      {
        // (momentum / pose rows in the device slice too: again all entries of the column before the sums)
        double jt2c[NGN - NCM];
#pragma unroll
        for (int r = NCM; r < NGN; r++)
          jt2c[r - NCM] = (term && r >= NCM + 6) ? 0.0 : sw.JT[r * NCOL + k];
#pragma unroll
        for (int r = NCM; r < NGN; r++)
          g += (term && r >= NCM + 6) ? 0.0 : jt2c[r - NCM] * sd.dual_()[r];
      }
      else
        for (int r = NCM; r < (term ? NCM + 6 : NGN); r++)
          g += full_gn_row<D>(sc, sw, r)[k] * sd.dual_()[r];
      if (k < NDX)
        sd.gx_()[k] = g;
      else
        sd.gu_()[k - NDX] = g;
      double cqv = 0.0;
      if constexpr (D::KINO)
      {
        if (!term)
        {
          // wrench-cone rows act on the control itself: D^T nu = A_cone^T nu on the wrench columns of the feet in contact
          if (k >= NDX && k < NDX + NCM && ((mask >> ((k - NDX) / D::FS)) & 1u))
            cqv = sd.yc_()[D::FS * __builtin_popcount(mask & ((1u << ((k - NDX) / D::FS)) - 1u)) + (k - NDX) % D::FS];
          // frame-velocity rows: Cv^T nu
          if (k < NDX)
            for (int r = 0; r < D::NVEL; r++)
              cqv += sw.Cv[r * NDX + k] * sc.nu[NU + NA + D::NCD + r];
        }
      }
      else if constexpr (D::NCONE > 0)
        if (!term)
        {
#pragma unroll
          for (int r = 0; r < NCM; r++)
            cqv += jtc[r] * sd.yc_()[r];
        }
      if constexpr (D::NLAND > 0)
        if (land != 0u && k < NDX) // land rows: C_x^T nu (rows of the state only)
          for (int f = 0; f < NF; f++)
            if ((land >> f) & 1u)
              for (int r = 0; r < D::NLAND1; r++)
                cqv += full_land_entry<D>(sc, sd, f, r, k) * sc.nu[NU + NA + D::NCONE + D::NLAND1 * f + r];
      sd.cq_()[k] = cqv;
    }
    SMPC_LANES_END_WAVE
    if (term)
    {
      // ---- terminal node: Q_N = Lxx + preg I, q_N = lx - lambda_H ----
      double * QN = b.QN + (size_t)inst * NDX * NDX;
      double * qN = b.qN + (size_t)inst * NDX;
      // terminal constraint c = com + tau vcom - ref (DCMPositionResidual, reference src/fulldynamics.cpp:433-455): the rows
      // C = [Jcom + tau dvcom/dq | tau Jcom] go where the (absent) force rows of JT would be, v | v+ | c behind them; the product
      // below then carries C^T C / mu, and q_N gets C^T v+
      const bool tcs = b.CN != nullptr;
      double * tv = &sw.JT[3 * NCOL];
      static_assert(NCM >= 4 && NCOL >= 9, "terminal constraint scratch inside the force rows");
      if (tcs)
      {
        SMPC_LANES(NT)
        {
          const double im = 1.0 / h.total_mass, tau = b.dcm_tau;
          const double * hrow = full_gn_row<D>(sc, sw, NCM); // momentum rows [dh/dq | Ag | 0]
          for (int idx = lane; idx < 3 * NCOL; idx += NT)
          {
            const int r = idx / NCOL, k = idx % NCOL;
            double v = 0.0;
            if (k < NV)
              v = (hrow[r * NCOL + NV + k] + tau * hrow[r * NCOL + k]) * im;
            else if (k < NDX)
              v = tau * hrow[r * NCOL + k] * im;
            sw.JT[idx] = v;
            if (k < NDX)
              b.CN[(size_t)inst * (3 * NDX + 3) + r * NDX + k] = v;
          }
          if (lane < 3)
          {
            const double c = sc.com[lane] + tau * sc.hg[lane] * im - b.dcm_ref[(size_t)inst * 3 + lane];
            const double v = b.vN[(size_t)inst * 3 + lane];
            const double vp = b.vN_e[(size_t)inst * 3 + lane] + c / h.mu;
            tv[lane] = v;
            tv[3 + lane] = vp;
            tv[6 + lane] = c;
            b.CN[(size_t)inst * (3 * NDX + 3) + 3 * NDX + lane] = h.mu * (vp - v);
          }
        }
        SMPC_LANES_END_WAVE
      }
      full_hessian_mfma<D>(sc, sd, sw, mg, true, preg, QN, (double *)nullptr, (double *)nullptr, true, tcs);
      SMPC_LANES(NT)
      {
        double dual = 0.0;
        for (int k = lane; k < NDX; k += NT)
        {
          double qn = sd.gx_()[k] - b.lams[(ib + sprev) * NDX + k];
          if (tcs)
          {
            for (int r = 0; r < 3; r++)
              qn += sw.JT[r * NCOL + k] * tv[r];
            dual = fmax(dual, fabs(qn)); // dual residual with the current multipliers; the Newton right-hand side uses v+
            for (int r = 0; r < 3; r++)
              qn += sw.JT[r * NCOL + k] * (h.mu * (tv[3 + r] - tv[r])) / h.mu;
          }
          else
            dual = fmax(dual, fabs(qn));
          qN[k] = qn;
        }
        sc.part[lane] = dual;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane == 0)
      {
        double dual = 0.0;
        for (int k = 0; k < NT; k++)
          dual = fmax(dual, sc.part[k]);
        double pen = 0.0, prim = 0.0;
        if (tcs)
          for (int r = 0; r < 3; r++)
          {
            const double vp = tv[3 + r], dv = vp - tv[r];
            pen += 0.5 * h.mu * (vp * vp + dv * dv);
            prim = fmax(prim, fabs(tv[6 + r]));
          }
        parts[0] = sc.red[0] + pen;
        parts[1] = sc.red[0];
        parts[2] = prim;
        parts[3] = dual;
      }
      SMPC_LANES_END_WAVE
      return;
    }
    double * lq = b.lq + ((size_t)inst * H + t) * D::LQ_STRIDE;
    const double dt = h.dt, mu = h.mu;
    ftick(fp, 14);
    // ---- [A | B] (lane = column), Lagrangian gradients q, r ----
    SMPC_LANES(NT)
    {
      double dual = 0.0;
      // (the previous stage's dynamics multiplier -- device memory -- is asked for before the column is assembled, not at the end of it)
      static_assert(NDX <= NT, "the state columns are one pass of the lanes");
      const double lam_prev = b.lams[(ib + sprev) * NDX + (lane < NDX ? lane : 0)];
      for (int col = lane; col < NXU; col += NT)
      {
        const bool isA = col < NDX;
        const int k = isA ? col : col - NDX;
        const int cc = isA ? k : 2 * NV + k; // column of [da_dq | da_dv | da_dtau]
        double * dst = lq + (isA ? D::O_A : D::O_B);
        const int ld = isA ? NDX : NU;
        double acc = 0.0; // (A^T lam_next)[k] or (B^T lam_next)[k], rows in order
        double r1c[NV]; // column cc of [da_dq | da_dv | da_dtau], read before anything of this column is stored
#pragma unroll
        for (int i = 0; i < NV; i++)
          r1c[i] = sw.r1(i, cc);
        double Dtop[6];
        for (int m = 0; m < 6; m++)
          Dtop[m] = dt * dt * r1c[m] + ((isA && k == NV + m) ? dt : 0.0);
        for (int i = 0; i < 6; i++)
        {
          double v;
          if (i < 3)
            v = sd.Je3[i * 3 + 0] * Dtop[0] + sd.Je3[i * 3 + 1] * Dtop[1] + sd.Je3[i * 3 + 2] * Dtop[2] + sd.JeQ[i * 3 + 0] * Dtop[3]
                + sd.JeQ[i * 3 + 1] * Dtop[4] + sd.JeQ[i * 3 + 2] * Dtop[5];
          else
            v = sd.Je3[(i - 3) * 3 + 0] * Dtop[3] + sd.Je3[(i - 3) * 3 + 1] * Dtop[4] + sd.Je3[(i - 3) * 3 + 2] * Dtop[5];
          if (isA && k < 6)
            v += sd.Jq[i * 6 + k];
          dst[i * ld + k] = v;
          acc += v * sc.lam_next[i];
        }
        for (int i = 6; i < NV; i++)
        {
          const double v = dt * dt * r1c[i] + ((isA && k == NV + i) ? dt : 0.0) + ((isA && k == i) ? 1.0 : 0.0);
          dst[i * ld + k] = v;
          acc += v * sc.lam_next[i];
        }
        for (int i = 0; i < NV; i++)
        {
          const double v = dt * r1c[i] + ((isA && k == NV + i) ? 1.0 : 0.0);
          dst[(NV + i) * ld + k] = v;
          acc += v * sc.lam_next[NV + i];
        }
        if (isA)
        {
          // C_x^T nu: the joint-box rows are unit selectors (unmasked Jacobian)
          double cn = 0.0;
          if (h.kinematics_limits && k >= 6 && k < NV)
            cn = sc.nu[NU + k - 6];
          double q = sd.gx_()[k] + acc + cn + sd.cq_()[k] - (t > 0 ? lam_prev : 0.0);
          if (t == 0)
            q = 0.0; // x_0 is pinned (force_initial_condition_, reference src/mpc.cpp:53)
          double qf = q;
          if constexpr (D::KINO) // folded frame-velocity rows: the Newton right-hand side gets Cv^T d / mu = Cv^T (nu+ - nu)
            for (int r = 0; r < (t == 0 ? 0 : D::NVEL); r++)
              qf += sw.Cv[r * NDX + k] * (sc.vplus[NU + NA + D::NCD + r] - sc.nu[NU + NA + D::NCD + r]);
          lq[D::O_q + k] = qf;
          lq[D::O_lx + k] = sd.gx_()[k];
          lq[D::O_f + k] = mu * (sc.lamp[k] - sc.lam_next[k]);
          lq[D::O_lpd + k] = 2.0 * sc.lamp[k] - sc.lam_next[k];
          dual = fmax(dual, fabs(q));
        }
        else
        {
          const double cn = h.torque_limits ? sc.nu[k] : 0.0;
          const double r = sd.gu_()[k] + acc + cn + sd.cq_()[NDX + k];
          lq[D::O_r + k] = r;
          lq[D::O_lu + k] = sd.gu_()[k];
          dual = fmax(dual, fabs(r));
        }
      }
      sc.part[lane] = dual;
    }
    SMPC_LANES_END_WAVE
    ftick(fp, 15);
    // ---- [Q S; S^T R] = H_0 + JT^T (W~ JT) + preg I on the matrix cores ----
    full_hessian_mfma<D>(sc, sd, sw, mg, false, preg, lq + D::O_Q, lq + D::O_S, lq + D::O_R, false, false, &fp);
    SMPC_LANES(NT)
    {
      for (int i = lane; i < NC; i += NT)
      {
        lq[D::O_d + i] = mu * (sc.vplus[i] - sc.nu[i]);
        lq[D::O_vpd + i] = sc.act[i] ? 2.0 * sc.vplus[i] - sc.nu[i] : 0.0;
        lq[D::O_act + i] = sc.act[i] ? 1.0 : 0.0;
      }
    }
    SMPC_LANES_END_WAVE
    ftick(fp, 16);
    if constexpr (D::NCONE > 0)
    {
      // dense cone rows of the knot: A_cone d lam / d(x, u) for the active rows, zero otherwise
      // (a stage without an active row -- the common case: soles flat on the ground -- would write 34 x 78 zeros over zeros; the
      //  block's flag says whether it holds anything else, and the phase is skipped when neither it nor this stage does.  The block's content
      //  is what the unconditional form leaves; the sweep's light grid never reads it in such a stage.)
      bool cone_rows = true;
      {
        unsigned any = 0u;
        for (int i = 0; i < D::NCONE; i++) // (wave-uniform: every lane reads the same flags)
          any |= sc.act[NU + NA + i] != 0 ? 1u : 0u;
        any = SMPC_UNIFORM_U32(any);
        const bool was = cone_block_dirty;
        cone_rows = any != 0u || was;
        if ((any != 0u) != was)
        {
          SMPC_LANES(NT)
          if (lane == 0)
            lq[D::O_cdirty] = any != 0u ? 1.0 : 0.0;
          SMPC_LANES_END_WAVE
        }
      }
      if (cone_rows)
      {
      if constexpr (D::KINO)
      {
        // CentroidalWrenchConeResidual: constant rows on the wrench of foot f (lane = column: a row's lanes store consecutive addresses, the
        // row index is a compile-time constant of the unrolled loops)
        SMPC_LANES(NT)
        for (int k = lane; k < NXU; k += NT)
        {
#pragma unroll
          for (int f = 0; f < NF; f++)
          {
            const int j = k - NDX - D::FS * f;
            const bool mine = j >= 0 && j < D::FS;
#pragma unroll
            for (int r = 0; r < D::NCONE1; r++)
            {
              const int i = D::NCONE1 * f + r;
              const double e = wrench_cone_entry(r, mine ? j : 0, h.fric_mu, h.Lfoot, h.Wfoot);
              const double acc = (mine && sc.act[NU + NA + i]) ? e : 0.0;
              if (k < NDX)
                lq[D::O_C + i * NDX + k] = acc;
              else
                lq[D::O_D + i * NU + k - NDX] = acc;
            }
          }
        }
        SMPC_LANES_END_WAVE
      }
      else
      {
        // lane = column k of (x, u): its NCM entries of d lam / d(x, u) are read once (the rows may lie in device memory), then every cone row of
        // every foot is a short sum with wave-uniform coefficients; a row's lanes store consecutive addresses
        SMPC_LANES(NT)
        for (int k = lane; k < NXU; k += NT)
        {
          double jt[NCM];
#pragma unroll
          for (int r = 0; r < NCM; r++)
            jt[r] = sw.JT[r * NCOL + k];
#pragma unroll
          for (int f = 0; f < NF; f++)
          {
            const bool on = (mask >> f) & 1u;
            const int c = on ? __builtin_popcount(mask & ((1u << f) - 1u)) : 0;
            double lf[D::FS];
#pragma unroll
            for (int j = 0; j < D::FS; j++)
            {
              // (the contact index c is wave-uniform; a select chain over the NF possible values keeps jt in registers)
              double v = jt[j];
#pragma unroll
              for (int cc = 1; cc < NF; cc++)
                v = c == cc ? jt[D::FS * cc + j] : v;
              lf[j] = v;
            }
#pragma unroll
            for (int r = 0; r < D::NCONE1; r++)
            {
              const int i = D::NCONE1 * f + r;
              double acc = 0.0;
#pragma unroll
              for (int j = 0; j < D::FS; j++)
                acc += wrench_cone_entry(r, j, h.fric_mu, h.Lfoot, h.Wfoot) * lf[j];
              acc = sc.act[NU + NA + i] ? acc : 0.0;
              if (k < NDX)
                lq[D::O_C + i * NDX + k] = acc;
              else
                lq[D::O_D + i * NU + k - NDX] = acc;
            }
          }
        }
        SMPC_LANES_END_WAVE
      }
      }
    }
    if constexpr (D::KINO)
    {
      // frame-velocity rows (folded into Q, q above): kept for the multiplier step of the forward sweep, dnu = (Cv dx + d) / mu
      SMPC_LANES(NT)
      for (int idx = lane; idx < D::NVEL * NDX; idx += NT)
        lq[D::O_V + idx] = sw.Cv[idx];
      SMPC_LANES_END_WAVE
    }
    if constexpr (D::NLAND > 0)
    {
      // land rows of the knot (equality rows: always active where present), behind the cone rows; no control columns
      SMPC_LANES(NT)
      for (int idx = lane; idx < D::NLAND * NXU; idx += NT)
      {
        const int i = idx / NXU, k = idx % NXU;
        const int f = i / D::NLAND1, r = i % D::NLAND1;
        const double v = (k < NDX && ((land >> f) & 1u)) ? full_land_entry<D>(sc, sd, f, r, k) : 0.0;
        if (k < NDX)
          lq[D::O_C + (D::NCONE + i) * NDX + k] = v;
        else
          lq[D::O_D + (D::NCONE + i) * NU + k - NDX] = v;
      }
      SMPC_LANES_END_WAVE
    }
    ftick(fp, 30);
    // (the maximum over the lanes' partial values in two levels: eight lanes over eight values each, then one lane over eight -- max is exact in any order)
    SMPC_LANES(NT)
    if (lane < 8)
    {
      double d8[8];
#pragma unroll
      for (int i = 0; i < 8; i++)
        d8[i] = sc.part[lane * 8 + i];
      double dual = 0.0;
#pragma unroll
      for (int i = 0; i < 8; i++)
        dual = fmax(dual, d8[i]);
      sc.part8[lane] = dual;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      if (lane == 0)
      {
        double dual = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++)
          dual = fmax(dual, sc.part8[k]);
        parts[0] = sc.red[0] + sc.red[1];
        parts[1] = sc.red[0];
        parts[2] = sc.red[2];
        parts[3] = dual;
      }
    }
    SMPC_LANES_END_WAVE
    ftick(fp, 29);
  }

  template <class D>
  SMPC_DEV void fdyn_deriv_body(const StageKernelArgs<D> & ka, int block)
  {
    // A block whose wide blocks lie in device memory owns slice `block` for its lifetime: the grid is as many blocks as stay resident
    // (ka.nres), each walks the work items block, block + nres, .. -- the slices are 50 MB that live in the caches instead of one slice per
    // item of the launch (5 GB at B = 1024, H = 100, every line of it written back once), and no slice is ever touched from two CUs.
    const int H = ka.b.H;
    if (ka.nwork > 0 && ka.slots > 0)
    {
      // list mode on the persistent grid: the (list entry, stage) pairs themselves are dealt round-robin (the slot walk below gives a block the
      // entries slot, slot + slots, ..: one or two per item, uneven once the list is longer than the slots)
      const int total = ka.b.und_list[ka.b.B] * (H + 1);
      for (int w = block; w < total; w += ka.nres)
        fdyn_deriv_one<D>(ka, ka.b.und_list[w / (H + 1)], w % (H + 1), block);
      return;
    }
    const int nwork = ka.nwork > 0 ? ka.nwork : block + 1, nres = ka.nwork > 0 ? ka.nres : 1;
    for (int w = block; w < nwork; w += nres)
    {
      const int slot = w / (H + 1), t = w % (H + 1);
      const int count = ka.slots > 0 ? ka.b.und_list[ka.b.B] : slot + 1;
      const int stride = ka.slots > 0 ? ka.slots : ka.b.B;
      for (int m = slot; m < count; m += stride)
        fdyn_deriv_one<D>(ka, ka.slots > 0 ? ka.b.und_list[m] : m, t, block);
    }
  }

  // =============================================================================================
  // fdyn_trial_body: one line-search candidate (LINEAR rollout point, reference src/mpc.cpp:44): re-evaluation, AL
  // multipliers, merit partials.  Same launch geometry as trial_body (smpc_kino_kernels.h).
  // =============================================================================================
  template <class D>
  SMPC_DEV void fdyn_trial_one(const StageKernelArgs<D> & ka, int inst, int t, int j)
  {
    typedef FullScratch<D, false> SC;
    constexpr int NT = 64;
    constexpr int NV = D::NV, NX = D::NX, NDX = D::NDX, NU = D::NU, NC = D::NC, NF = D::NF, NCM = D::NCM, NQ = D::NQ;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const bool term = t == H;
    const DevModel<D> & mg = *b.model;
    SMPC_LDS(SC, scs, 1);
    SC & sc = scs[0];
    const int st = ring_slot(ka.head, t, R), sn = ring_slot(ka.head, term ? t : t + 1, R);
    const size_t ib = (size_t)inst * R;
    double alpha = 1.0;
    for (int i = 0; i < j; i++)
      alpha *= 0.5;
    const unsigned mask = term ? 0u : (b.stages[t].mask & ((1u << NF) - 1u));
    const unsigned land = (D::NLAND > 0 && !term && mg.land_cstr) ? (b.stages[t].land & mask) : 0u; // feet with land_cstr rows at this stage
    const double * dx = b.dxs + ((size_t)inst * (H + 1) + t) * NDX;
    const size_t lt = (size_t)inst * H + (term ? 0 : t);
    // Every global load of the block's inputs before the first commit to LDS (see fdyn_deriv_one); the two states and their steps go to a
    // staging area in the late block (unused until the dynamics phases are through), the trial points are formed from there.
    double * const sx = sc.swp_(), * const sdx = sx + NX, * const sxn = sdx + NDX, * const sdxn = sxn + NX;
    static_assert(2 * (NX + NDX) <= SC::SWP_DOUBLES, "staging of the trial points fits the sweep scratch");
    SMPC_LANES(NT)
    {
      constexpr int PX = (NX + NT - 1) / NT, PU = (NU + NT - 1) / NT, PD = (NDX + NT - 1) / NT, PC = (NC + NT - 1) / NT;
      static_assert(NCM <= NT && NF * 3 <= NT, "one entry per lane");
      const int ts = term ? 0 : t;
      double vx[PX], vxn[PX], vxt[PX], vu[PU], vdu[PU], vur[PU], vl[PD], vdl[PD], vdx[PD], vdxn[PD], vn[PC], vdn[PC];
#pragma unroll
      for (int n = 0; n < PX; n++)
      {
        const int i = lane + n * NT < NX ? lane + n * NT : NX - 1;
        vx[n] = b.xs[(ib + st) * NX + i];
        vxn[n] = b.xs[(ib + sn) * NX + i];
        const double * pt = term ? &mg.x_term[i] : ((i >= NQ && i < NQ + 6) ? &b.vref[(ib + st) * 6 + (i - NQ)] : &b.stages[ts].x_tgt[i]);
        vxt[n] = *pt;
      }
#pragma unroll
      for (int n = 0; n < PU; n++)
      {
        const int i = lane + n * NT < NU ? lane + n * NT : NU - 1;
        vu[n] = b.us[(ib + st) * NU + i];
        vdu[n] = b.dus[lt * NU + i];
        vur[n] = b.stages[ts].u_ref[i];
      }
      const double vfr = b.stages[ts].f_ref[lane < NCM ? lane : NCM - 1];
      const double vfo = b.foot_ref[((size_t)inst * H + ts) * NF * 3 + (lane < NF * 3 ? lane : NF * 3 - 1)];
#pragma unroll
      for (int n = 0; n < PD; n++)
      {
        const int i = lane + n * NT < NDX ? lane + n * NT : NDX - 1;
        vl[n] = b.lams[(ib + st) * NDX + i];
        vdl[n] = b.dlams[lt * NDX + i];
        vdx[n] = dx[i];
        vdxn[n] = dx[(term ? 0 : NDX) + i];
      }
#pragma unroll
      for (int n = 0; n < PC; n++)
      {
        const int i = lane + n * NT < NC ? lane + n * NT : NC - 1;
        vn[n] = b.vs[(ib + st) * NC + i];
        vdn[n] = b.dvs[lt * NC + i];
      }
      SMPC_SCHED_FENCE();
      full_load_head<D, NT>(sc.h, &mg, lane);
#pragma unroll
      for (int n = 0; n < PX; n++)
        if (lane + n * NT < NX)
        {
          sx[lane + n * NT] = vx[n];
          sxn[lane + n * NT] = vxn[n];
          sc.x_tgt[lane + n * NT] = vxt[n];
        }
#pragma unroll
      for (int n = 0; n < PU; n++)
        if (lane + n * NT < NU)
        {
          sc.u[lane + n * NT] = term ? 0.0 : vu[n] + alpha * vdu[n];
          sc.u_ref[lane + n * NT] = term ? 0.0 : vur[n];
        }
      if (lane < NCM)
        sc.f_ref[lane] = term ? 0.0 : vfr;
      if (lane < NF * 3)
        sc.foot_ref[lane] = term ? 0.0 : vfo;
#pragma unroll
      for (int n = 0; n < PD; n++)
        if (lane + n * NT < NDX)
        {
          sc.lam_next[lane + n * NT] = term ? 0.0 : vl[n] + alpha * vdl[n];
          sdx[lane + n * NT] = vdx[n];
          sdxn[lane + n * NT] = vdxn[n];
        }
#pragma unroll
      for (int n = 0; n < PC; n++)
        if (lane + n * NT < NC)
          sc.nu[lane + n * NT] = term ? 0.0 : vn[n] + alpha * vdn[n];
    }
    SMPC_LANES_END_WAVE
    // trial points x_t (+) alpha dx_t and x_{t+1} (+) alpha dx_{t+1}
    SMPC_LANES(NT)
    {
      lanes_integrate<D>(sx, sdx, alpha, sc.x, lane, 0);
      lanes_integrate<D>(sxn, sdxn, alpha, sc.xn1, lane, 1);
    }
    SMPC_LANES_END_WAVE
    FullProf fp;
    full_dynamics_phases<D, false>(sc, (FullScratchDeriv<D> *)nullptr, mg, mask, !term, fp);
    full_eval_tail<D, false>(sc, (FullScratchDeriv<D> *)nullptr, mg, mask, land, term, b.lams_e + (ib + st) * NDX, b.vs_e + (ib + st) * NC);
    double * parts = b.partsT + (((size_t)inst * D::LS_N + j) * (H + 1) + t) * 2;
    SMPC_LANES(NT)
    {
      if (lane == 0)
      {
        double pen = 0.0, prim = 0.0;
        if (term && b.CN != nullptr)
          for (int r = 0; r < 3; r++)
          { // terminal constraint at the trial point, multipliers v + alpha dv
            const double c = sc.com[r] + b.dcm_tau * sc.hg[r] / sc.h.total_mass - b.dcm_ref[(size_t)inst * 3 + r];
            const double vp = b.vN_e[(size_t)inst * 3 + r] + c / sc.h.mu;
            const double dv = vp - (b.vN[(size_t)inst * 3 + r] + alpha * b.dvN[(size_t)inst * 3 + r]);
            pen += 0.5 * sc.h.mu * (vp * vp + dv * dv);
            prim = fmax(prim, fabs(c));
          }
        parts[0] = term ? sc.red[0] + pen : sc.red[0] + sc.red[1];
        parts[1] = term ? prim : sc.red[2];
      }
      if (!term && t < 2)
        for (int i = lane; i < NV; i += NT)
        {
          double * xd = b.xdotT + (((size_t)inst * D::LS_N + j) * 2 + t) * 2 * NV;
          xd[i] = sc.x[NQ + i];
          xd[NV + i] = sc.a[i];
        }
      // contact forces at the trial point: the accepted candidate's are the stage's forces (MPC::getContactForces)
      if (!term && b.forcesT != nullptr)
        for (int i = lane; i < NCM; i += NT)
        {
          const int f = i / D::FS;
          const bool on = (mask >> f) & 1u;
          const int c = __builtin_popcount(mask & ((1u << f) - 1u));
          b.forcesT[(((size_t)inst * D::LS_N + j) * H + t) * NCM + i] = on ? sc.lam[c * D::FS + i % D::FS] : 0.0;
        }
    }
    SMPC_LANES_END_WAVE
  }

  template <class D>
  SMPC_DEV void fdyn_trial_body(const StageKernelArgs<D> & ka, int block)
  {
    const int H = ka.b.H;
    const int slot = block / (H + 1), t = block % (H + 1);
    const int count = ka.slots > 0 ? ka.b.und_list[ka.b.B] : slot + 1;
    const int stride = ka.slots > 0 ? ka.slots : ka.b.B;
    for (int m = slot; m < count; m += stride)
    {
      const int inst = ka.slots > 0 ? ka.b.und_list[m] : m;
      if (ka.b.ls_sel[inst] >= 0) // decided (list mode: by an earlier batch of backtracking candidates -- the list is the one compacted before the first)
        continue;
      for (int jj = 0; jj < ka.nj; jj++)
        fdyn_trial_one<D>(ka, inst, t, ka.j0 + jj);
    }
  }

  // ---------------------------------------------------------------------------------------------------------------
  // state front end of a robot with any tree (RobotDataHandler::updateInternalData + getCentroidalState, reference
  // src/robot-handler.cpp:106-149) on the kinematics phases of the dense stage kernel; DF = a FullDims of the robot
  // ---------------------------------------------------------------------------------------------------------------
  template <class DF>
  SMPC_DEV void frontend_full_body(const FrontendArgs<DF> & ka, int block)
  {
    typedef FullScratch<DF, false> SC;
    constexpr int NT = 64, NX = DF::NX, NF = DF::NF;
    const int inst = block;
    const DevModel<DF> & mg = *ka.b.model;
    SMPC_LDS(SC, scs, 1);
    SC & sc = scs[0];
    SMPC_LANES(NT)
    {
      full_load_head<DF, NT>(sc.h, &mg, lane);
      for (int i = lane; i < NX; i += NT)
        sc.x[i] = ka.X[(size_t)inst * NX + i];
    }
    SMPC_LANES_END_WAVE
    FullProf fp;
    full_dynamics_phases<DF, false>(sc, (FullScratchDeriv<DF> *)nullptr, mg, 0u, false, fp);
    SMPC_LANES(NT)
    {
      if (ka.feet != nullptr && lane < NF * 3)
        ka.feet[(size_t)inst * NF * 3 + lane] = sc.footp[lane];
      if (ka.com != nullptr && lane < 3)
        ka.com[(size_t)inst * 3 + lane] = sc.com[lane];
      if (ka.hg != nullptr && lane < 6)
        ka.hg[(size_t)inst * 6 + lane] = sc.hg[lane];
      if (ka.cstate != nullptr && lane < 9)
        ka.cstate[(size_t)inst * 9 + lane] = lane < 3 ? sc.com[lane] : sc.hg[lane - 3];
    }
    SMPC_LANES_END_WAVE
  }

  // ---------------------------------------------------------------------------------------------------------------
  // The constrained forward dynamics alone, for any robot of the full-dynamics engine (MultibodyConstraintFwdDynamics::forward ->
  // pinocchio::constraintDynamics with the contacts of FullDynamicsOCP, reference src/fulldynamics.cpp:39,50-75,139): point feet (3-D
  // LOCAL) and flat feet (6-D LOCAL_WORLD_ALIGNED).  One wavefront per state; the phases are those of the stage kernel.
  // ---------------------------------------------------------------------------------------------------------------
  template <class D>
  struct FdynFdArgs
  {
    Buffers<D> b;          // only b.model is read
    const double * X;      // [n][NX] states (device)
    const double * tau;    // [n][NV - 6] joint torques (device)
    const unsigned * mask; // [n] contact bit per foot (device)
    double Kp[6], Kd[6];   // Baumgarte corrector (the first FS entries)
    double prox_accuracy, prox_mu;
    int prox_max_iter;
    double * a_out;   // [n][NV]
    double * lam_out; // [n][FS NF]: contact forces / wrenches ON the robot, feet in contact first (in order), rest 0
    int * iters_out;  // [n] proximal iterations taken (may be null)
  };
  template <class D>
  SMPC_DEV void fdyn_fd_body(const FdynFdArgs<D> & ka, int block)
  {
    typedef FullScratch<D, false> SC;
    constexpr int NT = 64, NX = D::NX, NV = D::NV, NU = D::NU, NCM = D::NCM, FS = D::FS;
    static_assert(!D::KINO, "the forward dynamics of the full model");
    const int inst = block;
    const DevModel<D> & mg = *ka.b.model;
    const unsigned mask = ka.mask[inst];
    SMPC_LDS(SC, scs, 1);
    SC & sc = scs[0];
    SMPC_LANES(NT)
    {
      full_load_head<D, NT>(sc.h, &mg, lane);
      for (int i = lane; i < NX; i += NT)
        sc.x[i] = ka.X[(size_t)inst * NX + i];
      for (int i = lane; i < NU; i += NT)
        sc.u[i] = ka.tau[(size_t)inst * NU + i];
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      if (lane < FS)
      {
        sc.h.Kp[lane] = ka.Kp[lane];
        sc.h.Kd[lane] = ka.Kd[lane];
      }
      if (lane == 0)
      {
        sc.h.prox_accuracy = ka.prox_accuracy;
        sc.h.prox_mu = ka.prox_mu;
        sc.h.prox_max_iter = ka.prox_max_iter;
      }
    }
    SMPC_LANES_END_WAVE
    FullProf fp;
    full_dynamics_phases<D, false>(sc, (FullScratchDeriv<D> *)nullptr, mg, mask, true, fp);
    SMPC_LANES(NT)
    {
      const int nact = FS * __builtin_popcount(mask);
      for (int i = lane; i < NV; i += NT)
        ka.a_out[(size_t)inst * NV + i] = sc.a[i];
      for (int i = lane; i < NCM; i += NT)
        ka.lam_out[(size_t)inst * NCM + i] = i < nact ? sc.lam[i] : 0.0;
      if (ka.iters_out != nullptr && lane == 0)
        ka.iters_out[inst] = sc.iters_[0];
    }
    SMPC_LANES_END_WAVE
  }
} // namespace smpc
