// smpc_riccati_kino.h -- structure-exploiting proximal Riccati sweep for the kinodynamics stage
// (HOT(4)/(5) of SolverProxDDP::run, reference src/mpc.cpp:212; LQ solver choice src/mpc.cpp:52).
//
// One 64-lane wavefront per instance, 8 per CU (19.8 KB LDS, 256 registers), no workgroup-wide barriers: phases are
// separated by wave-level ordering points only.  Same KKT system as riccati_body (smpc_solver_kernels.h), reorganised:
//
//   * semi-implicit Euler structure (reference src/kinodynamics.cpp:88): with the tangent split
//     [qb(6) | qj | vb(6) | vj], only the 12 rows G = qb u vb of A and B are dense; rows qj are
//     e_i + dt e_{v(i)} (A) / dt^2 e_a (B), rows vj are e_i / dt e_a.  Products with A, B cost 12/36 of dense.
//   * both eliminations are Schur complements of bordered symmetric matrices, computed by symmetric block sweeps on
//     the FP64 matrix cores (wave_block_sweep):
//         [ I + mu P     sqrt(mu) P   sqrt(mu) pt0 ]                       [ Q^ + C^T C/mu   S^   q^ + C^T d/mu ]
//         [ sqrt(mu) P   P            pt0          ]  -> P~, p~   and      [ S^^T            R^   r^            ]  -> P_t, p_t, K, k
//     (pt0 = p + P f; P~ = (I + mu P)^-1 P, p~ = (I - mu P~) pt0; K = -R^^-1 S^^T, k = -R^^-1 r^)
//   * constraint rows: joint-box rows are unit selectors (diagonal contribution), contact rows are dense.
//   * the GEMM-shaped products (P~[G,G] NAB, [A|B]^T P~ [A|B], C^T C) run on the matrix cores as well; operands are
//     read once from LDS, and blocks of <= 12 rows (TG, PEG) never leave the accumulator registers because the
//     accumulator layout of v_mfma_f64_16x16x4_f64 is the operand layout of the next product.
#pragma once
#include "smpc_solver_kernels.h"
#ifndef SMPC_KINO_RCP1
#define SMPC_KINO_RCP1 false // (experiment switch: one Newton step on the pivot reciprocals of the kinodynamics sweep)
#endif

namespace smpc
{
  // gains block v2 per (instance, stage)
  template <class D>
  struct GainsK
  {
    static constexpr int NDX = D::NDX, NU = D::NU;
    static constexpr int G_W = 0;                     // [K | k]^T = -[S^^T | r^]^T R^^-1, (NDX+1) x NU: K(u, x) at G_W + x NU + u, k(u) at G_W + NDX NU + u
                                                      // (the sweep stores its accumulator rows -- x -- as runs of u; the forward sweep reads columns)
    static constexpr int G_Pt = G_W + NU * (NDX + 1);       // P~, upper triangle packed row by row (the forward sweep is
    static constexpr int G_pn = G_Pt + NDX * (NDX + 1) / 2; //     HBM bound: half of P~ is 18 % of what it reads); p_{t+1}
    static constexpr int STRIDE = ((G_pn + NDX + 7) / 8) * 8;
    SMPC_HD static constexpr int pt_off(int i, int j) { return G_Pt + i * NDX - i * (i - 1) / 2 + (j - i); } // i <= j
  };

  template <class D>
  struct KinoIdx
  {
    static constexpr int NV = D::NV, NDX = D::NDX, NA = D::NA, NF = D::NF, NU = D::NU, NG = 12;
    SMPC_HD static int G(int g) { return g < 6 ? g : NV + g - 6; }             // dense row index
    SMPC_HD static bool isG(int i) { return i < 6 || (i >= NV && i < NV + 6); }
    SMPC_HD static bool isQj(int i) { return i >= 6 && i < NV; }
    SMPC_HD static bool isVj(int i) { return i >= NV + 6; }
  };

  // index of the upper tile (I <= J) of an NTI x NTI tile grid, row-major over the upper triangle
  template <int NTI>
  SMPC_HD constexpr int tix(int I, int J)
  {
    return I * NTI - I * (I - 1) / 2 + (J - I);
  }

  // Symmetric block sweep (block Gauss-Jordan without pivoting on an SPD pivot block) on the matrix cores, one wave.
  // T is a symmetric (16 NTI)^2 matrix held as upper 16x16 accumulator tiles (SMPC_ACC, tile tix(I, J)).  The NP
  // panels of 4 pivots  PIV0 + 4p .. PIV0 + 4p + 3  are swept one after the other:
  //     P_m := the stored entries pairing index m with the 4 pivots,   D := pivot block,   U_m := D^-1 P_m,
  //     T[i][j] -= U_i^T P_j   for every stored (i, j)          (one 16x16x4 MFMA per tile: a rank-4 update)
  //     entries pairing (pivot, m) := U_m                       (ALL only)
  // Every update is the symmetric bilinear form P_i^T D^-1 P_j, so upper storage needs no sign bookkeeping: after
  // sweeping a pivot set S the stored entry pairing s in S with an unswept j holds (T_SS^-1 T_Sj)_s, and the unswept
  // block holds the Schur complement T_jj' - T_jS T_SS^-1 T_Sj'.  ALL = false maintains only the tiles at or beyond
  // the current pivot's tile row (Schur complement of leading pivots: what a Cholesky-based elimination would give).
  // prow / urow: LDS, 4 x 16 NTI doubles each.  prof: optional phase timers (slots 36..39).
  // scheduling class of tile (I, J) in panel p of a sweep: 0 = not maintained any more, 1 = must be updated before the
  // next panel's pivot entries are gathered (or is rewritten by this panel's fix-up), 2 = can be updated later
  template <int NTI, bool ALL, int PIV0, int NP>
  SMPC_HD constexpr int sweep_tile_class(int p, int I, int J)
  {
    const int kb = PIV0 + 4 * p, Ip = kb / 16;
    const bool last = p == NP - 1;
    const int Ipn = last ? -1 : (kb + 4) / 16;
    if (!ALL)
    {
      if (I < Ip)
        return 0;
      // a tile row made of pivots only is dead once its last panel has been swept
      if (I == Ip && Ipn != Ip && PIV0 + 4 * NP >= 16 * (Ip + 1))
        return 0;
      return (last || I == Ipn) ? 1 : 2;
    }
    if (last)
      return 1;
    const bool fixup = (I == Ip) || (J == Ip);             // pivot rows / columns rewritten by (d)
    const bool gather = (I == Ipn) || (J == Ipn && I < Ipn); // read by the next (a)
    return (fixup || gather) ? 1 : 2;
  }

  // SKIPPABLE: bit p of `skip` set = panel p is left out (its 4 pivots couple to nothing: the caller deals with their rows).
  // RCP1: one Newton step on the pivot reciprocals instead of two (2e-15 instead of 1e-16 relative, tools/micro/rcp_accuracy.hip): two dependent
  // FMAs less per pivot, 8 per panel
  // Per-lane address bases of a sweep's LDS staging, formed ONCE by a kernel that sweeps in a loop (the lane index is re-materialised in every
  // lane phase, so the phases cannot share them otherwise: five integer instructions per base and phase, ~ 270 per stage of the kinodynamics
  // sweep): rc = (lane >> 4) * LDW + (lane & 15), cr = (lane & 15) * LDW + (lane >> 4).
  struct SweepBases
  {
    SMPC_PL(int, rc, 64);
    SMPC_PL(int, cr, 64);
  };
  template <int LDW>
  SMPC_DEV void sweep_bases_init(SweepBases & sb)
  {
    SMPC_LANES(64)
    {
      SMPC_PLV(sb.rc) = (lane >> 4) * LDW + (lane & 15);
      SMPC_PLV(sb.cr) = (lane & 15) * LDW + (lane >> 4);
    }
    SMPC_LANES_END_WAVE
  }
  template <int NT, int NTI, bool ALL, int PIV0, int NP, bool SKIPPABLE = false, bool RCP1 = false, class Acc>
  SMPC_DEV void wave_block_sweep(Acc & acc, double * prow, double * urow, double * prof, long long & tprev, unsigned skip = 0u, const SweepBases * sb = nullptr)
  {
    constexpr int LDW = 16 * NTI;
    static_assert(NT == 64 && PIV0 % 4 == 0 && PIV0 + 4 * NP <= LDW && LDW <= 2 * NT, "sweep geometry");
    // operands of two consecutive panels: the rank-4 updates that the next panel does not depend on are issued
    // between the next panel's gather / pivot-inverse phases, so the matrix pipe works while the VALU / LDS chain runs
    SMPC_PLA(double, aop, NT, 2 * NTI);
    SMPC_PLA(double, bop, NT, 2 * NTI);
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
      const int kb = PIV0 + 4 * p, Ip = kb / 16, c0 = kb % 16, vp = c0 / 4;
      const int ob = (p & 1) * NTI, obp = ((p + 1) & 1) * NTI; // operand sets of this / the previous panel
      const bool skp = SKIPPABLE && ((skip >> p) & 1u);                     // (wave-uniform)
      const bool prev = p > 0 && !(SKIPPABLE && ((skip >> (p > 0 ? p - 1 : 0)) & 1u)); // the previous panel ran: it deferred updates
      // (a) pivot entries -> prow[k][m]
      if (!skp)
      {
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
        const int brc = sb ? SMPC_PLV(sb->rc) : lr * LDW + lc, bcr = sb ? SMPC_PLV(sb->cr) : lc * LDW + lr;
#pragma unroll
        for (int J = Ip; J < NTI; J++)
          prow[brc + 16 * J] = SMPC_ACCV(acc, tix<NTI>(Ip, J), vp);
        if (ALL && lc >= c0 && lc < c0 + 4)
        {
#pragma unroll
          for (int I = 0; I < Ip; I++)
#pragma unroll
            for (int v = 0; v < 4; v++)
              prow[bcr - c0 * LDW + 16 * I + 4 * v] = SMPC_ACCV(acc, tix<NTI>(I, Ip), v);
        }
      }
      SMPC_LANES_END_WAVE
      }
      prof_tick(prof, 36, tprev);
      // deferred updates of the previous panel, first half
      if (prev)
      {
        int cnt = 0;
#pragma unroll
        for (int I = 0; I < NTI; I++)
#pragma unroll
          for (int J = I; J < NTI; J++)
            if (sweep_tile_class<NTI, ALL, PIV0, NP>(p - 1, I, J) == 2)
            {
              if ((cnt & 1) == 0)
                SMPC_MFMA(acc, tix<NTI>(I, J), aop, obp + I, bop, obp + J);
              cnt++;
            }
      }
      // (b) U_m = D^-1 P_m, lane = index m (every lane factors the 4x4 pivot block D = L diag(d) L^T itself; the substitution is a
      //     backward-stable solve -- two 2x2 block pivots with closed-form inverses save a fifth of this phase's instructions and were
      //     measured, but cost the ill-conditioned stage KKT blocks of the constrained problems half a digit: 1e-5 -> 4e-5 in a closed loop)
      if (!skp)
      {
      SMPC_LANES(NT)
      {
        const double * d = prow + kb;
        const double D00 = d[0], D01 = d[1], D02 = d[2], D03 = d[3];
        const double D11 = d[LDW + 1], D12 = d[LDW + 2], D13 = d[LDW + 3];
        const double D22 = d[2 * LDW + 2], D23 = d[2 * LDW + 3], D33 = d[3 * LDW + 3];
        auto rcp = [](double x) { return RCP1 ? SMPC_RCP1(x) : SMPC_RCP(x); };
        const double i0 = rcp(D00);
        const double l10 = D01 * i0, l20 = D02 * i0, l30 = D03 * i0;
        const double i1 = rcp(D11 - l10 * D01);
        const double t21 = D12 - l20 * D01, t31 = D13 - l30 * D01;
        const double l21 = t21 * i1, l31 = t31 * i1;
        const double i2 = rcp(D22 - l20 * D02 - l21 * t21);
        const double t32 = D23 - l30 * D02 - l31 * t21;
        const double l32 = t32 * i2;
        const double i3 = rcp(D33 - l30 * D03 - l31 * t31 - l32 * t32);
        // columns at or beyond the pivot's tile row only when the tiles before it are not maintained (ALL = false)
        const int M0 = ALL ? 0 : 16 * Ip; // (a constant once the panel loop is unrolled)
#pragma unroll
        for (int rr = 0; rr < (LDW - M0 + NT - 1) / NT; rr++)
        {
          const int m = M0 + lane + rr * NT;
          if (m < LDW)
          {
            const double a0 = prow[m], a1 = prow[LDW + m], a2 = prow[2 * LDW + m], a3 = prow[3 * LDW + m];
            const double y1 = a1 - l10 * a0;
            const double y2 = a2 - l20 * a0 - l21 * y1;
            const double y3 = a3 - l30 * a0 - l31 * y1 - l32 * y2;
            const double u3 = y3 * i3;
            const double u2 = y2 * i2 - l32 * u3;
            const double u1 = y1 * i1 - l21 * u2 - l31 * u3;
            const double u0 = a0 * i0 - l10 * u1 - l20 * u2 - l30 * u3;
            urow[m] = u0;
            urow[LDW + m] = u1;
            urow[2 * LDW + m] = u2;
            urow[3 * LDW + m] = u3;
          }
        }
      }
      SMPC_LANES_END_WAVE
      }
      prof_tick(prof, 37, tprev);
      // deferred updates of the previous panel, second half
      if (prev)
      {
        int cnt = 0;
#pragma unroll
        for (int I = 0; I < NTI; I++)
#pragma unroll
          for (int J = I; J < NTI; J++)
            if (sweep_tile_class<NTI, ALL, PIV0, NP>(p - 1, I, J) == 2)
            {
              if ((cnt & 1) == 1)
                SMPC_MFMA(acc, tix<NTI>(I, J), aop, obp + I, bop, obp + J);
              cnt++;
            }
      }
      // (c) operands of this panel's rank-4 updates ; the updates the next panel depends on
      if (skp)
        continue;
      SMPC_LANES(NT)
      {
        const int brc = sb ? SMPC_PLV(sb->rc) : (lane >> 4) * LDW + (lane & 15);
#pragma unroll
        for (int I = 0; I < NTI; I++)
        {
          SMPC_PLV(aop)[ob + I] = -urow[brc + 16 * I];
          SMPC_PLV(bop)[ob + I] = prow[brc + 16 * I];
        }
      }
      SMPC_LANES_END_WAVE
#pragma unroll
      for (int I = 0; I < NTI; I++)
#pragma unroll
        for (int J = I; J < NTI; J++)
          if (sweep_tile_class<NTI, ALL, PIV0, NP>(p, I, J) == 1)
            SMPC_MFMA(acc, tix<NTI>(I, J), aop, ob + I, bop, ob + J);
      prof_tick(prof, 38, tprev);
      // (d) pivot entries := U
      if (ALL)
      {
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
          const int brc = sb ? SMPC_PLV(sb->rc) : lr * LDW + lc, bcr = sb ? SMPC_PLV(sb->cr) : lc * LDW + lr;
#pragma unroll
          for (int J = Ip; J < NTI; J++)
            SMPC_ACCV(acc, tix<NTI>(Ip, J), vp) = urow[brc + 16 * J];
          if (lc >= c0 && lc < c0 + 4)
          {
#pragma unroll
            for (int I = 0; I <= Ip; I++)
#pragma unroll
              for (int v = 0; v < 4; v++)
                if (I < Ip || v != vp)
                  SMPC_ACCV(acc, tix<NTI>(I, Ip), v) = urow[bcr - c0 * LDW + 16 * I + 4 * v];
          }
        }
        SMPC_LANES_END_WAVE
        prof_tick(prof, 39, tprev);
      }
    }
    // (the last panel defers nothing)
  }

  // A (N x N, row-major, SPD) <- A^-1 in place: Schur complement of the bordered matrix [[A, I], [I, 0]] = -A^-1, by the symmetric
  // block sweep of the Riccati kernels (wave_block_sweep, 4 x 4 pivot blocks, rank-4 updates on the matrix cores)
  template <int N>
  SMPC_DEV void fwave_spd_inverse(double * A, double * swp)
  {
    constexpr int NT = 64, NP4 = ((N + 3) / 4) * 4, NTI = (2 * NP4 + 15) / 16, LDW = 16 * NTI;
    SMPC_ACC(t, NT, NTI * (NTI + 1) / 2);
    SMPC_LANES(NT)
    {
      const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
      for (int I = 0; I < NTI; I++)
#pragma unroll
        for (int J = I; J < NTI; J++)
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            const int row = 16 * I + lr + 4 * v, col = 16 * J + lc;
            const int r = row < col ? row : col, c = row < col ? col : row;
            double val = 0.0;
            if (c < N)
              val = A[r * N + c];
            else if (c < NP4)
              val = r == c ? 1.0 : 0.0;       // padding pivots
            else if (c < 2 * NP4 && r < NP4)
              val = (c - NP4 == r) ? 1.0 : 0.0; // identity border
            SMPC_ACCV(t, tix<NTI>(I, J), v) = val;
          }
    }
    SMPC_LANES_END_WAVE
    double * prof = nullptr;
    long long tprev = 0;
    wave_block_sweep<NT, NTI, false, 0, NP4 / 4>(t, swp, swp + 4 * LDW, prof, tprev);
    SMPC_LANES(NT)
    {
      const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
      for (int I = NP4 / 16; I < NTI; I++)
#pragma unroll
        for (int J = I; J < NTI; J++)
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            const int row = 16 * I + lr + 4 * v - NP4, col = 16 * J + lc - NP4;
            if (row >= 0 && row < N && col >= row && col < N)
            {
              const double val = -SMPC_ACCV(t, tix<NTI>(I, J), v);
              A[row * N + col] = val;
              A[col * N + row] = val;
            }
          }
    }
    SMPC_LANES_END_WAVE
  }

  template <class D>
  struct RiccatiKinoLds
  {
    static constexpr int NDX = D::NDX, NU = D::NU, NXU = D::NDX + D::NU, NG = 12;
    static constexpr int SWP = 4 * 16 * 5; // one operand block of the widest sweep (4 x 80)
    static constexpr int SCR_1 = 2 * SWP, SCR_2 = NG * NXU, SCR_3 = NG * NDX + 2 * 4 * 64;
    static constexpr int SCR = SCR_1 > SCR_2 ? (SCR_1 > SCR_3 ? SCR_1 : SCR_3) : (SCR_2 > SCR_3 ? SCR_2 : SCR_3);
    // One array, offsets by name (entries of different blocks are reached from one per-lane base with compile-time offsets).
    // P: rows 0 .. NDX-1: P_{t+1} -> P~ (-> E^T P~ E in place) -> P_t, full symmetric image, row stride NDX;
    //    row NDX: c = p_{t+1} - f / mu -> p~ (-> E^T p~), row NDX+1: p_{t+1} -> p_t -- the vector column of the bordered matrix of the first sweep is
    //    read through the same (column, row) addresses as the matrix entries: element (R, C) of a tile comes from P[C' * NDX + R'], C' = NDX, NDX + 1 for it
    // Z: two zeros: where the masked lanes of a gather read (an address select instead of a predicated load)
    // scr, by phase:  sweep operands (2 x 4 x 80)  ->  dense rows of A (NG x NDX) | of B (NG x NU)
    //                 ->  [Cc (NG x NDX) | sweep operands (2 x 4 x 64)]
    static constexpr int O_P = 0, O_Z = O_P + (NDX + 2) * NDX, O_SCR = O_Z + 2, O_qh = O_SCR + SCR, O_rh = O_qh + NDX, O_dc = O_rh + NU, O_boxd = O_dc + NG,
                         O_boxact = O_boxd + D::NA, O_cone = O_boxact + D::NA, N_W = O_cone + 8 * D::NF;
    double w[N_W]; // (cone: friction-cone rows of the stage (force_cone): active Jacobian rows (2 NF x 3) | d (2 NF))
    float sink[64]; // destination of the line touches (never read)
  };

  // =============================================================================================
  // riccati_kino_body: grid = B, NT = 64 lanes (one wavefront per instance)
  // =============================================================================================
  template <class D, bool EXT = false> // EXT: optional constraint blocks present (see deriv_body)
  SMPC_DEV void riccati_kino_body(const SolverArgs<D> & ka, int block)
  {
    constexpr int NT = 64; // exactly one wavefront per workgroup: phases end with SMPC_LANES_END_WAVE
    constexpr int NDX = D::NDX, NU = D::NU, NV = D::NV, NA = D::NA, NF = D::NF, NXU = NDX + NU, NG = 12;
    typedef KinoIdx<D> IX;
    typedef GainsK<D> GK;
    static_assert(NDX % 4 == 0 && NU % 4 == 0 && NG % 4 == 0, "pivot panels and K steps of 4");
    static_assert(2 * NDX + 1 <= 80 && NXU + 1 <= 64, "bordered matrices fit the 5x5 / 4x4 tile grids");
    const Buffers<D> & b = ka.b;
    const int H = b.H;
    const int inst = block;
    // (kernel constants in scalar registers: as vector registers three of them were spilled in the prologue and reloaded in every stage)
    const double mu = SMPC_UNIFORM_F64(b.model->mu), imu = SMPC_UNIFORM_F64(1.0 / mu), dt = SMPC_UNIFORM_F64(b.model->dt), smu = SMPC_UNIFORM_F64(sqrt(mu));
    SMPC_LDS(RiccatiKinoLds<D>, lds, 1);
    RiccatiKinoLds<D> & s = lds[0];
    typedef RiccatiKinoLds<D> LD;
    double * const sP = s.w + LD::O_P;         // image of P (+ vector rows); sP[LD::O_Z] = 0
    double * const scr = s.w + LD::O_SCR;
    double * const NAl = scr;                  // [NG][NDX]  dense rows of A                  phase 4
    double * const NBl = scr + NG * NDX;       // [NG][NU]   dense rows of B                  phase 4
    double * const Cc = scr;                   // [NG][NDX]  contact rows                     phase 5-6
    double * const sw1 = scr;                  // sweep operands of the first (5x5 tiles) sweep: 2 x 4 x 80
    double * const sw2 = scr + NG * NDX;       // ... of the second (4x4 tiles): 2 x 4 x 64
    double * const qh = s.w + LD::O_qh, * const rh = s.w + LD::O_rh, * const dcv = s.w + LD::O_dc, * const boxd = s.w + LD::O_boxd,
           * const boxact = s.w + LD::O_boxact, * const cone = s.w + LD::O_cone;
    constexpr int ZP = LD::O_Z - LD::O_P; // the zero slot, as an index into sP

    SMPC_LANES(NT)
    {
      for (int i = lane; i < NDX * NDX; i += NT)
        sP[i] = b.QN[(size_t)inst * NDX * NDX + i];
      for (int i = lane; i < NDX; i += NT)
        sP[(NDX + 1) * NDX + i] = b.qN[(size_t)inst * NDX + i];
      if (lane < 2)
        sP[ZP + lane] = 0.0;
      if (lane < NU)
        rh[lane] = 0.0; // (written by the stages of problems with friction-cone rows only)
    }
    SMPC_LANES_END_WAVE
    // f of the stage, one double per lane, fetched one stage ahead
    SMPC_PL(double, f_pf, NT);
    SMPC_LANES(NT)
    SMPC_PLV(f_pf) = b.lq[((size_t)inst * H + (H - 1)) * D::LQ_STRIDE + D::O_f + (lane < NDX ? lane : 0)];
    SMPC_LANES_END_WAVE

#ifdef SMPC_KINO_RICCATI_PROF
    double * prof = (b.dbg != nullptr && block == 0) ? b.dbg : nullptr; // optional phase timers (block 0 only)
#else
    // (phase timers: an experiment build, tools/variant_build.sh <name> -DSMPC_KINO_RICCATI_PROF -- their 63 uniform branches per stage cost 1 %)
    double * prof = nullptr;
#endif
    long long tprev = SMPC_CLOCK();
    // address bases of the two sweeps' LDS staging (80- and 64-wide operand rows), formed once for the 50 stages
    SweepBases sb5, sb4;
    sweep_bases_init<80>(sb5);
    sweep_bases_init<64>(sb4);
    for (int t = H - 1; t >= 0; t--)
    {
      const double * lq = b.lq + ((size_t)inst * H + t) * D::LQ_STRIDE;
      double * g = b.gains + ((size_t)inst * H + t) * GK::STRIDE;
      // matrix-core tile bookkeeping: upper tiles (I <= J) of 16-padded grids (3x3 for P, 4x4 for H^, 5x5 bordered)
      constexpr int T3I[6] = {0, 0, 0, 1, 1, 2}, T3J[6] = {0, 1, 2, 1, 2, 2};
      constexpr int T4I[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, T4J[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
      constexpr int T5I[15] = {0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 3, 3, 4}, T5J[15] = {0, 1, 2, 3, 4, 1, 2, 3, 4, 2, 3, 4, 3, 4, 4};
      constexpr int QT[6] = {0, 1, 2, 4, 5, 7}; // the T4 tiles that hold Q^ (rows, cols < NDX)
      // per-lane registers that live across phases
      SMPC_ACC(hacc, NT, 10); // H^ = [Q S; S^T R] + [A|B]^T P~ [A|B], upper tiles; swept in place
      static_assert(NDX == 36 && NV == 18 && NU == 24 && 3 * NF == 12 && NA == 12,
                    "lane maps of the stage body: x = [qb(6) qj(12) vb(6) vj(12)] on 36 = 2 * 16 + 4, u = [forces(12) accelerations(12)]");
      constexpr int NAQ = NG * NDX, NBQ = NG * NU, NA_PL = (NAQ + NT - 1) / NT, NB_PL = (NBQ + NT - 1) / NT, NAB_PL = NA_PL + NB_PL;
      SMPC_PLA(double, nab_pf, NT, NAB_PL);
      SMPC_PL(double, pf_dc, NT); // d of the contact rows, box selectors and their d: fetched before the products of (4d), used in (5)
      SMPC_PL(double, pf_ba, NT);
      SMPC_PL(double, pf_bd, NT);
      // ---- (1) save p_{t+1} ; vector column of the pivot rows c = p - f / mu (see (2)) ----
      SMPC_LANES(NT)
      if (lane < NDX)
      {
        const double pv = sP[(NDX + 1) * NDX + lane];
        g[GK::G_pn + lane] = pv;
        sP[NDX * NDX + lane] = pv - imu * SMPC_PLV(f_pf);
      }
      SMPC_LANES_END_WAVE
      prof_tick(prof, 0, tprev);
      prof_tick(prof, 1, tprev);
      // ---- (2b) register prefetch of the dense rows of [A|B]: the latency overlaps with the P~ sweep.  (The contact rows
      //           of C and the stage vectors are fetched after the sweep, once the [A|B] registers are free again: held
      //           across the sweep's 15 accumulator tiles they were spilled right after the load, i.e. waited for) ----
      SMPC_LANES(NT)
      {
        // rows G = {0 .. 5, NV .. NV + 5} of A and of B are two contiguous runs each: flat copies, one address select where a load straddles them
#pragma unroll
        for (int n = 0; n < NA_PL; n++)
        {
          const int lo = n * NT, hi = lo + NT - 1, idx = lo + lane;
          const int off = hi < 6 * NDX ? idx : ((lo >= 6 * NDX && hi < NAQ) ? idx + (NV - 6) * NDX : (idx < 6 * NDX ? idx : (idx < NAQ ? idx + (NV - 6) * NDX : 0)));
          SMPC_PLV(nab_pf)[n] = lq[D::O_A + off];
        }
#pragma unroll
        for (int n = 0; n < NB_PL; n++)
        {
          const int lo = n * NT, hi = lo + NT - 1, idx = lo + lane;
          const int off = hi < 6 * NU ? idx : ((lo >= 6 * NU && hi < NBQ) ? idx + (NV - 6) * NU : (idx < 6 * NU ? idx : (idx < NBQ ? idx + (NV - 6) * NU : 0)));
          SMPC_PLV(nab_pf)[NA_PL + n] = lq[D::O_B + off];
        }
      }
      SMPC_LANES_END_WAVE
      prof_tick(prof, 2, tprev);
      // ---- (2) P~ and p~ as the Schur complement of the bordered matrix (pivots: the first NDX rows, 9 panels of 4)
      //              [ P + I / mu    P    c ]       c = p - f / mu
      //              [ P             P    p ]
      //          = P - P (P + I/mu)^-1 P = (I + mu P)^-1 P = P~   and   p - mu P~ c = p + P~ (f - mu p) = (I - mu P~)(p + P f) = p~.
      //          (The same elimination as on [[I + mu P, sqrt(mu) P], [., P]] -- rows and columns of the pivot block scaled by 1 / sqrt(mu) --
      //          with nothing to scale: every entry is an entry of the LDS image, at a compile-time offset from one of four per-lane bases.) ----
      {
        SMPC_ACC(t1, NT, 15);
        static_assert(NDX == 36 && NDX % 4 == 0, "tile geometry of the bordered matrix: blocks start at 36 = 2 * 16 + 4");
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
          // element (R, C) <- P[C' * NDX + R'],  R' = R - NDX [R >= NDX],  C' = C - NDX [C >= NDX]  (C' = NDX: the vector column)
          const int bA = lc * NDX + lr;                                     // tile columns 0, 1, 3: C' = lc + const
          const int bW = (lc < 4 ? lc + 32 : lc - 4) * NDX + lr;            // tile column 2: C = 32 + lc wraps at NDX
          const int b4p = (lc < 9 ? 28 + lc : NDX + 1) * NDX + lr;          // tile column 4, pivot rows: P columns 28 .. 35, c, (padding: p)
          const int b4s = (lc < 8 ? 28 + lc : NDX + 1) * NDX + lr;          // tile column 4, other rows: P columns 28 .. 35, p, (padding: p)
#pragma unroll
          for (int tt = 0; tt < 15; tt++)
#pragma unroll
            for (int v = 0; v < 4; v++)
            {
              const int I = T5I[tt], J = T5J[tt], R0 = 16 * I + 4 * v; // (compile-time: NDX is a multiple of 4)
              double val = 0.0;
              if (R0 < 2 * NDX) // (rows beyond the vector row: padding)
              {
                const bool piv = R0 < NDX;
                const int Rp = piv ? R0 : R0 - NDX;
                const int base = J <= 1 ? bA + 16 * J * NDX : (J == 2 ? bW : (J == 3 ? bA + (48 - NDX) * NDX : (piv ? b4p : b4s)));
                val = sP[base + Rp];
                if (piv && I == J && lc == lr + 4 * v)
                  val += imu;
              }
              SMPC_ACCV(t1, tt, v) = val;
            }
        }
        SMPC_LANES_END_WAVE
        prof_tick(prof, 3, tprev);
wave_block_sweep<NT, 5, false, 0, NDX / 4, false, SMPC_KINO_RCP1>(t1, sw1, sw1 + LD::SWP, prof, tprev, 0u, &sb5);
        // P~ (rows / columns NDX .. 2 NDX of the grid) -> LDS image, both halves, and -> the gains block (upper triangle packed row by row), from
        // the registers; p~ -> row NDX of the image (c is dead).  One execution mask per group of stores (a predicate per store costs four scalar instructions and a branch);
        // of a diagonal tile the upper entries only (its two halves are rounded differently: the image stays exactly symmetric).
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
          const int bR = lr * NDX + lc, bT = lc * NDX + lr; // (row, column) and mirrored
          const int gl = lc - lr * (lr - 1) / 2;            // packed offset of (Rp + lr, Cp + lc): const(Rp, Cp) + lr (NDX - 1 - Rp) + gl
          auto put = [&](int tt, int v) SMPC_LAMBDA_INLINE {
            const int I = T5I[tt], J = T5J[tt];
            const int Rp = 16 * I + 4 * v - NDX, Cp = 16 * J - NDX; // row lr + Rp, column lc + Cp of P~
            const double val = SMPC_ACCV(t1, tt, v);
            sP[bR + Rp * NDX + Cp] = val;
            sP[bT + Cp * NDX + Rp] = val;
            g[GK::G_Pt + Rp * (NDX - 1) - Rp * (Rp - 1) / 2 + Cp + lr * (NDX - 1 - Rp) + gl] = val;
          };
          constexpr int t22 = tix<5>(2, 2), t23 = tix<5>(2, 3), t24 = tix<5>(2, 4), t33 = tix<5>(3, 3), t34 = tix<5>(3, 4), t44 = tix<5>(4, 4);
#pragma unroll
          for (int v = 0; v < 4; v++)
            if (lc >= lr + 4 * v)
            {
              if (v >= 1)
                put(t22, v); // (rows NDX + 4 (v - 1) ..: columns lc >= 4 follow from the mask)
              put(t33, v);
              if (v < 2 && lc < 8)
                put(t44, v);
            }
#pragma unroll
          for (int v = 1; v < 4; v++)
            put(t23, v);
          if (lc < 8)
          {
#pragma unroll
            for (int v = 1; v < 4; v++)
              put(t24, v);
#pragma unroll
            for (int v = 0; v < 4; v++)
              put(t34, v);
          }
          if (lc == 8)
          {
#pragma unroll
            for (int v = 1; v < 4; v++)
              sP[NDX * NDX + 4 * v - 4 + lr] = SMPC_ACCV(t1, t24, v);
#pragma unroll
            for (int v = 0; v < 4; v++)
              sP[NDX * NDX + 12 + 4 * v + lr] = SMPC_ACCV(t1, t34, v);
#pragma unroll
            for (int v = 0; v < 2; v++)
              sP[NDX * NDX + 28 + 4 * v + lr] = SMPC_ACCV(t1, t44, v);
          }
        }
        SMPC_LANES_END_WAVE
      }
      prof_tick(prof, 4, tprev);
      // ---- (3b) register prefetch of [Q S; S^T R] in accumulator-tile layout (element (row, col) of tile (I, J):
      //           row = 16 I + (lane >> 4) + 4 v, col = 16 J + (lane & 15)); the HBM/L2 latency overlaps with the
      //           P~ E passes and the TG product below.  Column NXU of the grid is the vector column of the second sweep: the derivative pass
      //           stores q / r there as well (smpc_kino_kernels.h) ----
      SMPC_LANES(NT)
      {
#pragma unroll
        for (int tt = 0; tt < 10; tt++)
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            // the knot keeps the tiles in this layout (Dims::O_T): one 512-byte run per accumulator register, no address arithmetic
            // (padding entries beyond the problem are zeros written once by lq_init_body; they only ever feed themselves)
            SMPC_ACCV(hacc, tt, v) = lq[D::O_T + (tt * 4 + v) * 64 + lane];
          }
      }
      SMPC_LANES_END_WAVE
      // commit the dense rows of [A|B]
      SMPC_LANES(NT)
      {
#pragma unroll
        for (int n = 0; n < NA_PL; n++)
          if (n * NT + NT <= NAQ || lane + n * NT < NAQ)
            NAl[lane + n * NT] = SMPC_PLV(nab_pf)[n];
#pragma unroll
        for (int n = 0; n < NB_PL; n++)
          if (n * NT + NT <= NBQ || lane + n * NT < NBQ)
            NBl[lane + n * NT] = SMPC_PLV(nab_pf)[NA_PL + n];
        // The contact rows of C, d and the box selectors are consumed after the products.  Eleven prefetch registers per lane
        // do not survive the products' accumulator tiles (the compiler spilled them right after the load, i.e. waited for
        // every one of them), so the lines are only TOUCHED here -- one 4-byte load per 64-byte line into an LDS sink, no
        // register and no wait -- which brings them into L2; the real loads in (5) then return in one L2 round trip.
        static_assert(NG * NDX <= 54 * 8 && NA + NG <= 24 && NA <= NT, "line touches: 54 lines of C, 3 of d, one per box row");
        SMPC_TOUCH(lq + (lane < 54 ? D::O_C + NA * NDX + lane * 8 : D::O_d + (lane < 57 ? (lane - 54) * 8 : 0)), s.sink);
        SMPC_TOUCH(lq + D::O_C + (lane < NA ? lane * NDX + 6 + lane : 0), s.sink);
      }
      SMPC_LANES_END_WAVE
      prof_tick(prof, 5, tprev);
      // ---- (4a) column pass, in place:  P[:, vj] += dt P[:, qj]   (P now holds P~ E on its J columns; row NDX: p~ -> E^T p~) ----
      // E_b = dt * E[:, vj], so every product with E_b is a scaled slice of the same matrix.
      SMPC_LANES(NT)
      {
        // lane -> (row group, column): NA columns x 5 row groups, no index division in the loop
        const int jp = lane % NA, ib = lane / NA;
        if (ib < 5)
          for (int i = ib; i < NDX + 1; i += 5)
            sP[i * NDX + NV + 6 + jp] += dt * sP[i * NDX + 6 + jp];
      }
      SMPC_LANES_END_WAVE
      static_assert(5 * NA <= NT, "lane map of the column pass");
      prof_tick(prof, 6, tprev);
      // ---- (4b, 4c) PEG = [(P~ E)[G,:] | (P~ E_b)[G,:]] (zero in the G / force columns), gathered from P straight
      //      into accumulator layout;  TG = P~[G,G] * NAB + PEG on the matrix cores (M = NG -> 16, N = NXU -> 64, K = NG).
      //      Both stay in registers: in accumulator layout a lane holds rows lr, lr + 4, lr + 8 of its column, which
      //      are exactly the rows 4 ks + lr it must supply as an MFMA operand at K-step ks of the next product.
      //      Gathers: per-lane bases + compile-time offsets. ----
      constexpr int KS = NG / 4;
      SMPC_ACC(tacc, NT, 4);
      SMPC_PLA(double, pegv, NT, 4 * KS);
      {
        SMPC_PLA(double, pgv, NT, KS);
        SMPC_PLA(double, nbv, NT, KS * 4);
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
          // dense row index G(4 v + lr), v = 0, 1, 2
          const int gq1 = lr < 2 ? lr + 4 : lr + NV - 2;
          const int rb[3] = {lr * NDX, gq1 * NDX, (lr + NV + 2) * NDX};
          // columns of [x | u] with a structured entry, by tile column: x not in G ; accelerations -> their vj column (x dt).  The lanes of the
          // other columns read a finite entry of the image and multiply it by zero (a scale per lane instead of an address select per entry)
          const double cs[4] = {lc >= 6 ? 1.0 : 0.0, (lc < 2 || lc >= 8) ? 1.0 : 0.0, lc < 4 ? 1.0 : 0.0, lc < NA ? dt : 0.0};
          const int pc[4] = {lc, 16 + lc, 32 + lc, NV + 6 + lc};
#pragma unroll
          for (int J = 0; J < 4; J++)
#pragma unroll
            for (int v = 0; v < 4; v++)
            {
              double pv = 0.0;
              if (v < KS)
              {
                pv = sP[rb[v] + pc[J]] * cs[J];
                SMPC_PLV(pegv)[J * KS + v] = pv;
              }
              SMPC_ACCV(tacc, J, v) = pv;
            }
          const int gcol = lc < 6 ? lc : lc + NV - 6; // G(lc), lc < NG
          const int na = lr * NDX + lc, nb = lr * NU + lc;
#pragma unroll
          for (int ks = 0; ks < KS; ks++)
          {
            SMPC_PLV(pgv)[ks] = sP[rb[ks] + gcol] * (lc < NG ? 1.0 : 0.0);
            SMPC_PLV(nbv)[ks * 4 + 0] = NAl[na + 4 * ks * NDX];
            SMPC_PLV(nbv)[ks * 4 + 1] = NAl[na + 4 * ks * NDX + 16];
            SMPC_PLV(nbv)[ks * 4 + 2] = scr[lc < 4 ? na + 4 * ks * NDX + 32 : NAQ + nb + 4 * ks * NU - 4];
            SMPC_PLV(nbv)[ks * 4 + 3] = lc < NU - 12 ? NBl[nb + 4 * ks * NU + 12] : 0.0;
          }
        }
        SMPC_LANES_END_WAVE
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
          for (int J = 0; J < 4; J++)
            SMPC_MFMA(tacc, J, pgv, ks, nbv, ks * 4 + J);
        // row pass, in place: P[vj, :] += dt P[qj, :]  (P[J,J] = E^T P~ E)
        SMPC_LANES(NT)
        if (lane < NDX)
          for (int ip = 0; ip < NA; ip++)
            sP[(NV + 6 + ip) * NDX + lane] += dt * sP[(6 + ip) * NDX + lane];
        SMPC_LANES_END_WAVE
      }
      prof_tick(prof, 7, tprev);
      // ---- (4d) H^ += E^T P~ [E|E_b] (structured) + NAB^T TG + PEG^T NAB  (matrix cores, K = 2 NG).  The vector column rides along:
      //      column NXU of TG := p~[G] gives NAB^T p~[G] by the first product, row NDX of the image (E^T p~) the structured part ----
      {
        SMPC_PLA(double, nav, NT, KS * 4);
        SMPC_PLA(double, tgv, NT, KS * 4);
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
          const int gq[3] = {lr, lr < 2 ? lr + 4 : lr + NV - 2, lr + NV + 2};
          const int na = lr * NDX + lc, nb = lr * NU + lc;
#pragma unroll
          for (int ks = 0; ks < KS; ks++)
          {
            SMPC_PLV(nav)[ks * 4 + 0] = NAl[na + 4 * ks * NDX];
            SMPC_PLV(nav)[ks * 4 + 1] = NAl[na + 4 * ks * NDX + 16];
            SMPC_PLV(nav)[ks * 4 + 2] = scr[lc < 4 ? na + 4 * ks * NDX + 32 : NAQ + nb + 4 * ks * NU - 4];
            SMPC_PLV(nav)[ks * 4 + 3] = lc < NU - 12 ? NBl[nb + 4 * ks * NU + 12] : 0.0;
#pragma unroll
            for (int J = 0; J < 3; J++)
              SMPC_PLV(tgv)[ks * 4 + J] = SMPC_ACCV(tacc, J, ks);
            // (TG is zero in the padding columns: its lanes lc >= NU - 12 of tile column 3 hold 0 + 0)
            SMPC_PLV(tgv)[ks * 4 + 3] = SMPC_ACCV(tacc, 3, ks) + sP[lc == NXU % 16 ? NDX * NDX + gq[ks] : ZP];
          }
          // structured term: entry (r, c) of [x | u] x [x | u] -> image entry (pr, pc) x scale: x: itself unless in G (nothing) ;
          // joint accelerations: their vj row / column x dt ; forces: nothing.  Rows by register (compile-time), columns by lane (masks).
          const int bR = lr * NDX + lc;
          const bool rm2 = lr >= 2;
          const double cs[4] = {lc >= 6 ? 1.0 : 0.0, (lc < 2 || lc >= 8) ? 1.0 : 0.0, lc < 4 ? 1.0 : 0.0, lc < NA ? dt : 0.0}; // column scale / mask
          double csa[4], csb[4]; // ... x row mask of the two registers whose four rows are partly in G
#pragma unroll
          for (int J = 0; J < 4; J++)
          {
            csa[J] = rm2 ? cs[J] : 0.0;
            csb[J] = rm2 ? 0.0 : cs[J];
          }
          const double csd = cs[3] * dt;
#pragma unroll
          for (int tt = 0; tt < 10; tt++)
#pragma unroll
            for (int v = 0; v < 4; v++)
            {
              const int I = T4I[tt], J = T4J[tt], R0 = 16 * I + 4 * v;
              // rows: 0..3 G | 4,5 G, 6,7 | 8..15 | 16,17, 18,19 G | 20..23 G | 24..35 | forces | 48..59 accelerations | padding
              const bool rnone = R0 < 4 || (R0 >= 20 && R0 < 24) || (R0 >= NDX && R0 < NDX + 12) || R0 >= NXU;
              if (rnone)
                continue;
              const bool racc = R0 >= NDX + 12; // (tile (3, 3) only)
              const int pr0 = racc ? R0 - NDX - 12 + NV + 6 : R0, pc0 = J == 3 ? NV + 6 : 16 * J;
              const double sc = racc ? csd : (R0 == 4 ? csa[J] : (R0 == 16 ? csb[J] : cs[J]));
              SMPC_ACCV(hacc, tt, v) += sc * sP[bR + pr0 * NDX + pc0];
            }
          // ... of the vector column: E^T p~ (row NDX of the image)
          if (lc == NXU % 16)
          {
#pragma unroll
            for (int I = 0; I < 4; I++)
#pragma unroll
              for (int v = 0; v < 4; v++)
              {
                const int R0 = 16 * I + 4 * v;
                const bool rnone = R0 < 4 || (R0 >= 20 && R0 < 24) || (R0 >= NDX && R0 < NDX + 12) || R0 >= NXU;
                if (rnone)
                  continue;
                const bool racc = R0 >= NDX + 12;
                const int pr0 = racc ? R0 - NDX - 12 + NV + 6 : R0;
                const bool m = R0 == 4 ? rm2 : (R0 == 16 ? !rm2 : true);
                const double pv = sP[m ? NDX * NDX + pr0 + lr : ZP];
                SMPC_ACCV(hacc, tix<4>(I, 3), v) += racc ? dt * pv : pv;
              }
          }
        }
        SMPC_LANES_END_WAVE
        // the dense rows of [A|B] are in registers: their block takes the contact rows of C (one contiguous run of the knot, 16 bytes per lane and
        // copy, no registers), the small vectors of (5) go to registers -- all of it lands while the products run
        SMPC_LANES(NT)
        {
          static_assert(4 * 2 * NT >= NG * NDX && 4 * 2 * NT <= LD::SCR && D::O_C + NA * NDX + 4 * 2 * NT <= D::LQ_STRIDE, "four copies of 64 x 16 bytes");
#pragma unroll
          for (int n = 0; n < 4; n++)
            SMPC_COPY16_TO_LDS(lq + D::O_C + NA * NDX + 2 * (lane + NT * n), Cc + 2 * NT * n);
          SMPC_PLV(pf_dc) = lq[D::O_d + NA + (lane < NG ? lane : 0)];
          SMPC_PLV(pf_ba) = lq[D::O_C + (lane < NA ? lane * NDX + 6 + lane : 0)]; // 1 if the box row is active
          SMPC_PLV(pf_bd) = lq[D::O_d + (lane < NA ? lane : 0)];
          if (t > 0)
            SMPC_PLV(f_pf) = lq[-(int)D::LQ_STRIDE + D::O_f + (lane < NDX ? lane : 0)]; // f of the next stage of the sweep
        }
        SMPC_LANES_END_WAVE
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
          for (int tt = 0; tt < 10; tt++)
          {
            SMPC_MFMA(hacc, tt, nav, ks * 4 + T4I[tt], tgv, ks * 4 + T4J[tt]);
            SMPC_MFMA(hacc, tt, pegv, T4I[tt] * KS + ks, nav, ks * 4 + T4J[tt]);
          }
      }
      prof_tick(prof, 8, tprev);
      // land_cstr rows present at this stage (wave-uniform: the stage descriptors are shared by the batch)
      const unsigned landrows = (EXT && b.ls != nullptr) ? (b.stages[t].land & b.stages[t].mask) : 0u;
      // ---- (5) Q^ += C^T C / mu + box  (matrix cores, K = NG), the vector column += C^T d / mu ----
      SMPC_COPY_TO_LDS_WAIT();
      SMPC_LANES(NT)
      {
        // (the contact rows are in Cc: asynchronous copy issued before the products of (4d))
        const int pl = SMPC_PIN(lane);
        const double * ekp = (EXT && b.es != nullptr) ? b.ek + ((size_t)inst * H + t) * 12 * NF : lq; // (cone rows: [D | d] are contiguous)
        const double vce = ekp[pl < 8 * NF ? pl : 0];
        const double vdc = SMPC_PLV(pf_dc), vba = SMPC_PLV(pf_ba), vbd = SMPC_PLV(pf_bd);
        if (lane < NG)
          dcv[lane] = vdc;
        if (lane < NA)
        {
          boxact[lane] = vba;
          boxd[lane] = vbd;
        }
        if constexpr (EXT)
          if (lane < 8 * NF)
            cone[lane] = b.es != nullptr ? vce : 0.0;
      }
      SMPC_LANES_END_WAVE
      {
        constexpr int KC = NG / 4;
        SMPC_PLA(double, cpv, NT, KC * 3); // B operands: Cc
        SMPC_PLA(double, cnv, NT, KC * 3); // A operands: Cc / mu
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
          for (int ks = 0; ks < KC; ks++)
#pragma unroll
            for (int I = 0; I < 3; I++)
            {
              const int col = 16 * I + lc;
              const double raw = Cc[(4 * ks + lr) * NDX + (col < NDX ? col : 0)];
              const double val = col < NDX ? raw : 0.0;
              SMPC_PLV(cpv)[ks * 3 + I] = val;
              SMPC_PLV(cnv)[ks * 3 + I] = imu * val;
            }
          if (lane < NDX)
          {
            const int i = lane;
            // (all reads first: a rolled loop waits for the LDS once per trip)
            double cc[NG];
#pragma unroll
            for (int r = 0; r < NG; r++)
              cc[r] = Cc[r * NDX + i];
            const bool bx = IX::isQj(i);
            const double bb = boxact[bx ? i - 6 : 0] * boxd[bx ? i - 6 : 0];
            double cd = bx ? bb : 0.0;
#pragma unroll
            for (int r = 0; r < NG; r++)
              cd += cc[r] * dcv[r];
            qh[i] = imu * cd; // (this phase's part of the vector column; added to it below)
          }
          if (landrows != 0u && lane < NV)
          { // q^ += c^T d / mu of the land rows (rows on q only)
            const double * lkp = b.lk + ((size_t)inst * H + t) * NF * (NV + 2);
            double acc = 0.0;
            for (int f = 0; f < NF; f++)
              if ((landrows >> f) & 1u)
                acc += lkp[f * NV + lane] * lkp[NF * NV + f];
            qh[lane] += imu * acc;
          }
          else if ((EXT && b.es != nullptr) && lane >= NDX && lane < NDX + 3 * NF)
          { // r^ += D^T d / mu of the friction-cone rows
            const int k = lane - NDX, f = k / 3;
            rh[k] = imu * (cone[(2 * f) * 3 + k % 3] * cone[6 * NF + 2 * f] + cone[(2 * f + 1) * 3 + k % 3] * cone[6 * NF + 2 * f + 1]);
          }
        }
        SMPC_LANES_END_WAVE
#pragma unroll
        for (int ks = 0; ks < KC; ks++)
#pragma unroll
          for (int tt = 0; tt < 6; tt++)
            SMPC_MFMA(hacc, QT[tt], cnv, ks * 3 + T3I[tt], cpv, ks * 3 + T3J[tt]);
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
          // box rows: unit selectors -> diagonal contribution on the qj entries (rows 6 .. NV - 1: registers (0, 0) v = 1, 2, 3 and (1, 1) v = 0);
          // all reads first, no predicated read-wait-add per register
          {
            double ba[4];
#pragma unroll
            for (int n = 0; n < 4; n++)
            {
              const int R0 = n < 3 ? 4 * (n + 1) : 16, v = n < 3 ? n + 1 : 0;
              const bool m = lc == lr + 4 * v && R0 + lr >= 6 && R0 + lr < NV;
              ba[n] = boxact[m ? R0 + lr - 6 : 0] * (m ? imu : 0.0);
            }
#pragma unroll
            for (int n = 0; n < 4; n++)
              SMPC_ACCV(hacc, tix<4>(n < 3 ? 0 : 1, n < 3 ? 0 : 1), n < 3 ? n + 1 : 0) += ba[n];
          }
          // land rows: Q^[q, q] += c^T c / mu (rank one per landing foot; at most a few stages of a horizon have any)
          if (landrows != 0u)
          {
            static_assert(NV <= 32, "the q-q block lives in tiles (0,0), (0,1), (1,1)");
            const double * lkp = b.lk + ((size_t)inst * H + t) * NF * (NV + 2);
            for (int f = 0; f < NF; f++)
              if ((landrows >> f) & 1u)
              {
                const double * cf = lkp + f * NV;
#pragma unroll
                for (int I = 0; I < 2; I++)
#pragma unroll
                  for (int J = I; J < 2; J++)
                  {
                    const int col = 16 * J + lc;
                    const double cc = cf[col < NV ? col : 0];
#pragma unroll
                    for (int v = 0; v < 4; v++)
                    {
                      const int row = 16 * I + lr + 4 * v;
                      const double cr = cf[row < NV ? row : 0];
                      if (row < NV && col < NV)
                        SMPC_ACCV(hacc, tix<4>(I, J), v) += imu * cr * cc;
                    }
                  }
              }
          }
          // friction-cone rows act on the force part of u only: R^ += D^T D / mu (3 x 3 block per foot), r^ += D^T d / mu
          if ((EXT && b.es != nullptr))
          {
            static_assert(NDX / 16 == 2 && NDX + 3 * NF <= 48, "the force block of R^ lives in tile (2, 2)");
#pragma unroll
            for (int v = 0; v < 4; v++)
            {
              const int row = 32 + lr + 4 * v - NDX, col = 32 + lc - NDX; // force indices
              if (row >= 0 && row < 3 * NF && col >= 0 && col < 3 * NF && row / 3 == col / 3)
              {
                const int f = row / 3;
                const double * d0 = &cone[(2 * f) * 3], * d1 = d0 + 3;
                SMPC_ACCV(hacc, tix<4>(2, 2), v) += imu * (d0[row % 3] * d0[col % 3] + d1[row % 3] * d1[col % 3]);
              }
            }
          }
          // vector column NXU (lanes with lc == NXU % 16 of the last tile column): += C^T d / mu (+ the force part of the cone rows)
          if (lc == NXU % 16)
          {
#pragma unroll
            for (int I = 0; I < 3; I++)
#pragma unroll
              for (int v = 0; v < 4; v++)
              {
                const int R0 = 16 * I + 4 * v;
                if (R0 < NDX)
                  SMPC_ACCV(hacc, tix<4>(I, 3), v) += qh[R0 + lr];
                else if (EXT && R0 < NDX + 3 * NF)
                  SMPC_ACCV(hacc, tix<4>(I, 3), v) += rh[R0 - NDX + lr];
              }
          }
        }
        SMPC_LANES_END_WAVE
      }
      static_assert(NXU / 16 == 3 && NXU % 16 != 0, "vector column lives in the last tile column of the 4x4 grid");
      prof_tick(prof, 9, tprev);
      // ---- (6) sweep the NU control pivots (rows NDX .. NXU) of
      //              [ Q^ + C^T C/mu   S^    q^ + C^T d/mu ]
      //              [ S^^T            R^    r^            ]
      //          in place:  x-x block -> P_t,  x-vector -> p_t,  stored (x, u) entries -> R^^-1 S^^T = -K,
      //          (u, vector) entries -> R^^-1 r^ = -k ----
      wave_block_sweep<NT, 4, true, NDX, NU / 4, false, SMPC_KINO_RCP1>(hacc, sw2, sw2 + 4 * 64, prof, tprev, 0u, &sb4);
      prof_tick(prof, 10, tprev);
      // P_t -> LDS image (both halves; of a diagonal tile the upper entries), p_t -> its row NDX + 1, [K | k]^T -> gains block, from the registers
      // (a 16-lane row of a tile = 12 consecutive u of one x); grouped by mask
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
        const int bR = lr * NDX + lc, bT = lc * NDX + lr;
        const int bK = lr * NU + lc; // K^T[x][u] of tile entry (x = R0 + lr, u = U0 + lc)
        auto putP = [&](int tt, int v) SMPC_LAMBDA_INLINE {
          const int R0 = 16 * T4I[tt] + 4 * v, C0 = 16 * T4J[tt];
          const double val = SMPC_ACCV(hacc, tt, v);
          sP[bR + R0 * NDX + C0] = val;
          sP[bT + C0 * NDX + R0] = val;
        };
        auto putK = [&](int tt, int v) SMPC_LAMBDA_INLINE {
          const int R0 = 16 * T4I[tt] + 4 * v, U0 = 16 * T4J[tt] - NDX;
          g[GK::G_W + bK + R0 * NU + U0] = -SMPC_ACCV(hacc, tt, v);
        };
        constexpr int h00 = tix<4>(0, 0), h01 = tix<4>(0, 1), h02 = tix<4>(0, 2), h03 = tix<4>(0, 3), h11 = tix<4>(1, 1), h12 = tix<4>(1, 2),
                      h13 = tix<4>(1, 3), h22 = tix<4>(2, 2), h23 = tix<4>(2, 3), h33 = tix<4>(3, 3);
#pragma unroll
        for (int v = 0; v < 4; v++)
          if (lc >= lr + 4 * v)
          {
            putP(h00, v);
            putP(h11, v);
            if (v == 0 && lc < 4)
              putP(h22, 0);
          }
#pragma unroll
        for (int v = 0; v < 4; v++)
          putP(h01, v);
        if (lc < 4)
        {
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            putP(h02, v);
            putP(h12, v);
          }
        }
        else
        {
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            putK(h02, v);
            putK(h12, v);
          }
          putK(h22, 0);
        }
        if (lc < 12)
        {
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            putK(h03, v);
            putK(h13, v);
          }
          putK(h23, 0);
        }
        else if (lc == 12)
        {
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            sP[(NDX + 1) * NDX + 4 * v + lr] = SMPC_ACCV(hacc, h03, v); // p_t
            sP[(NDX + 1) * NDX + 16 + 4 * v + lr] = SMPC_ACCV(hacc, h13, v);
          }
          sP[(NDX + 1) * NDX + 32 + lr] = SMPC_ACCV(hacc, h23, 0);
#pragma unroll
          for (int v = 1; v < 4; v++)
            g[GK::G_W + NDX * NU + 4 * v - 4 + lr] = -SMPC_ACCV(hacc, h23, v); // k
#pragma unroll
          for (int v = 0; v < 3; v++)
            g[GK::G_W + NDX * NU + 12 + 4 * v + lr] = -SMPC_ACCV(hacc, h33, v);
        }
      }
      SMPC_LANES_END_WAVE
      prof_tick(prof, 13, tprev);
    }
  }

  // =============================================================================================
  // forward_kino_body: grid = B, 64 lanes.  Forward sweep with the factored feedback + merit derivative.
  // The blocks a stage needs (W, L_R^-1, P~, p+, dense rows of [A|B], contact rows of C, vectors) are contiguous
  // chunks of the gains / LQ blocks: they are fetched with coalesced loads into registers one stage AHEAD (the
  // loads fly while the current stage computes from LDS) and committed to LDS at the stage boundary.
  // =============================================================================================
  template <class D>
  struct ForwardKinoLds
  {
    static constexpr int NDX = D::NDX, NU = D::NU, NG = 12;
    typedef GainsK<D> GK;
    static constexpr int N_W = NU * (NDX + 1), N_P = NDX * (NDX + 1) / 2;    // [K | k] ; packed P~
    static constexpr int N_A = 6 * NDX, N_B = 6 * NU, N_C = NG * NDX;        // row groups qb / vb of A, B; contact rows of C
    static constexpr int N_V = D::O_vpd + D::NC - D::O_f;                    // f d lx lu lpd vpd
    // flat staging index of a stage.  Early part (committed to LDS at the stage boundary):
    //   0 [K | k] | 1 p+ | 2,3 A rows qb, vb | 4,5 B rows qb, vb | 6 contact rows of C | 7 vectors | 8 box diag
    // late part: 9 packed P~, committed over [K | k] once du is formed (P~ is read last) | 10 pad
    static constexpr int O_pn = N_W, O_A0 = O_pn + NDX, O_A1 = O_A0 + N_A, O_B0 = O_A1 + N_A, O_B1 = O_B0 + N_B, O_Cc = O_B1 + N_B,
                         O_V = O_Cc + N_C, O_box = O_V + N_V, O_late = O_box + D::NA, N_STAGE = O_late + N_P;
    static_assert(N_P <= N_W, "P~ is staged over [K | k]");
    static constexpr int PER_LANE = (N_STAGE + 63) / 64;
    static constexpr int PER_EARLY = (O_late + 63) / 64;   // registers holding early entries (the last one may straddle)
    static constexpr int FULL_EARLY = O_late / 64;         // registers holding early entries only
    SMPC_HD static constexpr int start(int c)
    {
      return c == 0 ? 0 : c == 1 ? O_pn : c == 2 ? O_A0 : c == 3 ? O_A1 : c == 4 ? O_B0 : c == 5 ? O_B1 : c == 6 ? O_Cc : c == 7 ? O_V : c == 8 ? O_box
                                                                                                                                   : c == 9 ? O_late : N_STAGE;
    }
    SMPC_HD static constexpr int chunk(int idx)
    {
      return idx < O_pn ? 0 : idx < O_A0 ? 1 : idx < O_A1 ? 2 : idx < O_B0 ? 3 : idx < O_B1 ? 4 : idx < O_Cc ? 5 : idx < O_V ? 6 : idx < O_box ? 7 : idx < O_late ? 8
                                                                                                                                    : idx < N_STAGE ? 9 : 10;
    }
    SMPC_HD static constexpr bool from_gains(int c) { return c == 0 || c == 1 || c == 9; }
    // offset of the chunk's source: in the gains block (chunks 0, 1, 9) or in the LQ block (chunks 2 .. 7)
    SMPC_HD static constexpr int src(int c)
    {
      return c == 0 ? GK::G_W : c == 1 ? GK::G_pn : c == 9 ? GK::G_Pt : c == 2 ? D::O_A : c == 3 ? D::O_A + D::NV * NDX : c == 4 ? D::O_B
             : c == 5 ? D::O_B + D::NV * NU : c == 6 ? D::O_C + D::NA * NDX : D::O_f;
    }
    double st[PER_EARLY * 64];
    double dx[NDX], du[NU], y[NDX], part[64], lpd_prev[NDX];
  };

  template <class D, bool EXT = false> // EXT: optional constraint blocks present (see deriv_body)
  SMPC_DEV void forward_kino_body(const SolverArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NDX = D::NDX, NU = D::NU, NC = D::NC, NV = D::NV, NA = D::NA, NF = D::NF;
    typedef GainsK<D> GK;
    typedef KinoIdx<D> IX;
    typedef ForwardKinoLds<D> FL;
    static_assert(D::O_d == D::O_f + NDX && D::O_lx == D::O_d + NC && D::O_lu == D::O_lx + NDX && D::O_lpd == D::O_lu + NU &&
                    D::O_vpd == D::O_lpd + NDX,
                  "vector chunk must be contiguous");
    constexpr int PER = FL::PER_LANE;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const int inst = block;
    const double mu = b.model->mu, dt = b.model->dt;
    SMPC_LDS(FL, lds, 1);
    FL & s = lds[0];
    SMPC_PLA(double, pf, NT, PER);
    // fetch stage t into the per-lane registers (flat staging index -> source chunk).  Chunk boundaries are
    // compile-time: for most n all 64 lanes fall into one chunk (plain base + lane address); the few straddling
    // ones use an integer select chain -- no divergent branches.
    auto fetch = [&](int lane, double * r, int t, int n0, int n1) {
      const double * lq = b.lq + ((size_t)inst * H + t) * D::LQ_STRIDE;
      const double * g = b.gains + ((size_t)inst * H + t) * GK::STRIDE;
#pragma unroll
      for (int n = 0; n < PER; n++)
      {
        if (n < n0 || n >= n1)
          continue;
        const int lo = n * NT, hi = n * NT + NT - 1;
        if (FL::chunk(lo) == FL::chunk(hi))
        {
          const int c = FL::chunk(lo);
          if (FL::from_gains(c))
            r[n] = g[FL::src(c) + lo - FL::start(c) + lane];
          else if (c < 8)
            r[n] = lq[FL::src(c) + lo - FL::start(c) + lane];
          else if (c == 8)
            r[n] = lq[D::O_C + (lo + lane - FL::O_box) * (NDX + 1) + 6];
          else
            r[n] = 0.0;
        }
        else
        {
          const int idx = lo + lane;
          int off = 0;       // offset into lq or into the gains block
          bool ing = false;
#pragma unroll
          for (int c = 0; c < 10; c++)
          {
            const bool in = idx >= FL::start(c) && idx < FL::start(c + 1);
            if (c == 8)
              off = in ? D::O_C + (idx - FL::O_box) * (NDX + 1) + 6 : off;
            else
              off = in ? FL::src(c) + idx - FL::start(c) : off;
            ing = (in && FL::from_gains(c)) ? true : ing;
          }
          const double * src = ing ? g + off : lq + off; // address select, ONE load, no wait on its value
          r[n] = *src;                                   // (pad entries load lq[0]; never read back)
        }
      }
    };
    SMPC_LANES(NT)
    {
      if (lane < NDX)
      {
        s.dx[lane] = 0.0;
        s.lpd_prev[lane] = 0.0;
        b.dxs[((size_t)inst * (H + 1)) * NDX + lane] = 0.0;
      }
      s.part[lane] = 0.0;
      fetch(lane, SMPC_PLV(pf), 0, 0, PER);
    }
    SMPC_LANES_END_WAVE
    const double * W = s.st;              // [K | k] until du is formed, then the packed P~
    const double * pn = s.st + FL::O_pn;
    const double * vf = s.st + FL::O_V;
    const double *vd = vf + NDX, *vlx = vd + NC, *vlu = vlx + NDX, *vlpd = vlu + NU, *vvpd = vlpd + NDX;
    const double * boxact = s.st + FL::O_box;
    for (int t = 0; t < H; t++)
    {
      const size_t lt = (size_t)inst * H + t;
      // commit stage t to LDS, start fetching stage t+1
      SMPC_LANES(NT)
      {
#pragma unroll
        for (int n = 0; n < FL::PER_EARLY; n++)
          if (lane + n * NT < FL::O_late)
            s.st[lane + n * NT] = SMPC_PLV(pf)[n];
        if (t + 1 < H)
          fetch(lane, SMPC_PLV(pf), t + 1, 0, FL::FULL_EARLY);
      }
      SMPC_LANES_END_WAVE
      // ---- du = K dx + k ; dnu = (C dx + d)/mu ----
      // friction-cone row of lanes NDX .. NDX + 2 NF: Jacobian row (3) | d | vpd | the foot's other row (3) | activity of both rows
      SMPC_PLA(double, ce, NT, 10);
      SMPC_PL(double, gpart, NT); // lanes 48 .. 48 + 3 NF: stationarity residual of force component k without the B^T dlam term
      SMPC_PL(double, dlr, NT);   // lanes < NDX: dlam of this stage
      SMPC_LANES(NT)
      {
        if ((EXT && b.es != nullptr) && lane >= NDX && lane < NDX + 2 * NF)
        {
          const double * ekp = b.ek + lt * 12 * NF;
          const int i = lane - NDX, ip = i ^ 1;
          SMPC_PLV(ce)[0] = ekp[i * 3];
          SMPC_PLV(ce)[1] = ekp[i * 3 + 1];
          SMPC_PLV(ce)[2] = ekp[i * 3 + 2];
          SMPC_PLV(ce)[3] = ekp[6 * NF + i];
          SMPC_PLV(ce)[4] = ekp[8 * NF + i];
          SMPC_PLV(ce)[5] = ekp[ip * 3];
          SMPC_PLV(ce)[6] = ekp[ip * 3 + 1];
          SMPC_PLV(ce)[7] = ekp[ip * 3 + 2];
          SMPC_PLV(ce)[8] = ekp[10 * NF + i];
          SMPC_PLV(ce)[9] = ekp[10 * NF + ip];
        }
        if (lane < NU)
        {
          const double * Wc = &W[lane]; // column `lane` of [K | k]^T
          double acc = Wc[NDX * NU];
#pragma unroll 4
          for (int j = 0; j < NDX; j++)
            acc += Wc[j * NU] * s.dx[j];
          s.du[lane] = acc;
          b.dus[lt * NU + lane] = acc;
          s.part[lane] += vlu[lane] * acc;
        }
        else if (lane < NU + NC)
        {
          const int r = lane - NU;
          double acc = vd[r];
          if (r < NA)
            acc += boxact[r] * s.dx[6 + r];
          else
          {
            const double * Cr = &s.st[FL::O_Cc + (r - NA) * NDX];
#pragma unroll 4
            for (int j = 0; j < NDX; j++)
              acc += Cr[j] * s.dx[j];
          }
          const double dnu = acc / mu;
          b.dvs[lt * NC + r] = dnu;
          s.part[lane] += vvpd[r] * (mu * dnu - vd[r]) - vd[r] * dnu;
        }
        else if ((EXT && b.ls != nullptr) && lane < NU + NC + NF)
        { // land rows: dnu = (c dx_q + d) / mu  (absent rows: c = 0)
          const int f = lane - NU - NC;
          const double * lkp = b.lk + lt * NF * (NV + 2);
          const double dd = lkp[NF * NV + f], vpd = lkp[NF * NV + NF + f];
          double acc = dd;
          if (((b.stages[t].land & b.stages[t].mask) >> f) & 1u)
            for (int j = 0; j < NV; j++)
              acc += lkp[f * NV + j] * s.dx[j];
          const double dnu = acc / mu;
          b.dls[lt * NF + f] = dnu;
          s.part[lane] += vpd * (mu * dnu - dd) - dd * dnu;
        }
      }
      SMPC_LANES_END_WAVE
      // ---- y = A dx + B du + f - mu p_{t+1}  (dense rows G; unit rows by structure) ----
      SMPC_LANES(NT)
      {
        // [K | k] is dead: the packed P~ takes its place ; the registers that held it start fetching stage t+1
#pragma unroll
        for (int n = FL::FULL_EARLY; n < PER; n++)
        {
          const int idx = lane + n * NT;
          if (idx >= FL::O_late && idx < FL::N_STAGE)
            s.st[idx - FL::O_late] = SMPC_PLV(pf)[n];
        }
        if (t + 1 < H)
          fetch(lane, SMPC_PLV(pf), t + 1, FL::FULL_EARLY, PER);
      }
      if (lane < NDX)
      {
        const int i = lane;
        double acc;
        if (IX::isG(i))
        {
          const double * Ar = &s.st[(i < 6 ? FL::O_A0 + i * NDX : FL::O_A1 + (i - NV) * NDX)];
          const double * Br = &s.st[(i < 6 ? FL::O_B0 + i * NU : FL::O_B1 + (i - NV) * NU)];
          acc = 0.0;
#pragma unroll 4
          for (int j = 0; j < NDX; j++)
            acc += Ar[j] * s.dx[j];
#pragma unroll 4
          for (int j = 0; j < NU; j++)
            acc += Br[j] * s.du[j];
        }
        else if (IX::isQj(i))
          acc = s.dx[i] + dt * s.dx[i + NV] + dt * dt * s.du[3 * NF + i - 6];
        else
          acc = s.dx[i] + dt * s.du[3 * NF + i - NV - 6];
        s.part[lane] += (vlx[lane] - s.lpd_prev[lane]) * s.dx[lane] + vlpd[lane] * acc;
        s.y[lane] = acc + vf[lane] - mu * pn[lane];
      }
      else if ((EXT && b.es != nullptr) && lane >= 48 && lane < 48 + 3 * NF)
      {
        // Multiplier steps of the (eliminated) friction-cone rows.  dnu = (D du + d) / mu would divide the difference of two O(1)
        // numbers by mu: every digit du lost to the 1 / mu entries of R^ costs |delta du| / mu in dnu.  The stationarity row of the
        // foot's force components holds the same dnu among O(1) terms:  D^T dnu = -(R du + S^T dx + r + B^T dlam)  (knot R, S, r:
        // unfolded).  Here: everything but B^T dlam for force component k.
        const int k = lane - 48;
        const double * lq = b.lq + lt * D::LQ_STRIDE;
        double acc = lq[D::O_r + k];
#pragma unroll 4
        for (int j = 0; j < NU; j++)
          acc += lq[D::r_off(k, j)] * s.du[j];
#pragma unroll 4
        for (int i = 0; i < NDX; i++)
          acc += lq[D::s_off(i, k)] * s.dx[i];
        SMPC_PLV(gpart) = acc;
      }
      SMPC_LANES_END_WAVE
      // ---- w = P~ y (P~ symmetric, upper triangle packed) ; dx+ = y - mu w ; dlam+ = w + p_{t+1} ----
      SMPC_LANES(NT)
      if (lane < NDX)
      {
        double w = 0.0;
#pragma unroll 4
        for (int j = 0; j < NDX; j++)
        {
          const int lo = j < lane ? j : lane, hi = j < lane ? lane : j;
          w += s.st[GK::pt_off(lo, hi) - GK::G_Pt] * s.y[j];
        }
        const double dxn = s.y[lane] - mu * w;
        const double dl = w + pn[lane];
        b.dxs[((size_t)inst * (H + 1) + t + 1) * NDX + lane] = dxn;
        b.dlams[lt * NDX + lane] = dl;
        s.part[lane] -= vf[lane] * dl;
        s.lpd_prev[lane] = vlpd[lane];
        s.dx[lane] = dxn;
        SMPC_PLV(dlr) = dl;
      }
      SMPC_LANES_END_WAVE
      if (EXT && b.es != nullptr)
      {
        SMPC_LANES(NT)
        if (lane < NDX)
          s.y[lane] = SMPC_PLV(dlr); // (y is dead: it becomes dlam)
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        if (lane >= 48 && lane < 48 + 3 * NF)
        {
          const int k = lane - 48;
          double acc = SMPC_PLV(gpart);
#pragma unroll
          for (int m = 0; m < 12; m++)
            acc += s.st[(m < 6 ? FL::O_B0 + m * NU : FL::O_B1 + (m - 6) * NU) + k] * s.y[IX::G(m)];
          s.du[k] = -acc; // (du is dead until the next stage: it takes D^T dnu of the force components)
        }
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        if (lane >= NDX && lane < NDX + 2 * NF)
        {
          const int i = lane - NDX, f = i / 2;
          const double dd = SMPC_PLV(ce)[3];
          const double g0 = s.du[3 * f], g1 = s.du[3 * f + 1], g2 = s.du[3 * f + 2];
          const double a0 = SMPC_PLV(ce)[0], a1 = SMPC_PLV(ce)[1], a2 = SMPC_PLV(ce)[2];
          const double p0 = SMPC_PLV(ce)[5], p1 = SMPC_PLV(ce)[6], p2 = SMPC_PLV(ce)[7];
          double dnu = dd / mu; // inactive row: D = 0
          if (SMPC_PLV(ce)[8] != 0.0)
          {
            const double aa = a0 * a0 + a1 * a1 + a2 * a2, ag = a0 * g0 + a1 * g1 + a2 * g2;
            dnu = ag / aa;
            if (SMPC_PLV(ce)[9] != 0.0)
            { // both rows of the foot active: 2 x 2 normal equations (parallel rows: the single-row value stands)
              const double pp = p0 * p0 + p1 * p1 + p2 * p2, ap = a0 * p0 + a1 * p1 + a2 * p2, pg = p0 * g0 + p1 * g1 + p2 * g2;
              const double det = aa * pp - ap * ap;
              if (det > 1e-12 * aa * pp)
                dnu = (pp * ag - ap * pg) / det;
            }
          }
          b.des[lt * 2 * NF + i] = dnu;
          s.part[lane] += SMPC_PLV(ce)[4] * (mu * dnu - dd) - dd * dnu;
        }
        SMPC_LANES_END_WAVE
      }
    }
    SMPC_LANES(NT)
    if (lane < NDX)
    {
      const int sl = ring_slot(ka.head, H - 1, R);
      const double lamH = b.lams[((size_t)inst * R + sl) * NDX + lane];
      const double lxN = b.qN[(size_t)inst * NDX + lane] + lamH;
      s.part[lane] += (lxN - s.lpd_prev[lane]) * s.dx[lane];
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double sacc = 0.0;
      for (int i = 0; i < 64; i++)
        sacc += s.part[i];
      b.scal[(size_t)inst * SC_N + SC_DPHI0] = sacc;
      b.ls_sel[inst] = -1;
    }
    SMPC_LANES_END_WAVE
  }

  // K_t for (instance, stage) blocks: grid = B * nt, 64 lanes (lane = column of K)
  template <class D>
  struct GainOutArgs
  {
    Buffers<D> b;
    int nt;       // stages per instance to export (1 = K0 only, H = all)
    double * out; // [B][nt][NU][NDX]
  };
  template <class D>
  SMPC_DEV void gains_out_body(const GainOutArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NDX = D::NDX, NU = D::NU;
    typedef GainsK<D> GK;
    const Buffers<D> & b = ka.b;
    const int inst = block / ka.nt, t = block % ka.nt;
    const double * g = b.gains + ((size_t)inst * b.H + t) * GK::STRIDE;
    double * out = ka.out + ((size_t)inst * ka.nt + t) * NU * NDX;
    SMPC_LANES(NT)
    if (lane < NDX)
    {
#pragma unroll 4
      for (int i = 0; i < NU; i++)
        out[i * NDX + lane] = g[GK::G_W + lane * NU + i];
    }
    SMPC_LANES_END_WAVE
  }

  // Return set of a control step, packed on the device: row [x1 (NX) | u0 (NU) | K0 (NU x NDX)] per instance, `row` doubles apart (what a
  // controller consumes).  One block of 64 lanes per instance.
  template <class D>
  struct PackOutArgs
  {
    Buffers<D> b;
    int s1, s0;        // ring slots of xs[1], us[0]
    int R;             // ring length
    int g_off, g_str;  // [K k] of stage 0 inside an instance's gains; doubles per (instance, stage)
    int g_tr;          // 0: rows of NDX + 1 (K(u, x) at u (NDX + 1) + x) ; 1: the structured sweep's transposed block (x NU + u)
    size_t row;
    double * out;
  };
  template <class D>
  SMPC_DEV void pack_outputs_body(const PackOutArgs<D> & pa, int inst)
  {
    constexpr int NT = 64;
    constexpr int NDX = D::NDX, NU = D::NU, NX = D::NX;
    const Buffers<D> & b = pa.b;
    const double * g = b.gains + (size_t)inst * b.H * pa.g_str + pa.g_off;
    const double * x = b.xs + ((size_t)inst * pa.R + pa.s1) * NX;
    const double * u = b.us + ((size_t)inst * pa.R + pa.s0) * NU;
    double * out = pa.out + (size_t)inst * pa.row;
    SMPC_LANES(NT)
    {
      for (int i = lane; i < NX; i += NT)
        out[i] = x[i];
      for (int i = lane; i < NU; i += NT)
        out[NX + i] = u[i];
      for (int i = lane; i < NU * NDX; i += NT)
        out[NX + NU + i] = pa.g_tr ? g[(i % NDX) * NU + i / NDX] : g[(i / NDX) * (NDX + 1) + i % NDX];
    }
    SMPC_LANES_END_WAVE
  }
} // namespace smpc
