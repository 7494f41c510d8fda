// smpc_riccati_dense.h -- proximal Riccati backward sweep for stages with DENSE A, B on the FP64 matrix cores
// (v_mfma_f64_16x16x4_f64), one 64-lane wavefront per instance: the full-dynamics OCP (HOT(4) of SolverProxDDP::run, reference
// src/mpc.cpp:212; LQ solver choice src/mpc.cpp:52).
//
// Same KKT system and the same two symmetric block sweeps as riccati_kino_body (smpc_riccati_kino.h), without the semi-implicit
// row structure (A and B of the constrained dynamics are dense):
//   (2)  P~, p~ = Schur complement of  [[I + mu P, sqrt(mu) P, sqrt(mu) pt0], [., P, pt0]]        (NDX pivots, NT1 x NT1 tiles)
//   (4)  T  = P~ [A | B]            -- accumulator tiles; the accumulator layout of T IS the B-operand layout of the next product
//        H^ = [Q S; S^T R] + [A | B]^T T,  vector column [q; r] + [A | B]^T p~ (p~ rides as one more column of T)
//   (5)  box rows (unit selectors): diagonal terms act / mu on R^ (torque box) and Q^ (joint box), act d / mu on the vector column
//   (6)  sweep of the control pivots of [[Q^, S^, q^], [S^^T, R^, r^]] in place  ->  P_t, p_t, -K, -k
// Dense constraint rows (the wrench-cone rows of 6-D feet: C and D both dense through the contact-force derivatives) are NOT
// folded into R^ (D^T D / mu would swamp the weakly determined directions, as in the centroidal kernel): their multipliers are
// pivoted explicitly after the controls,
//        [[Q^, S^, C^T, q^], [., R^, D^T, r^], [., ., -mu I, d]]   pivots [u | nu]  ->  P_t, p_t, -K, -k, -Z, -z
// (the stage KKT matrix is quasi-definite: LDL^T without pivoting is stable, 4 x 4 pivot blocks are definite of either sign).
// Generic in (NDX, NU, NCD): tile grids are computed from the dimensions; pivots are padded to whole panels with unit rows.  When the grid
// would exceed 128 columns (Talos with wrench cones and land rows: 56 + 24 + 48 + 1), the first dense rows take the padding pivots of the
// last control panel (a mixed panel: the 4 x 4 pivot block is factored L diag(d) L^T in pivot order, signs of d free).
#pragma once
#include "smpc_full_model.h"
#include "smpc_riccati_kino.h"
#include <type_traits>

namespace smpc
{
  template <class D>
  struct RiccatiDenseGeom
  {
    static constexpr int NDX = D::NDX, NU = D::NU, NXU = NDX + NU;
    static constexpr int NUP = ((NU + 3) / 4) * 4;     // control pivots padded to whole panels
    static constexpr int NXUP = NDX + NUP;
    static constexpr int NCD = D::NCD;                 // dense constraint rows (multipliers pivoted explicitly)
    static constexpr int FILL0 = (NUP - NU) < NCD ? (NUP - NU) : NCD;
    static constexpr int FILL = (NXUP + ((NCD + 3) / 4) * 4 + 1 > 128) ? FILL0 : 0; // dense rows in the padding pivots of the control panels
    static constexpr int NCP = ((NCD - FILL + 3) / 4) * 4;
    static constexpr int NXC = NXUP + NCP;              // [x | u (+ FILL dense rows) | nu]
    // column of dense row i / dense row of column c (-1: none)
    SMPC_HD static constexpr int dcol(int i) { return i < FILL ? NXU + i : NXUP + i - FILL; }
    SMPC_HD static constexpr int drow(int c) { return (c >= NXU && c < NXU + FILL) ? c - NXU : ((c >= NXUP && c < NXUP + NCD - FILL) ? c - NXUP + FILL : -1); }
    static constexpr int NT1 = (2 * NDX + 1 + 15) / 16; // tile grid of the first sweep
    static constexpr int NTX = (NDX + 15) / 16;         // tile rows of P~ / T
    static constexpr int NTJ = (NXUP + 15) / 16;        // tile columns holding [A | B]
    static constexpr int VC = NXC;                      // vector column of the second grid
    static constexpr int NT2 = (NXC + 1 + 15) / 16;     // tile grid of the second sweep
    static constexpr int VT = VC / 16;                  // tile column of the vector column
    static_assert(VT >= NTJ, "the vector column lives in a tile column of its own (or of the multipliers)");
    static constexpr int NTM = NT1 > NT2 ? NT1 : NT2;
    static_assert(NDX % 4 == 0, "pivot panels and K steps of 4");
    static_assert(NT1 <= 8 && NT2 <= 8, "sweep geometry: at most 128 columns");
  };

  template <class D>
  struct RiccatiDenseLds
  {
    typedef RiccatiDenseGeom<D> GM;
    static constexpr int NDX = D::NDX, NU = D::NU;
    static constexpr int SWP = 4 * 16 * GM::NTM;
    static constexpr int N_SCR = 2 * SWP;
    // rows 0 .. NDX-1: P_{t+1} -> P~ -> P_t (full symmetric image); row NDX: c = p_{t+1} - f / mu; row NDX+1: p_{t+1} -> p_t -- the vector column of
    // the first sweep's bordered matrix is read through the same (column, row) addresses as the matrix entries (smpc_riccati_kino.h, round 6)
    double P[(NDX + 2) * NDX];
    double scr[N_SCR];        // sweep operands (2 x 4 x 16 NT)
    // ([A | B] is not staged: its 16 x 4 operand slices are read from the knot twice, coalesced along the rows, one K-step ahead
    //  of the products that consume them -- 36 KB of LDS per wave are worth more as resident waves)
    double pt[NDX];
    double boxa[D::NU + D::NA], boxd[D::NU + D::NA + GM::NCD]; // activity of the box rows ; d = mu (nu+ - nu) of all rows
    double cact[GM::NCP > 0 ? GM::NCP : 1];                    // activity of the dense rows (padded to whole panels)
  };

  template <class D>
  SMPC_DEV void riccati_dense_body(const SolverArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    typedef RiccatiDenseGeom<D> GM;
    typedef RiccatiDenseLds<D> LDS;
    constexpr int NDX = D::NDX, NU = D::NU, NA = D::NA, NXU = GM::NXU, NXUP = GM::NXUP, NT1 = GM::NT1, NTX = GM::NTX, NTJ = GM::NTJ, NT2 = GM::NT2,
                  VC = GM::VC, NUP = GM::NUP, NCD = GM::NCD, NCP = GM::NCP, NXC = GM::NXC, VT = GM::VT;
    const Buffers<D> & b = ka.b;
    const int H = b.H;
    const int inst = block;
    const double mu = b.model->mu, imu = 1.0 / mu, smu = sqrt(mu);
    SMPC_LDS(LDS, lds, 1);
    LDS & s = lds[0];
    double * sw = s.scr;
    SMPC_LANES(NT)
    {
      for (int i = lane; i < NDX * NDX; i += NT)
        s.P[i] = b.QN[(size_t)inst * NDX * NDX + i];
      for (int i = lane; i < NDX; i += NT)
        s.P[(NDX + 1) * NDX + i] = b.qN[(size_t)inst * NDX + i];
    }
    SMPC_LANES_END_WAVE
    static_assert(NDX <= NT, "one entry of f / p per lane");
    SMPC_PL(double, f_pf, NT); // f of the stage, fetched one stage ahead
    SMPC_LANES(NT)
    SMPC_PLV(f_pf) = b.lq[((size_t)inst * H + (H - 1)) * D::LQ_STRIDE + D::O_f + (lane < NDX ? lane : 0)];
    SMPC_LANES_END_WAVE
    double * prof = (b.dbg != nullptr && block == 0) ? b.dbg : nullptr; // optional phase timers (block 0 only): slots 40 .. 46
    long long tprev = prof ? SMPC_CLOCK() : 0;
    // address bases of the first sweep's LDS staging, formed once for all stages (wave_block_sweep; the second sweep has two grid widths and the
    // biped's instantiation no registers to spare: it computes its own)
    SweepBases sb1;
    sweep_bases_init<16 * NT1>(sb1);
    for (int t = H - 1; t >= 0; t--)
    {
      const double * lq = b.lq + ((size_t)inst * H + t) * D::LQ_STRIDE;
      double * g = b.gains + ((size_t)inst * H + t) * D::G_STRIDE;
      // pivot panels of the second sweep that hold inactive dense rows only: such a row is [0 ... -mu ... d] and couples to
      // nothing, so its panel is left out of the sweep and z = d / mu is written directly.  (The flags are fetched by the lanes of
      // phase (1) with their other loads; the mask is formed from LDS afterwards.)
      unsigned skip = 0u;
      // ---- (1) box rows ; save p_{t+1} ; c = p - f / mu (the vector column of the pivot rows of (2)) ; f of the next stage ----
      SMPC_LANES(NT)
      {
        if (lane < NDX)
        {
          const double pv = s.P[(NDX + 1) * NDX + lane];
          g[D::G_pn + lane] = pv;
          s.P[NDX * NDX + lane] = pv - imu * SMPC_PLV(f_pf);
        }
        if (t > 0)
          SMPC_PLV(f_pf) = lq[-(int)D::LQ_STRIDE + D::O_f + (lane < NDX ? lane : 0)];
        for (int i = lane; i < NU + NA; i += NT)
          s.boxa[i] = lq[D::O_act + i];
        for (int i = lane; i < NU + NA + NCD; i += NT)
          s.boxd[i] = lq[D::O_d + i];
        if constexpr (NCD > 0)
          for (int i = lane; i < GM::NCP; i += NT) // (rows behind the control panels; the FILL rows in a control panel are never skipped)
            s.cact[i] = i + GM::FILL < NCD ? lq[D::O_act + NU + NA + GM::FILL + i] : 0.0;
      }
      SMPC_LANES_END_WAVE
      if constexpr (NCD > 0)
      {
        unsigned m = 0u;
        SMPC_LANES(NT)
        {
          unsigned mm = 0u;
          for (int q = 0; q < GM::NCP / 4; q++)
            if (s.cact[4 * q] + s.cact[4 * q + 1] + s.cact[4 * q + 2] + s.cact[4 * q + 3] == 0.0)
              mm |= 1u << (GM::NUP / 4 + q);
          m = SMPC_UNIFORM_U32(mm); // (every lane computes the same value)
        }
        SMPC_LANES_END_WAVE
        skip = m;
      }
      prof_tick(prof, 40, tprev);
      // ---- (2) P~ and p~: Schur complement of the leading NDX pivots of  [[P + I / mu, P, c], [P, P, p]],  c = p - f / mu:
      //          P - P (P + I/mu)^-1 P = (I + mu P)^-1 P = P~,   p - mu P~ c = p + P~ (f - mu p) = p~   (the same elimination as on
      //          [[I + mu P, sqrt(mu) P], [., P]] with the pivot rows / columns scaled by 1 / sqrt(mu); no scaling, no P f product).
      //          Element (R, C) of a tile <- image[C' * NDX + R']: compile-time offsets from four per-lane bases (smpc_riccati_kino.h) ----
      {
        SMPC_ACC(t1, NT, NT1 * (NT1 + 1) / 2);
        constexpr int JW = NDX / 16, LW = NDX % 16;             // tile column holding column NDX, its first lane there
        constexpr int JV = (2 * NDX) / 16, LV = (2 * NDX) % 16; // tile column / lane of the vector column
        static_assert(JV < NT1 && (LW == 0 || JW != JV), "geometry of the bordered matrix");
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
          const int bA = lc * NDX + lr;
          const int bW = (lc < LW ? 16 * JW + lc : lc - LW) * NDX + lr;                       // tile column JW: C = 16 JW + lc wraps at NDX
          const int bVp = (lc < LV ? 16 * JV - NDX + lc : (lc == LV ? NDX : NDX + 1)) * NDX + lr; // tile column JV, pivot rows: ..., c, (padding: p)
          const int bVs = (lc < LV ? 16 * JV - NDX + lc : NDX + 1) * NDX + lr;                  // tile column JV, other rows: ..., p, (padding: p)
#pragma unroll
          for (int I = 0; I < NT1; I++)
#pragma unroll
            for (int J = I; J < NT1; J++)
#pragma unroll
              for (int v = 0; v < 4; v++)
              {
                const int R0 = 16 * I + 4 * v; // (compile-time: NDX is a multiple of 4)
                double val = 0.0;
                if (R0 < 2 * NDX)
                {
                  const bool piv = R0 < NDX;
                  const int Rp = piv ? R0 : R0 - NDX;
                  const int base = J == JV ? (piv ? bVp : bVs) : ((LW != 0 && J == JW) ? bW : (16 * J + 15 < NDX ? bA + 16 * J * NDX : bA + (16 * J - NDX) * NDX));
                  val = s.P[base + Rp];
                  if (piv && I == J && lc == lr + 4 * v)
                    val += imu;
                }
                SMPC_ACCV(t1, tix<NT1>(I, J), v) = val;
              }
        }
        SMPC_LANES_END_WAVE
        wave_block_sweep<NT, NT1, false, 0, NDX / 4>(t1, sw, sw + LDS::SWP, prof, tprev, 0u, &sb1);
        // P~ (rows / columns NDX .. 2 NDX of the grid) -> LDS image, both halves, upper entries of a diagonal tile only; p~ -> pt.  Stores grouped
        // by execution mask (a predicate per store costs four scalar instructions and a branch)
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
          const int bR = lr * NDX + lc, bT = lc * NDX + lr;
          auto put = [&](int I, int J, int v) SMPC_LAMBDA_INLINE {
            const int Rp = 16 * I + 4 * v - NDX, Cp = 16 * J - NDX;
            const double val = SMPC_ACCV(t1, tix<NT1>(I, J), v);
            s.P[bR + Rp * NDX + Cp] = val;
            s.P[bT + Cp * NDX + Rp] = val;
          };
          auto rows = [](int I, int v) { return 16 * I + 4 * v >= NDX && 16 * I + 4 * v < 2 * NDX; };
          // diagonal tiles: lanes lc >= lr + 4 v (and inside the block's columns)
#pragma unroll
          for (int v = 0; v < 4; v++)
            if (lc >= lr + 4 * v)
            {
#pragma unroll
              for (int I = JW; I <= JV; I++)
                if (rows(I, v))
                {
                  if (I == JV)
                  {
                    if (lc < LV)
                      put(I, I, v);
                  }
                  else if (I == JW && LW != 0)
                  {
                    if (lc >= LW)
                      put(I, I, v);
                  }
                  else
                    put(I, I, v);
                }
            }
          // off-diagonal tiles: all lanes, but for the tile column of the vector column
#pragma unroll
          for (int I = JW; I < JV; I++)
#pragma unroll
            for (int J = I + 1; J < JV; J++)
#pragma unroll
              for (int v = 0; v < 4; v++)
                if (rows(I, v))
                  put(I, J, v);
          if (lc < LV)
          {
#pragma unroll
            for (int I = JW; I < JV; I++)
#pragma unroll
              for (int v = 0; v < 4; v++)
                if (rows(I, v))
                  put(I, JV, v);
          }
          if (lc == LV)
          {
#pragma unroll
            for (int I = JW; I <= JV; I++)
#pragma unroll
              for (int v = 0; v < 4; v++)
                if (rows(I, v))
                  s.pt[16 * I + 4 * v - NDX + lr] = SMPC_ACCV(t1, tix<NT1>(I, JV), v);
          }
        }
        SMPC_LANES_END_WAVE
      }
      prof_tick(prof, 41, tprev);
      // Stages without an active dense row (the common case: wrench cones of feet that stand flat) run the LIGHT form of what follows: the
      // multiplier columns are left out of the second grid altogether (NCPX = 0: 6 instead of 8 tile columns for the Talos-class problems,
      // 21 instead of 36 accumulator tiles -- no spilled accumulators, 40 % fewer rank-4 updates in the control sweep); Z = 0, z = d / mu.
      auto tail = [&](auto ncpx_) SMPC_LAMBDA_INLINE {
        constexpr int NCPX = decltype(ncpx_)::value;                 // padded dense rows behind the control panels kept in the grid
        constexpr int NXCX = NCPX == 0 ? 16 * NTJ : NXUP + NCPX, VCX = NXCX, NT2X = (NXCX + 1 + 15) / 16, VTX = VCX / 16; // (light: the vector column opens a tile column of its own)
        static_assert(VTX >= NTJ, "the vector column lives in a tile column of its own (or of the multipliers)");
        auto drowx = [](int c) -> int { return (NCPX == 0 && c >= GM::NXUP) ? -1 : GM::drow(c); };
        // ---- (3) [A | B] into LDS ; P~ out (forward sweep) ; prefetch of [Q S q; S^T R r] in accumulator layout ----
        SMPC_ACC(hacc, NT, NT2X * (NT2X + 1) / 2);
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
  #pragma unroll
          for (int I = 0; I < NT2X; I++)
  #pragma unroll
            for (int J = I; J < NT2X; J++)
  #pragma unroll
              for (int v = 0; v < 4; v++)
              {
                const int row = 16 * I + lr + 4 * v, col = 16 * J + lc;
                // upper storage: row <= col inside a diagonal tile is not guaranteed: take (min, max)
                const int r0 = row < col ? row : col, c0 = row < col ? col : row;
                int off = D::O_Q; // entries outside the problem load Q[0][0] and are overwritten below
                if (c0 < NDX)
                  off = D::O_Q + r0 * NDX + c0;
                else if (c0 < NXU)
                  off = r0 < NDX ? D::O_S + r0 * NU + c0 - NDX : D::O_R + (r0 - NDX) * NU + c0 - NDX;
                else if (drowx(c0) >= 0 && r0 < NXU) // dense constraint rows: (x, nu_i) = C_i, (u, nu_i) = D_i
                  off = r0 < NDX ? D::O_C + drowx(c0) * NDX + r0 : D::O_D + drowx(c0) * NU + r0 - NDX;
                else if (c0 == VCX && r0 < NXU)
                  off = r0 < NDX ? D::O_q + r0 : D::O_r + r0 - NDX;
                SMPC_ACCV(hacc, tix<NT2X>(I, J), v) = lq[off];
              }
          if constexpr (D::PT_PACKED)
          {
            // upper triangle: row i (NDX - i entries from the diagonal on) and row NDX - 1 - i (i + 1 entries) share a pass of the lanes
            // (NDX + 1 <= 64 entries): half as many stores as one row per pass
            static_assert(NDX + 1 <= NT && NDX % 2 == 0, "two rows of P~ per pass of the lanes");
#pragma unroll
            for (int i = 0; i < NDX / 2; i++)
            {
              const int i2 = NDX - 1 - i, n1 = NDX - i;
              const bool first = lane < n1;
              const int src = first ? i * NDX + i + lane : i2 * NDX + i2 + (lane - n1);
              const int dst = first ? D::pt_row(i) + i + lane : D::pt_row(i2) + i2 + (lane - n1);
              if (lane < NDX + 1)
                g[dst] = s.P[src];
            }
          }
          else
            for (int idx = lane; idx < NDX * NDX; idx += NT)
              g[D::G_Pt + idx] = s.P[idx];
        }
        SMPC_LANES_END_WAVE
        prof_tick(prof, 42, tprev);
        // ---- (4) T = P~ [A | B] with p~ as column VCX ; H^ += [A | B]^T T ----
        {
          constexpr int NTT = NTJ + 1; // tile columns of T: [A | B] columns + the tile of the vector column
          SMPC_ACC(tacc, NT, NTX * NTT);
          SMPC_PLA(double, pav, NT, NTX);
          SMPC_PLA(double, abv, NT, NTJ);
          // operand slices of the next TWO K-steps, in flight while the current products run: a K-step of MFMAs (1.0 - 1.5 k cycles) is shorter
          // than a round trip to L2 / HBM (2 - 5 k), so with one slice ahead every K-step waited for its operands
          // (measured per instantiation: Go2 full dynamics 3.95 -> 3.82 ms, Talos kinodynamics 14.2 -> 12.3 ms per launch; the Talos full-dynamics
          //  sweep -- 56 states, 22 controls, up to 78 rows -- is at its register limit, the second slice is spilled there and costs 6 %: one ahead)
          constexpr bool TWO_AHEAD = !(NDX == 56 && NU == 22);
          SMPC_PLA(double, abn, NT, NTJ);
          SMPC_PLA(double, abn2, NT, (TWO_AHEAD ? NTJ : 1));
          // slice ks of [A | B | 0]: entry (4 ks + lr, 16 J + lc), address selected, loaded once, masked at the use
          auto ab_fetch = [&](int ks, int J, int lr, int lc) {
            const int r = 4 * ks + lr, c = 16 * J + lc;
            const double * src = c < NDX ? lq + D::O_A + r * NDX + c : lq + D::O_B + r * NU + (c < NXU ? c - NDX : 0);
            return *src;
          };
          SMPC_LANES(NT)
          {
            const int lr = lane >> 4, lc = lane & 15;
  #pragma unroll
            for (int J = 0; J < NTJ; J++)
            {
              SMPC_PLV(abn)[J] = ab_fetch(0, J, lr, lc);
              if constexpr (TWO_AHEAD)
                SMPC_PLV(abn2)[J] = ab_fetch(1, J, lr, lc);
            }
  #pragma unroll
            for (int I = 0; I < NTX; I++)
  #pragma unroll
              for (int J = 0; J < NTT; J++)
  #pragma unroll
                for (int v = 0; v < 4; v++)
                {
                  const int row = 16 * I + lr + 4 * v, col = 16 * (J < NTJ ? J : VTX) + lc;
                  const double pv = s.pt[row < NDX ? row : 0];
                  SMPC_ACCV(tacc, I * NTT + J, v) = (col == VCX && row < NDX) ? pv : 0.0;
                }
          }
          SMPC_LANES_END_WAVE
          for (int ks = 0; ks < NDX / 4; ks++)
          {
            SMPC_LANES(NT)
            {
              const int lr = lane >> 4, lc = lane & 15;
  #pragma unroll
              for (int I = 0; I < NTX; I++)
              {
                // A operand: P~[16 I + lc][4 ks + lr] (symmetric: read along the row of 4 ks + lr)
                const int r = 16 * I + lc;
                const double pv = s.P[(4 * ks + lr) * NDX + (r < NDX ? r : 0)];
                SMPC_PLV(pav)[I] = r < NDX ? pv : 0.0;
              }
  #pragma unroll
              for (int J = 0; J < NTJ; J++)
              {
                SMPC_PLV(abv)[J] = 16 * J + lc < NXU ? SMPC_PLV(abn)[J] : 0.0;
                // slice after next (the second product starts again at slice 0)
                if constexpr (TWO_AHEAD)
                {
                  SMPC_PLV(abn)[J] = SMPC_PLV(abn2)[J];
                  SMPC_PLV(abn2)[J] = ab_fetch((ks + 2) % (NDX / 4), J, lr, lc);
                }
                else
                  SMPC_PLV(abn)[J] = ab_fetch(ks + 1 < NDX / 4 ? ks + 1 : 0, J, lr, lc);
              }
            }
            SMPC_LANES_END_WAVE
  #pragma unroll
            for (int I = 0; I < NTX; I++)
  #pragma unroll
              for (int J = 0; J < NTJ; J++)
                SMPC_MFMA(tacc, I * NTT + J, pav, I, abv, J);
          }
          // H^(I, J) += sum_ks ABop(ks, I)^T T(ks, J): the B operand of K-step ks = 4 Ix + v is accumulator entry v of T's tile (Ix, J)
          SMPC_PLA(double, tbv, NT, NTT);
  #pragma unroll
          for (int Ix = 0; Ix < NTX; Ix++)
  #pragma unroll
            for (int v = 0; v < 4; v++)
            {
              if (16 * Ix + 4 * v >= NDX)
                continue;
              SMPC_LANES(NT)
              {
                const int lr = lane >> 4, lc = lane & 15;
                const int ks = 4 * Ix + v;
  #pragma unroll
                for (int J = 0; J < NTJ; J++)
                {
                  SMPC_PLV(abv)[J] = 16 * J + lc < NXU ? SMPC_PLV(abn)[J] : 0.0;
                  if constexpr (TWO_AHEAD)
                  {
                    SMPC_PLV(abn)[J] = SMPC_PLV(abn2)[J];
                    if (ks + 2 < NDX / 4)
                      SMPC_PLV(abn2)[J] = ab_fetch(ks + 2, J, lr, lc);
                  }
                  else if (ks + 1 < NDX / 4)
                    SMPC_PLV(abn)[J] = ab_fetch(ks + 1, J, lr, lc);
                }
  #pragma unroll
                for (int J = 0; J < NTT; J++)
                  SMPC_PLV(tbv)[J] = SMPC_ACCV(tacc, Ix * NTT + J, v);
              }
              SMPC_LANES_END_WAVE
  #pragma unroll
              for (int I = 0; I < NTJ; I++)
  #pragma unroll
                for (int J = I; J < NT2X; J++)
                  if (J < NTJ || J == VTX) // (T is zero in the multiplier columns)
                    SMPC_MFMA(hacc, tix<NT2X>(I, J), abv, I, tbv, J < NTJ ? J : NTJ);
            }
        }
        prof_tick(prof, 43, tprev);
        // ---- (5) box rows ; padding pivots ; entries outside the problem ----
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
  #pragma unroll
          for (int I = 0; I < NT2X; I++)
  #pragma unroll
            for (int J = I; J < NT2X; J++)
  #pragma unroll
              for (int v = 0; v < 4; v++)
              {
                const int row = 16 * I + lr + 4 * v, col = 16 * J + lc;
                const int r0 = row < col ? row : col, c0 = row < col ? col : row;
                double val = SMPC_ACCV(hacc, tix<NT2X>(I, J), v);
                // box index of a (state / control) index: joint box on x[6 .. 6 + NA), torque box on u
                const int bi = (r0 >= 6 && r0 < 6 + NA) ? NU + r0 - 6 : ((r0 >= NDX && r0 < NXU) ? r0 - NDX : -1);
                const double ba = s.boxa[bi >= 0 ? bi : 0], bd = s.boxd[bi >= 0 ? bi : 0];
                if (bi >= 0 && c0 == r0)
                  val += imu * ba;
                if (bi >= 0 && c0 == VCX)
                  val += imu * ba * bd;
                const bool rx = r0 < NXU, rn = drowx(r0) >= 0; // problem rows: (x, u) ; explicit multipliers
                const bool cx = c0 < NXU, cn = drowx(c0) >= 0;
                if (rn && c0 == r0)
                  val = -mu;                                   // multiplier block -mu I
                else if (rn && c0 == VCX)
                  val = s.boxd[NU + NA + drowx(r0)];         // d of the dense rows
                else if (!((rx && (cx || cn || c0 == VCX))))
                  val = (r0 == c0 && r0 < NXCX) ? 1.0 : 0.0;     // unit padding pivots, zero elsewhere
                SMPC_ACCV(hacc, tix<NT2X>(I, J), v) = val;
              }
        }
        SMPC_LANES_END_WAVE
        prof_tick(prof, 44, tprev);
        // ---- (6) sweep the control pivots in place:  x-x block -> P_t,  x-vector -> p_t,  (x, u) entries -> -K,  (u, vector) -> -k ----
        wave_block_sweep<NT, NT2X, true, NDX, (NUP + NCPX) / 4, (NCPX > 0)>(hacc, sw, sw + LDS::SWP, prof, tprev, skip);
        prof_tick(prof, 45, tprev);
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
  #pragma unroll
          for (int I = 0; I < NT2X; I++)
  #pragma unroll
            for (int J = I; J < NT2X; J++)
  #pragma unroll
              for (int v = 0; v < 4; v++)
              {
                const int row = 16 * I + lr + 4 * v, col = 16 * J + lc;
                const double val = SMPC_ACCV(hacc, tix<NT2X>(I, J), v);
                if (row < NDX)
                {
                  if (col < NDX)
                  {
                    if (row <= col)
                    {
                      s.P[row * NDX + col] = val;
                      s.P[col * NDX + row] = val;
                    }
                  }
                  else if (col < NXU)
                    g[D::G_K + (col - NDX) * (NDX + 1) + row] = -val; // K
                  else if (drowx(col) >= 0)
                    g[D::G_Z + drowx(col) * (NDX + 1) + row] = -val; // Z: multiplier feedback of the dense rows
                  else if (col == VCX)
                    s.P[(NDX + 1) * NDX + row] = val; // p_t
                }
                else if (row < NXU && col == VCX)
                  g[D::G_K + (row - NDX) * (NDX + 1) + NDX] = -val; // k
                else if (drowx(row) >= 0 && col == VCX)
                  g[D::G_Z + drowx(row) * (NDX + 1) + NDX] = ((skip >> ((row - NDX) / 4)) & 1u) ? val / mu : -val; // z
              }
        }
        SMPC_LANES_END_WAVE

        // (light grid: the dense rows that are not in the grid are all inactive -- Z = 0, z = d / mu.  Nothing is written for them: the forward
        //  sweeps form d / mu themselves for a row whose activity flag is 0 and never read its [Z z] -- round 5: 15.5 of the biped's 51 KB of
        //  gains per stage neither written nor read in the stages without an active cone row)
      };
      if constexpr (NCP > 0)
      {
        if (skip == (((1u << (NCP / 4)) - 1u) << (NUP / 4)))
          tail(std::integral_constant<int, 0>());
        else
          tail(std::integral_constant<int, NCP>());
      }
      else
        tail(std::integral_constant<int, 0>());
      prof_tick(prof, 46, tprev);
    }
  }
} // namespace smpc
