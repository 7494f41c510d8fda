// smpc_cent_split.h -- the centroidal OCP with point feet (BASELINE config "Go2 centroidal (9-dim state), H = 50") as a pipeline of
// kernels per ProxDDP iteration (round 5; the one-kernel form of smpc_cent_kernels.h stays as the cross-check, SMPC_CENT_FUSED=1).
// Reference: MPC::iterate (src/mpc.cpp:189-218) over a CentroidalOCP (src/centroidal-dynamics.cpp:39-106, 306-316).
//
// Why: cent_step_body is ONE dependent chain per instance (one wavefront, ~32 k cycles per stage and iteration), and 52 % of that chain is
// work that is local to a stage -- B x H independent point-mass problems.  Here that part runs one problem per LANE, and the serial part
// keeps its matrices in the accumulator registers of the FP64 matrix cores from one stage to the next:
//
//   cent_recede_body   grid B          ring advance, warm-start shift, Raibert foothold + Bezier swing references (src/mpc.cpp:201-207,278-309)
//   k x { cent_pre_body    grid B      lane = stage: evaluation, derivatives, stage cost / constraint blocks of the KKT matrix, merit terms;
//                                      hand-over = one contiguous 2 KB record per (instance, stage), written by a wave-wide transposition
//         cent_bwd_body    grid B      proximal Riccati recursion t = H-1 .. 0, one wavefront per instance.  P_{t+1} lives in ONE accumulator
//                                      tile; both bordered matrices are assembled in accumulator layout (no LDS image of them): sweep 1 from
//                                      the tile of P, the stage matrix by 18 gathered LDS reads of the record; P~ is the A operand of
//                                      P~ [A B] as it stands (symmetric: accumulator layout = operand layout), p~ = p + P~ (f - mu p) is three
//                                      more MFMAs whose result column is already where the second product needs it
//         cent_fwd_body    grid B      forward sweep: three matrix-vector products per stage fed by v_readlane broadcasts (no LDS)
//         cent_ls_body     grid B      line search (lane = stage), accept, regularisation }
//
// Index space of the stage KKT matrix (two 16-wide tile rows):  u 0..11 | nu_a 12..15 | x 16..24 | vector column 25 | 26, 27 unused |
// nu_b 28..31 (cone rows 0..3 -> nu_a, 4..7 -> nu_b).  The state block starts ON a tile boundary, so the Schur complement P_t of sweep 2
// (tile (1, 1), p_t in its column 9) IS the tile sweep 1 of the next stage starts from.
#pragma once
#include "smpc_cent_kernels.h"
#include "smpc_kino_lane.h"

namespace smpc
{
  // ---- hand-over record of a stage (doubles; the order is the order cent_pre_body produces them in) ----
  template <class D>
  struct CentRec
  {
    static constexpr int NF = D::NF, NU = D::NU, NC = D::NC;
    static_assert(NC <= 8 && NU <= 12, "cone rows live in the 4 + 4 spare slots of the two tile rows");
    // quarters 0, 1 (only the backward sweep reads them): the Hessian blocks
    static constexpr int O_RR = 0;                  // R = Luu + preg I as 3 x 3 foot blocks (fa <= fb; diagonal blocks: upper triangle)
    static constexpr int N_RR = 6 * NF + 9 * (NF * (NF - 1) / 2);
    static constexpr int O_QC = O_RR + N_RR;        // 6      CoM block of Q = Lxx + preg I, upper triangle
    static constexpr int O_S = O_QC + 6;            // 9 NF   Lxu of the CoM rows: [f][xi][jb]
    static constexpr int N_HESS = O_S + 9 * NF;
    // quarters 2, 3 (forward sweep AND backward sweep): [A B] entries, vectors, cone rows
    static constexpr int O_DTACT = ((N_HESS + 63) / 64) * 64; // NF dt act_f
    static constexpr int O_DTRP = O_DTACT + NF;     // 3 NF   +dt r_f   (r_f = 0 for a foot in the air)
    static constexpr int O_DTRN = O_DTRP + 3 * NF;  // 3 NF   -dt r_f
    static constexpr int O_DTFP = O_DTRN + 3 * NF;  // 3      +dt sum of the active forces
    static constexpr int O_DTFN = O_DTFP + 3;       // 3      -dt ...
    static constexpr int O_F = O_DTFN + 3;          // 9      mu (lam+ - lam)
    static constexpr int O_LPD = O_F + 9;           // 9      2 lam+ - lam
    static constexpr int O_Q = O_LPD + 9;           // 9      q
    static constexpr int O_GX = O_Q + 9;            // 9      lx + A^T (2 lam+ - lam)
    // per foot f (the pre-pass finishes one foot at a time): Jacobian rows of its two cone rows when ACTIVE (0 otherwise) 6, mu (nu+ - nu) 2,
    // r 3, gu = lu + B^T (2 lam+ - lam) + Cu^T vpd 3
    static constexpr int O_FT = O_GX + 9, FT_N = 14;
    static constexpr int O_ANY = O_FT + FT_N * NF;  // 1      any cone row active
    static constexpr int N = O_ANY + 1;
    static constexpr int Q_FWD = O_DTACT / 64;      // first quarter the forward sweep reads
    SMPC_HD static constexpr int d_off(int row, int k) { return O_FT + (row / 2) * FT_N + (row % 2) * 3 + k; }
    SMPC_HD static constexpr int dv_off(int row) { return O_FT + (row / 2) * FT_N + 6 + row % 2; }
    SMPC_HD static constexpr int r_off(int j) { return O_FT + (j / 3) * FT_N + 8 + j % 3; }
    SMPC_HD static constexpr int gu_off(int j) { return O_FT + (j / 3) * FT_N + 11 + j % 3; }
    static constexpr int STRIDE = ((N + 63) / 64) * 64;
    static constexpr int NLOAD = STRIDE / 64;
    static_assert(STRIDE % EV_CH == 0, "whole flushes");
    // gains of a stage, backward -> forward sweep: [K k] NU x 10 (rows of stride D::GKS: the read-out kernels of the handle use the same place),
    // P~ as packed upper triangle, p+, and -- read only when a cone row of the stage is active -- [Z z] NC x 10
    static constexpr int G_K = 0, G_Pt = NU * 10, G_pn = G_Pt + 45, G_Z = ((G_pn + 9 + 63) / 64) * 64, G_N = G_Z + NC * 10;
    static_assert(G_N <= D::G_STRIDE && G_K == D::G_K && D::GKS == 10, "gains block of the handle");
    SMPC_HD static constexpr int pt_off(int i, int j) { return i <= j ? G_Pt + i * 9 - i * (i - 1) / 2 + (j - i) : G_Pt + j * 9 - j * (j - 1) / 2 + (i - j); }
    // constants the gather of cent_bwd_body addresses behind the record in LDS
    static constexpr int C_ZERO = STRIDE, C_ONE = STRIDE + 1, C_DTM = STRIDE + 2, C_NMU = STRIDE + 3, C_QLM = STRIDE + 4, C_QAM = STRIDE + 10;
    static constexpr int STAGE_N = STRIDE + 16;
    // terminal node (own small block per instance): P_H 81, p_H 9, lx_N 9
    static constexpr int T_P = 0, T_p = 81, T_lx = 90, T_STRIDE = 104;

    SMPC_HD static constexpr int tri3(int a, int b) { return a * 3 - a * (a - 1) / 2 + (b - a); } // a <= b < 3
    SMPC_HD static constexpr int rr_base(int fa, int fb)
    {
      int o = 0;
      for (int a = 0; a < fa; a++)
        o += 6 + 9 * (NF - 1 - a);
      if (fb > fa)
        o += 6 + 9 * (fb - fa - 1);
      return o;
    }
    SMPC_HD static constexpr int rr_off(int i, int j) // i <= j
    {
      const int fa = i / 3, fb = j / 3, ia = i % 3, jb = j % 3;
      return O_RR + rr_base(fa, fb) + (fa == fb ? tri3(ia, jb) : ia * 3 + jb);
    }
    // index space of the stage KKT matrix
    static constexpr int XO = 16, ZC = 25;
    SMPC_HD static constexpr int cone_slot(int row) { return row < 4 ? 12 + row : 24 + row; }
    // entry (i, j) of the stage KKT matrix [[R, D^T, S^T, r], [D, -mu I, 0, d], [S, 0, Q, q]] in that index space -> offset in the LDS stage
    SMPC_HD static constexpr int m2_off(int i, int j)
    {
      if (i > j)
      {
        const int t = i;
        i = j;
        j = t;
      }
      // classes: 0 u, 1 nu, 2 x, 3 vector column, 4 unused
      const int ci = i < 12 ? 0 : (i < 16 ? 1 : (i < 25 ? 2 : (i == 25 ? 3 : (i < 28 ? 4 : 1))));
      const int cj = j < 12 ? 0 : (j < 16 ? 1 : (j < 25 ? 2 : (j == 25 ? 3 : (j < 28 ? 4 : 1))));
      const int ri = i < 16 ? i - 12 : i - 24, rj = j < 16 ? j - 12 : j - 24; // cone row of a nu index
      if (ci == 4 || cj == 4)
        return C_ZERO;
      if (ci == 0 && cj == 0)
        return (i < NU && j < NU) ? rr_off(i, j) : (i == j ? C_ONE : C_ZERO);
      if (ci == 0 && cj == 1)
        return (i < NU && rj < NC && i / 3 == rj / 2) ? d_off(rj, i % 3) : C_ZERO;
      if (ci == 0 && cj == 2)
        return (i < NU && j - XO < 3) ? O_S + (i / 3) * 9 + (j - XO) * 3 + i % 3 : C_ZERO;
      if (ci == 0 && cj == 3)
        return i < NU ? r_off(i) : C_ZERO;
      if (ci == 1 && cj == 1)
        return i == j ? C_NMU : C_ZERO;
      if (ci == 1 && cj == 2)
        return C_ZERO;
      if (ci == 2 && cj == 1) // (x index below a nu_b index)
        return C_ZERO;
      if (ci == 1 && cj == 3)
        return ri < NC ? dv_off(ri) : C_ZERO;
      if (ci == 3 && cj == 1)
        return rj < NC ? dv_off(rj) : C_ZERO;
      if (ci == 2 && cj == 2)
      {
        const int a = i - XO, b = j - XO;
        if (a / 3 != b / 3)
          return C_ZERO;
        return (a < 3 ? O_QC : (a < 6 ? C_QLM : C_QAM)) + tri3(a % 3, b % 3);
      }
      if (ci == 2 && cj == 3)
        return O_Q + (i - XO);
      return C_ZERO;
    }
    // [v]x (a, b) = sign * v[comp]: offset of +-dt v[comp] given the bases of +dt v and -dt v
    SMPC_HD static constexpr int skew_off(int a, int b, int pbase, int nbase)
    {
      if (a == b)
        return C_ZERO;
      const int comp = 3 - a - b;
      const bool pos = (a == 0 && b == 2) || (a == 1 && b == 0) || (a == 2 && b == 1);
      return (pos ? pbase : nbase) + comp;
    }
    // entry (k, col) of [A B] (row k of x+, column in the index space above)
    SMPC_HD static constexpr int ab_off(int k, int col)
    {
      if (k >= 9)
        return C_ZERO;
      if (col < NU)
      {
        const int f = col / 3, jj = col % 3;
        if (k >= 3 && k < 6)
          return k - 3 == jj ? O_DTACT + f : C_ZERO;
        if (k >= 6)
          return skew_off(k - 6, jj, O_DTRP + 3 * f, O_DTRN + 3 * f);
        return C_ZERO;
      }
      if (col >= XO && col < XO + 9)
      {
        const int i = col - XO;
        if (k == i)
          return C_ONE;
        if (i >= 3 && i < 6 && k == i - 3)
          return C_DTM;
        if (i < 3 && k >= 6)
          return skew_off(k - 6, i, O_DTFP, O_DTFN);
      }
      return C_ZERO;
    }
  };

  struct CentSplitBuffers
  {
    double * rec = nullptr;  // [B][H][CentRec::STRIDE]
    double * term = nullptr; // [B][CentRec::T_STRIDE]
  };
  template <class D>
  struct CentSplitArgs
  {
    CentStepArgs<D> a;
    CentSplitBuffers sb;
    int last;  // cent_ls_body: last iteration of the solver run (state derivatives of the accepted iterate)
    int inst0; // first instance of this launch (the batch runs as several parts on several streams: block b = instance inst0 + b)
  };

  // ============================================================================================================
  // recede: the prologue of cent_step_body as its own launch
  // ============================================================================================================
  template <class D>
  SMPC_DEV void cent_recede_body(const CentSplitArgs<D> & sa, int block)
  {
    constexpr int NT = 64, NU = D::NU, NC = D::NC, NF = D::NF;
    const CentStepArgs<D> & ka = sa.a;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head;
    const size_t inst = (size_t)(sa.inst0 + block), ib = inst * R;
    const CentDevModel<D> & md = *b.model;
    SMPC_LDS(double, rec, NF * 6);
    double * gsc = b.scal + inst * SC_N;
    SMPC_LANES(NT)
    if (lane < 16)
      gsc[lane] = lane == SC_PREG && !ka.reset_preg ? gsc[SC_PREG] : 0.0;
    SMPC_LANES_END_WAVE
    if (ka.shift)
    {
      const int s0 = ring_slot(head, 0, R), sHm1 = ring_slot(head, H - 1, R), sH = ring_slot(head, H, R), sHm2 = ring_slot(head, H - 2, R);
      SMPC_LANES(NT)
      {
        if (lane < 9)
        {
          b.xs[(ib + s0) * 9 + lane] = ka.cstate[inst * 9 + lane];
          b.xs[(ib + sH) * 9 + lane] = b.xs[(ib + sHm1) * 9 + lane];
          b.lams[(ib + sHm1) * 9 + lane] = 0.0;
        }
        if (lane < NU)
          b.us[(ib + sHm1) * NU + lane] = b.us[(ib + sHm2) * NU + lane];
        if (lane < NC)
          b.vs[(ib + sHm1) * NC + lane] = 0.0;
        if (lane >= 32 && lane < 38)
          b.vref[(ib + sHm1) * 6 + lane - 32] = md.mass * b.vbase[inst * 6 + lane - 32];
        if (lane < NF)
        {
          const int f = lane;
          const double * xm = ka.X + inst * ka.nx_mb;
          const V3 pf = ld3(ka.feet + (inst * NF + f) * 3);
          const V3 bp = ld3(xm);
          const M3 Rb = quat_to_R(Quat{xm[3], xm[4], xm[5], xm[6]});
          const V3 refp = Rb * ld3(md.foot_ref_p[f]) + bp;
          const double tw0 = -(refp.y - bp.y), tw1 = refp.x - bp.x;
          const double span = (double)(ka.T_fly + ka.T_contact) * ka.timestep;
          const double * vb = b.vbase + inst * 6;
          const V3 next = mk3(refp.x + (vb[0] + vb[5] * tw0) * span, refp.y + (vb[1] + vb[5] * tw1) * span, pf.z);
          double * ft = b.ftraj + (inst * NF + f) * 6;
          if (!(ka.land[f] < ka.T_fly))
          {
            st3(ft, pf);
            st3(ft + 3, next);
          }
          st3(&rec[f * 6], ld3(ft));
          st3(&rec[f * 6 + 3], ld3(ft + 3));
        }
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      for (int idx = lane; idx < H * NF; idx += NT)
      {
        const int k = idx / NF, f = idx % NF;
        const int t = ka.land[f] - k;
        const V3 p0 = ld3(&rec[f * 6]), p1 = ld3(&rec[f * 6 + 3]);
        V3 p;
        if (t < 0)
          p = p1;
        else if (t > ka.T_fly)
          p = p0;
        else
          p = bezier8(p0, p1, ka.swing_apex, float(ka.T_fly - t) / float(ka.T_fly));
        st3(b.foot + ((inst * H + k) * NF + f) * 3, p);
      }
      SMPC_LANES_END_WAVE
    }
    if (ka.set_centres)
    {
      SMPC_LANES(NT)
      for (int t = lane; t < H; t += NT)
      {
        const size_t sl = ib + ring_slot(head, t, R);
        for (int i = 0; i < NC; i++)
          b.vs_e[sl * NC + i] = b.vs[sl * NC + i];
        for (int i = 0; i < 9; i++)
          b.lams_e[sl * 9 + i] = b.lams[sl * 9 + i];
      }
      SMPC_LANES_END_WAVE
    }
  }

  // ============================================================================================================
  // pre-pass: lane = stage of the block's instance
  // ============================================================================================================
  // the transposing flush of the stream hand-over (ev_stream_flush of smpc_kino_lane.h), INLINED: the pre-pass has 16 flush sites and some 40
  // doubles live across each of them -- around a call they would all be saved and restored through scratch memory
  SMPC_DEV void cent_stream_flush(const double * park, const unsigned * poff, double * rec, int np, int c0, int lane)
  {
    constexpr int NT = 64, PB = NT / (EV_CH / 2);
    const int f = 2 * (lane % (EV_CH / 2));
    constexpr int NB = 4;
#pragma unroll
    for (int q0 = 0; q0 < NT / PB; q0 += NB)
    {
      double v0[NB], v1[NB];
      unsigned po[NB];
#pragma unroll
      for (int q = 0; q < NB; q++)
      {
        const int p0 = (q0 + q) * PB + lane / (EV_CH / 2), p = p0 < np ? p0 : np - 1;
        v0[q] = park[p * EV_PP + f];
        v1[q] = park[p * EV_PP + f + 1];
        po[q] = poff[p];
      }
#pragma unroll
      for (int q = 0; q < NB; q++)
        store2_nowait(rec + (size_t)po[q] + c0 + f, v0[q], v1[q]);
    }
  }

  template <class D>
  struct CentPreLds
  {
    CentDevModel<D> md;
    double park[64 * EV_PP];
    unsigned poff[64];
    double red[4][64];
  };

  template <class D>
  SMPC_DEV void cent_pre_body(const CentSplitArgs<D> & sa, int block)
  {
    typedef CentRec<D> RC;
    constexpr int NT = 64, NU = D::NU, NC = D::NC, NF = D::NF;
    const CentStepArgs<D> & ka = sa.a;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head;
    const size_t inst = (size_t)(sa.inst0 + block), ib = inst * R;
    SMPC_LDS(CentPreLds<D>, ldsv, 1);
    CentPreLds<D> & s = ldsv[0];
    double * gsc = b.scal + inst * SC_N;
    SMPC_LANES(NT)
    {
      constexpr int N = (int)(sizeof(CentDevModel<D>) / sizeof(double));
      const alias_double * src = reinterpret_cast<const alias_double *>(b.model);
      alias_double * dst = reinterpret_cast<alias_double *>(&s.md);
      for (int i = lane; i < N; i += NT)
        dst[i] = src[i];
    }
    SMPC_LANES_END_WAVE
    const CentDevModel<D> & md = s.md;
    const double mu = md.mu, dt = md.dt, mass = md.mass;
    const double imu = 1.0 / mu, imass = 1.0 / mass;
    const double preg0 = gsc[SC_PREG];
    const double preg = preg0 > 0.0 ? preg0 : ka.reg_init;
    SMPC_PL(double, acc_cost, NT);
    SMPC_PL(double, acc_pen, NT);
    SMPC_PL(double, acc_prim, NT);
    SMPC_PL(double, acc_dual, NT);
    SMPC_LANES(NT)
    {
      SMPC_PLV(acc_cost) = 0.0;
      SMPC_PLV(acc_pen) = 0.0;
      SMPC_PLV(acc_prim) = 0.0;
      SMPC_PLV(acc_dual) = 0.0;
    }
    SMPC_LANES_END_WAVE
    int spos = 0; // (uniform: declared outside the lane phase so that it lives in a scalar register)
    for (int t0 = 0; t0 <= H; t0 += NT)
    {
      const int np = H - t0 < NT ? H - t0 : NT; // regular stages of this sweep (lanes np .. : terminal node / idle)
      SMPC_LANES(NT)
      {
        const int t = t0 + lane;
        const bool reg = t < H, term = t == H;
        const int te = reg ? t : H - 1; // (terminal / idle lanes run along on the last stage: the flush is a wave-wide transposition)
        const size_t sl = ib + ring_slot(head, te, R), sl1 = ib + ring_slot(head, te + 1, R);
        const unsigned mask = b.stages[te].mask;
        // ---- terminal node: its own small block, plain stores ----
        if (term)
        {
          const double * xH = b.xs + (ib + ring_slot(head, H, R)) * 9;
          const double * lH = b.lams + (ib + ring_slot(head, H - 1, R)) * 9;
          double * tr = sa.sb.term + inst * RC::T_STRIDE;
          double cN = 0.0, dN = 0.0;
          for (int i = 0; i < 9; i++)
          {
            double g = 0.0;
            for (int j = 0; j < 9; j++)
            {
              double v = i == j ? preg : 0.0;
              if (i >= 3 && i / 3 == j / 3)
              {
                const double * W = i < 6 ? md.w_lm : md.w_am; // (pointer first: g++ 11 with -fsanitize=shift miscompiles the indexed conditional of two arrays)
                const double w = W[(i % 3) * 3 + j % 3];
                v += w;
                g += w * xH[j];
              }
              tr[RC::T_P + i * 9 + j] = v;
            }
            const double qn = g - lH[i];
            tr[RC::T_p + i] = qn;
            tr[RC::T_lx + i] = g;
            dN = fmax(dN, fabs(qn));
            cN += 0.5 * xH[i] * g;
          }
          SMPC_PLV(acc_cost) += cN;
          SMPC_PLV(acc_dual) = fmax(SMPC_PLV(acc_dual), dN);
        }
        // ---- hand-over plumbing ----
        int kpos = 0;
        spos = 0;
        const size_t pbase = (inst * H + te) * RC::STRIDE;
        double * const park = s.park + lane * EV_PP;
        s.poff[lane] = (unsigned)pbase;
        auto put = [&](double v) {
          if constexpr (SMPC_LOCKSTEP)
            park[kpos] = v;
          else
          {
            if (reg)
              sa.sb.rec[pbase + spos + kpos] = v;
          }
          kpos++;
          if (kpos == EV_CH)
          {
            if constexpr (SMPC_LOCKSTEP)
            {
              if (np > 0) // (uniform)
                cent_stream_flush(s.park, s.poff, sa.sb.rec, np, spos, lane);
            }
            spos += EV_CH;
            kpos = 0;
          }
        };
        auto put3 = [&](V3 v) {
          put(v.x);
          put(v.y);
          put(v.z);
        };
        // (the sequential test build checks that the record is produced in the order CentRec lays it out; nothing on the device)
#define CENT_REC_AT(off) SMPC_TEST_CHECK(spos + kpos == (off), "cent_pre_body: the record is produced in a different order than CentRec lays it out")
        // (the order of the sections is the order of the record: the Hessian blocks need the lever arms and W_aa [r_f]x of all feet at once and
        //  come first; everything after them re-reads its inputs -- the loads hit in L1 / L2 and the registers are free in between)
        const double * const xg = b.xs + sl * 9;
        const double * const xn = b.xs + sl1 * 9;
        const double * const ug = b.us + sl * NU;
        const double * const xt = b.stages[te].x_tgt;
        const M3 Waa = ldm3(md.w_aa);
        V3 fs = mk3(0, 0, 0), ts = mk3(0, 0, 0);
        V3 rf[NF];
        double act[NF];
        {
          const double * const pp = b.foot + (inst * H + te) * (3 * NF);
          const V3 c = ld3(xg);
#pragma unroll
          for (int f = 0; f < NF; f++)
          {
            const bool on = (mask >> f) & 1u;
            const V3 F = ld3(ug + 3 * f);
            const V3 r = ld3(pp + 3 * f) - c;
            act[f] = on ? 1.0 : 0.0;
            rf[f] = act[f] * r;
            fs = fs + act[f] * F;
            ts = ts + cross(rf[f], F);
          }
        }
        // ---- Hessian blocks (Gauss-Newton) ----
        {
          M3 Nf[NF];
#pragma unroll
          for (int f = 0; f < NF; f++)
            Nf[f] = Waa * skew(rf[f]);
          CENT_REC_AT(RC::O_RR);
          const double im2 = imass * imass;
#pragma unroll
          for (int fa = 0; fa < NF; fa++)
          {
            const M3 St = transpose(skew(rf[fa]));
#pragma unroll
            for (int fb = fa; fb < NF; fb++)
            {
              const M3 X = St * Nf[fb];
              const double xe[9] = {X.a00, X.a01, X.a02, X.a10, X.a11, X.a12, X.a20, X.a21, X.a22};
              const double aa = act[fa] * act[fb] * im2;
#pragma unroll
              for (int ia = 0; ia < 3; ia++)
#pragma unroll
                for (int jb = 0; jb < 3; jb++)
                  if (fa != fb || jb >= ia)
                    put(md.w_u[(3 * fa + ia) * NU + 3 * fb + jb] + (fa == fb && ia == jb ? preg : 0.0) + aa * md.w_la[ia * 3 + jb] + xe[ia * 3 + jb]);
            }
          }
          CENT_REC_AT(RC::O_QC);
          const M3 Sf = transpose(skew(fs));
          const M3 X = Sf * (Waa * skew(fs));
          const double xe[9] = {X.a00, X.a01, X.a02, X.a10, X.a11, X.a12, X.a20, X.a21, X.a22};
#pragma unroll
          for (int a = 0; a < 3; a++)
#pragma unroll
            for (int bb = a; bb < 3; bb++)
              put(md.w_com[a * 3 + bb] + (a == bb ? preg : 0.0) + xe[a * 3 + bb]);
          CENT_REC_AT(RC::O_S);
#pragma unroll
          for (int f = 0; f < NF; f++)
          {
            const M3 Y = Sf * Nf[f]; // [xi][jb]
            put(Y.a00);
            put(Y.a01);
            put(Y.a02);
            put(Y.a10);
            put(Y.a11);
            put(Y.a12);
            put(Y.a20);
            put(Y.a21);
            put(Y.a22);
          }
        }
        // (pad to the quarter boundary: what follows is what the forward sweep reads too)
#pragma unroll
        for (int i = RC::N_HESS; i < RC::O_DTACT; i++)
          put(0.0);
        // ---- [A B] entries ----
        CENT_REC_AT(RC::O_DTACT);
#pragma unroll
        for (int f = 0; f < NF; f++)
          put(dt * act[f]);
#pragma unroll
        for (int f = 0; f < NF; f++)
          put3(dt * rf[f]);
#pragma unroll
        for (int f = 0; f < NF; f++)
          put3((-dt) * rf[f]);
        put3(dt * fs);
        put3((-dt) * fs);
        // ---- defect, multiplier estimates ----
        const V3 gv = ld3(md.gravity);
        double l1[9], lpd[9];
        double pen = 0.0, prim = 0.0, dual = 0.0, cost = 0.0;
        {
          // lams rows t - 1 and t (lams[slot(t)] = lambda_{t+1}; row -1 is never used: q = 0 at t = 0)
          const double * const le = b.lams_e + sl * 9;
#pragma unroll
          for (int i = 0; i < 9; i++)
            l1[i] = b.lams[sl * 9 + i];
          const double xd[9] = {xg[3] * imass, xg[4] * imass, xg[5] * imass, mass * gv.x + fs.x, mass * gv.y + fs.y, mass * gv.z + fs.z, ts.x, ts.y, ts.z};
          double fv[9];
#pragma unroll
          for (int i = 0; i < 9; i++)
          {
            const double e = xg[i] + dt * xd[i] - xn[i];
            const double lp = le[i] + e * imu, dl = lp - l1[i];
            fv[i] = mu * dl;
            lpd[i] = 2.0 * lp - l1[i];
            pen += 0.5 * mu * (lp * lp + dl * dl);
            prim = fmax(prim, fabs(e));
          }
          CENT_REC_AT(RC::O_F);
#pragma unroll
          for (int i = 0; i < 9; i++)
            put(fv[i]);
#pragma unroll
          for (int i = 0; i < 9; i++)
            put(lpd[i]);
        }
        // ---- residuals, weighted residuals, cost ----
        const V3 rla = gv + imass * fs;
        const V3 wla = ldm3(md.w_la) * rla, waa = Waa * ts;
        cost += 0.5 * (dot(rla, wla) + dot(ts, waa));
        // ---- state gradient: lx = W rx + [fs]x^T waa ; q, gx ----
        {
          const double * const href = b.vref + sl * 6;
          const V3 rc = ld3(xg) - ld3(xt), rh = ld3(xg + 3) - ld3(href), rL = ld3(xg + 6) - ld3(href + 3);
          const V3 wc = ldm3(md.w_com) * rc, wh = ldm3(md.w_lm) * rh, wL = ldm3(md.w_am) * rL;
          cost += 0.5 * (dot(rc, wc) + dot(rh, wh) + dot(rL, wL));
          const V3 lxc = wc + cross(waa, fs);
          const double lx[9] = {lxc.x, lxc.y, lxc.z, wh.x, wh.y, wh.z, wL.x, wL.y, wL.z};
          // A^T l = [l_c + dt (l_L x fs); l_h + dt l_c / m; l_L]
          auto AT = [&](const double * l, double * o) {
            const V3 w = cross(mk3(l[6], l[7], l[8]), fs);
            o[0] = l[0] + dt * w.x;
            o[1] = l[1] + dt * w.y;
            o[2] = l[2] + dt * w.z;
            o[3] = l[3] + dt * imass * l[0];
            o[4] = l[4] + dt * imass * l[1];
            o[5] = l[5] + dt * imass * l[2];
            o[6] = l[6];
            o[7] = l[7];
            o[8] = l[8];
          };
          double al1[9], alp[9];
          AT(l1, al1);
          AT(lpd, alp);
          CENT_REC_AT(RC::O_Q);
#pragma unroll
          for (int i = 0; i < 9; i++)
          {
            // x_0 is fixed (force_initial_condition)
            const double q = t > 0 ? lx[i] + al1[i] - b.lams[(ib + ring_slot(head, te > 0 ? te - 1 : 0, R)) * 9 + i] : 0.0;
            dual = fmax(dual, fabs(q));
            put(q);
          }
#pragma unroll
          for (int i = 0; i < 9; i++)
            put(lx[i] + alp[i]);
        }
        // ---- one foot at a time: friction-cone rows (src/centroidal-dynamics.cpp:90-96), control gradient lu = W_u ru + act wla / m + [r_f]x^T waa,
        //      r = lu + B_f^T lam+ + Cu^T nu, gu = lu + B_f^T (2 lam+ - lam) + Cu^T vpd ;  B_f^T l = dt (act l_h + l_L x r_f) ----
        {
          double ru[NU];
#pragma unroll
          for (int i = 0; i < NU; i++)
            ru[i] = ug[i] - b.stages[te].u_ref[i];
          const double * const vv = b.vs + sl * NC;
          const double * const ve = b.vs_e + sl * NC;
          bool anyact = false;
#pragma unroll
          for (int f = 0; f < NF; f++)
          {
            const bool on = (mask >> f) & 1u;
            const V3 F = ld3(ug + 3 * f);
            double sv[3] = {0, 0, 0}, sp[3] = {0, 0, 0}, dvec[2];
            CENT_REC_AT(RC::d_off(2 * f, 0));
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
            {
              const int row = 2 * f + rr;
              const bool cone = rr == 1;
              const double cv = cone ? F.x * F.x + F.y * F.y - md.mu_fric * md.mu_fric * F.z * F.z : -F.z + md.cone_eps;
              const double z = cv + mu * ve[row];
              const double proj = fmin(z, 0.0);
              const double vp = on ? (z - proj) * imu : 0.0;
              const bool a = on && z != proj;
              if (on)
                prim = fmax(prim, fmax(cv, 0.0));
              const double vrow = vv[row];
              const double dv = vp - vrow;
              dvec[rr] = mu * dv;
              const double vpd = a ? 2.0 * vp - vrow : 0.0;
              // (the Jacobian row exists for every foot in contact -- the gradient of the Lagrangian uses it with the current multiplier --
              //  and enters the KKT matrix only when the row is active)
              const double c0 = on && cone ? 2.0 * F.x : 0.0, c1 = on && cone ? 2.0 * F.y : 0.0;
              const double c2 = on ? (cone ? -2.0 * md.mu_fric * md.mu_fric * F.z : -1.0) : 0.0;
              sv[0] += c0 * vrow;
              sv[1] += c1 * vrow;
              sv[2] += c2 * vrow;
              sp[0] += c0 * vpd;
              sp[1] += c1 * vpd;
              sp[2] += c2 * vpd;
              anyact = anyact || a;
              pen += 0.5 * mu * (vp * vp + dv * dv);
              put(a ? c0 : 0.0);
              put(a ? c1 : 0.0);
              put(a ? c2 : 0.0);
            }
            put(dvec[0]);
            put(dvec[1]);
            const V3 wf = cross(waa, rf[f]);
            const V3 b1 = dt * (act[f] * mk3(l1[3], l1[4], l1[5]) + cross(mk3(l1[6], l1[7], l1[8]), rf[f]));
            const V3 bp = dt * (act[f] * mk3(lpd[3], lpd[4], lpd[5]) + cross(mk3(lpd[6], lpd[7], lpd[8]), rf[f]));
            double rv[3], gv3[3];
#pragma unroll
            for (int k = 0; k < 3; k++)
            {
              const int j = 3 * f + k;
              double wuj = 0.0;
#pragma unroll
              for (int jj = 0; jj < NU; jj++)
                wuj += md.w_u[j * NU + jj] * ru[jj];
              cost += 0.5 * ru[j] * wuj;
              const double luj = wuj + act[f] * (k == 0 ? wla.x : (k == 1 ? wla.y : wla.z)) * imass + (k == 0 ? wf.x : (k == 1 ? wf.y : wf.z));
              rv[k] = luj + (k == 0 ? b1.x : (k == 1 ? b1.y : b1.z)) + sv[k];
              gv3[k] = luj + (k == 0 ? bp.x : (k == 1 ? bp.y : bp.z)) + sp[k];
              dual = fmax(dual, fabs(rv[k]));
            }
            CENT_REC_AT(RC::r_off(3 * f));
            put(rv[0]);
            put(rv[1]);
            put(rv[2]);
            put(gv3[0]);
            put(gv3[1]);
            put(gv3[2]);
          }
          CENT_REC_AT(RC::O_ANY);
          put(anyact ? 1.0 : 0.0);
        }
        CENT_REC_AT(RC::N);
#pragma unroll
        for (int i = RC::N; i < RC::STRIDE; i++)
          put(0.0);
        if (reg)
        {
          SMPC_PLV(acc_cost) += cost;
          SMPC_PLV(acc_pen) += pen;
          SMPC_PLV(acc_prim) = fmax(SMPC_PLV(acc_prim), prim);
          SMPC_PLV(acc_dual) = fmax(SMPC_PLV(acc_dual), dual);
        }
      }
      SMPC_LANES_END_WAVE
    }
    // ---- merit terms of the iterate: fixed-order reductions ----
    SMPC_LANES(NT)
    {
      s.red[0][lane] = SMPC_PLV(acc_cost);
      s.red[1][lane] = SMPC_PLV(acc_pen);
      s.red[2][lane] = SMPC_PLV(acc_prim);
      s.red[3][lane] = SMPC_PLV(acc_dual);
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < 4)
    {
      const double v = lane < 2 ? fold64<false>(s.red[lane]) : fold64<true>(s.red[lane]);
      s.red[lane][0] = v;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      gsc[SC_COST] = s.red[0][0];
      gsc[SC_PHI0] = s.red[0][0] + s.red[1][0];
      gsc[SC_PRIM] = s.red[2][0];
      gsc[SC_DUAL] = s.red[3][0];
    }
    SMPC_LANES_END_WAVE
  }

  // ============================================================================================================
  // backward sweep
  // ============================================================================================================
  // Symmetric block sweep of smpc_riccati_kino.h (wave_block_sweep) with the pivot panels and the scheduling classes of the tiles given by
  // a plan: Plan::NP panels of 4 pivots starting at Plan::kb(p); Plan::cls(p, I, J) = 0 tile not maintained any more, 1 update before the
  // next panel's gather, 2 update deferred into the next panel's gather / inverse phases.
  template <int NT, int NTI, bool ALL, class Plan, bool PROF = true, class Acc>
  SMPC_DEV void wave_block_sweep_plan(Acc & acc, double * prow, double * urow, double * prof, long long & tprev, const SweepBases * sb = nullptr)
  {
    constexpr int LDW = 16 * NTI, NP = Plan::NP;
    static_assert(NT == 64 && LDW <= NT, "sweep geometry");
    SMPC_PLA(double, aop, NT, 2 * NTI);
    SMPC_PLA(double, bop, NT, 2 * NTI);
#pragma unroll
    for (int p = 0; p < NP; p++)
    {
      const int kb = Plan::kb(p), Ip = kb / 16, c0 = kb % 16, vp = c0 / 4;
      const int ob = (p & 1) * NTI, obp = ((p + 1) & 1) * NTI;
      // (a) pivot entries -> prow[k][m]
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
        // (address bases formed once per kernel where the caller sweeps in a loop: SweepBases, smpc_riccati_kino.h)
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
      if constexpr (PROF)
        prof_tick(prof, 36, tprev);
      if (p > 0)
      {
        int cnt = 0;
#pragma unroll
        for (int I = 0; I < NTI; I++)
#pragma unroll
          for (int J = I; J < NTI; J++)
            if (Plan::cls(p - 1, I, J) == 2)
            {
              if ((cnt & 1) == 0)
                SMPC_MFMA(acc, tix<NTI>(I, J), aop, obp + I, bop, obp + J);
              cnt++;
            }
      }
      // (b) U_m = D^-1 P_m, lane = index m (every lane factors the 4 x 4 pivot block D = L diag(d) L^T itself)
      SMPC_LANES(NT)
      {
        const double * d = prow + kb;
        const double D00 = d[0], D01 = d[1], D02 = d[2], D03 = d[3];
        const double D11 = d[LDW + 1], D12 = d[LDW + 2], D13 = d[LDW + 3];
        const double D22 = d[2 * LDW + 2], D23 = d[2 * LDW + 3], D33 = d[3 * LDW + 3];
        const double i0 = SMPC_RCP1(D00);
        const double l10 = D01 * i0, l20 = D02 * i0, l30 = D03 * i0;
        const double i1 = SMPC_RCP1(D11 - l10 * D01);
        const double t21 = D12 - l20 * D01, t31 = D13 - l30 * D01;
        const double l21 = t21 * i1, l31 = t31 * i1;
        const double i2 = SMPC_RCP1(D22 - l20 * D02 - l21 * t21);
        const double t32 = D23 - l30 * D02 - l31 * t21;
        const double l32 = t32 * i2;
        const double i3 = SMPC_RCP1(D33 - l30 * D03 - l31 * t31 - l32 * t32);
        const int M0 = ALL ? 0 : 16 * Ip;
        const int m = M0 + lane;
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
      SMPC_LANES_END_WAVE
      if constexpr (PROF)
        prof_tick(prof, 37, tprev);
      if (p > 0)
      {
        int cnt = 0;
#pragma unroll
        for (int I = 0; I < NTI; I++)
#pragma unroll
          for (int J = I; J < NTI; J++)
            if (Plan::cls(p - 1, I, J) == 2)
            {
              if ((cnt & 1) == 1)
                SMPC_MFMA(acc, tix<NTI>(I, J), aop, obp + I, bop, obp + J);
              cnt++;
            }
      }
      // (c) operands of this panel's rank-4 updates ; the updates the next panel depends on
      SMPC_LANES(NT)
      {
        const int brc = sb ? SMPC_PLV(sb->rc) : (lane >> 4) * LDW + (lane & 15);
#pragma unroll
        for (int I = ALL ? 0 : Ip; I < NTI; I++)
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
          if (Plan::cls(p, I, J) == 1)
            SMPC_MFMA(acc, tix<NTI>(I, J), aop, ob + I, bop, ob + J);
      if constexpr (PROF)
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
        if constexpr (PROF)
        prof_tick(prof, 39, tprev);
      }
    }
    // the last panel's deferred updates (a plan may defer in its last panel only what nothing reads before the sweep returns: none here)
  }

  // sweep 1: Schur complement of the 9 (+ 3 unit) leading pivots of [[I + mu P, sqrt(mu) P], [., P]]
  struct CentPlan1
  {
    static constexpr int NP = 3;
    SMPC_HD static constexpr int kb(int p) { return 4 * p; }
    SMPC_HD static constexpr int cls(int p, int I, int J)
    {
      (void)J;
      const bool last = p == NP - 1;
      if (I == 0)
        return last ? 0 : 1; // the pivot tile row: dead after the last panel
      return last ? 1 : 2;
    }
  };
  // sweep 2: pivots u (3 panels) and, with active cone rows, the multipliers nu_a (12..15), nu_b (28..31)
  template <bool CONES>
  struct CentPlan2
  {
    static constexpr int NP = CONES ? 5 : 3;
    SMPC_HD static constexpr int kb(int p) { return p < 4 ? 4 * p : 28; }
    SMPC_HD static constexpr int cls(int p, int I, int J)
    {
      if (p == NP - 1)
        return 1;
      const int Ip = kb(p) / 16, Ipn = kb(p + 1) / 16;
      const bool fixup = (I == Ip) || (J == Ip);
      const bool gather = (I == Ipn) || (J == Ipn && I < Ipn);
      return (fixup || gather) ? 1 : 2;
    }
  };

  // PROF: in-kernel phase timers (SMPC_PHASE_PROFILE=1 handles launch this instantiation).  Without it the stage loop carries no timer branch:
  // the sweep is bound by instruction issue, and the 25 scalar branches per stage the (disabled) timers cost are 2 % of it
  template <class D, bool PROF = false>
  SMPC_DEV void cent_bwd_body(const CentSplitArgs<D> & sa, int block)
  {
    typedef CentRec<D> RC;
    constexpr int NT = 64, NU = D::NU, NC = D::NC;
    const CentStepArgs<D> & ka = sa.a;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H;
    const size_t inst = (size_t)(sa.inst0 + block);
    SMPC_LDS(double, stg, RC::STAGE_N);
    SMPC_LDS(double, swp, 8 * 32);
    double * const prow = swp;
    double * const urow = swp + 4 * 32;
    double * dbg = inst == 0 ? b.dbg : nullptr;
    long long tprev = SMPC_CLOCK();
    const CentDevModel<D> & mg = *b.model;
    const double mu = mg.mu, smu = sqrt(mg.mu);
    const double preg0 = b.scal[inst * SC_N + SC_PREG];
    const double preg = preg0 > 0.0 ? preg0 : ka.reg_init;

    SMPC_PLA(int, off2, NT, 12);
    SMPC_PLA(int, offab, NT, 6);
    SMPC_ACC(Pa, NT, 1);   // P_{t+1} (rows / columns 0..8), p_{t+1} in column 9: tile (1, 1) of the previous stage's sweep 2
    SMPC_ACC(m1, NT, 3);
    SMPC_ACC(m2, NT, 3);
    SMPC_ACC(tacc, NT, 2);
    SMPC_PLA(double, pin, NT, RC::NLOAD);
    SMPC_PLA(double, aop, NT, 6);
    SMPC_PLA(double, ptl, NT, 3);
    SMPC_PLA(double, top, NT, 6);

    SMPC_LANES(NT)
    {
      const int lr = lane >> 4, lc = lane & 15;
      // constants behind the record
      if (lane < 16)
      {
        double v = 0.0;
        if (lane == RC::C_ONE - RC::STRIDE)
          v = 1.0;
        else if (lane == RC::C_DTM - RC::STRIDE)
          v = mg.dt / mg.mass;
        else if (lane == RC::C_NMU - RC::STRIDE)
          v = -mu;
        else if (lane >= RC::C_QLM - RC::STRIDE)
        {
          const int e = (lane - (RC::C_QLM - RC::STRIDE)) % 6; // upper triangle of a 3 x 3 block: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
          const int a = e < 3 ? 0 : (e < 5 ? 1 : 2), bb = e < 3 ? e : (e < 5 ? e - 2 : 2);
          const double * W = lane < RC::C_QAM - RC::STRIDE ? mg.w_lm : mg.w_am;
          v = W[a * 3 + bb] + (a == bb ? preg : 0.0);
        }
        stg[RC::STRIDE + lane] = v;
      }
      // gather offsets of this lane's accumulator entries
#pragma unroll
      for (int I = 0; I < 2; I++)
#pragma unroll
        for (int J = I; J < 2; J++)
#pragma unroll
          for (int v = 0; v < 4; v++)
            SMPC_PLV(off2)[tix<2>(I, J) * 4 + v] = RC::m2_off(16 * I + lr + 4 * v, 16 * J + lc);
#pragma unroll
      for (int sk = 0; sk < 3; sk++)
#pragma unroll
        for (int I = 0; I < 2; I++)
          SMPC_PLV(offab)[sk * 2 + I] = RC::ab_off(4 * sk + lr, 16 * I + lc);
      // terminal node
      const double * tr = sa.sb.term + inst * RC::T_STRIDE;
#pragma unroll
      for (int v = 0; v < 4; v++)
      {
        const int r = lr + 4 * v;
        double val = 0.0;
        if (r < 9 && lc < 9)
          val = tr[RC::T_P + r * 9 + lc];
        else if (r < 9 && lc == 9)
          val = tr[RC::T_p + r];
        SMPC_ACCV(Pa, 0, v) = val;
      }
      const double * rp = sa.sb.rec + (inst * H + (H - 1)) * RC::STRIDE;
#pragma unroll
      for (int n = 0; n < RC::NLOAD; n++)
        SMPC_PLV(pin)[n] = rp[lane + n * NT];
    }
    SMPC_LANES_END_WAVE

    // address bases of the sweeps' LDS staging (32-wide operand rows, both sweeps), formed once for the H stages
    SweepBases sb2;
    sweep_bases_init<32>(sb2);
    for (int t = H - 1; t >= 0; t--)
    {
      double * g = b.gains + (inst * H + t) * D::G_STRIDE;
      // ---- record -> LDS ; prefetch of stage t - 1 ; sweep-1 tiles from P ; p_{t+1} out ----
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
        for (int n = 0; n < RC::NLOAD; n++)
          stg[lane + n * NT] = SMPC_PLV(pin)[n];
        if (t > 0)
        {
          const double * rp = sa.sb.rec + (inst * H + (t - 1)) * RC::STRIDE;
#pragma unroll
          for (int n = 0; n < RC::NLOAD; n++)
            SMPC_PLV(pin)[n] = rp[lane + n * NT];
        }
#pragma unroll
        for (int v = 0; v < 4; v++)
        {
          const int r = lr + 4 * v;
          const double pv = (r < 9 && lc < 9) ? SMPC_ACCV(Pa, 0, v) : 0.0;
          SMPC_ACCV(m1, tix<2>(1, 1), v) = pv;
          SMPC_ACCV(m1, tix<2>(0, 1), v) = smu * pv;
          SMPC_ACCV(m1, tix<2>(0, 0), v) = mu * pv + (r == lc ? 1.0 : 0.0);
          if (lc == 9 && r < 9)
            g[RC::G_pn + r] = SMPC_ACCV(Pa, 0, v);
        }
      }
      SMPC_LANES_END_WAVE
      if constexpr (PROF)
        CENT_FINE_TICK(5);
      wave_block_sweep_plan<NT, 2, false, CentPlan1, PROF>(m1, prow, urow, CENT_FINE_DBG, tprev, &sb2);
      // ---- P~ (tile (1, 1)): out for the forward sweep; operand of the products.  p~ = p + P~ (f - mu p) ----
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
        for (int v = 0; v < 4; v++)
        {
          const int r = lr + 4 * v;
          if (lc < 9 && r <= lc)
            g[RC::pt_off(r, lc)] = SMPC_ACCV(m1, tix<2>(1, 1), v);
        }
#pragma unroll
        for (int sk = 0; sk < 3; sk++)
        {
          const int k = 4 * sk + lr;
          SMPC_PLV(ptl)[sk] = SMPC_ACCV(m1, tix<2>(1, 1), sk);
          SMPC_PLV(aop)[sk * 2 + 0] = stg[SMPC_PLV(offab)[sk * 2 + 0]];
          // the vector column (ZC) of [A B] is structurally zero: it carries f - mu p, so that column ZC of P~ [A B] is P~ (f - mu p) for free.
          // (As A operand of the second product the same entries land in ROW ZC of the stage matrix -- a row no sweep ever reads.)
          const double ab1 = stg[SMPC_PLV(offab)[sk * 2 + 1]];
          SMPC_PLV(aop)[sk * 2 + 1] = (lc == RC::ZC - 16 && k < 9) ? stg[RC::O_F + (k < 9 ? k : 0)] - mu * SMPC_ACCV(Pa, 0, sk) : ab1;
        }
#pragma unroll
        for (int J = 0; J < 2; J++)
#pragma unroll
          for (int v = 0; v < 4; v++)
            SMPC_ACCV(tacc, J, v) = 0.0;
#pragma unroll
        for (int n = 0; n < 12; n++)
          SMPC_ACCV(m2, n / 4, n % 4) = stg[SMPC_PLV(off2)[n]];
      }
      SMPC_LANES_END_WAVE
      if constexpr (PROF)
        CENT_FINE_TICK(7);
#pragma unroll
      for (int sk = 0; sk < 3; sk++)
#pragma unroll
        for (int J = 0; J < 2; J++)
          SMPC_MFMA(tacc, J, ptl, sk, aop, sk * 2 + J);
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
        for (int sk = 0; sk < 3; sk++)
        {
          SMPC_PLV(top)[sk * 2 + 0] = SMPC_ACCV(tacc, 0, sk);
          // column ZC of the right factor is p~ (so that the vector column receives [A B]^T p~)
          SMPC_PLV(top)[sk * 2 + 1] = SMPC_ACCV(tacc, 1, sk) + ((lc == RC::ZC - 16 && 4 * sk + lr < 9) ? SMPC_ACCV(Pa, 0, sk) : 0.0);
        }
      }
      SMPC_LANES_END_WAVE
#pragma unroll
      for (int sk = 0; sk < 3; sk++)
#pragma unroll
        for (int I = 0; I < 2; I++)
#pragma unroll
          for (int J = I; J < 2; J++)
            SMPC_MFMA(m2, tix<2>(I, J), aop, sk * 2 + I, top, sk * 2 + J);
      if constexpr (PROF)
        CENT_FINE_TICK(9);
      // ---- sweep 2: pivots u, then the multipliers of the active cone rows (quasi-definite KKT matrix: explicit pivots) ----
      const bool anyact = SMPC_UNIFORM_U32(stg[RC::O_ANY] != 0.0 ? 1u : 0u) != 0u;
      if (anyact)
        wave_block_sweep_plan<NT, 2, true, CentPlan2<true>, PROF>(m2, prow, urow, CENT_FINE_DBG, tprev, &sb2);
      else
        wave_block_sweep_plan<NT, 2, true, CentPlan2<false>, PROF>(m2, prow, urow, CENT_FINE_DBG, tprev, &sb2);
      // ---- gains out ; P_t, p_t stay in the accumulators ----
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
        if (lc < 10)
        {
#pragma unroll
          for (int v = 0; v < 3; v++)
            g[D::G_K + (lr + 4 * v) * D::GKS + lc] = -SMPC_ACCV(m2, tix<2>(0, 1), v);
          if (anyact)
          {
            if (lr < NC)
              g[RC::G_Z + lr * 10 + lc] = -SMPC_ACCV(m2, tix<2>(0, 1), 3);
            if (4 + lr < NC)
              g[RC::G_Z + (4 + lr) * 10 + lc] = -SMPC_ACCV(m2, tix<2>(1, 1), 3);
          }
        }
#pragma unroll
        for (int v = 0; v < 4; v++)
          SMPC_ACCV(Pa, 0, v) = SMPC_ACCV(m2, tix<2>(1, 1), v);
      }
      SMPC_LANES_END_WAVE
      if constexpr (PROF)
        CENT_FINE_TICK(8);
    }
  }

  // ============================================================================================================
  // forward sweep: dx_0 = 0;  du = K dx + k,  dnu = Z dx + z,  y = A dx + B du + f - mu p+,  w = P~ y,  dx+ = y - mu w,  dlam+ = w + p+
  // lane i < 9 carries dx_i; every product is fed by v_readlane broadcasts of the previous one
  // ============================================================================================================
  template <class D>
  SMPC_DEV void cent_fwd_body(const CentSplitArgs<D> & sa, int block)
  {
    typedef CentRec<D> RC;
    constexpr int NT = 64, NU = D::NU, NC = D::NC, NF = D::NF;
    // what the sweep reads of a stage: gains [K k | Z z | P~ | p+] (one contiguous run) and three of the four 64-double quarters of the
    // pre-pass record ([A B] entries in the first, vectors in the last two) -- coalesced loads one stage ahead, handed to the lanes through LDS
    constexpr int NG = RC::G_Z / NT; // [K k | P~ | p+]; [Z z] comes straight from memory in the stages that have an active cone row
    constexpr int QF = RC::Q_FWD, NR = RC::NLOAD - QF;
    static_assert(NR == 2 && RC::G_Z % NT == 0, "quarters of the record the forward sweep needs");
    const CentStepArgs<D> & ka = sa.a;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head;
    const size_t inst = (size_t)(sa.inst0 + block), ib = inst * R;
    const CentDevModel<D> & mg = *b.model;
    const double mu = mg.mu, imu = 1.0 / mg.mu, dtm = mg.dt / mg.mass;
    SMPC_LDS(double, gb, NG * NT);
    SMPC_LDS(double, rb, RC::STRIDE);
    SMPC_LDS(double, red, 64);
    SMPC_PL(double, dxr, NT);
    SMPC_PL(double, dur, NT);
    SMPC_PL(double, yr, NT);
    SMPC_PL(double, acc_dphi, NT);
    SMPC_PLA(double, gpre, NT, NG);
    SMPC_PLA(double, rpre, NT, NR);
    SMPC_PLA(int, po, NT, 9); // lane i < 9: places of row i of the packed P~
    SMPC_LANES(NT)
    {
      SMPC_PLV(acc_dphi) = 0.0;
      SMPC_PLV(dxr) = 0.0;
      SMPC_PLV(dur) = 0.0;
      SMPC_PLV(yr) = 0.0;
      if (lane < 9)
        b.dxs[(ib + ring_slot(head, 0, R)) * 9 + lane] = 0.0;
      const double * g = b.gains + (inst * H) * D::G_STRIDE;
      const double * rc = sa.sb.rec + (inst * H) * RC::STRIDE;
#pragma unroll
      for (int n = 0; n < NG; n++)
        SMPC_PLV(gpre)[n] = g[lane + n * NT];
#pragma unroll
      for (int n = 0; n < NR; n++)
        SMPC_PLV(rpre)[n] = rc[lane + (QF + n) * NT];
#pragma unroll
      for (int j = 0; j < 9; j++)
        SMPC_PLV(po)[j] = RC::pt_off(lane < 9 ? lane : 0, j);
    }
    SMPC_LANES_END_WAVE
    for (int t = 0; t < H; t++)
    {
      const size_t sl = ib + ring_slot(head, t, R), sl1 = ib + ring_slot(head, t + 1, R); // (the steps live on the iterate's ring slots)
      SMPC_LANES(NT)
      {
#pragma unroll
        for (int n = 0; n < NG; n++)
          gb[lane + n * NT] = SMPC_PLV(gpre)[n];
#pragma unroll
        for (int n = 0; n < NR; n++)
          rb[lane + (QF + n) * NT] = SMPC_PLV(rpre)[n];
        if (t + 1 < H)
        {
          const double * g = b.gains + (inst * H + t + 1) * D::G_STRIDE;
          const double * rc = sa.sb.rec + (inst * H + t + 1) * RC::STRIDE;
#pragma unroll
          for (int n = 0; n < NG; n++)
            SMPC_PLV(gpre)[n] = g[lane + n * NT];
#pragma unroll
          for (int n = 0; n < NR; n++)
            SMPC_PLV(rpre)[n] = rc[lane + (QF + n) * NT];
        }
      }
      SMPC_LANES_END_WAVE
      const bool any = SMPC_UNIFORM_U32(rb[RC::O_ANY] != 0.0 ? 1u : 0u) != 0u;
      // du = K dx + k (lanes 0..NU-1) ; dnu = Z dx + z, or d / mu without an active row (lanes 16..)
      SMPC_LANES(NT)
      {
        const bool urow = lane < NU, vrow = lane >= 16 && lane < 16 + NC;
        double a = 0.0;
        if (urow)
        {
          const double * row = gb + RC::G_K + lane * D::GKS;
          a = row[9];
#pragma unroll
          for (int j = 0; j < 9; j++)
            a += row[j] * SMPC_XLANE(dxr, j);
        }
        else if (vrow && any) // (rare: the [Z z] row of this stage straight from memory)
        {
          const double * row = b.gains + (inst * H + t) * D::G_STRIDE + RC::G_Z + (lane - 16) * 10;
          a = row[9];
#pragma unroll
          for (int j = 0; j < 9; j++)
            a += row[j] * SMPC_XLANE(dxr, j);
        }
        if (urow)
        {
          SMPC_PLV(dur) = a;
          b.dus[sl * NU + lane] = a;
          SMPC_PLV(acc_dphi) += rb[RC::gu_off(lane < NU ? lane : 0)] * a;
        }
        if (vrow)
        {
          const double d = rb[RC::dv_off(lane - 16 < NC && lane >= 16 ? lane - 16 : 0)];
          const double dn = any ? a : d * imu;
          b.dvs[sl * NC + lane - 16] = dn;
          SMPC_PLV(acc_dphi) -= d * dn;
        }
        if (lane < 9)
          SMPC_PLV(acc_dphi) += rb[RC::O_GX + lane] * SMPC_PLV(dxr);
      }
      SMPC_LANES_END_WAVE
      // y = A dx + B du + f - mu p+
      SMPC_LANES(NT)
      {
        const V3 dc = mk3(SMPC_XLANE(dxr, 0), SMPC_XLANE(dxr, 1), SMPC_XLANE(dxr, 2));
        const V3 dh = mk3(SMPC_XLANE(dxr, 3), SMPC_XLANE(dxr, 4), SMPC_XLANE(dxr, 5));
        V3 sf = mk3(0, 0, 0);
        V3 tq = cross(ld3(rb + RC::O_DTFP), dc); // dt [fs]x dc
#pragma unroll
        for (int f = 0; f < NF; f++)
        {
          const V3 duf = mk3(SMPC_XLANE(dur, 3 * f), SMPC_XLANE(dur, 3 * f + 1), SMPC_XLANE(dur, 3 * f + 2));
          sf = sf + rb[RC::O_DTACT + f] * duf;
          tq = tq + cross(ld3(rb + RC::O_DTRP + 3 * f), duf);
        }
        const V3 lin = dtm * dh;
        const int k = lane % 3;
        const double add = lane < 3 ? (k == 0 ? lin.x : (k == 1 ? lin.y : lin.z)) : (lane < 6 ? (k == 0 ? sf.x : (k == 1 ? sf.y : sf.z)) : (k == 0 ? tq.x : (k == 1 ? tq.y : tq.z)));
        if (lane < 9)
          SMPC_PLV(yr) = SMPC_PLV(dxr) + add + rb[RC::O_F + lane] - mu * gb[RC::G_pn + lane];
      }
      SMPC_LANES_END_WAVE
      // w = P~ y ; dx+ = y - mu w ; dlam+ = w + p+
      SMPC_LANES(NT)
      {
        double w = 0.0;
#pragma unroll
        for (int j = 0; j < 9; j++)
          w += gb[SMPC_PLV(po)[j]] * SMPC_XLANE(yr, j);
        if (lane < 9)
        {
          const double dxn = SMPC_PLV(yr) - mu * w;
          const double dl = w + gb[RC::G_pn + lane];
          b.dxs[sl1 * 9 + lane] = dxn;
          b.dlams[sl * 9 + lane] = dl;
          SMPC_PLV(acc_dphi) -= rb[RC::O_LPD + lane] * dxn + rb[RC::O_F + lane] * dl;
          SMPC_PLV(dxr) = dxn;
        }
      }
      SMPC_LANES_END_WAVE
    }
    // terminal gradient lx_N . dx_H ; reduction (lane order: deterministic)
    SMPC_LANES(NT)
    {
      if (lane < 9)
        SMPC_PLV(acc_dphi) += sa.sb.term[inst * RC::T_STRIDE + RC::T_lx + lane] * SMPC_PLV(dxr);
      red[lane] = SMPC_PLV(acc_dphi);
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
      b.scal[inst * SC_N + SC_DPHI0] = fold64<false>(red);
    SMPC_LANES_END_WAVE
  }

  template <class D>
  struct CentLsLds
  {
    CentDevModel<D> md;
    double red[64];
    double rows[64 * D::NU]; // lane-private strips of the control residual (dense weight, rolled product)
    double sc[16];
  };
  // The steps dxs, dus, dvs, dlams of this pipeline live on the SAME ring slots as the iterate ([B][R][.]: entry of stage t at ring_slot(head, t)),
  // so a trial point is w[e] + alpha dw[e] element by element and accepting a step is a flat axpy over the instance's arrays.
  template <class D>
  SMPC_DEV void cent_ls_body(const CentSplitArgs<D> & sa, int block)
  {
    constexpr int NT = 64, NU = D::NU, NC = D::NC, NF = D::NF;
    static_assert(NU <= 12 && NC <= 12, "row buffer");
    const CentStepArgs<D> & ka = sa.a;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head;
    const size_t inst = (size_t)(sa.inst0 + block), ib = inst * R;
    SMPC_LDS(CentLsLds<D>, ldsv, 1);
    CentLsLds<D> & s = ldsv[0];
    double * const red = s.red;
    double * gsc = b.scal + inst * SC_N;
    SMPC_LANES(NT)
    {
      constexpr int N = (int)(sizeof(CentDevModel<D>) / sizeof(double));
      const alias_double * src = reinterpret_cast<const alias_double *>(b.model);
      alias_double * dst = reinterpret_cast<alias_double *>(&s.md);
      for (int i = lane; i < N; i += NT)
        dst[i] = src[i];
      if (lane < 16)
        s.sc[lane] = gsc[lane];
    }
    SMPC_LANES_END_WAVE
    const CentDevModel<D> & md = s.md;
    const double mu = md.mu, imu = 1.0 / md.mu, imass = 1.0 / md.mass;
    const double preg = s.sc[SC_PREG] > 0.0 ? s.sc[SC_PREG] : ka.reg_init;
    SMPC_PL(double, acc_cost, NT);
    SMPC_PL(double, acc_pen, NT);
    SMPC_PL(double, acc_prim, NT);
    double alpha = 1.0;
    int accepted = -1, jlast = 0;
    for (int j = 0; j < D::LS_N; j++)
    {
      jlast = j;
      SMPC_LANES(NT)
      {
        double cst = 0.0, pen = 0.0, prm = 0.0;
        for (int t = lane; t <= H; t += NT)
        {
          const size_t sl = ib + ring_slot(head, t, R);
          const double * xg = b.xs + sl * 9;
          const double * dxg = b.dxs + sl * 9;
          if (t == H)
          {
            const V3 hh = mk3(xg[3] + alpha * dxg[3], xg[4] + alpha * dxg[4], xg[5] + alpha * dxg[5]);
            const V3 LL = mk3(xg[6] + alpha * dxg[6], xg[7] + alpha * dxg[7], xg[8] + alpha * dxg[8]);
            cst += 0.5 * dot(hh, ldm3(md.w_lm) * hh) + 0.5 * dot(LL, ldm3(md.w_am) * LL);
            continue;
          }
          const size_t sl1 = ib + ring_slot(head, t + 1, R);
          CentTrial<D> q;
          q.x = xg;
          q.dx = dxg;
          q.xn = b.xs + sl1 * 9;
          q.dxn = b.dxs + sl1 * 9;
          q.u = b.us + sl * NU;
          q.du = b.dus + sl * NU;
          q.v = b.vs + sl * NC;
          q.dv = b.dvs + sl * NC;
          q.l1 = b.lams + sl * 9;
          q.dl = b.dlams + sl * 9;
          q.ve = b.vs_e + sl * NC;
          q.l1e = b.lams_e + sl * 9;
          q.p = b.foot + (inst * H + t) * (3 * NF);
          q.uref = b.stages[t].u_ref;
          q.xtgt = b.stages[t].x_tgt;
          q.href = b.vref + sl * 6;
          q.alpha = alpha;
          q.mask = b.stages[t].mask;
          double c1, p1, r1;
          cent_stage_merit<D>(md, q, c1, p1, r1, nullptr, &s.rows[lane * NU]);
          cst += c1;
          pen += p1;
          prm = fmax(prm, r1);
        }
        SMPC_PLV(acc_cost) = cst;
        SMPC_PLV(acc_pen) = pen;
        SMPC_PLV(acc_prim) = prm;
      }
      SMPC_LANES_END_WAVE
      for (int which = 0; which < 3; which++)
      {
        SMPC_LANES(NT)
        red[lane] = which == 0 ? SMPC_PLV(acc_cost) : (which == 1 ? SMPC_PLV(acc_pen) : SMPC_PLV(acc_prim));
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        if (lane == 0)
        {
          if (which == 0)
            s.sc[SC_COST_NEW] = fold64<false>(red);
          else if (which == 1)
            s.sc[SC_PHI_NEW] = s.sc[SC_COST_NEW] + fold64<false>(red);
          else
            s.sc[SC_PRIM_NEW] = fold64<true>(red);
        }
        SMPC_LANES_END_WAVE
      }
      if (s.sc[SC_PHI_NEW] <= s.sc[SC_PHI0] + ka.armijo_c1 * alpha * s.sc[SC_DPHI0])
      {
        accepted = j;
        break;
      }
      if (j + 1 < D::LS_N)
        alpha *= 0.5;
    }
    // ---- accept (the last candidate is taken when none passes, like the restated solver): flat axpy; the slot of stage H carries no
    //      control / multiplier ----
    SMPC_LANES(NT)
    {
      const int sH = ring_slot(head, H, R);
      for (int e = lane; e < R * 9; e += NT)
      {
        b.xs[ib * 9 + e] += alpha * b.dxs[ib * 9 + e];
        if (e / 9 != sH)
          b.lams[ib * 9 + e] += alpha * b.dlams[ib * 9 + e];
      }
      for (int e = lane; e < R * NU; e += NT)
        if (e / NU != sH)
          b.us[ib * NU + e] += alpha * b.dus[ib * NU + e];
      for (int e = lane; e < R * NC; e += NT)
        if (e / NC != sH)
          b.vs[ib * NC + e] += alpha * b.dvs[ib * NC + e];
      if (lane == 0)
      {
        s.sc[SC_ALPHA] = alpha;
        s.sc[SC_LS_FAILED] = accepted < 0 ? 1.0 : 0.0;
        s.sc[SC_LS_INDEX] = (double)jlast;
        s.sc[SC_PREG] = accepted < 0 ? fmin(preg * ka.reg_inc, ka.reg_max) : fmax(preg * ka.reg_dec, ka.reg_min);
      }
    }
    SMPC_LANES_END_WAVE
    // ---- outputs: solver scalars ; xdot at t = 0, 1 of the accepted iterate (MPC::getStateDerivative) after the last iteration ----
    SMPC_LANES(NT)
    {
      if (lane < 16)
        gsc[lane] = s.sc[lane];
      if (lane >= 32 && lane < 34 && sa.last)
      {
        // xdot = [h / m; m g + sum f; sum (p_f - c) x f] of the accepted iterate
        const int t = lane - 32;
        const size_t sl = ib + ring_slot(head, t, R);
        const double * xg = b.xs + sl * 9;
        const unsigned mask = b.stages[t].mask;
        const V3 c = ld3(xg);
        V3 fs = mk3(0, 0, 0), ts = mk3(0, 0, 0);
        for (int f = 0; f < NF; f++)
          if ((mask >> f) & 1u)
          {
            const V3 F = ld3(b.us + sl * NU + 3 * f);
            fs = fs + F;
            ts = ts + cross(ld3(b.foot + (inst * H + t) * (3 * NF) + 3 * f) - c, F);
          }
        double * xo = b.xdot01 + (inst * 2 + t) * 9;
        for (int i = 0; i < 3; i++)
        {
          xo[i] = xg[3 + i] * imass;
          xo[3 + i] = md.mass * md.gravity[i] + (i == 0 ? fs.x : (i == 1 ? fs.y : fs.z));
          xo[6 + i] = i == 0 ? ts.x : (i == 1 ? ts.y : ts.z);
        }
      }
    }
    SMPC_LANES_END_WAVE
  }

  // ============================================================================================================
  // line search, polynomial form (horizons of at most 63 stages: one lane per stage and one for the terminal node)
  // ============================================================================================================
  // Along the search direction every residual of the centroidal stage is a polynomial in the step size: states, forces and multipliers move
  // linearly, the torque sum (p_f - c) x f_f is bilinear, the friction cone quadratic.  So the merit of stage t at w + alpha dw is
  //     Q_t(alpha)  (one quartic: all costs, the dynamics penalty, the penalty of the rows of feet in the air)
  //   + the cone rows of the feet in contact (a quadratic z(alpha) each, with the projection max(z, 0) applied per candidate).
  // cent_ls_body re-reads iterate and step -- some 150 doubles a lane, 64 cache lines per load instruction -- for EVERY candidate; here they
  // are read once, the quartic stays in 10 registers, the cone rows in LDS, and a candidate costs ~100 instructions and one reduction.
  template <class D>
  struct CentLsPolyLds
  {
    static constexpr int NCK = 9; // per foot in contact: z0 = a + b alpha ; z1 = q0 + q1 alpha + q2 alpha^2 ; v0, dv0, v1, dv1
    CentDevModel<D> md;
    double red[64];
    double cone[D::NF * NCK][64];
    double sc[16];
  };
  template <class D>
  SMPC_DEV void cent_ls_poly_body(const CentSplitArgs<D> & sa, int block)
  {
    typedef CentLsPolyLds<D> LD;
    constexpr int NT = 64, NU = D::NU, NC = D::NC, NF = D::NF, NCK = LD::NCK;
    const CentStepArgs<D> & ka = sa.a;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head;
    const size_t inst = (size_t)(sa.inst0 + block), ib = inst * R;
    SMPC_LDS(LD, ldsv, 1);
    LD & s = ldsv[0];
    double * const red = s.red;
    double * gsc = b.scal + inst * SC_N;
    SMPC_LANES(NT)
    {
      constexpr int N = (int)(sizeof(CentDevModel<D>) / sizeof(double));
      const alias_double * src = reinterpret_cast<const alias_double *>(b.model);
      alias_double * dst = reinterpret_cast<alias_double *>(&s.md);
      for (int i = lane; i < N; i += NT)
        dst[i] = src[i];
      if (lane < 16)
        s.sc[lane] = gsc[lane];
    }
    SMPC_LANES_END_WAVE
    const CentDevModel<D> & md = s.md;
    const double mu = md.mu, imu = 1.0 / md.mu, imass = 1.0 / md.mass, dt = md.dt;
    const double preg = s.sc[SC_PREG] > 0.0 ? s.sc[SC_PREG] : ka.reg_init;
    // per-lane polynomial state
    SMPC_PLA(double, qc, NT, 5); // cost part of the quartic
    SMPC_PLA(double, qp, NT, 5); // penalty part
    SMPC_PLA(double, e0, NT, 9);
    SMPC_PLA(double, e1, NT, 9);
    SMPC_PLA(double, e2, NT, 3);
    SMPC_PL(unsigned, lmask, NT);
    SMPC_LANES(NT)
    {
      const int t = lane;
      double * QC = SMPC_PLV(qc), *QP = SMPC_PLV(qp);
#pragma unroll
      for (int i = 0; i < 5; i++)
        QC[i] = QP[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 9; i++)
        SMPC_PLV(e0)[i] = SMPC_PLV(e1)[i] = 0.0;
#pragma unroll
      for (int i = 0; i < 3; i++)
        SMPC_PLV(e2)[i] = 0.0;
      SMPC_PLV(lmask) = 0u;
      // 1/2 (r0 + alpha r1)^T W (r0 + alpha r1) into a quadratic
      auto quad = [&](const double * W, V3 r0, V3 r1, double * Q) {
        const M3 Wm = ldm3(W);
        const V3 w0 = Wm * r0, w1 = Wm * r1;
        Q[0] += 0.5 * dot(r0, w0);
        Q[1] += dot(r1, w0);
        Q[2] += 0.5 * dot(r1, w1);
      };
      if (t <= H)
      {
        const size_t sl = ib + ring_slot(head, t, R);
        const double * xg = b.xs + sl * 9;
        const double * dxg = b.dxs + sl * 9;
        const V3 c0 = ld3(xg), dc = ld3(dxg), h0 = ld3(xg + 3), dh = ld3(dxg + 3), L0 = ld3(xg + 6), dL = ld3(dxg + 6);
        if (t == H)
        {
          quad(md.w_lm, h0, dh, QC);
          quad(md.w_am, L0, dL, QC);
        }
        else
        {
          const unsigned mask = b.stages[t].mask;
          SMPC_PLV(lmask) = mask;
          const size_t lt = inst * H + t;
          // ---- forces: sums, torque sums, cone rows, control cost ----
          V3 fs0 = mk3(0, 0, 0), fs1 = mk3(0, 0, 0), ts0 = mk3(0, 0, 0), ts1 = mk3(0, 0, 0);
          double ru[NU], du[NU];
#pragma unroll
          for (int f = 0; f < NF; f++)
          {
            const bool on = (mask >> f) & 1u;
            const V3 F0 = ld3(b.us + sl * NU + 3 * f), dF = ld3(b.dus + sl * NU + 3 * f), ur = ld3(b.stages[t].u_ref + 3 * f);
            ru[3 * f] = F0.x - ur.x;
            ru[3 * f + 1] = F0.y - ur.y;
            ru[3 * f + 2] = F0.z - ur.z;
            du[3 * f] = dF.x;
            du[3 * f + 1] = dF.y;
            du[3 * f + 2] = dF.z;
            const double v0 = b.vs[sl * NC + 2 * f], dv0 = b.dvs[sl * NC + 2 * f], v1 = b.vs[sl * NC + 2 * f + 1], dv1 = b.dvs[sl * NC + 2 * f + 1];
            // (vp - v(alpha))^2 + vp^2 = 2 vp^2 - 2 vp v(alpha) + v(alpha)^2: the last term is a polynomial whether the foot is in contact or not
            QP[0] += 0.5 * mu * (v0 * v0 + v1 * v1);
            QP[1] += mu * (v0 * dv0 + v1 * dv1);
            QP[2] += 0.5 * mu * (dv0 * dv0 + dv1 * dv1);
            double ck[NCK] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (on)
            {
              const V3 r0 = ld3(b.foot + lt * (3 * NF) + 3 * f) - c0;
              fs0 = fs0 + F0;
              fs1 = fs1 + dF;
              ts0 = ts0 + cross(r0, F0);
              ts1 = ts1 + cross(r0, dF);
              const double mf2 = md.mu_fric * md.mu_fric;
              ck[0] = -F0.z + md.cone_eps + mu * b.vs_e[sl * NC + 2 * f];
              ck[1] = -dF.z;
              ck[2] = F0.x * F0.x + F0.y * F0.y - mf2 * F0.z * F0.z + mu * b.vs_e[sl * NC + 2 * f + 1];
              ck[3] = 2.0 * (F0.x * dF.x + F0.y * dF.y - mf2 * F0.z * dF.z);
              ck[4] = dF.x * dF.x + dF.y * dF.y - mf2 * dF.z * dF.z;
              ck[5] = v0;
              ck[6] = dv0;
              ck[7] = v1;
              ck[8] = dv1;
            }
#pragma unroll
            for (int k = 0; k < NCK; k++)
              s.cone[f * NCK + k][lane] = ck[k];
          }
          {
            double c00 = 0.0, c01 = 0.0, c11 = 0.0;
#pragma unroll
            for (int i = 0; i < NU; i++)
            {
              double wr = 0.0, wd = 0.0;
#pragma unroll
              for (int jj = 0; jj < NU; jj++)
              {
                wr += md.w_u[i * NU + jj] * ru[jj];
                wd += md.w_u[i * NU + jj] * du[jj];
              }
              c00 += ru[i] * wr;
              c01 += du[i] * wr;
              c11 += du[i] * wd;
            }
            QC[0] += 0.5 * c00;
            QC[1] += c01;
            QC[2] += 0.5 * c11;
          }
          ts1 = ts1 - cross(dc, fs0);
          const V3 ts2 = mk3(0, 0, 0) - cross(dc, fs1);
          // ---- costs ----
          const V3 g = ld3(md.gravity);
          quad(md.w_com, c0 - ld3(b.stages[t].x_tgt), dc, QC);
          quad(md.w_lm, h0 - ld3(b.vref + sl * 6), dh, QC);
          quad(md.w_am, L0 - ld3(b.vref + sl * 6 + 3), dL, QC);
          quad(md.w_la, g + imass * fs0, imass * fs1, QC);
          {
            const M3 Wm = ldm3(md.w_aa);
            const V3 w0 = Wm * ts0, w1 = Wm * ts1, w2 = Wm * ts2;
            QC[0] += 0.5 * dot(ts0, w0);
            QC[1] += dot(ts1, w0);
            QC[2] += 0.5 * dot(ts1, w1) + dot(ts2, w0);
            QC[3] += dot(ts2, w1);
            QC[4] += 0.5 * dot(ts2, w2);
          }
          // ---- dynamics defect e(alpha) = e0 + alpha e1 + alpha^2 e2 and its penalty 1/2 mu (lp^2 + (lp - lam(alpha))^2), lp = lam_e + e / mu ----
          const size_t sl1 = ib + ring_slot(head, t + 1, R);
          const double xd0[9] = {h0.x * imass, h0.y * imass, h0.z * imass, md.mass * g.x + fs0.x, md.mass * g.y + fs0.y, md.mass * g.z + fs0.z, ts0.x, ts0.y, ts0.z};
          const double xd1[9] = {dh.x * imass, dh.y * imass, dh.z * imass, fs1.x, fs1.y, fs1.z, ts1.x, ts1.y, ts1.z};
          const double xd2[3] = {ts2.x, ts2.y, ts2.z};
#pragma unroll
          for (int i = 0; i < 9; i++)
          {
            const double a0 = xg[i] + dt * xd0[i] - b.xs[sl1 * 9 + i];
            const double a1 = dxg[i] + dt * xd1[i] - b.dxs[sl1 * 9 + i];
            const double a2 = i >= 6 ? dt * xd2[i - 6] : 0.0;
            SMPC_PLV(e0)[i] = a0;
            SMPC_PLV(e1)[i] = a1;
            if (i >= 6)
              SMPC_PLV(e2)[i - 6] = a2;
            const double p0 = b.lams_e[sl * 9 + i] + a0 * imu, p1 = a1 * imu, p2 = a2 * imu;
            const double d0 = p0 - b.lams[sl * 9 + i], d1 = p1 - b.dlams[sl * 9 + i];
            QP[0] += 0.5 * mu * (p0 * p0 + d0 * d0);
            QP[1] += mu * (p0 * p1 + d0 * d1);
            QP[2] += 0.5 * mu * (p1 * p1 + d1 * d1 + 2.0 * p2 * (p0 + d0));
            QP[3] += mu * p2 * (p1 + d1);
            QP[4] += mu * p2 * p2;
          }
        }
      }
    }
    SMPC_LANES_END_WAVE
    // ---- candidates alpha = 1, 1/2, ... : merit from the stored polynomials ----
    double alpha = 1.0;
    int accepted = -1, jlast = 0;
    for (int j = 0; j < D::LS_N; j++)
    {
      jlast = j;
      SMPC_LANES(NT)
      {
        const double * QC = SMPC_PLV(qc), *QP = SMPC_PLV(qp);
        double phi = ((((QC[4] + QP[4]) * alpha + (QC[3] + QP[3])) * alpha + (QC[2] + QP[2])) * alpha + (QC[1] + QP[1])) * alpha + (QC[0] + QP[0]);
        const unsigned mask = SMPC_PLV(lmask);
#pragma unroll
        for (int f = 0; f < NF; f++)
          if ((mask >> f) & 1u)
          {
            const double z0 = s.cone[f * NCK + 0][lane] + alpha * s.cone[f * NCK + 1][lane];
            const double z1 = s.cone[f * NCK + 2][lane] + alpha * (s.cone[f * NCK + 3][lane] + alpha * s.cone[f * NCK + 4][lane]);
            const double vp0 = fmax(z0, 0.0) * imu, vp1 = fmax(z1, 0.0) * imu;
            const double va0 = s.cone[f * NCK + 5][lane] + alpha * s.cone[f * NCK + 6][lane], va1 = s.cone[f * NCK + 7][lane] + alpha * s.cone[f * NCK + 8][lane];
            phi += mu * (vp0 * (vp0 - va0) + vp1 * (vp1 - va1));
          }
        red[lane] = phi;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane == 0)
        s.sc[SC_PHI_NEW] = fold64<false>(red);
      SMPC_LANES_END_WAVE
      if (s.sc[SC_PHI_NEW] <= s.sc[SC_PHI0] + ka.armijo_c1 * alpha * s.sc[SC_DPHI0])
      {
        accepted = j;
        break;
      }
      if (j + 1 < D::LS_N)
        alpha *= 0.5;
    }
    // ---- cost and primal infeasibility at the accepted candidate (reported, not part of the test) ----
    SMPC_LANES(NT)
    {
      const double * QC = SMPC_PLV(qc);
      red[lane] = (((QC[4] * alpha + QC[3]) * alpha + QC[2]) * alpha + QC[1]) * alpha + QC[0];
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
      s.sc[SC_COST_NEW] = fold64<false>(red);
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      double prm = 0.0;
#pragma unroll
      for (int i = 0; i < 9; i++)
        prm = fmax(prm, fabs(SMPC_PLV(e0)[i] + alpha * (SMPC_PLV(e1)[i] + (i >= 6 ? alpha * SMPC_PLV(e2)[i - 6] : 0.0))));
      const unsigned mask = SMPC_PLV(lmask);
      if (mask != 0u)
      {
        const size_t sl = ib + ring_slot(head, lane, R);
#pragma unroll
        for (int f = 0; f < NF; f++)
          if ((mask >> f) & 1u)
          {
            const double c0 = s.cone[f * NCK + 0][lane] + alpha * s.cone[f * NCK + 1][lane] - mu * b.vs_e[sl * NC + 2 * f];
            const double c1 = s.cone[f * NCK + 2][lane] + alpha * (s.cone[f * NCK + 3][lane] + alpha * s.cone[f * NCK + 4][lane]) - mu * b.vs_e[sl * NC + 2 * f + 1];
            prm = fmax(prm, fmax(fmax(c0, 0.0), fmax(c1, 0.0)));
          }
      }
      red[lane] = prm;
    }
    SMPC_LANES_END_WAVE
    // ---- accept (the last candidate is taken when none passes, like the restated solver): flat axpy; the slot of stage H carries no
    //      control / multiplier ----
    SMPC_LANES(NT)
    {
      if (lane == 0)
        s.sc[SC_PRIM_NEW] = fold64<true>(red);
      const int sH = ring_slot(head, H, R);
      for (int e = lane; e < R * 9; e += NT)
      {
        b.xs[ib * 9 + e] += alpha * b.dxs[ib * 9 + e];
        if (e / 9 != sH)
          b.lams[ib * 9 + e] += alpha * b.dlams[ib * 9 + e];
      }
      for (int e = lane; e < R * NU; e += NT)
        if (e / NU != sH)
          b.us[ib * NU + e] += alpha * b.dus[ib * NU + e];
      for (int e = lane; e < R * NC; e += NT)
        if (e / NC != sH)
          b.vs[ib * NC + e] += alpha * b.dvs[ib * NC + e];
      if (lane == 0)
      {
        s.sc[SC_ALPHA] = alpha;
        s.sc[SC_LS_FAILED] = accepted < 0 ? 1.0 : 0.0;
        s.sc[SC_LS_INDEX] = (double)jlast;
        s.sc[SC_PREG] = accepted < 0 ? fmin(preg * ka.reg_inc, ka.reg_max) : fmax(preg * ka.reg_dec, ka.reg_min);
      }
    }
    SMPC_LANES_END_WAVE
    // ---- outputs: solver scalars ; xdot at t = 0, 1 of the accepted iterate (MPC::getStateDerivative) after the last iteration ----
    SMPC_LANES(NT)
    {
      if (lane < 16)
        gsc[lane] = s.sc[lane];
      if (lane >= 32 && lane < 34 && sa.last)
      {
        const int t = lane - 32;
        const size_t sl = ib + ring_slot(head, t, R);
        const double * xg = b.xs + sl * 9;
        const unsigned mask = b.stages[t].mask;
        const V3 c = ld3(xg);
        V3 fs = mk3(0, 0, 0), ts = mk3(0, 0, 0);
        for (int f = 0; f < NF; f++)
          if ((mask >> f) & 1u)
          {
            const V3 F = ld3(b.us + sl * NU + 3 * f);
            fs = fs + F;
            ts = ts + cross(ld3(b.foot + (inst * H + t) * (3 * NF) + 3 * f) - c, F);
          }
        double * xo = b.xdot01 + (inst * 2 + t) * 9;
        for (int i = 0; i < 3; i++)
        {
          xo[i] = xg[3 + i] * imass;
          xo[3 + i] = md.mass * md.gravity[i] + (i == 0 ? fs.x : (i == 1 ? fs.y : fs.z));
          xo[6 + i] = i == 0 ? ts.x : (i == 1 ? ts.y : ts.z);
        }
      }
    }
    SMPC_LANES_END_WAVE
  }
} // namespace smpc
