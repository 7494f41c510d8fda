// smpc_kino_deriv2.h -- derivative pass of a kinodynamics stage, one wavefront per (instance, stage), STARTING from the evaluation
// lane_tree_body has done for the problem (smpc_kino_lane.h): HOT(2) derivatives + HOT(3) LQ knot without the serial chain of
// HOT(1).  What is left has lane parallelism throughout:
//   load      the problem's block (strided: one field per lane and load) + the iterate, all loads in flight together
//   rows      kino_rows: defect / residual / constraint rows, AL multipliers, merit terms (lane = row)
//   bodies    momenta, accelerations and net forces of the bodies for the solved accelerations (lane = joint); velocity-product
//             matrices (lane = (body, column))
//   composite subtree sums of inertias / momenta / forces / velocity products: lane = scalar, the joints' values in registers
//   columns   kino_deriv_columns (smpc_kino_stage.h), then the knot: deriv_stage_knot / deriv_terminal_node (smpc_kino_kernels.h)
// Same LDS scratch type as the one-kernel path (KinoScratch<D, true>, 8 resident wavefronts per CU); same knot, bit for bit the same
// code from the derivative columns on.
#pragma once
#include "smpc_kino_lane.h"

namespace smpc
{
  template <class D, bool STREAM>
  SMPC_DEV void deriv2_one(const StageKernelArgs<D> & ka, int inst, int t);

  // Commit code of a field of the stream (host side, once per handle: the engine turns the field ids the tree kernel recorded into these):
  // (head index + 1) << 16 | (offset in doubles from the start of the derivative kernel's LDS scratch + 1); 0 parts: not a head field /
  // nothing to commit after the rows.  The same mapping as the tile form's commit phase below.
  template <class D>
  inline int deriv2_commit_code(const int * foot_joint, int id)
  {
    typedef KinoScratch<D, true> KS;
    typedef EvLayout<D> L;
    constexpr int NV = D::NV, NJ = D::NJ, NF = D::NF;
    constexpr int N_TREE = NV * 6 + NJ * 6 + NJ * 6 + NJ * 10;
    alignas(16) static char probe[sizeof(KS)]; // (addresses only: no object is created or read)
    KS * sc = reinterpret_cast<KS *>(probe);
    auto off = [&](const void * p) { return (int)((reinterpret_cast<const char *>(p) - probe) / (std::ptrdiff_t)sizeof(double)); };
    int h = -1, d = -1;
    if (id >= 0 && id < L::HEAD)
      h = id;
    else if (id >= L::O_S && id < L::O_S + L::N_DERIV)
    {
      const int i = id - L::O_S;
      if (i < N_TREE)
        d = off(sc->S) + i; // S | vel | acc | Ic contiguous
      else
      {
        const int r = i - N_TREE;
        if (r < 9 * NF)
          d = off(sc->oR) + foot_joint[r / 9] * 9 + r % 9;
        else if (r < 9 * NF + 3)
          d = off(sc->com) + r - 9 * NF;
        else if (r < 9 * NF + 9)
          d = off(sc->b0) + r - 9 * NF - 3;
        else if (r < 9 * NF + 45)
          d = off(sc->Agbi) + r - 9 * NF - 9;
        else
          d = off(static_cast<KinoScratchDerivPart<D> *>(sc)->Je3) + r - 9 * NF - 45; // Je3 | JeQ | Jq | Jl contiguous
      }
    }
    return ((h + 1) << 16) | (d + 1);
  }

  // grid = B * (H+1) (slots == 0) or slots * (H+1) walking the compacted list of instances that rejected the tentative full step
  // STREAM: the problem's fields arrive as one contiguous run in production order (Buffers::evd, ev_order) instead of the strided tile
  template <class D, bool STREAM = false>
  SMPC_DEV void deriv2_body(const StageKernelArgs<D> & ka, int block)
  {
    const int H = ka.b.H;
    int slot, t;
    xcd_problem(block, H, slot, t); // (XCD-aware: see xcd_problem)
    if (slot >= (ka.slots > 0 ? ka.slots : ka.b.B))
      return;
    const int count = ka.slots > 0 ? (slot < ka.slots ? ka.b.und_list[ka.b.B] : 0) : (slot < ka.b.B ? slot + 1 : 0); // (padding blocks: idle)
    const int stride = ka.slots > 0 ? ka.slots : ka.b.B;
    for (int m = slot; m < count; m += stride)
      deriv2_one<D, STREAM>(ka, ka.slots > 0 ? ka.b.und_list[m] : m, t);
  }

  template <class D, bool STREAM>
  SMPC_DEV void deriv2_one(const StageKernelArgs<D> & ka, int inst, int t)
  {
    typedef KinoScratch<D, true> KinoScratchT;
    typedef EvLayout<D> L;
    constexpr int NT = 64;
    constexpr int NV = D::NV, NX = D::NX, NDX = D::NDX, NU = D::NU, NC = D::NC, NF = D::NF, NA = D::NA, NJ = D::NJ;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const bool term = t == H;
    const DevModel<D> & mg = *b.model;
    SMPC_LDS(KinoScratchT, scs, 1);
    KinoScratchT & sc = scs[0];
    const DevModelSmall<D> & md = sc.ml;
    const int st = ring_slot(ka.head, t, R), sn = ring_slot(ka.head, term ? t : t + 1, R);
    const size_t ib = (size_t)inst * R;
    const double preg = b.scal[(size_t)inst * SC_N + SC_PREG];
    const LaneBlk blk = lane_block<D>(b, inst, t);

    StageIn<D> in;
    in.md = &mg;
    in.terminal = term;
    in.mask = term ? 0u : b.stages[t].mask;
    in.u_ref = term ? nullptr : b.stages[t].u_ref;
    in.x_tgt = term ? mg.x_term : b.stages[t].x_tgt;
    in.foot_ref = term ? nullptr : b.foot_ref + ((size_t)inst * H + t) * NF * 3;
    in.C_rows = term ? nullptr : b.lq + ((size_t)inst * H + t) * D::LQ_STRIDE + D::O_C + (size_t)NA * NDX;
    long long tprev = SMPC_CLOCK();
    in.prof = (b.dbg != nullptr && inst == 0 && t == 17 && ka.slots == 0) ? b.dbg : nullptr;
    in.tprev = &tprev;
    const double * vref = term ? nullptr : b.vref + (ib + st) * 6;

    // the row scratch lives in the block the derivative columns fill later (dh_dq ... cn)
    static_assert(RowsScratch<D>::N <= 3 * 6 * NV + NF * 3 * NV + 3 * NV + NDX, "row scratch inside the column block");
    const RowsScratch<D> rs(reinterpret_cast<double *>(static_cast<KinoScratchDerivPart<D> *>(&sc)) + offsetof(KinoScratchDerivPart<D>, dh_dq) / sizeof(double));
    // the block's derivative part: [S | vel | acc | I] is one run of tree block A, [Je3 | JeQ | Jq | Jl] one run of the derivative part
    constexpr int N_TREE = NV * 6 + NJ * 6 + NJ * 6 + NJ * 10;
    static_assert(L::O_vel == L::O_S + NV * 6 && L::O_acc == L::O_vel + NJ * 6 && L::O_I == L::O_acc + NJ * 6 && L::O_oRf == L::O_S + N_TREE,
                  "block layout: S | vel | acc | I contiguous");
    static_assert(offsetof(KinoScratchT, vel) == offsetof(KinoScratchT, S) + NV * 6 * sizeof(double)
                    && offsetof(KinoScratchT, acc) == offsetof(KinoScratchT, vel) + NJ * 6 * sizeof(double)
                    && offsetof(KinoScratchT, Ic) == offsetof(KinoScratchT, acc) + NJ * 6 * sizeof(double),
                  "scratch layout: S | vel | acc | Ic contiguous");
    static_assert(L::O_JeQ == L::O_Je3 + 9 && L::O_Jq == L::O_JeQ + 9 && L::O_Jl == L::O_Jq + 36, "block layout: SE(3) Jacobians contiguous");
    constexpr int O_REST = L::O_oRf - L::O_S; // fields after the tree run: oRf 9 NF | com 3 | dab 6 | Agbi 36 | Je3 .. Jl 90
    constexpr int NLOAD = STREAM ? EvStream<D>::NLOAD : (L::N_DERIV + NT - 1) / NT;
    const double * const pblk = STREAM ? b.evd + ((size_t)(b.ev_inst0 + inst) * (H + 1) + t) * EvStream<D>::STRIDE : nullptr;
    // the two runs as plain double pointers into the scratch (an index past the end of the first member array of a run must not be
    // visible to the optimiser as an out-of-bounds subscript)
    double * const run_tree = reinterpret_cast<double *>(&sc) + offsetof(KinoScratchT, S) / sizeof(double);
    double * const run_se3 = reinterpret_cast<double *>(static_cast<KinoScratchDerivPart<D> *>(&sc)) + offsetof(KinoScratchDerivPart<D>, Je3) / sizeof(double);

    SMPC_PL(double, plam, NT);
    SMPC_PL(double, lame, NT);
    SMPC_PL(double, pnu, NT);
    SMPC_PL(double, nue, NT);
    SMPC_PLA(double, vb, NT, NLOAD);
    SMPC_PLA(int, oi, NT, NLOAD); // STREAM: field id of each loaded element (EvLayout offsets; < HEAD: head of candidate 0)
    SMPC_LANES(NT)
    {
      // every global load of the block back to back (index clamped, one wait), then committed to LDS
      static_assert(NX + 9 <= NT && NU <= NT && NDX <= NT && NC <= NT, "one element per lane");
      const double vx = b.xs[(ib + st) * NX + (lane < NX ? lane : 0)];
      const double vxn = *(lane < NX ? b.xs + (ib + sn) * NX + lane : mg.w_frame + (lane < NX + 9 ? lane - NX : 0)); // (+ w_frame in spare lanes)
      const double vu = b.us[(ib + st) * NU + (lane < NU ? lane : 0)];
      const double vl = b.lams[(ib + st) * NDX + (lane < NDX ? lane : 0)];
      const double vn = b.vs[(ib + st) * NC + (lane < NC ? lane : 0)];
      const double vle = b.lams_e[(ib + st) * NDX + (lane < NDX ? lane : 0)];
      const double vne = b.vs_e[(ib + st) * NC + (lane < NC ? lane : 0)];
      const double vh = STREAM ? 0.0 : blk[L::O_head + (lane < L::HEAD ? lane : 0)];
      const double vxt = *((vref != nullptr && lane >= D::NQ && lane < D::NQ + 6) ? vref + (lane - D::NQ) : in.x_tgt + (lane < NX ? lane : 0));
      const double vur = term ? 0.0 : in.u_ref[lane < NU ? lane : 0];
      const double vfr = term ? 0.0 : in.foot_ref[lane < NF * 3 ? lane : 0];
      ModelLoad<D, NT> ml;
      ml.issue(&mg, lane);
      SMPC_SCHED_FENCE(); // (the scheduler otherwise sinks one of the model's loads below the strided ones, and its commit then waits for all)
      // (the per-joint part of the block is issued last and committed after the row phase, which does not need it: the commits below wait
      //  for the loads above only -- the counter is in order --, and the latency of these strided loads hides behind the row phase)
#pragma unroll
      for (int n = 0; n < NLOAD; n++)
      {
        if constexpr (STREAM)
        {
          const int idx = lane + n * NT < EvStream<D>::STRIDE ? lane + n * NT : EvStream<D>::STRIDE - 1; // (the last position is padding)
          SMPC_PLV(vb)[n] = pblk[idx];
          SMPC_PLV(oi)[n] = b.ev_order[idx];
        }
        else
          SMPC_PLV(vb)[n] = blk[L::O_S + (lane + n * NT < L::N_DERIV ? lane + n * NT : 0)];
      }
      SMPC_SCHED_FENCE();
      ml.commit(sc, lane);
      if (lane < NX)
      {
        sc.x[lane] = vx;
        rs.px[lane] = vx;
        rs.pxn[lane] = vxn;
        rs.xt[lane] = vxt;
      }
      if (lane < NU)
        rs.uref[lane] = vur;
      if (lane < NF * 3)
        rs.fref[lane] = vfr;
      else if (lane < NX + 9)
        sc.wframe_()[lane - NX] = vxn;
      if (lane < NU)
      {
        sc.u[lane] = term ? 0.0 : vu;
        rs.pu[lane] = term ? 0.0 : vu;
      }
      if (lane < NDX)
        sc.lam_next[lane] = term ? 0.0 : vl;
      if (lane < NC)
        sc.nu[lane] = term ? 0.0 : vn;
      if constexpr (STREAM)
      {
        // the head's 54 fields sit among the others (production order): the rows need them now, the rest is committed after the rows
#pragma unroll
        for (int n = 0; n < NLOAD; n++)
        {
          const int h = (SMPC_PLV(oi)[n] >> 16) - 1; // (commit code: deriv2_commit_code)
          const double v = SMPC_PLV(vb)[n];
          // two unconditional stores per element, the destinations by address select (a predicate per store costs a compare, two scalar
          // mask instructions and a branch: 44 of them per block): elements that are no head fields go to a dump slot (red[3], unused)
          static_assert(L::H_hg == L::H_footp + 3 * NF && L::H_hd == L::H_hg + 6 && L::HEAD == L::H_hd + 6, "footp | hg | hd end the head");
          double * const scb0 = reinterpret_cast<double *>(&sc);
          const int o_dump = (int)(&sc.red[3] - scb0), o_head = (int)(rs.head - scb0);
          const int o_fp = (int)(sc.footp - scb0) - L::H_footp, o_hg = (int)(sc.hg - scb0) - L::H_hg, o_hd = (int)(sc.hd - scb0) - L::H_hd;
          const int d1 = h >= 0 ? o_head + h : o_dump;
          int d2 = h >= L::H_footp ? h + o_fp : o_dump;
          d2 = h >= L::H_hg ? h + o_hg : d2;
          d2 = h >= L::H_hd ? h + o_hd : d2;
          scb0[d1] = v;
          scb0[d2] = v;
        }
      }
      else
      {
        rs.head[lane] = vh;
        if (lane >= L::H_footp && lane < L::H_footp + 3 * NF)
          sc.footp[lane - L::H_footp] = vh;
        if (lane >= L::H_hg && lane < L::H_hg + 6)
          sc.hg[lane - L::H_hg] = vh;
        if (lane >= L::H_hd && lane < L::H_hd + 6)
          sc.hd[lane - L::H_hd] = vh;
      }
      SMPC_PLV(plam) = term ? 0.0 : vl;
      SMPC_PLV(lame) = vle;
      SMPC_PLV(pnu) = term ? 0.0 : vn;
      SMPC_PLV(nue) = vne;
    }
    SMPC_LANES_END_WAVE
    static_assert(offsetof(KinoScratchDerivPart<D>, JeQ) == offsetof(KinoScratchDerivPart<D>, Je3) + 9 * sizeof(double)
                    && offsetof(KinoScratchDerivPart<D>, Jq) == offsetof(KinoScratchDerivPart<D>, JeQ) + 9 * sizeof(double)
                    && offsetof(KinoScratchDerivPart<D>, Jl) == offsetof(KinoScratchDerivPart<D>, Jq) + 36 * sizeof(double),
                  "scratch layout: SE(3) Jacobians contiguous");
    if (in.prof) prof_tick(in.prof, 15, tprev);

    // ---- rows: defect, residuals, constraint rows, multipliers, merit terms ----
    SMPC_PL(double, lamp_r, NT);
    SMPC_PL(double, vplus_r, NT);
    SMPC_PL(int, act_r, NT);
    SMPC_PL(double, wres_r, NT);
    SMPC_PL(double, wru_r, NT);
    double red[3];
    kino_rows<D, true>(rs, md, mg, sc.wframe_(), in.mask, term, plam, lame, pnu, nue, lamp_r, vplus_r, act_r, wres_r, wru_r, red);
    if (in.prof) prof_tick(in.prof, 30, tprev);
    if constexpr (STREAM)
    {
      double * const scb = reinterpret_cast<double *>(&sc);
      SMPC_LANES(NT)
      {
#pragma unroll
        for (int n = 0; n < NLOAD; n++)
        {
          const int d = (SMPC_PLV(oi)[n] & 0xFFFF) - 1; // destination in the scratch, from the commit code of the element
          if (d >= 0)
            scb[d] = SMPC_PLV(vb)[n];
        }
      }
      SMPC_LANES_END_WAVE
    }
    else
    {
    SMPC_LANES(NT)
    {
#pragma unroll
      for (int n = 0; n < NLOAD; n++)
      {
        const int i = lane + n * NT;
        if (i < N_TREE)
          run_tree[i] = SMPC_PLV(vb)[n]; // S | vel | acc | Ic (= the bodies' own inertias until the composite phase)
        else if (i < L::N_DERIV)
        {
          const int r = i - O_REST;
          if (r < 9 * NF)
            sc.oR[md.foot_joint[r / 9] * 9 + r % 9] = SMPC_PLV(vb)[n]; // (md: the LDS copy of the small model block -- a global read here would wait for every load in flight)
          else if (r < 9 * NF + 3)
            sc.com[r - 9 * NF] = SMPC_PLV(vb)[n];
          else if (r < 9 * NF + 9)
            sc.b0[r - 9 * NF - 3] = SMPC_PLV(vb)[n]; // (b0 carries the base's spatial acceleration here)
          else if (r < 9 * NF + 45)
            sc.Agbi[r - 9 * NF - 9] = SMPC_PLV(vb)[n];
          else
            run_se3[r - 9 * NF - 45] = SMPC_PLV(vb)[n]; // Je3 | JeQ | Jq | Jl
        }
      }
    }
    SMPC_LANES_END_WAVE
    }

    // ---- bodies: inertia copy, accelerations for the solved base acceleration, momenta, net forces (lane = joint) ----
    SMPC_LANES(NT)
    if (lane < NJ)
    {
      const int j = lane;
      const SI I = ldsi(&sc.Ic[j * 10]);
      stsi(&sc.I_()[j * 10], I);
      const SV v = ldsv(&sc.vel[j * 6]);
      const SV a = ldsv(&sc.acc[j * 6]) + ldsv(sc.b0);
      stsv(&sc.acc[j * 6], a);
      const SV h = I * v;
      stsv(&sc.hc[j * 6], h);
      stsv(&sc.Fc[j * 6], I * a + crf(v, h));
    }
    SMPC_LANES_END_WAVE
    // ---- per-body "velocity product" matrices  B_l y = v_l x* (I_l y) - I_l (v_l x y)  (lane = (body, column)) ----
    double * Bm = reinterpret_cast<double *>(&sc) + offsetof(KinoScratchEval<D>, cval) / sizeof(double); // (late block | WJl | JWJ)
    SMPC_LANES(NT)
    for (int idx = lane; idx < NJ * 6; idx += NT)
    {
      const int l = idx / 6, m = idx % 6;
      const SI Il = ldsi(&sc.I_()[l * 10]);
      const SV vl = ldsv(&sc.vel[l * 6]);
      const V3 e = mk3(m % 3 == 0, m % 3 == 1, m % 3 == 2), z = mk3(0, 0, 0);
      const SV y = m < 3 ? SV{e, z} : SV{z, e};
      const SV col = crf(vl, Il * y) - Il * crm(vl, y);
      double * dst = &Bm[l * 36 + m];
      dst[0] = col.l.x;
      dst[6] = col.l.y;
      dst[12] = col.l.z;
      dst[18] = col.a.x;
      dst[24] = col.a.y;
      dst[30] = col.a.z;
    }
    SMPC_LANES_END_WAVE
    if (in.prof) prof_tick(in.prof, 16, tprev);
    // ---- composites, leaf -> root: lane = one scalar of (Ic | hc | Fc | B); the joints' values of that scalar in registers, the
    //      tree walked with scalar branches on the (uniform) parent table: no LDS round trip per joint ----
    SMPC_LANES(NT)
    if (lane < 58)
    {
      double * base = lane < 10 ? sc.Ic : (lane < 16 ? sc.hc : (lane < 22 ? sc.Fc : Bm));
      const int stride = lane < 10 ? 10 : (lane < 22 ? 6 : 36);
      const int e = lane < 10 ? lane : (lane < 16 ? lane - 10 : (lane < 22 ? lane - 16 : lane - 22));
      double val[NJ];
#pragma unroll
      for (int j = 0; j < NJ; j++)
        val[j] = base[j * stride + e];
#pragma unroll
      for (int j = NJ - 1; j >= 1; j--)
      {
        const int par = md.parent[j];
#pragma unroll
        for (int pp = 0; pp < j; pp++)
          if (par == pp)
            val[pp] += val[j];
      }
#pragma unroll
      for (int j = 0; j < NJ; j++)
        base[j * stride + e] = val[j];
    }
    SMPC_LANES_END_WAVE
    if (in.prof) prof_tick(in.prof, 19, tprev);
    // ---- centroidal map columns ----
    SMPC_LANES(NT)
    if (lane < NV)
    {
      const int k = lane;
      const int j = k < 6 ? 0 : k - 5;
      const V3 com = ld3(sc.com);
      const SV c = ldsi(&sc.Ic[j * 10]) * ldsv(&sc.S[k * 6]);
      const V3 ang = c.a - cross(com, c.l);
      sc.Ag[0 * NV + k] = c.l.x;
      sc.Ag[1 * NV + k] = c.l.y;
      sc.Ag[2 * NV + k] = c.l.z;
      sc.Ag[3 * NV + k] = ang.x;
      sc.Ag[4 * NV + k] = ang.y;
      sc.Ag[5 * NV + k] = ang.z;
    }
    SMPC_LANES_END_WAVE
    if (in.prof) prof_tick(in.prof, 20, tprev);
    kino_deriv_columns<D>(sc, in, Bm);
    // ---- the rows' results into the late block (it held the velocity-product matrices until here) ----
    SMPC_LANES(NT)
    {
      if (lane < NDX)
      {
        sc.Wrx[lane] = SMPC_PLV(wres_r);
        sc.lamp[lane] = SMPC_PLV(lamp_r);
      }
      else if (lane < NDX + 6)
        sc.Whg[lane - NDX] = SMPC_PLV(wres_r);
      else if (lane < NDX + 12)
        sc.Whd[lane - NDX - 6] = SMPC_PLV(wres_r);
      else if (lane < NDX + 12 + 3 * NF)
        sc.Wrf[lane - NDX - 12] = SMPC_PLV(wres_r);
      if (lane < NU)
        sc.Wru[lane] = SMPC_PLV(wru_r);
      if (lane < NC)
      {
        sc.vplus[lane] = SMPC_PLV(vplus_r);
        sc.act[lane] = SMPC_PLV(act_r);
      }
      if (lane < 3)
        sc.red[lane] = lane == 0 ? red[0] : (lane == 1 ? red[1] : red[2]);
    }
    SMPC_LANES_END_WAVE
    double * parts = b.parts0 + ((size_t)inst * (H + 1) + t) * 4;
    if (term)
      deriv_terminal_node<D, false>(ka, sc, inst, t, preg, parts);
    else
      deriv_stage_knot<D, false>(ka, sc, in, inst, t, preg, parts, false, 0u, tprev);
  }
} // namespace smpc
