// smpc_cent_kernels.h -- the centroidal OCP (BASELINE config "Go2 centroidal (9-dim state), H = 50") as ONE fused
// kernel per control step: one wavefront per MPC instance runs the whole of MPC::iterate's solver part
// (reference src/mpc.cpp:189-218 over a CentroidalOCP, src/centroidal-dynamics.cpp:39-106, 306-316):
//   recede (ring advance, warm-start shift, Raibert foothold + Bezier swing references -> contact positions)
//   k ProxDDP iterations, each:
//     backward  t = H-1 .. 0 : stage evaluation + derivatives + LQ knot, built in LDS from the iterate; proximal Riccati
//                              step as two symmetric block sweeps on the FP64 matrix cores (wave_block_sweep):
//                              [[I + mu P, sqrt(mu) P, sqrt(mu) pt0], [., P, pt0]]      -> P~, p~        (9 pivots)
//                              [[R^, D^T, S^^T, r^], [D, -mu I, C, d], [., ., Q^, q^]]  -> K, k, Z, z, P_t, p_t
//                              (the stage KKT matrix is quasi-definite: the friction-cone rows have D != 0, so the
//                               multiplier block is pivoted explicitly instead of folding D^T D / mu into R^)
//     forward   t = 0 .. H-1 : (dx, du, dnu, dlam) and the directional derivative of the merit
//     line search            : lane = stage trial evaluations, Armijo backtracking (alpha = 1, 1/2, ..)
//   The stage matrices are 9 x 9 / 9 x 12: nothing but the per-stage gains leaves the CU between the phases of an
//   iteration (gains go through L2 to the forward pass).  Algorithm and constants: DESIGN.md 2
//   ("solver constants"); SURVEY App. B.1, B.4, B.5.
#pragma once
#include "smpc_riccati_kino.h"
#include "smpc_solver_kernels.h"

// phase timers (block 0 only, when the handle was created with SMPC_PHASE_PROFILE=1; a uniform branch otherwise).  They are
// always compiled in: the basic-block boundaries they add between the phase groups of the stage loop also happen to give
// the better instruction schedule (measured: -8 % kernel time against the same code without them).
#define CENT_FINE_TICK(n) prof_tick(dbg, n, tprev)
#define CENT_FINE_DBG dbg

namespace smpc
{
  // FS_: contact force size.  3 (point feet): this file; 6 (flat feet, wrench cones): the specialisation in smpc_cent6_kernels.h
  template <int NF_, int FS_ = 3>
  struct CentDims
  {
    static_assert(FS_ == 3, "6-D feet: CentDims<NF, 6> of smpc_cent6_kernels.h");
    static constexpr int NF = NF_, FS = 3;
    static constexpr int NX = 9, NDX = 9;
    static constexpr int NU = 3 * NF_;
    static constexpr int NC = 2 * NF_;            // rows 2f, 2f+1: friction-cone block of foot f
    static constexpr int NUP = (NU + 3) / 4 * 4;  // pivot panels are 4 wide: pad rows are decoupled unit pivots
    static constexpr int NCP = (NC + 3) / 4 * 4;
    static constexpr int VO = NUP;                // sweep 2 index space: u | nu | x | vector column
    static constexpr int XO = NUP + NCP;
    static constexpr int ZC = XO + 9;
    static constexpr int NP2 = (NUP + NCP) / 4;
    static constexpr int X1 = 12, Z1 = 21;        // sweep 1 index space: 9 pivots (+3 pad) | x | vector column
    static constexpr int LDM = 32;
    static constexpr int R1 = Z1 + 1, R2 = XO + 10; // rows of the two bordered matrices that are ever non-zero
    static constexpr int LD1 = Z1 + 1;              // row stride of the sweep-1 matrix (columns 0 .. Z1)
    static_assert(ZC < LDM && NU <= 16 && NC <= 16 && 3 * NU <= 64, "the stage KKT matrix must fit two 16 x 16 tile rows");
    // per (instance, stage) record written by the backward pass for the forward pass
    static constexpr int G_K = 0;                 // [K | k]  NU x 10
    static constexpr int GKS = 10;                // row stride of [K | k]
    static constexpr int G_Z = G_K + NU * 10;     // [Z | z]  NC x 10
    static constexpr int G_Pt = G_Z + NC * 10;    // P~_{t+1} 9 x 9
    static constexpr int G_pn = G_Pt + 81;        // p_{t+1}
    static constexpr int G_fs = G_pn + 9;         // sum of the active forces (A = I + dt [[0, I/m, 0], [0], [[fs]x, 0, 0]])
    static constexpr int G_r = G_fs + 3;          // act_f (p_f - c) per foot (B rows of the angular momentum)
    static constexpr int G_act = G_r + 3 * NF;    // contact flags as 0 / 1
    static constexpr int G_f = G_act + NF;        // mu (lam+ - lam)
    static constexpr int G_gx = G_f + 9;          // lx + A^T (2 lam+ - lam)
    static constexpr int G_gu = G_gx + 9;         // lu + B^T (2 lam+ - lam) + Cu^T vpd
    static constexpr int G_lpd = G_gu + NU;       // 2 lam+ - lam
    static constexpr int G_d = G_lpd + 9;         // mu (nu+ - nu)
    static constexpr int G_N = G_d + NC;
    static constexpr int G_REGS = (G_N + 63) / 64;
    static constexpr int G_STRIDE = G_REGS * 64;
    static_assert(G_STRIDE <= R1 * LD1, "the forward-pass record is staged over the sweep-1 matrix");
    // stage inputs gathered one stage ahead by the backward pass
    static constexpr int I_x = 0, I_xn = 9, I_l1 = 18, I_l1e = 27, I_l0 = 36, I_u = 45, I_v = I_u + NU, I_ve = I_v + NC;
    static constexpr int I_p = I_ve + NC, I_ur = I_p + 3 * NF, I_xt = I_ur + NU, I_N = I_xt + 9;
    static constexpr int I_REGS = (I_N + 63) / 64;
    static constexpr int LS_N = 10;
  };

  template <class D>
  struct CentDevModel
  {
    double mass, dt, mu, mu_fric, cone_eps, pad_;
    double Lfoot, Wfoot; // half length / half width of the sole (wrench cones of 6-D feet)
    double gravity[3];
    double w_com[9], w_lm[9], w_am[9], w_la[9], w_aa[9];
    double w_u[D::NU * D::NU];
    double foot_ref_p[D::NF][3];
  };
  template <class D>
  struct CentStage // shared by the batch (phase-aligned), one per horizon stage
  {
    unsigned mask, pad;
    double u_ref[D::NU];
    double x_tgt[9]; // [com_ref; h_ref; L_ref]
  };
  // (the pointers are the same for every foot type: CentBuffersBase is what the C ABI reads through CentEngineBase)
  struct CentBuffersBase
  {
    int B = 0, H = 0, R = 0;
    double *xs = nullptr, *us = nullptr, *vs = nullptr, *lams = nullptr; // rings [B][R][.]; lams[slot(t)] = lambda_{t+1}
    double *vs_e = nullptr, *lams_e = nullptr;
    double *dxs = nullptr, *dus = nullptr, *dvs = nullptr, *dlams = nullptr; // [B][H+1][9], [B][H][.] linear in t
    double * foot = nullptr;   // [B][H][NF*3] contact positions (MPC::setReferencePose -> contact map)
    double * ftraj = nullptr;  // [B][NF][6] swing start / end
    // velocity command per instance [B][6] and the momentum references [m v_lin; m v_ang] of every stage in the horizon
    // (ring [B][R][6]: what setVelocityBase wrote when the stage entered, src/mpc.cpp:312, src/centroidal-dynamics.cpp:227-239)
    double *vbase = nullptr, *vref = nullptr;
    double * gains = nullptr;  // [B][H][G_STRIDE]
    double * scal = nullptr;   // [B][SC_N]
    double * xdot01 = nullptr; // [B][2][9]
    double * zeros = nullptr;  // [64] zeros (address target of masked-out prefetch slots)
    double * dbg = nullptr;    // [64] optional in-kernel phase timers (block 0 only; null = off)
  };
  template <class D>
  struct CentBuffers : CentBuffersBase
  {
    CentStage<D> * stages = nullptr;
    CentDevModel<D> * model = nullptr;
  };
  template <class D>
  struct CentStepArgs
  {
    CentBuffers<D> b;
    int head;
    int shift;        // 1: control step (warm-start shift at the new head, references); 0: iterate in place (cold solve)
    int set_centres;  // 1: AL centres := current multipliers
    int reset_preg;   // 1: regularisation restarts (every solver run of the MPC)
    int iters;
    const double * X; // [B][nx_mb] measured multibody states (base pose for the Raibert heuristic)
    int nx_mb;
    const double * cstate; // [B][9]  getCentroidalState of the measured state (front-end kernel)
    const double * feet;   // [B][NF*3] measured foot positions (front-end kernel)
    int land[D::NF];
    int T_fly, T_contact;
    double swing_apex, timestep;
    double armijo_c1, reg_init, reg_min, reg_max, reg_inc, reg_dec;
  };

  template <class D>
  struct CentLds
  {
    static constexpr int NU = D::NU, NC = D::NC, NF = D::NF;
    CentDevModel<D> md;
    double in[D::I_N > 64 ? D::I_N : 64]; // stage inputs (I_* offsets); also the 64-entry reduction buffer between the passes
    double P[81], p[9], Pt[81], pt[9], pt0[9];
    double ABp[9 * D::LDM];          // [A B] in sweep-2 column order (B at 0, A at XO)
    // bordered matrices of the two sweeps (built here, swept in registers): zero-filled once, after that only the
    // structural non-zeros are rewritten per stage (results leave the accumulators directly, never through these)
    double M1[D::R1 * D::LD1], M2[D::R2 * D::LDM];
    double fs[3], ts[3], rf[3 * NF], act[NF], Cu[NC * 3], cact[NC];
    double f[9], dvec[NC], lpd[9], vpd[NC];
    double ru[NU], rx[9], rla[3], N[NF * 9], G[9], wla[3], waa[3], wrx[9], wu[NU];
    double lx[9], lu[NU], q[9], r[NU], gxp[9], gu[NU];
    double dx[9], du[NU], y[9], w[9];
    double sc[16];
  };

  SMPC_HD double skew_el(V3 v, int a, int b)
  {
    // [v]x (a, b)
    if (a == b)
      return 0.0;
    const int k = 3 - a - b; // the remaining axis
    const double c = k == 0 ? v.x : (k == 1 ? v.y : v.z);
    const bool pos = (a == 0 && b == 2) || (a == 1 && b == 0) || (a == 2 && b == 1);
    return pos ? c : -c;
  }

  // deterministic wave reduction of a per-lane value through LDS: red[lane] then lane 0 folds in lane order
  template <bool MAX>
  SMPC_DEV double fold64(const double * red)
  {
    double a = red[0];
    for (int i = 1; i < 64; i++)
      a = MAX ? fmax(a, red[i]) : a + red[i];
    return a;
  }

  // Stage merit terms at the trial point  w + alpha dw  (each lane evaluates one whole stage: used by the line search;
  // the point is formed entry by entry from the iterate and the step, so no per-lane copy of it is kept in registers).
  // Returns cost, penalty and primal infeasibility of stage t; xdot optionally.
  template <class D>
  struct CentTrial
  {
    const double *x, *dx, *xn, *dxn, *u, *du, *v, *dv, *l1, *dl; // iterate and step (dxn = dx of stage t + 1)
    const double *ve, *l1e, *p, *uref, *xtgt, *href; // xtgt: CoM reference (shared), href: momentum references (per instance)
    double alpha;
    unsigned mask;
  };
  template <class D>
  SMPC_DEV void cent_stage_merit(const CentDevModel<D> & md, const CentTrial<D> & q, double & cost, double & pen, double & prim, double * xdot, double * ru_lds)
  {
    constexpr int NF = D::NF, NU = D::NU;
    const double al = q.alpha;
    const double imu = 1.0 / md.mu, imass = 1.0 / md.mass;
    auto X = [&](int i) { return q.x[i] + al * q.dx[i]; };
    const V3 c = mk3(X(0), X(1), X(2)), h = mk3(X(3), X(4), X(5)), L = mk3(X(6), X(7), X(8));
    V3 fs = mk3(0, 0, 0), ts = mk3(0, 0, 0);
    pen = 0.0;
    prim = 0.0;
    double cu = 0.0;
#pragma unroll
    for (int f = 0; f < NF; f++)
    {
      const bool on = (q.mask >> f) & 1u;
      const V3 F = mk3(q.u[3 * f] + al * q.du[3 * f], q.u[3 * f + 1] + al * q.du[3 * f + 1], q.u[3 * f + 2] + al * q.du[3 * f + 2]);
      // control residual to a lane-private LDS strip: the NU x NU weight is applied by a rolled loop below (a fully
      // unrolled form keeps all of W_u live and spills)
      ru_lds[3 * f] = F.x - q.uref[3 * f];
      ru_lds[3 * f + 1] = F.y - q.uref[3 * f + 1];
      ru_lds[3 * f + 2] = F.z - q.uref[3 * f + 2];
      double vp0 = 0.0, vp1 = 0.0;
      if (on)
      {
        fs = fs + F;
        ts = ts + cross(ld3(q.p + 3 * f) - c, F);
        const double c0 = -F.z + md.cone_eps, c1 = F.x * F.x + F.y * F.y - md.mu_fric * md.mu_fric * F.z * F.z;
        const double z0 = c0 + md.mu * q.ve[2 * f], z1 = c1 + md.mu * q.ve[2 * f + 1];
        vp0 = (z0 - fmin(z0, 0.0)) * imu;
        vp1 = (z1 - fmin(z1, 0.0)) * imu;
        prim = fmax(prim, fmax(fmax(c0, 0.0), fmax(c1, 0.0)));
      }
      const double d0 = vp0 - (q.v[2 * f] + al * q.dv[2 * f]), d1 = vp1 - (q.v[2 * f + 1] + al * q.dv[2 * f + 1]);
      pen += 0.5 * md.mu * (vp0 * vp0 + d0 * d0);
      pen += 0.5 * md.mu * (vp1 * vp1 + d1 * d1);
    }
    const V3 g = ld3(md.gravity);
    double xd[9];
    xd[0] = h.x * imass;
    xd[1] = h.y * imass;
    xd[2] = h.z * imass;
    xd[3] = md.mass * g.x + fs.x;
    xd[4] = md.mass * g.y + fs.y;
    xd[5] = md.mass * g.z + fs.z;
    xd[6] = ts.x;
    xd[7] = ts.y;
    xd[8] = ts.z;
#pragma unroll
    for (int i = 0; i < 9; i++)
    {
      const double e = X(i) + md.dt * xd[i] - (q.xn[i] + al * q.dxn[i]);
      const double lp = q.l1e[i] + e * imu, dl = lp - (q.l1[i] + al * q.dl[i]);
      pen += 0.5 * md.mu * (lp * lp + dl * dl);
      prim = fmax(prim, fabs(e));
      if (xdot)
        xdot[i] = xd[i];
    }
    auto quad3 = [](const double * W, V3 r) {
      const V3 Wr = ldm3(W) * r;
      return 0.5 * dot(r, Wr);
    };
    cost = quad3(md.w_com, c - ld3(q.xtgt));
#pragma unroll 1
    for (int i = 0; i < NU; i++)
    {
      double wr = 0.0;
#pragma unroll 4
      for (int j = 0; j < NU; j++)
        wr += md.w_u[i * NU + j] * ru_lds[j];
      cu += ru_lds[i] * wr;
    }
    cost += 0.5 * cu;
    cost += quad3(md.w_lm, h - ld3(q.href));
    cost += quad3(md.w_am, L - ld3(q.href + 3));
    cost += quad3(md.w_la, g + imass * fs);
    cost += quad3(md.w_aa, ts);
  }

  // address of stage-input slot s of stage t (backward-pass prefetch); masked-out slots read a zero
  template <class D>
  SMPC_DEV const double * cent_in_addr(const CentBuffers<D> & b, size_t inst, int head, int t, int s)
  {
    constexpr int NU = D::NU, NC = D::NC, NF = D::NF;
    const int R = b.R, H = b.H;
    const size_t ib = inst * R;
    const int st = ring_slot(head, t, R), st1 = ring_slot(head, t + 1, R), stm = ring_slot(head, t > 0 ? t - 1 : 0, R);
    const double * a = b.zeros;
    if (s < D::I_xn)
      a = b.xs + (ib + st) * 9 + s;
    else if (s < D::I_l1)
      a = b.xs + (ib + st1) * 9 + (s - D::I_xn);
    else if (s < D::I_l1e)
      a = b.lams + (ib + st) * 9 + (s - D::I_l1);
    else if (s < D::I_l0)
      a = b.lams_e + (ib + st) * 9 + (s - D::I_l1e);
    else if (s < D::I_u)
      a = t > 0 ? b.lams + (ib + stm) * 9 + (s - D::I_l0) : b.zeros;
    else if (s < D::I_v)
      a = b.us + (ib + st) * NU + (s - D::I_u);
    else if (s < D::I_ve)
      a = b.vs + (ib + st) * NC + (s - D::I_v);
    else if (s < D::I_p)
      a = b.vs_e + (ib + st) * NC + (s - D::I_ve);
    else if (s < D::I_ur)
      a = b.foot + (inst * H + t) * (3 * NF) + (s - D::I_p);
    else if (s < D::I_xt)
      a = b.stages[t].u_ref + (s - D::I_ur);
    else if (s < D::I_xt + 3)
      a = b.stages[t].x_tgt + (s - D::I_xt);
    else if (s < D::I_N)
      a = b.vref + (ib + st) * 6 + (s - D::I_xt - 3); // momentum references are per instance
    return a;
  }

  template <class D>
  SMPC_DEV void cent_step_body(const CentStepArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NU = D::NU, NC = D::NC, NF = D::NF, LDM = D::LDM;
    constexpr int NUP = D::NUP, VO = D::VO, XO = D::XO, ZC = D::ZC, X1 = D::X1, Z1 = D::Z1;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head;
    const size_t inst = (size_t)block;
    const size_t ib = inst * R;
    SMPC_LDS(CentLds<D>, ldsv, 1);
    CentLds<D> & s = ldsv[0];
    double * const rec = s.M1;  // forward pass / recede scratch: record of the current stage (the backward pass is over)
    double * const red = s.in;  // reductions happen between the passes
    // pivot rows / U rows of the block sweeps live in the [A B] area: sweep 1 runs before [A B] of the stage is built,
    // sweep 2 after its last reader (the operand loads of the products)
    static_assert(9 * LDM >= 8 * LDM, "sweep scratch inside ABp");
    double * const prow = s.ABp;
    double * const urow = s.ABp + 4 * LDM;
    double * dbg = block == 0 ? b.dbg : nullptr;
    long long tprev = SMPC_CLOCK();
    double * gsc = b.scal + inst * SC_N;

    // ---- model constants -> LDS ----
    SMPC_LANES(NT)
    {
      constexpr int N = (int)(sizeof(CentDevModel<D>) / sizeof(double));
      const alias_double * src = reinterpret_cast<const alias_double *>(b.model);
      alias_double * dst = reinterpret_cast<alias_double *>(&s.md);
      for (int i = lane; i < N; i += NT)
        dst[i] = src[i];
      if (lane < 16)
        s.sc[lane] = lane == SC_PREG && !ka.reset_preg ? gsc[SC_PREG] : 0.0;
    }
    SMPC_LANES_END_WAVE
    const CentDevModel<D> & md = s.md;
    const double mu = md.mu, dt = md.dt, mass = md.mass;
    const double imu = 1.0 / mu, imass = 1.0 / mass; // (the stage loop multiplies: an FP64 division is ~35 instructions)

    // ---- recede: warm-start shift on the ring (src/mpc.cpp:201-207), references (src/mpc.cpp:278-309) ----
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

    SMPC_PL(double, acc_cost, NT);
    SMPC_PL(double, acc_pen, NT);
    SMPC_PL(double, acc_prim, NT);
    SMPC_PL(double, acc_dual, NT);
    SMPC_PL(double, acc_dphi, NT);
    SMPC_PLA(double, pin, NT, D::I_REGS);
    SMPC_PLA(double, prec, NT, D::G_REGS);
    SMPC_ACC(macc, NT, 3);
    SMPC_ACC(tacc, NT, 2);
    SMPC_PLA(double, aop, NT, 6);
    SMPC_PLA(double, pop, NT, 3);
    SMPC_PLA(double, top, NT, 6);

    for (int it = 0; it < ka.iters; it++)
    {
      const double preg = s.sc[SC_PREG] > 0.0 ? s.sc[SC_PREG] : ka.reg_init;
      // sweep templates: zero pattern and the decoupled unit pivots of the pad rows (the forward pass and the line search
      // of the previous iteration used these areas as scratch)
      SMPC_LANES(NT)
      {
#pragma unroll
        for (int n = 0; n < (D::R1 * D::LD1 + NT - 1) / NT; n++)
          if (lane + n * NT < D::R1 * D::LD1)
            s.M1[lane + n * NT] = 0.0;
#pragma unroll
        for (int n = 0; n < (D::R2 * LDM + NT - 1) / NT; n++)
          if (lane + n * NT < D::R2 * LDM)
            s.M2[lane + n * NT] = 0.0;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        if (lane >= 9 && lane < X1)
          s.M1[lane * D::LD1 + lane] = 1.0;
        if ((lane >= NU && lane < NUP) || (lane >= VO + NC && lane < XO))
          s.M2[lane * LDM + lane] = 1.0;
      }
      SMPC_LANES_END_WAVE
      // =====================================================================================
      // backward pass
      // =====================================================================================
      SMPC_LANES(NT)
      {
        SMPC_PLV(acc_cost) = 0.0;
        SMPC_PLV(acc_pen) = 0.0;
        SMPC_PLV(acc_prim) = 0.0;
        SMPC_PLV(acc_dual) = 0.0;
        // terminal node: P = Lxx_N + preg I, p = lx_N - lambda_H ; prefetch of stage H-1
        const double * xH = b.xs + (ib + ring_slot(head, H, R)) * 9;
        const double * lH = b.lams + (ib + ring_slot(head, H - 1, R)) * 9;
#pragma unroll
        for (int n = 0; n < D::I_REGS; n++)
          SMPC_PLV(pin)[n] = *cent_in_addr<D>(b, inst, head, H - 1, lane + n * NT);
        for (int idx = lane; idx < 81; idx += NT)
        {
          const int i = idx / 9, j = idx % 9;
          double v = i == j ? preg : 0.0;
          if (i >= 3 && i / 3 == j / 3)
          {
            const double * W = i < 6 ? md.w_lm : md.w_am;
            v += W[(i % 3) * 3 + j % 3];
          }
          s.P[idx] = v;
        }
        if (lane < 9)
        {
          double g = 0.0;
          if (lane >= 3)
          {
            const double * W = lane < 6 ? md.w_lm : md.w_am;
            const int bo = lane < 6 ? 3 : 6;
            for (int j = 0; j < 3; j++)
              g += W[(lane - bo) * 3 + j] * xH[bo + j];
          }
          const double qn = g - lH[lane];
          s.p[lane] = qn;
          SMPC_PLV(acc_dual) = fabs(qn);
        }
        if (lane == 9)
          SMPC_PLV(acc_cost) = 0.5 * dot(ld3(xH + 3), ldm3(md.w_lm) * ld3(xH + 3)) + 0.5 * dot(ld3(xH + 6), ldm3(md.w_am) * ld3(xH + 6));
      }
      SMPC_LANES_END_WAVE

      for (int t = H - 1; t >= 0; t--)
      {
        const unsigned mask = b.stages[t].mask;
        double * g = b.gains + (inst * H + t) * D::G_STRIDE;
        // ---- sweep 1 first: P~ = (I + mu P)^-1 P depends on P_{t+1} only, so it runs while the record stores of the previous
        //      stage are still being acknowledged (the stage inputs are committed after it; gfx950 has one counter for loads
        //      and stores).  M1 = [[I + mu P, sqrt(mu) P], [., P]] (pad pivots 9..11: unit diagonal); the vector part follows
        //      from (I + mu P)^-1 = I - mu P~ once the defect is known:  p~ = pt0 - mu P~ pt0 ----
        SMPC_LANES(NT)
        {
          const double smu = sqrt(mu);
#pragma unroll
          for (int n = 0; n < 2; n++)
          {
            const int idx = lane + n * NT;
            if (idx < 81)
            {
              const int i = idx / 9, j = idx % 9;
              const double pv = s.P[idx];
              s.M1[i * D::LD1 + j] = mu * pv + (i == j ? 1.0 : 0.0);
              s.M1[i * D::LD1 + X1 + j] = smu * pv;
              s.M1[(X1 + j) * D::LD1 + i] = smu * pv;
              s.M1[(X1 + i) * D::LD1 + X1 + j] = pv;
            }
          }
        }
        SMPC_LANES_END_WAVE
        CENT_FINE_TICK(5);
        // ---- sweep 1: P~, p~ ----
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
          for (int I = 0; I < 2; I++)
#pragma unroll
            for (int J = I; J < 2; J++)
#pragma unroll
              for (int vv = 0; vv < 4; vv++)
              {
                const int r = 16 * I + lr + 4 * vv;
                SMPC_ACCV(macc, tix<2>(I, J), vv) = r < D::R1 && 16 * J + lc < D::LD1 ? s.M1[r * D::LD1 + 16 * J + lc] : 0.0;
              }
        }
        SMPC_LANES_END_WAVE
        CENT_FINE_TICK(6);
        wave_block_sweep<NT, 2, false, 0, 3>(macc, prow, urow, CENT_FINE_DBG, tprev);
        // P~ (upper triangle of the Schur block is authoritative, mirrored) straight out of the accumulators
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
          for (int I = 0; I < 2; I++)
#pragma unroll
            for (int J = I; J < 2; J++)
#pragma unroll
              for (int vv = 0; vv < 4; vv++)
              {
                const int r = 16 * I + lr + 4 * vv - X1, c = 16 * J + lc - X1;
                const double a = SMPC_ACCV(macc, tix<2>(I, J), vv);
                if (r >= 0 && r < 9 && c >= r && c < 9)
                {
                  s.Pt[r * 9 + c] = a;
                  s.Pt[c * 9 + r] = a;
                }
              }
        }
        SMPC_LANES_END_WAVE
        CENT_FINE_TICK(7);
        // ---- stage inputs -> LDS ; prefetch of stage t-1 ----
        SMPC_LANES(NT)
        {
#pragma unroll
          for (int n = 0; n < D::I_REGS; n++)
            if (lane + n * NT < D::I_N)
              s.in[lane + n * NT] = SMPC_PLV(pin)[n];
          if (t > 0)
          {
#pragma unroll
            for (int n = 0; n < D::I_REGS; n++)
              SMPC_PLV(pin)[n] = *cent_in_addr<D>(b, inst, head, t - 1, lane + n * NT);
          }
          if (lane < 9)
            g[D::G_pn + lane] = s.p[lane];
        }
        SMPC_LANES_END_WAVE
        CENT_FINE_TICK(0);
        CENT_FINE_TICK(15); // cost of a tick itself
        const double *x = s.in + D::I_x, *xn = s.in + D::I_xn, *l1 = s.in + D::I_l1, *l1e = s.in + D::I_l1e, *l0 = s.in + D::I_l0;
        const double *u = s.in + D::I_u, *v = s.in + D::I_v, *ve = s.in + D::I_ve, *pp = s.in + D::I_p, *uref = s.in + D::I_ur, *xtgt = s.in + D::I_xt;
        // ---- point quantities: forces, lever arms, defect, multiplier estimates ----
        SMPC_LANES(NT)
        {
          const V3 c = ld3(x);
          V3 fs = mk3(0, 0, 0), ts = mk3(0, 0, 0);
#pragma unroll
          for (int f = 0; f < NF; f++)
            if ((mask >> f) & 1u)
            {
              const V3 F = ld3(u + 3 * f);
              fs = fs + F;
              ts = ts + cross(ld3(pp + 3 * f) - c, F);
            }
          if (lane < 9)
          {
            const V3 gv = ld3(md.gravity);
            const int k = lane % 3;
            const double hk = k == 0 ? x[3] : (k == 1 ? x[4] : x[5]);
            const double fk = k == 0 ? fs.x : (k == 1 ? fs.y : fs.z);
            const double tk = k == 0 ? ts.x : (k == 1 ? ts.y : ts.z);
            const double gk = k == 0 ? gv.x : (k == 1 ? gv.y : gv.z);
            const double xd = lane < 3 ? hk * imass : (lane < 6 ? mass * gk + fk : tk);
            const double e = x[lane] + dt * xd - xn[lane];
            const double lp = l1e[lane] + e * imu, dl = lp - l1[lane];
            s.f[lane] = mu * dl;
            s.lpd[lane] = 2.0 * lp - l1[lane];
            SMPC_PLV(acc_pen) += 0.5 * mu * (lp * lp + dl * dl);
            SMPC_PLV(acc_prim) = fmax(SMPC_PLV(acc_prim), fabs(e));
            s.rx[lane] = x[lane] - xtgt[lane];
            if (lane < 3)
            {
              s.fs[lane] = fk;
              s.ts[lane] = tk;
              s.rla[lane] = gk + fk * imass;
            }
          }
          if (lane >= 16 && lane < 16 + NC)
          {
            const int row = lane - 16, f = row / 2;
            const bool on = (mask >> f) & 1u;
            const V3 F = ld3(u + 3 * f);
            double vp = 0.0, act = 0.0, c0 = 0.0, c1 = 0.0, c2 = 0.0;
            if (on)
            {
              const bool cone = row & 1;
              const double cv = cone ? F.x * F.x + F.y * F.y - md.mu_fric * md.mu_fric * F.z * F.z : -F.z + md.cone_eps;
              const double z = cv + mu * ve[row];
              const double proj = fmin(z, 0.0);
              vp = (z - proj) * imu;
              act = z != proj ? 1.0 : 0.0;
              SMPC_PLV(acc_prim) = fmax(SMPC_PLV(acc_prim), fmax(cv, 0.0));
              c0 = cone ? 2.0 * F.x : 0.0;
              c1 = cone ? 2.0 * F.y : 0.0;
              c2 = cone ? -2.0 * md.mu_fric * md.mu_fric * F.z : -1.0;
            }
            const double dv = vp - v[row];
            s.dvec[row] = mu * dv;
            s.vpd[row] = act != 0.0 ? 2.0 * vp - v[row] : 0.0;
            s.cact[row] = act;
            s.Cu[row * 3 + 0] = c0;
            s.Cu[row * 3 + 1] = c1;
            s.Cu[row * 3 + 2] = c2;
            SMPC_PLV(acc_pen) += 0.5 * mu * (vp * vp + dv * dv);
          }
          if (lane >= 32 && lane < 32 + NU)
          {
            const int i = lane - 32;
            s.ru[i] = u[i] - uref[i];
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < NU; j++)
              a += md.w_u[i * NU + j] * (u[j] - uref[j]);
            s.wu[i] = a;
          }
          if (lane >= 48 && lane < 48 + NF)
          {
            const int f = lane - 48;
            const bool on = (mask >> f) & 1u;
            const V3 r = ld3(pp + 3 * f) - c;
            s.act[f] = on ? 1.0 : 0.0;
            st3(&s.rf[3 * f], on ? r : mk3(0, 0, 0));
          }
        }
        SMPC_LANES_END_WAVE
        CENT_FINE_TICK(1);
        // ---- small products: N_f = W_aa [r_f]x, G = W_aa [fs]x, W r of every residual ; [A B] ----
        static_assert(NF * 9 + 9 + 15 <= NT, "one small product per lane");
        SMPC_LANES(NT)
        {
          const V3 fsv = ld3(s.fs);
          if (lane < NF * 9 + 9)
          {
            // lanes 0 .. 9 NF - 1: N_f(i, j) ; next 9 lanes: G(i, j)
            const bool isN = lane < NF * 9;
            const int e = isN ? lane % 9 : lane - NF * 9, i = e / 3, j = e % 3;
            const V3 r = isN ? ld3(&s.rf[3 * (lane / 9)]) : fsv;
            const double a = md.w_aa[i * 3 + 0] * skew_el(r, 0, j) + md.w_aa[i * 3 + 1] * skew_el(r, 1, j) + md.w_aa[i * 3 + 2] * skew_el(r, 2, j);
            if (isN)
              s.N[lane] = a;
            else
              s.G[e] = a;
          }
          else if (lane < NF * 9 + 9 + 15)
          {
            const int e = lane - NF * 9 - 9; // 0..2 la, 3..5 aa, 6..14 x blocks
            const int k = e % 3;
            const double * W = e < 3 ? md.w_la : (e < 6 ? md.w_aa : (e < 9 ? md.w_com : (e < 12 ? md.w_lm : md.w_am)));
            const double * rv = e < 3 ? s.rla : (e < 6 ? s.ts : s.rx + (e - 6) / 3 * 3);
            const double a = W[k * 3] * rv[0] + W[k * 3 + 1] * rv[1] + W[k * 3 + 2] * rv[2];
            double * dst = e < 3 ? s.wla + e : (e < 6 ? s.waa + (e - 3) : s.wrx + (e - 6));
            *dst = a;
          }
          // [A B] in sweep-2 column order: zero, then the structural non-zeros
          for (int idx = lane; idx < 9 * LDM; idx += NT)
            s.ABp[idx] = 0.0;
        }
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        {
          const V3 fsv = ld3(s.fs);
          if (lane < NU)
          { // B: rows 3..5 <- dt act I ; rows 6..8 <- dt [r_f]x (r_f = 0 for a foot in the air)
            const int f = lane / 3, jj = lane % 3;
            const V3 r = ld3(&s.rf[3 * f]);
            s.ABp[(3 + jj) * LDM + lane] = dt * s.act[f];
#pragma unroll
            for (int k = 0; k < 3; k++)
              if (k != jj)
                s.ABp[(6 + k) * LDM + lane] = dt * skew_el(r, k, jj);
          }
          else if (lane >= 16 && lane < 25)
          { // A = I + dt [[0, I/m, 0], [0, 0, 0], [[fs]x, 0, 0]], column i = lane - 16
            const int i = lane - 16;
            s.ABp[i * LDM + XO + i] = 1.0;
            if (i >= 3 && i < 6)
              s.ABp[(i - 3) * LDM + XO + i] = dt * imass;
            if (i < 3)
            {
#pragma unroll
              for (int k = 0; k < 3; k++)
                if (k != i)
                  s.ABp[(6 + k) * LDM + XO + i] = dt * skew_el(fsv, k, i);
            }
          }
        }
        SMPC_LANES_END_WAVE
        CENT_FINE_TICK(2);
        // ---- cost gradient, cost ; pt0 = p + P f ----
        SMPC_LANES(NT)
        {
          const V3 fsv = ld3(s.fs), waa = ld3(s.waa);
          if (lane < 9)
          {
            double a = s.wrx[lane];
            if (lane < 3)
            {
              const V3 w = cross(waa, fsv); // [fs]x^T w = w x fs
              a += lane == 0 ? w.x : (lane == 1 ? w.y : w.z);
            }
            s.lx[lane] = a;
            double pa = s.p[lane];
#pragma unroll
            for (int j = 0; j < 9; j++)
              pa += s.P[lane * 9 + j] * s.f[j];
            s.pt0[lane] = pa;
            // cost terms are accumulated where their factors are at hand (per-lane partial sums, folded once per iteration)
            SMPC_PLV(acc_cost) += 0.5 * s.rx[lane] * s.wrx[lane];
            if (lane < 3)
              SMPC_PLV(acc_cost) += 0.5 * (s.rla[lane] * s.wla[lane] + s.ts[lane] * s.waa[lane]);
          }
          if (lane >= 16 && lane < 16 + NU)
          {
            const int j = lane - 16, f = j / 3, k = j % 3;
            const V3 w = cross(waa, ld3(&s.rf[3 * f])); // [r]x^T w (r = 0 for a foot in the air)
            s.lu[j] = s.wu[j] + s.act[f] * s.wla[k] * imass + (k == 0 ? w.x : (k == 1 ? w.y : w.z));
            SMPC_PLV(acc_cost) += 0.5 * s.ru[j] * s.wu[j];
          }
        }
        SMPC_LANES_END_WAVE
        CENT_FINE_TICK(3);
        // ---- knot vectors q, r and the merit-gradient pieces ; sweep-1 matrix ----
        SMPC_LANES(NT)
        {
          if (lane < 9)
          {
            double a = 0.0, gx = 0.0;
#pragma unroll
            for (int k = 0; k < 9; k++)
            {
              a += s.ABp[k * LDM + XO + lane] * l1[k];
              gx += s.ABp[k * LDM + XO + lane] * s.lpd[k];
            }
            double pa = 0.0;
#pragma unroll
            for (int k = 0; k < 9; k++)
              pa += s.Pt[lane * 9 + k] * s.pt0[k];
            s.pt[lane] = s.pt0[lane] - mu * pa; // p~ = (I - mu P~) pt0
            const double q = t > 0 ? s.lx[lane] + a - l0[lane] : 0.0; // x_0 is fixed (force_initial_condition)
            s.q[lane] = q;
            s.gxp[lane] = s.lx[lane] + gx;
            SMPC_PLV(acc_dual) = fmax(SMPC_PLV(acc_dual), fabs(q));
          }
          if (lane >= 16 && lane < 16 + NU)
          {
            const int j = lane - 16, f = j / 3, k = j % 3;
            double a = 0.0, gu = 0.0;
#pragma unroll
            for (int kk = 0; kk < 9; kk++)
            {
              a += s.ABp[kk * LDM + j] * l1[kk];
              gu += s.ABp[kk * LDM + j] * s.lpd[kk];
            }
            for (int rr = 0; rr < 2; rr++)
            {
              a += s.Cu[(2 * f + rr) * 3 + k] * v[2 * f + rr];
              gu += s.Cu[(2 * f + rr) * 3 + k] * s.vpd[2 * f + rr];
            }
            const double r = s.lu[j] + a;
            s.r[j] = r;
            s.gu[j] = s.lu[j] + gu;
            SMPC_PLV(acc_dual) = fmax(SMPC_PLV(acc_dual), fabs(r));
          }
        }
        SMPC_LANES_END_WAVE
        CENT_FINE_TICK(4);
        // ---- sweep-2 matrix: cost / constraint part ----
        SMPC_LANES(NT)
        {
          const V3 fsv = ld3(s.fs);
          // R = Luu + preg I
#pragma unroll
          for (int n = 0; n < (NU * NU + NT - 1) / NT; n++)
          {
            const int idx = lane + n * NT;
            if (idx >= NU * NU)
              break;
            const int i = idx / NU, j = idx % NU;
            const int fa = i / 3, ia = i % 3, fb = j / 3, jb = j % 3;
            const V3 ra = ld3(&s.rf[3 * fa]);
            double a = md.w_u[idx] + (i == j ? preg : 0.0) + s.act[fa] * s.act[fb] * md.w_la[ia * 3 + jb] * (imass * imass);
#pragma unroll
            for (int k = 0; k < 3; k++)
              a += skew_el(ra, k, ia) * s.N[fb * 9 + k * 3 + jb];
            s.M2[i * LDM + j] = a;
          }
          // Q = Lxx + preg I (block diagonal)
#pragma unroll
          for (int n = 0; n < 2; n++)
          {
            const int idx = lane + n * NT;
            if (idx >= 81)
              break;
            const int xl = idx / 9, xi = idx % 9;
            double a = xl == xi ? preg : 0.0;
            if (xl / 3 == xi / 3)
            {
              const double * W = xl < 3 ? md.w_com : (xl < 6 ? md.w_lm : md.w_am);
              a += W[(xl % 3) * 3 + xi % 3];
              if (xl < 3)
              {
#pragma unroll
                for (int k = 0; k < 3; k++)
                  a += skew_el(fsv, k, xl) * s.G[k * 3 + xi];
              }
            }
            s.M2[(XO + xl) * LDM + XO + xi] = a;
          }
          // S^T: Lxu(xi, j), only the com rows
          if (lane < 3 * NU)
          {
            const int idx = lane;
            const int xi = idx / NU, j = idx % NU, f = j / 3, jb = j % 3;
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++)
              a += skew_el(fsv, k, xi) * s.N[f * 9 + k * 3 + jb];
            s.M2[j * LDM + XO + xi] = a;
            s.M2[(XO + xi) * LDM + j] = a;
          }
          // D^T (active rows), -mu I, unit pad pivots
          if (lane < NC * 3)
          {
            const int row = lane / 3, k = lane % 3, f = row / 2;
            const double a = s.cact[row] != 0.0 ? s.Cu[row * 3 + k] : 0.0;
            s.M2[(3 * f + k) * LDM + VO + row] = a;
            s.M2[(VO + row) * LDM + 3 * f + k] = a;
          }
          else if (lane < NC * 3 + NC)
          {
            const int row = lane - NC * 3;
            s.M2[(VO + row) * LDM + VO + row] = -mu;
          }
          // vector column
          if (lane >= 32 && lane < 32 + NU)
          {
            s.M2[(lane - 32) * LDM + ZC] = s.r[lane - 32];
            s.M2[ZC * LDM + lane - 32] = s.r[lane - 32];
          }
          else if (lane >= 32 + NU && lane < 32 + NU + NC)
          {
            const int row = lane - 32 - NU;
            s.M2[(VO + row) * LDM + ZC] = s.dvec[row];
            s.M2[ZC * LDM + VO + row] = s.dvec[row];
          }
          else if (lane >= 32 + NU + NC && lane < 32 + NU + NC + 9)
          {
            const int i = lane - 32 - NU - NC;
            s.M2[(XO + i) * LDM + ZC] = s.q[i];
            s.M2[ZC * LDM + XO + i] = s.q[i];
          }
        }
        SMPC_LANES_END_WAVE
        CENT_FINE_TICK(8);
        // ---- M2 += [A B]^T P~ [A B] (and the vector column += [A B]^T p~) on the matrix cores ----
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
          for (int sk = 0; sk < 3; sk++)
          {
            const int k = 4 * sk + lr;
            SMPC_PLV(pop)[sk] = lc < 9 && k < 9 ? s.Pt[lc * 9 + k] : 0.0;
#pragma unroll
            for (int I = 0; I < 2; I++)
              SMPC_PLV(aop)[sk * 2 + I] = k < 9 ? s.ABp[k * LDM + 16 * I + lc] : 0.0;
          }
#pragma unroll
          for (int J = 0; J < 2; J++)
#pragma unroll
            for (int vv = 0; vv < 4; vv++)
              SMPC_ACCV(tacc, J, vv) = 0.0;
#pragma unroll
          for (int I = 0; I < 2; I++)
#pragma unroll
            for (int J = I; J < 2; J++)
#pragma unroll
              for (int vv = 0; vv < 4; vv++)
              {
                const int r = 16 * I + lr + 4 * vv;
                SMPC_ACCV(macc, tix<2>(I, J), vv) = r < D::R2 ? s.M2[r * LDM + 16 * J + lc] : 0.0;
              }
        }
        SMPC_LANES_END_WAVE
#pragma unroll
        for (int sk = 0; sk < 3; sk++)
#pragma unroll
          for (int J = 0; J < 2; J++)
            SMPC_MFMA(tacc, J, pop, sk, aop, sk * 2 + J);
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
          for (int sk = 0; sk < 3; sk++)
          {
            SMPC_PLV(top)[sk * 2 + 0] = SMPC_ACCV(tacc, 0, sk);
            // column ZC of the right factor is p~ (so that the vector column receives [A B]^T p~)
            const int row = lr + 4 * sk;
            SMPC_PLV(top)[sk * 2 + 1] = lc == ZC - 16 ? (row < 9 ? s.pt[row] : 0.0) : SMPC_ACCV(tacc, 1, sk);
          }
        }
        SMPC_LANES_END_WAVE
#pragma unroll
        for (int sk = 0; sk < 3; sk++)
#pragma unroll
          for (int I = 0; I < 2; I++)
#pragma unroll
            for (int J = I; J < 2; J++)
              SMPC_MFMA(macc, tix<2>(I, J), aop, sk * 2 + I, top, sk * 2 + J);
        CENT_FINE_TICK(9);
        // ---- sweep 2: pivots = [u | nu] ----
        // no active cone row: D = 0, the multiplier pivots -mu are decoupled (Z = 0, z = d / mu) and their panels are skipped
        bool anyact = false;
#pragma unroll
        for (int row = 0; row < NC; row++)
          anyact = anyact || s.cact[row] != 0.0;
        if (anyact)
          wave_block_sweep<NT, 2, true, 0, D::NP2>(macc, prow, urow, CENT_FINE_DBG, tprev);
        else
          wave_block_sweep<NT, 2, true, 0, NUP / 4>(macc, prow, urow, CENT_FINE_DBG, tprev);
        // ---- gains, P_t, p_t straight out of the accumulators ; record for the forward pass ----
        SMPC_LANES(NT)
        {
          const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
          for (int I = 0; I < 2; I++)
#pragma unroll
            for (int J = I; J < 2; J++)
#pragma unroll
              for (int vv = 0; vv < 4; vv++)
              {
                const int r = 16 * I + lr + 4 * vv, c = 16 * J + lc - XO; // c: 0..8 state columns, 9 vector column
                const double a = SMPC_ACCV(macc, tix<2>(I, J), vv);
                if (c >= 0 && c < 10)
                {
                  if (r < NU)
                    g[D::G_K + r * 10 + c] = -a;
                  else if (r >= VO && r < VO + NC)
                    g[D::G_Z + (r - VO) * 10 + c] = anyact ? -a : a * imu;
                  else if (r >= XO && r < XO + 9)
                  {
                    const int i = r - XO;
                    if (c == 9)
                      s.p[i] = a;
                    else if (c >= i)
                    {
                      s.P[i * 9 + c] = a;
                      s.P[c * 9 + i] = a;
                    }
                  }
                }
              }
          if (lane < 9)
          {
            g[D::G_f + lane] = s.f[lane];
            g[D::G_gx + lane] = s.gxp[lane];
            g[D::G_lpd + lane] = s.lpd[lane];
          }
#pragma unroll
          for (int n = 0; n < 2; n++)
            if (lane + n * NT < 81)
              g[D::G_Pt + lane + n * NT] = s.Pt[lane + n * NT];
          if (lane < 3)
            g[D::G_fs + lane] = s.fs[lane];
          if (lane < 3 * NF)
            g[D::G_r + lane] = s.rf[lane];
          if (lane < NF)
            g[D::G_act + lane] = s.act[lane];
          if (lane < NU)
            g[D::G_gu + lane] = s.gu[lane];
          if (lane < NC)
            g[D::G_d + lane] = s.dvec[lane];
        }
        SMPC_LANES_END_WAVE
      }

      prof_tick(dbg, 10, tprev);
      // =====================================================================================
      // forward pass: dx_0 = 0
      // =====================================================================================
      SMPC_LANES(NT)
      {
        SMPC_PLV(acc_dphi) = 0.0;
        const double * g0 = b.gains + (inst * H) * D::G_STRIDE;
#pragma unroll
        for (int n = 0; n < D::G_REGS; n++)
          SMPC_PLV(prec)[n] = g0[lane + n * NT];
        if (lane < 9)
        {
          s.dx[lane] = 0.0;
          b.dxs[(inst * (H + 1)) * 9 + lane] = 0.0;
        }
      }
      SMPC_LANES_END_WAVE
      for (int t = 0; t < H; t++)
      {
        SMPC_LANES(NT)
        {
#pragma unroll
          for (int n = 0; n < D::G_REGS; n++)
            rec[lane + n * NT] = SMPC_PLV(prec)[n];
          if (t + 1 < H)
          {
            const double * gn = b.gains + (inst * H + t + 1) * D::G_STRIDE;
#pragma unroll
            for (int n = 0; n < D::G_REGS; n++)
              SMPC_PLV(prec)[n] = gn[lane + n * NT];
          }
        }
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        {
          if (lane < NU)
          {
            const double * K = rec + D::G_K + lane * 10;
            double a = K[9];
            for (int j = 0; j < 9; j++)
              a += K[j] * s.dx[j];
            s.du[lane] = a;
            b.dus[(inst * H + t) * NU + lane] = a;
            SMPC_PLV(acc_dphi) += rec[D::G_gu + lane] * a;
          }
          if (lane >= 16 && lane < 16 + NC)
          {
            const int row = lane - 16;
            const double * Z = rec + D::G_Z + row * 10;
            double a = Z[9];
            for (int j = 0; j < 9; j++)
              a += Z[j] * s.dx[j];
            b.dvs[(inst * H + t) * NC + row] = a;
            SMPC_PLV(acc_dphi) -= rec[D::G_d + row] * a;
          }
          if (lane >= 32 && lane < 41)
            SMPC_PLV(acc_dphi) += rec[D::G_gx + lane - 32] * s.dx[lane - 32];
        }
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        if (lane < 9)
        {
          // y = A dx + B du + f - mu p_{t+1}
          double a = s.dx[lane];
          const int k = lane % 3;
          if (lane < 3)
            a += dt * s.dx[3 + lane] * imass;
          else if (lane < 6)
          {
            double sf = 0.0;
            for (int f = 0; f < NF; f++)
              sf += rec[D::G_act + f] * s.du[3 * f + k];
            a += dt * sf;
          }
          else
          {
            V3 tq = cross(ld3(rec + D::G_fs), ld3(s.dx)); // [fs]x dc
            for (int f = 0; f < NF; f++)
              tq = tq + cross(ld3(rec + D::G_r + 3 * f), ld3(&s.du[3 * f]));
            a += dt * (k == 0 ? tq.x : (k == 1 ? tq.y : tq.z));
          }
          s.y[lane] = a + rec[D::G_f + lane] - mu * rec[D::G_pn + lane];
        }
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        if (lane < 9)
        {
          double w = 0.0;
          for (int j = 0; j < 9; j++)
            w += rec[D::G_Pt + lane * 9 + j] * s.y[j];
          const double dxn = s.y[lane] - mu * w;
          const double dl = w + rec[D::G_pn + lane];
          s.w[lane] = dxn;
          b.dxs[(inst * (H + 1) + t + 1) * 9 + lane] = dxn;
          b.dlams[(inst * H + t) * 9 + lane] = dl;
          SMPC_PLV(acc_dphi) -= rec[D::G_lpd + lane] * dxn + rec[D::G_f + lane] * dl;
        }
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        if (lane < 9)
          s.dx[lane] = s.w[lane];
        SMPC_LANES_END_WAVE
      }
      prof_tick(dbg, 11, tprev);
      // terminal gradient lx_N . dx_H ; reductions
      SMPC_LANES(NT)
      {
        if (lane >= 3 && lane < 9)
        {
          const double * xH = b.xs + (ib + ring_slot(head, H, R)) * 9;
          const double * W = lane < 6 ? md.w_lm : md.w_am;
          const int bo = lane < 6 ? 3 : 6;
          double gN = 0.0;
          for (int j = 0; j < 3; j++)
            gN += W[(lane - bo) * 3 + j] * xH[bo + j];
          SMPC_PLV(acc_dphi) += gN * s.dx[lane];
        }
      }
      SMPC_LANES_END_WAVE
      // five reductions through LDS (lane order: deterministic)
      for (int which = 0; which < 5; which++)
      {
        SMPC_LANES(NT)
        red[lane] = which == 0 ? SMPC_PLV(acc_cost) : (which == 1 ? SMPC_PLV(acc_pen) : (which == 2 ? SMPC_PLV(acc_prim) : (which == 3 ? SMPC_PLV(acc_dual) : SMPC_PLV(acc_dphi))));
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        if (lane == 0)
        {
          if (which == 0)
            s.sc[SC_COST] = fold64<false>(red);
          else if (which == 1)
            s.sc[SC_PHI0] = s.sc[SC_COST] + fold64<false>(red);
          else if (which == 2)
            s.sc[SC_PRIM] = fold64<true>(red);
          else if (which == 3)
            s.sc[SC_DUAL] = fold64<true>(red);
          else
            s.sc[SC_DPHI0] = fold64<false>(red);
        }
        SMPC_LANES_END_WAVE
      }

      prof_tick(dbg, 12, tprev);
      // =====================================================================================
      // line search (lane = stage; lane / slot H carries the terminal cost)
      // =====================================================================================
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
            const double * xg = b.xs + (ib + ring_slot(head, t, R)) * 9;
            const double * dxg = b.dxs + (inst * (H + 1) + t) * 9;
            if (t == H)
            {
              const V3 hh = mk3(xg[3] + alpha * dxg[3], xg[4] + alpha * dxg[4], xg[5] + alpha * dxg[5]);
              const V3 LL = mk3(xg[6] + alpha * dxg[6], xg[7] + alpha * dxg[7], xg[8] + alpha * dxg[8]);
              cst += 0.5 * dot(hh, ldm3(md.w_lm) * hh) + 0.5 * dot(LL, ldm3(md.w_am) * LL);
              continue;
            }
            const size_t sl = ib + ring_slot(head, t, R);
            CentTrial<D> q;
            q.x = xg;
            q.dx = dxg;
            q.xn = b.xs + (ib + ring_slot(head, t + 1, R)) * 9;
            q.dxn = dxg + 9;
            q.u = b.us + sl * NU;
            q.du = b.dus + (inst * H + t) * NU;
            q.v = b.vs + sl * NC;
            q.dv = b.dvs + (inst * H + t) * NC;
            q.l1 = b.lams + sl * 9;
            q.dl = b.dlams + (inst * H + t) * 9;
            q.ve = b.vs_e + sl * NC;
            q.l1e = b.lams_e + sl * 9;
            q.p = b.foot + (inst * H + t) * (3 * NF);
            q.uref = b.stages[t].u_ref;
            q.xtgt = b.stages[t].x_tgt;
            q.href = b.vref + sl * 6;
            q.alpha = alpha;
            q.mask = b.stages[t].mask;
            double c1, p1, r1;
            cent_stage_merit<D>(md, q, c1, p1, r1, nullptr, &s.M2[lane * NU]);
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
      prof_tick(dbg, 13, tprev);
      // ---- accept (the last candidate is taken when none passes, like the restated solver) ----
      SMPC_LANES(NT)
      {
        for (int t = lane; t <= H; t += NT)
        {
          double * xg = b.xs + (ib + ring_slot(head, t, R)) * 9;
          const double * dxg = b.dxs + (inst * (H + 1) + t) * 9;
          for (int i = 0; i < 9; i++)
            xg[i] += alpha * dxg[i];
          if (t < H)
          {
            const size_t sl = ib + ring_slot(head, t, R);
            for (int i = 0; i < NU; i++)
              b.us[sl * NU + i] += alpha * b.dus[(inst * H + t) * NU + i];
            for (int i = 0; i < NC; i++)
              b.vs[sl * NC + i] += alpha * b.dvs[(inst * H + t) * NC + i];
            for (int i = 0; i < 9; i++)
              b.lams[sl * 9 + i] += alpha * b.dlams[(inst * H + t) * 9 + i];
          }
        }
        if (lane == 0)
        {
          s.sc[SC_ALPHA] = alpha;
          s.sc[SC_LS_FAILED] = accepted < 0 ? 1.0 : 0.0;
          s.sc[SC_LS_INDEX] = (double)jlast;
          s.sc[SC_PREG] = accepted < 0 ? fmin(preg * ka.reg_inc, ka.reg_max) : fmax(preg * ka.reg_dec, ka.reg_min);
        }
      }
      SMPC_LANES_END_WAVE
    }

    prof_tick(dbg, 14, tprev);
    // ---- outputs: solver scalars, xdot at t = 0, 1 of the accepted iterate (MPC::getStateDerivative) ----
    SMPC_LANES(NT)
    {
      if (lane < 16)
        gsc[lane] = s.sc[lane];
      if (lane >= 32 && lane < 34 && ka.iters > 0)
      {
        const int t = lane - 32;
        const size_t sl = ib + ring_slot(head, t, R);
        CentTrial<D> q;
        q.x = q.dx = b.xs + sl * 9;
        q.xn = q.dxn = b.xs + (ib + ring_slot(head, t + 1, R)) * 9;
        q.u = q.du = b.us + sl * NU;
        q.v = q.dv = b.vs + sl * NC;
        q.l1 = q.dl = b.lams + sl * 9;
        q.ve = b.vs_e + sl * NC;
        q.l1e = b.lams_e + sl * 9;
        q.p = b.foot + (inst * H + t) * (3 * NF);
        q.uref = b.stages[t].u_ref;
        q.xtgt = b.stages[t].x_tgt;
        q.href = b.vref + sl * 6;
        q.alpha = 0.0; // the accepted iterate itself
        q.mask = b.stages[t].mask;
        double c1, p1, r1, xd[9];
        cent_stage_merit<D>(md, q, c1, p1, r1, xd, &s.M2[lane * NU]);
        for (int i = 0; i < 9; i++)
          b.xdot01[(inst * 2 + t) * 9 + i] = xd[i];
      }
    }
    SMPC_LANES_END_WAVE
  }

  // Targets between MPC knots (reference src/interpolator.cpp:44-78 as used in examples/talos_centroidal.py: interpolateLinear
  // over the centroidal states, their derivatives and the contact forces) and the Riccati feedback on the centroidal state:
  //   x(d) = interpolateLinear(xs[0 .. knots-1]),  xdot(d) = interpolateLinear(getStateDerivative(0), (1)),
  //   f(d) = interpolateLinear(us[0], us[1]),      u = f(d) - K_0 (x(d) - x_meas)        (the state space is a vector space)
  template <class D>
  struct CentInterpArgs
  {
    CentBuffers<D> b;
    int head, knots;
    double delay, timestep;
    const double * x_meas;                // [B][9] centroidal state of the measured multibody state, or null
    double *x_out, *xdot_out, *f_out;     // [B][9], [B][9], [B][NU] (device), any may be null
    double * u_out;                       // [B][NU] Riccati feedback (needs x_meas), may be null
    // targets of the CentroidalID controller (examples/talos_centroidal.py:218-243), any may be null: centre of mass and its velocity
    // (linear momentum / mass) [B][3], foot references between stages 0 and 1 and their velocities [B][3 NF]
    double *com_out = nullptr, *vcom_out = nullptr, *fp_out = nullptr, *fv_out = nullptr;
    double mass = 1.0;
  };
  template <class D>
  SMPC_DEV void cent_interp_body(const CentInterpArgs<D> & ka, int block)
  {
    constexpr int NT = 64, NU = D::NU;
    const CentBuffers<D> & b = ka.b;
    const size_t inst = (size_t)block;
    const int R = b.R, H = b.H;
    const size_t step = (size_t)(ka.delay / ka.timestep);
    const double sx = (ka.delay - (double)step * ka.timestep) / ka.timestep;
    const bool lastx = step >= (size_t)ka.knots - 1; // beyond the last interval: the last knot (reference :50-53)
    const bool last2 = step >= 1;                    // two knots for xdot and the forces
    SMPC_LDS(double, e, 9);
    SMPC_LANES(NT)
    {
      if (lane < 9)
      {
        const int t0 = lastx ? ka.knots - 1 : (int)step, t1 = lastx ? ka.knots - 1 : (int)step + 1;
        const double x0 = b.xs[(inst * R + ring_slot(ka.head, t0, R)) * 9 + lane], x1 = b.xs[(inst * R + ring_slot(ka.head, t1, R)) * 9 + lane];
        const double xi = lastx ? x0 : x1 * sx + x0 * (1.0 - sx);
        if (ka.x_out)
          ka.x_out[inst * 9 + lane] = xi;
        if (ka.xdot_out)
        {
          const double d0 = b.xdot01[(inst * 2) * 9 + lane], d1 = b.xdot01[(inst * 2 + 1) * 9 + lane];
          ka.xdot_out[inst * 9 + lane] = last2 ? d1 : d1 * sx + d0 * (1.0 - sx);
        }
        e[lane] = ka.x_meas ? xi - ka.x_meas[inst * 9 + lane] : 0.0;
        if (ka.com_out && lane < 3)
          ka.com_out[inst * 3 + lane] = xi;
        if (ka.vcom_out && lane >= 3 && lane < 6)
          ka.vcom_out[inst * 3 + lane - 3] = xi / ka.mass;
      }
      if (ka.fp_out && lane >= 16 && lane < 16 + 3 * D::NF)
      {
        const int i = lane - 16;
        const double p0 = b.foot[(inst * H) * (3 * D::NF) + i], p1 = b.foot[(inst * H + 1) * (3 * D::NF) + i];
        ka.fp_out[inst * 3 * D::NF + i] = last2 ? p1 : (1.0 - sx) * p0 + sx * p1;
        ka.fv_out[inst * 3 * D::NF + i] = (p1 - p0) / ka.timestep;
      }
    }
    SMPC_LANES_END_WAVE
    static_assert(16 + 3 * D::NF <= 64, "lane map of the foot targets");
    SMPC_LANES(NT)
    if (lane < NU)
    {
      const double u0 = b.us[(inst * R + ring_slot(ka.head, 0, R)) * NU + lane], u1 = b.us[(inst * R + ring_slot(ka.head, 1, R)) * NU + lane];
      const double ui = last2 ? u1 : u1 * sx + u0 * (1.0 - sx);
      if (ka.f_out)
        ka.f_out[inst * NU + lane] = ui;
      if (ka.u_out)
      {
        const double * K = b.gains + (inst * H) * D::G_STRIDE + D::G_K + lane * D::GKS;
        double a = ui;
        for (int j = 0; j < 9; j++)
          a -= K[j] * e[j];
        ka.u_out[inst * NU + lane] = a;
      }
    }
    SMPC_LANES_END_WAVE
  }

  // K_t of every stage -> dense [B][H][NU][9] (MPC::Ks_) or K_0 only
  template <class D>
  struct CentGainsOutArgs
  {
    CentBuffers<D> b;
    double * out;
    int all;
  };
  template <class D>
  SMPC_DEV void cent_gains_out_body(const CentGainsOutArgs<D> & ka, int block)
  {
    constexpr int NT = 64, NU = D::NU;
    const int H = ka.b.H;
    const int inst = ka.all ? block / H : block, t = ka.all ? block % H : 0;
    const double * g = ka.b.gains + ((size_t)inst * H + t) * D::G_STRIDE + D::G_K;
    double * o = ka.out + (size_t)block * NU * 9;
    SMPC_LANES(NT)
    for (int idx = lane; idx < NU * 9; idx += NT)
      o[idx] = g[(idx / 9) * D::GKS + idx % 9];
    SMPC_LANES_END_WAVE
  }
} // namespace smpc
