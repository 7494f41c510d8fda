// smpc_cent6_kernels.h -- the centroidal OCP of a robot with 6-D (flat) feet (reference CentroidalOCP with force_size == 6:
// src/centroidal-dynamics.cpp:39-106 -- the Talos configuration of examples/talos_centroidal.py:39-96 and tests/test_utils.cpp:199-218):
//   control u = [(f_i, tau_i) per foot]; the contact torques add to the angular-momentum rate and to the angular_acc residual;
//   constraint per foot in contact: CentroidalWrenchConeResidual, 17 constant linear rows A_cone(mu, L, W) u_i <= 0 (:90-95).
// Unlike the point-foot problem (cent_step_body: the whole control step fused in one wavefront, smpc_cent_kernels.h) a stage of this one
// carries up to 34 explicit multiplier pivots; it runs as (instance x stage) kernels around the dense proximal Riccati sweep on the FP64
// matrix cores (riccati_dense_body, smpc_riccati_dense.h) -- the sweep the full-dynamics and the 6-D kinodynamics OCPs use:
//   cent6_recede_body   B blocks          ring advance, warm-start shift, Raibert + Bezier -> contact positions, AL centres
//   cent6_deriv_body    B (H + 1) blocks  stage evaluation + derivatives + LQ knot in the dense layout (state padded 9 -> 12: the sweep's
//                                         pivot panels are 4 wide; the pad rows are zero and decoupled)
//   riccati_dense_body  B blocks          backward sweep, wrench-cone multipliers pivoted explicitly
//   cent6_forward_body  B blocks          (dx, du, dnu, dlam) and the directional derivative of the merit
//   cent6_trial_body    B (H + 1) blocks  stage merit at line-search candidates (the full step; then alpha = 1/2, 1/4, .. where it failed)
//   cent6_ls_body       B blocks          Armijo test over the candidates' merits, step, regularisation update
// Algorithm and constants: those of cent_step_body (smpc_cent_kernels.h), i.e. of the solver stack in DESIGN.md section 2.
#pragma once
#include "smpc_cent_kernels.h"
#include "smpc_full_stage.h" // wrench_cone_entry, full_dynamics_phases (front end of a robot with any tree)
#include "smpc_riccati_dense.h"

namespace smpc
{
  // dimensions the dense sweep sees
  template <int NF_>
  struct Cent6Dims
  {
    static constexpr int NF = NF_, FS = 6, NJ = 1;
    static constexpr int NX = 9, NXR = 9;
    static constexpr int NDX = 12;                 // padded tangent dimension (9 real)
    static constexpr int NU = 6 * NF_, NA = 0, NV = 9, NQ = 9;
    static constexpr int NCONE1 = 17, NCONE = 17 * NF_, NLAND = 0, NCD = NCONE, NVEL = 0;
    static constexpr int NC = NU + NA + NCD;       // knot rows: [control box (absent) | wrench-cone rows]
    static constexpr int NXU = NDX + NU;
    static constexpr int O_A = 0;
    static constexpr int O_B = O_A + NDX * NDX;
    static constexpr int O_Q = O_B + NDX * NU;
    static constexpr int O_S = O_Q + NDX * NDX;
    static constexpr int O_R = O_S + NDX * NU;
    static constexpr int O_C = O_R + NU * NU;
    static constexpr int O_D = O_C + NCD * NDX;
    static constexpr int O_V = O_D + NCD * NU;
    static constexpr int O_q = O_V;
    static constexpr int O_r = O_q + NDX;
    static constexpr int O_f = O_r + NU;
    static constexpr int O_d = O_f + NDX;
    static constexpr int O_lx = O_d + NC;
    static constexpr int O_lu = O_lx + NDX;
    static constexpr int O_lpd = O_lu + NU;
    static constexpr int O_vpd = O_lpd + NDX;
    static constexpr int O_act = O_vpd + NC;
    static constexpr int O_cdirty = O_act + NC;    // 1.0 while the dense rows [C | D] of this block hold a nonzero entry (cent6_deriv_body; allocations are zero-filled)
    static constexpr int LQ_STRIDE = ((O_cdirty + 1 + 7) / 8) * 8;
    static constexpr int G_K = 0;
    static constexpr int G_Z = G_K + NU * (NDX + 1);
    static constexpr bool PT_PACKED = false; // (P~ as a full NDX x NDX image: cent6_forward_body reads rows of it)
    static constexpr int G_Pt = G_Z + NCD * (NDX + 1);
    static constexpr int G_pn = G_Pt + NDX * NDX;
    static constexpr int G_STRIDE = ((G_pn + NDX + 7) / 8) * 8;
    static constexpr int LS_N = 10;
  };
  template <int NF_>
  struct DevModel<Cent6Dims<NF_>>
  {
    double mu;
  };
  template <int NF_>
  struct StageShared<Cent6Dims<NF_>>
  {
    unsigned mask, land;
  };

  // CentDims of a robot with 6-D feet: what the host engine and the interpolation / gain read-out kernels need
  template <int NF_>
  struct CentDims<NF_, 6>
  {
    typedef Cent6Dims<NF_> DD;
    static constexpr int NF = NF_, FS = 6;
    static constexpr int NX = 9, NDX = 9;
    static constexpr int NU = 6 * NF_;
    static constexpr int NC = 17 * NF_; // rows 17 f ..: wrench-cone block of foot f
    static constexpr int G_K = DD::G_K, GKS = DD::NDX + 1, G_STRIDE = DD::G_STRIDE; // [K | k] rows of the dense gains block
    static constexpr int LS_N = 10;
  };

  template <class D>
  struct Cent6Args
  {
    CentBuffers<D> b;
    Buffers<typename D::DD> sb; // what the dense sweep reads: lq, gains, QN, qN, model (mu), dbg
    double *parts0 = nullptr;   // [B][H+1][4] phi, cost, prim, dual at the current point
    double *partsT = nullptr;   // [B][LS_N][H+1][2] phi, prim of every line-search candidate (cent6_trial_body)
    double *xdotT = nullptr;    // [B][LS_N][2][9] state derivative of stages 0, 1 at every candidate
    int head;
    int j0 = 0, nj = 0;         // line search: candidate range of the launch
    int shift, set_centres, reset_preg;
    const double * X;
    int nx_mb;
    const double *cstate, *feet;
    int land[D::NF];
    int T_fly, T_contact;
    double swing_apex, timestep;
    double armijo_c1, reg_init, reg_min, reg_max, reg_inc, reg_dec;
  };

  // ---------------------------------------------------------------------------------------------------------------
  // recede: warm-start shift on the ring (src/mpc.cpp:201-207), references (src/mpc.cpp:278-309), AL centres := multipliers,
  // regularisation restart.  One wavefront per instance (the recede part of cent_step_body, generic in NU / NC).
  // ---------------------------------------------------------------------------------------------------------------
  template <class D>
  SMPC_DEV void cent6_recede_body(const Cent6Args<D> & ka, int block)
  {
    constexpr int NT = 64, NU = D::NU, NC = D::NC, NF = D::NF;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head;
    const size_t inst = (size_t)block, ib = inst * R;
    SMPC_LDS(double, rec, D::NF * 6);
    SMPC_LDS(CentDevModel<D>, mds, 1);
    SMPC_LANES(NT)
    {
      constexpr int N = (int)(sizeof(CentDevModel<D>) / sizeof(double));
      const alias_double * src = reinterpret_cast<const alias_double *>(b.model);
      alias_double * dst = reinterpret_cast<alias_double *>(&mds[0]);
      for (int i = lane; i < N; i += NT)
        dst[i] = src[i];
    }
    SMPC_LANES_END_WAVE
    const CentDevModel<D> & md = mds[0];
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
        for (int i = lane; i < NU; i += NT)
          b.us[(ib + sHm1) * NU + i] = b.us[(ib + sHm2) * NU + i];
        for (int i = lane; i < NC; i += NT)
          b.vs[(ib + sHm1) * NC + i] = 0.0;
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
    SMPC_LANES(NT)
    {
      if (ka.set_centres)
        for (int t = lane; t < H; t += NT)
        {
          const size_t sl = ib + ring_slot(head, t, R);
          for (int i = 0; i < NC; i++)
            b.vs_e[sl * NC + i] = b.vs[sl * NC + i];
          for (int i = 0; i < 9; i++)
            b.lams_e[sl * 9 + i] = b.lams[sl * 9 + i];
        }
      if (ka.reset_preg && lane == 0)
        b.scal[inst * SC_N + SC_PREG] = ka.reg_init;
    }
    SMPC_LANES_END_WAVE
  }

  // stage quantities at a point (x, u): contact sums, state derivative; shared by the derivative pass and the line search
  template <class D>
  struct Cent6Point
  {
    V3 fs, ts; // sum of the contact forces ; sum of (p - c) x f + tau
  };
  template <class D, class FX, class FU>
  SMPC_DEV Cent6Point<D> cent6_point(unsigned mask, const double * pp, FX x, FU u)
  {
    Cent6Point<D> q;
    q.fs = mk3(0, 0, 0);
    q.ts = mk3(0, 0, 0);
    const V3 c = mk3(x(0), x(1), x(2));
#pragma unroll
    for (int f = 0; f < D::NF; f++)
      if ((mask >> f) & 1u)
      {
        const V3 F = mk3(u(6 * f), u(6 * f + 1), u(6 * f + 2));
        q.fs = q.fs + F;
        q.ts = q.ts + cross(ld3(pp + 3 * f) - c, F) + mk3(u(6 * f + 3), u(6 * f + 4), u(6 * f + 5));
      }
    return q;
  }

  // ---------------------------------------------------------------------------------------------------------------
  // derivative pass: one wavefront per (instance, stage); t == H is the terminal node (linear + angular momentum cost,
  // src/centroidal-dynamics.cpp:306-316).  Lane = entry; every entry is a closed-form expression of the stage inputs.
  // ---------------------------------------------------------------------------------------------------------------
  template <class D>
  struct Cent6DerivLds
  {
    static constexpr int NU = D::NU, NC = D::NC, NF = D::NF, NXU = 9 + D::NU;
    CentDevModel<D> md;
    double x[9], xn[9], l1[9], l1e[9], l0[9], u[NU], v[NC], ve[NC], pp[3 * NF], uref[NU], xt[9];
    double f[9], lamp[9], lpd[9], dvec[NC], vpd[NC], vplus[NC], act[NC];
    double res[6 + 9 + NU], wres[6 + 9 + NU]; // residuals [la (3) | aa (3) | x blocks (9) | u (NU)] and W r
    double J[6 * NXU];  // Jacobian of [linear_acc ; angular_acc] w.r.t. (x | u)
    double WJ[6 * NXU];
    double A[81], B[9 * NU];
    double g[NXU];      // cost gradient (lx | lu)
    double yc[NU];      // D^T nu: A_cone^T nu on the wrench of every foot in contact
    double red[64], red2[64], red3[64];
    double cost;
  };
  template <class D>
  SMPC_DEV void cent6_deriv_body(const Cent6Args<D> & ka, int block)
  {
    typedef typename D::DD DD;
    constexpr int NT = 64, NU = D::NU, NC = D::NC, NF = D::NF, NXU = 9 + D::NU, NP = DD::NDX;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head;
    const size_t inst = (size_t)(block / (H + 1));
    const int t = block % (H + 1);
    const bool term = t == H;
    const size_t ib = inst * R;
    SMPC_LDS(Cent6DerivLds<D>, ldsv, 1);
    Cent6DerivLds<D> & s = ldsv[0];
    const int st = ring_slot(head, t, R), st1 = ring_slot(head, term ? t : t + 1, R), stm = ring_slot(head, t > 0 ? t - 1 : 0, R);
    const unsigned mask = term ? 0u : b.stages[t].mask;
    const double preg = b.scal[inst * SC_N + SC_PREG];
    // (flag of the knot's dense rows: a scalar load here, used where the knot is written)
    const bool dense_dirty = term ? false : ka.sb.lq[(inst * H + t) * DD::LQ_STRIDE + DD::O_cdirty] != 0.0;
    double * parts = ka.parts0 + (inst * (H + 1) + t) * 4;
    // every global load of the block's inputs before the first commit to LDS (indices clamped, the terminal node reads stage 0's shared
    // entries and does not use them): written as one loop per array the phase was a chain of memory round trips per block
    SMPC_LANES(NT)
    {
      constexpr int N = (int)(sizeof(CentDevModel<D>) / sizeof(double)), PM = (N + NT - 1) / NT;
      static_assert(NU <= NT && NC <= NT && 3 * NF <= NT, "one entry per lane");
      const alias_double * src = reinterpret_cast<const alias_double *>(b.model);
      alias_double * dst = reinterpret_cast<alias_double *>(&s.md);
      const int ts = term ? 0 : t, l9 = lane < 9 ? lane : 8;
      double vm[PM];
#pragma unroll
      for (int n = 0; n < PM; n++)
        vm[n] = src[lane + n * NT < N ? lane + n * NT : N - 1];
      const double vx = b.xs[(ib + st) * 9 + l9], vxn = b.xs[(ib + st1) * 9 + l9];
      const double vl1 = b.lams[(ib + st) * 9 + l9], vl1e = b.lams_e[(ib + st) * 9 + l9], vl0 = b.lams[(ib + stm) * 9 + l9];
      const double vxt = *(l9 < 3 ? &b.stages[ts].x_tgt[l9] : &b.vref[(ib + st) * 6 + l9 - 3]);
      const int lu = lane < NU ? lane : NU - 1, lc = lane < NC ? lane : NC - 1, lf = lane < 3 * NF ? lane : 3 * NF - 1;
      const double vu = b.us[(ib + st) * NU + lu], vur = b.stages[ts].u_ref[lu];
      const double vv = b.vs[(ib + st) * NC + lc], vve = b.vs_e[(ib + st) * NC + lc];
      const double vpp = b.foot[(inst * H + ts) * (3 * NF) + lf];
      SMPC_SCHED_FENCE();
#pragma unroll
      for (int n = 0; n < PM; n++)
        if (lane + n * NT < N)
          dst[lane + n * NT] = vm[n];
      if (lane < 9)
      {
        s.x[lane] = vx;
        s.xn[lane] = vxn;
        s.l1[lane] = term ? 0.0 : vl1;
        s.l1e[lane] = term ? 0.0 : vl1e;
        s.l0[lane] = t > 0 ? vl0 : 0.0;
        s.xt[lane] = term ? 0.0 : vxt;
      }
      if (!term)
      {
        if (lane < NU)
        {
          s.u[lane] = vu;
          s.uref[lane] = vur;
        }
        if (lane < NC)
        {
          s.v[lane] = vv;
          s.ve[lane] = vve;
        }
        if (lane < 3 * NF)
          s.pp[lane] = vpp;
      }
    }
    SMPC_LANES_END_WAVE
    const CentDevModel<D> & md = s.md;
    const double mu = md.mu, imu = 1.0 / mu, dt = md.dt, mass = md.mass, imass = 1.0 / mass;
    if (term)
    {
      // ---- terminal node: Q_N = blockdiag(0, w_lm, w_am) + preg I, q_N = lx_N - lambda_H (padded to NP) ----
      double * QN = ka.sb.QN + inst * NP * NP;
      double * qN = ka.sb.qN + inst * NP;
      SMPC_LANES(NT)
      {
        for (int idx = lane; idx < NP * NP; idx += NT)
        {
          const int i = idx / NP, j = idx % NP;
          double v = 0.0;
          if (i < 9 && j < 9)
          {
            v = i == j ? preg : 0.0;
            if (i >= 3 && i / 3 == j / 3)
            {
              const double * W = i < 6 ? md.w_lm : md.w_am; // (pointer first: g++ 11 with -fsanitize=shift miscompiles the indexed conditional of two arrays)
              v += W[(i % 3) * 3 + j % 3];
            }
          }
          QN[idx] = v;
        }
        double dual = 0.0;
        if (lane < NP)
        {
          double qn = 0.0;
          if (lane < 9)
          {
            double g = 0.0;
            if (lane >= 3)
            {
              const double * W = lane < 6 ? md.w_lm : md.w_am;
              const int bo = lane < 6 ? 3 : 6;
              for (int j = 0; j < 3; j++)
                g += W[(lane - bo) * 3 + j] * s.x[bo + j];
            }
            qn = g - s.l0[lane];
            dual = fabs(qn);
          }
          qN[lane] = qn;
        }
        s.red[lane] = dual;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane == 0)
      {
        double dual = 0.0;
        for (int i = 0; i < 9; i++)
          dual = fmax(dual, s.red[i]);
        const V3 h = ld3(s.x + 3), L = ld3(s.x + 6);
        const double cost = 0.5 * dot(h, ldm3(md.w_lm) * h) + 0.5 * dot(L, ldm3(md.w_am) * L);
        parts[0] = cost;
        parts[1] = cost;
        parts[2] = 0.0;
        parts[3] = dual;
      }
      SMPC_LANES_END_WAVE
      return;
    }
    // ---- point quantities, defect, multiplier estimates, residuals ----
    SMPC_LANES(NT)
    {
      const Cent6Point<D> q = cent6_point<D>(mask, s.pp, [&](int i) { return s.x[i]; }, [&](int i) { return s.u[i]; });
      const V3 gv = ld3(md.gravity);
      double pen = 0.0, prim = 0.0;
      if (lane < 9)
      {
        const int k = lane % 3;
        const double xd = lane < 3 ? s.x[3 + k] * imass : (lane < 6 ? mass * v3c(gv, k) + v3c(q.fs, k) : v3c(q.ts, k));
        const double e = s.x[lane] + dt * xd - s.xn[lane];
        const double lp = s.l1e[lane] + e * imu, dl = lp - s.l1[lane];
        s.lamp[lane] = lp;
        s.f[lane] = mu * dl;
        s.lpd[lane] = 2.0 * lp - s.l1[lane];
        pen += 0.5 * mu * (lp * lp + dl * dl);
        prim = fmax(prim, fabs(e));
        s.res[6 + lane] = s.x[lane] - s.xt[lane];
        if (lane < 3)
        {
          s.res[lane] = v3c(gv, k) + v3c(q.fs, k) * imass; // linear_acc: CoM acceleration g + sum f / m
          s.res[3 + lane] = v3c(q.ts, k);                  // angular_acc
        }
      }
      for (int i = lane; i < NU; i += NT)
        s.res[15 + i] = s.u[i] - s.uref[i];
      // wrench-cone rows of the feet in contact (negative orthant)
      for (int row = lane; row < NC; row += NT)
      {
        const int f = row / 17, r = row % 17;
        double vp = 0.0, act = 0.0;
        if ((mask >> f) & 1u)
        {
          double cv = 0.0;
          for (int j = 0; j < 6; j++)
            cv += wrench_cone_entry(r, j, md.mu_fric, md.Lfoot, md.Wfoot) * s.u[6 * f + j];
          const double z = cv + mu * s.ve[row];
          const double proj = fmin(z, 0.0);
          vp = (z - proj) * imu;
          act = z != proj ? 1.0 : 0.0;
          prim = fmax(prim, fmax(cv, 0.0));
        }
        const double dv = vp - s.v[row];
        s.vplus[row] = vp;
        s.dvec[row] = mu * dv;
        s.vpd[row] = act != 0.0 ? 2.0 * vp - s.v[row] : 0.0;
        s.act[row] = act;
        pen += 0.5 * mu * (vp * vp + dv * dv);
      }
      s.red[lane] = pen;
      s.red2[lane] = prim;
    }
    SMPC_LANES_END_WAVE
    // ---- weighted residuals ; Jacobian of [linear_acc ; angular_acc] ; [A B] ; D^T nu ----
    SMPC_LANES(NT)
    {
      if (lane < 15)
      { // 3 x 3 weights: la, aa, com, lm, am
        const int blk = lane / 3, k = lane % 3;
        const double * W = blk == 0 ? md.w_la : (blk == 1 ? md.w_aa : (blk == 2 ? md.w_com : (blk == 3 ? md.w_lm : md.w_am)));
        const double * r = s.res + 3 * blk;
        s.wres[lane] = W[k * 3] * r[0] + W[k * 3 + 1] * r[1] + W[k * 3 + 2] * r[2];
      }
      for (int i = lane; i < NU; i += NT)
      {
        double a = 0.0;
        for (int j = 0; j < NU; j++)
          a += md.w_u[i * NU + j] * s.res[15 + j];
        s.wres[15 + i] = a;
      }
      const Cent6Point<D> q = cent6_point<D>(mask, s.pp, [&](int i) { return s.x[i]; }, [&](int i) { return s.u[i]; });
      const V3 c = ld3(s.x);
      for (int idx = lane; idx < 6 * NXU; idx += NT)
      {
        const int r = idx / NXU, k = idx % NXU;
        double v = 0.0;
        if (k >= 9)
        {
          const int f = (k - 9) / 6, j = (k - 9) % 6;
          if ((mask >> f) & 1u)
          {
            if (r < 3)
              v = (j < 3 && j == r) ? imass : 0.0;
            else if (j < 3) // d((p - c) x f)/df_j = (p - c) x e_j
              v = v3c(cross(ld3(s.pp + 3 * f) - c, mk3(j == 0, j == 1, j == 2)), r - 3);
            else
              v = (j - 3 == r - 3) ? 1.0 : 0.0;
          }
        }
        else if (k < 3 && r >= 3) // d(sum (p - c) x f)/dc_k = fs x e_k
          v = v3c(cross(q.fs, mk3(k == 0, k == 1, k == 2)), r - 3);
        s.J[idx] = v;
      }
      for (int idx = lane; idx < 81; idx += NT)
      {
        const int i = idx / 9, j = idx % 9;
        double v = i == j ? 1.0 : 0.0;
        if (i < 3 && j == 3 + i)
          v += dt * imass;
        if (i >= 6 && j < 3)
          v += dt * v3c(cross(q.fs, mk3(j == 0, j == 1, j == 2)), i - 6);
        s.A[idx] = v;
      }
      for (int idx = lane; idx < 9 * NU; idx += NT)
      {
        const int i = idx / NU, k = idx % NU, f = k / 6, j = k % 6;
        double v = 0.0;
        if ((mask >> f) & 1u)
        {
          if (i >= 3 && i < 6)
            v = (j < 3 && j == i - 3) ? dt : 0.0;
          else if (i >= 6)
            v = j < 3 ? dt * v3c(cross(ld3(s.pp + 3 * f) - c, mk3(j == 0, j == 1, j == 2)), i - 6) : ((j - 3 == i - 6) ? dt : 0.0);
        }
        s.B[idx] = v;
      }
      for (int k = lane; k < NU; k += NT)
      {
        const int f = k / 6, j = k % 6;
        double a = 0.0;
        if ((mask >> f) & 1u)
          for (int r = 0; r < 17; r++)
            a += wrench_cone_entry(r, j, md.mu_fric, md.Lfoot, md.Wfoot) * s.v[17 * f + r];
        s.yc[k] = a;
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      for (int idx = lane; idx < 6 * NXU; idx += NT)
      {
        const int r = idx / NXU, k = idx % NXU;
        const double * W = r < 3 ? md.w_la : md.w_aa;
        const int r0 = r < 3 ? 0 : 3, rr = r - r0;
        s.WJ[idx] = W[rr * 3] * s.J[r0 * NXU + k] + W[rr * 3 + 1] * s.J[(r0 + 1) * NXU + k] + W[rr * 3 + 2] * s.J[(r0 + 2) * NXU + k];
      }
      // cost gradient: direct terms (x blocks, control) + J^T W r of the two acceleration residuals
      for (int k = lane; k < NXU; k += NT)
      {
        double g = k < 9 ? s.wres[6 + k] : s.wres[15 + k - 9];
        for (int r = 0; r < 6; r++)
          g += s.J[r * NXU + k] * s.wres[r];
        s.g[k] = g;
      }
      if (lane == 0)
      {
        double c = 0.0;
        for (int i = 0; i < 15 + NU; i++)
          c += s.res[i] * s.wres[i];
        s.cost = 0.5 * c;
      }
    }
    SMPC_LANES_END_WAVE
    // ---- knot ----
    double * lq = ka.sb.lq + (inst * H + t) * DD::LQ_STRIDE;
    SMPC_LANES(NT)
    {
      double dual = 0.0;
      // A, B (padded rows / columns zero)
      for (int idx = lane; idx < NP * NP; idx += NT)
      {
        const int i = idx / NP, j = idx % NP;
        lq[DD::O_A + idx] = (i < 9 && j < 9) ? s.A[i * 9 + j] : 0.0;
        // Q = Lxx + preg I:  direct blocks (com, lm, am weights) + J_x^T W J_x
        double q = 0.0;
        if (i < 9 && j < 9)
        {
          q = i == j ? preg : 0.0;
          if (i / 3 == j / 3)
          {
            const double * W = i < 3 ? md.w_com : (i < 6 ? md.w_lm : md.w_am);
            q += W[(i % 3) * 3 + j % 3];
          }
          for (int r = 0; r < 6; r++)
            q += s.J[r * NXU + i] * s.WJ[r * NXU + j];
        }
        lq[DD::O_Q + idx] = q;
      }
      for (int idx = lane; idx < NP * NU; idx += NT)
      {
        const int i = idx / NU, k = idx % NU;
        lq[DD::O_B + idx] = i < 9 ? s.B[i * NU + k] : 0.0;
        double sv = 0.0;
        if (i < 9)
          for (int r = 0; r < 6; r++)
            sv += s.J[r * NXU + i] * s.WJ[r * NXU + 9 + k];
        lq[DD::O_S + idx] = sv;
      }
      for (int idx = lane; idx < NU * NU; idx += NT)
      {
        const int i = idx / NU, k = idx % NU;
        double rv = md.w_u[idx] + (i == k ? preg : 0.0);
        for (int r = 0; r < 6; r++)
          rv += s.J[r * NXU + 9 + i] * s.WJ[r * NXU + 9 + k];
        lq[DD::O_R + idx] = rv;
      }
      // dense rows: C = 0, D = A_cone on the wrench of the foot for the active rows.  A stage without an active row (soles flat on the ground:
      // most of them) would write 34 x 24 zeros over a block that is zero already: the block's flag says whether it holds anything else, and
      // the rows are written only if it or this stage does -- the content is what the unconditional form leaves (45 % of the knot's doubles;
      // the sweep's light grid never reads them in such a stage)
      double anya = 0.0;
      for (int row = 0; row < NC; row++) // (every lane reads the same flags)
        anya += s.act[row];
      const bool any_dense = anya != 0.0;
      if (any_dense || dense_dirty)
      {
        for (int idx = lane; idx < NC * NP; idx += NT)
          lq[DD::O_C + idx] = 0.0;
        for (int idx = lane; idx < NC * NU; idx += NT)
        {
          const int row = idx / NU, k = idx % NU, f = row / 17, r = row % 17;
          lq[DD::O_D + idx] = (s.act[row] != 0.0 && k / 6 == f) ? wrench_cone_entry(r, k % 6, md.mu_fric, md.Lfoot, md.Wfoot) : 0.0;
        }
      }
      if (any_dense != dense_dirty && lane == 0)
        lq[DD::O_cdirty] = any_dense ? 1.0 : 0.0;
      // vectors
      if (lane < NP)
      {
        double q = 0.0, lx = 0.0, ff = 0.0, lpd = 0.0;
        if (lane < 9)
        {
          lx = s.g[lane];
          double acc = 0.0;
          for (int i = 0; i < 9; i++)
            acc += s.A[i * 9 + lane] * s.l1[i];
          q = t == 0 ? 0.0 : lx + acc - s.l0[lane]; // x_0 is pinned (force_initial_condition_, src/mpc.cpp:53)
          dual = fmax(dual, fabs(q));
          ff = s.f[lane];
          lpd = s.lpd[lane];
        }
        lq[DD::O_q + lane] = q;
        lq[DD::O_lx + lane] = lx;
        lq[DD::O_f + lane] = ff;
        lq[DD::O_lpd + lane] = lpd;
      }
      for (int k = lane; k < NU; k += NT)
      {
        double acc = 0.0;
        for (int i = 0; i < 9; i++)
          acc += s.B[i * NU + k] * s.l1[i];
        const double r = s.g[9 + k] + acc + s.yc[k];
        lq[DD::O_r + k] = r;
        lq[DD::O_lu + k] = s.g[9 + k];
        dual = fmax(dual, fabs(r));
        // control-box rows of the dense layout: absent
        lq[DD::O_d + k] = 0.0;
        lq[DD::O_vpd + k] = 0.0;
        lq[DD::O_act + k] = 0.0;
      }
      for (int row = lane; row < NC; row += NT)
      {
        lq[DD::O_d + NU + row] = s.dvec[row];
        lq[DD::O_vpd + NU + row] = s.vpd[row];
        lq[DD::O_act + NU + row] = s.act[row];
      }
      s.red3[lane] = dual;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double pen = 0.0, prim = 0.0, dual = 0.0;
      for (int i = 0; i < 64; i++)
      {
        pen += s.red[i];
        prim = fmax(prim, s.red2[i]);
        dual = fmax(dual, s.red3[i]);
      }
      parts[0] = s.cost + pen;
      parts[1] = s.cost;
      parts[2] = prim;
      parts[3] = dual;
    }
    SMPC_LANES_END_WAVE
  }

  // ---------------------------------------------------------------------------------------------------------------
  // forward sweep + directional derivative of the merit, one wavefront per instance (the recursion of forward_full_body,
  // smpc_full_solver.h, on the padded knot / gains layout; nine real state rows)
  // ---------------------------------------------------------------------------------------------------------------
  template <class D>
  SMPC_DEV void cent6_forward_body(const Cent6Args<D> & ka, int block)
  {
    typedef typename D::DD DD;
    constexpr int NT = 64, NU = D::NU, NC = D::NC, NP = DD::NDX;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const size_t inst = (size_t)block;
    const double mu = ka.sb.model->mu;
    SMPC_LDS(double, dx, 9);
    SMPC_LDS(double, du, D::NU);
    SMPC_LDS(double, y, 9);
    SMPC_LDS(double, part, 64);
    SMPC_LDS(double, lpd_prev, 9);
    SMPC_LANES(NT)
    {
      if (lane < 9)
      {
        dx[lane] = 0.0;
        lpd_prev[lane] = 0.0;
        b.dxs[(inst * (H + 1)) * 9 + lane] = 0.0;
      }
      part[lane] = 0.0;
    }
    SMPC_LANES_END_WAVE
    // What a lane needs of a stage -- its rows of [K k], [Z z], [A | B], P~ and the vector entries beside them -- is read one stage AHEAD into
    // registers (none of it depends on the sweep's state): the loads of stage t + 1 fly while stage t computes.  (First form: every phase
    // waited for its own loads, 3 x 100 exposed round trips to device memory per launch: 0.39 ms.)
    static_assert(NU <= NT && NC <= NT, "one row of K / Z per lane");
    SMPC_PLA(double, kr, NT, 10);  // lane < NU: K row (9) | k
    SMPC_PLA(double, zr, NT, 10);  // lane < NC: Z row (9) | z
    SMPC_PLA(double, ab, NT, 9 + D::NU); // lane < 9: A row | B row
    SMPC_PLA(double, pr, NT, 9);   // lane < 9: P~ row
    SMPC_PLA(double, sv, NT, 8);   // lu_i | d_r, vpd_r | f_i, pn_i, lx_i, lpd_i | act_r
    auto fetch = [&](int t) {
      const double * lq = ka.sb.lq + (inst * H + t) * DD::LQ_STRIDE;
      const double * g = ka.sb.gains + (inst * H + t) * DD::G_STRIDE;
      SMPC_LANES(NT)
      {
        const int i = lane < NU ? lane : 0, r = lane < NC ? lane : 0, x = lane < 9 ? lane : 0;
#pragma unroll
        for (int j = 0; j < 9; j++)
        {
          SMPC_PLV(kr)[j] = g[DD::G_K + i * (NP + 1) + j];
          SMPC_PLV(zr)[j] = g[DD::G_Z + r * (NP + 1) + j];
          SMPC_PLV(ab)[j] = lq[DD::O_A + x * NP + j];
          SMPC_PLV(pr)[j] = g[DD::G_Pt + x * NP + j];
        }
        SMPC_PLV(kr)[9] = g[DD::G_K + i * (NP + 1) + NP];
        SMPC_PLV(zr)[9] = g[DD::G_Z + r * (NP + 1) + NP];
#pragma unroll
        for (int j = 0; j < NU; j++)
          SMPC_PLV(ab)[9 + j] = lq[DD::O_B + x * NU + j];
        SMPC_PLV(sv)[0] = lq[DD::O_lu + i];
        SMPC_PLV(sv)[1] = lq[DD::O_d + NU + r];
        SMPC_PLV(sv)[2] = lq[DD::O_vpd + NU + r];
        SMPC_PLV(sv)[3] = lq[DD::O_f + x];
        SMPC_PLV(sv)[4] = g[DD::G_pn + x];
        SMPC_PLV(sv)[5] = lq[DD::O_lx + x];
        SMPC_PLV(sv)[6] = lq[DD::O_lpd + x];
        SMPC_PLV(sv)[7] = lq[DD::O_act + NU + r];
      }
      SMPC_LANES_END_WAVE
    };
    SMPC_PLA(double, ck, NT, 10);
    SMPC_PLA(double, cz, NT, 10);
    SMPC_PLA(double, cab, NT, 9 + D::NU);
    SMPC_PLA(double, cp, NT, 9);
    SMPC_PLA(double, cs, NT, 8);
    fetch(0);
    for (int t = 0; t < H; t++)
    {
      const size_t lt = inst * H + t;
      SMPC_LANES(NT)
      { // this stage's operands out of the prefetch registers
#pragma unroll
        for (int j = 0; j < 10; j++)
        {
          SMPC_PLV(ck)[j] = SMPC_PLV(kr)[j];
          SMPC_PLV(cz)[j] = SMPC_PLV(zr)[j];
        }
#pragma unroll
        for (int j = 0; j < 9 + NU; j++)
          SMPC_PLV(cab)[j] = SMPC_PLV(ab)[j];
#pragma unroll
        for (int j = 0; j < 9; j++)
          SMPC_PLV(cp)[j] = SMPC_PLV(pr)[j];
#pragma unroll
        for (int j = 0; j < 8; j++)
          SMPC_PLV(cs)[j] = SMPC_PLV(sv)[j];
      }
      SMPC_LANES_END_WAVE
      if (t + 1 < H)
        fetch(t + 1);
      SMPC_LANES(NT)
      if (lane < NU)
      {
        double acc = SMPC_PLV(ck)[9];
#pragma unroll
        for (int j = 0; j < 9; j++)
          acc += SMPC_PLV(ck)[j] * dx[j];
        du[lane] = acc;
        b.dus[lt * NU + lane] = acc;
        part[lane] += SMPC_PLV(cs)[0] * acc;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        if (lane < NC)
        {
          const double d = SMPC_PLV(cs)[1];
          double dnu = SMPC_PLV(cz)[9];
#pragma unroll
          for (int j = 0; j < 9; j++)
            dnu += SMPC_PLV(cz)[j] * dx[j];
          if (SMPC_PLV(cs)[7] == 0.0)
            dnu = d / mu; // inactive row: Z = 0, z = d / mu -- the sweep writes no [Z z] for it in the stages of the light grid (what was read is stale)
          b.dvs[lt * NC + lane] = dnu;
          part[lane] += SMPC_PLV(cs)[2] * (mu * dnu - d) - d * dnu;
        }
        if (lane < 9)
        {
          double acc = 0.0;
#pragma unroll
          for (int j = 0; j < 9; j++)
            acc += SMPC_PLV(cab)[j] * dx[j];
#pragma unroll
          for (int j = 0; j < NU; j++)
            acc += SMPC_PLV(cab)[9 + j] * du[j];
          const double fi = SMPC_PLV(cs)[3], pn = SMPC_PLV(cs)[4];
          part[lane] += (SMPC_PLV(cs)[5] - lpd_prev[lane]) * dx[lane] + SMPC_PLV(cs)[6] * acc;
          y[lane] = acc + fi - mu * pn;
        }
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane < 9)
      {
        double w = 0.0;
#pragma unroll
        for (int j = 0; j < 9; j++)
          w += SMPC_PLV(cp)[j] * y[j];
        const double dxn = y[lane] - mu * w;
        const double dl = w + SMPC_PLV(cs)[4];
        b.dxs[(inst * (H + 1) + t + 1) * 9 + lane] = dxn;
        b.dlams[lt * 9 + lane] = dl;
        part[lane] -= SMPC_PLV(cs)[3] * dl;
        lpd_prev[lane] = SMPC_PLV(cs)[6];
        dx[lane] = dxn;
      }
      SMPC_LANES_END_WAVE
    }
    SMPC_LANES(NT)
    if (lane < 9)
    {
      const int sl = ring_slot(ka.head, H - 1, R);
      const double lamH = b.lams[(inst * R + sl) * 9 + lane];
      const double lxN = ka.sb.qN[inst * NP + lane] + lamH;
      part[lane] += (lxN - lpd_prev[lane]) * dx[lane];
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double sacc = 0.0;
      for (int i = 0; i < 64; i++)
        sacc += part[i];
      b.scal[inst * SC_N + SC_DPHI0] = sacc;
    }
    SMPC_LANES_END_WAVE
  }

  // ---------------------------------------------------------------------------------------------------------------
  // line search (round 5), in two kinds of launches around the host-free decision:
  //   cent6_trial_body (grid B (H + 1))  candidates j0 .. j0 + nj - 1 of one stage per wavefront: the stage inputs are read once (coalesced, into
  //                                      LDS); lane = one term of the stage merit (a wrench-cone row, a dynamics row, a row of the control cost,
  //                                      one of the five 3 x 3 quadratic costs); lane j then adds the terms of candidate j in a fixed order.
  //                                      Writes partsT[inst][j][t] = (phi, prim) and, for stages 0 and 1, the state derivative at the candidate.
  //   cent6_ls_body (grid B)             adds the stage terms of its candidates in a fixed order, Armijo test, step, regularisation update.
  // The engine launches trial(0, 1), ls(0, 1), trial(1, LS_N - 1), ls(1, LS_N - 1): an instance whose full step passes the Armijo test is
  // finished by the first pair (its blocks of the second pair exit at once); the backtracking candidates alpha = 1/2, 1/4, .. are evaluated for
  // the others only, all at once.  SC_LS_INDEX < 0 marks an instance as undecided between the pairs.
  // (The first form -- lane = stage, one candidate after the other inside cent6_ls_body -- ran 101 stage evaluations of per-lane strided loads on
  // one wavefront per instance: 0.72 ms of the 2.8 ms iteration at B = 1024.)
  // ---------------------------------------------------------------------------------------------------------------
  template <class D>
  struct Cent6TrialLds
  {
    static constexpr int NU = D::NU, NC = D::NC, NF = D::NF, LS_N = D::LS_N;
    static constexpr int NTERM = NC + 9 + NU + 5; // lanes with a term
    double x[9], dx[9], xn[9], dxn[9], l1[9], dl[9], l1e[9], xt[9];
    double u[NU], du[NU], uref[NU], v[NC], dv[NC], ve[NC], pp[3 * NF];
    double val[LS_N][NTERM], prm[LS_N][NC + 9];
  };
  template <class D>
  SMPC_DEV void cent6_trial_body(const Cent6Args<D> & ka, int block)
  {
    constexpr int NT = 64, NU = D::NU, NC = D::NC, NF = D::NF, LS_N = D::LS_N;
    constexpr int L_DYN = NC, L_CU = NC + 9, L_Q = NC + 9 + NU, L_END = L_Q + 5;
    static_assert(L_END <= NT && LS_N <= NT, "one term of the stage merit per lane");
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head, j0 = ka.j0, j1 = ka.j0 + ka.nj;
    const size_t inst = (size_t)(block / (H + 1));
    const int t = block % (H + 1);
    if (j0 > 0 && b.scal[inst * SC_N + SC_LS_INDEX] >= 0.0)
      return; // decided by an earlier candidate
    const bool term = t == H;
    const size_t ib = inst * R;
    SMPC_LDS(Cent6TrialLds<D>, ldsv, 1);
    Cent6TrialLds<D> & s = ldsv[0];
    const int st = ring_slot(head, t, R), st1 = ring_slot(head, term ? t : t + 1, R);
    const unsigned mask = term ? 0u : b.stages[t].mask;
    // (every global load before the first commit to LDS, as in cent6_deriv_body)
    SMPC_LANES(NT)
    {
      static_assert(NU <= NT && NC <= NT && 3 * NF <= NT, "one entry per lane");
      const int ts = term ? 0 : t, l9 = lane < 9 ? lane : 8;
      const double vx = b.xs[(ib + st) * 9 + l9], vdx = b.dxs[(inst * (H + 1) + t) * 9 + l9];
      const double vxn = b.xs[(ib + st1) * 9 + l9], vdxn = b.dxs[(inst * (H + 1) + (term ? t : t + 1)) * 9 + l9];
      const double vl1 = b.lams[(ib + st) * 9 + l9], vdl = b.dlams[(inst * H + ts) * 9 + l9], vl1e = b.lams_e[(ib + st) * 9 + l9];
      const double vxt = *(l9 < 3 ? &b.stages[ts].x_tgt[l9] : &b.vref[(ib + st) * 6 + l9 - 3]);
      const int lu = lane < NU ? lane : NU - 1, lc = lane < NC ? lane : NC - 1, lf = lane < 3 * NF ? lane : 3 * NF - 1;
      const double vu = b.us[(ib + st) * NU + lu], vdu = b.dus[(inst * H + ts) * NU + lu], vur = b.stages[ts].u_ref[lu];
      const double vv = b.vs[(ib + st) * NC + lc], vdv = b.dvs[(inst * H + ts) * NC + lc], vve = b.vs_e[(ib + st) * NC + lc];
      const double vpp = b.foot[(inst * H + ts) * (3 * NF) + lf];
      SMPC_SCHED_FENCE();
      if (lane < 9)
      {
        s.x[lane] = vx;
        s.dx[lane] = vdx;
        s.xn[lane] = vxn;
        s.dxn[lane] = term ? 0.0 : vdxn;
        s.l1[lane] = term ? 0.0 : vl1;
        s.dl[lane] = term ? 0.0 : vdl;
        s.l1e[lane] = term ? 0.0 : vl1e;
        s.xt[lane] = term ? 0.0 : vxt;
      }
      if (!term)
      {
        if (lane < NU)
        {
          s.u[lane] = vu;
          s.du[lane] = vdu;
          s.uref[lane] = vur;
        }
        if (lane < NC)
        {
          s.v[lane] = vv;
          s.dv[lane] = vdv;
          s.ve[lane] = vve;
        }
        if (lane < 3 * NF)
          s.pp[lane] = vpp;
      }
    }
    SMPC_LANES_END_WAVE
    const CentDevModel<D> & md = *b.model; // (the same 1.6 KB for every block: read where it lies, through the scalar / L1 caches)
    double * out = ka.partsT + (inst * LS_N * (H + 1) + t) * 2; // candidate j: + j (H + 1) 2
    auto alpha_of = [](int j) {
      double al = 1.0;
      for (int k = 0; k < j; k++)
        al *= 0.5;
      return al;
    };
    if (term)
    { // terminal node: momentum cost at x_H + alpha dx_H
      SMPC_LANES(NT)
      if (lane >= j0 && lane < j1)
      {
        const double al = alpha_of(lane);
        const V3 h = mk3(s.x[3] + al * s.dx[3], s.x[4] + al * s.dx[4], s.x[5] + al * s.dx[5]);
        const V3 L = mk3(s.x[6] + al * s.dx[6], s.x[7] + al * s.dx[7], s.x[8] + al * s.dx[8]);
        out[(size_t)lane * (H + 1) * 2] = 0.5 * dot(h, ldm3(md.w_lm) * h) + 0.5 * dot(L, ldm3(md.w_am) * L);
        out[(size_t)lane * (H + 1) * 2 + 1] = 0.0;
      }
      SMPC_LANES_END_WAVE
      return;
    }
    const double mu = md.mu, imu = 1.0 / mu, imass = 1.0 / md.mass;
    SMPC_LANES(NT)
    {
      const V3 g = ld3(md.gravity);
      // Every term is a function of alpha through a few polynomials (the point moves along a line; the only product of two moving quantities
      // is (p - c) x f in the angular-momentum rate): their coefficients once, Horner per candidate.
      //   cone row:      A_cone (u + alpha du) = c0 + alpha c1 ; multiplier v + alpha dv = e0 + alpha e1
      //   control cost:  row i of W (u + alpha du - u_ref) = c0 + alpha c1 ; (u + alpha du - u_ref)_i = e0 + alpha e1
      //   dynamics row:  defect e = c0 + alpha c1 + alpha^2 c2 ; multiplier l1 + alpha dl = e0 + alpha e1 ; xdot = r0.x + alpha r1.x + alpha^2 r2.x
      //   3 x 3 costs:   residual r0 + alpha r1 + alpha^2 r2
      double c0 = 0.0, c1 = 0.0, c2 = 0.0, e0 = 0.0, e1 = 0.0;
      V3 r0 = mk3(0, 0, 0), r1 = r0, r2 = r0;
      bool on = false;
      // sums over the feet in contact: force fs0 + alpha fs1, torque about the CoM ts0 + alpha ts1 + alpha^2 ts2
      auto contact_sums = [&](V3 & fs0, V3 & fs1, V3 & ts0, V3 & ts1, V3 & ts2) {
        fs0 = fs1 = ts0 = ts1 = ts2 = mk3(0, 0, 0);
        const V3 cc0 = ld3(s.x), cc1 = ld3(s.dx);
#pragma unroll
        for (int f = 0; f < NF; f++)
          if ((mask >> f) & 1u)
          {
            const V3 F0 = ld3(s.u + 6 * f), F1 = ld3(s.du + 6 * f), p = ld3(s.pp + 3 * f) - cc0;
            fs0 = fs0 + F0;
            fs1 = fs1 + F1;
            ts0 = ts0 + cross(p, F0) + ld3(s.u + 6 * f + 3);
            ts1 = ts1 + cross(p, F1) - cross(cc1, F0) + ld3(s.du + 6 * f + 3);
            ts2 = ts2 - cross(cc1, F1);
          }
      };
      if (lane < L_DYN)
      {
        const int row = lane, f = row / 17, r = row % 17;
        on = (mask >> f) & 1u;
        if (on)
          for (int k = 0; k < 6; k++)
          {
            const double a = wrench_cone_entry(r, k, md.mu_fric, md.Lfoot, md.Wfoot);
            c0 += a * s.u[6 * f + k];
            c1 += a * s.du[6 * f + k];
          }
        e0 = s.v[row];
        e1 = s.dv[row];
      }
      else if (lane < L_CU)
      {
        const int i = lane - L_DYN;
        V3 fs0, fs1, ts0, ts1, ts2;
        contact_sums(fs0, fs1, ts0, ts1, ts2);
        // xdot_i = r0.x + alpha r1.x + alpha^2 r2.x
        if (i < 3)
          r0 = mk3(s.x[3 + i] * imass, 0, 0), r1 = mk3(s.dx[3 + i] * imass, 0, 0);
        else if (i < 6)
          r0 = mk3(md.mass * v3c(g, i - 3) + v3c(fs0, i - 3), 0, 0), r1 = mk3(v3c(fs1, i - 3), 0, 0);
        else
          r0 = mk3(v3c(ts0, i - 6), 0, 0), r1 = mk3(v3c(ts1, i - 6), 0, 0), r2 = mk3(v3c(ts2, i - 6), 0, 0);
        c0 = s.x[i] + md.dt * r0.x - s.xn[i];
        c1 = s.dx[i] + md.dt * r1.x - s.dxn[i];
        c2 = md.dt * r2.x;
        e0 = s.l1[i];
        e1 = s.dl[i];
      }
      else if (lane < L_Q)
      {
        const int i = lane - L_CU;
        for (int k = 0; k < NU; k++)
        {
          c0 += md.w_u[i * NU + k] * (s.u[k] - s.uref[k]);
          c1 += md.w_u[i * NU + k] * s.du[k];
        }
        e0 = s.u[i] - s.uref[i];
        e1 = s.du[i];
      }
      else if (lane < L_END)
      {
        const int k = lane - L_Q;
        if (k < 3)
          r0 = ld3(s.x + 3 * k) - ld3(s.xt + 3 * k), r1 = ld3(s.dx + 3 * k);
        else
        {
          V3 fs0, fs1, ts0, ts1, ts2;
          contact_sums(fs0, fs1, ts0, ts1, ts2);
          if (k == 3)
            r0 = g + imass * fs0, r1 = imass * fs1;
          else
            r0 = ts0, r1 = ts1, r2 = ts2;
        }
      }
      const double * Wq = lane == L_Q ? md.w_com : (lane == L_Q + 1 ? md.w_lm : (lane == L_Q + 2 ? md.w_am : (lane == L_Q + 3 ? md.w_la : md.w_aa)));
      const M3 Wm = (lane >= L_Q && lane < L_END) ? ldm3(Wq) : M3{};
      const double vel = lane < L_DYN ? s.ve[lane] : 0.0, l1e = (lane >= L_DYN && lane < L_CU) ? s.l1e[lane - L_DYN] : 0.0;
      for (int j = j0; j < j1; j++)
      {
        const double al = alpha_of(j);
        double val = 0.0, prm = 0.0;
        if (lane < L_DYN)
        { // penalty of the multiplier estimate of a cone row
          double vp = 0.0;
          if (on)
          {
            const double cv = c0 + al * c1;
            const double z = cv + mu * vel;
            vp = (z - fmin(z, 0.0)) * imu;
            prm = fmax(cv, 0.0);
          }
          const double d = vp - (e0 + al * e1);
          val = 0.5 * mu * (vp * vp + d * d);
        }
        else if (lane < L_CU)
        { // dynamics row: defect e_i and its multiplier estimate
          const double e = c0 + al * (c1 + al * c2);
          const double lp = l1e + e * imu, dd = lp - (e0 + al * e1);
          val = 0.5 * mu * (lp * lp + dd * dd);
          prm = fabs(e);
          if (t < 2)
            ka.xdotT[((inst * LS_N + j) * 2 + t) * 9 + lane - L_DYN] = r0.x + al * (r1.x + al * r2.x);
        }
        else if (lane < L_Q)
          val = (e0 + al * e1) * (c0 + al * c1);
        else if (lane < L_END)
        {
          const V3 r = r0 + al * (r1 + al * r2);
          val = 0.5 * dot(r, Wm * r);
        }
        if (lane < L_END)
          s.val[j][lane] = val;
        if (lane < L_CU)
          s.prm[j][lane] = prm;
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane >= j0 && lane < j1)
    { // candidate `lane`: penalty terms row by row, then the costs
      const double * vj = s.val[lane], *pj = s.prm[lane];
      double pen = 0.0, prim = 0.0, cu = 0.0;
      for (int i = 0; i < L_CU; i++)
      {
        pen += vj[i];
        prim = fmax(prim, pj[i]);
      }
      for (int i = L_CU; i < L_Q; i++)
        cu += vj[i];
      double cost = vj[L_Q];
      cost += 0.5 * cu;
      cost += vj[L_Q + 1];
      cost += vj[L_Q + 2];
      cost += vj[L_Q + 3];
      cost += vj[L_Q + 4];
      out[(size_t)lane * (H + 1) * 2] = cost + pen;
      out[(size_t)lane * (H + 1) * 2 + 1] = prim;
    }
    SMPC_LANES_END_WAVE
  }

  template <class D>
  SMPC_DEV void cent6_ls_body(const Cent6Args<D> & ka, int block)
  {
    constexpr int NT = 64, NU = D::NU, NC = D::NC, LS_N = D::LS_N;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head, j0 = ka.j0, j1 = ka.j0 + ka.nj;
    const size_t inst = (size_t)block, ib = inst * R;
    SMPC_LDS(double, sp, LS_N * NT * 2); // per candidate: (phi, prim) summed over the stages lane, lane + 64, ..
    SMPC_LDS(double, sp0, NT * 4);       // phi, cost, prim, dual at the current point, the same partial sums
    SMPC_LDS(double, cand, 2 * LS_N);
    SMPC_LDS(double, res, 8);
    double * sc = b.scal + inst * SC_N;
    if (j0 > 0 && sc[SC_LS_INDEX] >= 0.0)
      return; // decided by an earlier candidate
    // merits: stage terms added in a fixed order -- lane L adds its stages L, L + 64, .., then one lane per candidate adds the 64 partial sums
    SMPC_LANES(NT)
    {
      for (int j = j0; j < j1; j++)
      {
        const double * src = ka.partsT + (inst * LS_N + j) * (H + 1) * 2;
        double phi = 0.0, prim = 0.0;
        for (int t = lane; t <= H; t += NT)
        {
          phi += src[2 * t];
          prim = fmax(prim, src[2 * t + 1]);
        }
        sp[(j * NT + lane) * 2] = phi;
        sp[(j * NT + lane) * 2 + 1] = prim;
      }
      if (j0 == 0)
      {
        const double * src0 = ka.parts0 + inst * (H + 1) * 4;
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        for (int t = lane; t <= H; t += NT)
        {
          p0 += src0[4 * t];
          p1 += src0[4 * t + 1];
          p2 = fmax(p2, src0[4 * t + 2]);
          p3 = fmax(p3, src0[4 * t + 3]);
        }
        sp0[lane * 4] = p0;
        sp0[lane * 4 + 1] = p1;
        sp0[lane * 4 + 2] = p2;
        sp0[lane * 4 + 3] = p3;
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      if (lane >= j0 && lane < j1)
      {
        double phi = 0.0, prim = 0.0;
        for (int l = 0; l < NT; l++)
        {
          phi += sp[(lane * NT + l) * 2];
          prim = fmax(prim, sp[(lane * NT + l) * 2 + 1]);
        }
        cand[2 * lane] = phi;
        cand[2 * lane + 1] = prim;
      }
      if (lane == 32)
      {
        if (j0 == 0)
        {
          double phi = 0.0, cost = 0.0, prim = 0.0, dual = 0.0;
          for (int l = 0; l < NT; l++)
          {
            const double * p = sp0 + l * 4;
            phi += p[0];
            cost += p[1];
            prim = fmax(prim, p[2]);
            dual = fmax(dual, p[3]);
          }
          sc[SC_PHI0] = phi;
          sc[SC_COST] = cost;
          sc[SC_PRIM] = prim;
          sc[SC_DUAL] = dual;
          res[0] = phi;
        }
        else
          res[0] = sc[SC_PHI0];
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      const double phi0 = res[0], dphi0 = sc[SC_DPHI0];
      res[2] = -1.0;
      for (int j = j0; j < j1; j++)
      {
        double alpha = 1.0;
        for (int k = 0; k < j; k++)
          alpha *= 0.5;
        const bool ok = cand[2 * j] <= phi0 + ka.armijo_c1 * alpha * dphi0;
        if (ok || j == LS_N - 1)
        {
          sc[SC_LS_FAILED] = ok ? 0.0 : 1.0;
          sc[SC_ALPHA] = alpha;
          sc[SC_PHI_NEW] = cand[2 * j];
          sc[SC_PRIM_NEW] = cand[2 * j + 1];
          const double preg = sc[SC_PREG];
          sc[SC_PREG] = ok ? fmax(preg * ka.reg_dec, ka.reg_min) : fmin(preg * ka.reg_inc, ka.reg_max);
          res[1] = alpha;
          res[2] = (double)j;
          break;
        }
      }
      sc[SC_LS_INDEX] = res[2]; // (-1: undecided, the launch of the next candidates takes over)
    }
    SMPC_LANES_END_WAVE
    if (res[2] < 0.0)
      return;
    const double alpha = res[1];
    const int sel = (int)res[2];
    // ---- state derivative of stages 0, 1 at the accepted point (MPC::getStateDerivative), then the step (coalesced over the horizon) ----
    SMPC_LANES(NT)
    {
      if (lane < 18)
        b.xdot01[inst * 18 + lane] = ka.xdotT[(inst * LS_N + sel) * 18 + lane];
      for (int idx = lane; idx < (H + 1) * 9; idx += NT)
        b.xs[(ib + ring_slot(head, idx / 9, R)) * 9 + idx % 9] += alpha * b.dxs[inst * (H + 1) * 9 + idx];
      for (int idx = lane; idx < H * NU; idx += NT)
        b.us[(ib + ring_slot(head, idx / NU, R)) * NU + idx % NU] += alpha * b.dus[inst * H * NU + idx];
      for (int idx = lane; idx < H * NC; idx += NT)
        b.vs[(ib + ring_slot(head, idx / NC, R)) * NC + idx % NC] += alpha * b.dvs[inst * H * NC + idx];
      for (int idx = lane; idx < H * 9; idx += NT)
        b.lams[(ib + ring_slot(head, idx / 9, R)) * 9 + idx % 9] += alpha * b.dlams[inst * H * 9 + idx];
    }
    SMPC_LANES_END_WAVE
  }

  // (frontend_full_body: smpc_full_stage.h)
} // namespace smpc
