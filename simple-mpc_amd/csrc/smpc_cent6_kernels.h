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
//   cent6_ls_body       B blocks          Armijo backtracking (lane = stage trial evaluations), step, regularisation update
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
    static constexpr int LQ_STRIDE = ((O_act + NC + 7) / 8) * 8;
    static constexpr int G_K = 0;
    static constexpr int G_Z = G_K + NU * (NDX + 1);
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
    int head;
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
  SMPC_HD double v3c(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

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
    double * parts = ka.parts0 + (inst * (H + 1) + t) * 4;
    SMPC_LANES(NT)
    {
      constexpr int N = (int)(sizeof(CentDevModel<D>) / sizeof(double));
      const alias_double * src = reinterpret_cast<const alias_double *>(b.model);
      alias_double * dst = reinterpret_cast<alias_double *>(&s.md);
      for (int i = lane; i < N; i += NT)
        dst[i] = src[i];
      if (lane < 9)
      {
        s.x[lane] = b.xs[(ib + st) * 9 + lane];
        s.xn[lane] = b.xs[(ib + st1) * 9 + lane];
        s.l1[lane] = term ? 0.0 : b.lams[(ib + st) * 9 + lane];
        s.l1e[lane] = term ? 0.0 : b.lams_e[(ib + st) * 9 + lane];
        s.l0[lane] = t > 0 ? b.lams[(ib + stm) * 9 + lane] : 0.0;
        s.xt[lane] = term ? 0.0 : (lane < 3 ? b.stages[t].x_tgt[lane] : b.vref[(ib + st) * 6 + lane - 3]);
      }
      if (!term)
      {
        for (int i = lane; i < NU; i += NT)
        {
          s.u[i] = b.us[(ib + st) * NU + i];
          s.uref[i] = b.stages[t].u_ref[i];
        }
        for (int i = lane; i < NC; i += NT)
        {
          s.v[i] = b.vs[(ib + st) * NC + i];
          s.ve[i] = b.vs_e[(ib + st) * NC + i];
        }
        for (int i = lane; i < 3 * NF; i += NT)
          s.pp[i] = b.foot[(inst * H + t) * (3 * NF) + i];
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
              v += (i < 6 ? md.w_lm : md.w_am)[(i % 3) * 3 + j % 3];
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
            q += (i < 3 ? md.w_com : (i < 6 ? md.w_lm : md.w_am))[(i % 3) * 3 + j % 3];
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
      // dense rows: C = 0, D = A_cone on the wrench of the foot for the active rows
      for (int idx = lane; idx < NC * NP; idx += NT)
        lq[DD::O_C + idx] = 0.0;
      for (int idx = lane; idx < NC * NU; idx += NT)
      {
        const int row = idx / NU, k = idx % NU, f = row / 17, r = row % 17;
        lq[DD::O_D + idx] = (s.act[row] != 0.0 && k / 6 == f) ? wrench_cone_entry(r, k % 6, md.mu_fric, md.Lfoot, md.Wfoot) : 0.0;
      }
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
    for (int t = 0; t < H; t++)
    {
      const double * lq = ka.sb.lq + (inst * H + t) * DD::LQ_STRIDE;
      const double * g = ka.sb.gains + (inst * H + t) * DD::G_STRIDE;
      const size_t lt = inst * H + t;
      SMPC_LANES(NT)
      for (int i = lane; i < NU; i += NT)
      {
        const double * Kr = g + DD::G_K + i * (NP + 1);
        double acc = Kr[NP];
        for (int j = 0; j < 9; j++)
          acc += Kr[j] * dx[j];
        du[i] = acc;
        b.dus[lt * NU + i] = acc;
        part[lane] += lq[DD::O_lu + i] * acc;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      {
        for (int r = lane; r < NC; r += NT)
        {
          const double * Zr = g + DD::G_Z + r * (NP + 1);
          double dnu = Zr[NP];
          for (int j = 0; j < 9; j++)
            dnu += Zr[j] * dx[j];
          const double d = lq[DD::O_d + NU + r];
          b.dvs[lt * NC + r] = dnu;
          part[lane] += lq[DD::O_vpd + NU + r] * (mu * dnu - d) - d * dnu;
        }
        if (lane < 9)
        {
          const double * Ar = lq + DD::O_A + lane * NP;
          const double * Br = lq + DD::O_B + lane * NU;
          double acc = 0.0;
          for (int j = 0; j < 9; j++)
            acc += Ar[j] * dx[j];
          for (int j = 0; j < NU; j++)
            acc += Br[j] * du[j];
          const double fi = lq[DD::O_f + lane], pn = g[DD::G_pn + lane];
          part[lane] += (lq[DD::O_lx + lane] - lpd_prev[lane]) * dx[lane] + lq[DD::O_lpd + lane] * acc;
          y[lane] = acc + fi - mu * pn;
        }
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane < 9)
      {
        const double * Pr = g + DD::G_Pt + lane * NP;
        double w = 0.0;
        for (int j = 0; j < 9; j++)
          w += Pr[j] * y[j];
        const double dxn = y[lane] - mu * w;
        const double dl = w + g[DD::G_pn + lane];
        b.dxs[(inst * (H + 1) + t + 1) * 9 + lane] = dxn;
        b.dlams[lt * 9 + lane] = dl;
        part[lane] -= lq[DD::O_f + lane] * dl;
        lpd_prev[lane] = lq[DD::O_lpd + lane];
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

  // merit terms of stage t at the trial point w + alpha dw (cost, penalty, primal infeasibility; xdot optionally)
  template <class D>
  SMPC_DEV void cent6_stage_merit(const CentDevModel<D> & md, const CentBuffers<D> & b, size_t inst, int head, int t, double al, double & cost, double & pen,
                                  double & prim, double * xdot)
  {
    constexpr int NU = D::NU, NC = D::NC, NF = D::NF;
    const int H = b.H, R = b.R;
    const size_t ib = inst * R;
    const int st = ring_slot(head, t, R), st1 = ring_slot(head, t + 1, R);
    const double * x = b.xs + (ib + st) * 9, *xn = b.xs + (ib + st1) * 9, *dx = b.dxs + (inst * (H + 1) + t) * 9, *dxn = dx + 9;
    const double * u = b.us + (ib + st) * NU, *du = b.dus + (inst * H + t) * NU;
    const double * v = b.vs + (ib + st) * NC, *dv = b.dvs + (inst * H + t) * NC, *ve = b.vs_e + (ib + st) * NC;
    const double * l1 = b.lams + (ib + st) * 9, *dl = b.dlams + (inst * H + t) * 9, *l1e = b.lams_e + (ib + st) * 9;
    const double * pp = b.foot + (inst * H + t) * (3 * NF);
    const double * uref = b.stages[t].u_ref;
    const unsigned mask = b.stages[t].mask;
    const double mu = md.mu, imu = 1.0 / mu, imass = 1.0 / md.mass;
    auto X = [&](int i) { return x[i] + al * dx[i]; };
    auto U = [&](int i) { return u[i] + al * du[i]; };
    const Cent6Point<D> q = cent6_point<D>(mask, pp, X, U);
    const V3 g = ld3(md.gravity);
    pen = 0.0;
    prim = 0.0;
    for (int row = 0; row < NC; row++)
    {
      const int f = row / 17, r = row % 17;
      double vp = 0.0;
      if ((mask >> f) & 1u)
      {
        double cv = 0.0;
        for (int j = 0; j < 6; j++)
          cv += wrench_cone_entry(r, j, md.mu_fric, md.Lfoot, md.Wfoot) * U(6 * f + j);
        const double z = cv + mu * ve[row];
        vp = (z - fmin(z, 0.0)) * imu;
        prim = fmax(prim, fmax(cv, 0.0));
      }
      const double d = vp - (v[row] + al * dv[row]);
      pen += 0.5 * mu * (vp * vp + d * d);
    }
    double xd[9];
    for (int k = 0; k < 3; k++)
    {
      xd[k] = X(3 + k) * imass;
      xd[3 + k] = md.mass * v3c(g, k) + v3c(q.fs, k);
      xd[6 + k] = v3c(q.ts, k);
    }
    for (int i = 0; i < 9; i++)
    {
      const double e = X(i) + md.dt * xd[i] - (xn[i] + al * dxn[i]);
      const double lp = l1e[i] + e * imu, dd = lp - (l1[i] + al * dl[i]);
      pen += 0.5 * mu * (lp * lp + dd * dd);
      prim = fmax(prim, fabs(e));
      if (xdot)
        xdot[i] = xd[i];
    }
    auto quad3 = [](const double * W, V3 r) { return 0.5 * dot(r, ldm3(W) * r); };
    const V3 c = mk3(X(0), X(1), X(2)), h = mk3(X(3), X(4), X(5)), L = mk3(X(6), X(7), X(8));
    const double * href = b.vref + (ib + st) * 6;
    cost = quad3(md.w_com, c - ld3(b.stages[t].x_tgt));
    double cu = 0.0;
    for (int i = 0; i < NU; i++)
    {
      double wr = 0.0;
      for (int j = 0; j < NU; j++)
        wr += md.w_u[i * NU + j] * (U(j) - uref[j]);
      cu += (U(i) - uref[i]) * wr;
    }
    cost += 0.5 * cu;
    cost += quad3(md.w_lm, h - ld3(href));
    cost += quad3(md.w_am, L - ld3(href + 3));
    cost += quad3(md.w_la, g + imass * q.fs);
    cost += quad3(md.w_aa, q.ts);
  }

  // ---------------------------------------------------------------------------------------------------------------
  // line search + step: one wavefront per instance; lane = stage trial evaluations, fixed-order reductions; Armijo backtracking
  // alpha = 1, 1/2, .. (the last candidate is taken on failure), regularisation update, x <- x + alpha dx, ...
  // ---------------------------------------------------------------------------------------------------------------
  template <class D>
  SMPC_DEV void cent6_ls_body(const Cent6Args<D> & ka, int block)
  {
    constexpr int NT = 64, NU = D::NU, NC = D::NC;
    const CentBuffers<D> & b = ka.b;
    const int H = b.H, R = b.R, head = ka.head;
    const size_t inst = (size_t)block, ib = inst * R;
    constexpr int MAXS = 256;
    SMPC_LDS(double, sphi, MAXS);
    SMPC_LDS(double, sprim, MAXS);
    SMPC_LDS(double, res, 8);
    SMPC_LDS(CentDevModel<D>, mds, 1);
    double * sc = b.scal + inst * SC_N;
    SMPC_LANES(NT)
    {
      constexpr int N = (int)(sizeof(CentDevModel<D>) / sizeof(double));
      const alias_double * src = reinterpret_cast<const alias_double *>(b.model);
      alias_double * dst = reinterpret_cast<alias_double *>(&mds[0]);
      for (int i = lane; i < N; i += NT)
        dst[i] = src[i];
      if (lane == 0)
      {
        double phi = 0.0, cost = 0.0, prim = 0.0, dual = 0.0;
        for (int t = 0; t <= H; t++)
        {
          const double * p = ka.parts0 + (inst * (H + 1) + t) * 4;
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
    }
    SMPC_LANES_END_WAVE
    const CentDevModel<D> & md = mds[0];
    const double phi0 = res[0], dphi0 = sc[SC_DPHI0];
    double alpha = 1.0;
    int sel = -1;
    for (int j = 0; j < D::LS_N; j++)
    {
      SMPC_LANES(NT)
      for (int t = lane; t <= H; t += NT)
      {
        double cost, pen = 0.0, prim = 0.0;
        if (t < H)
          cent6_stage_merit<D>(md, b, inst, head, t, alpha, cost, pen, prim, (double *)nullptr);
        else
        { // terminal node: momentum cost at x_H + alpha dx_H
          const double * xH = b.xs + (ib + ring_slot(head, H, R)) * 9, *dxH = b.dxs + (inst * (H + 1) + H) * 9;
          const V3 h = mk3(xH[3] + alpha * dxH[3], xH[4] + alpha * dxH[4], xH[5] + alpha * dxH[5]);
          const V3 L = mk3(xH[6] + alpha * dxH[6], xH[7] + alpha * dxH[7], xH[8] + alpha * dxH[8]);
          cost = 0.5 * dot(h, ldm3(md.w_lm) * h) + 0.5 * dot(L, ldm3(md.w_am) * L);
        }
        sphi[t] = cost + pen;
        sprim[t] = prim;
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane == 0)
      {
        double phi = 0.0, prim = 0.0;
        for (int t = 0; t <= H; t++)
        {
          phi += sphi[t];
          prim = fmax(prim, sprim[t]);
        }
        res[1] = phi;
        res[2] = prim;
      }
      SMPC_LANES_END_WAVE
      const bool ok = res[1] <= phi0 + ka.armijo_c1 * alpha * dphi0;
      if (ok || j == D::LS_N - 1)
      {
        sel = j;
        SMPC_LANES(NT)
        if (lane == 0)
        {
          sc[SC_LS_FAILED] = ok ? 0.0 : 1.0;
          sc[SC_ALPHA] = alpha;
          sc[SC_PHI_NEW] = res[1];
          sc[SC_PRIM_NEW] = res[2];
          sc[SC_LS_INDEX] = (double)j;
          const double preg = sc[SC_PREG];
          sc[SC_PREG] = ok ? fmax(preg * ka.reg_dec, ka.reg_min) : fmin(preg * ka.reg_inc, ka.reg_max);
        }
        SMPC_LANES_END_WAVE
        break;
      }
      alpha *= 0.5;
    }
    (void)sel;
    // ---- state derivative of stages 0, 1 at the accepted point (MPC::getStateDerivative), then the step ----
    SMPC_LANES(NT)
    if (lane < 2)
    {
      double cost, pen, prim, xd[9];
      cent6_stage_merit<D>(md, b, inst, head, lane, alpha, cost, pen, prim, xd);
      for (int i = 0; i < 9; i++)
        b.xdot01[(inst * 2 + lane) * 9 + i] = xd[i];
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    for (int t = lane; t <= H; t += NT)
    {
      const size_t sl = ib + ring_slot(head, t, R);
      for (int i = 0; i < 9; i++)
        b.xs[sl * 9 + i] += alpha * b.dxs[(inst * (H + 1) + t) * 9 + i];
      if (t < H)
      {
        for (int i = 0; i < NU; i++)
          b.us[sl * NU + i] += alpha * b.dus[(inst * H + t) * NU + i];
        for (int i = 0; i < NC; i++)
          b.vs[sl * NC + i] += alpha * b.dvs[(inst * H + t) * NC + i];
        for (int i = 0; i < 9; i++)
          b.lams[sl * 9 + i] += alpha * b.dlams[(inst * H + t) * 9 + i];
      }
    }
    SMPC_LANES_END_WAVE
    static_assert(MAXS >= 201, "stages per instance");
  }

  // (frontend_full_body: smpc_full_stage.h)
} // namespace smpc
