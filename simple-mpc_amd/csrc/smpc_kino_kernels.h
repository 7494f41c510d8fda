// smpc_kino_kernels.h -- kernel bodies that run one wavefront per (instance, stage):
//   deriv_body : HOT(1)+(2)+(3) at the current iterate -> LQ knot in HBM + merit partials
//   trial_body : HOT(6) one line-search candidate: LINEAR rollout point (reference src/mpc.cpp:44),
//                re-evaluation, AL multipliers, merit partials
// (reference call site: SolverProxDDP::run, src/mpc.cpp:212; stage composition src/kinodynamics.cpp:40-152)
#pragma once
#include "smpc_kino_stage.h"

namespace smpc
{
  template <class D>
  struct StageKernelArgs
  {
    Buffers<D> b;
    int head;
    int j0, nj; // trial kernel: candidate range
    int slots;  // trial kernel: 0 = block per (instance, stage); > 0 = `slots` instance slots walking und_list
    double * wide = nullptr; // full-dynamics derivative kernel (D::WIDE_DEV): device scratch, one FullDerivWide slice per block OF THE GRID (smpc_full_stage.h)
    int nwork = 0, nres = 0; // the same kernel: nwork (instance-slot, stage) items on a grid of nres persistent blocks (0: one item per block)
  };

  // x (+) alpha*dx into dst (NX); SE3 part by `se3lane`, vector part by lanes
  template <class D>
  SMPC_DEV void lanes_integrate(const double * x, const double * dx, double alpha, double * dst, int lane, int se3lane)
  {
    constexpr int NV = D::NV, NQ = D::NQ;
    if (lane == se3lane)
    {
      const V3 dv = alpha * ld3(dx), dw = alpha * ld3(dx + 3);
      const M3 R0 = quat_to_R(Quat{x[3], x[4], x[5], x[6]});
      const SE3 E = exp6(dv, dw);
      st3(dst, ld3(x) + R0 * E.p);
      Quat qn = quat_mul(Quat{x[3], x[4], x[5], x[6]}, quat_exp(dw));
      const double n = 1.0 / sqrt(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
      dst[3] = qn.x * n;
      dst[4] = qn.y * n;
      dst[5] = qn.z * n;
      dst[6] = qn.w * n;
    }
    for (int i = 6 + lane; i < 2 * NV; i += 64)
    {
      if (i < NV)
        dst[i + 1] = x[i + 1] + alpha * dx[i];
      else
        dst[NQ + i - NV] = x[NQ + i - NV] + alpha * dx[i];
    }
  }
  // semi-implicit Euler step of a simulated robot, in place: v <- v + a dt ; q <- q (+) v dt   (one wavefront per robot)
  template <class D>
  struct SimStepArgs
  {
    double * X;       // [n][NX] states, updated in place
    const double * a; // [n][NV] accelerations
    double dt;
  };
  template <class D>
  SMPC_DEV void sim_integrate_body(const SimStepArgs<D> & ka, int block)
  {
    constexpr int NT = 64, NV = D::NV, NQ = D::NQ, NX = D::NX;
    SMPC_LDS(double, dx, 2 * NV);
    double * x = ka.X + (size_t)block * NX;
    const double * a = ka.a + (size_t)block * NV;
    SMPC_LANES(NT)
    for (int i = lane; i < NV; i += NT)
    {
      const double dv = a[i] * ka.dt;
      dx[NV + i] = dv;
      dx[i] = (x[NQ + i] + dv) * ka.dt;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    lanes_integrate<D>(x, dx, 1.0, x, lane, 0);
    SMPC_LANES_END_WAVE
  }
  // e = xa (-) xb  (tangent at xb): SE3 part by se3lane, rest by lanes
  template <class D>
  SMPC_DEV void lanes_difference(const double * xb, const double * xa, double * e, int lane, int se3lane)
  {
    constexpr int NV = D::NV, NQ = D::NQ;
    if (lane == se3lane)
    {
      const SE3 Mb{quat_to_R(Quat{xb[3], xb[4], xb[5], xb[6]}), ld3(xb)};
      const SE3 Ma{quat_to_R(Quat{xa[3], xa[4], xa[5], xa[6]}), ld3(xa)};
      V3 v, w;
      log6(se3_mul(se3_inv(Mb), Ma), v, w);
      st3(e, v);
      st3(e + 3, w);
    }
    for (int i = 6 + lane; i < 2 * NV; i += 64)
    {
      if (i < NV)
        e[i] = xa[i + 1] - xb[i + 1];
      else
        e[i] = xa[NQ + i - NV] - xb[NQ + i - NV];
    }
  }

  // =============================================================================================
  // deriv_body: grid = B * (H+1); block (inst, t); t == H is the terminal node.
  // =============================================================================================
  template <class D, bool EXT = false>
  SMPC_DEV void deriv_one(const StageKernelArgs<D> & ka, int inst, int t);

  // grid = B * (H+1) (slots == 0) or slots * (H+1) walking the compacted list of instances that rejected the
  // tentative full step (slots > 0) -- one loop, one inlined copy of the stage body
  // EXT: the problem has optional constraint blocks (friction cones, land rows, terminal constraint); the default instantiation
  // carries none of their code
  template <class D, bool EXT = false>
  SMPC_DEV void deriv_body(const StageKernelArgs<D> & ka, int block)
  {
    const int H = ka.b.H;
    const int slot = block / (H + 1), t = block % (H + 1);
    const int count = ka.slots > 0 ? ka.b.und_list[ka.b.B] : slot + 1;
    const int stride = ka.slots > 0 ? ka.slots : ka.b.B;
    for (int m = slot; m < count; m += stride)
      deriv_one<D, EXT>(ka, ka.slots > 0 ? ka.b.und_list[m] : m, t);
  }

  // ---- terminal node of the derivative pass: Q_N, q_N (+ the folded terminal constraint).  The scratch holds the derivative columns,
  //      Jl, Wrx, Whg, red[0] of the terminal evaluation ----
  template <class D, bool EXT>
  SMPC_DEV void deriv_terminal_node(const StageKernelArgs<D> & ka, KinoScratch<D, true> & sc, int inst, int t, double preg, double * parts)
  {
    constexpr int NT = 64;
    constexpr int NV = D::NV, NDX = D::NDX, NF = D::NF;
    const Buffers<D> & b = ka.b;
    const int R = b.R;
    const DevModel<D> & mg = *b.model;
    const DevModelSmall<D> & md = sc.ml;
    const size_t ib = (size_t)inst * R;
    const int sprev = ring_slot(ka.head, t > 0 ? t - 1 : 0, R);
      // ---- terminal node: Q_N = Lxx + preg I, q_N = lx - lambda_H ----
      SMPC_LANES(NT)
      {
        for (int idx = lane; idx < NDX * 6; idx += NT)
        {
          const int a = idx / 6, k = idx % 6;
          double s = 0.0;
          for (int bb = 0; bb < 6; bb++)
            s += mg.w_x[a * NDX + bb] * sc.Jl[bb * 6 + k];
          sc.WJl[idx] = s;
        }
        for (int idx = lane; idx < 6 * NDX; idx += NT)
        {
          const int a = idx / NDX, k = idx % NDX;
          double s = 0.0;
          for (int bb = 0; bb < 6; bb++)
            s += 10.0 * md.w_cent[a * 6 + bb] * (k < NV ? sc.dh_dq[bb * NV + k] : sc.Ag[bb * NV + k - NV]);
          sc.WJc()[idx] = s;
        }
      }
      SMPC_LANES_END_WAVE
      double * QN = b.QN + (size_t)inst * NDX * NDX;
      double * qN = b.qN + (size_t)inst * NDX;
      // terminal constraint c = com + tau vcom - ref (DCMPositionResidual): rows C = [Jcom + tau dvcom/dq | tau Jcom] into the (dead)
      // weighted-Jacobian block, v+ = v_e + c / mu; folded below: Q_N += C^T C / mu, q_N += C^T v+
      const bool tcs = EXT && b.CN != nullptr;
      double * tC = sc.Jfoot; // 3 x NDX, then v | v+ | c  (the terminal node has no foot rows)
      static_assert(NF * 3 * NV >= 3 * NDX + 9, "terminal constraint rows fit the foot Jacobian block");
      if (tcs)
      {
        SMPC_LANES(NT)
        {
          const double im = 1.0 / md.total_mass, tau = b.dcm_tau;
          double cr[3];
          if (lane < NDX)
            for (int r = 0; r < 3; r++)
              cr[r] = lane < NV ? (sc.Ag[r * NV + lane] + tau * sc.dh_dq[r * NV + lane]) * im : tau * sc.Ag[r * NV + lane - NV] * im;
          double v = 0.0, vp = 0.0, c = 0.0;
          if (lane < 3)
          {
            c = sc.com[lane] + tau * sc.hg[lane] * im - b.dcm_ref[(size_t)inst * 3 + lane];
            v = b.vN[(size_t)inst * 3 + lane];
            vp = b.vN_e[(size_t)inst * 3 + lane] + c / md.mu;
          }
          if (lane < NDX)
            for (int r = 0; r < 3; r++)
            {
              tC[r * NDX + lane] = cr[r];
              b.CN[(size_t)inst * (3 * NDX + 3) + r * NDX + lane] = cr[r];
            }
          if (lane < 3)
          {
            tC[3 * NDX + lane] = v;
            tC[3 * NDX + 3 + lane] = vp;
            tC[3 * NDX + 6 + lane] = c;
            b.CN[(size_t)inst * (3 * NDX + 3) + 3 * NDX + lane] = md.mu * (vp - v);
          }
        }
        SMPC_LANES_END_WAVE
      }
      SMPC_LANES(NT)
      if (lane < NDX)
      {
        const int k = lane;
        double dual = 0.0;
        // gradient
        double g = 0.0;
        if (k < 6)
          for (int a = 0; a < 6; a++)
            g += sc.Jl[a * 6 + k] * sc.Wrx[a];
        else
          g = sc.Wrx[k];
        for (int a = 0; a < 6; a++)
          g += (k < NV ? sc.dh_dq[a * NV + k] : sc.Ag[a * NV + k - NV]) * sc.Whg[a];
        double qn = g - b.lams[(ib + sprev) * NDX + k];
        if (tcs)
        {
          for (int r = 0; r < 3; r++)
            qn += tC[r * NDX + k] * tC[3 * NDX + r];
          dual = fabs(qn); // dual residual with the current multipliers; the Newton right-hand side uses v+
          for (int r = 0; r < 3; r++)
            qn += tC[r * NDX + k] * (md.mu * (tC[3 * NDX + 3 + r] - tC[3 * NDX + r])) / md.mu;
        }
        else
          dual = fabs(qn);
        qN[k] = qn;
        sc.rx[k] = dual; // reuse as dual-infeasibility scratch
        for (int i = 0; i < NDX; i++)
        {
          double v;
          if (i < 6)
          {
            v = 0.0;
            for (int a = 0; a < 6; a++)
              v += sc.Jl[a * 6 + i] * (k < 6 ? sc.WJl[a * 6 + k] : mg.w_x[a * NDX + k]);
          }
          else
            v = k < 6 ? sc.WJl[i * 6 + k] : mg.w_x[i * NDX + k];
          for (int a = 0; a < 6; a++)
            v += (i < NV ? sc.dh_dq[a * NV + i] : sc.Ag[a * NV + i - NV]) * sc.WJc()[a * NDX + k];
          if (i == k)
            v += preg;
          if (tcs)
            for (int r = 0; r < 3; r++)
              v += tC[r * NDX + i] * tC[r * NDX + k] / md.mu;
          QN[i * NDX + k] = v;
        }
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      if (lane == 0)
      {
        double dual = 0.0;
        for (int k = 0; k < NDX; k++)
          dual = fmax(dual, sc.rx[k]);
        double pen = 0.0, prim = 0.0;
        if (tcs)
          for (int r = 0; r < 3; r++)
          {
            const double vp = tC[3 * NDX + 3 + r], dv = vp - tC[3 * NDX + r];
            pen += 0.5 * md.mu * (vp * vp + dv * dv);
            prim = fmax(prim, fabs(tC[3 * NDX + 6 + r]));
          }
        parts[0] = sc.red[0] + pen;
        parts[1] = sc.red[0];
        parts[2] = prim;
        parts[3] = dual;
      }
      SMPC_LANES_END_WAVE
  }

  // ---- LQ knot of a running stage from the scratch of the derivative pass: [A | B], gradients, Q, S, R on the matrix cores, C, d.
  //      The scratch holds the derivative columns, ab_d*, the SE(3) Jacobians, weighted residuals, multiplier estimates, red[0..2] ----
  template <class D, bool EXT>
  SMPC_DEV void deriv_stage_knot(const StageKernelArgs<D> & ka, KinoScratch<D, true> & sc, const StageIn<D> & in, int inst, int t, double preg,
                                 double * parts, bool cones, unsigned land, long long & tprev)
  {
    constexpr int NT = 64;
    constexpr int NV = D::NV, NDX = D::NDX, NU = D::NU, NC = D::NC, NF = D::NF, NA = D::NA;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const DevModel<D> & mg = *b.model;
    const DevModelSmall<D> & md = sc.ml;
    const size_t ib = (size_t)inst * R;
    const int sprev = ring_slot(ka.head, t > 0 ? t - 1 : 0, R);
    double * lq = b.lq + ((size_t)inst * H + t) * D::LQ_STRIDE;
    const double dt = md.dt;
    const double mu = md.mu;

    // ---- [A | B]: lane k < NDX owns column k of A, lane NDX+k owns column k of B ----
    SMPC_LANES(NT)
    if (lane < NDX + NU)
    {
      const bool isA = lane < NDX;
      const int k = isA ? lane : lane - NDX;
      const double lam_prev_v = b.lams[(ib + sprev) * NDX + (isA ? k : 0)]; // lambda_t (used at the end of the phase)
      // top block source column: Dtop[m] (m<6) = d(dx_q)[m]/d(col).  Branch-free: the three sources (d/dq, d/dv, d/du blocks) by a per-lane pointer
      // and stride, the lane-dependent extra terms by selects -- all LDS reads of the phase are issued unconditionally, side by side
      double Dtop[6], Dbot[6];
      const bool isAv = isA && k >= NV; // a velocity column of A
      const int kk = isAv ? k - NV : k;
      const double * const dsrc = isA ? (isAv ? sc.ab_dv() + kk : sc.ab_dq() + k) : sc.ab_du() + k;
      const int dstride = isA ? NV : NU;
#pragma unroll
      for (int m = 0; m < 6; m++)
      {
        Dbot[m] = dt * dsrc[m * dstride];
        Dtop[m] = dt * Dbot[m] + ((isAv && m == kk) ? dt : 0.0);
      }
      double acc = 0.0; // (A^T lam_next)[k] or (B^T lam_next)[k]
      double * dst = lq + (isA ? D::O_A : D::O_B);
      const int ld = isA ? NDX : NU;
      // only the 12 rows G = qb u vb are state dependent; rows qj / vj are the constant integrator pattern
      // (e_i + dt e_{v(i)} | dt^2 e_a and e_i | dt e_a), written once by lq_init_body
      // (the multiplier products are accumulated in row order, like a dense A^T lam)
      const double jq_on = (isA && k < 6) ? 1.0 : 0.0;
      const int jq_k = (isA && k < 6) ? k : 0;
#pragma unroll
      for (int gi = 0; gi < 12; gi++)
      {
        const int i = gi < 6 ? gi : NV + gi - 6;
        if (gi == 6)
        {
          // rows qj: lam_next[k] (A, k in qj), dt lam_next[k - NV] (A, k in vj), dt^2 lam_next[6 + k - 3 NF] (B, accelerations)
          const int li = isA ? kk : (k >= 3 * NF ? 6 + k - 3 * NF : 0);
          const double ls = isA ? ((k >= 6 && k < NV) ? 1.0 : (k >= NV + 6 ? dt : 0.0)) : (k >= 3 * NF ? dt * dt : 0.0);
          acc += ls * sc.lam_next[li];
        }
        double v;
        if (gi < 6)
        {
          // Je row i = [J3 Q; 0 J3]
          if (i < 3)
            v = sc.Je3[i * 3 + 0] * Dtop[0] + sc.Je3[i * 3 + 1] * Dtop[1] + sc.Je3[i * 3 + 2] * Dtop[2] + sc.JeQ[i * 3 + 0] * Dtop[3]
                + sc.JeQ[i * 3 + 1] * Dtop[4] + sc.JeQ[i * 3 + 2] * Dtop[5];
          else
            v = sc.Je3[(i - 3) * 3 + 0] * Dtop[3] + sc.Je3[(i - 3) * 3 + 1] * Dtop[4] + sc.Je3[(i - 3) * 3 + 2] * Dtop[5];
          v += jq_on * sc.Jq[i * 6 + jq_k];
        }
        else
          v = Dbot[i - NV] + ((isA && k == i) ? 1.0 : 0.0);
        dst[i * ld + k] = v;
        acc += v * sc.lam_next[i];
      }
      // rows vj
      if (isA)
        acc += k >= NV + 6 ? sc.lam_next[k] : 0.0;
      else
        acc += k >= 3 * NF ? dt * sc.lam_next[NV + 6 + k - 3 * NF] : 0.0;
      // Lagrangian gradient pieces: cost gradient + multipliers
      if (isA)
      {
        double g;
        if (k < 6)
        {
          g = 0.0;
          for (int a = 0; a < 6; a++)
            g += sc.Jl[a * 6 + k] * sc.Wrx[a];
        }
        else
          g = sc.Wrx[k];
        for (int a = 0; a < 6; a++)
          g += (k < NV ? sc.dh_dq[a * NV + k] : sc.Ag[a * NV + k - NV]) * sc.Whg[a];
        if (k < NV)
        {
          for (int a = 0; a < 3; a++)
            g += sc.dtgt[a * NV + k] * sc.Whd[3 + a];
          for (int fa = 0; fa < NF * 3; fa++)
            g += sc.Jfoot[fa * NV + k] * sc.Wrf[fa];
        }
        // C_x^T nu: box rows are unit selectors, the contact rows were contracted with nu when their columns were formed
        double cn = sc.cn[k];
        if (md.kinematics_limits && k >= 6 && k < NV)
          cn += sc.nu[k - 6];
        if (land != 0u && k < NV)
        { // C_x^T nu of the land rows (current multipliers)
          const double * lsc = kino_land_scratch<D, true>(sc);
          for (int f = 0; f < NF; f++)
            if ((land >> f) & 1u)
              cn += sc.Jfoot[(3 * f + 2) * NV + k] * lsc[2 * NF + f];
        }
        double q = g + acc + cn - (t > 0 ? lam_prev_v : 0.0);
        if (t == 0)
          q = 0.0; // x_0 is pinned (force_initial_condition_, reference src/mpc.cpp:53)
        lq[D::O_q + k] = q;
        lq[D::t_off(k, D::NXU)] = q; // (column NXU of the tile grid: the vector column of the structured sweep, loaded with its tiles)
        lq[D::O_lx + k] = g;
        lq[D::O_f + k] = mu * (sc.lamp[k] - sc.lam_next[k]);
        lq[D::O_lpd + k] = 2.0 * sc.lamp[k] - sc.lam_next[k];
        sc.rx[k] = fabs(q); // reuse as dual-infeasibility scratch
      }
      else
      {
        double g = sc.Wru[k];
        if (k < 3 * NF)
        {
          const int f = k / 3, j = k % 3;
          if ((in.mask >> f) & 1u)
          {
            const V3 rr = ld3(&sc.footp[f * 3]) - ld3(sc.com);
            const V3 xc = cross(rr, mk3(j == 0, j == 1, j == 2));
            g += sc.Whd[j] + xc.x * sc.Whd[3] + xc.y * sc.Whd[4] + xc.z * sc.Whd[5];
          }
        }
        double r = g + acc;
        if (cones && k < 3 * NF)
        { // C_u^T nu of the friction-cone rows (unmasked Jacobian, current multipliers)
          const int f = k / 3;
          const double * cs = kino_cone_scratch<D, true>(sc);
          double j0[3], j1[3];
          kino_cone_jac(0, sc.u[3 * f], sc.u[3 * f + 1], sc.u[3 * f + 2], b.cone_mu2, j0);
          kino_cone_jac(1, sc.u[3 * f], sc.u[3 * f + 1], sc.u[3 * f + 2], b.cone_mu2, j1);
          if ((in.mask >> f) & 1u)
            r += j0[k % 3] * cs[6 * NF + 2 * f] + j1[k % 3] * cs[6 * NF + 2 * f + 1];
        }
        lq[D::O_r + k] = r;
        lq[D::t_off(NDX + k, D::NXU)] = r;
        lq[D::O_lu + k] = g;
        sc.ru[k] = fabs(r);
      }
    }
    SMPC_LANES_END_WAVE

    if (in.prof) prof_tick(in.prof, 33, tprev);
    // ---- state-cost tables (after the [A|B] assembly: they take over the block that held d a_b / du);
    //      the weighted Gauss-Newton Jacobians are formed on the matrix cores below ----
    SMPC_LANES(NT)
    {
      const bool wdiag = md.w_diag != 0; // diagonal w_x, w_u: their entries come from the LDS model block
      for (int idx = lane; idx < NDX * 6; idx += NT)
      {
        const int a = idx / 6, k = idx % 6;
        double s = 0.0;
        if (wdiag)
          s = a < 6 ? md.wxd[a] * sc.Jl[a * 6 + k] : 0.0;
        else
          for (int bb = 0; bb < 6; bb++)
            s += mg.w_x[a * NDX + bb] * sc.Jl[bb * 6 + k];
        sc.WJl[idx] = s;
      }
      if (lane < 36)
      {
        // JWJ = Jl^T w_x[0:6,0:6] Jl (base block of the state Hessian)
        const int i = lane / 6, j = lane % 6;
        double s = 0.0;
        for (int a = 0; a < 6; a++)
        {
          double t = 0.0;
          if (wdiag)
            t = md.wxd[a] * sc.Jl[a * 6 + j];
          else
            for (int bb = 0; bb < 6; bb++)
              t += mg.w_x[a * NDX + bb] * sc.Jl[bb * 6 + j];
          s += sc.Jl[a * 6 + i] * t;
        }
        sc.JWJ[lane] = s;
      }
    }
    SMPC_LANES_END_WAVE
    if (in.prof) prof_tick(in.prof, 32, tprev);
    // ---- Q, S, R on the matrix cores:  H = [Q S; S^T R] = H_0 + J^T (W J)  with the stacked Gauss-Newton Jacobian
    //        rows  0.. 5  centroidal momentum      J = [dh_dq | Ag | 0]                    W J = WJc
    //        rows  8..13  momentum derivative      J = [[0; dtgt] | 0 | Ju (contact) | 0]  W J = [WD | 0 | WJu | 0]
    //        rows 16..27  foot positions           J = [Jfoot | 0 | 0]                     W J = WJf
    //      (K = 28 with two zero rows after each 6-row group, 7 K-steps of 4),  H_0 = state / control weight blocks
    //      + preg I.  Upper 16x16 tiles of the 64-padded (x, u) grid; tile column 3 (joint-acceleration columns of u)
    //      has no Jacobian entries and is written from the weights directly. ----
    {
      constexpr int T3I[6] = {0, 0, 0, 1, 1, 2}, T3J[6] = {0, 1, 2, 1, 2, 2};
      SMPC_ACC(qacc, NT, 6);
      SMPC_ACC(wjacc, NT, 6);
      SMPC_PLA(double, av, NT, 7 * 3);
      SMPC_PLA(double, bv, NT, 7 * 3);
      SMPC_PLA(double, wop, NT, 7);
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
        for (int tt = 0; tt < 6; tt++)
#pragma unroll
          for (int v = 0; v < 4; v++)
            SMPC_ACCV(wjacc, tt, v) = 0.0;
        // H_0 in accumulator layout: the weight entries come from global memory -- all loads are issued first
        // (one per accumulator entry, address-selected), the table look-ups and selects follow
        double wv[24];
        const bool wdiag = md.w_diag != 0;
        if (wdiag)
        {
#pragma unroll
          for (int tt = 0; tt < 6; tt++)
#pragma unroll
            for (int v = 0; v < 4; v++)
            {
              const int row = 16 * T3I[tt] + lr + 4 * v, col = 16 * T3J[tt] + lc;
              const double dv = row < NDX ? md.wxd[row] : md.wud[row < NDX + NU ? row - NDX : 0];
              wv[tt * 4 + v] = row == col ? dv : 0.0;
            }
        }
        else
        {
#pragma unroll
          for (int tt = 0; tt < 6; tt++)
#pragma unroll
            for (int v = 0; v < 4; v++)
            {
              const int row = 16 * T3I[tt] + lr + 4 * v, col = 16 * T3J[tt] + lc;
              const bool inx = row >= 6 && row < NDX && col >= 6 && col < NDX;
              const bool inu = row >= NDX && row < NDX + NU && col >= NDX && col < NDX + NU;
              const double * src = inx ? &mg.w_x[row * NDX + col] : (inu ? &mg.w_u[(row - NDX) * NU + col - NDX] : &mg.w_x[0]);
              wv[tt * 4 + v] = *src;
            }
        }
#pragma unroll
        for (int tt = 0; tt < 6; tt++)
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            const int row = 16 * T3I[tt] + lr + 4 * v, col = 16 * T3J[tt] + lc;
            const bool inx = row >= 6 && row < NDX && col >= 6 && col < NDX;
            const bool inu = row >= NDX && row < NDX + NU && col >= NDX && col < NDX + NU;
            double val = (inx || inu) ? wv[tt * 4 + v] : 0.0;
            if (row < NDX && col < NDX && (row < 6 || col < 6))
              val = row < 6 ? (col < 6 ? sc.JWJ[row * 6 + col] : sc.WJl[col * 6 + row]) : sc.WJl[row * 6 + col];
            SMPC_ACCV(qacc, tt, v) = val + ((row == col && row < NDX + NU) ? preg : 0.0);
          }
        // operands
#pragma unroll
        for (int ks = 0; ks < 7; ks++)
#pragma unroll
          for (int I = 0; I < 3; I++)
          {
            const int col = 16 * I + lc;
            double a = 0.0;
            if (ks < 2)
            {
              const int r = 4 * ks + lr;
              if (r < 6 && col < NDX)
              {
                a = col < NV ? sc.dh_dq[r * NV + col] : sc.Ag[r * NV + col - NV];
              }
            }
            else if (ks < 4)
            {
              const int r = 4 * (ks - 2) + lr;
              if (r < 6)
              {
                if (col < NV)
                {
                  a = r >= 3 ? sc.dtgt[(r - 3) * NV + col] : 0.0;
                }
                else if (col >= NDX && col < NDX + 3 * NF)
                {
                  const int k = col - NDX, f = k / 3, j = k % 3;
                  if ((in.mask >> f) & 1u)
                  {
                    // Ju[:, 3f+j] = [e_j ; (p_f - c) x e_j]
                    const V3 rr = ld3(&sc.footp[f * 3]) - ld3(sc.com);
                    const V3 xc = cross(rr, mk3(j == 0, j == 1, j == 2));
                    a = r < 3 ? (r == j ? 1.0 : 0.0) : (r == 3 ? xc.x : (r == 4 ? xc.y : xc.z));
                  }
                }
              }
            }
            else
            {
              const int r = 4 * (ks - 4) + lr;
              if (col < NV)
              {
                a = sc.Jfoot[r * NV + col];
              }
            }
            SMPC_PLV(av)[ks * 3 + I] = a;
          }
        // block-diagonal weight W~ = diag(w_cent, 0, w_centder, 0, w_frame x NF) as the A operand of W~ J~:
        // entry (16 R + lc, 4 ks + lr), tile row R = ks / 4
#pragma unroll
        for (int ks = 0; ks < 7; ks++)
        {
          const int r = 16 * (ks / 4) + lc, k = 4 * ks + lr;
          double w = 0.0;
          if (ks < 2)
            w = (r < 6 && k < 6) ? md.w_cent[r * 6 + k] : 0.0;
          else if (ks < 4)
            w = (r >= 8 && r < 14 && k < 14) ? md.w_centder[(r - 8) * 6 + k - 8] : 0.0;
          else
            w = (r < 16 + 3 * NF && (r - 16) / 3 == (k - 16) / 3) ? sc.wframe_()[((r - 16) % 3) * 3 + (k - 16) % 3] : 0.0;
          SMPC_PLV(wop)[ks] = w;
        }
      }
      SMPC_LANES_END_WAVE
      // W~ J~ (28 x 48) on the matrix cores; its accumulator layout (rows lr + 4 v of tile row R) is the B-operand
      // layout of K-step 4 R + v of the next product, so the weighted Jacobian never leaves the registers
#pragma unroll
      for (int ks = 0; ks < 7; ks++)
#pragma unroll
        for (int J = 0; J < 3; J++)
          if (ks < 4 || J < 2)
            SMPC_MFMA(wjacc, (ks / 4) * 3 + J, wop, ks, av, ks * 3 + J);
      SMPC_LANES(NT)
      {
#pragma unroll
        for (int ks = 0; ks < 7; ks++)
#pragma unroll
          for (int J = 0; J < 3; J++)
            SMPC_PLV(bv)[ks * 3 + J] = SMPC_ACCV(wjacc, (ks / 4) * 3 + J, ks % 4);
      }
      SMPC_LANES_END_WAVE
#pragma unroll
      for (int ks = 0; ks < 7; ks++)
#pragma unroll
        for (int tt = 0; tt < 6; tt++)
          if (ks < 4 || T3J[tt] < 2) // the foot rows only reach the q columns (tile columns 0, 1)
            SMPC_MFMA(qacc, tt, av, ks * 3 + T3I[tt], bv, ks * 3 + T3J[tt]);
      SMPC_LANES(NT)
      {
        const int lr = lane >> 4, lc = lane & 15;
#pragma unroll
        for (int tt = 0; tt < 6; tt++)
#pragma unroll
          for (int v = 0; v < 4; v++)
          {
            // the knot keeps [Q S; S^T R] as tiles in this very layout (Dims::O_T): off-diagonal tiles are stored as they are, of a diagonal
            // tile the upper triangle goes to its own place and to its mirror image (so that the tile is exactly symmetric for the sweep)
            const int I = T3I[tt], J = T3J[tt];
            const double val = SMPC_ACCV(qacc, tt, v);
            double * tile = lq + D::O_T + (I * D::NTT - I * (I - 1) / 2 + (J - I)) * 256;
            if (I != J)
              tile[v * 64 + lane] = val;
            else if (lr + 4 * v <= lc)
            {
              tile[v * 64 + lane] = val;
              tile[(lc >> 2) * 64 + (lc & 3) * 16 + lr + 4 * v] = val; // entry (lc, lr + 4 v) of the tile
            }
          }
        // columns / rows of the joint accelerations (u indices >= UC): no Jacobian entries.  Their S columns (zero)
        // and the off-diagonal R entries (weights) are constant and written once by lq_init_body; only the diagonal
        // carries the current primal regularisation
        constexpr int UC = 16 * 3 - NDX; // u columns covered by the tiles
        if (lane >= UC && lane < NU)
          lq[D::r_off(lane, lane)] = (md.w_diag ? md.wud[lane] : mg.w_u[lane * NU + lane]) + preg;
      }
      SMPC_LANES_END_WAVE
    }
    SMPC_LANES(NT)
    {
      // joint-box rows of C: unit selectors when active (the zeros of these rows are written once by lq_init_body, the
      // contact rows with the constraint Jacobian columns), d, vpd
      if (lane < NA)
        lq[D::O_C + lane * NDX + 6 + lane] = sc.act[lane] ? 1.0 : 0.0;
      if (lane < NC)
      {
        lq[D::O_d + lane] = mu * (sc.vplus[lane] - sc.nu[lane]);
        lq[D::O_vpd + lane] = sc.act[lane] ? 2.0 * sc.vplus[lane] - sc.nu[lane] : 0.0;
      }
    }
    SMPC_LANES_END_WAVE
    if (in.prof) prof_tick(in.prof, 35, tprev);
    // dual infeasibility = max |q_k|, |r_k| (sc.rx, sc.ru): a tree over the lanes, in place -- a loop of one lane over the 60 values is 5 % of the
    // kernel's instructions.  (max is exact and order-independent: the same number as the loop's.)
    static_assert(NU <= NDX && NDX % 4 == 0 && NDX / 4 >= 5, "tree of the dual-infeasibility maximum");
    SMPC_LANES(NT)
    if (lane < NU)
      sc.rx[lane] = fmax(sc.rx[lane], sc.ru[lane]);
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < NDX / 2)
      sc.rx[lane] = fmax(sc.rx[lane], sc.rx[lane + NDX / 2]);
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < NDX / 4)
      sc.rx[lane] = fmax(sc.rx[lane], sc.rx[lane + NDX / 4]);
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double dual = 0.0;
      for (int k = 0; k < NDX / 4; k++)
        dual = fmax(dual, sc.rx[k]);
      parts[0] = sc.red[0] + sc.red[1];
      parts[1] = sc.red[0];
      parts[2] = sc.red[2];
      parts[3] = dual;
    }
    SMPC_LANES_END_WAVE
  }

  template <class D, bool EXT>
  SMPC_DEV void deriv_one(const StageKernelArgs<D> & ka, int inst, int t)
  {
    typedef KinoScratch<D, true> KinoScratchT;
    constexpr int NT = 64;
    constexpr int NV = D::NV, NX = D::NX, NDX = D::NDX, NU = D::NU, NC = D::NC, NF = D::NF, NA = D::NA;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const bool term = t == H;
    const DevModel<D> & mg = *b.model; // global: large weights only
    SMPC_LDS(KinoScratchT, scs, 1);
    KinoScratchT & sc = scs[0];
    const DevModelSmall<D> & md = sc.ml; // LDS copy (filled in the load phase)
    const int st = ring_slot(ka.head, t, R);
    const size_t ib = (size_t)inst * R;
    const double * xg = b.xs + (ib + st) * NX;
    const double * xn_g = b.xs + (ib + ring_slot(ka.head, term ? t : t + 1, R)) * NX;
    const int sprev = ring_slot(ka.head, t > 0 ? t - 1 : 0, R);
    const double preg = b.scal[(size_t)inst * SC_N + SC_PREG];

    StageIn<D> in;
    in.md = &mg;
    in.terminal = term;
    in.mask = term ? 0u : b.stages[t].mask;
    in.u_ref = term ? nullptr : b.stages[t].u_ref;
    in.x_tgt = term ? mg.x_term : b.stages[t].x_tgt;
    in.foot_ref = term ? nullptr : b.foot_ref + ((size_t)inst * H + t) * NF * 3;
    in.C_rows = term ? nullptr : b.lq + ((size_t)inst * H + t) * D::LQ_STRIDE + D::O_C + (size_t)NA * NDX;
    long long tprev = SMPC_CLOCK();
    in.prof = (b.dbg != nullptr && inst == 0 && t == 17 && ka.slots == 0) ? b.dbg : nullptr; // optional phase timers: one mid-horizon block
    in.tprev = &tprev;

    // block inputs that are consumed once, by the lane that loaded them, stay in registers: lambda_t (q of the knot) and
    // x_{t+1} (committed to LDS only for the defect phase, into the then idle late block)
    SMPC_PL(double, xn_r, NT);
    SMPC_LANES(NT)
    {
      // all global loads of the block are issued back to back (index clamped, one wait), then committed to LDS
      static_assert(NX <= NT && NU <= NT && NDX <= NT && NC <= NT, "one element per lane");
      const double vx = xg[lane < NX ? lane : 0];
      const double vu = b.us[(ib + st) * NU + (lane < NU ? lane : 0)];
      const double vl = b.lams[(ib + st) * NDX + (lane < NDX ? lane : 0)];
      const double vn = b.vs[(ib + st) * NC + (lane < NC ? lane : 0)];
      // state_cost target: shared pose part, per-instance base-velocity part (address select, one load)
      const double vxt = *((!term && lane >= D::NQ && lane < D::NQ + 6) ? b.vref + (ib + st) * 6 + (lane - D::NQ) : in.x_tgt + (lane < NX ? lane : 0));
      const double vur = term ? 0.0 : in.u_ref[lane < NU ? lane : 0];
      const double vfr = term ? 0.0 : in.foot_ref[lane < NF * 3 ? lane : 0];
      static_assert(NX + 9 <= NT && D::NV >= 9, "w_frame rides in the spare lanes of the x_{t+1} register");
      const double vxn = *(lane < NX ? xn_g + lane : mg.w_frame + (lane < NX + 9 ? lane - NX : 0));
      lanes_load_model<D, NT>(sc, &mg, lane);
      if (lane < NX)
      {
        sc.x[lane] = vx;
        sc.in_x_tgt[lane] = vxt;
      }
      SMPC_PLV(xn_r) = vxn;
      if (lane < NU)
        sc.in_u_ref[lane] = vur;
      if (lane < NF * 3)
        sc.in_foot_ref[lane] = vfr;
      if (lane < NU)
        sc.u[lane] = term ? 0.0 : vu;
      if (lane < NDX)
      {
        sc.lam_next[lane] = term ? 0.0 : vl;
      }
      if (lane < NC)
        sc.nu[lane] = term ? 0.0 : vn;
    }
    SMPC_LANES_END_WAVE

    if (in.prof) prof_tick(in.prof, 15, tprev);
    kino_tree_phases<D, true>(sc, in);

    if (!term)
    {
      static_assert(KinoScratchEval<D>::LATE_DOUBLES >= NX, "x_{t+1} is staged at the start of the late block");
      SMPC_LANES(NT)
      {
        if (lane < NX)
          sc.cval[lane] = SMPC_PLV(xn_r); // (cval | Wrx ...: idle until the cost phase)
        else if (lane < NX + 9)
          sc.wframe_()[lane - NX] = SMPC_PLV(xn_r); // w_frame into the (now dead) acceleration vector
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      lanes_difference<D>(sc.cval, sc.xnext, sc.e, lane, 61);
      SMPC_LANES_END_WAVE
    }
    if (in.prof) prof_tick(in.prof, 29, tprev);
    kino_cost_constraints<D, true>(sc, in);
    if (in.prof) prof_tick(in.prof, 30, tprev);

    double * parts = b.parts0 + ((size_t)inst * (H + 1) + t) * 4;

    if (term)
    {
      deriv_terminal_node<D, EXT>(ka, sc, inst, t, preg, parts);
      return;
    }

    // ---- multipliers, active set ----
    kino_multipliers<D, true>(sc, in, b.lams_e + (ib + st) * NDX, b.vs_e + (ib + st) * NC);
    const bool cones = EXT && b.es != nullptr;
    if (cones)
      kino_cone_rows<D, true>(sc, in, b.cone_mu2, b.es + (ib + st) * 2 * NF, (const double *)nullptr, 0.0, b.es_e + (ib + st) * 2 * NF,
                              b.ek + ((size_t)inst * H + t) * 12 * NF);
    const unsigned land = (EXT && b.ls != nullptr) ? (b.stages[t].land & in.mask) : 0u;
    if (EXT && b.ls != nullptr)
      kino_land_rows<D, true>(sc, in, land, b.land_z, b.ls + (ib + st) * NF, (const double *)nullptr, 0.0, b.ls_e + (ib + st) * NF,
                              b.lk + ((size_t)inst * H + t) * NF * (NV + 2));
    if (in.prof) prof_tick(in.prof, 31, tprev);

    deriv_stage_knot<D, EXT>(ka, sc, in, inst, t, preg, parts, cones, land, tprev);
  }

  // =============================================================================================
  // frontend_body: grid = B, 64 lanes.  State feedback front-end: what
  // RobotDataHandler::updateInternalData(x, false) + getCentroidalState provide for a measured multibody state
  // (reference src/robot-handler.cpp:106-149): foot positions, centre of mass, centroidal momentum, and the
  // centroidal state [com; h_lin; h_ang].
  // =============================================================================================
  template <class D>
  struct FrontendArgs
  {
    Buffers<D> b;
    const double * X;                     // [B][NX] measured states (device)
    double *feet, *com, *hg, *cstate;     // [B][NF*3], [B][3], [B][6], [B][9] (device), any may be null
  };
  template <class D>
  SMPC_DEV void frontend_body(const FrontendArgs<D> & ka, int block)
  {
    typedef KinoScratch<D, false> KinoScratchT;
    constexpr int NT = 64;
    constexpr int NX = D::NX, NU = D::NU, NF = D::NF;
    const Buffers<D> & b = ka.b;
    const int inst = block;
    const DevModel<D> & mg = *b.model;
    SMPC_LDS(KinoScratchT, scs, 1);
    KinoScratchT & sc = scs[0];
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
    kino_tree_phases<D, false, true>(sc, in);
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

  // =============================================================================================
  // interp_body: grid = B, 64 lanes.  Targets between MPC knots for the whole-body controller that follows the MPC
  // (reference src/interpolator.cpp:5-78 used as in examples/go2_kinodynamics.py:276-284):
  //   x    interpolateState over xs[0 .. knots-1]: q on the manifold (q0 (+) s (q1 (-) q0)), v linear
  //   acc  interpolateLinear over [getStateDerivative(t)[nv:] with the joint part replaced by us[t][3 nf:]], t = 0, 1
  //   f    interpolateLinear over us[t][: 3 nf], t = 0, 1
  // =============================================================================================
  template <class D>
  struct InterpArgs
  {
    Buffers<D> b;
    int head, knots;
    double delay, timestep;
    double *x_out, *acc_out, *f_out; // device, any may be null
    double * u_out = nullptr;        // [B][NU] interpolateLinear over us[0], us[1] (device, may be null)
  };
  template <class D>
  SMPC_DEV void interp_body(const InterpArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NX = D::NX, NDX = D::NDX, NV = D::NV, NU = D::NU, NF = D::NF;
    const Buffers<D> & b = ka.b;
    const int inst = block, R = b.R;
    const size_t step = (size_t)(ka.delay / ka.timestep);
    const double s = (ka.delay - (double)step * ka.timestep) / ka.timestep;
    SMPC_LDS(double, e, D::NDX);
    if (ka.x_out != nullptr)
    {
      double * xo = ka.x_out + (size_t)inst * NX;
      if (step >= (size_t)ka.knots - 1)
      {
        const double * xl = b.xs + ((size_t)inst * R + ring_slot(ka.head, ka.knots - 1, R)) * NX;
        SMPC_LANES(NT)
        if (lane < NX)
          xo[lane] = xl[lane];
        SMPC_LANES_END_WAVE
      }
      else
      {
        const double * x0 = b.xs + ((size_t)inst * R + ring_slot(ka.head, (int)step, R)) * NX;
        const double * x1 = b.xs + ((size_t)inst * R + ring_slot(ka.head, (int)step + 1, R)) * NX;
        SMPC_LANES(NT)
        lanes_difference<D>(x0, x1, e, lane, 0);
        SMPC_LANES_END_WAVE
        SMPC_LANES(NT)
        lanes_integrate<D>(x0, e, s, xo, lane, 0);
        SMPC_LANES_END_WAVE
      }
    }
    static_assert(NDX <= NT && NU <= NT, "one entry per lane");
    // two knots (t = 0, 1) for the accelerations and the forces
    const bool last = step >= 1;
    const double w1 = last ? 1.0 : s, w0 = last ? 0.0 : 1.0 - s;
    const double * u0 = b.us + ((size_t)inst * R + ring_slot(ka.head, 0, R)) * NU;
    const double * u1 = b.us + ((size_t)inst * R + ring_slot(ka.head, 1, R)) * NU;
    SMPC_LANES(NT)
    {
      if (ka.acc_out != nullptr && lane < NV)
      {
        const double * xd = b.xdot01 + (size_t)inst * 4 * NV;
        const double a0 = lane < 6 ? xd[NV + lane] : u0[3 * NF + lane - 6];
        const double a1 = lane < 6 ? xd[2 * NV + NV + lane] : u1[3 * NF + lane - 6];
        ka.acc_out[(size_t)inst * NV + lane] = a1 * w1 + a0 * w0;
      }
      if (ka.f_out != nullptr && lane < 3 * NF)
        ka.f_out[(size_t)inst * 3 * NF + lane] = u1[lane] * w1 + u0[lane] * w0;
      if (ka.u_out != nullptr && lane < NU)
        ka.u_out[(size_t)inst * NU + lane] = u1[lane] * w1 + u0[lane] * w0;
    }
    SMPC_LANES_END_WAVE
  }

  // Riccati feedback application (reference examples/go2_fulldynamics.py:283-285):
  //   u = u_interp - K_0 (x_interp (-) x_meas),   x_interp / u_interp from interp_body, K_0 the first-stage feedback gain
  template <class D>
  struct FeedbackArgs
  {
    Buffers<D> b;
    const double *X_meas, *x_interp, *u_interp; // [B][NX], [B][NX], [B][NU] (device)
    const double * K0;                          // [B][NU][NDX] (device)
    double * u_out;                             // [B][NU] (device)
  };
  template <class D>
  SMPC_DEV void feedback_body(const FeedbackArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NX = D::NX, NDX = D::NDX, NU = D::NU;
    const int inst = block;
    SMPC_LDS(double, e, D::NDX);
    SMPC_LANES(NT)
    lanes_difference<D>(ka.X_meas + (size_t)inst * NX, ka.x_interp + (size_t)inst * NX, e, lane, 0); // x_interp (-) x_meas
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < NU)
    {
      const double * Kr = ka.K0 + ((size_t)inst * NU + lane) * NDX;
      double acc = ka.u_interp[(size_t)inst * NU + lane];
      for (int j = 0; j < NDX; j++)
        acc -= Kr[j] * e[j];
      ka.u_out[(size_t)inst * NU + lane] = acc;
    }
    SMPC_LANES_END_WAVE
  }

  // interpolation of explicit knot lists (the reference's Interpolator class on host data): one block
  //   kind 0: state knots [n][NX], 1: configuration knots [n][NQ], 2: linear [n][dim]
  template <class D>
  struct InterpKnotsArgs
  {
    int kind, n, dim;
    double delay, timestep;
    const double * knots; // device
    double * out;         // device
  };
  template <class D>
  SMPC_DEV void interp_knots_body(const InterpKnotsArgs<D> & ka, int)
  {
    constexpr int NT = 64;
    constexpr int NX = D::NX, NQ = D::NQ, NV = D::NV;
    const size_t step = (size_t)(ka.delay / ka.timestep);
    const double s = (ka.delay - (double)step * ka.timestep) / ka.timestep;
    SMPC_LDS(double, buf, 3 * D::NX + D::NDX);
    double *x0 = buf, *x1 = buf + NX, *xo = buf + 2 * NX, *e = buf + 3 * NX;
    if (step >= (size_t)ka.n - 1)
    {
      SMPC_LANES(NT)
      for (int i = lane; i < ka.dim; i += NT)
        ka.out[i] = ka.knots[(size_t)(ka.n - 1) * ka.dim + i];
      SMPC_LANES_END_WAVE
      return;
    }
    const double * k0 = ka.knots + step * ka.dim;
    const double * k1 = ka.knots + (step + 1) * ka.dim;
    if (ka.kind == 2)
    {
      SMPC_LANES(NT)
      for (int i = lane; i < ka.dim; i += NT)
        ka.out[i] = k1[i] * s + k0[i] * (1.0 - s);
      SMPC_LANES_END_WAVE
      return;
    }
    SMPC_LANES(NT)
    if (lane < NX)
    {
      x0[lane] = lane < ka.dim ? k0[lane] : 0.0;
      x1[lane] = lane < ka.dim ? k1[lane] : 0.0;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    lanes_difference<D>(x0, x1, e, lane, 0);
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    lanes_integrate<D>(x0, e, s, xo, lane, 0);
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < ka.dim)
      ka.out[lane] = (lane < NQ) ? xo[lane] : k1[lane] * s + k0[lane] * (1.0 - s);
    SMPC_LANES_END_WAVE
    (void)NV;
  }

  // =============================================================================================
  // lq_init_body: grid = B * H, run once: the state-independent rows (qj, vj) of A and B of every knot
  // (semi-implicit Euler: q+ = q + dt (v + dt a), v+ = v + dt a, reference src/kinodynamics.cpp:88)
  // =============================================================================================
  template <class D>
  SMPC_DEV void lq_init_body(const StageKernelArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NV = D::NV, NDX = D::NDX, NU = D::NU, NF = D::NF;
    static_assert(NDX + NU <= NT, "one lane per column");
    const double dt = ka.b.model->dt;
    double * lq = ka.b.lq + (size_t)block * D::LQ_STRIDE;
    SMPC_LANES(NT)
    if (lane < NDX + NU)
    {
      const bool isA = lane < NDX;
      const int k = isA ? lane : lane - NDX;
      for (int i = 6; i < NDX; i++)
      {
        if (i >= NV && i < NV + 6)
          continue;
        double v;
        if (i < NV)
          v = isA ? (k == i ? 1.0 : 0.0) + (k == NV + i ? dt : 0.0) : ((k == 3 * NF + i - 6) ? dt * dt : 0.0);
        else
          v = isA ? ((k == i) ? 1.0 : 0.0) : ((k == 3 * NF + i - NV - 6) ? dt : 0.0);
        lq[(isA ? D::O_A + i * NDX : D::O_B + i * NU) + k] = v;
      }
    }
    SMPC_LANES_END_WAVE
    // blocks without Jacobian entries: S columns and off-diagonal R entries of the joint accelerations, the zeros of the
    // joint-box rows of C
    SMPC_LANES(NT)
    {
      // (first the whole tile block: the padding entries beyond the problem are loaded by the sweep and must be finite)
      for (int idx = lane; idx < D::N_T; idx += NT)
        lq[D::O_T + idx] = 0.0;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      constexpr int UC = 16 * 3 - NDX, NA = D::NA;
      const DevModel<D> & mg = *ka.b.model;
      for (int idx = lane; idx < NU * NU; idx += NT)
      {
        const int i = idx / NU, j = idx % NU;
        if ((i >= UC || j >= UC) && i != j)
        { // off-diagonal weights of the joint accelerations; inside a diagonal tile both triangles are kept
          if (i < j)
            lq[D::r_off(i, j)] = mg.w_u[idx];
          else if ((NDX + i) / 16 == (NDX + j) / 16)
            lq[D::O_T + ((((NDX + j) / 16) * D::NTT - ((NDX + j) / 16) * ((NDX + j) / 16 - 1) / 2) * 4 + ((NDX + i) % 16) / 4) * 64 + (((NDX + i) % 16) % 4) * 16 + (NDX + j) % 16] = mg.w_u[idx];
        }
      }
      for (int idx = lane; idx < NA * NDX; idx += NT)
        lq[D::O_C + idx] = 0.0;
    }
    SMPC_LANES_END_WAVE
  }

  // =============================================================================================
  // trial_body: grid = B * (H+1); block (inst, t) evaluates the candidates j0 .. j0+nj-1 one after the other.
  // The backtracking launch (slots > 0) has grid = slots * (H+1) and walks the compacted list of instances that
  // rejected alpha = 1: the common case (everybody accepted) costs a few thousand empty blocks, not B * (H+1).
  // =============================================================================================
  template <class D, bool EXT = false>
  SMPC_DEV void trial_one(const StageKernelArgs<D> & ka, int inst, int t, int j);

  // EXT: the problem has optional constraint blocks (friction cones, land rows, terminal constraint); the default instantiation
  // carries none of their code
  template <class D, bool EXT = false>
  SMPC_DEV void trial_body(const StageKernelArgs<D> & ka, int block)
  {
    const int H = ka.b.H;
    const int slot = block / (H + 1), t = block % (H + 1);
    // slots > 0: backtracking launch over the compacted list of instances that rejected alpha = 1, `slots` at a time;
    // slots == 0: block per (instance, stage).  One loop (one inlined copy of the stage evaluation) serves both.
    const int count = ka.slots > 0 ? ka.b.und_list[ka.b.B] : slot + 1;
    const int stride = ka.slots > 0 ? ka.slots : ka.b.B;
    for (int m = slot; m < count; m += stride)
    {
      const int inst = ka.slots > 0 ? ka.b.und_list[m] : m;
      if (ka.slots == 0 && ka.b.ls_sel[inst] >= 0)
        break; // already accepted an earlier candidate (uniform across the workgroup)
      for (int jj = 0; jj < ka.nj; jj++)
        trial_one<D, EXT>(ka, inst, t, ka.j0 + jj);
    }
  }

  template <class D, bool EXT>
  SMPC_DEV void trial_one(const StageKernelArgs<D> & ka, int inst, int t, int j)
  {
    typedef KinoScratch<D, false> KinoScratchT;
    constexpr int NT = 64;
    constexpr int NV = D::NV, NX = D::NX, NDX = D::NDX, NU = D::NU, NC = D::NC, NF = D::NF;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const bool term = t == H;
    const DevModel<D> & mg = *b.model; // global: large weights only
    SMPC_LDS(KinoScratchT, scs, 1);
    KinoScratchT & sc = scs[0];
    const DevModelSmall<D> & md = sc.ml; // LDS copy (filled in the load phase)
    const int st = ring_slot(ka.head, t, R);
    const size_t ib = (size_t)inst * R;
    double alpha = 1.0;
    for (int i = 0; i < j; i++)
      alpha *= 0.5;

    StageIn<D> in;
    in.md = &mg;
    in.terminal = term;
    in.mask = term ? 0u : b.stages[t].mask;
    in.u_ref = term ? nullptr : b.stages[t].u_ref;
    in.x_tgt = term ? mg.x_term : b.stages[t].x_tgt;
    in.foot_ref = term ? nullptr : b.foot_ref + ((size_t)inst * H + t) * NF * 3;

    const double * dx = b.dxs + ((size_t)inst * (H + 1) + t) * NDX;
    SMPC_LANES(NT)
    {
      // all global loads of the block back to back (index clamped), then committed to LDS: raw x_t, x_{t+1} and their
      // steps (the manifold update follows), u, lam, nu at the trial point, the stage inputs
      static_assert(NX <= NT && NU <= NT && NDX <= NT && NC <= NT, "one element per lane");
      const int sn = ring_slot(ka.head, term ? t : t + 1, R);
      const size_t lt = (size_t)inst * H + (term ? 0 : t);
      const double vx = b.xs[(ib + st) * NX + (lane < NX ? lane : 0)];
      const double vxn = b.xs[(ib + sn) * NX + (lane < NX ? lane : 0)];
      const double vdx = dx[lane < NDX ? lane : 0];
      const double vdxn = dx[(term ? 0 : NDX) + (lane < NDX ? lane : 0)];
      const double vu = b.us[(ib + st) * NU + (lane < NU ? lane : 0)], vdu = b.dus[lt * NU + (lane < NU ? lane : 0)];
      const double vl = b.lams[(ib + st) * NDX + (lane < NDX ? lane : 0)], vdl = b.dlams[lt * NDX + (lane < NDX ? lane : 0)];
      const double vn = b.vs[(ib + st) * NC + (lane < NC ? lane : 0)], vdn = b.dvs[lt * NC + (lane < NC ? lane : 0)];
      // state_cost target: shared pose part, per-instance base-velocity part (address select, one load)
      const double vxt = *((!term && lane >= D::NQ && lane < D::NQ + 6) ? b.vref + (ib + st) * 6 + (lane - D::NQ) : in.x_tgt + (lane < NX ? lane : 0));
      const double vur = term ? 0.0 : in.u_ref[lane < NU ? lane : 0];
      const double vfr = term ? 0.0 : in.foot_ref[lane < NF * 3 ? lane : 0];
      const double vwf = mg.w_frame[lane < 9 ? lane : 0];
      lanes_load_model<D, NT>(sc, &mg, lane);
      if (lane < NX)
      {
        sc.x[lane] = vx;
        sc.xn1[lane] = vxn;
        sc.in_x_tgt[lane] = vxt;
      }
      if (lane < NDX)
      {
        sc.e[lane] = vdx;   // (temporaries: e and rx are computed later)
        sc.rx[lane] = vdxn;
        sc.lam_next[lane] = term ? 0.0 : vl + alpha * vdl;
      }
      if (lane < NU)
      {
        sc.u[lane] = term ? 0.0 : vu + alpha * vdu;
        sc.in_u_ref[lane] = vur;
      }
      if (lane < NC)
        sc.nu[lane] = term ? 0.0 : vn + alpha * vdn;
      if (lane < NF * 3)
        sc.in_foot_ref[lane] = vfr;
      if (lane < 9)
        sc.wframe_()[lane] = vwf;
    }
    SMPC_LANES_END_WAVE
    // x_t (+) alpha dx_t and x_{t+1} (+) alpha dx_{t+1} in place: the two SE(3) updates run side by side on lanes 0 / 1
    SMPC_LANES(NT)
    {
      if (lane < 2)
      {
        double * x = lane == 0 ? sc.x : sc.xn1;
        const double * d = lane == 0 ? sc.e : sc.rx;
        const V3 dv = alpha * ld3(d), dw = alpha * ld3(d + 3);
        const Quat q0{x[3], x[4], x[5], x[6]};
        const SE3 E = exp6(dv, dw);
        st3(x, ld3(x) + quat_to_R(q0) * E.p);
        Quat qn = quat_mul(q0, quat_exp(dw));
        const double n = 1.0 / sqrt(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
        x[3] = qn.x * n;
        x[4] = qn.y * n;
        x[5] = qn.z * n;
        x[6] = qn.w * n;
      }
      else if (lane >= 6 && lane < 2 * NV)
      {
        // linear entries: tangent index i <-> x[i + 1] (joints) or x[NQ + i - NV] (velocities)
        const int i = lane, o = i < NV ? i + 1 : D::NQ + i - NV;
        sc.x[o] += alpha * sc.e[i];
        sc.xn1[o] += alpha * sc.rx[i];
      }
    }
    SMPC_LANES_END_WAVE

    kino_tree_phases<D, false>(sc, in);
    if (!term)
    {
      SMPC_LANES(NT)
      lanes_difference<D>(sc.xn1, sc.xnext, sc.e, lane, 61);
      SMPC_LANES_END_WAVE
    }
    kino_cost_constraints<D, false>(sc, in);
    double * parts = b.partsT + (((size_t)inst * D::LS_N + j) * (H + 1) + t) * 2;
    if (term)
    {
      SMPC_LANES(NT)
      if (lane == 0)
      {
        double pen = 0.0, prim = 0.0;
        if (EXT && b.CN != nullptr)
          for (int r = 0; r < 3; r++)
          { // terminal constraint at the trial point, multipliers v + alpha dv
            const double c = sc.com[r] + b.dcm_tau * sc.hg[r] / md.total_mass - b.dcm_ref[(size_t)inst * 3 + r];
            const double vp = b.vN_e[(size_t)inst * 3 + r] + c / md.mu;
            const double dv = vp - (b.vN[(size_t)inst * 3 + r] + alpha * b.dvN[(size_t)inst * 3 + r]);
            pen += 0.5 * md.mu * (vp * vp + dv * dv);
            prim = fmax(prim, fabs(c));
          }
        parts[0] = sc.red[0] + pen;
        parts[1] = prim;
      }
      SMPC_LANES_END_WAVE
      return;
    }
    kino_multipliers<D, false>(sc, in, b.lams_e + (ib + st) * NDX, b.vs_e + (ib + st) * NC);
    if (EXT && b.es != nullptr)
      kino_cone_rows<D, false>(sc, in, b.cone_mu2, b.es + (ib + st) * 2 * NF, b.des + ((size_t)inst * H + t) * 2 * NF, alpha,
                               b.es_e + (ib + st) * 2 * NF, (double *)nullptr);
    if (EXT && b.ls != nullptr)
      kino_land_rows<D, false>(sc, in, b.stages[t].land, b.land_z, b.ls + (ib + st) * NF, b.dls + ((size_t)inst * H + t) * NF, alpha,
                               b.ls_e + (ib + st) * NF, (double *)nullptr);
    SMPC_LANES(NT)
    {
      if (lane == 0)
      {
        parts[0] = sc.red[0] + sc.red[1];
        parts[1] = sc.red[2];
      }
      if (t < 2 && lane < NV)
      {
        double * xd = b.xdotT + (((size_t)inst * D::LS_N + j) * 2 + t) * 2 * NV;
        xd[lane] = sc.x[D::NQ + lane];
        xd[NV + lane] = sc.a[lane];
      }
    }
    SMPC_LANES_END_WAVE
  }
} // namespace smpc
