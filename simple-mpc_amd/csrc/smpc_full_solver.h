// smpc_full_solver.h -- proximal Riccati backward / forward sweeps for stages with DENSE A, B (full-dynamics OCP): the same
// recursion as riccati_body / forward_body (smpc_solver_kernels.h; reference gar::ProximalRiccatiSolver, src/mpc.cpp:52;
// SURVEY App. B.5), with the constraint rows of the full-dynamics stage:
//   torque box rows  (unit selectors on u):  R^_ii += act_i / mu,  r^_i += act_i d_i / mu,   dnu_i = (act_i du_i + d_i) / mu
//   joint box rows   (unit selectors on x):  P_t(6+i, 6+i) += act_i / mu,  p_t(6+i) += act_i d_i / mu,  dnu = (act dx_{6+i} + d) / mu
// (eliminating a multiplier row whose Jacobian is a unit selector adds a diagonal term: no loss of accuracy, unlike the dense
// cone rows of the centroidal kernel).  This file holds the model-independent VALU version (cross-check, any size); the
// matrix-core version is riccati_dense_body in smpc_riccati_dense.h.
#pragma once
#include "smpc_full_model.h"
#include "smpc_solver_kernels.h"

namespace smpc
{
  template <class D>
  struct RiccatiFullLds
  {
    static constexpr int NDX = D::NDX, NU = D::NU, NC = D::NC, NXU = D::NDX + D::NU;
    double P[NDX * NDX];
    double MT[NDX * NXU];
    double AB[NDX * NXU];
    double QS[NDX * NXU];
    double Rh[NU * NU];
    double p[NDX], pt[NDX], qh[NDX], rh[NU], f[NDX], d[NC], act[D::NU + D::NA], col[NDX], wr[NU], kk[NU];
  };

  template <class D, int NT>
  SMPC_DEV void riccati_full_body(const SolverArgs<D> & ka, int block)
  {
    static_assert(NT >= 64, "needs one lane per right-hand side column");
    constexpr int NDX = D::NDX, NU = D::NU, NC = D::NC, NXU = NDX + NU, NA = D::NA;
    constexpr int TL = (NDX % 3 == 0 && NXU % 3 == 0 && NU % 3 == 0) ? 3 : 2;
    static_assert(NDX % TL == 0 && NXU % TL == 0 && NU % TL == 0, "register tiles must divide the dimensions");
    static_assert(NDX + 1 <= NT, "one lane per right-hand side column");
    static_assert(D::NCD == 0, "dense rows (cones, landing feet) are handled by the matrix-core sweep only");
    const Buffers<D> & b = ka.b;
    const int H = b.H;
    const int inst = block;
    const double mu = b.model->mu, imu = 1.0 / mu;
    SMPC_LDS(RiccatiFullLds<D>, lds, 1);
    RiccatiFullLds<D> & s = lds[0];
    SMPC_LANES(NT)
    {
      for (int i = lane; i < NDX * NDX; i += NT)
        s.P[i] = b.QN[(size_t)inst * NDX * NDX + i];
      for (int i = lane; i < NDX; i += NT)
        s.p[i] = b.qN[(size_t)inst * NDX + i];
    }
    SMPC_LANES_END
    for (int t = H - 1; t >= 0; t--)
    {
      const double * lq = b.lq + ((size_t)inst * H + t) * D::LQ_STRIDE;
      double * g = b.gains + ((size_t)inst * H + t) * D::G_STRIDE;
      SMPC_LANES(NT)
      {
        for (int idx = lane; idx < NDX * NDX; idx += NT)
        {
          const int i = idx / NDX, j = idx % NDX;
          s.AB[i * NXU + j] = lq[D::O_A + idx];
          s.MT[i * NDX + j] = mu * s.P[idx] + (i == j ? 1.0 : 0.0);
        }
        for (int idx = lane; idx < NDX * NU; idx += NT)
        {
          const int i = idx / NU, j = idx % NU;
          s.AB[i * NXU + NDX + j] = lq[D::O_B + idx];
        }
        for (int i = lane; i < NDX; i += NT)
        {
          s.f[i] = lq[D::O_f + i];
          g[D::G_pn + i] = s.p[i];
        }
        for (int i = lane; i < NC; i += NT)
          s.d[i] = lq[D::O_d + i];
        for (int i = lane; i < NU + NA; i += NT)
          s.act[i] = lq[D::O_act + i];
      }
      SMPC_LANES_END
      SMPC_LANES(NT)
      if (lane < NDX)
      {
        double acc = s.p[lane];
        for (int j = 0; j < NDX; j++)
          acc += s.P[lane * NDX + j] * s.f[j];
        s.pt[lane] = acc;
      }
      SMPC_LANES_END
      wg_cholesky<NDX, NT>(s.MT, NDX, s.col);
      SMPC_LANES(NT)
      lane_chol_solve<NDX, NT>(
        s.MT, NDX, NDX + 1, lane, [&](int i, int c) { return c < NDX ? s.P[i * NDX + c] : s.pt[i]; }, [&](int, int, double) {},
        [&](int i, int c, double v) {
          if (c < NDX)
            s.P[i * NDX + c] = v;
          else
            s.pt[i] = v;
        });
      SMPC_LANES_END
      SMPC_LANES(NT)
      for (int idx = lane; idx < NDX * NDX; idx += NT)
      {
        const int i = idx / NDX, j = idx % NDX;
        if (j < i)
        {
          const double v = 0.5 * (s.P[i * NDX + j] + s.P[j * NDX + i]);
          s.P[i * NDX + j] = v;
          s.P[j * NDX + i] = v;
        }
      }
      SMPC_LANES_END
      SMPC_LANES(NT)
      {
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
        mm_tn<NDX, NXU, NDX, TL, TL, NT>(
          s.P, NDX, s.AB, NXU, lane, [](int, int) { return 0.0; }, [&](int i, int j, double v) { s.MT[i * NXU + j] = v; });
      }
      SMPC_LANES_END
      SMPC_LANES(NT)
      {
        mm_tn<NDX, NXU, NDX, TL, TL, NT>(
          s.AB, NXU, s.MT, NXU, lane, [&](int i, int j) { return j < NDX ? lq[D::O_Q + (i < j ? i : j) * NDX + (i < j ? j : i)] : lq[D::O_S + i * NU + j - NDX]; }, // upper triangle is authoritative
          [&](int i, int j, double v) { s.QS[i * NXU + j] = v; });
        mm_tn<NU, NU, NDX, TL, TL, NT>(
          s.AB + NDX, NXU, s.MT + NDX, NXU, lane,
          [&](int i, int j) { return lq[D::O_R + (i < j ? i : j) * NU + (i < j ? j : i)] + (i == j ? imu * s.act[i] : 0.0); }, // torque box rows
          [&](int i, int j, double v) { s.Rh[i * NU + j] = v; });
        for (int c = lane; c < NXU; c += NT)
        {
          double acc = c < NDX ? lq[D::O_q + c] : lq[D::O_r + c - NDX] + imu * s.act[c - NDX] * s.d[c - NDX];
          for (int k = 0; k < NDX; k++)
            acc += s.AB[k * NXU + c] * s.pt[k];
          if (c < NDX)
            s.qh[c] = acc;
          else
            s.rh[c - NDX] = acc;
        }
      }
      SMPC_LANES_END
      wg_cholesky<NU, NT>(s.Rh, NU, s.col);
      SMPC_LANES(NT)
      lane_chol_solve<NU, NT>(
        s.Rh, NU, NDX + 1, lane, [&](int i, int c) { return c < NDX ? s.QS[c * NXU + NDX + i] : s.rh[i]; },
        [&](int i, int c, double v) { s.MT[i * (NDX + 1) + c] = v; },
        [&](int i, int c, double v) { g[D::G_K + i * (NDX + 1) + c] = -v; });
      SMPC_LANES_END
      // P_t = Q^ - W^T W + box ;  p_t = q^ - W^T w_r + box
      SMPC_LANES(NT)
      {
        mm_tn<NDX, NDX, NU, TL, TL, NT>(
          s.MT, NDX + 1, s.MT, NDX + 1, lane, [](int, int) { return 0.0; },
          [&](int i, int j, double v) {
            const bool bx = i == j && i >= 6 && i < 6 + NA;
            s.P[i * NDX + j] = s.QS[i * NXU + j] - v + (bx ? imu * s.act[NU + i - 6] : 0.0);
          });
        if (lane >= NT - 64 && lane < NT - 64 + NDX)
        {
          const int i = lane - (NT - 64);
          double acc = s.qh[i];
          for (int m = 0; m < NU; m++)
            acc -= s.MT[m * (NDX + 1) + i] * s.MT[m * (NDX + 1) + NDX];
          if (i >= 6 && i < 6 + NA)
            acc += imu * s.act[NU + i - 6] * s.d[NU + i - 6];
          s.p[i] = acc;
        }
      }
      SMPC_LANES_END
      static_assert(NDX <= 64, "one lane per row in the p_t phase");
      SMPC_LANES(NT)
      for (int idx = lane; idx < NDX * NDX; idx += NT)
      {
        const int i = idx / NDX, j = idx % NDX;
        if (j < i)
        {
          const double v = 0.5 * (s.P[i * NDX + j] + s.P[j * NDX + i]);
          s.P[i * NDX + j] = v;
          s.P[j * NDX + i] = v;
        }
      }
      SMPC_LANES_END
    }
  }

  // forward sweep + merit directional derivative, one wavefront per instance (dense A, B, P~ from the gains block)
  template <class D>
  SMPC_DEV void forward_full_body(const SolverArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NDX = D::NDX, NU = D::NU, NC = D::NC, NA = D::NA;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const int inst = block;
    const double mu = b.model->mu, dt = b.model->dt;
    constexpr int NV = D::NV;
    SMPC_LDS(double, dx, D::NDX);
    SMPC_LDS(double, du, D::NU);
    SMPC_LDS(double, y, D::NDX);
    SMPC_LDS(double, ab, D::NDX);
    SMPC_LDS(double, part, 64);
    SMPC_LDS(double, lpd_prev, D::NDX);
    SMPC_LANES(NT)
    {
      for (int i = lane; i < NDX; i += NT)
      {
        dx[i] = 0.0;
        lpd_prev[i] = 0.0;
        b.dxs[((size_t)inst * (H + 1)) * NDX + i] = 0.0;
      }
      part[lane] = 0.0;
    }
    SMPC_LANES_END_WAVE
    for (int t = 0; t < H; t++)
    {
      const double * lq = b.lq + ((size_t)inst * H + t) * D::LQ_STRIDE;
      const double * g = b.gains + ((size_t)inst * H + t) * D::G_STRIDE;
      const size_t lt = (size_t)inst * H + t;
      // du = k + K dx
      SMPC_LANES(NT)
      for (int i = lane; i < NU; i += NT)
      {
        const double * Kr = g + D::G_K + i * (NDX + 1);
        double acc = Kr[NDX];
        for (int j = 0; j < NDX; j++)
          acc += Kr[j] * dx[j];
        du[i] = acc;
        b.dus[lt * NU + i] = acc;
        part[lane] += lq[D::O_lu + i] * acc;
      }
      SMPC_LANES_END_WAVE
      // dnu of the box rows ; y = A dx + B du + f - mu p_{t+1}
      SMPC_LANES(NT)
      {
        for (int r = lane; r < NU + NA; r += NT)
        {
          const double d = lq[D::O_d + r], act = lq[D::O_act + r];
          const double lin = r < NU ? du[r] : dx[6 + r - NU];
          const double dnu = (act * lin + d) / mu;
          b.dvs[lt * NC + r] = dnu;
          part[lane] += lq[D::O_vpd + r] * (mu * dnu - d) - d * dnu;
        }
        // dense rows (wrench cones): dnu = Z dx + z from the explicitly pivoted multipliers
        for (int r = NU + NA + lane; r < NU + NA + D::NCD; r += NT)
        {
          const double d = lq[D::O_d + r];
          double dnu = d / mu; // inactive row: Z = 0, z = d / mu (riccati_dense_body writes no [Z z] for it in the stages of the light grid)
          if (lq[D::O_act + r] != 0.0)
          {
            const double * Zr = g + D::G_Z + (r - NU - NA) * (NDX + 1);
            dnu = Zr[NDX];
            for (int j = 0; j < NDX; j++)
              dnu += Zr[j] * dx[j];
          }
          b.dvs[lt * NC + r] = dnu;
          part[lane] += lq[D::O_vpd + r] * (mu * dnu - d) - d * dnu;
        }
        // kinodynamics variant: the folded frame-velocity rows (state only): dnu = (Cv dx + d) / mu
        if constexpr (D::NVEL > 0)
          for (int r = lane; r < D::NVEL; r += NT)
          {
            const double * Cr = lq + D::O_V + r * NDX;
            const int row = NU + NA + D::NCD + r;
            const double d = lq[D::O_d + row];
            double acc = d;
            for (int j = 0; j < NDX; j++)
              acc += Cr[j] * dx[j];
            const double dnu = acc / mu;
            b.dvs[lt * NC + row] = dnu;
            part[lane] += lq[D::O_vpd + row] * (mu * dnu - d) - d * dnu;
          }
        // Rows of [A B]: the joint-position rows are e_i + dt x (the joint-velocity row of the same joint) -- q_i+ = q_i + dt v_i+ (semi-implicit
        // Euler; assembled exactly so by fdyn_deriv_body) --, so only the base rows and the velocity rows are read: 34 of the biped's 56 rows,
        // a quarter fewer of the bytes that bound this kernel (the kinodynamics variant: the 12 base rows).  (A dx + B du)_i of the other rows follows from the velocity rows below.
        for (int i = lane; i < NDX; i += NT)
        {
          if (i >= 6 && i < NV)
            continue;
          if constexpr (D::KINO)
            if (i >= NV + 6)
            {
              // kinodynamics variant: the joint accelerations are controls, v_i+ = v_i + dt u_i: a unit entry and one dt (fdyn_deriv_body, r1())
              ab[i] = dx[i] + dt * du[D::NCM + (i - NV) - 6];
              continue;
            }
          const double * Ar = lq + D::O_A + i * NDX;
          const double * Br = lq + D::O_B + i * NU;
          double acc = 0.0;
          for (int j = 0; j < NDX; j++)
            acc += Ar[j] * dx[j];
          for (int j = 0; j < NU; j++)
            acc += Br[j] * du[j];
          ab[i] = acc;
        }
      }
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      for (int i = lane; i < NDX; i += NT)
      {
        const double acc = (i >= 6 && i < NV) ? dx[i] + dt * ab[NV + i] : ab[i];
        const double fi = lq[D::O_f + i], pn = g[D::G_pn + i];
        part[lane] += (lq[D::O_lx + i] - lpd_prev[i]) * dx[i] + lq[D::O_lpd + i] * acc;
        y[i] = acc + fi - mu * pn;
      }
      SMPC_LANES_END_WAVE
      // w = P~ y ; dx+ = y - mu w ; dlam+ = w + p_{t+1}
      SMPC_LANES(NT)
      for (int i = lane; i < NDX; i += NT)
      {
        // row i of the packed upper triangle: (j, i) for j < i -- the lanes of a wave read consecutive addresses --, (i, j) from the diagonal on
        static_assert(D::PT_PACKED, "forward_full_body reads the packed P~ of the FullDims sweeps");
        double w = 0.0;
#pragma unroll
        for (int j = 0; j < NDX; j++)
        {
          const double * src = j < i ? g + D::pt_row(j) + i : g + D::pt_row(i) + j;
          w += *src * y[j];
        }
        const double dxn = y[i] - mu * w;
        const double dl = w + g[D::G_pn + i];
        b.dxs[((size_t)inst * (H + 1) + t + 1) * NDX + i] = dxn;
        b.dlams[lt * NDX + i] = dl;
        part[lane] -= lq[D::O_f + i] * dl;
        lpd_prev[i] = lq[D::O_lpd + i];
        dx[i] = dxn; // (only y is read in this phase)
      }
      SMPC_LANES_END_WAVE
    }
    SMPC_LANES(NT)
    for (int i = lane; i < NDX; i += NT)
    {
      const int sl = ring_slot(ka.head, H - 1, R);
      const double lamH = b.lams[((size_t)inst * R + sl) * NDX + i];
      const double lxN = b.qN[(size_t)inst * NDX + i] + lamH;
      part[lane] += (lxN - lpd_prev[i]) * dx[i];
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double sacc = 0.0;
      for (int i = 0; i < 64; i++)
        sacc += part[i];
      b.scal[(size_t)inst * SC_N + SC_DPHI0] = sacc;
      b.ls_sel[inst] = -1;
    }
    SMPC_LANES_END_WAVE
  }
  // K_t of stages 0 .. nt-1 out of the gains block: [B][nt][NU][NDX] dense (grid = B * nt)
  template <class D>
  struct FullGainOutArgs
  {
    Buffers<D> b;
    int nt;
    double * out;
  };
  template <class D>
  SMPC_DEV void full_gains_out_body(const FullGainOutArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NDX = D::NDX, NU = D::NU;
    const Buffers<D> & b = ka.b;
    const int inst = block / ka.nt, t = block % ka.nt;
    const double * g = b.gains + ((size_t)inst * b.H + t) * D::G_STRIDE + D::G_K;
    double * out = ka.out + ((size_t)inst * ka.nt + t) * NU * NDX;
    SMPC_LANES(NT)
    for (int idx = lane; idx < NU * NDX; idx += NT)
      out[idx] = g[(idx / NDX) * (NDX + 1) + idx % NDX];
    SMPC_LANES_END_WAVE
  }

  // Targets between MPC knots for the whole-body loop of the full-dynamics examples (reference examples/go2_fulldynamics.py:
  // 268-292 with src/interpolator.cpp:5-78): x = interpolateState(xs[0 .. knots-1]), acc = interpolateLinear of the state
  // derivatives' acceleration part at t = 0, 1, forces = interpolateLinear of MPC::getContactForces(0 / 1), u = interpolateLinear
  // of us[0], us[1].  grid = B.
  template <class D>
  struct FullInterpArgs
  {
    Buffers<D> b;
    int head, knots;
    double delay, timestep;
    double *x_out, *acc_out, *f_out, *u_out; // device, any may be null
  };
  template <class D>
  SMPC_DEV void full_interp_body(const FullInterpArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NX = D::NX, NDX = D::NDX, NV = D::NV, NU = D::NU, NCM = D::NCM;
    const Buffers<D> & b = ka.b;
    const int inst = block, R = b.R;
    const size_t step = (size_t)(ka.delay / ka.timestep);
    const double s = (ka.delay - (double)step * ka.timestep) / ka.timestep;
    SMPC_LDS(double, e, D::NDX);
    SMPC_LDS(double, xo, D::NX);
    if (ka.x_out != nullptr)
    {
      const bool last = step >= (size_t)ka.knots - 1;
      const double * x0 = b.xs + ((size_t)inst * R + ring_slot(ka.head, last ? ka.knots - 1 : (int)step, R)) * NX;
      const double * x1 = b.xs + ((size_t)inst * R + ring_slot(ka.head, last ? ka.knots - 1 : (int)step + 1, R)) * NX;
      SMPC_LANES(NT)
      lanes_difference<D>(x0, x1, e, lane, 0);
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      lanes_integrate<D>(x0, e, last ? 0.0 : s, xo, lane, 0);
      SMPC_LANES_END_WAVE
      SMPC_LANES(NT)
      for (int i = lane; i < NX; i += NT)
        ka.x_out[(size_t)inst * NX + i] = last ? x0[i] : xo[i];
      SMPC_LANES_END_WAVE
    }
    const bool last1 = step >= 1;
    const double w1 = last1 ? 1.0 : s, w0 = last1 ? 0.0 : 1.0 - s;
    const double * u0 = b.us + ((size_t)inst * R + ring_slot(ka.head, 0, R)) * NU;
    const double * u1 = b.us + ((size_t)inst * R + ring_slot(ka.head, 1, R)) * NU;
    SMPC_LANES(NT)
    {
      if (ka.acc_out != nullptr)
        for (int i = lane; i < NV; i += NT)
        {
          const double * xd = b.xdot01 + (size_t)inst * 4 * NV;
          ka.acc_out[(size_t)inst * NV + i] = xd[2 * NV + NV + i] * w1 + xd[NV + i] * w0;
        }
      if (ka.f_out != nullptr)
        for (int i = lane; i < NCM; i += NT)
        {
          const double * f = b.forces + (size_t)inst * b.H * NCM;
          ka.f_out[(size_t)inst * NCM + i] = f[NCM + i] * w1 + f[i] * w0;
        }
      if (ka.u_out != nullptr)
        for (int i = lane; i < NU; i += NT)
          ka.u_out[(size_t)inst * NU + i] = u1[i] * w1 + u0[i] * w0;
    }
    SMPC_LANES_END_WAVE
    (void)NDX;
  }
} // namespace smpc
