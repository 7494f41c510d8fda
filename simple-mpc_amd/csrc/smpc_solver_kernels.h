// smpc_solver_kernels.h -- model-independent kernel bodies of the batched ProxDDP iteration:
//   riccati_body : HOT(4) proximal Riccati backward sweep, one 256-lane workgroup per instance,
//                  per-stage blocks staged in LDS (reference: gar::ProximalRiccatiSolver::backward
//                  selected at src/mpc.cpp:52; derivation in DESIGN.md)
//   forward_body : HOT(5) gains -> (dx, du, dnu, dlam) + directional derivative of the merit
//   select_body  : HOT(6) Armijo test over the evaluated line-search candidates
//   apply_body   : accept the step (LINEAR rollout x (+) alpha dx, reference src/mpc.cpp:44)
//   recede_body  : per control step: ring advance, warm-start shift (src/mpc.cpp:201-207), FK of the
//                  measured state, Raibert foothold + Bezier swing references
//                  (src/mpc.cpp:278-324, src/foot-trajectory.cpp:41-96)
#pragma once
#include "smpc_kino_kernels.h"

namespace smpc
{
  template <class D>
  struct SolverArgs
  {
    Buffers<D> b;
    int head;
    int j0, nj;
    double armijo_c1, reg_min, reg_max, reg_inc, reg_dec;
    int mode = 0;  // apply_body: 0 accept alpha, 1 tentative full step (with backup), 2 restore the backup
    int slots = 0; // > 0: the launch walks the compacted list of undecided instances, `slots` at a time
    // >= 0: SolverProxDDP::run's convergence test (reference src/mpc.cpp:43,212: TOL is the solver's tolerance): an instance whose
    // primal and dual infeasibilities are below it at the start of an iteration takes no step in that and the following iterations
    double stop_tol = -1.0;
  };

  // out(i,j) = init(i,j) + sum_k X[k*ldx + i] * Y[k*ldy + j]  for an M x N output, register tiles TM x TN
  template <int M, int N, int K, int TM, int TN, int NT, class Init, class Store>
  SMPC_DEV void mm_tn(const double * X, int ldx, const double * Y, int ldy, int lane, Init init, Store store)
  {
    static_assert(M % TM == 0 && N % TN == 0, "tile must divide the output");
    constexpr int TJ = N / TN, NTILES = (M / TM) * TJ;
    for (int tile = lane; tile < NTILES; tile += NT)
    {
      const int i0 = (tile / TJ) * TM, j0 = (tile % TJ) * TN;
      double acc[TM][TN];
#pragma unroll
      for (int a = 0; a < TM; a++)
#pragma unroll
        for (int c = 0; c < TN; c++)
          acc[a][c] = init(i0 + a, j0 + c);
#pragma unroll 4
      for (int k = 0; k < K; k++)
      {
        double xv[TM], yv[TN];
#pragma unroll
        for (int a = 0; a < TM; a++)
          xv[a] = X[k * ldx + i0 + a];
#pragma unroll
        for (int c = 0; c < TN; c++)
          yv[c] = Y[k * ldy + j0 + c];
#pragma unroll
        for (int a = 0; a < TM; a++)
#pragma unroll
          for (int c = 0; c < TN; c++)
            acc[a][c] += xv[a] * yv[c];
      }
#pragma unroll
      for (int a = 0; a < TM; a++)
#pragma unroll
        for (int c = 0; c < TN; c++)
          store(i0 + a, j0 + c, acc[a][c]);
    }
  }

  // In-place lower Cholesky of the N x N matrix M (leading dimension ld) by the whole workgroup.
  // Two phases per column: scaled column into `col`, then trailing update.
  template <int N, int NT>
  SMPC_DEV void wg_cholesky(double * Mx, int ld, double * col)
  {
    for (int k = 0; k < N; k++)
    {
      SMPC_LANES(NT)
      if (lane >= k && lane < N)
      {
        const double d = sqrt(Mx[k * ld + k]);
        col[lane] = lane == k ? d : Mx[lane * ld + k] / d;
      }
      SMPC_LANES_END
      SMPC_LANES(NT)
      {
        const int n = N - k; // rows k..N-1
        for (int idx = lane; idx < n * n; idx += NT)
        {
          const int i = k + idx / n, j = k + idx % n;
          if (j == k)
            Mx[i * ld + k] = col[i];
          else if (j <= i && i > k)
            Mx[i * ld + j] -= col[i] * col[j];
        }
      }
      SMPC_LANES_END
    }
  }

  // Solve (L L^T) X = RHS for NRHS columns held column-wise: lane c owns column c, kept in registers.
  // get(i, c) reads RHS(i, c); put(i, c, v) stores the result.  fwd_only: stop after L y = rhs.
  template <int N, int NT, class Get, class PutY, class PutX>
  SMPC_DEV void lane_chol_solve(const double * L, int ld, int ncols, int lane, Get get, PutY puty, PutX putx)
  {
    if (lane < ncols)
    {
      double col[N];
#pragma unroll
      for (int i = 0; i < N; i++)
        col[i] = get(i, lane);
#pragma unroll
      for (int i = 0; i < N; i++)
      {
        double s = col[i];
#pragma unroll
        for (int k = 0; k < i; k++)
          s -= L[i * ld + k] * col[k];
        col[i] = s / L[i * ld + i];
        puty(i, lane, col[i]);
      }
#pragma unroll
      for (int i = N - 1; i >= 0; i--)
      {
        double s = col[i];
#pragma unroll
        for (int k = i + 1; k < N; k++)
          s -= L[k * ld + i] * col[k];
        col[i] = s / L[i * ld + i];
        putx(i, lane, col[i]);
      }
    }
  }

  // =============================================================================================
  // riccati_body: grid = B, 256 lanes.  Backward sweep t = H-1 .. 0 for one instance.
  // =============================================================================================
  template <class D>
  struct RiccatiLds
  {
    static constexpr int NDX = D::NDX, NU = D::NU, NC = D::NC, NXU = D::NDX + D::NU;
    double P[NDX * NDX];   // P_{t+1} -> P~ -> P_t
    double MT[NDX * NXU];  // M = I + mu P (Cholesky factor), then [TA | TB]; later W = L_R^-1 [S^^T r^]
    double AB[NDX * NXU];  // [A | B]; later C
    double QS[NDX * NXU];  // [Q^ | S^]
    double Rh[NU * NU];
    double p[NDX], pt[NDX], qh[NDX], rh[NU], f[NDX], d[NC], col[NDX], wr[NU], kk[NU];
  };

  template <class D, int NT>
  SMPC_DEV void riccati_body(const SolverArgs<D> & ka, int block)
  {
    static_assert(NT >= 64, "needs one lane per right-hand side column");
    constexpr int NDX = D::NDX, NU = D::NU, NC = D::NC, NXU = NDX + NU;
    static_assert(NDX % 3 == 0 && NXU % 3 == 0 && NU % 3 == 0, "register tiles of 3 must divide the dimensions");
    const Buffers<D> & b = ka.b;
    const int H = b.H;
    const int inst = block;
    const double mu = b.model->mu;
    SMPC_LDS(RiccatiLds<D>, lds, 1);
    RiccatiLds<D> & s = lds[0];

    // terminal: P = Q_N, p = q_N
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
      // ---- load [A|B], f; M = I + mu P; save p_{t+1} ----
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
      }
      SMPC_LANES_END
      // pt = p + P f
      SMPC_LANES(NT)
      if (lane < NDX)
      {
        double acc = s.p[lane];
#pragma unroll 4
        for (int j = 0; j < NDX; j++)
          acc += s.P[lane * NDX + j] * s.f[j];
        s.pt[lane] = acc;
      }
      SMPC_LANES_END
      wg_cholesky<NDX, NT>(s.MT, NDX, s.col);
      // P~ = M^-1 P (columns 0..NDX-1), p~ = M^-1 pt (column NDX)
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
      // symmetrise P~ and store it for the forward pass
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
      // [TA|TB] = P~ [A|B]  (P~ symmetric -> X^T Y form); also stream P~ out
      SMPC_LANES(NT)
      {
        for (int idx = lane; idx < NDX * NDX; idx += NT)
          g[D::G_Pt + idx] = s.P[idx];
        mm_tn<NDX, NXU, NDX, 3, 3, NT>(
          s.P, NDX, s.AB, NXU, lane, [](int, int) { return 0.0; }, [&](int i, int j, double v) { s.MT[i * NXU + j] = v; });
      }
      SMPC_LANES_END
      // [Q^|S^] = [Q|S] + A^T [TA|TB];  R^ = R + B^T TB;  q^ = q + A^T p~;  r^ = r + B^T p~
      SMPC_LANES(NT)
      {
        mm_tn<NDX, NXU, NDX, 3, 3, NT>(
          s.AB, NXU, s.MT, NXU, lane, [&](int i, int j) { return j < NDX ? lq[D::q_off(i, j)] : lq[D::s_off(i, j - NDX)]; }, // (tile layout of the knot, Dims::O_T)
          [&](int i, int j, double v) { s.QS[i * NXU + j] = v; });
        mm_tn<NU, NU, NDX, 3, 3, NT>(
          s.AB + NDX, NXU, s.MT + NDX, NXU, lane, [&](int i, int j) { return lq[D::r_off(i, j)]; },
          [&](int i, int j, double v) { s.Rh[i * NU + j] = v; });
        if (lane < NXU)
        {
          double acc = lane < NDX ? lq[D::O_q + lane] : lq[D::O_r + lane - NDX];
#pragma unroll 4
          for (int k = 0; k < NDX; k++)
            acc += s.AB[k * NXU + lane] * s.pt[k];
          if (lane < NDX)
            s.qh[lane] = acc;
          else
            s.rh[lane - NDX] = acc;
        }
      }
      SMPC_LANES_END
      wg_cholesky<NU, NT>(s.Rh, NU, s.col);
      // W = L_R^-1 [S^^T r^] (kept, NU x (NDX+1), in MT);  [K k] = -R^^-1 [S^^T r^]
      SMPC_LANES(NT)
      lane_chol_solve<NU, NT>(
        s.Rh, NU, NDX + 1, lane, [&](int i, int c) { return c < NDX ? s.QS[c * NXU + NDX + i] : s.rh[i]; },
        [&](int i, int c, double v) { s.MT[i * (NDX + 1) + c] = v; },
        [&](int i, int c, double v) {
          g[D::G_K + i * (NDX + 1) + c] = -v;
          if (c == NDX)
            s.kk[i] = -v;
        });
      SMPC_LANES_END
      // load C (active rows) over [A|B]
      SMPC_LANES(NT)
      for (int idx = lane; idx < NC * NDX; idx += NT)
        s.AB[idx] = lq[D::O_C + idx];
      SMPC_LANES_END
      // P_t = Q^ - W^T W + C^T C / mu ;  p_t = q^ - W^T w_r + C^T d / mu
      SMPC_LANES(NT)
      {
        const double imu = 1.0 / mu;
        mm_tn<NDX, NDX, NU, 3, 3, NT>(
          s.MT, NDX + 1, s.MT, NDX + 1, lane, [](int, int) { return 0.0; },
          [&](int i, int j, double v) { s.P[i * NDX + j] = s.QS[i * NXU + j] - v; });
        if (lane >= NT - 64 && lane < NT - 64 + NDX)
        {
          const int i = lane - (NT - 64);
          double acc = s.qh[i];
#pragma unroll 4
          for (int m = 0; m < NU; m++)
            acc -= s.MT[m * (NDX + 1) + i] * s.MT[m * (NDX + 1) + NDX];
          double cd = 0.0;
#pragma unroll 4
          for (int r = 0; r < NC; r++)
            cd += s.AB[r * NDX + i] * s.d[r];
          s.p[i] = acc + imu * cd;
        }
      }
      SMPC_LANES_END
      SMPC_LANES(NT)
      {
        const double imu = 1.0 / mu;
        mm_tn<NDX, NDX, NC, 3, 3, NT>(
          s.AB, NDX, s.AB, NDX, lane, [](int, int) { return 0.0; }, [&](int i, int j, double v) { s.P[i * NDX + j] += imu * v; });
      }
      SMPC_LANES_END
      // symmetrise P_t
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

  // =============================================================================================
  // forward_body: grid = B, 64 lanes.  Forward sweep + merit directional derivative.
  // =============================================================================================
  template <class D>
  SMPC_DEV void forward_body(const SolverArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NDX = D::NDX, NU = D::NU, NC = D::NC;
    static_assert(NDX <= 64 && NU + NC <= 64, "one lane per row");
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const int inst = block;
    const double mu = b.model->mu;
    SMPC_LDS(double, dx, D::NDX);
    SMPC_LDS(double, du, D::NU);
    SMPC_LDS(double, y, D::NDX);
    SMPC_LDS(double, part, 64);
    SMPC_LDS(double, lpd_prev, D::NDX);
    SMPC_LANES(NT)
    {
      if (lane < NDX)
      {
        dx[lane] = 0.0;
        lpd_prev[lane] = 0.0;
        b.dxs[((size_t)inst * (H + 1)) * NDX + lane] = 0.0;
      }
      part[lane] = 0.0;
    }
    SMPC_LANES_END
    for (int t = 0; t < H; t++)
    {
      const double * lq = b.lq + ((size_t)inst * H + t) * D::LQ_STRIDE;
      const double * g = b.gains + ((size_t)inst * H + t) * D::G_STRIDE;
      const size_t lt = (size_t)inst * H + t;
      // du = k + K dx  (lanes 0..NU-1) ; dnu = (C dx + d)/mu (lanes NU..NU+NC-1)
      SMPC_LANES(NT)
      {
        if (lane < NU)
        {
          const double * Kr = g + D::G_K + lane * (NDX + 1);
          double acc = Kr[NDX];
#pragma unroll 4
          for (int j = 0; j < NDX; j++)
            acc += Kr[j] * dx[j];
          du[lane] = acc;
          b.dus[lt * NU + lane] = acc;
          part[lane] += lq[D::O_lu + lane] * acc;
        }
        else if (lane < NU + NC)
        {
          const int r = lane - NU;
          const double * Cr = lq + D::O_C + r * NDX;
          double acc = lq[D::O_d + r];
#pragma unroll 4
          for (int j = 0; j < NDX; j++)
            acc += Cr[j] * dx[j];
          const double dnu = acc / mu;
          b.dvs[lt * NC + r] = dnu;
          // vpd (mu dnu - d) - d dnu
          part[lane] += lq[D::O_vpd + r] * (mu * dnu - lq[D::O_d + r]) - lq[D::O_d + r] * dnu;
        }
      }
      SMPC_LANES_END
      // y = A dx + B du + f - mu p_{t+1}
      SMPC_LANES(NT)
      if (lane < NDX)
      {
        const double * Ar = lq + D::O_A + lane * NDX;
        const double * Br = lq + D::O_B + lane * NU;
        double acc = 0.0;
#pragma unroll 4
        for (int j = 0; j < NDX; j++)
          acc += Ar[j] * dx[j];
#pragma unroll 4
        for (int j = 0; j < NU; j++)
          acc += Br[j] * du[j];
        const double fi = lq[D::O_f + lane], pn = g[D::G_pn + lane];
        // lx.dx - lpd_{t-1}.dx + lpd_t.(A dx + B du)
        part[lane] += (lq[D::O_lx + lane] - lpd_prev[lane]) * dx[lane] + lq[D::O_lpd + lane] * acc;
        y[lane] = acc + fi - mu * pn;
      }
      SMPC_LANES_END
      // w = P~ y ; dx+ = y - mu w ; dlam+ = w + p_{t+1}
      SMPC_LANES(NT)
      if (lane < NDX)
      {
        const double * Pr = g + D::G_Pt + lane * NDX;
        double w = 0.0;
#pragma unroll 4
        for (int j = 0; j < NDX; j++)
          w += Pr[j] * y[j];
        const double dxn = y[lane] - mu * w;
        const double dl = w + g[D::G_pn + lane];
        b.dxs[((size_t)inst * (H + 1) + t + 1) * NDX + lane] = dxn;
        b.dlams[lt * NDX + lane] = dl;
        part[lane] -= lq[D::O_f + lane] * dl;
        lpd_prev[lane] = lq[D::O_lpd + lane];
        dx[lane] = dxn; // dx is not read in this phase (only y is), so it can be advanced in place
      }
      SMPC_LANES_END
    }
    // terminal gradient term (lxN - lpd_{H-1}) . dx_H with lxN = qN + lambda_H
    SMPC_LANES(NT)
    if (lane < NDX)
    {
      const int sl = ring_slot(ka.head, H - 1, R);
      const double lamH = b.lams[((size_t)inst * R + sl) * NDX + lane];
      const double lxN = b.qN[(size_t)inst * NDX + lane] + lamH;
      part[lane] += (lxN - lpd_prev[lane]) * dx[lane];
    }
    SMPC_LANES_END
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double sacc = 0.0;
      for (int i = 0; i < 64; i++)
        sacc += part[i];
      b.scal[(size_t)inst * SC_N + SC_DPHI0] = sacc;
      b.ls_sel[inst] = -1;
    }
    SMPC_LANES_END
  }

  // =============================================================================================
  // select_body: grid = ceil(B/64), one lane per instance.
  // =============================================================================================
  template <class D>
  SMPC_DEV void select_body(const SolverArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    const Buffers<D> & b = ka.b;
    const int H = b.H;
    SMPC_LANES(NT)
    {
      const int inst = block * NT + lane;
      if (inst < b.B)
      {
        double * sc = b.scal + (size_t)inst * SC_N;
        if (ka.j0 == 0)
        {
          double phi = 0.0, cost = 0.0, prim = 0.0, dual = 0.0;
          // (eight stages' partials are read together, then added in stage order: the sums are those of the plain loop, the chain of dependent
          //  round trips to memory is an eighth as long -- a lane owns an instance here, its reads are strided)
          const double * p0 = b.parts0 + (size_t)inst * (H + 1) * 4;
          int t = 0;
          for (; t + 8 <= H + 1; t += 8)
          {
            double v[8][4];
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
              for (int c = 0; c < 4; c++)
                v[k][c] = p0[(t + k) * 4 + c];
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
              phi += v[k][0];
              cost += v[k][1];
              prim = fmax(prim, v[k][2]);
              dual = fmax(dual, v[k][3]);
            }
          }
          for (; t <= H; t++)
          {
            const double * p = p0 + t * 4;
            phi += p[0];
            cost += p[1];
            prim = fmax(prim, p[2]);
            dual = fmax(dual, p[3]);
          }
          sc[SC_PHI0] = phi;
          sc[SC_COST] = cost;
          sc[SC_PRIM] = prim;
          sc[SC_DUAL] = dual;
          if (ka.stop_tol >= 0.0 && fmax(prim, dual) <= ka.stop_tol)
          { // converged: alpha = 0, the iterate stays (no candidate is looked at, the regularisation is left alone)
            b.ls_sel[inst] = 0;
            sc[SC_ALPHA] = 0.0;
            sc[SC_PHI_NEW] = phi;
            sc[SC_PRIM_NEW] = prim;
            sc[SC_LS_INDEX] = 0.0;
            sc[SC_LS_FAILED] = 0.0;
          }
        }
        if (b.ls_sel[inst] < 0)
        {
          const double phi0 = sc[SC_PHI0], dphi0 = sc[SC_DPHI0];
          double alpha = 1.0;
          for (int i = 0; i < ka.j0; i++)
            alpha *= 0.5;
          int sel = -1;
          double phi_sel = 0.0, prim_sel = 0.0;
          for (int jj = 0; jj < ka.nj; jj++)
          {
            const int j = ka.j0 + jj;
            double phi = 0.0, prim = 0.0;
            const double * pT = b.partsT + ((size_t)inst * D::LS_N + j) * (H + 1) * 2;
            int t = 0;
            for (; t + 8 <= H + 1; t += 8)
            {
              double v[8][2];
#pragma unroll
              for (int k = 0; k < 8; k++)
              {
                v[k][0] = pT[(t + k) * 2];
                v[k][1] = pT[(t + k) * 2 + 1];
              }
#pragma unroll
              for (int k = 0; k < 8; k++)
              {
                phi += v[k][0];
                prim = fmax(prim, v[k][1]);
              }
            }
            for (; t <= H; t++)
            {
              phi += pT[t * 2];
              prim = fmax(prim, pT[t * 2 + 1]);
            }
            const bool ok = phi <= phi0 + ka.armijo_c1 * alpha * dphi0;
            const bool last = j == D::LS_N - 1;
            if (ok || last)
            {
              sel = j;
              phi_sel = phi;
              prim_sel = prim;
              sc[SC_LS_FAILED] = ok ? 0.0 : 1.0;
              break;
            }
            alpha *= 0.5;
          }
          if (sel >= 0)
          {
            b.ls_sel[inst] = sel;
            sc[SC_ALPHA] = alpha;
            sc[SC_PHI_NEW] = phi_sel;
            sc[SC_PRIM_NEW] = prim_sel;
            sc[SC_LS_INDEX] = (double)sel;
            const double preg = sc[SC_PREG];
            sc[SC_PREG] = sc[SC_LS_FAILED] != 0.0 ? fmin(preg * ka.reg_inc, ka.reg_max) : fmax(preg * ka.reg_dec, ka.reg_min);
            // (copies of the accepted candidate's outputs: eight entries are read before the first is written, as above)
            auto copy = [](double * dst, const double * src, int n) {
              int i = 0;
              for (; i + 8 <= n; i += 8)
              {
                double v[8];
#pragma unroll
                for (int k = 0; k < 8; k++)
                  v[k] = src[i + k];
#pragma unroll
                for (int k = 0; k < 8; k++)
                  dst[i + k] = v[k];
              }
              for (; i < n; i++)
                dst[i] = src[i];
            };
            copy(b.xdot01 + (size_t)inst * 4 * D::NV, b.xdotT + ((size_t)inst * D::LS_N + sel) * 4 * D::NV, 4 * D::NV);
            if (b.forcesT != nullptr)
              copy(b.forces + (size_t)inst * H * b.nforce, b.forcesT + ((size_t)inst * D::LS_N + sel) * H * b.nforce, H * b.nforce);
          }
        }
      }
    }
    SMPC_LANES_END
  }

  // =============================================================================================
  // compact_body: grid = 1, 64 lanes: ordered list of the instances whose line search is still undecided
  // (deterministic: lane l scans a contiguous range, exclusive prefix over the 64 counts)
  // =============================================================================================
  template <class D>
  SMPC_DEV void compact_body(const SolverArgs<D> & ka, int)
  {
    constexpr int NT = 64;
    const Buffers<D> & b = ka.b;
    SMPC_LDS(int, cnt, NT + 1);
    const int per = (b.B + NT - 1) / NT;
    SMPC_LANES(NT)
    {
      int c = 0;
      for (int i = lane * per; i < (lane + 1) * per && i < b.B; i++)
        c += b.ls_sel[i] < 0 ? 1 : 0;
      cnt[lane] = c;
    }
    SMPC_LANES_END
    SMPC_LANES(NT)
    if (lane == 0)
    {
      int run = 0;
      for (int l = 0; l < NT; l++)
      {
        const int c = cnt[l];
        cnt[l] = run;
        run += c;
      }
      cnt[NT] = run;
      b.und_list[b.B] = run;
    }
    SMPC_LANES_END
    SMPC_LANES(NT)
    {
      int pos = cnt[lane];
      for (int i = lane * per; i < (lane + 1) * per && i < b.B; i++)
        if (b.ls_sel[i] < 0)
          b.und_list[pos++] = i;
    }
    SMPC_LANES_END
  }

  // =============================================================================================
  // apply_body: grid = B * (H+1), 64 lanes: accept the step of size alpha for (inst, t)
  // =============================================================================================
  template <class D>
  SMPC_DEV void apply_inst(const SolverArgs<D> & ka, int inst);

  // grid = B (slots == 0) or `slots` walking the compacted list of undecided instances (slots > 0); one wave per instance
  template <class D>
  SMPC_DEV void apply_body(const SolverArgs<D> & ka, int block)
  {
    const int count = ka.slots > 0 ? ka.b.und_list[ka.b.B] : block + 1;
    const int stride = ka.slots > 0 ? ka.slots : ka.b.B;
    for (int m = block; m < count; m += stride)
      apply_inst<D>(ka, ka.slots > 0 ? ka.b.und_list[m] : m);
  }

  // mode 0: x <- x (+) alpha dx, u, nu, lam += alpha d.  (alpha of the instance's accepted candidate)
  // mode 1: tentative full step: the same with alpha = 1, the old values saved to the backup buffers, and the
  //         regularisation moved as after a successful line search (reference ProxDDP: decrease on success)
  // mode 2: restore the backup (the full step was rejected)
  // A SIMD pass costs the same for 1 or 64 lanes: the SE(3) parts of all H+1 nodes are integrated side by side
  // (lane = node), everything else is a flat coalesced stream over the instance's ring.
  template <class D>
  SMPC_DEV void apply_inst(const SolverArgs<D> & ka, int inst)
  {
    constexpr int NT = 64;
    constexpr int NX = D::NX, NDX = D::NDX, NU = D::NU, NC = D::NC, NQ = D::NQ, NV = D::NV;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const size_t ib = (size_t)inst * R;
    const bool tent = ka.mode == 1, restore = ka.mode == 2;
    const double alpha = tent ? 1.0 : b.scal[(size_t)inst * SC_N + SC_ALPHA];
    // ---- states: base pose on the manifold (lane = node), joints / velocities linearly ----
    for (int t0 = 0; t0 <= H; t0 += NT)
    {
      SMPC_LANES(NT)
      {
        const int t = t0 + lane;
        if (t <= H)
        {
          const int st = ring_slot(ka.head, t, R);
          double * x = b.xs + (ib + st) * NX;
          double * xb = b.xs_b + (ib + st) * NX;
          const double * dx = b.dxs + ((size_t)inst * (H + 1) + t) * NDX;
          if (restore)
          {
            for (int i = 0; i < 7; i++)
              x[i] = xb[i];
          }
          else
          {
            const V3 dv = alpha * ld3(dx), dw = alpha * ld3(dx + 3);
            const Quat q0{x[3], x[4], x[5], x[6]};
            const V3 p0 = ld3(x);
            if (tent)
            {
              st3(xb, p0);
              xb[3] = q0.x;
              xb[4] = q0.y;
              xb[5] = q0.z;
              xb[6] = q0.w;
            }
            const SE3 E = exp6(dv, dw);
            st3(x, p0 + quat_to_R(q0) * E.p);
            Quat qn = quat_mul(q0, quat_exp(dw));
            const double n = 1.0 / sqrt(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
            x[3] = qn.x * n;
            x[4] = qn.y * n;
            x[5] = qn.z * n;
            x[6] = qn.w * n;
          }
        }
      }
      SMPC_LANES_END_WAVE
    }
    SMPC_LANES(NT)
    {
      // flat streams over the instance's ring: entry i of node t of `dst` (n entries per node behind offset `doff`, `dn` doubles per ring slot)
      // += alpha * entry `soff + i` of node t of `step` (`sn` doubles per node, `nodes_s` nodes per instance).  Four entries per lane are read
      // before the first is written: a store and the next entry's loads are the same arrays as far as the compiler knows, and taken one by one
      // every entry waited for the round trip of the one before it (0.16 ms per launch for 4096 quadrupeds, 0.34 ms for 1024 bipeds)
      auto stream = [&](double * dst, double * bak, const double * step, int n, int nodes, int dn, int doff, int sn, int soff, int nodes_s) {
        const int total = nodes * n;
        for (int idx0 = lane; idx0 < total; idx0 += 4 * NT)
        {
          size_t o[4];
          double v[4], d[4];
#pragma unroll
          for (int k = 0; k < 4; k++)
          {
            const int idx = idx0 + k * NT;
            const bool ok = idx < total;
            const int t = ok ? idx / n : 0, i = ok ? idx % n : 0;
            o[k] = (ib + ring_slot(ka.head, t, R)) * dn + doff + i;
            v[k] = restore ? bak[o[k]] : dst[o[k]];
            d[k] = restore ? 0.0 : step[((size_t)inst * nodes_s + t) * sn + soff + i];
          }
#pragma unroll
          for (int k = 0; k < 4; k++)
            if (idx0 + k * NT < total)
            {
              if (restore)
                dst[o[k]] = v[k];
              else
              {
                if (tent)
                  bak[o[k]] = v[k];
                dst[o[k]] = v[k] + alpha * d[k];
              }
            }
        }
      };
      stream(b.xs, b.xs_b, b.dxs, NX - 7, H + 1, NX, 7, NDX, 6, H + 1); // linear entries of a state: x[7 + k] <-> dx[6 + k]
      stream(b.us, b.us_b, b.dus, NU, H, NU, 0, NU, 0, H);               // controls and multipliers: ring slot of node t
      stream(b.vs, b.vs_b, b.dvs, NC, H, NC, 0, NC, 0, H);
      stream(b.lams, b.lams_b, b.dlams, NDX, H, NDX, 0, NDX, 0, H);
      if (b.es != nullptr) // multipliers of the friction-cone rows
        stream(b.es, b.es_b, b.des, 2 * D::NF, H, 2 * D::NF, 0, 2 * D::NF, 0, H);
      if (b.ls != nullptr) // multipliers of the land_cstr rows
        stream(b.ls, b.ls_b, b.dls, D::NF, H, D::NF, 0, D::NF, 0, H);
      if (b.CN != nullptr && lane < 3)
      { // multipliers of the terminal constraint
        const size_t o = (size_t)inst * 3 + lane;
        if (restore)
          b.vN[o] = b.vN_b[o];
        else
        {
          const double v = b.vN[o];
          if (tent)
            b.vN_b[o] = v;
          b.vN[o] = v + alpha * b.dvN[o];
        }
      }
      if (tent && lane == 0)
      {
        double * sc = b.scal + (size_t)inst * SC_N;
        sc[SC_PREG_OLD] = sc[SC_PREG];
        sc[SC_PREG] = fmax(sc[SC_PREG] * ka.reg_dec, ka.reg_min);
      }
    }
    SMPC_LANES_END_WAVE
    static_assert(NQ == NV + 1, "free-flyer state layout");
  }

  // =============================================================================================
  // term_step_body: grid = ceil(B / 64), lane = instance; after the forward sweep of a problem with a terminal constraint.
  // The derivative pass folded the rows into the terminal node (Q_N += C^T C / mu, q_N += C^T v+), so the sweeps are unchanged;
  // here dv_N = (C dx_H + d) / mu, and the directional derivative of the merit gets what the fold leaves out:
  // C^T (v+ - v) . dx_H - d . dv_N = -|d|^2 / mu   (d = mu (v+ - v)).
  // =============================================================================================
  template <class D>
  SMPC_DEV void term_step_body(const SolverArgs<D> & ka, int block)
  {
    constexpr int NT = 64, NDX = D::NDX;
    const Buffers<D> & b = ka.b;
    SMPC_LANES(NT)
    {
      const int inst = block * NT + lane;
      if (inst < b.B && b.CN != nullptr)
      {
        const double * cn = b.CN + (size_t)inst * (3 * NDX + 3);
        const double * dx = b.dxs + ((size_t)inst * (b.H + 1) + b.H) * NDX;
        const double mu = b.model->mu;
        double corr = 0.0;
        for (int r = 0; r < 3; r++)
        {
          double acc = cn[3 * NDX + r];
          for (int i = 0; i < NDX; i++)
            acc += cn[r * NDX + i] * dx[i];
          b.dvN[(size_t)inst * 3 + r] = acc / mu;
          corr += cn[3 * NDX + r] * cn[3 * NDX + r];
        }
        b.scal[(size_t)inst * SC_N + SC_DPHI0] -= corr / mu;
      }
    }
    SMPC_LANES_END_WAVE
  }

  // =============================================================================================
  // merit0_body: grid = ceil(B / 64), lane = instance: merit, cost and infeasibilities of the current point from the
  // per-stage partials of the derivative pass (fixed stage order).  slots > 0: the listed instances only.
  // =============================================================================================
  template <class D>
  SMPC_DEV void merit0_body(const SolverArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    const Buffers<D> & b = ka.b;
    const int H = b.H;
    SMPC_LANES(NT)
    {
      const int m = block * NT + lane;
      const int count = ka.slots > 0 ? b.und_list[b.B] : b.B;
      if (m < count)
      {
        const int inst = ka.slots > 0 ? b.und_list[m] : m;
        double * sc = b.scal + (size_t)inst * SC_N;
        double phi = 0.0, cost = 0.0, prim = 0.0, dual = 0.0;
        for (int t = 0; t <= H; t++)
        {
          const double * p = b.parts0 + ((size_t)inst * (H + 1) + t) * 4;
          phi += p[0];
          cost += p[1];
          prim = fmax(prim, p[2]);
          dual = fmax(dual, p[3]);
        }
        sc[SC_PHI0] = phi;
        sc[SC_COST] = cost;
        sc[SC_PRIM] = prim;
        sc[SC_DUAL] = dual;
      }
    }
    SMPC_LANES_END_WAVE
  }

  // =============================================================================================
  // spec_select_body: grid = ceil(B / 64), lane = instance.  The derivative pass has just been run at the tentative
  // point x (+) dx: its merit is the line-search value phi(alpha = 1).  Armijo accepted -> the step is final (and the
  // knot of the next iteration is already there); rejected -> mark the instance undecided and put the old
  // regularisation back (the backtracking path restores the iterate, tries alpha = 2^-1 .. and re-derives).
  // =============================================================================================
  template <class D>
  SMPC_DEV void spec_select_body(const SolverArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    const Buffers<D> & b = ka.b;
    const int H = b.H;
    SMPC_LANES(NT)
    {
      const int inst = block * NT + lane;
      if (inst < b.B)
      {
        double * sc = b.scal + (size_t)inst * SC_N;
        double phi = 0.0, cost = 0.0, prim = 0.0, dual = 0.0;
        for (int t = 0; t <= H; t++)
        {
          const double * p = b.parts0 + ((size_t)inst * (H + 1) + t) * 4;
          phi += p[0];
          cost += p[1];
          prim = fmax(prim, p[2]);
          dual = fmax(dual, p[3]);
        }
        const bool ok = phi <= sc[SC_PHI0] + ka.armijo_c1 * sc[SC_DPHI0];
        if (ok)
        {
          b.ls_sel[inst] = 0;
          sc[SC_ALPHA] = 1.0;
          sc[SC_PHI_NEW] = phi;
          sc[SC_PRIM_NEW] = prim;
          sc[SC_LS_INDEX] = 0.0;
          sc[SC_LS_FAILED] = 0.0;
          sc[SC_PHI0] = phi;
          sc[SC_COST] = cost;
          sc[SC_PRIM] = prim;
          sc[SC_DUAL] = dual;
        }
        else
        {
          b.ls_sel[inst] = -1;
          sc[SC_PREG] = sc[SC_PREG_OLD];
        }
      }
    }
    SMPC_LANES_END_WAVE
  }

  // =============================================================================================
  // recede_body: grid = B, 64 lanes.  `head` is the NEW head (already advanced by the host).
  // =============================================================================================
  template <class D>
  struct RecedeArgs
  {
    Buffers<D> b;
    int head;
    const double * X;   // [B][NX] measured states (device)
    int land[D::NF];    // first landing time per foot or -1 (src/mpc.cpp:283-285)
    int T_fly, T_contact;
    double swing_apex, timestep;
    // (velocity commands are per instance: Buffers::vbase)
    int shift;          // 1: regular control step; 0: only (re)generate references
    double reg_init;
  };

  SMPC_HD V3 bezier8(V3 p0, V3 p1, double apex, float tf)
  {
    V3 mid = 0.75 * p0 + 0.25 * p1;
    mid.z += apex;
    const double u = (double)tf, uo = 1.0 - u;
    double bc = 1.0, tn = 1.0;
    V3 tmp = uo * p0;
    for (int i = 1; i < 8; i++)
    {
      tn = tn * u;
      bc = bc * (double)(8 - i + 1) / (double)i;
      const V3 cp = i < 4 ? p0 : (i == 4 ? mid : p1);
      tmp = uo * (tmp + (tn * bc) * cp);
    }
    return tmp + (tn * u) * p1;
  }

  template <class D>
  SMPC_DEV void recede_body(const RecedeArgs<D> & ka, int block)
  {
    constexpr int NT = 64;
    constexpr int NX = D::NX, NDX = D::NDX, NU = D::NU, NC = D::NC, NF = D::NF, NJ = D::NJ;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const int inst = block;
    const DevModel<D> & md = *b.model;
    const size_t ib = (size_t)inst * R;
    const double * xm = ka.X + (size_t)inst * NX;
    SMPC_LDS(double, oR, D::NJ * 9);
    SMPC_LDS(double, op, D::NJ * 3);
    SMPC_LDS(double, fp, D::NF * 3);
    SMPC_LDS(double, fse, D::NF * 6);
    const int s0 = ring_slot(ka.head, 0, R), sHm1 = ring_slot(ka.head, H - 1, R), sH = ring_slot(ka.head, H, R);
    const int sHm2 = ring_slot(ka.head, H - 2, R);
    // ---- warm-start shift on the ring (head already advanced): x_0 := measured, duplicate the tail ----
    SMPC_LANES(NT)
    if (ka.shift)
    {
      for (int i = lane; i < NX; i += NT)
      {
        b.xs[(ib + s0) * NX + i] = xm[i];
        b.xs[(ib + sH) * NX + i] = b.xs[(ib + sHm1) * NX + i];
      }
      for (int i = lane; i < NU; i += NT)
        b.us[(ib + sHm1) * NU + i] = b.us[(ib + sHm2) * NU + i];
      for (int i = lane; i < NC; i += NT)
        b.vs[(ib + sHm1) * NC + i] = 0.0;
      if (b.es != nullptr && lane < 2 * NF)
        b.es[(ib + sHm1) * 2 * NF + lane] = 0.0;
      if (b.ls != nullptr && lane < NF)
        b.ls[(ib + sHm1) * NF + lane] = 0.0;
      if (lane < 6)
        b.vref[(ib + sHm1) * 6 + lane] = b.vbase[(size_t)inst * 6 + lane]; // setVelocityBase(H-1, velocity_base_)
      for (int i = lane; i < NDX; i += NT)
        b.lams[(ib + sHm1) * NDX + i] = 0.0;
      if (lane == 0)
        b.scal[(size_t)inst * SC_N + SC_PREG] = ka.reg_init; // regularisation restarts with every solver run
    }
    SMPC_LANES_END
    // ---- FK of the measured state (reference: RobotDataHandler::updateInternalData, src/robot-handler.cpp:114-127) ----
    for (int lvl = 0; lvl < md.nlevels; lvl++)
    {
      SMPC_LANES(NT)
      if (lane < NJ && md.level[lane] == lvl)
      {
        const int j = lane;
        M3 Rm;
        V3 p;
        if (j == 0)
        {
          Rm = quat_to_R(Quat{xm[3], xm[4], xm[5], xm[6]});
          p = ld3(xm);
        }
        else
        {
          const int par = md.parent[j];
          const double ang = xm[6 + j];
          const double sn = sin(ang), c = cos(ang);
          const int jt = md.jtype[j];
          M3 Rq = jt == 1 ? M3{1, 0, 0, 0, c, -sn, 0, sn, c} : (jt == 2 ? M3{c, 0, sn, 0, 1, 0, -sn, 0, c} : M3{c, -sn, 0, sn, c, 0, 0, 0, 1});
          const M3 Rp = ldm3(&oR[par * 9]);
          Rm = Rp * (ldm3(md.jpR[j]) * Rq);
          p = ld3(&op[par * 3]) + Rp * ld3(md.jpp[j]);
        }
        stm3(&oR[j * 9], Rm);
        st3(&op[j * 3], p);
      }
      SMPC_LANES_END
    }
    // ---- Raibert foothold, swing start/end update (src/mpc.cpp:280-302) ----
    SMPC_LANES(NT)
    if (lane < NF)
    {
      const int f = lane;
      const int j = md.foot_joint[f];
      const V3 pf = ldm3(&oR[j * 9]) * ld3(md.foot_p[f]) + ld3(&op[j * 3]);
      st3(&fp[f * 3], pf);
      const V3 bp = ld3(&op[0]);
      const V3 refp = ldm3(&oR[0]) * ld3(md.foot_ref_p[f]) + bp;
      const double tw0 = -(refp.y - bp.y), tw1 = refp.x - bp.x;
      const double span = (double)(ka.T_fly + ka.T_contact) * ka.timestep;
      const double * vb = b.vbase + (size_t)inst * 6;
      const V3 next = mk3(refp.x + (vb[0] + vb[5] * tw0) * span, refp.y + (vb[1] + vb[5] * tw1) * span, pf.z);
      double * ft = b.ftraj + ((size_t)inst * NF + f) * 6;
      const bool update = !(ka.land[f] < ka.T_fly);
      if (update)
      {
        st3(ft, pf);
        st3(ft + 3, next);
      }
      st3(&fse[f * 6], ld3(ft));
      st3(&fse[f * 6 + 3], ld3(ft + 3));
    }
    SMPC_LANES_END
    // ---- sample the references for every horizon stage (src/foot-trajectory.cpp:64-82) ----
    SMPC_LANES(NT)
    for (int idx = lane; idx < H * NF; idx += NT)
    {
      const int k = idx / NF, f = idx % NF;
      const int t = ka.land[f] - k;
      const V3 p0 = ld3(&fse[f * 6]), p1 = ld3(&fse[f * 6 + 3]);
      V3 p;
      if (t < 0)
        p = p1;
      else if (t > ka.T_fly)
        p = p0;
      else
        p = bezier8(p0, p1, ka.swing_apex, float(ka.T_fly - t) / float(ka.T_fly));
      st3(b.foot_ref + (((size_t)inst * H + k) * NF + f) * 3, p);
    }
    SMPC_LANES_END
    // ---- updateTerminalConstraint: mean of the last foot references, at the CoM height of the reference state (src/mpc.cpp:313-323) ----
    if (b.CN != nullptr)
    {
      SMPC_LANES(NT)
      if (lane < NF)
      {
        const int f = lane, t = ka.land[f] - (H - 1);
        const V3 p0 = ld3(&fse[f * 6]), p1 = ld3(&fse[f * 6 + 3]);
        st3(&fp[f * 3], t < 0 ? p1 : (t > ka.T_fly ? p0 : bezier8(p0, p1, ka.swing_apex, float(ka.T_fly - t) / float(ka.T_fly))));
      }
      SMPC_LANES_END
      SMPC_LANES(NT)
      if (lane < 3)
      {
        double acc = 0.0;
        for (int f = 0; f < NF; f++)
          acc += fp[f * 3 + lane];
        b.dcm_ref[(size_t)inst * 3 + lane] = acc / (double)NF + (lane == 2 ? b.com0z : 0.0);
      }
      SMPC_LANES_END
    }
  }

  // the same n <= 8 values into a strided array of `count` records (per-stage reference setters, broadcast over the batch)
  struct FillStridedArgs
  {
    double * base;
    size_t stride;
    int count, n;
    double v[8];
  };
  SMPC_DEV void fill_strided_body(const FillStridedArgs & ka, int block)
  {
    constexpr int NT = 64;
    SMPC_LANES(NT)
    {
      const int r = block * NT + lane;
      if (r < ka.count)
        for (int i = 0; i < ka.n; i++)
          ka.base[(size_t)r * ka.stride + i] = ka.v[i];
    }
    SMPC_LANES_END_WAVE
  }

  // CentroidalFwdDynamics + IntegratorEuler with derivatives (reference src/centroidal-dynamics.cpp:79-81; SURVEY 8a row
  // a6, App. B.1), batched: lane = instance (the model is 9-dimensional: one instance per lane is the wide mapping).
  struct CentroidalArgs
  {
    double mass, dt, g[3];
    int nf, batch;
    const double *X, *U, *pos;       // [B][9], [B][3 nf], [B][nf][3] (device)
    const unsigned char * contact;   // [B][nf] (device)
    double *Xn, *A, *Bm;             // [B][9], [B][81], [B][9][3 nf] (device); A / Bm may be null
  };
  SMPC_DEV void centroidal_body(const CentroidalArgs & ka, int block)
  {
    constexpr int NT = 64;
    SMPC_LANES(NT)
    {
      const int inst = block * NT + lane;
      if (inst < ka.batch)
      {
        const int nf = ka.nf, nu = 3 * nf;
        const double * x = ka.X + (size_t)inst * 9;
        const double * u = ka.U + (size_t)inst * nu;
        const double dt = ka.dt, im = 1.0 / ka.mass;
        double hd[3] = {ka.mass * ka.g[0], ka.mass * ka.g[1], ka.mass * ka.g[2]}, Ld[3] = {0, 0, 0};
        double * A = ka.A ? ka.A + (size_t)inst * 81 : nullptr;
        double * Bm = ka.Bm ? ka.Bm + (size_t)inst * 9 * nu : nullptr;
        if (A)
        {
          for (int i = 0; i < 81; i++)
            A[i] = 0.0;
          for (int i = 0; i < 9; i++)
            A[i * 9 + i] = 1.0;
          for (int i = 0; i < 3; i++)
            A[i * 9 + 3 + i] = dt * im;
        }
        if (Bm)
          for (int i = 0; i < 9 * nu; i++)
            Bm[i] = 0.0;
        for (int f = 0; f < nf; f++)
        {
          if (!ka.contact[(size_t)inst * nf + f])
            continue;
          const double * F = u + 3 * f;
          const double * p = ka.pos + ((size_t)inst * nf + f) * 3;
          const double r[3] = {p[0] - x[0], p[1] - x[1], p[2] - x[2]};
          for (int i = 0; i < 3; i++)
            hd[i] += F[i];
          Ld[0] += r[1] * F[2] - r[2] * F[1];
          Ld[1] += r[2] * F[0] - r[0] * F[2];
          Ld[2] += r[0] * F[1] - r[1] * F[0];
          if (A)
          {
            // d(r x F)/dc = [F]x
            A[6 * 9 + 1] += -dt * F[2];
            A[6 * 9 + 2] += dt * F[1];
            A[7 * 9 + 0] += dt * F[2];
            A[7 * 9 + 2] += -dt * F[0];
            A[8 * 9 + 0] += -dt * F[1];
            A[8 * 9 + 1] += dt * F[0];
          }
          if (Bm)
          {
            for (int i = 0; i < 3; i++)
              Bm[(3 + i) * nu + 3 * f + i] = dt;
            // d(r x F)/dF = [r]x
            Bm[6 * nu + 3 * f + 1] = -dt * r[2];
            Bm[6 * nu + 3 * f + 2] = dt * r[1];
            Bm[7 * nu + 3 * f + 0] = dt * r[2];
            Bm[7 * nu + 3 * f + 2] = -dt * r[0];
            Bm[8 * nu + 3 * f + 0] = -dt * r[1];
            Bm[8 * nu + 3 * f + 1] = dt * r[0];
          }
        }
        double * xn = ka.Xn + (size_t)inst * 9;
        for (int i = 0; i < 3; i++)
        {
          xn[i] = x[i] + dt * x[3 + i] * im;
          xn[3 + i] = x[3 + i] + dt * hd[i];
          xn[6 + i] = x[6 + i] + dt * Ld[i];
        }
      }
    }
    SMPC_LANES_END
  }

  // torque += viscous * v + dry * sign(v), elementwise over [B][nu] (reference src/friction-compensation.cpp:22-37)
  struct FrictionArgs
  {
    const double *dry, *viscous, *velocity; // [nu], [nu], [B][nu] (device)
    double * torque;                        // [B][nu] in / out (device)
    int nu;
    size_t total; // B * nu
  };
  SMPC_DEV void friction_body(const FrictionArgs & ka, int block)
  {
    constexpr int NT = 256;
    SMPC_LANES(NT)
    {
      const size_t i = (size_t)block * NT + lane;
      if (i < ka.total)
      {
        const int j = (int)(i % (size_t)ka.nu);
        const double v = ka.velocity[i];
        const double sgn = (double)((v > 0.0) - (v < 0.0));
        ka.torque[i] += ka.viscous[j] * v + ka.dry[j] * sgn;
      }
    }
    SMPC_LANES_END
  }

  // gather one horizon node t of every instance out of the ring into a dense [B][NX] device buffer
  template <class D>
  struct GatherArgs
  {
    Buffers<D> b;
    int head, t;
    double * out;
  };
  template <class D>
  SMPC_DEV void gather_x_body(const GatherArgs<D> & ka, int block)
  {
    constexpr int NT = 256;
    const Buffers<D> & b = ka.b;
    const int st = ring_slot(ka.head, ka.t, b.R);
    SMPC_LANES(NT)
    {
      const size_t i = (size_t)block * NT + lane;
      if (i < (size_t)b.B * D::NX)
      {
        const size_t inst = i / D::NX, k = i % D::NX;
        ka.out[i] = b.xs[(inst * b.R + st) * D::NX + k];
      }
    }
    SMPC_LANES_END
  }

  // copy current multipliers into the AL centres (start of SolverProxDDP::run: prev_vs / prev_lams)
  template <class D>
  SMPC_DEV void centres_body(const SolverArgs<D> & ka, int block)
  {
    constexpr int NT = 256;
    const Buffers<D> & b = ka.b;
    const size_t n_v = (size_t)b.B * b.R * D::NC, n_l = (size_t)b.B * b.R * D::NDX;
    SMPC_LANES(NT)
    {
      const size_t i = (size_t)block * NT + lane;
      if (i < n_v)
        b.vs_e[i] = b.vs[i];
      if (i < n_l)
        b.lams_e[i] = b.lams[i];
    }
    SMPC_LANES_END
  }
} // namespace smpc
