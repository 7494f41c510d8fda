// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_proxddp.hpp: one ProxDDP iteration as driven by the reference through
// `solver_->run(problem, xs_, us_)` (src/mpc.cpp:212; solver settings src/mpc.cpp:43-53,91):
// LINEAR rollout, SERIAL proximal Riccati, force_initial_condition = true, mu = mu_init.
// The algorithm itself lives in Aligator 0.16.0 (un-vendored); this restates it from the papers
// (Jallet et al., "PROXDDP" 2023/25; "Parallel and Proximal Constrained LQR" 2024) -- SURVEY.md
// Appendix B.4/B.5.  Constants that could not be checked upstream are chosen here and documented
// in DESIGN.md ("solver constants").
//
// Per iteration, with centres (lam_e, nu_e) = multipliers at the start of the control step:
//   1. evaluate stages: e_{t+1} = f(x_t,u_t) (-) x_{t+1}, cost, c_t
//   2. lam+ = lam_e + e/mu ;  z = c + mu nu_e ;  nu+ = (z - Proj_C(z))/mu ; active rows
//      merit phi = sum cost + sum mu/2 (|lam+|^2 + |lam+ - lam|^2) + sum mu/2 (|nu+|^2 + |nu+ - nu|^2)
//   3. derivatives; Lagrangian gradients with the current (lam, nu)
//   4. LQ knot: Q,S,R (Gauss-Newton + preg I), q = Lx, r = Lu, A, B, f = mu (lam+ - lam),
//      C,D = active rows, d = mu (nu+ - nu)
//   5. proximal Riccati backward / forward  -> (dx, du, dnu, dlam)
//   6. backtracking Armijo line search on phi over alpha in {1, 1/2, ..., 2^-(LS_N-1)}
#pragma once
#include "orc_cent.hpp"

namespace orc
{
  struct SolverConsts
  {
    static constexpr int LS_N = 10;          // candidate step sizes 2^0 .. 2^-9
    static constexpr double ARMIJO_C1 = 1e-4;
    static constexpr double REG_INIT = 1e-9, REG_MIN = 1e-10, REG_MAX = 1e9;
    static constexpr double REG_INC = 10.0, REG_DEC = 1.0 / 3.0;
    static constexpr double STALL_REL = 1e-9; // cold solve: stop when |dphi0| <= STALL_REL max(1,|phi0|)
  };

  struct Knot
  {
    Mat Q, S, R, A, B, C, D;
    Vec q, r, f, d;
  };
  struct Gains
  {
    Mat K, Z, Pt; // K (nu x ndx), Z (nc x ndx), Pt = P~_{t+1} (ndx x ndx)
    Vec k, z, pnext;
  };

  // Proximal Riccati (derivation in DESIGN.md).
  // Solves, for t = 0..H-1 (dx_0 = 0):
  //   Q dx + S du + A^T dl+ + C^T dnu - dl + q = 0 ;  S^T dx + R du + B^T dl+ + D^T dnu + r = 0
  //   A dx + B du - dx+ + f - mu dl+ = 0 ;             C dx + D du + d - mu dnu = 0
  //   Q_N dx_N - dl_N + q_N = 0
  inline bool & fold_u_rows()
  {
    static bool on = false;
    return on;
  }

  inline void prox_riccati(
    const std::vector<Knot> & kn, const Mat & QN, const Vec & qN, double mu, std::vector<Vec> & dxs,
    std::vector<Vec> & dus, std::vector<Vec> & dvs, std::vector<Vec> & dlams, std::vector<Mat> * K_out = nullptr)
  {
    const int H = (int)kn.size();
    const int ndx = QN.r;
    std::vector<Gains> G(H);
    Mat P = QN;
    Vec p = qN;
    for (int t = H - 1; t >= 0; t--)
    {
      const Knot & k = kn[t];
      const int nu = k.R.r, nc = k.C.r;
      // P~ = (I + mu P)^-1 P ; p~ = (I + mu P)^-1 (p + P f)
      Mat Mm = P;
      for (auto & e : Mm.a)
        e *= mu;
      for (int i = 0; i < ndx; i++)
        Mm(i, i) += 1.0;
      bool ok = cholesky(Mm);
      assert(ok);
      (void)ok;
      Mat Pt = P;
      chol_solve_inplace(Mm, Pt);
      // symmetrise
      for (int i = 0; i < ndx; i++)
        for (int j = 0; j < i; j++)
        {
          double s = 0.5 * (Pt(i, j) + Pt(j, i));
          Pt(i, j) = Pt(j, i) = s;
        }
      Vec pt = p;
      axpy(pt, mul(P, k.f));
      chol_solve_inplace(Mm, pt);
      Mat TA = mul(Pt, k.A), TB = mul(Pt, k.B);
      Mat Qh = k.Q, Sh = k.S, Rh = k.R;
      add_inplace(Qh, mulTN(k.A, TA));
      add_inplace(Sh, mulTN(k.A, TB));
      add_inplace(Rh, mulTN(k.B, TB));
      Vec qh = k.q, rh = k.r;
      axpy(qh, mulT(k.A, pt));
      axpy(rh, mulT(k.B, pt));
      // (numerical experiment, off by default: eliminate the rows that act on u only -- R^ += D^T D / mu, r^ += D^T d / mu -- the
      //  way a kernel that folds them would; fold_u_rows() is set by the test that measures what the fold costs in accuracy)
      Mat Dk = k.D;
      std::vector<char> folded(nc, 0);
      if (fold_u_rows())
        for (int i = 0; i < nc; i++)
        {
          bool hasD = false, hasC = false;
          for (int j = 0; j < nu; j++)
            hasD = hasD || k.D(i, j) != 0.0;
          for (int j = 0; j < ndx; j++)
            hasC = hasC || k.C(i, j) != 0.0;
          if (!hasD || hasC)
            continue;
          folded[i] = 1;
          for (int a = 0; a < nu; a++)
          {
            rh[a] += k.D(i, a) * k.d[i] / mu;
            for (int bb = 0; bb < nu; bb++)
              Rh(a, bb) += k.D(i, a) * k.D(i, bb) / mu;
            Dk(i, a) = 0.0;
          }
        }
      // stage KKT: [Rh D^T; D -mu I] [K k; Z z] = -[Sh^T rh; C d]
      Mat LR = Rh;
      ok = cholesky(LR);
      assert(ok);
      Mat W(nu, ndx + 1); // L^-1 [Sh^T rh]
      for (int i = 0; i < nu; i++)
      {
        for (int j = 0; j < ndx; j++)
          W(i, j) = Sh(j, i);
        W(i, ndx) = rh[i];
      }
      for (int j = 0; j <= ndx; j++)
        solve_L(LR, &W.a[j], ndx + 1);
      Mat Y(nu, nc); // L^-1 D^T
      for (int i = 0; i < nu; i++)
        for (int j = 0; j < nc; j++)
          Y(i, j) = Dk(j, i);
      for (int j = 0; j < nc; j++)
        solve_L(LR, &Y.a[j], nc);
      Mat Sc = mulTN(Y, Y);
      for (int i = 0; i < nc; i++)
        Sc(i, i) += mu;
      ok = cholesky(Sc);
      assert(ok);
      Mat Zz(nc, ndx + 1);
      Mat YtW = mulTN(Y, W);
      for (int i = 0; i < nc; i++)
      {
        for (int j = 0; j < ndx; j++)
          Zz(i, j) = k.C(i, j) - YtW(i, j);
        Zz(i, ndx) = k.d[i] - YtW(i, ndx);
      }
      chol_solve_inplace(Sc, Zz);
      Mat Kk = W;
      add_inplace(Kk, mul(Y, Zz));
      for (int j = 0; j <= ndx; j++)
        solve_LT(LR, &Kk.a[j], ndx + 1);
      for (auto & e : Kk.a)
        e = -e;
      Gains & g = G[t];
      g.K.resize(nu, ndx);
      g.k.assign(nu, 0.0);
      g.Z.resize(nc, ndx);
      g.z.assign(nc, 0.0);
      for (int i = 0; i < nu; i++)
      {
        for (int j = 0; j < ndx; j++)
          g.K(i, j) = Kk(i, j);
        g.k[i] = Kk(i, ndx);
      }
      for (int i = 0; i < nc; i++)
      {
        for (int j = 0; j < ndx; j++)
          g.Z(i, j) = Zz(i, j);
        g.z[i] = Zz(i, ndx);
        if (folded[i])
        { // dv = (D du + d) / mu
          for (int j = 0; j < ndx; j++)
          {
            double acc = 0.0;
            for (int a = 0; a < nu; a++)
              acc += k.D(i, a) * g.K(a, j);
            g.Z(i, j) = acc / mu;
          }
          double acc = k.d[i];
          for (int a = 0; a < nu; a++)
            acc += k.D(i, a) * g.k[a];
          g.z[i] = acc / mu;
        }
      }
      g.Pt = Pt;
      g.pnext = p;
      // P_t = Qh + Sh K + C^T Z ; p_t = qh + Sh k + C^T z
      Mat Pn = Qh;
      add_inplace(Pn, mul(Sh, g.K));
      add_inplace(Pn, mulTN(k.C, g.Z));
      for (int i = 0; i < ndx; i++)
        for (int j = 0; j < i; j++)
        {
          double s = 0.5 * (Pn(i, j) + Pn(j, i));
          Pn(i, j) = Pn(j, i) = s;
        }
      Vec pn = qh;
      axpy(pn, mul(Sh, g.k));
      axpy(pn, mulT(k.C, g.z));
      P = Pn;
      p = pn;
    }
    // forward
    dxs.assign(H + 1, Vec(ndx, 0.0));
    dus.resize(H);
    dvs.resize(H);
    dlams.assign(H + 1, Vec(ndx, 0.0));
    for (int t = 0; t < H; t++)
    {
      const Knot & k = kn[t];
      const Gains & g = G[t];
      Vec du = g.k;
      axpy(du, mul(g.K, dxs[t]));
      Vec dv = g.z;
      axpy(dv, mul(g.Z, dxs[t]));
      Vec y = mul(k.A, dxs[t]);
      axpy(y, mul(k.B, du));
      axpy(y, k.f);
      axpy(y, g.pnext, -mu);
      Vec w = mul(g.Pt, y);
      Vec dxn = y;
      axpy(dxn, w, -mu);
      Vec dl = w;
      axpy(dl, g.pnext);
      dus[t] = du;
      dvs[t] = dv;
      dxs[t + 1] = dxn;
      dlams[t + 1] = dl;
    }
    if (K_out)
    {
      K_out->resize(H);
      for (int t = 0; t < H; t++)
        (*K_out)[t] = G[t].K;
    }
  }

  struct OcpInstance
  {
    std::vector<StageRef> stages; // H
    Vec x_tgt_term;               // terminal state_cost target
    // terminal equality constraint DCMPositionResidual(com_ref, tau): com + tau vcom = dcm_ref (createProblem(..., terminal_constraint = true):
    // reference src/ocp-handler.cpp:133-136, src/kinodynamics.cpp:366-377)
    bool term_cstr = false;
    double dcm_ref[3] = {0, 0, 0};
    double dcm_tau = 0;
  };
  struct SolverState
  {
    std::vector<Vec> xs, us, vs, lams; // lams[0] unused (initial condition is forced)
    Vec vN;                            // multipliers of the terminal constraint (3 when the problem has one)
    double preg = SolverConsts::REG_INIT;
    std::vector<Mat> Ks;               // feedback gains of the last iteration
    std::vector<Vec> xdot;             // continuous xdot per stage at the accepted point
  };
  struct IterInfo
  {
    double phi0 = 0, dphi0 = 0, alpha = 0, phi_new = 0, prim_infeas = 0, dual_infeas = 0, cost = 0, prim_new = 0, cost_new = 0;
    int ls_index = 0, ls_failed = 0;
  };

  // Model: KinoModel (orc_kino.hpp) or CentModel (orc_cent.hpp)
  template <class Model>
  struct ProxDDPT
  {
    const Model & md;
    double mu;
    ProxDDPT(const Model & m, double mu_) : md(m), mu(mu_) {}

    struct Eval
    {
      std::vector<StageEval> ev;
      std::vector<Vec> e;      // defects, index t+1
      std::vector<Vec> lam_plus, v_plus;
      std::vector<std::vector<char>> active;
      Vec vN_plus, cN;
      double cost = 0, phi = 0, prim = 0;
    };

    // steps 1-2 at (xs, us, vs, lams) with centres (vs_e, lams_e)
    void evaluate(
      Rigid & R, const OcpInstance & ocp, const std::vector<Vec> & xs, const std::vector<Vec> & us,
      const std::vector<Vec> & vs, const std::vector<Vec> & lams, const std::vector<Vec> & vs_e,
      const std::vector<Vec> & lams_e, Eval & E, const Vec * vN = nullptr, const Vec * vN_e = nullptr) const
    {
      const int H = (int)ocp.stages.size();
      E.ev.resize(H);
      E.e.assign(H + 1, Vec(md.ndx, 0.0));
      E.lam_plus.assign(H + 1, Vec(md.ndx, 0.0));
      E.v_plus.assign(H, Vec(md.nc, 0.0));
      E.active.assign(H, std::vector<char>(md.nc, 0));
      double cost = 0, pen = 0, prim = 0;
      for (int t = 0; t < H; t++)
      {
        md.eval(R, ocp.stages[t], xs[t].data(), us[t].data(), E.ev[t]);
        cost += E.ev[t].cost;
        md.difference(xs[t + 1].data(), E.ev[t].xnext.data(), E.e[t + 1].data());
        for (int i = 0; i < md.ndx; i++)
        {
          const double lp = lams_e[t + 1][i] + E.e[t + 1][i] / mu;
          E.lam_plus[t + 1][i] = lp;
          const double dl = lp - lams[t + 1][i];
          pen += 0.5 * mu * (lp * lp + dl * dl);
          prim = std::fmax(prim, std::fabs(E.e[t + 1][i]));
        }
        for (int i = 0; i < md.nc; i++)
        {
          const int kind = md.row_kind(ocp.stages[t], i);
          double vp = 0;
          char act = 0;
          if (kind != ROW_ABSENT)
          {
            const double c = E.ev[t].c[i];
            const double z = c + mu * vs_e[t][i];
            double proj = z;
            if (kind == ROW_EQ)
              proj = 0.0;
            else if (kind == ROW_NEG)
              proj = std::fmin(z, 0.0);
            else if (kind == ROW_BOX)
              proj = std::fmin(std::fmax(z, md.row_lo_v(i)), md.row_hi_v(i));
            vp = (z - proj) / mu;
            act = (z != proj) || kind == ROW_EQ;
            double viol = 0;
            if (kind == ROW_EQ)
              viol = std::fabs(c);
            else if (kind == ROW_NEG)
              viol = std::fmax(c, 0.0);
            else
              viol = std::fmax(std::fmax(c - md.row_hi_v(i), md.row_lo_v(i) - c), 0.0);
            prim = std::fmax(prim, viol);
          }
          E.v_plus[t][i] = vp;
          E.active[t][i] = act;
          const double dv = vp - vs[t][i];
          pen += 0.5 * mu * (vp * vp + dv * dv);
        }
      }
      cost += md.term_eval(R, ocp.x_tgt_term, xs[H].data());
      if (ocp.term_cstr)
      {
        E.cN.assign(3, 0.0);
        E.vN_plus.assign(3, 0.0);
        md.term_cstr(R, xs[H].data(), ocp.dcm_ref, ocp.dcm_tau, E.cN.data(), nullptr);
        for (int i = 0; i < 3; i++)
        {
          const double vp = (*vN_e)[i] + E.cN[i] / mu; // equality row: always active
          E.vN_plus[i] = vp;
          const double dv = vp - (*vN)[i];
          pen += 0.5 * mu * (vp * vp + dv * dv);
          prim = std::fmax(prim, std::fabs(E.cN[i]));
        }
      }
      E.cost = cost;
      E.phi = cost + pen;
      E.prim = prim;
    }

    // One full iteration; vs_e / lams_e are the AL centres.
    IterInfo iterate(
      Rigid & R, const OcpInstance & ocp, SolverState & S, const std::vector<Vec> & vs_e,
      const std::vector<Vec> & lams_e, std::vector<Knot> * knots_out = nullptr, const Vec * vN_e = nullptr, double stop_tol = -1.0) const
    {
      const int H = (int)ocp.stages.size();
      const int ndx = md.ndx, nu = md.nu, nc = md.nc;
      IterInfo info;
      Eval E0;
      evaluate(R, ocp, S.xs, S.us, S.vs, S.lams, vs_e, lams_e, E0, &S.vN, vN_e);
      info.phi0 = E0.phi;
      info.cost = E0.cost;
      info.prim_infeas = E0.prim;
      // derivatives + knots
      std::vector<StageDer> der(H);
      std::vector<Knot> kn(H);
      double dual = 0;
      for (int t = 0; t < H; t++)
      {
        md.deriv(R, ocp.stages[t], S.xs[t].data(), S.us[t].data(), der[t]);
        Knot & k = kn[t];
        k.Q = der[t].Lxx;
        k.S = der[t].Lxu;
        k.R = der[t].Luu;
        for (int i = 0; i < ndx; i++)
          k.Q(i, i) += S.preg;
        for (int i = 0; i < nu; i++)
          k.R(i, i) += S.preg;
        k.A = der[t].A;
        k.B = der[t].B;
        k.q = der[t].lx;
        axpy(k.q, mulT(der[t].A, S.lams[t + 1]));
        axpy(k.q, mulT(der[t].Cx, S.vs[t]));
        if (t >= 1)
          axpy(k.q, S.lams[t], -1.0);
        else
          std::fill(k.q.begin(), k.q.end(), 0.0); // x_0 is fixed (force_initial_condition)
        k.r = der[t].lu;
        axpy(k.r, mulT(der[t].B, S.lams[t + 1]));
        axpy(k.r, mulT(der[t].Cu, S.vs[t]));
        k.f.assign(ndx, 0.0);
        for (int i = 0; i < ndx; i++)
          k.f[i] = mu * (E0.lam_plus[t + 1][i] - S.lams[t + 1][i]);
        k.C.resize(nc, ndx);
        k.D.resize(nc, nu);
        k.d.assign(nc, 0.0);
        for (int i = 0; i < nc; i++)
        {
          if (E0.active[t][i])
          {
            for (int j = 0; j < ndx; j++)
              k.C(i, j) = der[t].Cx(i, j);
            for (int j = 0; j < nu; j++)
              k.D(i, j) = der[t].Cu(i, j);
          }
          k.d[i] = mu * (E0.v_plus[t][i] - S.vs[t][i]);
        }
        dual = std::fmax(dual, std::fmax(norm_inf(k.q), norm_inf(k.r)));
      }
      Vec lxN;
      Mat LxxN;
      md.term_deriv(R, ocp.x_tgt_term, S.xs[H].data(), lxN, LxxN);
      Vec qN = lxN;
      axpy(qN, S.lams[H], -1.0);
      Mat QN = LxxN;
      for (int i = 0; i < ndx; i++)
        QN(i, i) += S.preg;
      // terminal constraint rows C dx - mu dv + mu (v+ - v) = 0, eliminated: QN += C^T C / mu, qN + C^T v += C^T (v+ - v)
      Mat CN;
      Vec dN(3, 0.0);
      if (ocp.term_cstr)
      {
        Vec c3(3);
        md.term_cstr(R, S.xs[H].data(), ocp.dcm_ref, ocp.dcm_tau, c3.data(), &CN);
        axpy(qN, mulT(CN, S.vN));
        for (int i = 0; i < 3; i++)
          dN[i] = mu * (E0.vN_plus[i] - S.vN[i]);
      }
      dual = std::fmax(dual, norm_inf(qN));
      info.dual_infeas = dual;
      if (stop_tol >= 0.0 && std::fmax(info.prim_infeas, dual) <= stop_tol)
      { // SolverProxDDP::run's convergence test (reference src/mpc.cpp:43,212): converged, no step
        if (knots_out)
          *knots_out = kn;
        info.phi_new = info.phi0;
        info.prim_new = info.prim_infeas;
        return info;
      }
      if (ocp.term_cstr)
      {
        for (int r = 0; r < 3; r++)
          for (int i = 0; i < ndx; i++)
          {
            qN[i] += CN(r, i) * dN[r] / mu;
            for (int j = 0; j < ndx; j++)
              QN(i, j) += CN(r, i) * CN(r, j) / mu;
          }
      }
      if (knots_out)
        *knots_out = kn;

      std::vector<Vec> dxs, dus, dvs, dlams;
      prox_riccati(kn, QN, qN, mu, dxs, dus, dvs, dlams, &S.Ks);
      Vec dvN(3, 0.0);
      if (ocp.term_cstr)
        for (int r = 0; r < 3; r++)
        {
          double acc = dN[r];
          for (int i = 0; i < ndx; i++)
            acc += CN(r, i) * dxs[H][i];
          dvN[r] = acc / mu;
        }

      // directional derivative of the merit
      double dphi = 0;
      for (int t = 0; t < H; t++)
      {
        Vec lpd(ndx), lpd_t(ndx, 0.0), vpd(nc, 0.0);
        for (int i = 0; i < ndx; i++)
          lpd[i] = 2.0 * E0.lam_plus[t + 1][i] - S.lams[t + 1][i];
        if (t >= 1)
          for (int i = 0; i < ndx; i++)
            lpd_t[i] = 2.0 * E0.lam_plus[t][i] - S.lams[t][i];
        for (int i = 0; i < nc; i++)
          if (E0.active[t][i])
            vpd[i] = 2.0 * E0.v_plus[t][i] - S.vs[t][i];
        Vec gx = der[t].lx;
        axpy(gx, mulT(der[t].A, lpd));
        axpy(gx, lpd_t, -1.0);
        axpy(gx, mulT(der[t].Cx, vpd));
        Vec gu = der[t].lu;
        axpy(gu, mulT(der[t].B, lpd));
        axpy(gu, mulT(der[t].Cu, vpd));
        dphi += dot(gx, dxs[t]) + dot(gu, dus[t]);
        for (int i = 0; i < ndx; i++)
          dphi -= mu * (E0.lam_plus[t + 1][i] - S.lams[t + 1][i]) * dlams[t + 1][i];
        for (int i = 0; i < nc; i++)
          dphi -= mu * (E0.v_plus[t][i] - S.vs[t][i]) * dvs[t][i];
      }
      {
        Vec gx = lxN;
        for (int i = 0; i < ndx; i++)
          gx[i] -= 2.0 * E0.lam_plus[H][i] - S.lams[H][i];
        if (ocp.term_cstr)
          for (int r = 0; r < 3; r++)
          {
            const double vpd = 2.0 * E0.vN_plus[r] - S.vN[r];
            for (int i = 0; i < ndx; i++)
              gx[i] += CN(r, i) * vpd;
            dphi -= mu * (E0.vN_plus[r] - S.vN[r]) * dvN[r];
          }
        dphi += dot(gx, dxs[H]);
      }
      info.dphi0 = dphi;

      // line search
      std::vector<Vec> txs(H + 1), tus(H), tvs(H), tl(H + 1);
      Vec tvN = S.vN;
      Eval Et;
      double alpha = 1.0;
      int accepted = -1;
      for (int j = 0; j < SolverConsts::LS_N; j++)
      {
        for (int t = 0; t <= H; t++)
        {
          Vec adx = dxs[t];
          for (auto & e : adx)
            e *= alpha;
          txs[t].assign(md.nx, 0.0);
          md.integrate(S.xs[t].data(), adx.data(), txs[t].data());
          tl[t] = S.lams[t];
          axpy(tl[t], dlams[t], alpha);
        }
        for (int t = 0; t < H; t++)
        {
          tus[t] = S.us[t];
          axpy(tus[t], dus[t], alpha);
          tvs[t] = S.vs[t];
          axpy(tvs[t], dvs[t], alpha);
        }
        if (ocp.term_cstr)
          for (int r = 0; r < 3; r++)
            tvN[r] = S.vN[r] + alpha * dvN[r];
        evaluate(R, ocp, txs, tus, tvs, tl, vs_e, lams_e, Et, &tvN, vN_e);
        info.ls_index = j;
        if (Et.phi <= info.phi0 + SolverConsts::ARMIJO_C1 * alpha * dphi)
        {
          accepted = j;
          break;
        }
        if (j + 1 < SolverConsts::LS_N)
          alpha *= 0.5;
      }
      info.alpha = alpha;
      info.phi_new = Et.phi;
      info.prim_new = Et.prim;
      info.cost_new = Et.cost;
      info.ls_failed = accepted < 0;
      S.xs = txs;
      S.us = tus;
      S.vs = tvs;
      S.lams = tl;
      S.vN = tvN;
      S.xdot.resize(H);
      for (int t = 0; t < H; t++)
        S.xdot[t] = Et.ev[t].xdot;
      if (accepted < 0)
        S.preg = std::fmin(S.preg * SolverConsts::REG_INC, SolverConsts::REG_MAX);
      else
        S.preg = std::fmax(S.preg * SolverConsts::REG_DEC, SolverConsts::REG_MIN);
      return info;
    }
  };
  typedef ProxDDPT<KinoModel> ProxDDP;
} // namespace orc
