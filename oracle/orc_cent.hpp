// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_cent.hpp: the centroidal OCP of the reference (BASELINE config "Go2 centroidal, H=50"), restated:
//   CentroidalOCP::createStage         src/centroidal-dynamics.cpp:39-106   (6 cost components, one cone block per
//                                                                            foot in contact)
//   CentroidalOCP::createTerminalCost  src/centroidal-dynamics.cpp:306-316  (linear + angular momentum)
//   setReferencePose / setVelocityBase / setPoseBase / setReferenceState
//                                      src/centroidal-dynamics.cpp:151-169, 227-239, 249-257, 286-291
// The residual / dynamics classes are Aligator 0.16.0 (un-vendored, aligator/modelling/centroidal/*,
// aligator/modelling/dynamics/centroidal-fwd.hpp); restated from their published definitions:
//   state x = [c; h; L]   (CoM, linear momentum, angular momentum about the CoM),  u = [f_1 .. f_nf] (3-D forces)
//   xdot = [h / m ;  m g + sum_{contact} f_i ;  sum_{contact} (p_i - c) x f_i],   x+ = x + dt xdot   (IntegratorEuler)
//   residuals: com c - c_ref | control u - u_ref | linear_mom h - h_ref | angular_mom L - L_ref |
//              linear_acc g + sum f_i / m | angular_acc sum (p_i - c) x f_i        (Gauss-Newton Hessians)
//   constraint per foot in contact (CentroidalFrictionConeResidual, set = NegativeOrthant), epsilon = 1e-4:
//              [ -f_z + epsilon ;  f_x^2 + f_y^2 - mu^2 f_z^2 ] <= 0
// 6-D feet (force_size 6; the Talos configuration of examples/talos_centroidal.py, tests/test_utils.cpp:199-218): u = [(f_i, tau_i) per
//   foot]; the contact torques add to the angular momentum rate and to the angular_acc residual ([UPSTREAM-RECALL] aligator
//   centroidal-fwd.hxx / angular-acceleration.hxx: xdot.tail<3>() += u.segment(i * 6 + 3, 3)); constraint per foot in contact:
//   CentroidalWrenchConeResidual (src/centroidal-dynamics.cpp:90-95), 17 constant linear rows A_cone(mu, L, W) u_i <= 0.
// Choice where the upstream scaling could not be checked: linear_acc is the CoM acceleration
// g + sum f / m (not the force balance m g + sum f).
#pragma once
#include "orc_kino.hpp"

namespace orc
{
  struct CentSettings // include/simple-mpc/centroidal-dynamics.hpp:27-43
  {
    double timestep = 0.01;
    Mat w_u, w_com, w_linear_mom, w_angular_mom, w_linear_acc, w_angular_acc;
    double gravity[3] = {0, 0, -9.81};
    double mu = 0.8, Lfoot = 0.1, Wfoot = 0.075;
    int force_size = 3;
  };

  struct CentModel
  {
    static constexpr double CONE_EPS = 1e-4; // src/centroidal-dynamics.cpp:96
    const smpc_robot_model * M;
    double mass;
    CentSettings s;
    int nq = 9, nv = 9, nx = 9, ndx = 9, nu, nf, nc, fs = 3, nc1 = 2;
    Mat Acone;

    CentModel(const smpc_robot_model * m, const CentSettings & st) : M(m), mass(m->total_mass), s(st)
    {
      nf = m->nfeet;
      fs = st.force_size;
      nc1 = fs == 6 ? 17 : 2;
      nu = fs * nf;
      nc = nc1 * nf; // rows nc1 f ..: cone block of foot f (present while the foot is in contact): friction cone (2) / wrench cone (17)
      Acone = wrench_cone_matrix(st.mu, st.Lfoot, st.Wfoot);
    }
    int row_kind(const StageRef & r, int row) const { return ((r.mask >> (row / nc1)) & 1u) ? ROW_NEG : ROW_ABSENT; }
    int force_ref_index(int f) const { return fs * f; }
    double row_lo_v(int) const { return 0.0; }
    double row_hi_v(int) const { return 0.0; }
    void integrate(const double * x, const double * dx, double * out) const
    {
      for (int i = 0; i < 9; i++)
        out[i] = x[i] + dx[i];
    }
    void difference(const double * x0, const double * x1, double * out) const
    {
      for (int i = 0; i < 9; i++)
        out[i] = x1[i] - x0[i];
    }
    static double quad3(const Mat & W, const V3 & r)
    {
      double sacc = 0;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          sacc += r[i] * W(i, j) * r[j];
      return 0.5 * sacc;
    }
    static Mat skew(const V3 & a)
    {
      Mat S(3, 3);
      S(0, 1) = -a[2];
      S(0, 2) = a[1];
      S(1, 0) = a[2];
      S(1, 2) = -a[0];
      S(2, 0) = -a[1];
      S(2, 1) = a[0];
      return S;
    }

    void eval(Rigid &, const StageRef & r, const double * x, const double * u, StageEval & o) const
    {
      const double dt = s.timestep;
      const V3 c = v3(x[0], x[1], x[2]), h = v3(x[3], x[4], x[5]), L = v3(x[6], x[7], x[8]);
      const V3 g = v3(s.gravity[0], s.gravity[1], s.gravity[2]);
      V3 fsum = v3(0, 0, 0), tsum = v3(0, 0, 0);
      for (int f = 0; f < nf; f++)
        if ((r.mask >> f) & 1u)
        {
          const V3 F = v3(u[fs * f], u[fs * f + 1], u[fs * f + 2]);
          fsum = fsum + F;
          tsum = tsum + cross(r.foot_ref[f] - c, F);
          if (fs == 6)
            tsum = tsum + v3(u[6 * f + 3], u[6 * f + 4], u[6 * f + 5]);
        }
      o.xdot.assign(9, 0.0);
      for (int i = 0; i < 3; i++)
      {
        o.xdot[i] = h[i] / mass;
        o.xdot[3 + i] = mass * g[i] + fsum[i];
        o.xdot[6 + i] = tsum[i];
      }
      o.xnext.assign(9, 0.0);
      for (int i = 0; i < 9; i++)
        o.xnext[i] = x[i] + dt * o.xdot[i];
      double cost = 0;
      cost += quad3(s.w_com, c - v3(r.x_tgt[0], r.x_tgt[1], r.x_tgt[2]));
      Vec ru(nu);
      for (int i = 0; i < nu; i++)
        ru[i] = u[i] - r.u_ref[i];
      cost += 0.5 * dot(ru, mul(s.w_u, ru));
      cost += quad3(s.w_linear_mom, h - v3(r.x_tgt[3], r.x_tgt[4], r.x_tgt[5]));
      cost += quad3(s.w_angular_mom, L - v3(r.x_tgt[6], r.x_tgt[7], r.x_tgt[8]));
      cost += quad3(s.w_linear_acc, g + (1.0 / mass) * fsum);
      cost += quad3(s.w_angular_acc, tsum);
      o.cost = cost;
      o.c.assign(nc, 0.0);
      for (int f = 0; f < nf; f++)
        if ((r.mask >> f) & 1u)
        {
          if (fs == 6)
          {
            for (int i = 0; i < 17; i++)
            {
              double acc = 0.0;
              for (int j = 0; j < 6; j++)
                acc += Acone(i, j) * u[6 * f + j];
              o.c[17 * f + i] = acc;
            }
            continue;
          }
          const double fx = u[3 * f], fy = u[3 * f + 1], fz = u[3 * f + 2];
          o.c[2 * f] = -fz + CONE_EPS;
          o.c[2 * f + 1] = fx * fx + fy * fy - s.mu * s.mu * fz * fz;
        }
    }

    void deriv(Rigid &, const StageRef & r, const double * x, const double * u, StageDer & o) const
    {
      const double dt = s.timestep;
      const V3 c = v3(x[0], x[1], x[2]), h = v3(x[3], x[4], x[5]), L = v3(x[6], x[7], x[8]);
      const V3 g = v3(s.gravity[0], s.gravity[1], s.gravity[2]);
      o.A = Mat::identity(9);
      o.B.resize(9, nu);
      V3 fsum = v3(0, 0, 0), tsum = v3(0, 0, 0);
      Mat Jaa_c(3, 3);      // d angular_acc / dc = sum [f]x
      Mat Jaa_u(3, nu);     // d angular_acc / du = [p - c]x per foot in contact
      Mat Jla_u(3, nu);     // d linear_acc / du = I / m per foot in contact
      for (int i = 0; i < 3; i++)
        o.A(i, 3 + i) = dt / mass;
      for (int f = 0; f < nf; f++)
        if ((r.mask >> f) & 1u)
        {
          const V3 F = v3(u[fs * f], u[fs * f + 1], u[fs * f + 2]);
          const V3 rr = r.foot_ref[f] - c;
          fsum = fsum + F;
          tsum = tsum + cross(rr, F);
          if (fs == 6)
            tsum = tsum + v3(u[6 * f + 3], u[6 * f + 4], u[6 * f + 5]);
          const Mat Fx = skew(F), Rx = skew(rr);
          for (int i = 0; i < 3; i++)
          {
            for (int j = 0; j < 3; j++)
            {
              Jaa_c(i, j) += Fx(i, j);
              Jaa_u(i, fs * f + j) = Rx(i, j);
              o.B(6 + i, fs * f + j) = dt * Rx(i, j);
            }
            Jla_u(i, fs * f + i) = 1.0 / mass;
            o.B(3 + i, fs * f + i) = dt;
            if (fs == 6)
            { // contact torque
              Jaa_u(i, 6 * f + 3 + i) = 1.0;
              o.B(6 + i, 6 * f + 3 + i) = dt;
            }
          }
        }
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
          o.A(6 + i, j) = dt * Jaa_c(i, j);
      o.lx.assign(9, 0.0);
      o.lu.assign(nu, 0.0);
      o.Lxx.resize(9, 9);
      o.Lxu.resize(9, nu);
      o.Luu.resize(nu, nu);
      auto vec3 = [](const V3 & a) { return Vec{a[0], a[1], a[2]}; };
      // blocks of x: residual on block b (0 com, 1 h, 2 L) with weight W
      auto add_xblock = [&](const Mat & W, const V3 & res, int b) {
        Vec Wr = mul(W, vec3(res));
        for (int i = 0; i < 3; i++)
        {
          o.lx[3 * b + i] += Wr[i];
          for (int j = 0; j < 3; j++)
            o.Lxx(3 * b + i, 3 * b + j) += W(i, j);
        }
      };
      add_xblock(s.w_com, c - v3(r.x_tgt[0], r.x_tgt[1], r.x_tgt[2]), 0);
      add_xblock(s.w_linear_mom, h - v3(r.x_tgt[3], r.x_tgt[4], r.x_tgt[5]), 1);
      add_xblock(s.w_angular_mom, L - v3(r.x_tgt[6], r.x_tgt[7], r.x_tgt[8]), 2);
      { // control
        Vec ru(nu);
        for (int i = 0; i < nu; i++)
          ru[i] = u[i] - r.u_ref[i];
        axpy(o.lu, mul(s.w_u, ru));
        add_inplace(o.Luu, s.w_u);
      }
      { // linear acceleration
        Vec Wr = mul(s.w_linear_acc, vec3(g + (1.0 / mass) * fsum));
        axpy(o.lu, mulT(Jla_u, Wr));
        add_inplace(o.Luu, mulTN(Jla_u, mul(s.w_linear_acc, Jla_u)));
      }
      { // angular acceleration: J = [Jaa_c 0 0 | Jaa_u]
        Vec Wr = mul(s.w_angular_acc, vec3(tsum));
        Vec gx = mulT(Jaa_c, Wr);
        for (int i = 0; i < 3; i++)
          o.lx[i] += gx[i];
        axpy(o.lu, mulT(Jaa_u, Wr));
        Mat WJc = mul(s.w_angular_acc, Jaa_c), WJu = mul(s.w_angular_acc, Jaa_u);
        Mat cc = mulTN(Jaa_c, WJc), cu = mulTN(Jaa_c, WJu);
        for (int i = 0; i < 3; i++)
        {
          for (int j = 0; j < 3; j++)
            o.Lxx(i, j) += cc(i, j);
          for (int j = 0; j < nu; j++)
            o.Lxu(i, j) += cu(i, j);
        }
        add_inplace(o.Luu, mulTN(Jaa_u, WJu));
      }
      o.Cx.resize(nc, 9);
      o.Cu.resize(nc, nu);
      for (int f = 0; f < nf; f++)
        if ((r.mask >> f) & 1u)
        {
          if (fs == 6)
          {
            for (int i = 0; i < 17; i++)
              for (int j = 0; j < 6; j++)
                o.Cu(17 * f + i, 6 * f + j) = Acone(i, j);
            continue;
          }
          const double fx = u[3 * f], fy = u[3 * f + 1], fz = u[3 * f + 2];
          o.Cu(2 * f, 3 * f + 2) = -1.0;
          o.Cu(2 * f + 1, 3 * f) = 2.0 * fx;
          o.Cu(2 * f + 1, 3 * f + 1) = 2.0 * fy;
          o.Cu(2 * f + 1, 3 * f + 2) = -2.0 * s.mu * s.mu * fz;
        }
    }

    // terminal cost: linear + angular momentum with zero references (src/centroidal-dynamics.cpp:306-316)
    double term_eval(Rigid &, const Vec &, const double * x) const
    {
      return quad3(s.w_linear_mom, v3(x[3], x[4], x[5])) + quad3(s.w_angular_mom, v3(x[6], x[7], x[8]));
    }
    // CentroidalOCP::createTerminalConstraint leaves the constraint out (reference src/centroidal-dynamics.cpp:318-328)
    void term_cstr(Rigid &, const double *, const double *, double, double *, Mat *) const {}
    void term_deriv(Rigid &, const Vec &, const double * x, Vec & lx, Mat & Lxx) const
    {
      lx.assign(9, 0.0);
      Lxx.resize(9, 9);
      for (int b = 1; b <= 2; b++)
      {
        const Mat & W = b == 1 ? s.w_linear_mom : s.w_angular_mom;
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
          {
            lx[3 * b + i] += W(i, j) * x[3 * b + j];
            Lxx(3 * b + i, 3 * b + j) = W(i, j);
          }
      }
    }
  };
} // namespace orc
