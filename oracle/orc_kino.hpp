// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_kino.hpp: the kinodynamics stage of the reference, restated.
//   composition  : src/kinodynamics.cpp:40-152 (createStage), :352-364 (createTerminalCost)
//   dynamics     : Aligator KinodynamicsFwdDynamics + IntegratorSemiImplEuler
//                  (constructed at src/kinodynamics.cpp:85-88)  [UPSTREAM-RECALL, SURVEY App. B.2]
//   costs        : state / control / centroidal / centroidal_derivative / <foot>_pose
//                  (src/kinodynamics.cpp:60-83)
//   constraints  : joint box (src/kinodynamics.cpp:91-101), LOCAL frame velocity = 0 per contact
//                  foot (src/kinodynamics.cpp:110-133)
// force_size 3 (point feet): force_cone rows = CentroidalFrictionConeResidual per foot in contact; land_cstr: the height of a landing
// foot pinned to its contact pose (FrameTranslationResidual, z slice).
// force_size 6 (flat feet, the Talos configuration of examples/talos_kinodynamics.py and tests/test_utils.cpp:147-197): u = [(f_i, tau_i) per
// foot ; joint accelerations]; the contact torques enter the angular momentum rate [UPSTREAM-RECALL aligator kinodynamics-fwd.hxx /
// centroidal-momentum-derivative.hxx: hdot.angular += u.segment(i * 6 + 3, 3)]; <foot>_pose_cost is FramePlacementResidual
// log6(M_ref^-1 oMf) (src/kinodynamics.cpp:66-72); the contact constraint is the whole 6-row LOCAL FrameVelocityResidual (:105-123);
// force_cone adds CentroidalWrenchConeResidual per foot in contact: 17 constant linear rows A_cone u_i <= 0 (:114-119); land_cstr has no
// rows for 6-D feet (the branch at :134-146 is the 3-D one).
#pragma once
#include "orc_rigid.hpp"

namespace orc
{
  struct KinoSettings // mirrors KinodynamicsSettings (include/simple-mpc/kinodynamics.hpp:24-51)
  {
    double timestep = 0.01;
    Mat w_x, w_u, w_frame, w_cent, w_centder;
    Vec qmin, qmax;
    double gravity[3] = {0, 0, -9.81};
    double mu = 0.8, Lfoot = 0.01, Wfoot = 0.01;
    int force_size = 3;
    bool kinematics_limits = true, force_cone = false, land_cstr = false;
  };

  enum RowKind
  {
    ROW_ABSENT = 0,
    ROW_EQ = 1,
    ROW_BOX = 2,
    ROW_NEG = 3
  };

  struct StageRef
  {
    unsigned mask = 0xF;      // contact bit per foot
    unsigned land = 0;        // bit per foot: the foot lands at this stage (in contact here, not in the stage before of the cycle)
    Vec u_ref;                // nu   (control_cost target, src/kinodynamics.cpp:61,229-240)
    Vec x_tgt;                // nx   (state_cost target)
    std::vector<V3> foot_ref; // nf   (<foot>_pose_cost references)
  };

  struct StageEval
  {
    Vec xnext, xdot, c;
    double cost = 0;
  };
  struct StageDer
  {
    Mat A, B, Lxx, Lxu, Luu, Cx, Cu;
    Vec lx, lu;
  };

  // wrench cone of a rectangular sole (half length L, half width W, friction mu) on the contact wrench [f ; tau]: A lam <= 0
  inline Mat wrench_cone_matrix(double mu, double L, double W)
  {
    const double m = mu * (L + W);
    const double rows[17][6] = {
      {0, 0, -1, 0, 0, 0},
      {-1, 0, -mu, 0, 0, 0}, {1, 0, -mu, 0, 0, 0}, {0, -1, -mu, 0, 0, 0}, {0, 1, -mu, 0, 0, 0},
      {0, 0, -W, -1, 0, 0}, {0, 0, -W, 1, 0, 0}, {0, 0, -L, 0, -1, 0}, {0, 0, -L, 0, 1, 0},
      {W, L, -m, -mu, -mu, -1}, {W, -L, -m, -mu, mu, -1}, {-W, L, -m, mu, -mu, -1}, {-W, -L, -m, mu, mu, -1},
      {W, L, -m, mu, mu, 1}, {W, -L, -m, mu, -mu, 1}, {-W, L, -m, -mu, mu, 1}, {-W, -L, -m, -mu, -mu, 1}};
    Mat A(17, 6);
    for (int i = 0; i < 17; i++)
      for (int j = 0; j < 6; j++)
        A(i, j) = rows[i][j];
    return A;
  }

  struct KinoModel
  {
    const smpc_robot_model * M;
    KinoSettings s;
    int nq, nv, nx, ndx, nu, nf, nc, fs = 3;
    Mat Acone; // 17 x 6 (6-D feet)
    std::vector<int> row_kind_base; // per-row kind when present
    Vec row_lo, row_hi;

    KinoModel(const smpc_robot_model * m, const KinoSettings & st) : M(m), s(st)
    {
      nq = m->nq;
      nv = m->nv;
      nx = nq + nv;
      ndx = 2 * nv;
      nf = m->nfeet;
      fs = st.force_size;
      nu = nv - 6 + fs * nf;
      Acone = wrench_cone_matrix(st.mu, st.Lfoot, st.Wfoot);
      configure();
    }
    std::vector<double> land_z; // height of the contact poses the cycle stages are created with: feet at the reference state
    void configure()
    {
      nc = (nv - 6) + fs * nf + (s.force_cone ? ncone1() * nf : 0) + (has_land() ? nf : 0);
      Rigid R(M);
      Vec q(M->q_ref, M->q_ref + nq);
      R.fk(q.data());
      land_z.resize(nf);
      for (int f = 0; f < nf; f++)
        land_z[f] = R.foot_p[f][2];
    }
    int ncone1() const { return fs == 6 ? 17 : 2; }                  // cone rows per foot: wrench cone / friction cone
    bool has_land() const { return s.land_cstr && fs == 3; }         // (no land rows for 6-D feet, src/kinodynamics.cpp:134)
    int cone_base() const { return (nv - 6) + fs * nf; }
    int land_base() const { return cone_base() + (s.force_cone ? ncone1() * nf : 0); }
    // fixed row layout: rows [0, nv-6) joint box, rows nv-6+3f.. frame velocity of foot f, then (force_cone) two friction-cone rows
    // per foot: CentroidalFrictionConeResidual(ndx, nu, f, mu, 1e-4) in NegativeOrthant (reference src/kinodynamics.cpp:124-129):
    //   [ -f_z + epsilon ; f_x^2 + f_y^2 - mu^2 f_z^2 ] <= 0   ([UPSTREAM-RECALL] aligator centroidal-friction-cone.hxx; the same
    //   residual as the centroidal OCP's, orc_cent.hpp)
    int row_kind(const StageRef & r, int row) const
    {
      if (row < nv - 6)
        return s.kinematics_limits ? ROW_BOX : ROW_ABSENT;
      // land_cstr: one equality row per foot that lands at this stage, FrameTranslationResidual(contact pose) sliced to z
      // (reference src/kinodynamics.cpp:134-146; land flags of the cycle stages: src/mpc.cpp:167-178)
      if (has_land() && row >= land_base())
        return (((r.land & r.mask) >> (row - land_base())) & 1u) ? ROW_EQ : ROW_ABSENT;
      if (row >= cone_base())
        return ((r.mask >> ((row - cone_base()) / ncone1())) & 1u) ? ROW_NEG : ROW_ABSENT;
      int f = (row - (nv - 6)) / fs;
      return ((r.mask >> f) & 1u) ? ROW_EQ : ROW_ABSENT;
    }
    double row_lo_v(int row) const { return row < nv - 6 ? s.qmin[row] : 0.0; }
    double row_hi_v(int row) const { return row < nv - 6 ? s.qmax[row] : 0.0; }
    void integrate(const double * x, const double * dx, double * out) const { x_integrate(nq, nv, x, dx, out); }
    void difference(const double * x0, const double * x1, double * out) const { x_difference(nq, nv, x0, x1, out); }
    // where a stage's reference vector keeps the force reference of foot f (here: the control reference itself)
    int force_ref_index(int f) const { return fs * f; }
    int n_uref() const { return nu; }

    // ---- continuous dynamics  xdot = (v, a) ----
    // hdot target from contact forces: [m g + sum f ; sum (p_f - c) x f]
    SV hdot_target(const Rigid & R, const StageRef & r, const double * u) const
    {
      SV hd{v3(R.mass * s.gravity[0], R.mass * s.gravity[1], R.mass * s.gravity[2]), v3(0, 0, 0)};
      for (int f = 0; f < nf; f++)
        if ((r.mask >> f) & 1u)
        {
          V3 F = v3(u[fs * f], u[fs * f + 1], u[fs * f + 2]);
          hd.l = hd.l + F;
          hd.a = hd.a + cross(R.foot_p[f] - R.com, F);
          if (fs == 6)
            hd.a = hd.a + v3(u[6 * f + 3], u[6 * f + 4], u[6 * f + 5]);
        }
      return hd;
    }

    // forward: fills R (fk, velocities, forces with the solved acceleration), returns a (nv)
    Vec forward(Rigid & R, const StageRef & r, const double * x, const double * u, Mat * Agb_inv_out = nullptr) const
    {
      const double * q = x;
      const double * v = x + nq;
      R.fk(q);
      R.velocities(v);
      R.forces(v, nullptr);
      SV b = R.dAg_v();
      Mat Ag = R.Ag();
      SV hd = hdot_target(R, r, u);
      Vec rhs = sv_vec(hd - b);
      const double * aj = u + fs * nf;
      for (int i = 0; i < 6; i++)
        for (int k = 6; k < nv; k++)
          rhs[i] -= Ag(i, k) * aj[k - 6];
      Mat Agb(6, 6);
      for (int i = 0; i < 6; i++)
        for (int k = 0; k < 6; k++)
          Agb(i, k) = Ag(i, k);
      Mat Agb_inv = inverse(Agb);
      Vec ab = mul(Agb_inv, rhs);
      Vec a(nv);
      for (int i = 0; i < 6; i++)
        a[i] = ab[i];
      for (int k = 6; k < nv; k++)
        a[k] = aj[k - 6];
      if (Agb_inv_out)
        *Agb_inv_out = Agb_inv;
      return a;
    }

    // da/dq, da/dv (nv x nv), da/du (nv x nu).  R must hold forward() results; recomputes forces.
    void dforward(Rigid & R, const StageRef & r, const double * x, const double * u, const Vec & a, const Mat & Agb_inv,
                  Mat & da_dq, Mat & da_dv, Mat & da_du, Mat * dh_dq_out = nullptr) const
    {
      const double * v = x + nq;
      R.forces(v, a.data());
      R.compute_Bc();
      Mat dh_dq, dhdot_dq, dhdot_dv;
      R.centroidal_derivatives(dh_dq, dhdot_dq, dhdot_dv);
      if (dh_dq_out)
        *dh_dq_out = dh_dq;
      Mat Ag = R.Ag();
      // d hdot_target / dq
      Mat dtgt(6, nv);
      for (int f = 0; f < nf; f++)
        if ((r.mask >> f) & 1u)
        {
          V3 F = v3(u[fs * f], u[fs * f + 1], u[fs * f + 2]);
          for (int k = 0; k < nv; k++)
          {
            V3 dp = R.Jfoot_col(f, k) - R.Jcom_col(k);
            V3 t = cross(dp, F);
            for (int i = 0; i < 3; i++)
              dtgt(3 + i, k) += t[i];
          }
        }
      Mat rq = dtgt;
      add_inplace(rq, dhdot_dq, -1.0);
      Mat ab_dq = mul(Agb_inv, rq);
      Mat ab_dv = mul(Agb_inv, dhdot_dv);
      da_dq.resize(nv, nv);
      da_dv.resize(nv, nv);
      da_du.resize(nv, nu);
      for (int i = 0; i < 6; i++)
        for (int k = 0; k < nv; k++)
        {
          da_dq(i, k) = ab_dq(i, k);
          da_dv(i, k) = -ab_dv(i, k);
        }
      // forces
      for (int f = 0; f < nf; f++)
        if ((r.mask >> f) & 1u)
        {
          M3 X = skew(R.foot_p[f] - R.com);
          Mat G(6, fs);
          for (int i = 0; i < 3; i++)
          {
            G(i, i) = 1.0;
            for (int j = 0; j < 3; j++)
              G(3 + i, j) = X(i, j);
            if (fs == 6)
              G(3 + i, 3 + i) = 1.0; // contact torque
          }
          Mat AG = mul(Agb_inv, G);
          for (int i = 0; i < 6; i++)
            for (int j = 0; j < fs; j++)
              da_du(i, fs * f + j) = AG(i, j);
        }
      // joint accelerations
      Mat Agj(6, nv - 6);
      for (int i = 0; i < 6; i++)
        for (int k = 6; k < nv; k++)
          Agj(i, k - 6) = Ag(i, k);
      Mat AA = mul(Agb_inv, Agj);
      for (int i = 0; i < 6; i++)
        for (int k = 0; k < nv - 6; k++)
          da_du(i, fs * nf + k) = -AA(i, k);
      for (int k = 0; k < nv - 6; k++)
        da_du(6 + k, fs * nf + k) = 1.0;
    }

    // <foot>_pose_cost residual and its Jacobian w.r.t. q: translation error (3-D feet) or log6(M_ref^-1 oMf) with the LOCAL 6-D frame
    // Jacobian (FramePlacementResidual; M_ref = (identity rotation, reference translation), as MPC::updateStepTrackerReferences sets
    // it, src/mpc.cpp:304-308)
    Vec pose_residual(const Rigid & R, const StageRef & r, int f, Mat * Jq = nullptr) const
    {
      const int l = M->foot_joint[f];
      if (fs == 3)
      {
        V3 e = R.foot_p[f] - r.foot_ref[f];
        if (Jq)
        {
          Jq->resize(3, nv);
          for (int k = 0; k < nv; k++)
          {
            V3 c = R.Jfoot_col(f, k);
            for (int i = 0; i < 3; i++)
              (*Jq)(i, k) = c[i];
          }
        }
        return Vec{e[0], e[1], e[2]};
      }
      const SE3 Mref{m3_id(), r.foot_ref[f]}, Mf{R.oMi[l].R, R.foot_p[f]};
      Vec e(6);
      log6(inv(Mref) * Mf, e.data());
      if (Jq)
      {
        const Mat Jl = Jlog6(inv(Mref) * Mf);
        const M3 Rt = tr(Mf.R);
        Mat Jloc(6, nv);
        for (int k = 0; k < nv; k++)
          if (R.is_ancestor_dof(k, l))
          {
            const V3 lin = Rt * R.Jfoot_col(f, k), ang = Rt * R.S[k].a;
            for (int i = 0; i < 3; i++)
            {
              Jloc(i, k) = lin[i];
              Jloc(3 + i, k) = ang[i];
            }
          }
        *Jq = mul(Jl, Jloc);
      }
      return e;
    }

    // ---- cost pieces ----
    static double quad(const Mat & W, const Vec & r)
    {
      Vec Wr = mul(W, r);
      return 0.5 * dot(r, Wr);
    }

    // Evaluate dynamics + cost + constraints at (x,u).
    void eval(Rigid & R, const StageRef & r, const double * x, const double * u, StageEval & o) const
    {
      const double dt = s.timestep;
      Vec a = forward(R, r, x, u);
      o.xdot.assign(2 * nv, 0.0);
      for (int i = 0; i < nv; i++)
      {
        o.xdot[i] = x[nq + i];
        o.xdot[nv + i] = a[i];
      }
      // semi-implicit Euler: v+ = v + dt a ; q+ = q (+) dt v+
      Vec dx(ndx);
      for (int i = 0; i < nv; i++)
      {
        dx[nv + i] = dt * a[i];
        dx[i] = dt * (x[nq + i] + dx[nv + i]);
      }
      o.xnext.assign(nx, 0.0);
      x_integrate(nq, nv, x, dx.data(), o.xnext.data());
      // costs
      double cost = 0;
      Vec rx(ndx);
      x_difference(nq, nv, r.x_tgt.data(), x, rx.data());
      cost += quad(s.w_x, rx);
      Vec ru(nu);
      for (int i = 0; i < nu; i++)
        ru[i] = u[i] - r.u_ref[i];
      cost += quad(s.w_u, ru);
      cost += quad(s.w_cent, sv_vec(R.hg()));
      cost += quad(s.w_centder, sv_vec(hdot_target(R, r, u)));
      for (int f = 0; f < nf; f++)
        cost += quad(s.w_frame, pose_residual(R, r, f));
      o.cost = cost;
      // constraints
      o.c.assign(nc, 0.0);
      if (s.kinematics_limits)
        for (int i = 0; i < nv - 6; i++)
          o.c[i] = x[7 + i]; // (x (-) neutral)[6+i]
      for (int f = 0; f < nf; f++)
        if ((r.mask >> f) & 1u)
        {
          const int l = M->foot_joint[f];
          V3 vw = R.vel[l].l + cross(R.vel[l].a, R.foot_p[f]);
          V3 c = tr(R.oMi[l].R) * vw;
          for (int i = 0; i < 3; i++)
            o.c[nv - 6 + fs * f + i] = c[i];
          if (fs == 6)
          {
            V3 w = tr(R.oMi[l].R) * R.vel[l].a;
            for (int i = 0; i < 3; i++)
              o.c[nv - 6 + 6 * f + 3 + i] = w[i];
            if (s.force_cone)
              for (int i = 0; i < 17; i++)
              {
                double acc = 0.0;
                for (int j = 0; j < 6; j++)
                  acc += Acone(i, j) * u[6 * f + j];
                o.c[cone_base() + 17 * f + i] = acc;
              }
            continue;
          }
          if (s.force_cone)
          {
            const double fx = u[3 * f], fy = u[3 * f + 1], fz = u[3 * f + 2];
            o.c[cone_base() + 2 * f] = -fz + 1e-4;
            o.c[cone_base() + 2 * f + 1] = fx * fx + fy * fy - s.mu * s.mu * fz * fz;
          }
          if (s.land_cstr && ((r.land >> f) & 1u))
            o.c[land_base() + f] = R.foot_p[f][2] - land_z[f];
        }
    }

    // Derivatives at (x,u): A,B, cost gradient + Gauss-Newton Hessian, constraint Jacobians.
    void deriv(Rigid & R, const StageRef & r, const double * x, const double * u, StageDer & o) const
    {
      const double dt = s.timestep;
      Mat Agb_inv;
      Vec a = forward(R, r, x, u, &Agb_inv);
      Mat da_dq, da_dv, da_du, dh_dq;
      dforward(R, r, x, u, a, Agb_inv, da_dq, da_dv, da_du, &dh_dq);
      // d(dx)/dx and d(dx)/du, dx = [dt (v + dt a); dt a]
      Mat Dx(ndx, ndx), Du(ndx, nu);
      for (int i = 0; i < nv; i++)
      {
        for (int k = 0; k < nv; k++)
        {
          Dx(nv + i, k) = dt * da_dq(i, k);
          Dx(nv + i, nv + k) = dt * da_dv(i, k);
          Dx(i, k) = dt * dt * da_dq(i, k);
          Dx(i, nv + k) = dt * dt * da_dv(i, k);
        }
        Dx(i, nv + i) += dt;
        for (int k = 0; k < nu; k++)
        {
          Du(nv + i, k) = dt * da_du(i, k);
          Du(i, k) = dt * dt * da_du(i, k);
        }
      }
      double nu6[6];
      for (int i = 0; i < 6; i++)
        nu6[i] = dt * (x[nq + i] + dt * a[i]);
      Mat Je = Jexp6(nu6);                       // d (x (+) dx) / d dx, SE3 block
      Mat Jq = action_matrix(inv(exp6(nu6)));    // d (x (+) dx) / d x, SE3 block
      // transport: first 6 rows of Dx, Du get multiplied by Je
      o.A.resize(ndx, ndx);
      o.B.resize(ndx, nu);
      for (int i = 0; i < ndx; i++)
      {
        for (int k = 0; k < ndx; k++)
        {
          double sacc;
          if (i < 6)
          {
            sacc = 0;
            for (int m = 0; m < 6; m++)
              sacc += Je(i, m) * Dx(m, k);
          }
          else
            sacc = Dx(i, k);
          o.A(i, k) = sacc;
        }
        for (int k = 0; k < nu; k++)
        {
          double sacc;
          if (i < 6)
          {
            sacc = 0;
            for (int m = 0; m < 6; m++)
              sacc += Je(i, m) * Du(m, k);
          }
          else
            sacc = Du(i, k);
          o.B(i, k) = sacc;
        }
      }
      for (int i = 0; i < 6; i++)
        for (int k = 0; k < 6; k++)
          o.A(i, k) += Jq(i, k);
      for (int i = 6; i < ndx; i++)
        o.A(i, i) += 1.0;

      // ---- costs ----
      o.lx.assign(ndx, 0.0);
      o.lu.assign(nu, 0.0);
      o.Lxx.resize(ndx, ndx);
      o.Lxu.resize(ndx, nu);
      o.Luu.resize(nu, nu);
      auto add_cost = [&](const Mat & W, const Vec & res, const Mat & Jx, const Mat * Ju) {
        Vec Wr = mul(W, res);
        axpy(o.lx, mulT(Jx, Wr));
        Mat WJx = mul(W, Jx);
        add_inplace(o.Lxx, mulTN(Jx, WJx));
        if (Ju)
        {
          axpy(o.lu, mulT(*Ju, Wr));
          Mat WJu = mul(W, *Ju);
          add_inplace(o.Lxu, mulTN(Jx, WJu));
          add_inplace(o.Luu, mulTN(*Ju, WJu));
        }
      };
      { // state cost: r = x (-) x_tgt, J = blockdiag(Jlog6(Mt^-1 M), I)
        Vec rx(ndx);
        x_difference(nq, nv, r.x_tgt.data(), x, rx.data());
        SE3 Mt{quat_to_R(r.x_tgt.data() + 3), v3(r.x_tgt[0], r.x_tgt[1], r.x_tgt[2])};
        SE3 Mx{quat_to_R(x + 3), v3(x[0], x[1], x[2])};
        Mat Jl = Jlog6(inv(Mt) * Mx);
        Mat Jx = Mat::identity(ndx);
        for (int i = 0; i < 6; i++)
          for (int k = 0; k < 6; k++)
            Jx(i, k) = Jl(i, k);
        add_cost(s.w_x, rx, Jx, nullptr);
      }
      { // control cost
        Vec ru(nu);
        for (int i = 0; i < nu; i++)
          ru[i] = u[i] - r.u_ref[i];
        Vec Wr = mul(s.w_u, ru);
        axpy(o.lu, Wr);
        add_inplace(o.Luu, s.w_u);
      }
      Mat Ag = R.Ag();
      { // centroidal momentum cost: r = hg(q,v)
        Mat Jx(6, ndx);
        for (int i = 0; i < 6; i++)
          for (int k = 0; k < nv; k++)
          {
            Jx(i, k) = dh_dq(i, k);
            Jx(i, nv + k) = Ag(i, k);
          }
        add_cost(s.w_cent, sv_vec(R.hg()), Jx, nullptr);
      }
      { // centroidal momentum derivative cost: r = hdot_target(q,u)
        Mat Jx(6, ndx), Ju(6, nu);
        for (int f = 0; f < nf; f++)
          if ((r.mask >> f) & 1u)
          {
            V3 F = v3(u[fs * f], u[fs * f + 1], u[fs * f + 2]);
            for (int k = 0; k < nv; k++)
            {
              V3 t = cross(R.Jfoot_col(f, k) - R.Jcom_col(k), F);
              for (int i = 0; i < 3; i++)
                Jx(3 + i, k) += t[i];
            }
            M3 X = skew(R.foot_p[f] - R.com);
            for (int i = 0; i < 3; i++)
            {
              Ju(i, fs * f + i) = 1.0;
              for (int j = 0; j < 3; j++)
                Ju(3 + i, fs * f + j) = X(i, j);
              if (fs == 6)
                Ju(3 + i, 6 * f + 3 + i) = 1.0;
            }
          }
        add_cost(s.w_centder, sv_vec(hdot_target(R, r, u)), Jx, &Ju);
      }
      for (int f = 0; f < nf; f++)
      { // foot pose cost (translation / placement)
        Mat Jq;
        const Vec e = pose_residual(R, r, f, &Jq);
        Mat Jx(fs, ndx);
        for (int i = 0; i < fs; i++)
          for (int k = 0; k < nv; k++)
            Jx(i, k) = Jq(i, k);
        add_cost(s.w_frame, e, Jx, nullptr);
      }
      // ---- constraints ----
      o.Cx.resize(nc, ndx);
      o.Cu.resize(nc, nu);
      if (s.kinematics_limits)
        for (int i = 0; i < nv - 6; i++)
          o.Cx(i, 6 + i) = 1.0;
      for (int f = 0; f < nf; f++)
        if ((r.mask >> f) & 1u)
        {
          if (fs == 6)
          {
            Vec c6;
            Mat dq, dv;
            R.foot_local_velocity6(f, c6, dq, dv);
            for (int i = 0; i < 6; i++)
              for (int k = 0; k < nv; k++)
              {
                o.Cx(nv - 6 + 6 * f + i, k) = dq(i, k);
                o.Cx(nv - 6 + 6 * f + i, nv + k) = dv(i, k);
              }
            if (s.force_cone)
              for (int i = 0; i < 17; i++)
                for (int j = 0; j < 6; j++)
                  o.Cu(cone_base() + 17 * f + i, 6 * f + j) = Acone(i, j);
            continue;
          }
          V3 c;
          Mat dq, dv;
          R.foot_local_velocity(f, c, dq, dv);
          for (int i = 0; i < 3; i++)
            for (int k = 0; k < nv; k++)
            {
              o.Cx(nv - 6 + 3 * f + i, k) = dq(i, k);
              o.Cx(nv - 6 + 3 * f + i, nv + k) = dv(i, k);
            }
          if (s.force_cone)
          {
            const int cb = cone_base() + 2 * f;
            o.Cu(cb, 3 * f + 2) = -1.0;
            o.Cu(cb + 1, 3 * f) = 2.0 * u[3 * f];
            o.Cu(cb + 1, 3 * f + 1) = 2.0 * u[3 * f + 1];
            o.Cu(cb + 1, 3 * f + 2) = -2.0 * s.mu * s.mu * u[3 * f + 2];
          }
          if (s.land_cstr && ((r.land >> f) & 1u))
          { // d p_z / dq = (R_l J_local)_z: the local velocity Jacobian turned into the world frame
            const M3 & Rl = R.oMi[M->foot_joint[f]].R;
            for (int k = 0; k < nv; k++)
              o.Cx(land_base() + f, k) = Rl(2, 0) * dv(0, k) + Rl(2, 1) * dv(1, k) + Rl(2, 2) * dv(2, k);
          }
        }
    }

    // ---- terminal cost: state (target x_tgt) + centroidal with 10 w_cent ----
    double term_eval(Rigid & R, const Vec & x_tgt, const double * x) const
    {
      R.fk(x);
      R.velocities(x + nq);
      Vec rx(ndx);
      x_difference(nq, nv, x_tgt.data(), x, rx.data());
      Mat W10 = s.w_cent;
      for (auto & e : W10.a)
        e *= 10.0;
      return quad(s.w_x, rx) + quad(W10, sv_vec(R.hg()));
    }
    // terminal constraint DCMPositionResidual: c = com(q) + tau vcom(q, v) - ref, C = [Jcom + tau dvcom/dq | tau Jcom]
    // (reference src/kinodynamics.cpp:366-377; [UPSTREAM-RECALL] aligator DCMPositionResidualTpl: centerOfMass + alpha vcom - dcm_ref,
    // Jacobian from jacobianCenterOfMass and getCenterOfMassVelocityDerivatives)
    void term_cstr(Rigid & R, const double * x, const double ref[3], double tau, double * c, Mat * C) const
    {
      R.fk(x);
      R.velocities(x + nq);
      const SV h = R.hg();
      const double m = M->total_mass;
      for (int i = 0; i < 3; i++)
        c[i] = R.com[i] + tau * h.l[i] / m - ref[i];
      if (!C)
        return;
      Vec zero(nv, 0.0);
      R.forces(x + nq, zero.data());
      R.compute_Bc();
      Mat dh_dq, d1, d2;
      R.centroidal_derivatives(dh_dq, d1, d2);
      const Mat Ag = R.Ag();
      C->resize(3, ndx);
      for (int i = 0; i < 3; i++)
        for (int k = 0; k < nv; k++)
        {
          (*C)(i, k) = (Ag(i, k) + tau * dh_dq(i, k)) / m;
          (*C)(i, nv + k) = tau * Ag(i, k) / m;
        }
    }
    void term_deriv(Rigid & R, const Vec & x_tgt, const double * x, Vec & lx, Mat & Lxx) const
    {
      R.fk(x);
      R.velocities(x + nq);
      Vec zero(nv, 0.0);
      R.forces(x + nq, zero.data());
      R.compute_Bc();
      Mat dh_dq, d1, d2;
      R.centroidal_derivatives(dh_dq, d1, d2);
      Mat Ag = R.Ag();
      lx.assign(ndx, 0.0);
      Lxx.resize(ndx, ndx);
      {
        Vec rx(ndx);
        x_difference(nq, nv, x_tgt.data(), x, rx.data());
        SE3 Mt{quat_to_R(x_tgt.data() + 3), v3(x_tgt[0], x_tgt[1], x_tgt[2])};
        SE3 Mx{quat_to_R(x + 3), v3(x[0], x[1], x[2])};
        Mat Jl = Jlog6(inv(Mt) * Mx);
        Mat Jx = Mat::identity(ndx);
        for (int i = 0; i < 6; i++)
          for (int k = 0; k < 6; k++)
            Jx(i, k) = Jl(i, k);
        Vec Wr = mul(s.w_x, rx);
        axpy(lx, mulT(Jx, Wr));
        add_inplace(Lxx, mulTN(Jx, mul(s.w_x, Jx)));
      }
      {
        Mat W10 = s.w_cent;
        for (auto & e : W10.a)
          e *= 10.0;
        Mat Jx(6, ndx);
        for (int i = 0; i < 6; i++)
          for (int k = 0; k < nv; k++)
          {
            Jx(i, k) = dh_dq(i, k);
            Jx(i, nv + k) = Ag(i, k);
          }
        Vec Wr = mul(W10, sv_vec(R.hg()));
        axpy(lx, mulT(Jx, Wr));
        add_inplace(Lxx, mulTN(Jx, mul(W10, Jx)));
      }
    }
  };
} // namespace orc
