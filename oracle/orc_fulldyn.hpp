// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_fulldyn.hpp: the stage model of the full-dynamics OCP, restating what
// FullDynamicsOCP::createStage composes from Aligator pieces [REF src/fulldynamics.cpp:78-214] for 3-D feet
// (force_size 3, the Go2 configuration of examples/go2_fulldynamics.py and benchmark/go2.cpp):
//   state  x = (q, v) on SE3 x R^(nq-7) x R^nv, control u = joint torques (nu = nv - 6) [REF src/ocp-handler.cpp:15-18]
//   dynamics   MultibodyConstraintFwdDynamics(actuation [0; I], contacts of the stage) + IntegratorSemiImplEuler
//              [REF src/fulldynamics.cpp:139-140]  -> orc_full.hpp
//   costs      state_cost (x (-) x_ref, w_x), control_cost (u - 0, w_u), centroidal_cost (hg(q, v), w_cent),
//              <foot>_pose_cost (FrameTranslationResidual, w_frame) for EVERY foot, <foot>_force_cost
//              (ContactForceResidual: contact force of the constrained dynamics - reference, w_forces) for the feet in
//              contact [REF src/fulldynamics.cpp:88-137]
//   constraints  torque box  umin <= u <= umax  (ControlErrorResidual, rows 0 .. nu-1) if torque_limits;
//              joint box  qmin <= (x (-) neutral)[6 .. nv) <= qmax  (rows nu .. nu+nv-7) if kinematics_limits
//              [REF src/fulldynamics.cpp:144-162]
//   terminal   state_cost + 10 x centroidal_cost [REF src/fulldynamics.cpp:418-430]
// 6-D feet (force_size 6, the Talos configuration of examples/talos_fulldynamics.py): <foot>_pose_cost is a FramePlacementResidual
//   log6(M_ref^-1 oMf) with the LOCAL frame Jacobian [REF src/fulldynamics.cpp:103-109], the force cost is on the 6-D contact
//   wrench, and with force_cone every foot in contact adds MultibodyWrenchConeResidual rows in the negative orthant
//   [REF src/fulldynamics.cpp:163-173]: 17 linear rows A_cone lam (unilaterality, friction pyramid, centre of pressure inside
//   the sole, yaw torque bounds; [UPSTREAM-RECALL] Aligator's wrench-cone matrix after Caron et al. 2015), rows after the boxes.
// land_cstr rows [REF :175-209]: 6 LOCAL_WORLD_ALIGNED frame-velocity rows of a landing 6-D foot, 3 + the height of the contact pose for a 3-D foot.
// StageRef::u_ref carries [control reference (nu) ; force reference per foot (force_size nf)] for this model.
#pragma once
#include "orc_full.hpp"
#include "orc_kino.hpp"

namespace orc
{
  struct FullSettings // mirrors FullDynamicsSettings (include/simple-mpc/fulldynamics.hpp:28-65)
  {
    double timestep = 0.01;
    Mat w_x, w_u, w_cent, w_forces, w_frame;
    double gravity[3] = {0, 0, -9.81};
    double mu = 0.8, Lfoot = 0.1, Wfoot = 0.075;
    int force_size = 3;
    bool torque_limits = true, kinematics_limits = true, force_cone = false, land_cstr = false;
    Vec umin, umax, qmin, qmax;
    double Kp[6] = {0, 0, 0, 0, 0, 0}, Kd[6] = {0, 0, 0, 0, 0, 0};
  };

  // (wrench_cone_matrix: orc_kino.hpp -- the kinodynamics and centroidal OCPs of 6-D feet use the same 17 x 6 matrix on u)
  struct FullModel
  {
    const smpc_robot_model * M;
    FullSettings s;
    int nq, nv, nx, ndx, nu, nf, nc, fs, ncone1, nland1;
    Mat Acone;
    std::vector<double> land_z; // height of the contact poses the cycle stages are created with: feet at the reference state

    FullModel(const smpc_robot_model * m, const FullSettings & st) : M(m), s(st)
    {
      nq = m->nq;
      nv = m->nv;
      nx = nq + nv;
      ndx = 2 * nv;
      nu = nv - 6;
      nf = m->nfeet;
      fs = st.force_size;
      // force_cone: 17 wrench-cone rows per 6-D foot [REF src/fulldynamics.cpp:163-173]; 5 friction-pyramid rows per 3-D foot (MultibodyFrictionConeResidual
      // [REF :185-190]; [UPSTREAM-RECALL] its 5 x 3 matrix: unilaterality and +-f_x, +-f_y <= mu f_z on the contact force in the contact's frame =
      // rows 0 .. 4 / columns 0 .. 2 of the wrench-cone matrix)
      ncone1 = st.force_cone ? (fs == 6 ? 17 : 5) : 0;
      // land_cstr: a foot that lands at this stage keeps a zero LOCAL_WORLD_ALIGNED frame velocity (6 rows of a 6-D foot [REF :175-181]; the 3
      // linear rows of a 3-D foot and the height of the contact pose, FrameTranslationResidual sliced to z [REF :191-210]); equality rows
      nland1 = st.land_cstr ? (fs == 6 ? 6 : 4) : 0;
      nc = nu + (nv - 6) + ncone1 * nf + nland1 * nf;
      Acone = wrench_cone_matrix(st.mu, st.Lfoot, st.Wfoot);
      if (st.land_cstr)
      {
        Rigid R(m);
        Vec q(m->q_ref, m->q_ref + nq);
        R.fk(q.data());
        land_z.resize(nf);
        for (int f = 0; f < nf; f++)
          land_z[f] = R.foot_p[f][2];
      }
    }
    int land_base() const { return 2 * nu + ncone1 * nf; }
    int row_kind(const StageRef & r, int row) const
    {
      if (row < nu)
        return s.torque_limits ? ROW_BOX : ROW_ABSENT;
      if (row < 2 * nu)
        return s.kinematics_limits ? ROW_BOX : ROW_ABSENT;
      if (row < land_base())
        return ((r.mask >> ((row - 2 * nu) / ncone1)) & 1u) ? ROW_NEG : ROW_ABSENT; // cone rows of a foot in contact
      return (((r.land & r.mask) >> ((row - land_base()) / nland1)) & 1u) ? ROW_EQ : ROW_ABSENT; // rows of a landing foot
    }
    double row_lo_v(int row) const { return row < nu ? s.umin[row] : s.qmin[row - nu]; }
    double row_hi_v(int row) const { return row < nu ? s.umax[row] : s.qmax[row - nu]; }
    void integrate(const double * x, const double * dx, double * out) const { x_integrate(nq, nv, x, dx, out); }
    void difference(const double * x0, const double * x1, double * out) const { x_difference(nq, nv, x0, x1, out); }
    int force_ref_index(int f) const { return nu + fs * f; }
    int n_uref() const { return nu + fs * nf; }

    ConstraintDynamics dynamics() const
    {
      ConstraintDynamics cd(M);
      cd.fs = fs;
      for (int i = 0; i < 3; i++)
        cd.gravity[i] = s.gravity[i];
      for (int i = 0; i < fs; i++)
      {
        cd.Kp[i] = s.Kp[i];
        cd.Kd[i] = s.Kd[i];
      }
      return cd;
    }
    // <foot>_pose_cost residual: translation error (3-D feet) or log6(M_ref^-1 oMf), M_ref = (identity rotation, reference
    // translation) as MPC::updateStepTrackerReferences sets it [REF src/mpc.cpp:304-308]
    Vec pose_residual(const ConstraintDynamics & cd, const StageRef & r, int f) const
    {
      if (fs == 3)
      {
        const V3 e = cd.R.foot_p[f] - r.foot_ref[f];
        return Vec{e[0], e[1], e[2]};
      }
      const SE3 Mref{m3_id(), r.foot_ref[f]}, Mf{cd.foot_R(f), cd.R.foot_p[f]};
      Vec e(6);
      log6(inv(Mref) * Mf, e.data());
      return e;
    }
    static double quad(const Mat & W, const Vec & r) { return 0.5 * dot(r, mul(W, r)); }

    void eval(Rigid & R, const StageRef & r, const double * x, const double * u, StageEval & o) const
    {
      const double dt = s.timestep;
      ConstraintDynamics cd = dynamics();
      cd.compute(x, x + nq, u, r.mask);
      o.xdot.assign(2 * nv, 0.0);
      Vec dx(ndx);
      for (int i = 0; i < nv; i++)
      {
        o.xdot[i] = x[nq + i];
        o.xdot[nv + i] = cd.a[i];
        dx[nv + i] = dt * cd.a[i];
        dx[i] = dt * (x[nq + i] + dx[nv + i]);
      }
      o.xnext.assign(nx, 0.0);
      x_integrate(nq, nv, x, dx.data(), o.xnext.data());
      double cost = 0;
      Vec rx(ndx);
      x_difference(nq, nv, r.x_tgt.data(), x, rx.data());
      cost += quad(s.w_x, rx);
      Vec ru(nu);
      for (int i = 0; i < nu; i++)
        ru[i] = u[i] - r.u_ref[i];
      cost += quad(s.w_u, ru);
      cost += quad(s.w_cent, sv_vec(cd.R.hg()));
      for (int f = 0; f < nf; f++)
        cost += quad(s.w_frame, pose_residual(cd, r, f));
      for (size_t c = 0; c < cd.feet.size(); c++)
      {
        const int f = cd.feet[c];
        Vec e(fs);
        for (int i = 0; i < fs; i++)
          e[i] = cd.lam[fs * c + i] - r.u_ref[nu + fs * f + i];
        cost += quad(s.w_forces, e);
      }
      o.cost = cost;
      o.c.assign(nc, 0.0);
      for (size_t c = 0; c < cd.feet.size() && ncone1 > 0; c++)
      {
        const int f = cd.feet[c];
        for (int i = 0; i < ncone1; i++)
        {
          double acc = 0;
          for (int j = 0; j < fs; j++)
            acc += Acone(i, j) * cd.lam[fs * c + j];
          o.c[2 * nu + ncone1 * f + i] = acc;
        }
      }
      for (int f = 0; f < nf && nland1 > 0; f++)
        if (((r.land & r.mask) >> f) & 1u)
        {
          const int j = M->foot_joint[f], row = land_base() + nland1 * f;
          const V3 p = cd.R.foot_p[f], w = cd.R.vel[j].a;
          const V3 vp = cd.R.vel[j].l + cross(w, p);
          for (int i = 0; i < 3; i++)
          {
            o.c[row + i] = vp[i];
            if (fs == 6)
              o.c[row + 3 + i] = w[i];
          }
          if (fs == 3)
            o.c[row + 3] = p[2] - land_z[f];
        }
      if (s.torque_limits)
        for (int i = 0; i < nu; i++)
          o.c[i] = u[i];
      if (s.kinematics_limits)
        for (int i = 0; i < nv - 6; i++)
          o.c[nu + i] = x[7 + i];
      R = cd.R;
    }

    void deriv(Rigid & R, const StageRef & r, const double * x, const double * u, StageDer & o) const
    {
      const double dt = s.timestep;
      ConstraintDynamics cd = dynamics();
      cd.compute(x, x + nq, u, r.mask);
      cd.derivatives(x + nq);
      const Vec & a = cd.a;
      // d(dx)/dx and d(dx)/du, dx = [dt (v + dt a); dt a]  (semi-implicit Euler, as orc_kino.hpp)
      Mat Dx(ndx, ndx), Du(ndx, nu);
      for (int i = 0; i < nv; i++)
      {
        for (int k = 0; k < nv; k++)
        {
          Dx(nv + i, k) = dt * cd.da_dq(i, k);
          Dx(nv + i, nv + k) = dt * cd.da_dv(i, k);
          Dx(i, k) = dt * dt * cd.da_dq(i, k);
          Dx(i, nv + k) = dt * dt * cd.da_dv(i, k);
        }
        Dx(i, nv + i) += dt;
        for (int k = 0; k < nu; k++)
        {
          Du(nv + i, k) = dt * cd.da_dtau(i, k);
          Du(i, k) = dt * dt * cd.da_dtau(i, k);
        }
      }
      double nu6[6];
      for (int i = 0; i < 6; i++)
        nu6[i] = dt * (x[nq + i] + dt * a[i]);
      const Mat Je = Jexp6(nu6);
      const Mat Jq = action_matrix(inv(exp6(nu6)));
      o.A.resize(ndx, ndx);
      o.B.resize(ndx, nu);
      for (int i = 0; i < ndx; i++)
      {
        for (int k = 0; k < ndx; k++)
        {
          double sacc = 0;
          if (i < 6)
            for (int m = 0; m < 6; m++)
              sacc += Je(i, m) * Dx(m, k);
          else
            sacc = Dx(i, k);
          o.A(i, k) = sacc;
        }
        for (int k = 0; k < nu; k++)
        {
          double sacc = 0;
          if (i < 6)
            for (int m = 0; m < 6; m++)
              sacc += Je(i, m) * Du(m, k);
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

      o.lx.assign(ndx, 0.0);
      o.lu.assign(nu, 0.0);
      o.Lxx.resize(ndx, ndx);
      o.Lxu.resize(ndx, nu);
      o.Luu.resize(nu, nu);
      auto add_cost = [&](const Mat & W, const Vec & res, const Mat & Jx, const Mat * Ju) {
        const Vec Wr = mul(W, res);
        axpy(o.lx, mulT(Jx, Wr));
        const Mat WJx = mul(W, Jx);
        add_inplace(o.Lxx, mulTN(Jx, WJx));
        if (Ju)
        {
          axpy(o.lu, mulT(*Ju, Wr));
          const Mat WJu = mul(W, *Ju);
          add_inplace(o.Lxu, mulTN(Jx, WJu));
          add_inplace(o.Luu, mulTN(*Ju, WJu));
        }
      };
      { // state cost
        Vec rx(ndx);
        x_difference(nq, nv, r.x_tgt.data(), x, rx.data());
        SE3 Mt{quat_to_R(r.x_tgt.data() + 3), v3(r.x_tgt[0], r.x_tgt[1], r.x_tgt[2])};
        SE3 Mx{quat_to_R(x + 3), v3(x[0], x[1], x[2])};
        const Mat Jl = Jlog6(inv(Mt) * Mx);
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
        axpy(o.lu, mul(s.w_u, ru));
        add_inplace(o.Luu, s.w_u);
      }
      { // centroidal momentum cost (cd.derivatives left forces(v, a) and Bc in cd.R)
        Mat dh_dq, d1, d2;
        cd.R.centroidal_derivatives(dh_dq, d1, d2);
        const Mat Ag = cd.R.Ag();
        Mat Jx(6, ndx);
        for (int i = 0; i < 6; i++)
          for (int k = 0; k < nv; k++)
          {
            Jx(i, k) = dh_dq(i, k);
            Jx(i, nv + k) = Ag(i, k);
          }
        add_cost(s.w_cent, sv_vec(cd.R.hg()), Jx, nullptr);
      }
      for (int f = 0; f < nf; f++)
      { // foot pose cost
        Mat Jx(fs, ndx);
        if (fs == 3)
        {
          for (int k = 0; k < nv; k++)
          {
            const V3 c = cd.R.Jfoot_col(f, k);
            for (int i = 0; i < 3; i++)
              Jx(i, k) = c[i];
          }
        }
        else
        {
          // FramePlacementResidual: Jlog6(M_ref^-1 oMf) * (LOCAL 6-D frame Jacobian)
          const SE3 Mref{m3_id(), r.foot_ref[f]}, Mf{cd.foot_R(f), cd.R.foot_p[f]};
          const Mat Jl = Jlog6(inv(Mref) * Mf);
          const M3 Rt = tr(Mf.R);
          Mat Jloc(6, nv);
          for (int k = 0; k < nv; k++)
            if (cd.R.is_ancestor_dof(k, M->foot_joint[f]))
            {
              const V3 lin = Rt * cd.R.Jfoot_col(f, k), ang = Rt * cd.R.S[k].a;
              for (int i = 0; i < 3; i++)
              {
                Jloc(i, k) = lin[i];
                Jloc(3 + i, k) = ang[i];
              }
            }
          const Mat Jq = mul(Jl, Jloc);
          for (int i = 0; i < 6; i++)
            for (int k = 0; k < nv; k++)
              Jx(i, k) = Jq(i, k);
        }
        add_cost(s.w_frame, pose_residual(cd, r, f), Jx, nullptr);
      }
      for (size_t c = 0; c < cd.feet.size(); c++)
      { // contact force cost
        const int f = cd.feet[c];
        Vec e(fs);
        Mat Jx(fs, ndx), Ju(fs, nu);
        for (int i = 0; i < fs; i++)
        {
          e[i] = cd.lam[fs * c + i] - r.u_ref[nu + fs * f + i];
          for (int k = 0; k < nv; k++)
          {
            Jx(i, k) = cd.dlam_dq(fs * (int)c + i, k);
            Jx(i, nv + k) = cd.dlam_dv(fs * (int)c + i, k);
          }
          for (int k = 0; k < nu; k++)
            Ju(i, k) = cd.dlam_dtau(fs * (int)c + i, k);
        }
        add_cost(s.w_forces, e, Jx, &Ju);
      }
      o.Cx.resize(nc, ndx);
      o.Cu.resize(nc, nu);
      for (size_t c = 0; c < cd.feet.size() && ncone1 > 0; c++)
      { // wrench cone rows: A_cone * d lam / d(x, u)
        const int f = cd.feet[c];
        for (int i = 0; i < ncone1; i++)
        {
          const int row = 2 * nu + ncone1 * f + i;
          for (int j = 0; j < fs; j++)
          {
            const double aij = Acone(i, j);
            if (aij == 0.0)
              continue;
            for (int k = 0; k < nv; k++)
            {
              o.Cx(row, k) += aij * cd.dlam_dq(fs * (int)c + j, k);
              o.Cx(row, nv + k) += aij * cd.dlam_dv(fs * (int)c + j, k);
            }
            for (int k = 0; k < nu; k++)
              o.Cu(row, k) += aij * cd.dlam_dtau(fs * (int)c + j, k);
          }
        }
      }
      for (int f = 0; f < nf && nland1 > 0; f++)
        if (((r.land & r.mask) >> f) & 1u)
        { // frame velocity (LOCAL_WORLD_ALIGNED) and height of a landing foot: functions of the state only
          const int l = M->foot_joint[f], row = land_base() + nland1 * f;
          const V3 p = cd.R.foot_p[f], w = cd.R.vel[l].a;
          const V3 vp = cd.R.vel[l].l + cross(w, p);
          for (int k = 0; k < nv; k++)
          {
            if (!cd.R.is_ancestor_dof(k, l))
              continue;
            const int lam_ = M->parent[cd.R.dof2j[k]];
            const SV d = lam_ >= 0 ? crm(cd.R.vel[lam_], cd.R.S[k]) : sv_zero(); // non-rigid part of d(v_l)/dq_k (orc_full.hpp)
            const V3 sa = cd.R.S[k].a, vv = cd.R.S[k].l + cross(sa, p); // d(point position)/dq_k = d(point velocity)/dv_k
            const V3 vq = d.l + cross(d.a, p) + cross(sa, vp), wq = d.a + cross(sa, w);
            for (int i = 0; i < 3; i++)
            {
              o.Cx(row + i, k) = vq[i];
              o.Cx(row + i, nv + k) = vv[i];
              if (fs == 6)
              {
                o.Cx(row + 3 + i, k) = wq[i];
                o.Cx(row + 3 + i, nv + k) = sa[i];
              }
            }
            if (fs == 3)
              o.Cx(row + 3, k) = vv[2];
          }
        }
      if (s.torque_limits)
        for (int i = 0; i < nu; i++)
          o.Cu(i, i) = 1.0;
      if (s.kinematics_limits)
        for (int i = 0; i < nv - 6; i++)
          o.Cx(nu + i, 6 + i) = 1.0;
      R = cd.R;
    }

    // terminal cost = the kinodynamics one (state + 10 x centroidal): delegate to the restatement in orc_kino.hpp
    KinoModel terminal_model() const
    {
      KinoSettings ks;
      ks.timestep = s.timestep;
      ks.w_x = s.w_x;
      ks.w_cent = s.w_cent;
      ks.w_u = Mat(1, 1);
      ks.w_frame = s.w_frame;
      ks.w_centder = s.w_cent;
      return KinoModel(M, ks);
    }
    double term_eval(Rigid & R, const Vec & x_tgt, const double * x) const { return terminal_model().term_eval(R, x_tgt, x); }
    void term_deriv(Rigid & R, const Vec & x_tgt, const double * x, Vec & lx, Mat & Lxx) const
    {
      terminal_model().term_deriv(R, x_tgt, x, lx, Lxx);
    }
    // DCM terminal constraint of src/fulldynamics.cpp:433-446: the same residual as the kinodynamics one
    void term_cstr(Rigid & R, const double * x, const double ref[3], double tau, double * c, Mat * C) const
    {
      terminal_model().term_cstr(R, x, ref, tau, c, C);
    }
  };
} // namespace orc
