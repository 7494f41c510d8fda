// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_mpc_cent.hpp: the receding-horizon host state machine of the reference (orc_mpc.hpp cites it line by line)
// driving the centroidal OCP (orc_cent.hpp).  What differs from the kinodynamics case:
//   - the problem state is RobotDataHandler::getCentroidalState() of the measured multibody state
//     (src/mpc.cpp:200, src/centroidal-dynamics.cpp:261-264, src/robot-handler.cpp:142-149): [com; hg.linear; hg.angular]
//   - setReferencePose writes the CONTACT POSITION of the stage's contact map (dynamics + both acceleration residuals,
//     src/centroidal-dynamics.cpp:151-169): the swing / foothold references of src/mpc.cpp:278-309 become p_i
//   - setReferenceState(t, x_ref) = setPoseBase(x_ref[0:3]) + setVelocityBase(x_ref[3:9]) and setVelocityBase stores
//     MOMENTUM references m v (src/centroidal-dynamics.cpp:227-239, 286-291)
//   - x_reference_ starts as getReferenceState(0) of the default problem = 0 (com_ref_ = 0, src/centroidal-dynamics.cpp:36)
//   - the default problem has identity contact poses, i.e. p_i = 0 (src/ocp-handler.cpp:117)
//   - no terminal constraint (src/centroidal-dynamics.cpp:318-337)
#pragma once
#include "orc_mpc.hpp"

namespace orc
{
  struct BatchMPCCent
  {
    const smpc_robot_model * M;
    CentModel md;
    MPCSettings st;
    int H, B, nf, nx_mb;
    bool walking = true;
    double velocity_base[6] = {0, 0, 0, 0, 0, 0};
    std::vector<std::array<double, 6>> vbase_inst;          // one command per instance
    std::vector<std::vector<std::array<double, 6>>> vref;   // momentum references [m v_lin; m v_ang] per horizon stage
    Vec x_reference; // MPC::x_reference_ (9)
    Vec x_model_ref; // multibody reference state
    V3 com0;
    CycleTimer timer;
    std::vector<StageRef> horizon, cycle_horizon;
    StageRef standing_stage;
    std::vector<OcpInstance> ocp;
    std::vector<SolverState> sol;
    std::vector<std::vector<FootTraj>> ftraj;
    std::vector<IterInfo> last_info;
    std::vector<IterInfo> cold_trace;
    bool keep_knots = false;
    std::vector<std::vector<Knot>> last_knots;

    static Vec centroidal_state(Rigid & R, const smpc_robot_model * m, const double * x)
    {
      R.fk(x);
      R.velocities(x + m->nq);
      const SV hg = R.hg();
      return Vec{R.com[0], R.com[1], R.com[2], hg.l[0], hg.l[1], hg.l[2], hg.a[0], hg.a[1], hg.a[2]};
    }

    BatchMPCCent(const smpc_robot_model * m, const CentSettings & cs, const MPCSettings & ms, int H_, int B_, double gravity_arg)
    : M(m), md(m, cs), st(ms), H(H_), B(B_)
    {
      nf = m->nfeet;
      nx_mb = m->nq + m->nv;
      x_model_ref.assign(nx_mb, 0.0);
      for (int i = 0; i < m->nq; i++)
        x_model_ref[i] = m->q_ref[i];
      x_reference.assign(9, 0.0);
      Rigid R(m);
      const Vec x0 = centroidal_state(R, m, x_model_ref.data());
      com0 = R.com;
      StageRef def;
      def.mask = (1u << nf) - 1u;
      def.u_ref.assign(md.nu, 0.0);
      for (int f = 0; f < nf; f++)
        def.u_ref[md.fs * f + 2] = -m->total_mass * gravity_arg / (double)nf;
      def.x_tgt.assign(9, 0.0);
      def.foot_ref.assign(nf, v3(0, 0, 0));
      horizon.assign(H, def);
      standing_stage = def;
      for (int f = 0; f < nf; f++)
        standing_stage.foot_ref[f] = R.foot_p[f];
      OcpInstance o;
      o.stages = horizon;
      o.x_tgt_term.assign(9, 0.0);
      SolverState s0;
      s0.xs.assign(H + 1, x0);
      s0.us.assign(H, def.u_ref);
      s0.vs.assign(H, Vec(md.nc, 0.0));
      s0.lams.assign(H + 1, Vec(md.ndx, 0.0));
      ProxDDPT<CentModel> solver(md, st.mu_init);
      std::vector<Vec> vs_e = s0.vs, lams_e = s0.lams;
      for (int it = 0; it < 100; it++)
      {
        IterInfo info = solver.iterate(R, o, s0, vs_e, lams_e);
        cold_trace.push_back(info);
        if (std::fmax(info.prim_infeas, info.dual_infeas) <= st.TOL)
          break;
        if (std::fabs(info.dphi0) <= SolverConsts::STALL_REL * std::fmax(1.0, std::fabs(info.phi0)))
          break;
        if (info.dual_infeas <= st.TOL)
        {
          vs_e = s0.vs;
          lams_e = s0.lams;
        }
      }
      ocp.assign(B, o);
      sol.assign(B, s0);
      ftraj.assign(B, std::vector<FootTraj>(nf));
      for (int b = 0; b < B; b++)
        for (int f = 0; f < nf; f++)
          ftraj[b][f] = FootTraj{R.foot_p[f], R.foot_p[f]};
      last_info.resize(B);
      vbase_inst.assign(B, std::array<double, 6>{{0, 0, 0, 0, 0, 0}});
      vref.assign(B, std::vector<std::array<double, 6>>(H, std::array<double, 6>{{0, 0, 0, 0, 0, 0}}));
    }
    void set_velocity_all(const double * v6)
    {
      for (auto & v : vbase_inst)
        for (int i = 0; i < 6; i++)
          v[i] = v6[i];
    }
    void setVelocityBaseBatched(const double * V)
    {
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 6; i++)
          vbase_inst[b][i] = V[(size_t)b * 6 + i];
    }

    void generateCycleHorizon(const std::vector<std::vector<char>> & cs)
    {
      timer.generate(cs, H, nf);
      Rigid R(M);
      R.fk(x_model_ref.data());
      cycle_horizon.clear();
      for (auto & state : timer.contact_states)
      {
        int active = 0;
        for (int f = 0; f < nf; f++)
          active += state[f] ? 1 : 0;
        StageRef sr;
        sr.mask = 0;
        sr.u_ref.assign(md.nu, 0.0);
        for (int f = 0; f < nf; f++)
          if (state[f])
          {
            sr.mask |= 1u << f;
            sr.u_ref[md.fs * f + 2] = st.support_force / (double)active;
          }
        // com_ref_ of the OCP at creation time: the last setPoseBase call, i.e. x_reference_[0:3] once a control step
        // has run (src/centroidal-dynamics.cpp:249-257); momentum references start at zero
        sr.x_tgt.assign(9, 0.0);
        for (int i = 0; i < 3; i++)
          sr.x_tgt[i] = com_ref_member[i];
        sr.foot_ref.resize(nf);
        for (int f = 0; f < nf; f++)
          sr.foot_ref[f] = R.foot_p[f];
        cycle_horizon.push_back(sr);
      }
    }
    double com_ref_member[3] = {0, 0, 0}; // CentroidalOCP::com_ref_

    void switchToWalk(const double * v6)
    {
      walking = true;
      for (int i = 0; i < 6; i++)
        velocity_base[i] = v6[i];
      set_velocity_all(v6);
    }
    void switchToStand()
    {
      walking = false;
      for (int i = 0; i < 6; i++)
        velocity_base[i] = 0;
      set_velocity_all(velocity_base);
    }

    // X: [B][nq+nv] measured multibody states
    void iterate(const double * X)
    {
      int last_support = 0;
      for (int f = 0; f < nf; f++)
        last_support += (horizon[H - 1].mask >> f) & 1u;
      StageRef incoming;
      if (walking || last_support < nf)
      {
        incoming = cycle_horizon[0];
        std::rotate(cycle_horizon.begin(), cycle_horizon.begin() + 1, cycle_horizon.end());
        timer.recede_cycle();
      }
      else
      {
        incoming = standing_stage;
        timer.update_timing(true);
      }
      horizon.erase(horizon.begin());
      horizon.push_back(incoming);
      // setReferenceState(H-1, x_reference_) then setVelocityBase(H-1, velocity_base_): momentum references are m v
      for (int i = 0; i < 3; i++)
      {
        horizon[H - 1].x_tgt[i] = x_reference[i];
        com_ref_member[i] = x_reference[i];
        horizon[H - 1].x_tgt[3 + i] = md.mass * velocity_base[i];
        horizon[H - 1].x_tgt[6 + i] = md.mass * velocity_base[3 + i];
      }
      if (keep_knots)
        last_knots.resize(B);

      ProxDDPT<CentModel> solver(md, st.mu_init);
#pragma omp parallel for schedule(dynamic)
      for (int b = 0; b < B; b++)
      {
        Rigid R(M);
        const double * x = X + (size_t)b * nx_mb;
        const Vec x0 = centroidal_state(R, M, x);
        OcpInstance & o = ocp[b];
        SolverState & S = sol[b];
        o.stages.erase(o.stages.begin());
        o.stages.push_back(incoming);
        for (int t = 0; t < H; t++)
        {
          o.stages[t].mask = horizon[t].mask;
          o.stages[t].u_ref = horizon[t].u_ref;
          o.stages[t].x_tgt = horizon[t].x_tgt;
        }
        const std::array<double, 6> & vb = vbase_inst[b];
        std::array<double, 6> mv;
        for (int i = 0; i < 6; i++)
          mv[i] = md.mass * vb[i];
        vref[b].erase(vref[b].begin());
        vref[b].push_back(mv);
        for (int t = 0; t < H; t++)
          for (int i = 0; i < 6; i++)
            o.stages[t].x_tgt[3 + i] = vref[b][t][i];
        S.vs.erase(S.vs.begin());
        S.vs.push_back(Vec(md.nc, 0.0));
        S.lams.erase(S.lams.begin() + 1);
        S.lams.push_back(Vec(md.ndx, 0.0));
        const V3 base_p = R.oMi[0].p;
        for (int f = 0; f < nf; f++)
        {
          int land = -1;
          if (!timer.land[f].empty())
            land = timer.land[f][0];
          const bool update = !(land < st.T_fly);
          V3 refp = R.oMi[0].R * v3(M->foot_ref_p[f][0], M->foot_ref_p[f][1], M->foot_ref_p[f][2]) + base_p;
          double tw0 = -(refp[1] - base_p[1]);
          double tw1 = refp[0] - base_p[0];
          V3 next;
          const double span = (double)(st.T_fly + st.T_contact) * st.timestep;
          next[0] = refp[0] + (vb[0] + vb[5] * tw0) * span;
          next[1] = refp[1] + (vb[1] + vb[5] * tw1) * span;
          next[2] = R.foot_p[f][2];
          FootTraj & ft = ftraj[b][f];
          if (update)
          {
            ft.initial = R.foot_p[f];
            ft.final_ = next;
          }
          for (int k = 0; k < H; k++)
          {
            const int t = land - k;
            V3 p;
            if (t < 0)
              p = ft.final_;
            else if (t > st.T_fly)
              p = ft.initial;
            else
              p = bezier8(ft.initial, ft.final_, st.swing_apex, float(st.T_fly - t) / float(st.T_fly));
            o.stages[k].foot_ref[f] = p;
          }
        }
        S.xs.erase(S.xs.begin());
        S.xs[0] = x0;
        S.xs.push_back(S.xs.back());
        S.us.erase(S.us.begin());
        S.us.push_back(S.us.back());
        std::vector<Vec> vs_e = S.vs, lams_e = S.lams;
        S.preg = SolverConsts::REG_INIT;
        for (int it = 0; it < st.max_iters; it++)
          last_info[b] = solver.iterate(R, o, S, vs_e, lams_e, keep_knots ? &last_knots[b] : nullptr);
      }
    }
  };
} // namespace orc
