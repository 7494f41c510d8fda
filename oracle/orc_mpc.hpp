// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_mpc.hpp: the receding-horizon host state machine of the reference, restated for a batch of
// phase-aligned instances:
//   MPC::MPC                         src/mpc.cpp:19-99
//   MPC::generateCycleHorizon        src/mpc.cpp:101-187
//   MPC::iterate                     src/mpc.cpp:189-218
//   MPC::recedeWithCycle             src/mpc.cpp:220-254
//   MPC::updateCycleTiming           src/mpc.cpp:256-276
//   MPC::updateStepTrackerReferences src/mpc.cpp:278-324
//   FootTrajectory                   src/foot-trajectory.cpp:20-96 (degree-8 Bezier, float parameter)
//   OCPHandler::createProblem        src/ocp-handler.cpp:96-137 (default all-contact horizon)
#pragma once
#include "orc_proxddp.hpp"
#include <algorithm>
#include <array>

namespace orc
{
  struct MPCSettings // include/simple-mpc/mpc.hpp:29-49
  {
    double swing_apex = 0.15, support_force = 1000, TOL = 1e-4, mu_init = 1e-8;
    int max_iters = 1, num_threads = 2, T_fly = 80, T_contact = 20;
    int T = 100;
    double timestep = 0.01;
  };

  // Integer gait bookkeeping (src/mpc.cpp:101-132, 220-276). Robot independent.
  struct CycleTimer
  {
    int H = 0, nf = 0;
    std::vector<std::vector<char>> contact_states; // extended cycle, [i][foot]
    std::vector<std::vector<int>> takeoff, land;   // per foot

    void generate(const std::vector<std::vector<char>> & cs, int H_, int nf_)
    {
      H = H_;
      nf = nf_;
      contact_states = cs;
      const int m = H / (int)cs.size();
      for (int i = 0; i < m; i++)
        contact_states.insert(contact_states.end(), cs.begin(), cs.end());
      takeoff.assign(nf, std::vector<int>());
      land.assign(nf, std::vector<int>());
      const int n = (int)contact_states.size();
      for (int f = 0; f < nf; f++)
      {
        for (int i = 1; i < n; i++)
        {
          if (!contact_states[i][f] && contact_states[i - 1][f])
            takeoff[f].push_back(i + H);
          if (contact_states[i][f] && !contact_states[i - 1][f])
            land[f].push_back(i + H);
        }
        if (contact_states[n - 1][f] && !contact_states[0][f])
          takeoff[f].push_back(n - 1 + H);
        if (!contact_states[n - 1][f] && contact_states[0][f])
          land[f].push_back(n - 1 + H);
      }
    }
    void update_timing(bool only_horizon)
    {
      for (int f = 0; f < nf; f++)
      {
        for (auto & t : land[f])
          if (!only_horizon || t < H)
            t -= 1;
        if (!land[f].empty() && land[f][0] < 0)
          land[f].erase(land[f].begin());
        for (auto & t : takeoff[f])
          if (!only_horizon || t < H)
            t -= 1;
        if (!takeoff[f].empty() && takeoff[f][0] < 0)
          takeoff[f].erase(takeoff[f].begin());
      }
    }
    // walking branch of recedeWithCycle: rotate the cycle and push new switch times
    void recede_cycle()
    {
      std::rotate(contact_states.begin(), contact_states.begin() + 1, contact_states.end());
      const int n = (int)contact_states.size();
      for (int f = 0; f < nf; f++)
      {
        if (!contact_states[n - 1][f] && contact_states[n - 2][f])
          takeoff[f].push_back(n + H);
        if (contact_states[n - 1][f] && !contact_states[n - 2][f])
          land[f].push_back(n + H);
      }
      update_timing(false);
    }
  };

  // Degree-8 Bezier swing curve (src/foot-trajectory.cpp:41-62) evaluated with ndcurves' Horner
  // scheme [UPSTREAM-RECALL ndcurves bezier_curve::evalHorner]; the curve parameter is computed in
  // single precision (src/foot-trajectory.cpp:18,76).
  inline V3 bezier8(const V3 & p0, const V3 & p1, double apex, float tf)
  {
    V3 mid = 0.75 * p0 + 0.25 * p1;
    mid[2] += apex;
    const V3 * cp[9] = {&p0, &p0, &p0, &p0, &mid, &p1, &p1, &p1, &p1};
    const double u = (double)tf;
    const double uo = 1.0 - u;
    double bc = 1.0, tn = 1.0;
    V3 tmp = uo * (*cp[0]);
    for (int i = 1; i < 8; i++)
    {
      tn = tn * u;
      bc = bc * (double)(8 - i + 1) / (double)i;
      tmp = uo * (tmp + (tn * bc) * (*cp[i]));
    }
    return tmp + (tn * u) * (*cp[8]);
  }

  struct FootTraj // per instance, per foot
  {
    V3 initial, final_;
  };

  // Model = KinoModel (orc_kino.hpp) or FullModel (orc_fulldyn.hpp): the host state machine is the same; the models say where a
  // stage keeps its force references (Model::force_ref_index) and how long its reference vector is (Model::n_uref)
  template <class Model, class ModelSettings>
  struct BatchMPCT
  {
    const smpc_robot_model * M;
    Model md;
    MPCSettings st;
    int H, B, nf;
    bool walking = true;
    double velocity_base[6] = {0, 0, 0, 0, 0, 0};
    // one velocity command per instance (the reference's MPC has one velocity_base_; a batch = B such objects) and the
    // velocity part of the state_cost target each horizon stage received when it entered (setVelocityBase, src/mpc.cpp:312)
    std::vector<std::array<double, 6>> vbase_inst;
    std::vector<std::vector<std::array<double, 6>>> vref;
    Vec x_reference;  // MPC::x_reference_
    Vec x_model_ref;  // model reference state
    V3 com0;
    CycleTimer timer;
    // shared stage descriptors (mask, u_ref, x_tgt); foot refs are per instance
    std::vector<StageRef> horizon;       // H, current problem stages (shared fields)
    std::vector<StageRef> cycle_horizon; // extended cycle
    StageRef standing_stage;
    std::vector<OcpInstance> ocp;        // per instance (copies of horizon + own foot refs)
    std::vector<SolverState> sol;
    std::vector<std::vector<FootTraj>> ftraj; // [b][f]
    std::vector<IterInfo> last_info;
    bool keep_knots = false;
    std::vector<std::vector<Knot>> last_knots; // [b][t], filled when keep_knots (tests only)

    bool terminal_constraint = false;
    BatchMPCT(const smpc_robot_model * m, const ModelSettings & ks, const MPCSettings & ms, int H_, int B_, double gravity_arg,
              bool terminal_constraint_ = false)
    : M(m), md(m, ks), st(ms), H(H_), B(B_), terminal_constraint(terminal_constraint_)
    {
      nf = m->nfeet;
      x_model_ref.assign(md.nx, 0.0);
      for (int i = 0; i < md.nq; i++)
        x_model_ref[i] = m->q_ref[i];
      x_reference = x_model_ref;
      Rigid R(m);
      R.fk(x_model_ref.data());
      com0 = R.com;
      const std::vector<V3> ref_feet(R.foot_p.begin(), R.foot_p.begin() + nf); // FootTrajectory starting poses: the feet at the reference state (src/mpc.cpp:26-36)
      // default problem (OCPHandler::createProblem): all contacts, identity contact poses
      StageRef def;
      def.mask = (1u << nf) - 1u;
      def.u_ref.assign(md.n_uref(), 0.0);
      for (int f = 0; f < nf; f++)
        def.u_ref[md.force_ref_index(f) + 2] = -m->total_mass * gravity_arg / (double)nf;
      def.x_tgt = x_model_ref;
      def.foot_ref.assign(nf, v3(0, 0, 0));
      horizon.assign(H, def);
      standing_stage = def; // force ref = getReferenceForce(0, foot0) (src/mpc.cpp:57)
      for (int f = 0; f < nf; f++)
        standing_stage.foot_ref[f] = R.foot_p[f];
      // cold solve once (all instances share x0 = reference state) -- src/mpc.cpp:72-91
      OcpInstance o;
      o.stages = horizon;
      o.x_tgt_term = x_model_ref;
      if (terminal_constraint)
      { // createTerminalConstraint(x0.head<3>()): the reference is the base position until the first iterate (src/ocp-handler.cpp:133-136)
        o.term_cstr = true;
        for (int i = 0; i < 3; i++)
          o.dcm_ref[i] = x_model_ref[i];
        o.dcm_tau = std::sqrt(x_model_ref[2] / 9.81);
      }
      SolverState s0;
      s0.vN.assign(3, 0.0);
      s0.xs.assign(H + 1, x_model_ref);
      s0.us.assign(H, Vec(def.u_ref.begin(), def.u_ref.begin() + md.nu)); // getReferenceControl(0) (src/mpc.cpp:75)
      s0.vs.assign(H, Vec(md.nc, 0.0));
      s0.lams.assign(H + 1, Vec(md.ndx, 0.0));
      ProxDDPT<Model> solver(md, st.mu_init);
      std::vector<Vec> vs_e = s0.vs, lams_e = s0.lams;
      Vec vN_e = s0.vN;
      for (int it = 0; it < 100; it++)
      {
        IterInfo info = solver.iterate(R, o, s0, vs_e, lams_e, nullptr, &vN_e);
        cold_trace.push_back(info);
        if (std::fmax(info.prim_infeas, info.dual_infeas) <= st.TOL)
          break;
        // stalled: the predicted merit decrease is below what FP64 can resolve (cf. Aligator's
        // ls_params.dphi_thresh early exit) -- DESIGN.md "solver constants"
        if (std::fabs(info.dphi0) <= SolverConsts::STALL_REL * std::fmax(1.0, std::fabs(info.phi0)))
          break;
        if (info.dual_infeas <= st.TOL)
        {
          vs_e = s0.vs;
          lams_e = s0.lams;
          vN_e = s0.vN;
        }
      }
      ocp.assign(B, o);
      sol.assign(B, s0);
      ftraj.assign(B, std::vector<FootTraj>(nf));
      for (int b = 0; b < B; b++)
        for (int f = 0; f < nf; f++)
          ftraj[b][f] = FootTraj{ref_feet[f], ref_feet[f]}; // (not R.foot_p: the cold solve above has moved R to its last trial point)
      last_info.resize(B);
      vbase_inst.assign(B, std::array<double, 6>{{0, 0, 0, 0, 0, 0}});
      vref.assign(B, std::vector<std::array<double, 6>>(H, std::array<double, 6>{{0, 0, 0, 0, 0, 0}}));
    }
    std::vector<IterInfo> cold_trace;
    bool early_exit_on_tol = false; // SolverProxDDP::run's convergence test inside iterate (reference src/mpc.cpp:43,212)
    void set_velocity_all(const double * v6)
    {
      for (auto & v : vbase_inst)
        for (int i = 0; i < 6; i++)
          v[i] = v6[i];
    }
    void setVelocityBaseBatched(const double * V) // [B][6]
    {
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 6; i++)
          vbase_inst[b][i] = V[(size_t)b * 6 + i];
    }

    void generateCycleHorizon(const std::vector<std::vector<char>> & cs)
    {
      timer.generate(cs, H, nf);
      Rigid R(M);
      R.fk(x_model_ref.data()); // data handler still holds the reference state (src/mpc.cpp:26,162)
      cycle_horizon.clear();
      std::vector<char> previous(nf, 1); // land flags: in contact here, not in the stage before (src/mpc.cpp:133-137,167-185)
      for (auto & state : timer.contact_states)
      {
        int active = 0;
        for (int f = 0; f < nf; f++)
          active += state[f] ? 1 : 0;
        StageRef sr;
        sr.mask = 0;
        sr.u_ref.assign(md.n_uref(), 0.0);
        for (int f = 0; f < nf; f++)
          if (state[f])
          {
            sr.mask |= 1u << f;
            sr.u_ref[md.force_ref_index(f) + 2] = st.support_force / (double)active;
          }
        sr.x_tgt = x_model_ref;
        sr.foot_ref.resize(nf);
        for (int f = 0; f < nf; f++)
          sr.foot_ref[f] = R.foot_p[f];
        sr.land = 0;
        for (int f = 0; f < nf; f++)
        {
          if (state[f] && !previous[f])
            sr.land |= 1u << f;
          previous[f] = state[f] ? 1 : 0;
        }
        cycle_horizon.push_back(sr);
      }
    }

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

    // X: [B][nx] measured states
    void iterate(const double * X)
    {
      // ---- recedeWithCycle (shared part) ----
      int last_support = 0;
      for (int f = 0; f < nf; f++)
        last_support += (horizon[H - 1].mask >> f) & 1u;
      StageRef incoming;
      const bool cyc = walking || last_support < nf;
      if (cyc)
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
      // setReferenceState(H-1, x_reference_) ; setVelocityBase(H-1, velocity_base_)
      horizon[H - 1].x_tgt = x_reference;
      for (int i = 0; i < 6; i++)
        horizon[H - 1].x_tgt[md.nq + i] = velocity_base[i];

      ProxDDPT<Model> solver(md, st.mu_init);
#pragma omp parallel for schedule(dynamic)
      for (int b = 0; b < B; b++)
      {
        Rigid R(M);
        const double * x = X + (size_t)b * md.nx;
        R.fk(x); // updateInternalData(x, false): FK + frames (+ hg, unused here)
        OcpInstance & o = ocp[b];
        SolverState & S = sol[b];
        // rotate per-instance stages and multipliers (replaceStageCircular + cycleProblem)
        o.stages.erase(o.stages.begin());
        o.stages.push_back(incoming);
        for (int t = 0; t < H; t++)
        {
          o.stages[t].mask = horizon[t].mask;
          o.stages[t].land = horizon[t].land;
          o.stages[t].u_ref = horizon[t].u_ref;
          o.stages[t].x_tgt = horizon[t].x_tgt;
        }
        const std::array<double, 6> & vb = vbase_inst[b];
        vref[b].erase(vref[b].begin());
        vref[b].push_back(vb);
        for (int t = 0; t < H; t++)
          for (int i = 0; i < 6; i++)
            o.stages[t].x_tgt[md.nq + i] = vref[b][t][i];
        S.vs.erase(S.vs.begin());
        S.vs.push_back(Vec(md.nc, 0.0));
        S.lams.erase(S.lams.begin() + 1);
        S.lams.push_back(Vec(md.ndx, 0.0));
        // ---- updateStepTrackerReferences ----
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
        if (o.term_cstr)
        { // updateTerminalConstraint: mean of the last foot references, at the CoM height of the reference state (src/mpc.cpp:313-323)
          V3 cr = v3(0, 0, 0);
          for (int f = 0; f < nf; f++)
            cr = cr + o.stages[H - 1].foot_ref[f];
          for (int i = 0; i < 3; i++)
            o.dcm_ref[i] = cr[i] / (double)nf;
          o.dcm_ref[2] += com0[2];
        }
        // ---- warm start shift (src/mpc.cpp:201-207) ----
        S.xs.erase(S.xs.begin());
        S.xs[0].assign(x, x + md.nx);
        S.xs.push_back(S.xs.back());
        S.us.erase(S.us.begin());
        S.us.push_back(S.us.back());
        // ---- solver run: max_iters iterations, centres = incoming multipliers ----
        std::vector<Vec> vs_e = S.vs, lams_e = S.lams;
        const Vec vN_e = S.vN;
        S.preg = SolverConsts::REG_INIT; // regularisation restarts with every solver run
        for (int it = 0; it < st.max_iters; it++)
        {
          last_info[b] = solver.iterate(R, o, S, vs_e, lams_e, keep_knots ? &last_knots[b] : nullptr, &vN_e, early_exit_on_tol ? st.TOL : -1.0);
          if (early_exit_on_tol && std::fmax(last_info[b].prim_infeas, last_info[b].dual_infeas) <= st.TOL)
            break;
        }
      }
    }
  };
  typedef BatchMPCT<KinoModel, KinoSettings> BatchMPC;
} // namespace orc
