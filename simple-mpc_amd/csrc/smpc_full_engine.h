// smpc_full_engine.h -- host side of the batched FULL-DYNAMICS MPC engine (reference FullDynamicsOCP under the MPC class:
// src/fulldynamics.cpp:30-455, src/mpc.cpp:19-392).  Same state machine, ring discipline and launch sequence as KinoEngine
// (smpc_engine.h); the stage kernels are fdyn_deriv_body / fdyn_trial_body (smpc_full_stage.h), the sweeps are dense
// (riccati_dense_body on the matrix cores, or the VALU cross-check riccati_full_body with SMPC_RICCATI=valu).
#pragma once
#include <chrono>
#include "smpc_engine.h"
#include "smpc_full_solver.h"
#include "smpc_riccati_dense.h"
#include "smpc_full_stage.h"

namespace smpc
{
  struct HostFullSettings // FullDynamicsSettings, include/simple-mpc/fulldynamics.hpp:28-65
  {
    double timestep;
    std::vector<double> w_x, w_u, w_cent, w_forces, w_frame, umin, umax, qmin, qmax, Kp, Kd;
    std::vector<double> w_centder; // kinodynamics variant (FullDims<..., KIN = 1>): KinodynamicsSettings::w_centder
    double gravity[3];
    double mu, Lfoot, Wfoot;
    int force_size, torque_limits, kinematics_limits, force_cone, land_cstr;
    int terminal_constraint = 0; // createProblem(..., terminal_constraint)
  };

  // what the C ABI needs from a full-dynamics engine of any robot shape
  struct FullEngineBase
  {
    int B = 0, H = 0, R = 0, head = 0;
    int device_id = 0;
    int dims[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // nq nv nx ndx nu nc nf H
    int force_size = 3;
    GaitTimer timer;
    stream_t stream;
    bool profiling = false;
    double kernel_ms[KID_N] = {0};
    long kernel_calls[KID_N] = {0};
    int cold_iters = 0;
    std::vector<double> cold_trace;
    std::vector<double> x_reference;
    virtual void set_early_exit(bool on) = 0;
    virtual ~FullEngineBase() {}
    virtual void generate_cycle_horizon(const unsigned char * cs, int n) = 0;
    virtual void switch_to_walk(const double * v6) = 0;
    virtual void switch_to_stand() = 0;
    virtual void set_velocity_base_batched(const double * V) = 0;
    virtual void iterate_host(const double * X) = 0;
    virtual void iterate_device(const double * Xd) = 0;
    virtual void sync() = 0;
    virtual void gather_x_device(int t, double * out_dev) = 0;
    virtual void get(int what, double * out) = 0; // 0 xs 1 us 2 K0 3 Ks 4 vs 5 lams 6 xdot01 7 foot refs 8 info 9 contact forces
    virtual void set_stage_reference(int t, int what, const double * v, int n) = 0;
    virtual void get_stage_reference(int t, int what, double * v, int n) = 0;
    virtual void set_reference_pose(int t, int foot, const double * p3) = 0;
    virtual void get_reference_pose(int t, int foot, int inst, double * p3) = 0;
    virtual void set_reference_rotation(int t, int foot, const double * R9) = 0;
    virtual void get_reference_rotation(int t, int foot, double * R9) = 0;
    virtual unsigned contact_mask(int t) const = 0;
    virtual void update_internal_data(const double * X, double * feet, double * com, double * hg, double * cstate) = 0;
    virtual void full_forward_dynamics(int n, const double * X, const double * tau, const unsigned * mask, const double * Kp, const double * Kd,
                                       double prox_accuracy, double prox_mu, int prox_max_iter, double * a, double * lam, int * iters, double * kernel_ms) = 0;
    virtual size_t state_io(StateIO & io) = 0;
    virtual void collect_profile() = 0;
    virtual int lq_size() const = 0;
    virtual void debug_lq(int inst, int t, double * out) = 0;
    virtual void debug_steps(double * dxs, double * dus) = 0;
    virtual void debug_terminal(int inst, double * QN, double * qN) = 0;
    virtual bool phase_cycles(double * out64) = 0;
    virtual void interpolate(double delay, int knots, double * x_out, double * acc_out, double * f_out) = 0;
    // device-resident forms (nothing crosses the host between two MPC steps): the interpolated targets written into the
    // inverse-dynamics engine's device buffers, the simulated robot stepped in place, ordering with another stream
    virtual void interpolate_device(double delay, int knots, double * x_dev, double * acc_dev, double * f_dev) = 0;
    virtual void sim_step_device(double * X_dev, const double * tau_dev, unsigned mask, const double * Kp, const double * Kd, double dt) = 0;
    virtual void wait_stream(stream_t other) = 0;
    virtual void riccati_feedback(double delay, const double * X, double * u_out) = 0;
  };

  template <class D>
  class FullEngine : public FullEngineBase
  {
  public:
    Buffers<D> buf;
    double * deriv_wide = nullptr; // D::WIDE_DEV: R1 / JT slices of the derivative kernel's blocks (FullDerivWide, smpc_full_stage.h)
    int n_res = 0;                 // blocks of its persistent grid (compute units x resident blocks per unit)
    // The batch as parts on streams of their own (round 5; SMPC_FULL_PARTS=n, default 1): one wavefront per instance is all the sweeps have, so
    // at B = 1024 riccati_dense_body runs one wave per SIMD for its whole duration; with two parts the sweep of one could run beside the stage
    // kernel of the other.  Measured on the biped (B = 1024, H = 100): 10.77 k control-steps/s with 1 part, 10.29 k with 2, 10.43 k with 3 --
    // three derivative blocks fill a CU's LDS (3 x 53 KB), a sweep block (36 KB) finds no room beside them and the two kernels take turns as
    // before; the quadruped (B = 4096): 70.5 k -> 71.3 k.  Kept off.  Instances are independent: bit-identical results (tests).
    static constexpr int MAX_PARTS = 4;
    int n_parts = 1;
    stream_t cur{};                      // the stream the launches of the moment go to
    stream_t part_stream[MAX_PARTS] = {}; // [0] = stream
    event_t ev_fork{}, ev_join[MAX_PARTS] = {};
    int * und_part[MAX_PARTS] = {nullptr, nullptr, nullptr, nullptr};
    HostMpcSettings ms;
    std::vector<StageShared<D>> horizon, cycle;
    StageShared<D> standing;
    bool walking = true;
    double velocity_base[6] = {0, 0, 0, 0, 0, 0};
    std::vector<double> x_model_ref;
    double * X_dev = nullptr;
    double * stage_out = nullptr;
    size_t stage_out_bytes = 0;
    bool valu_riccati = xcheck_env("SMPC_RICCATI") && std::string(xcheck_env("SMPC_RICCATI")) == "valu";
    bool speculative_ls = xcheck_env("SMPC_NO_SPECULATIVE_LS") == nullptr;
    bool early_exit_on_tol = false; // smpc_set_early_exit_on_tol
    void set_early_exit(bool on) override { early_exit_on_tol = on; }
    bool aux_launches = false;
    std::vector<std::pair<int, std::pair<event_t, event_t>>> pending_events;
    static constexpr int LS_SLOTS = 64;
#ifndef SMPC_FDYN_DERIV_MINW
#define SMPC_FDYN_DERIV_MINW 1
#endif
    static constexpr int DERIV_MINW = D::NV <= 20 ? SMPC_FDYN_DERIV_MINW : 1; // (register-cap experiment: -DSMPC_FDYN_DERIV_MINW=2)
    static constexpr int TRIAL_MINW = D::NV <= 20 ? 2 : 1; // waves per SIMD the evaluation kernel's register budget is capped for
    static constexpr double ARMIJO_C1 = 1e-4, REG_INIT = 1e-9, REG_MIN = 1e-10, REG_MAX = 1e9, REG_INC = 10.0, REG_DEC = 1.0 / 3.0, STALL_REL = 1e-9;
    double ref_foot_pos[D::NF][3];

    FullEngine(const smpc_robot_model * rm, const HostFullSettings & fs, const HostMpcSettings & ms_, int batch, double gravity_arg, int device)
    : ms(ms_)
    {
      AllocScope ctor_scope; // (a throw below releases what was allocated so far: smpc_alloc_scope.h)
      if (rm->njoints != D::NJ || rm->nfeet != D::NF)
        throw std::runtime_error("robot shape (njoints, nfeet) does not match this kernel instantiation");
      if (fs.force_size != D::FS)
        throw std::runtime_error("force size in settings does not match reference force size");
      if (batch <= 0)
        throw std::runtime_error("batch must be positive");
      if ((int)fs.Kp.size() != D::FS)
        throw std::runtime_error("Force must be of same size as Kp correction"); // src/fulldynamics.cpp:41-44
      if ((int)fs.Kd.size() != D::FS)
        throw std::runtime_error("Force must be of same size as Kd correction"); // src/fulldynamics.cpp:45-48
      if ((int)fs.w_x.size() != D::NDX * D::NDX || (int)fs.w_u.size() != D::NU * D::NU || (int)fs.w_cent.size() != 36
          || (int)fs.w_forces.size() != D::FS * D::FS || (int)fs.w_frame.size() != D::FS * D::FS || (int)fs.umin.size() != D::NU
          || (int)fs.umax.size() != D::NU || (int)fs.qmin.size() != D::NA || (int)fs.qmax.size() != D::NA)
        throw std::runtime_error("full-dynamics settings: weight / limit sizes do not match the robot");
      if (fs.land_cstr && D::NLAND1 == 0)
        throw std::runtime_error("internal: land_cstr needs the instantiation with land rows");
      if (fs.force_cone && D::NCONE1 == 0)
        throw std::runtime_error("internal: force_cone needs the instantiation with cone rows");
      device_id = device;
      set_device(device);
      stream = stream_create();
      cur = stream;
      part_stream[0] = stream;
      B = batch;
      H = ms.T;
      R = H + 1;
      force_size = D::FS;
      const int dd[8] = {D::NQ, D::NV, D::NX, D::NDX, D::NU, D::NC, D::NF, H};
      std::copy(dd, dd + 8, dims);
      // ---- model table ----
      std::vector<DevModel<D>> hm(1);
      DevModel<D> & m = hm[0];
      std::memset(&m, 0, sizeof(m));
      fill_tree_model<D>(rm, m);
      m.dt = fs.timestep;
      for (int i = 0; i < 3; i++)
        m.gravity[i] = fs.gravity[i];
      std::copy(fs.w_x.begin(), fs.w_x.end(), m.w_x);
      std::copy(fs.w_u.begin(), fs.w_u.end(), m.w_u);
      m.w_diag = 1;
      for (int i = 0; i < D::NDX; i++)
        for (int j = 0; j < D::NDX; j++)
          if (i != j && m.w_x[i * D::NDX + j] != 0.0)
            m.w_diag = 0;
      for (int i = 0; i < D::NU; i++)
        for (int j = 0; j < D::NU; j++)
          if (i != j && m.w_u[i * D::NU + j] != 0.0)
            m.w_diag = 0;
      if (xcheck_env("SMPC_FORCE_DENSE_WEIGHTS"))
        m.w_diag = 0;
      for (int i = 0; i < D::NDX; i++)
        m.wxd[i] = m.w_x[i * D::NDX + i];
      for (int i = 0; i < D::NU; i++)
        m.wud[i] = m.w_u[i * D::NU + i];
      std::copy(fs.w_cent.begin(), fs.w_cent.end(), m.w_cent);
      std::copy(fs.w_forces.begin(), fs.w_forces.end(), m.w_forces);
      std::copy(fs.w_frame.begin(), fs.w_frame.end(), m.w_frame);
      if constexpr (D::KINO)
      {
        if ((int)fs.w_centder.size() != 36)
          throw std::runtime_error("kinodynamics settings: w_centder must be 6 x 6");
        std::copy(fs.w_centder.begin(), fs.w_centder.end(), m.w_centder);
      }
      std::copy(fs.Kp.begin(), fs.Kp.end(), m.Kp);
      std::copy(fs.Kd.begin(), fs.Kd.end(), m.Kd);
      std::copy(fs.umin.begin(), fs.umin.end(), m.umin);
      std::copy(fs.umax.begin(), fs.umax.end(), m.umax);
      std::copy(fs.qmin.begin(), fs.qmin.end(), m.qmin);
      std::copy(fs.qmax.begin(), fs.qmax.end(), m.qmax);
      m.fric_mu = fs.mu;
      m.Lfoot = fs.Lfoot;
      m.Wfoot = fs.Wfoot;
      m.prox_accuracy = 1e-9; // ProximalSettings(1e-9, 1e-10, 10), src/fulldynamics.cpp:39
      m.prox_mu = 1e-10;
      m.prox_max_iter = 10;
      m.torque_limits = fs.torque_limits;
      m.kinematics_limits = fs.kinematics_limits;
      m.force_cone = fs.force_cone;
      m.mu = ms.mu_init;
      x_model_ref.assign(D::NX, 0.0);
      for (int i = 0; i < D::NQ; i++)
        x_model_ref[i] = rm->q_ref[i];
      x_reference = x_model_ref;
      for (int i = 0; i < D::NX; i++)
        m.x_term[i] = x_model_ref[i];
      m.land_cstr = fs.land_cstr;
      {
        // contact poses of the cycle stages: the feet at the reference state (src/mpc.cpp:162); land_cstr pins the height of a landing 3-D foot to them
        std::vector<double> ft((size_t)D::NF * 6);
        host_foot_positions(m, x_model_ref.data(), ft.data());
        for (int f = 0; f < D::NF; f++)
          m.land_z[f] = ft[f * 6 + 2];
      }
      // ---- buffers ----
      buf.B = B;
      buf.H = H;
      buf.R = R;
      auto dalloc = [&](size_t n) { return (double *)dev_alloc(n * sizeof(double)); };
      const size_t BR = (size_t)B * R, BH = (size_t)B * H;
      buf.xs = dalloc(BR * D::NX);
      buf.us = dalloc(BR * D::NU);
      buf.vs = dalloc(BR * D::NC);
      buf.lams = dalloc(BR * D::NDX);
      buf.vs_e = dalloc(BR * D::NC);
      buf.lams_e = dalloc(BR * D::NDX);
      buf.xs_b = dalloc(BR * D::NX);
      buf.us_b = dalloc(BR * D::NU);
      buf.vs_b = dalloc(BR * D::NC);
      buf.lams_b = dalloc(BR * D::NDX);
      buf.dxs = dalloc((size_t)B * (H + 1) * D::NDX);
      buf.dus = dalloc(BH * D::NU);
      buf.dvs = dalloc(BH * D::NC);
      buf.dlams = dalloc(BH * D::NDX);
      buf.foot_ref = dalloc(BH * D::NF * 3);
      buf.ftraj = dalloc((size_t)B * D::NF * 6);
      buf.vbase = dalloc((size_t)B * 6);
      buf.vref = dalloc(BR * 6);
      buf.lq = dalloc(BH * D::LQ_STRIDE);
      buf.gains = dalloc(BH * (size_t)D::G_STRIDE);
      buf.QN = dalloc((size_t)B * D::NDX * D::NDX);
      buf.qN = dalloc((size_t)B * D::NDX);
      buf.parts0 = dalloc((size_t)B * (H + 1) * 4);
      buf.partsT = dalloc((size_t)B * D::LS_N * (H + 1) * 2);
      buf.scal = dalloc((size_t)B * SC_N);
      buf.xdotT = dalloc((size_t)B * D::LS_N * 4 * D::NV);
      buf.xdot01 = dalloc((size_t)B * 4 * D::NV);
      buf.nforce = D::NCM;
      buf.forcesT = dalloc(BH * D::LS_N * D::NCM);
      buf.forces = dalloc(BH * D::NCM);
      buf.ls_sel = (int *)dev_alloc((size_t)B * sizeof(int));
      buf.und_list = (int *)dev_alloc((size_t)(B + 1) * sizeof(int));
      buf.stages = (StageShared<D> *)dev_alloc((size_t)H * sizeof(StageShared<D>));
      buf.model = (DevModel<D> *)dev_alloc(sizeof(DevModel<D>));
      X_dev = dalloc((size_t)B * D::NX);
      {
        const char * pe = std::getenv("SMPC_FULL_PARTS");
        n_parts = pe ? std::atoi(pe) : 1;
        if (n_parts < 1 || n_parts > MAX_PARTS || B < 64 * n_parts)
          n_parts = 1;
        if constexpr (D::WIDE_DEV)
        {
          // one slice per RESIDENT block of fdyn_deriv_body (its grid is persistent: smpc_full_stage.h), per part of the batch: LDS decides how
          // many blocks a CU holds
          n_res = dev_cu_count(device_id) * (int)(160 * 1024 / sizeof(FullScratch<D, true>));
          deriv_wide = (double *)dev_alloc((size_t)n_res * n_parts * sizeof(FullDerivWide<D>));
        }
        if (n_parts > 1)
        {
          ev_fork = event_create();
          for (int i = 1; i < n_parts; i++)
          {
            part_stream[i] = stream_create();
            ev_join[i] = event_create();
          }
          for (int i = 0; i < n_parts; i++)
            und_part[i] = (int *)dev_alloc((size_t)(B + 1) * sizeof(int));
        }
      }
      if (fs.terminal_constraint)
        alloc_terminal_constraint<D>(buf, x_model_ref.data(), host_com_height(m, x_model_ref.data()), stream);
      if (std::getenv("SMPC_PHASE_PROFILE"))
        buf.dbg = dalloc(64);
      h2d(buf.model, hm.data(), sizeof(DevModel<D>), stream);
      stream_sync(stream);
      // ---- default problem (OCPHandler::createProblem, src/ocp-handler.cpp:96-137) ----
      StageShared<D> def;
      std::memset(&def, 0, sizeof(def));
      def.mask = (1u << D::NF) - 1u;
      for (int f = 0; f < D::NF; f++)
      {
        def.f_ref[D::FS * f + 2] = -rm->total_mass * gravity_arg / (double)D::NF;
        if constexpr (D::KINO) // the force references are the head of the control reference (computeControlFromForces, src/kinodynamics.cpp:229-240)
          def.u_ref[D::FS * f + 2] = def.f_ref[D::FS * f + 2];
      }
      for (int i = 0; i < D::NX; i++)
        def.x_tgt[i] = x_model_ref[i];
      horizon.assign(H, def);
      standing = def;
      cold_solve(def, m);
      ref_rot.init(H, D::NF);
      ctor_scope.commit();
    }
    ~FullEngine()
    {
      for (double * p : {buf.CN, buf.vN, buf.vN_e, buf.vN_b, buf.dvN, buf.dcm_ref})
        dev_free(p);
      for (double * p : {buf.xs_b, buf.us_b, buf.vs_b, buf.lams_b, buf.xs, buf.us, buf.vs, buf.lams, buf.vs_e, buf.lams_e, buf.dxs, buf.dus, buf.dvs, buf.dlams, buf.foot_ref,
                         buf.ftraj, buf.vbase, buf.vref, buf.lq, buf.gains, buf.QN, buf.qN, buf.parts0, buf.partsT, buf.scal, buf.xdotT, buf.xdot01, buf.forces,
                         buf.forcesT, X_dev, stage_out, deriv_wide})
        dev_free(p);
      dev_free(buf.ls_sel);
      dev_free(buf.und_list);
      for (int i = 0; i < MAX_PARTS; i++)
        dev_free(und_part[i]);
      if (n_parts > 1)
      {
        event_destroy(ev_fork);
        for (int i = 1; i < n_parts; i++)
        {
          event_destroy(ev_join[i]);
          stream_destroy(part_stream[i]);
        }
      }
      dev_free(buf.stages);
      dev_free(buf.model);
      dev_free(sim_a);
      dev_free(sim_lam);
      dev_free(sim_mask);
      if (ev_handoff_valid)
        event_destroy(ev_handoff);
      stream_destroy(stream);
    }
    FullEngine(const FullEngine &) = delete;
    FullEngine & operator=(const FullEngine &) = delete;

    SolverArgs<D> solver_args(const Buffers<D> & b, int j0 = 0, int nj = 0) const
    {
      SolverArgs<D> a;
      a.b = b;
      a.head = head;
      a.j0 = j0;
      a.nj = nj;
      a.armijo_c1 = ARMIJO_C1;
      a.reg_min = REG_MIN;
      a.reg_max = REG_MAX;
      a.reg_inc = REG_INC;
      a.reg_dec = REG_DEC;
      a.stop_tol = early_exit_on_tol ? ms.TOL : -1.0;
      return a;
    }
    template <class Args, void (*Body)(const Args &, int), int NT, int MINW = 1>
    void timed_launch(int kid, int grid, const Args & a, bool aux = false)
    {
      set_device(device_id);
      aux = aux || aux_launches;
      event_t e0{}, e1{};
      if (profiling)
      {
        e0 = event_create();
        e1 = event_create();
        event_record(e0, cur);
      }
      if (aux)
        launch<Args, Body, NT, MINW, 1>(grid, cur, a);
      else
        launch<Args, Body, NT, MINW, 0>(grid, cur, a);
      if (profiling)
      {
        event_record(e1, cur);
        pending_events.push_back({kid, {e0, e1}});
      }
      kernel_calls[kid]++;
    }
    void collect_profile() override
    {
      stream_sync(stream);
      for (auto & pe : pending_events)
      {
        kernel_ms[pe.first] += event_elapsed_ms(pe.second.first, pe.second.second);
        event_destroy(pe.second.first);
        event_destroy(pe.second.second);
      }
      pending_events.clear();
    }
    StageKernelArgs<D> stage_args(const Buffers<D> & b, int slots = 0) const
    {
      StageKernelArgs<D> sk;
      sk.b = b;
      sk.head = head;
      sk.j0 = 0;
      sk.nj = 0;
      sk.slots = slots;
      // (a view of a part of the batch: the slices of its blocks lie behind those of the parts before it)
      const long long i0 = (long long)((size_t)(b.xs - buf.xs) / ((size_t)R * D::NX));
      const int part = (int)((i0 * n_parts + B - 1) / B);
      sk.wide = deriv_wide ? deriv_wide + (size_t)part * n_res * (sizeof(FullDerivWide<D>) / sizeof(double)) : nullptr;
      return sk;
    }
    void launch_deriv(const Buffers<D> & b, int slots = 0)
    {
      StageKernelArgs<D> sk = stage_args(b, slots);
      int grid = (slots > 0 ? slots : b.B) * (H + 1);
      if constexpr (D::WIDE_DEV)
      {
        sk.nwork = grid;
        sk.nres = grid < n_res ? grid : n_res;
        grid = sk.nres;
      }
      timed_launch<StageKernelArgs<D>, fdyn_deriv_body<D>, 64, DERIV_MINW>(slots > 0 ? KID_SELECT : KID_DERIV, grid, sk, slots > 0);
    }
    void launch_sweeps(const Buffers<D> & b)
    {
      if constexpr (D::NCD == 0 && kCrossCheck)
      {
        if (valu_riccati)
        {
          timed_launch<SolverArgs<D>, riccati_full_body<D, 256>, 256>(KID_RICCATI, b.B, solver_args(b)); // cross-check (box rows only)
          timed_launch<SolverArgs<D>, forward_full_body<D>, 64>(KID_FORWARD, b.B, solver_args(b));
          launch_term_step(b);
          return;
        }
      }
      timed_launch<SolverArgs<D>, riccati_dense_body<D>, 64, (RiccatiDenseGeom<D>::NT2 > 6 ? 1 : 2)>(KID_RICCATI, b.B, solver_args(b));
      timed_launch<SolverArgs<D>, forward_full_body<D>, 64>(KID_FORWARD, b.B, solver_args(b));
      launch_term_step(b);
    }
    void launch_term_step(const Buffers<D> & b)
    {
      if (b.CN != nullptr)
        timed_launch<SolverArgs<D>, term_step_body<D>, 64>(KID_FORWARD, (b.B + 63) / 64, solver_args(b));
    }
    int launch_backtracking(const Buffers<D> & b)
    {
      const int slots = b.B < LS_SLOTS ? b.B : LS_SLOTS;
      timed_launch<SolverArgs<D>, compact_body<D>, 64>(KID_SELECT, 1, solver_args(b));
      return slots;
    }
    // The backtracking candidates alpha = 1/2, 1/4, .. of the instances in und_list, in two batches over the same list: the first LS_FIRST, then --
    // for the instances none of them decided (fdyn_trial_body skips the others) -- the rest.  One batch of all nine cost what a full trial of the
    // batch costs (the biped: 4.4 ms per launch, three launches per control step, with ~ 10 % of the instances backtracking); most of them
    // accept 1/2 or 1/4.  Same decisions: select_body takes the first candidate that passes, in order.
    static constexpr int LS_FIRST = 2;
    void launch_backtracking_trials(const Buffers<D> & b, StageKernelArgs<D> sk)
    {
      const int nb = (b.B + 63) / 64;
      for (int j0 = 1; j0 < D::LS_N; j0 += (j0 == 1 ? LS_FIRST : D::LS_N))
      {
        sk.j0 = j0;
        sk.nj = j0 == 1 ? (LS_FIRST < D::LS_N - 1 ? LS_FIRST : D::LS_N - 1) : D::LS_N - j0;
        timed_launch<StageKernelArgs<D>, fdyn_trial_body<D>, 64, TRIAL_MINW>(KID_SELECT, sk.slots * (H + 1), sk, true);
        timed_launch<SolverArgs<D>, select_body<D>, 64>(KID_SELECT, nb, solver_args(b, sk.j0, sk.nj));
      }
    }
    void launch_line_search(const Buffers<D> & b)
    {
      StageKernelArgs<D> sk = stage_args(b);
      sk.j0 = 0;
      sk.nj = 1;
      timed_launch<StageKernelArgs<D>, fdyn_trial_body<D>, 64, TRIAL_MINW>(KID_TRIAL, b.B * (H + 1), sk);
      timed_launch<SolverArgs<D>, select_body<D>, 64>(KID_SELECT, (b.B + 63) / 64, solver_args(b, 0, 1));
      const int slots = launch_backtracking(b);
      sk.slots = slots;
      launch_backtracking_trials(b, sk);
      timed_launch<SolverArgs<D>, apply_body<D>, 64>(KID_APPLY, b.B, solver_args(b));
    }
    void run_iteration(const Buffers<D> & b)
    {
      launch_deriv(b);
      launch_sweeps(b);
      launch_line_search(b);
    }
    // k ProxDDP iterations of one control step; tentative full steps as in KinoEngine::run_iterations
    void run_iterations(const Buffers<D> & b, int k)
    {
      if (!speculative_ls || k <= 1 || early_exit_on_tol) // (the convergence test belongs to the sequential scheme)
      {
        for (int it = 0; it < k; it++)
          run_iteration(b);
        return;
      }
      const int nb = (b.B + 63) / 64;
      launch_deriv(b);
      timed_launch<SolverArgs<D>, merit0_body<D>, 64>(KID_SELECT, nb, solver_args(b));
      for (int it = 0; it < k; it++)
      {
        launch_sweeps(b);
        if (it == k - 1)
        {
          launch_line_search(b);
          break;
        }
        SolverArgs<D> sa = solver_args(b);
        sa.mode = 1;
        timed_launch<SolverArgs<D>, apply_body<D>, 64>(KID_APPLY, b.B, sa);
        launch_deriv(b);
        timed_launch<SolverArgs<D>, spec_select_body<D>, 64>(KID_SELECT, nb, solver_args(b));
        const int slots = launch_backtracking(b);
        sa = solver_args(b);
        sa.slots = slots;
        sa.mode = 2;
        timed_launch<SolverArgs<D>, apply_body<D>, 64>(KID_SELECT, slots, sa, true);
        StageKernelArgs<D> sk = stage_args(b, slots);
        launch_backtracking_trials(b, sk);
        sa.mode = 0;
        timed_launch<SolverArgs<D>, apply_body<D>, 64>(KID_SELECT, slots, sa, true);
        launch_deriv(b, slots);
        timed_launch<SolverArgs<D>, merit0_body<D>, 64>(KID_SELECT, nb, sa);
      }
    }
    void copy_centres(const Buffers<D> & b)
    {
      d2d(b.vs_e, b.vs, (size_t)b.B * R * D::NC * sizeof(double), cur);
      d2d(b.lams_e, b.lams, (size_t)b.B * R * D::NDX * sizeof(double), cur);
      if (b.CN != nullptr)
        d2d(b.vN_e, b.vN, (size_t)b.B * 3 * sizeof(double), cur);
    }
    void upload_stages() { stage_ring.upload(buf.stages, horizon.data(), (size_t)H * sizeof(StageShared<D>), stream); }
    UploadRing stage_ring;

    // reference: src/mpc.cpp:72-91.  All instances share x0 = reference state: solve instance 0, broadcast.
    void cold_solve(const StageShared<D> & def, const DevModel<D> & m)
    {
      std::vector<double> xs0((size_t)R * D::NX), us0((size_t)R * D::NU, 0.0);
      for (int t = 0; t < R; t++)
        std::copy(x_model_ref.begin(), x_model_ref.end(), xs0.begin() + (size_t)t * D::NX);
      for (int t = 0; t < R; t++)
        std::copy(def.u_ref, def.u_ref + D::NU, us0.begin() + (size_t)t * D::NU); // getReferenceControl(0) (src/mpc.cpp:75)
      head = 0;
      h2d(buf.xs, xs0.data(), xs0.size() * sizeof(double), stream);
      h2d(buf.us, us0.data(), us0.size() * sizeof(double), stream);
      std::vector<double> sc0(SC_N, 0.0);
      sc0[SC_PREG] = REG_INIT;
      h2d(buf.scal, sc0.data(), SC_N * sizeof(double), stream);
      upload_stages();
      dev_zero(buf.foot_ref, (size_t)H * D::NF * 3 * sizeof(double), stream); // identity contact poses (src/ocp-handler.cpp:116)
      Buffers<D> b1 = buf;
      b1.B = 1;
      aux_launches = true;
      copy_centres(b1);
      std::vector<double> sc(SC_N);
      cold_trace.clear();
      const int cold_max = std::getenv("SMPC_COLD_MAX_ITERS") ? std::atoi(std::getenv("SMPC_COLD_MAX_ITERS")) : 100; // (diagnostics)
      for (int it = 0; it < cold_max; it++)
      {
        run_iteration(b1);
        d2h(sc.data(), buf.scal, SC_N * sizeof(double), stream);
        stream_sync(stream);
        cold_iters = it + 1;
        cold_trace.insert(cold_trace.end(), {sc[SC_PHI0], sc[SC_PRIM], sc[SC_DUAL], sc[SC_ALPHA]});
        if (std::fmax(sc[SC_PRIM], sc[SC_DUAL]) <= ms.TOL)
          break;
        if (std::fabs(sc[SC_DPHI0]) <= STALL_REL * std::fmax(1.0, std::fabs(sc[SC_PHI0])))
          break;
        if (sc[SC_DUAL] <= ms.TOL)
          copy_centres(b1);
      }
      aux_launches = false;
      auto bc = [&](double * p, size_t per) {
        for (size_t done = 1; done < (size_t)B;)
        {
          const size_t n = std::min(done, (size_t)B - done);
          d2d(p + done * per, p, n * per * sizeof(double), stream);
          done += n;
        }
      };
      bc(buf.xs, (size_t)R * D::NX);
      bc(buf.us, (size_t)R * D::NU);
      bc(buf.vs, (size_t)R * D::NC);
      bc(buf.lams, (size_t)R * D::NDX);
      bc(buf.scal, SC_N);
      bc(buf.forces, (size_t)H * D::NCM);
      if (buf.CN != nullptr)
      {
        bc(buf.vN, 3);
        bc(buf.dcm_ref, 3);
      }
      // swing start / end = foot positions at the reference state (FootTrajectory ctor, src/foot-trajectory.cpp:20-39)
      std::vector<double> ft((size_t)D::NF * 6);
      host_foot_positions(m, x_model_ref.data(), ft.data());
      h2d(buf.ftraj, ft.data(), ft.size() * sizeof(double), stream);
      stream_sync(stream);
      bc(buf.ftraj, (size_t)D::NF * 6);
      stream_sync(stream);
      for (int f = 0; f < D::NF; f++)
        for (int i = 0; i < 3; i++)
          ref_foot_pos[f][i] = ft[f * 6 + i];
    }
    static void host_foot_positions(const DevModel<D> & m, const double * x, double * out)
    {
      M3 Rj[D::NJ];
      V3 pj[D::NJ];
      for (int j = 0; j < D::NJ; j++)
      {
        if (j == 0)
        {
          Rj[0] = quat_to_R(Quat{x[3], x[4], x[5], x[6]});
          pj[0] = ld3(x);
        }
        else
        {
          const double ang = x[6 + j], s = std::sin(ang), c = std::cos(ang);
          const int jt = m.jtype[j];
          M3 Rq = jt == 1 ? M3{1, 0, 0, 0, c, -s, 0, s, c} : (jt == 2 ? M3{c, 0, s, 0, 1, 0, -s, 0, c} : M3{c, -s, 0, s, c, 0, 0, 0, 1});
          Rj[j] = Rj[m.parent[j]] * (ldm3(m.jpR[j]) * Rq);
          pj[j] = pj[m.parent[j]] + Rj[m.parent[j]] * ld3(m.jpp[j]);
        }
      }
      for (int f = 0; f < D::NF; f++)
      {
        const V3 p = Rj[m.foot_joint[f]] * ld3(m.foot_p[f]) + pj[m.foot_joint[f]];
        st3(out + f * 6, p);
        st3(out + f * 6 + 3, p);
      }
    }

    void generate_cycle_horizon(const unsigned char * cs, int n) override
    {
      if (n <= 0)
        throw std::runtime_error("contact sequence must not be empty");
      timer.generate(cs, n, D::NF, H);
      cycle.clear();
      unsigned previous = (1u << D::NF) - 1u; // land flags: in contact here, not in the stage before (src/mpc.cpp:133-137,167-185)
      for (auto & st : timer.states)
      {
        int active = 0;
        for (int f = 0; f < D::NF; f++)
          active += st[f] ? 1 : 0;
        StageShared<D> s;
        std::memset(&s, 0, sizeof(s));
        for (int f = 0; f < D::NF; f++)
          if (st[f])
          {
            s.mask |= 1u << f;
            s.f_ref[D::FS * f + 2] = ms.support_force / (double)active; // src/mpc.cpp:149-167
            if constexpr (D::KINO)
              s.u_ref[D::FS * f + 2] = s.f_ref[D::FS * f + 2];
          }
        s.land = s.mask & ~previous;
        previous = s.mask;
        for (int i = 0; i < D::NX; i++)
          s.x_tgt[i] = x_model_ref[i];
        cycle.push_back(s);
      }
    }
    void upload_velocity(const double * V, bool broadcast)
    {
      set_device(device_id);
      std::vector<double> hh((size_t)B * 6);
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 6; i++)
          hh[(size_t)b * 6 + i] = broadcast ? V[i] : V[(size_t)b * 6 + i];
      h2d(buf.vbase, hh.data(), hh.size() * sizeof(double), stream);
      stream_sync(stream);
    }
    void switch_to_walk(const double * v6) override
    {
      walking = true;
      for (int i = 0; i < 6; i++)
        velocity_base[i] = v6[i];
      upload_velocity(v6, true);
    }
    void switch_to_stand() override
    {
      walking = false;
      for (int i = 0; i < 6; i++)
        velocity_base[i] = 0.0;
      upload_velocity(velocity_base, true);
    }
    void set_velocity_base_batched(const double * V) override
    {
      for (int i = 0; i < 6; i++)
        velocity_base[i] = V[i];
      upload_velocity(V, false);
    }

    void iterate_device(const double * Xd) override
    {
      ref_rot.reset(); // (every control step rewrites every stage's reference pose with the identity rotation: src/mpc.cpp:303-309)
      if (cycle.empty())
        throw std::runtime_error("generateCycleHorizon must be called before iterate");
      int last_support = 0;
      for (int f = 0; f < D::NF; f++)
        last_support += (horizon[H - 1].mask >> f) & 1u;
      StageShared<D> incoming;
      if (walking || last_support < D::NF)
      {
        incoming = cycle[0];
        std::rotate(cycle.begin(), cycle.begin() + 1, cycle.end());
        timer.recede_cycle();
      }
      else
      {
        incoming = standing;
        timer.update_timing(true);
      }
      horizon.erase(horizon.begin());
      horizon.push_back(incoming);
      for (int i = 0; i < D::NX; i++)
        horizon[H - 1].x_tgt[i] = x_reference[i];
      for (int i = 0; i < 6; i++)
        horizon[H - 1].x_tgt[D::NQ + i] = velocity_base[i];
      upload_stages();
      head = head + 1 == R ? 0 : head + 1;
      RecedeArgs<D> ra;
      ra.b = buf;
      ra.head = head;
      ra.X = Xd;
      for (int f = 0; f < D::NF; f++)
        ra.land[f] = timer.land[f].empty() ? -1 : timer.land[f][0];
      ra.T_fly = ms.T_fly;
      ra.T_contact = ms.T_contact;
      ra.swing_apex = ms.swing_apex;
      ra.timestep = ms.timestep;
      ra.shift = 1;
      ra.reg_init = REG_INIT;
      timed_launch<RecedeArgs<D>, recede_body<D>, 64>(KID_RECEDE, B, ra);
      if (n_parts > 1 && buf.CN == nullptr && !profiling) // (per-launch event timings mean nothing once launches overlap: one part while profiling)
      {
        event_record(ev_fork, stream);
        for (int i = 0; i < n_parts; i++)
        {
          const int i0 = (int)((long long)B * i / n_parts), i1 = (int)((long long)B * (i + 1) / n_parts);
          const Buffers<D> part = slice(buf, i0, i1 - i0, und_part[i]);
          cur = part_stream[i];
          if (i > 0)
            stream_wait_event(cur, ev_fork);
          copy_centres(part);
          run_iterations(part, ms.max_iters);
        }
        for (int i = 1; i < n_parts; i++)
        {
          event_record(ev_join[i], part_stream[i]);
          stream_wait_event(stream, ev_join[i]);
        }
        cur = stream;
        return;
      }
      copy_centres(buf);
      run_iterations(buf, ms.max_iters);
    }
    // instances i0 .. i0 + n of every per-instance array (problems without a terminal constraint)
    Buffers<D> slice(const Buffers<D> & b, int i0, int n, int * und) const
    {
      Buffers<D> s = b;
      s.B = n;
      const size_t o = (size_t)i0, Rs = (size_t)R, Hs = (size_t)H;
      auto adv = [&](double *& p, size_t per) {
        if (p)
          p += o * per;
      };
      adv(s.xs, Rs * D::NX); adv(s.us, Rs * D::NU); adv(s.vs, Rs * D::NC); adv(s.lams, Rs * D::NDX);
      adv(s.vs_e, Rs * D::NC); adv(s.lams_e, Rs * D::NDX);
      adv(s.xs_b, Rs * D::NX); adv(s.us_b, Rs * D::NU); adv(s.vs_b, Rs * D::NC); adv(s.lams_b, Rs * D::NDX);
      adv(s.dxs, (Hs + 1) * D::NDX); adv(s.dus, Hs * D::NU); adv(s.dvs, Hs * D::NC); adv(s.dlams, Hs * D::NDX);
      adv(s.foot_ref, Hs * D::NF * 3); adv(s.ftraj, (size_t)D::NF * 6); adv(s.vbase, 6); adv(s.vref, Rs * 6);
      adv(s.lq, Hs * D::LQ_STRIDE); adv(s.gains, Hs * (size_t)D::G_STRIDE);
      adv(s.QN, (size_t)D::NDX * D::NDX); adv(s.qN, D::NDX);
      adv(s.parts0, (Hs + 1) * 4); adv(s.partsT, (size_t)D::LS_N * (Hs + 1) * 2); adv(s.scal, SC_N);
      adv(s.xdotT, (size_t)D::LS_N * 4 * D::NV); adv(s.xdot01, (size_t)4 * D::NV);
      adv(s.forcesT, Hs * D::LS_N * D::NCM); adv(s.forces, Hs * D::NCM);
      s.ls_sel = b.ls_sel + i0;
      s.und_list = und;
      return s;
    }
    void iterate_host(const double * X) override
    {
      set_device(device_id);
      h2d(X_dev, X, (size_t)B * D::NX * sizeof(double), stream);
      iterate_device(X_dev);
      stream_sync(stream);
    }
    void sync() override
    {
      set_device(device_id);
      stream_sync(stream);
    }
    void gather_x_device(int t, double * out_dev) override
    {
      if (t < 0 || t > H)
        throw std::runtime_error("Stage index exceeds stage vector size");
      GatherArgs<D> ga;
      ga.b = buf;
      ga.head = head;
      ga.t = t;
      ga.out = out_dev;
      launch<GatherArgs<D>, gather_x_body<D>, 256>((int)(((size_t)B * D::NX + 255) / 256), stream, ga);
    }
    void check_stage(int t) const
    {
      if (t < 0 || t >= H)
        throw std::runtime_error("Stage index exceeds stage vector size");
    }
    void fill_strided(double * base, size_t stride, int count, const double * v, int n)
    {
      set_device(device_id);
      FillStridedArgs fa;
      fa.base = base;
      fa.stride = stride;
      fa.count = count;
      fa.n = n;
      for (int i = 0; i < n; i++)
        fa.v[i] = v[i];
      launch<FillStridedArgs, fill_strided_body, 64>((count + 63) / 64, stream, fa);
      stream_sync(stream);
    }
    // what: 0 = control target (nu), 1 = state target (nx), 2 = contact-force references (force_size * nfeet)
    void set_stage_reference(int t, int what, const double * v, int n) override
    {
      check_stage(t);
      if (what == 0)
      {
        if (n != D::NU)
          throw std::runtime_error("u_ref not of the right size");
        std::copy(v, v + n, horizon[t].u_ref);
        if constexpr (D::KINO) // getReferenceForce is a segment of the reference control (src/kinodynamics.cpp:258-265): keep the mirror in step
          std::copy(v, v + D::NCM, horizon[t].f_ref);
      }
      else if (what == 1)
      {
        if (n != D::NX)
          throw std::runtime_error("x_ref not of the right size");
        std::copy(v, v + n, horizon[t].x_tgt);
        fill_strided(buf.vref + (size_t)ring_slot(head, t, R) * 6, (size_t)R * 6, B, v + D::NQ, 6);
      }
      else if (what == 2)
      {
        if (n != D::NCM)
          throw std::runtime_error("Reference forces do not have the right dimension");
        std::copy(v, v + n, horizon[t].f_ref);
        if constexpr (D::KINO)
          std::copy(v, v + n, horizon[t].u_ref);
      }
      else
        throw std::runtime_error("unknown stage reference");
    }
    void get_stage_reference(int t, int what, double * v, int n) override
    {
      check_stage(t);
      if (what == 0 && n == D::NU)
        std::copy(horizon[t].u_ref, horizon[t].u_ref + n, v);
      else if (what == 1 && n == D::NX)
      {
        std::copy(horizon[t].x_tgt, horizon[t].x_tgt + n, v);
        get_linear(buf.vref + (size_t)ring_slot(head, t, R) * 6, 6, v + D::NQ);
      }
      else if (what == 2 && n == D::NCM)
        std::copy(horizon[t].f_ref, horizon[t].f_ref + n, v);
      else
        throw std::runtime_error("unknown stage reference or wrong size");
    }
    void set_reference_pose(int t, int foot, const double * p3) override
    {
      check_stage(t);
      if (foot < 0 || foot >= D::NF)
        throw std::runtime_error("unknown end effector");
      ref_rot.set(t, foot, nullptr); // (a translation: identity rotation)
      fill_strided(buf.foot_ref + ((size_t)t * D::NF + foot) * 3, (size_t)H * D::NF * 3, B, p3, 3);
    }
    void get_reference_pose(int t, int foot, int inst, double * p3) override
    {
      check_stage(t);
      if (foot < 0 || foot >= D::NF || inst < 0 || inst >= B)
        throw std::runtime_error("unknown end effector or instance");
      get_linear(buf.foot_ref + (((size_t)inst * H + t) * D::NF + foot) * 3, 3, p3);
    }
    RefRotations ref_rot; // rotations of the foot reference placements: API state (smpc_model.h)
    void set_reference_rotation(int t, int foot, const double * R9) override
    {
      check_stage(t);
      if (foot < 0 || foot >= D::NF)
        throw std::runtime_error("unknown end effector");
      ref_rot.set(t, foot, R9);
    }
    void get_reference_rotation(int t, int foot, double * R9) override
    {
      check_stage(t);
      if (foot < 0 || foot >= D::NF)
        throw std::runtime_error("unknown end effector");
      ref_rot.get(t, foot, R9);
    }
    unsigned contact_mask(int t) const override
    {
      check_stage(t);
      return horizon[t].mask;
    }
    size_t state_io(StateIO & io) override
    {
      set_device(device_id);
      stream_sync(stream);
      io.tag(0x534d504346554c4cLL, "kind (full dynamics)");
      io.tag(B, "batch");
      io.tag(H, "horizon");
      io.tag(D::NX, "nx");
      io.tag(D::NU, "nu");
      io.tag(buf.CN != nullptr ? 1 : 0, "terminal constraint");
      io.pod(head);
      io.pod(walking);
      io.host(velocity_base, sizeof(velocity_base));
      io.vec(x_reference);
      io.vec(horizon);
      io.vec(cycle);
      io.timer(timer);
      const size_t BR = (size_t)B * R;
      io.dev(buf.xs, BR * D::NX * sizeof(double));
      io.dev(buf.us, BR * D::NU * sizeof(double));
      io.dev(buf.vs, BR * D::NC * sizeof(double));
      io.dev(buf.lams, BR * D::NDX * sizeof(double));
      io.dev(buf.ftraj, (size_t)B * D::NF * 6 * sizeof(double));
      io.dev(buf.foot_ref, (size_t)B * H * D::NF * 3 * sizeof(double));
      io.dev(buf.vbase, (size_t)B * 6 * sizeof(double));
      io.dev(buf.vref, BR * 6 * sizeof(double));
      io.dev(buf.scal, (size_t)B * SC_N * sizeof(double));
      io.dev(buf.xdot01, (size_t)B * 4 * D::NV * sizeof(double));
      io.dev(buf.forces, (size_t)B * H * D::NCM * sizeof(double));
      if (buf.CN != nullptr)
        io.dev(buf.vN, (size_t)B * 3 * sizeof(double));
      if (io.mode == StateIO::LOAD)
        upload_stages();
      stream_sync(stream);
      return io.pos;
    }
    // state feedback front-end on measured states X [B][nq + nv] (host): feet [B][NF][3], com [B][3], hg [B][6], centroidal state [B][9]
    // (host outputs, any may be null) -- RobotDataHandler::updateInternalData + getCentroidalState on the stage kernel's kinematics
    void update_internal_data(const double * X, double * feet, double * com, double * hg, double * cstate) override
    {
      set_device(device_id);
      const size_t nf = (size_t)B * D::NF * 3, nc = (size_t)B * 3, nh = (size_t)B * 6, ns = (size_t)B * 9;
      double * st = staging((nf + nc + nh + ns) * sizeof(double));
      h2d(X_dev, X, (size_t)B * D::NX * sizeof(double), stream);
      FrontendArgs<D> fa;
      fa.b = buf;
      fa.X = X_dev;
      fa.feet = st;
      fa.com = st + nf;
      fa.hg = st + nf + nc;
      fa.cstate = st + nf + nc + nh;
      launch<FrontendArgs<D>, frontend_full_body<D>, 64, 1, 1>(B, stream, fa);
      if (feet)
        d2h(feet, st, nf * sizeof(double), stream);
      if (com)
        d2h(com, st + nf, nc * sizeof(double), stream);
      if (hg)
        d2h(hg, st + nf + nc, nh * sizeof(double), stream);
      if (cstate)
        d2h(cstate, st + nf + nc + nh, ns * sizeof(double), stream);
      stream_sync(stream);
    }
    // constrained forward dynamics of n states (host buffers): a [n][NV], lam [n][FS NF] (feet in contact first), iters [n] (may be null)
    void full_forward_dynamics(int n, const double * X, const double * tau, const unsigned * mask, const double * Kp, const double * Kd,
                               double prox_accuracy, double prox_mu, int prox_max_iter, double * a, double * lam, int * iters, double * kernel_ms) override
    {
      if constexpr (D::KINO)
        throw std::runtime_error("full_forward_dynamics: the kinodynamics variant has no constrained forward dynamics (use a full-dynamics handle)");
      else
      {
        if (n < 1)
          throw std::runtime_error("full_forward_dynamics: n must be positive");
        set_device(device_id);
        constexpr int NV = D::NV, NX = D::NX, NCM = D::NCM, NU = D::NU;
        // staging layout (doubles): X | tau | a | lam | mask (unsigned) | iters (int)
        const size_t oX = 0, oT = oX + (size_t)n * NX, oA = oT + (size_t)n * NU, oL = oA + (size_t)n * NV, oM = oL + (size_t)n * NCM,
                     oI = oM + ((size_t)n + 1) / 2, total = oI + ((size_t)n + 1) / 2;
        double * st = staging(total * sizeof(double));
        h2d(st + oX, X, (size_t)n * NX * sizeof(double), stream);
        h2d(st + oT, tau, (size_t)n * NU * sizeof(double), stream);
        h2d(st + oM, mask, (size_t)n * sizeof(unsigned), stream);
        FdynFdArgs<D> fa;
        fa.b = buf;
        fa.X = st + oX;
        fa.tau = st + oT;
        fa.mask = reinterpret_cast<const unsigned *>(st + oM);
        for (int i = 0; i < 6; i++)
        {
          fa.Kp[i] = (Kp && i < D::FS) ? Kp[i] : 0.0;
          fa.Kd[i] = (Kd && i < D::FS) ? Kd[i] : 0.0;
        }
        fa.prox_accuracy = prox_accuracy > 0 ? prox_accuracy : 1e-9; // ProximalSettings(1e-9, 1e-10, 10), src/fulldynamics.cpp:39
        fa.prox_mu = prox_mu > 0 ? prox_mu : 1e-10;
        fa.prox_max_iter = prox_max_iter > 0 ? prox_max_iter : 10;
        fa.a_out = st + oA;
        fa.lam_out = st + oL;
        fa.iters_out = reinterpret_cast<int *>(st + oI);
        stream_sync(stream);
        const auto t0 = std::chrono::steady_clock::now();
        launch<FdynFdArgs<D>, fdyn_fd_body<D>, 64, 1, 1>(n, stream, fa);
        stream_sync(stream);
        if (kernel_ms)
          *kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        d2h(a, st + oA, (size_t)n * NV * sizeof(double), stream);
        d2h(lam, st + oL, (size_t)n * NCM * sizeof(double), stream);
        if (iters)
          d2h(iters, st + oI, (size_t)n * sizeof(int), stream);
        stream_sync(stream);
      }
    }
    double * staging(size_t bytes)
    {
      set_device(device_id);
      if (bytes > stage_out_bytes)
      {
        dev_free(stage_out);
        stage_out = (double *)dev_alloc(bytes);
        stage_out_bytes = bytes;
      }
      return stage_out;
    }
    void get_ring(const double * src, int n, int count, double * out)
    {
      set_device(device_id);
      stream_sync(stream);
      std::vector<double> tmp((size_t)B * R * n);
      d2h(tmp.data(), src, tmp.size() * sizeof(double), stream);
      stream_sync(stream);
      for (int b = 0; b < B; b++)
        for (int t = 0; t < count; t++)
          std::memcpy(out + ((size_t)b * count + t) * n, tmp.data() + ((size_t)b * R + ring_slot(head, t, R)) * n, n * sizeof(double));
    }
    void get_linear(const double * src, size_t n, double * out)
    {
      set_device(device_id);
      stream_sync(stream);
      d2h(out, src, n * sizeof(double), stream);
      stream_sync(stream);
    }
    void get_K(double * out, bool all)
    {
      stream_sync(stream);
      const int nt = all ? H : 1;
      const size_t n = (size_t)B * nt * D::NU * D::NDX;
      double * dev = staging(n * sizeof(double));
      FullGainOutArgs<D> ga;
      ga.b = buf;
      ga.nt = nt;
      ga.out = dev;
      launch<FullGainOutArgs<D>, full_gains_out_body<D>, 64>(B * nt, stream, ga);
      d2h(out, dev, n * sizeof(double), stream);
      stream_sync(stream);
    }
    void get(int what, double * out) override
    {
      switch (what)
      {
      case 0:
        return get_ring(buf.xs, D::NX, H + 1, out);
      case 1:
        return get_ring(buf.us, D::NU, H, out);
      case 2:
        return get_K(out, false);
      case 3:
        return get_K(out, true);
      case 4:
        return get_ring(buf.vs, D::NC, H, out);
      case 5:
      {
        // device arrays hold lambda_{t+1} at stage t; the API returns lams[0..H] with lams[0] = 0
        std::vector<double> tmp((size_t)B * H * D::NDX);
        get_ring(buf.lams, D::NDX, H, tmp.data());
        for (int b = 0; b < B; b++)
        {
          double * o = out + (size_t)b * (H + 1) * D::NDX;
          std::memset(o, 0, D::NDX * sizeof(double));
          std::memcpy(o + D::NDX, tmp.data() + (size_t)b * H * D::NDX, (size_t)H * D::NDX * sizeof(double));
        }
        return;
      }
      case 6:
        return get_linear(buf.xdot01, (size_t)B * 4 * D::NV, out);
      case 7:
        return get_linear(buf.foot_ref, (size_t)B * H * D::NF * 3, out);
      case 8:
        return get_linear(buf.scal, (size_t)B * SC_N, out);
      case 9:
        return get_linear(buf.forces, (size_t)B * H * D::NCM, out);
      default:
        throw std::runtime_error("unknown output");
      }
    }
    // interpolated whole-body targets at `delay` after the last solve; host outputs, any may be null
    void interpolate(double delay, int knots, double * x_out, double * acc_out, double * f_out) override
    {
      if (knots < 2 || knots > H + 1)
        throw std::runtime_error("interpolate: knots must be in [2, horizon + 1]");
      if (!(delay >= 0.0))
        throw std::runtime_error("interpolate: delay must be non-negative");
      const size_t nx = (size_t)B * D::NX, na = (size_t)B * D::NV, nf = (size_t)B * D::NCM;
      double * st = staging((nx + na + nf) * sizeof(double));
      FullInterpArgs<D> ia;
      ia.b = buf;
      ia.head = head;
      ia.knots = knots;
      ia.delay = delay;
      ia.timestep = ms.timestep;
      ia.x_out = x_out ? st : nullptr;
      ia.acc_out = acc_out ? st + nx : nullptr;
      ia.f_out = f_out ? st + nx + na : nullptr;
      ia.u_out = nullptr;
      launch<FullInterpArgs<D>, full_interp_body<D>, 64>(B, stream, ia);
      if (x_out)
        d2h(x_out, st, nx * sizeof(double), stream);
      if (acc_out)
        d2h(acc_out, st + nx, na * sizeof(double), stream);
      if (f_out)
        d2h(f_out, st + nx + na, nf * sizeof(double), stream);
      stream_sync(stream);
    }
    void interpolate_device(double delay, int knots, double * x_dev, double * acc_dev, double * f_dev) override
    {
      if (knots < 2 || knots > H + 1)
        throw std::runtime_error("interpolate: knots must be in [2, horizon + 1]");
      if (!(delay >= 0.0))
        throw std::runtime_error("interpolate: delay must be non-negative");
      set_device(device_id);
      FullInterpArgs<D> ia;
      ia.b = buf;
      ia.head = head;
      ia.knots = knots;
      ia.delay = delay;
      ia.timestep = ms.timestep;
      ia.x_out = x_dev;
      ia.acc_out = acc_dev;
      ia.f_out = f_dev;
      ia.u_out = nullptr;
      launch<FullInterpArgs<D>, full_interp_body<D>, 64>(B, stream, ia);
    }
    // One step of a simulated batch resident in HBM: constrained forward dynamics of the feet in `mask` (Baumgarte gains Kp, Kd [FS];
    // proximal settings of record), then semi-implicit Euler over dt, X updated in place.  Asynchronous on this engine's stream.
    double *sim_a = nullptr, *sim_lam = nullptr;
    unsigned * sim_mask = nullptr;
    unsigned sim_mask_value = ~0u;
    void sim_step_device(double * X_dev_, const double * tau_dev, unsigned mask, const double * Kp, const double * Kd, double dt) override
    {
      if constexpr (D::KINO)
        throw std::runtime_error("sim_step_device: the kinodynamics variant has no constrained forward dynamics (use a full-dynamics handle)");
      else
      {
        set_device(device_id);
        if (!sim_a)
        {
          sim_a = (double *)dev_alloc((size_t)B * D::NV * sizeof(double));
          sim_lam = (double *)dev_alloc((size_t)B * D::NCM * sizeof(double));
          sim_mask = (unsigned *)dev_alloc((size_t)B * sizeof(unsigned));
        }
        if (mask != sim_mask_value)
        {
          std::vector<unsigned> m(B, mask);
          h2d(sim_mask, m.data(), m.size() * sizeof(unsigned), stream);
          stream_sync(stream); // (m goes out of scope)
          sim_mask_value = mask;
        }
        FdynFdArgs<D> fa;
        fa.b = buf;
        fa.X = X_dev_;
        fa.tau = tau_dev;
        fa.mask = sim_mask;
        for (int i = 0; i < 6; i++)
        {
          fa.Kp[i] = (Kp && i < D::FS) ? Kp[i] : 0.0;
          fa.Kd[i] = (Kd && i < D::FS) ? Kd[i] : 0.0;
        }
        fa.prox_accuracy = 1e-9; // ProximalSettings(1e-9, 1e-10, 10), src/fulldynamics.cpp:39
        fa.prox_mu = 1e-10;
        fa.prox_max_iter = 10;
        fa.a_out = sim_a;
        fa.lam_out = sim_lam;
        fa.iters_out = nullptr;
        launch<FdynFdArgs<D>, fdyn_fd_body<D>, 64, 1, 1>(B, stream, fa);
        SimStepArgs<D> sa;
        sa.X = X_dev_;
        sa.a = sim_a;
        sa.dt = dt;
        launch<SimStepArgs<D>, sim_integrate_body<D>, 64, 1, 1>(B, stream, sa);
      }
    }
    event_t ev_handoff{};
    bool ev_handoff_valid = false;
    void wait_stream(stream_t other) override
    {
      set_device(device_id);
      if (!ev_handoff_valid)
      {
        ev_handoff = event_create();
        ev_handoff_valid = true;
      }
      event_record(ev_handoff, stream);
      stream_wait_event(other, ev_handoff);
    }
    // u = u_interp - K_0 (x_interp (-) x_meas) at `delay` after the last solve (reference examples/go2_fulldynamics.py:271-285)
    void riccati_feedback(double delay, const double * X, double * u_out) override
    {
      if (!(delay >= 0.0))
        throw std::runtime_error("riccati_feedback: delay must be non-negative");
      const size_t nx = (size_t)B * D::NX, nu = (size_t)B * D::NU, nk = (size_t)B * D::NU * D::NDX;
      double * st = staging((nx + 2 * nu + nk) * sizeof(double));
      double *xi = st, *ui = st + nx, *uo = ui + nu, *k0 = uo + nu;
      h2d(X_dev, X, nx * sizeof(double), stream);
      FullInterpArgs<D> ia;
      ia.b = buf;
      ia.head = head;
      ia.knots = 2;
      ia.delay = delay;
      ia.timestep = ms.timestep;
      ia.x_out = xi;
      ia.acc_out = nullptr;
      ia.f_out = nullptr;
      ia.u_out = ui;
      launch<FullInterpArgs<D>, full_interp_body<D>, 64>(B, stream, ia);
      FullGainOutArgs<D> ga;
      ga.b = buf;
      ga.nt = 1;
      ga.out = k0;
      launch<FullGainOutArgs<D>, full_gains_out_body<D>, 64>(B, stream, ga);
      FeedbackArgs<D> fa;
      fa.b = buf;
      fa.X_meas = X_dev;
      fa.x_interp = xi;
      fa.u_interp = ui;
      fa.K0 = k0;
      fa.u_out = uo;
      launch<FeedbackArgs<D>, feedback_body<D>, 64>(B, stream, fa);
      d2h(u_out, uo, nu * sizeof(double), stream);
      stream_sync(stream);
    }
    bool phase_cycles(double * out64) override
    {
      if (!buf.dbg)
        return false;
      get_linear(buf.dbg, 64, out64);
      return true;
    }
    int lq_size() const override { return D::LQ_STRIDE; }
    void debug_lq(int inst, int t, double * out) override
    {
      if (inst < 0 || inst >= B || t < 0 || t >= H)
        throw std::runtime_error("Stage index exceeds stage vector size");
      get_linear(buf.lq + ((size_t)inst * H + t) * D::LQ_STRIDE, D::LQ_STRIDE, out);
      // the derivative pass writes the upper 16 x 16 tiles of Q and R only (readers take the upper triangle): mirror here
      for (int i = 0; i < D::NDX; i++)
        for (int j = 0; j < i; j++)
          out[D::O_Q + i * D::NDX + j] = out[D::O_Q + j * D::NDX + i];
      for (int i = 0; i < D::NU; i++)
        for (int j = 0; j < i; j++)
          out[D::O_R + i * D::NU + j] = out[D::O_R + j * D::NU + i];
    }
    void debug_steps(double * dxs, double * dus) override
    {
      get_linear(buf.dxs, (size_t)B * (H + 1) * D::NDX, dxs);
      get_linear(buf.dus, (size_t)B * H * D::NU, dus);
    }
    void debug_terminal(int inst, double * QN, double * qN) override
    {
      if (inst < 0 || inst >= B)
        throw std::runtime_error("instance index out of range");
      get_linear(buf.QN + (size_t)inst * D::NDX * D::NDX, D::NDX * D::NDX, QN);
      get_linear(buf.qN + (size_t)inst * D::NDX, D::NDX, qN);
    }
  };
} // namespace smpc
