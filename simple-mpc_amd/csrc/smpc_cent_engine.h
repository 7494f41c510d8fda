// smpc_cent_engine.h -- host side of the batched centroidal MPC: the reference's MPC class (smpc_engine.h cites it line
// by line) over a CentroidalOCP (reference src/centroidal-dynamics.cpp).  One control step = two launches:
//   frontend_body   measured multibody states -> getCentroidalState + foot positions (src/mpc.cpp:192,200)
//   cent_step_body  recede + references + k ProxDDP iterations, one wavefront per instance (smpc_cent_kernels.h)
// What differs from the kinodynamics host logic:
//   - the problem state is RobotDataHandler::getCentroidalState() of the measured multibody state
//     (src/mpc.cpp:200, src/centroidal-dynamics.cpp:261-264, src/robot-handler.cpp:142-149)
//   - setReferencePose writes the contact POSITION of the stage's contact map (src/centroidal-dynamics.cpp:151-169)
//   - setReferenceState = setPoseBase + setVelocityBase, and setVelocityBase stores momentum references m v
//     (src/centroidal-dynamics.cpp:227-239, 286-291)
//   - x_reference_ starts as getReferenceState(0) of the default problem = 0 (src/centroidal-dynamics.cpp:36)
//   - the default problem has identity contact poses, i.e. contact positions at the origin (src/ocp-handler.cpp:117)
//   - no terminal constraint (src/centroidal-dynamics.cpp:318-337)
#pragma once
#include "smpc_cent6_kernels.h"
#include "smpc_cent_split.h"
#include "smpc_engine.h"

namespace smpc
{
  struct HostCentSettings // include/simple-mpc/centroidal-dynamics.hpp:27-43
  {
    double timestep;
    std::vector<double> w_u, w_com, w_linear_mom, w_angular_mom, w_linear_acc, w_angular_acc;
    double gravity[3];
    double mu;
    double Lfoot = 0.1, Wfoot = 0.075; // sole half-length / half-width (wrench cones of 6-D feet)
    int force_size = 3;
  };

  enum CentKernelId
  {
    CKID_FRONTEND = 0,
    CKID_STEP,    // recede (point feet with SMPC_CENT_FUSED=1: the one-kernel control step)
    CKID_DERIV,   // point feet (smpc_cent_split.h): lane-per-stage pre-pass ; 6-D feet (smpc_cent6_kernels.h): stage evaluation + derivatives + knot
    CKID_RICCATI, //   backward sweep
    CKID_FORWARD, //   forward sweep
    CKID_LS,      //   line search + step
    CKID_TRIAL,   //   6-D feet: stage merits of the line-search candidates (cent6_trial_body)
    CKID_N
  };

  template <class DK>
  struct cent_is_full_dims
  {
    static constexpr bool value = false;
  };
  template <int NJ_, int NF_, int FS_, int CN_, int LN_, int KIN_>
  struct cent_is_full_dims<FullDims<NJ_, NF_, FS_, CN_, LN_, KIN_>>
  {
    static constexpr bool value = true;
  };
  // buffers of the dense path (6-D feet); empty for point feet
  template <class DC, int FS>
  struct Cent6Extra
  {
  };
  template <class DC>
  struct Cent6Extra<DC, 6>
  {
    Buffers<typename DC::DD> sb;
    double *parts0 = nullptr, *partsT = nullptr, *xdotT = nullptr; // (Cent6Args, smpc_cent6_kernels.h)
  };

  // what the C ABI needs from a centroidal engine of any robot shape / foot type
  struct CentEngineBase
  {
    int B = 0, H = 0, R = 0, head = 0;
    int nu = 0, nc = 0, nf = 0, nq_mb = 0, nv_mb = 0;
    int device_id = 0; // every entry point makes this the current device first
    GaitTimer timer;
    stream_t stream;
    double x_reference[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int cold_iters = 0;
    std::vector<double> cold_trace;
    bool profiling = false;
    double kernel_ms[CKID_N] = {0};
    long kernel_calls[CKID_N] = {0};
    virtual ~CentEngineBase() {}
    virtual const CentBuffersBase & bufs() const = 0;
    virtual void collect_profile() = 0;
    virtual void generate_cycle_horizon(const unsigned char * cs, int n) = 0;
    virtual void switch_to_walk(const double * v6) = 0;
    virtual void switch_to_stand() = 0;
    virtual void set_velocity_base_batched(const double * V) = 0;
    virtual void iterate_device(const double * Xd) = 0;
    virtual void iterate_host(const double * X) = 0;
    virtual void sync() = 0;
    virtual void set_stage_reference(int t, int what, const double * v, int n) = 0;
    virtual void get_stage_reference(int t, int what, double * v, int n) = 0;
    virtual void set_reference_pose(int t, int foot, const double * p3) = 0;
    virtual void get_reference_pose(int t, int foot, int inst, double * p3) = 0;
    virtual void set_reference_rotation(int t, int foot, const double * R9) = 0;
    virtual void get_reference_rotation(int t, int foot, double * R9) = 0;
    virtual unsigned contact_mask(int t) const = 0;
    virtual void update_internal_data(const double * X, double * feet, double * com, double * hg, double * cstate) = 0;
    virtual void interpolate_device_id(double delay, int knots, double * com, double * vcom, double * fp, double * fv, double * f) = 0;
    virtual void wait_stream(stream_t other) = 0;
    virtual void interpolate(double delay, int knots, const double * X_meas, double * x_out, double * xdot_out, double * f_out, double * u_out) = 0;
    virtual size_t state_io(StateIO & io) = 0;
    virtual void get_K(double * out, bool all) = 0;
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
  };

  // DK: Dims of the multibody robot (front-end FK; a FullDims selects the front end of the dense stage kernels), DC: CentDims
  template <class DK, class DC>
  class CentEngine : public CentEngineBase
  {
  public:
    CentBuffers<DC> buf;
    const CentBuffersBase & bufs() const override { return buf; }
    Cent6Extra<DC, DC::FS> x6; // 6-D feet: knots, dense gains, terminal node, merit partials
    CentSplitBuffers sbuf;     // point feet: hand-over records of the kernel pipeline (smpc_cent_split.h)
    bool fused = false;        // point feet: SMPC_CENT_FUSED=1 runs the one-kernel control step (cross-check)
    // point feet: the batch runs as `nparts` contiguous parts, each on its own stream (SMPC_CENT_PARTS; instances are independent): the
    // backward sweep is bound by VALU / matrix-core issue, the pre-pass, forward sweep and line search by memory -- launches of different
    // parts fill each other's idle units.  Part p > 0 starts one kernel late (it waits for part p - 1's first pre-pass), so that the parts do
    // not march in step.
    int nparts = 1;
    int ls_passes = 1;      // 6-D feet: SMPC_CENT6_LS_PASSES=2 evaluates the full step first, the backtracking candidates where it failed
    bool ls_direct = false; // SMPC_CENT_LS=direct: the per-candidate re-evaluation (cent_ls_body) also for horizons the polynomial form covers
    std::vector<stream_t> part_stream; // [nparts - 1] (part 0 runs on `stream`)
    std::vector<event_t> part_event;   // [nparts - 1] completion of a part, [nparts] stagger events
    event_t ev_fork{};
    Buffers<DK> fk; // only .model is used (front-end kernel)
    HostMpcSettings ms;
    std::vector<CentStage<DC>> horizon, cycle;
    CentStage<DC> standing;
    bool walking = true;
    double velocity_base[6] = {0, 0, 0, 0, 0, 0};
    double com_ref_member[3] = {0, 0, 0}; // CentroidalOCP::com_ref_ (last setPoseBase)
    double mass;
    std::vector<double> x_model_ref;
    UploadRing stage_ring; // pinned staging of the per-step stage table
    double *X_dev = nullptr, *cstate_dev = nullptr, *feet_dev = nullptr;
    std::vector<std::pair<int, std::pair<event_t, event_t>>> pending_events;
    static constexpr double ARMIJO_C1 = 1e-4, REG_INIT = 1e-9, REG_MIN = 1e-10, REG_MAX = 1e9, REG_INC = 10.0, REG_DEC = 1.0 / 3.0, STALL_REL = 1e-9;

    CentEngine(const smpc_robot_model * rm, const HostCentSettings & cs, const HostMpcSettings & ms_, int batch, double gravity_arg, int device)
    : ms(ms_)
    {
      AllocScope ctor_scope; // (a throw below releases what was allocated so far: smpc_alloc_scope.h)
      if (batch <= 0)
        throw std::runtime_error("batch must be positive");
      if (cs.force_size != DC::FS)
        throw std::runtime_error("force size in settings does not match reference force size");
      if (rm->nfeet != DC::NF || rm->njoints != DK::NJ)
        throw std::runtime_error("robot shape (njoints, nfeet) does not match this kernel instantiation");
      if ((int)cs.w_u.size() != DC::NU * DC::NU || cs.w_com.size() != 9 || cs.w_linear_mom.size() != 9 || cs.w_angular_mom.size() != 9
          || cs.w_linear_acc.size() != 9 || cs.w_angular_acc.size() != 9)
        throw std::runtime_error("centroidal settings: weight sizes do not match the robot");
      if (ms.T < 2)
        throw std::runtime_error("horizon must have at least 2 stages");
      if (DC::FS == 6 && ms.T + 1 > 256) // (before anything is allocated: a constructor that throws runs no destructor)
        throw std::runtime_error("centroidal OCP with 6-D feet: at most 255 stages");
      device_id = device;
      set_device(device);
      stream = stream_create();
      B = batch;
      H = ms.T;
      R = H + 1;
      nu = DC::NU;
      nc = DC::NC;
      nf = DC::NF;
      nq_mb = DK::NQ;
      nv_mb = DK::NV;
      mass = rm->total_mass;
      // ---- models ----
      std::vector<DevModel<DK>> hk(1);
      std::memset(&hk[0], 0, sizeof(DevModel<DK>));
      fill_tree_model<DK>(rm, hk[0]);
      fk.model = (DevModel<DK> *)dev_alloc(sizeof(DevModel<DK>));
      h2d(fk.model, hk.data(), sizeof(DevModel<DK>), stream);
      std::vector<CentDevModel<DC>> hc(1);
      CentDevModel<DC> & m = hc[0];
      std::memset(&m, 0, sizeof(m));
      m.mass = mass;
      m.dt = cs.timestep;
      m.mu = ms.mu_init;
      m.mu_fric = cs.mu;
      m.cone_eps = 1e-4; // src/centroidal-dynamics.cpp:96
      m.Lfoot = cs.Lfoot;
      m.Wfoot = cs.Wfoot;
      for (int i = 0; i < 3; i++)
        m.gravity[i] = cs.gravity[i];
      std::copy(cs.w_com.begin(), cs.w_com.end(), m.w_com);
      std::copy(cs.w_linear_mom.begin(), cs.w_linear_mom.end(), m.w_lm);
      std::copy(cs.w_angular_mom.begin(), cs.w_angular_mom.end(), m.w_am);
      std::copy(cs.w_linear_acc.begin(), cs.w_linear_acc.end(), m.w_la);
      std::copy(cs.w_angular_acc.begin(), cs.w_angular_acc.end(), m.w_aa);
      std::copy(cs.w_u.begin(), cs.w_u.end(), m.w_u);
      for (int f = 0; f < DC::NF; f++)
        for (int i = 0; i < 3; i++)
          m.foot_ref_p[f][i] = rm->foot_ref_p[f][i];
      // ---- buffers ----
      buf.B = B;
      buf.H = H;
      buf.R = R;
      auto dalloc = [&](size_t n) { return (double *)dev_alloc(n * sizeof(double)); };
      const size_t BR = (size_t)B * R, BH = (size_t)B * H;
      buf.xs = dalloc(BR * 9);
      buf.us = dalloc(BR * DC::NU);
      buf.vs = dalloc(BR * DC::NC);
      buf.lams = dalloc(BR * 9);
      buf.vs_e = dalloc(BR * DC::NC);
      buf.lams_e = dalloc(BR * 9);
      buf.dxs = dalloc((size_t)B * (H + 1) * 9);
      // (steps: [B][H][.] linear in t for the one-kernel / 6-D paths; the kernel pipeline of point feet keeps them on the iterate's ring slots, [B][R][.])
      buf.dus = dalloc(BR * DC::NU);
      buf.dvs = dalloc(BR * DC::NC);
      buf.dlams = dalloc(BR * 9);
      buf.foot = dalloc(BH * DC::NF * 3);
      buf.ftraj = dalloc((size_t)B * DC::NF * 6);
      buf.vbase = dalloc((size_t)B * 6);
      buf.vref = dalloc(BR * 6);
      buf.gains = dalloc(BH * DC::G_STRIDE);
      if constexpr (DC::FS == 6)
      {
        typedef typename DC::DD DD;
        x6.sb.B = B;
        x6.sb.H = H;
        x6.sb.R = R;
        x6.sb.lq = dalloc(BH * DD::LQ_STRIDE);
        x6.sb.gains = buf.gains; // (the dense gains block: [K | k] rows at G_K with stride GKS, read by the interpolation / gain read-out kernels)
        x6.sb.QN = dalloc((size_t)B * DD::NDX * DD::NDX);
        x6.sb.qN = dalloc((size_t)B * DD::NDX);
        x6.sb.model = (DevModel<DD> *)dev_alloc(sizeof(DevModel<DD>));
        DevModel<DD> hm;
        hm.mu = ms.mu_init;
        h2d(x6.sb.model, &hm, sizeof(hm), stream);
        stream_sync(stream);
        x6.parts0 = dalloc((size_t)B * (H + 1) * 4);
        x6.partsT = dalloc((size_t)B * DC::LS_N * (H + 1) * 2);
        x6.xdotT = dalloc((size_t)B * DC::LS_N * 18);
        ls_passes = (xcheck_env("SMPC_CENT6_LS_PASSES") && xcheck_env("SMPC_CENT6_LS_PASSES")[0] == '2' && DC::LS_N > 1) ? 2 : 1;
      }
      if constexpr (DC::FS == 3)
      {
        fused = xcheck_env("SMPC_CENT_FUSED") != nullptr && xcheck_env("SMPC_CENT_FUSED")[0] == '1';
        if (!fused)
        {
          if (BH * CentRec<DC>::STRIDE >= ((size_t)1 << 32))
            throw std::runtime_error("centroidal OCP: batch x horizon too large for the 32-bit record offsets");
          sbuf.rec = dalloc(BH * CentRec<DC>::STRIDE);
          sbuf.term = dalloc((size_t)B * CentRec<DC>::T_STRIDE);
          const char * le = xcheck_env("SMPC_CENT_LS");
          ls_direct = le != nullptr && le[0] == 'd';
          const char * pe = std::getenv("SMPC_CENT_PARTS");
          nparts = pe ? std::atoi(pe) : (B >= 1024 ? 2 : 1);
          if (nparts < 1 || nparts > 8 || B < 64 * nparts)
            nparts = 1;
          ev_fork = event_create();
          for (int p = 1; p < nparts; p++)
            part_stream.push_back(stream_create());
          for (int p = 0; p < 2 * nparts; p++)
            part_event.push_back(event_create());
        }
      }
      buf.scal = dalloc((size_t)B * SC_N);
      buf.xdot01 = dalloc((size_t)B * 18);
      buf.zeros = dalloc(64);
      if (std::getenv("SMPC_PHASE_PROFILE"))
        buf.dbg = dalloc(64);
      buf.stages = (CentStage<DC> *)dev_alloc((size_t)H * sizeof(CentStage<DC>));
      buf.model = (CentDevModel<DC> *)dev_alloc(sizeof(CentDevModel<DC>));
      X_dev = dalloc((size_t)B * DK::NX);
      cstate_dev = dalloc((size_t)B * 9);
      feet_dev = dalloc((size_t)B * DC::NF * 3);
      h2d(buf.model, hc.data(), sizeof(CentDevModel<DC>), stream);
      stream_sync(stream);
      x_model_ref.assign(DK::NX, 0.0);
      for (int i = 0; i < DK::NQ; i++)
        x_model_ref[i] = rm->q_ref[i];

      // ---- default problem (OCPHandler::createProblem, src/ocp-handler.cpp:96-137): all feet in contact, identity
      //      contact poses, zero references ----
      CentStage<DC> def;
      std::memset(&def, 0, sizeof(def));
      def.mask = (1u << DC::NF) - 1u;
      for (int f = 0; f < DC::NF; f++)
        def.u_ref[DC::FS * f + 2] = -mass * gravity_arg / (double)DC::NF;
      horizon.assign(H, def);
      standing = def;
      cold_solve(def);
      ref_rot.init(H, DC::NF);
      ctor_scope.commit();
    }
    ~CentEngine()
    {
      for (double * p : {buf.xs, buf.us, buf.vs, buf.lams, buf.vs_e, buf.lams_e, buf.dxs, buf.dus, buf.dvs, buf.dlams, buf.foot, buf.ftraj, buf.vbase, buf.vref, buf.gains, buf.scal, buf.xdot01,
                         buf.zeros, buf.dbg, X_dev, cstate_dev, feet_dev, stage_out, sbuf.rec, sbuf.term})
        dev_free(p);
      if (ev_handoff_valid)
        event_destroy(ev_handoff);
      if (!part_event.empty())
      {
        event_destroy(ev_fork);
        for (auto & e : part_event)
          event_destroy(e);
        for (auto & st : part_stream)
          stream_destroy(st);
      }
      if constexpr (DC::FS == 6)
      {
        for (double * p : {x6.sb.lq, x6.sb.QN, x6.sb.qN, x6.parts0, x6.partsT, x6.xdotT})
          dev_free(p);
        dev_free(x6.sb.model);
      }
      dev_free(buf.stages);
      dev_free(buf.model);
      dev_free(fk.model);
      stream_destroy(stream);
    }
    CentEngine(const CentEngine &) = delete;
    CentEngine & operator=(const CentEngine &) = delete;

    template <class Args, void (*Body)(const Args &, int), int NT, int MINW = 1>
    void timed_launch(int kid, int grid, const Args & a, bool aux = false, const stream_t * on = nullptr)
    {
      set_device(device_id);
      const stream_t st = on ? *on : stream;
      event_t e0{}, e1{};
      if (profiling)
      {
        e0 = event_create();
        e1 = event_create();
        event_record(e0, st);
      }
      if (aux)
        launch<Args, Body, NT, MINW, 1>(grid, st, a);
      else
        launch<Args, Body, NT, MINW, 0>(grid, st, a);
      if (profiling)
      {
        event_record(e1, st);
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

    void upload_stages()
    {
      stage_ring.upload(buf.stages, horizon.data(), (size_t)H * sizeof(CentStage<DC>), stream);
    }
    void launch_frontend(const double * Xd, bool aux = false, int inst0 = 0, int count = -1, const stream_t * on = nullptr)
    {
      if (count < 0)
        count = B;
      FrontendArgs<DK> fa;
      fa.b = fk;
      fa.X = Xd + (size_t)inst0 * DK::NX;
      fa.feet = feet_dev + (size_t)inst0 * DC::NF * 3;
      fa.com = nullptr;
      fa.hg = nullptr;
      fa.cstate = cstate_dev + (size_t)inst0 * 9;
      if constexpr (cent_is_full_dims<DK>::value)
        timed_launch<FrontendArgs<DK>, frontend_full_body<DK>, 64>(CKID_FRONTEND, count, fa, aux, on);
      else
        timed_launch<FrontendArgs<DK>, frontend_body<DK>, 64>(CKID_FRONTEND, count, fa, aux, on);
    }
    // first instance / instance count of part p (whole wavefront groups of 64)
    void part_range(int p, int np, int & i0, int & n) const
    {
      const int per = ((B + np - 1) / np + 63) / 64 * 64;
      i0 = p * per < B ? p * per : B;
      n = i0 + per <= B ? per : B - i0;
    }
    // the kernel pipeline of one part of the batch, on stream `on` (null: the engine's stream); with_frontend: the state front end of the part first
    void launch_split_part(const CentStepArgs<DC> & a, bool aux, int inst0, int count, const stream_t * on, const double * Xfront, event_t * after_first_pre, const event_t * wait_before_pre)
    {
      if (count <= 0)
        return;
      if (Xfront)
        launch_frontend(Xfront, aux, inst0, count, on);
      CentSplitArgs<DC> c;
      c.a = a;
      c.sb = sbuf;
      c.last = 0;
      c.inst0 = inst0;
      timed_launch<CentSplitArgs<DC>, cent_recede_body<DC>, 64>(CKID_STEP, count, c, aux, on);
      if (wait_before_pre)
        stream_wait_event(on ? *on : stream, *wait_before_pre);
      for (int it = 0; it < a.iters; it++)
      {
        c.last = it + 1 == a.iters ? 1 : 0;
        timed_launch<CentSplitArgs<DC>, cent_pre_body<DC>, 64, 2>(CKID_DERIV, count, c, aux, on);
        if (it == 0 && after_first_pre)
          event_record(*after_first_pre, on ? *on : stream);
        if (buf.dbg != nullptr)
          timed_launch<CentSplitArgs<DC>, cent_bwd_body<DC, true>, 64, 4>(CKID_RICCATI, count, c, aux, on);
        else
          timed_launch<CentSplitArgs<DC>, cent_bwd_body<DC, false>, 64, 4>(CKID_RICCATI, count, c, aux, on);
        timed_launch<CentSplitArgs<DC>, cent_fwd_body<DC>, 64, 4>(CKID_FORWARD, count, c, aux, on);
        // (line search: the polynomial form needs one lane per stage and the terminal node; longer horizons re-evaluate per candidate)
        if (H + 1 <= 64 && !ls_direct)
          timed_launch<CentSplitArgs<DC>, cent_ls_poly_body<DC>, 64, 2>(CKID_LS, count, c, aux, on);
        else
          timed_launch<CentSplitArgs<DC>, cent_ls_body<DC>, 64, 1>(CKID_LS, count, c, aux, on);
      }
    }
    // one solver run of `a.iters` iterations (with the recede / centre bookkeeping the flags of `a` ask for)
    void launch_step(const CentStepArgs<DC> & a, bool aux = false)
    {
      if constexpr (DC::FS == 6)
      {
        Cent6Args<DC> c;
        c.b = buf;
        c.sb = x6.sb;
        c.parts0 = x6.parts0;
        c.partsT = x6.partsT;
        c.xdotT = x6.xdotT;
        c.head = a.head;
        c.shift = a.shift;
        c.set_centres = a.set_centres;
        c.reset_preg = a.reset_preg;
        c.X = a.X;
        c.nx_mb = a.nx_mb;
        c.cstate = a.cstate;
        c.feet = a.feet;
        for (int f = 0; f < DC::NF; f++)
          c.land[f] = a.land[f];
        c.T_fly = a.T_fly;
        c.T_contact = a.T_contact;
        c.swing_apex = a.swing_apex;
        c.timestep = a.timestep;
        c.armijo_c1 = a.armijo_c1;
        c.reg_init = a.reg_init;
        c.reg_min = a.reg_min;
        c.reg_max = a.reg_max;
        c.reg_inc = a.reg_inc;
        c.reg_dec = a.reg_dec;
        typedef typename DC::DD DD;
        SolverArgs<DD> sa;
        sa.b = x6.sb;
        sa.head = a.head;
        sa.j0 = 0;
        sa.nj = 0;
        sa.armijo_c1 = a.armijo_c1;
        sa.reg_min = a.reg_min;
        sa.reg_max = a.reg_max;
        sa.reg_inc = a.reg_inc;
        sa.reg_dec = a.reg_dec;
        timed_launch<Cent6Args<DC>, cent6_recede_body<DC>, 64>(CKID_STEP, B, c, aux);
        for (int it = 0; it < a.iters; it++)
        {
          timed_launch<Cent6Args<DC>, cent6_deriv_body<DC>, 64>(CKID_DERIV, B * (H + 1), c, aux);
          timed_launch<SolverArgs<DD>, riccati_dense_body<DD>, 64, 2>(CKID_RICCATI, B, sa, aux);
          timed_launch<Cent6Args<DC>, cent6_forward_body<DC>, 64>(CKID_FORWARD, B, c, aux);
          // line search: every candidate in one pair of launches -- a launch of B (H + 1) one-wave blocks costs 0.3 ms whatever it evaluates (measured:
          // the full step alone 0.30 ms, the nine backtracking candidates 0.32 ms), so the split `full step first, the rest where it failed'
          // (SMPC_CENT6_LS_PASSES=2; the kernels take any candidate range) only pays when no instance backtracks
          for (int pass = 0; pass < ls_passes; pass++)
          {
            c.j0 = pass;
            c.nj = ls_passes == 1 ? DC::LS_N : (pass == 0 ? 1 : DC::LS_N - 1);
            timed_launch<Cent6Args<DC>, cent6_trial_body<DC>, 64>(CKID_TRIAL, B * (H + 1), c, aux);
            timed_launch<Cent6Args<DC>, cent6_ls_body<DC>, 64>(CKID_LS, B, c, aux);
          }
        }
      }
      else if (fused)
      {
        if constexpr (kCrossCheck) // (the one-kernel form: SMPC_CENT_FUSED=1 of a cross-check build)
          timed_launch<CentStepArgs<DC>, cent_step_body<DC>, 64, 2>(CKID_STEP, B, a, aux);
      }
      else
        launch_split_part(a, aux, 0, B, nullptr, nullptr, nullptr, nullptr);
    }
    CentStepArgs<DC> step_args(const double * Xd) const
    {
      CentStepArgs<DC> a;
      a.b = buf;
      a.head = head;
      a.shift = 0;
      a.set_centres = 0;
      a.reset_preg = 0;
      a.iters = 1;
      a.X = Xd;
      a.nx_mb = DK::NX;
      a.cstate = cstate_dev;
      a.feet = feet_dev;
      for (int f = 0; f < DC::NF; f++)
        a.land[f] = -1;
      a.T_fly = ms.T_fly;
      a.T_contact = ms.T_contact;
      a.swing_apex = ms.swing_apex;
      a.timestep = ms.timestep;
      a.armijo_c1 = ARMIJO_C1;
      a.reg_init = REG_INIT;
      a.reg_min = REG_MIN;
      a.reg_max = REG_MAX;
      a.reg_inc = REG_INC;
      a.reg_dec = REG_DEC;
      return a;
    }

    // MPC constructor (src/mpc.cpp:62-91): every instance starts from the solution of the default problem at the
    // reference state.  All B wavefronts run the same cold solve (it is identical work, and cheaper than a broadcast).
    void cold_solve(const CentStage<DC> & def)
    {
      std::vector<double> X((size_t)B * DK::NX);
      for (int b = 0; b < B; b++)
        std::copy(x_model_ref.begin(), x_model_ref.end(), X.begin() + (size_t)b * DK::NX);
      h2d(X_dev, X.data(), X.size() * sizeof(double), stream);
      launch_frontend(X_dev, true);
      std::vector<double> cst((size_t)B * 9), feet((size_t)B * DC::NF * 3);
      d2h(cst.data(), cstate_dev, cst.size() * sizeof(double), stream);
      d2h(feet.data(), feet_dev, feet.size() * sizeof(double), stream);
      stream_sync(stream);
      std::vector<double> xs((size_t)B * R * 9), us((size_t)B * R * DC::NU), ft((size_t)B * DC::NF * 6);
      for (int b = 0; b < B; b++)
      {
        for (int t = 0; t < R; t++)
        {
          std::copy(cst.begin() + (size_t)b * 9, cst.begin() + (size_t)(b + 1) * 9, xs.begin() + ((size_t)b * R + t) * 9);
          std::copy(def.u_ref, def.u_ref + DC::NU, us.begin() + ((size_t)b * R + t) * DC::NU);
        }
        for (int f = 0; f < DC::NF; f++)
          for (int i = 0; i < 3; i++)
            ft[((size_t)b * DC::NF + f) * 6 + i] = ft[((size_t)b * DC::NF + f) * 6 + 3 + i] = feet[((size_t)b * DC::NF + f) * 3 + i];
      }
      h2d(buf.xs, xs.data(), xs.size() * sizeof(double), stream);
      h2d(buf.us, us.data(), us.size() * sizeof(double), stream);
      h2d(buf.ftraj, ft.data(), ft.size() * sizeof(double), stream);
      upload_stages();
      stream_sync(stream);
      bool centres = true;
      std::vector<double> sc(SC_N);
      for (int it = 0; it < 100; it++)
      {
        CentStepArgs<DC> a = step_args(X_dev);
        a.set_centres = centres ? 1 : 0;
        a.reset_preg = it == 0 ? 1 : 0;
        launch_step(a, true);
        d2h(sc.data(), buf.scal, SC_N * sizeof(double), stream);
        stream_sync(stream);
        cold_iters = it + 1;
        cold_trace.insert(cold_trace.end(), {sc[SC_PHI0], sc[SC_PRIM], sc[SC_DUAL], sc[SC_ALPHA]});
        centres = false;
        if (std::fmax(sc[SC_PRIM], sc[SC_DUAL]) <= ms.TOL)
          break;
        if (std::fabs(sc[SC_DPHI0]) <= STALL_REL * std::fmax(1.0, std::fabs(sc[SC_PHI0])))
          break;
        if (sc[SC_DUAL] <= ms.TOL)
          centres = true;
      }
      for (int i = 0; i < CKID_N; i++)
        kernel_calls[i] = 0;
    }

    void generate_cycle_horizon(const unsigned char * cs, int n) override
    {
      if (n <= 0)
        throw std::runtime_error("contact sequence must not be empty");
      timer.generate(cs, n, DC::NF, H);
      cycle.clear();
      for (auto & st : timer.states)
      {
        int active = 0;
        for (int f = 0; f < DC::NF; f++)
          active += st[f] ? 1 : 0;
        CentStage<DC> s;
        std::memset(&s, 0, sizeof(s));
        for (int f = 0; f < DC::NF; f++)
          if (st[f])
          {
            s.mask |= 1u << f;
            s.u_ref[DC::FS * f + 2] = ms.support_force / (double)active;
          }
        for (int i = 0; i < 3; i++)
          s.x_tgt[i] = com_ref_member[i];
        cycle.push_back(s);
      }
    }
    void upload_velocity(const double * V, bool broadcast)
    {
      set_device(device_id);
      std::vector<double> h((size_t)B * 6);
      for (int b = 0; b < B; b++)
        for (int i = 0; i < 6; i++)
          h[(size_t)b * 6 + i] = broadcast ? V[i] : V[(size_t)b * 6 + i];
      h2d(buf.vbase, h.data(), h.size() * sizeof(double), stream);
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
      for (int f = 0; f < DC::NF; f++)
        last_support += (horizon[H - 1].mask >> f) & 1u;
      CentStage<DC> incoming;
      if (walking || last_support < DC::NF)
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
      // setReferenceState(H-1, x_reference_) ; setVelocityBase(H-1, velocity_base_): momentum references are m v
      for (int i = 0; i < 3; i++)
      {
        horizon[H - 1].x_tgt[i] = x_reference[i];
        com_ref_member[i] = x_reference[i];
        horizon[H - 1].x_tgt[3 + i] = mass * velocity_base[i];
        horizon[H - 1].x_tgt[6 + i] = mass * velocity_base[3 + i];
      }
      upload_stages();
      head = head + 1 == R ? 0 : head + 1;
      CentStepArgs<DC> a = step_args(Xd);
      a.shift = 1;
      a.set_centres = 1;
      a.reset_preg = 1;
      a.iters = ms.max_iters;
      for (int f = 0; f < DC::NF; f++)
        a.land[f] = timer.land[f].empty() ? -1 : timer.land[f][0];
      if constexpr (DC::FS == 3)
      {
        if (!fused && nparts > 1 && !profiling) // (per-kernel event times mean something only when launches do not overlap: one part while profiling)
        {
          // fork: every part's stream sees what the engine's stream has done so far (the stage table upload, the caller's writes of Xd)
          event_record(ev_fork, stream);
          for (int p = 0; p < nparts; p++)
          {
            int i0, n;
            part_range(p, nparts, i0, n);
            const stream_t * on = p == 0 ? nullptr : &part_stream[p - 1];
            if (p > 0)
              stream_wait_event(*on, ev_fork);
            launch_split_part(a, false, i0, n, on, Xd, &part_event[nparts + p], p > 0 ? &part_event[nparts + p - 1] : nullptr);
          }
          // join
          for (int p = 1; p < nparts; p++)
          {
            event_record(part_event[p - 1], part_stream[p - 1]);
            stream_wait_event(stream, part_event[p - 1]);
          }
          return;
        }
      }
      launch_frontend(Xd);
      launch_step(a);
    }
    // ---- per-stage references (OCPHandler setters / getters of the centroidal OCP, reference src/centroidal-dynamics.cpp:
    //      120-304), broadcast over the batch ----
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
    // what: 0 = control target (nu); 1 = reference state as get / setReferenceState define it: [com_ref; v_lin; v_ang]
    // (setReferenceState = setPoseBase + setVelocityBase, which stores the momenta m v; getReferenceState divides by m)
    void set_stage_reference(int t, int what, const double * v, int n) override
    {
      check_stage(t);
      if (what == 0)
      {
        if (n != DC::NU)
          throw std::runtime_error("u_ref not of the right size");
        std::copy(v, v + n, horizon[t].u_ref);
      }
      else if (what == 1)
      {
        if (n != 9)
          throw std::runtime_error("x_ref not of the right size");
        double mv[6];
        for (int i = 0; i < 3; i++)
        {
          horizon[t].x_tgt[i] = v[i];
          com_ref_member[i] = v[i]; // setPoseBase updates CentroidalOCP::com_ref_
        }
        for (int i = 0; i < 6; i++)
          horizon[t].x_tgt[3 + i] = mv[i] = mass * v[3 + i];
        fill_strided(buf.vref + (size_t)ring_slot(head, t, R) * 6, (size_t)R * 6, B, mv, 6);
      }
      else
        throw std::runtime_error("unknown stage reference");
    }
    void get_stage_reference(int t, int what, double * v, int n) override
    {
      check_stage(t);
      if (what == 0 && n == DC::NU)
        std::copy(horizon[t].u_ref, horizon[t].u_ref + n, v);
      else if (what == 1 && n == 9)
      {
        double mv[6];
        get_linear(buf.vref + (size_t)ring_slot(head, t, R) * 6, 6, mv); // instance 0
        for (int i = 0; i < 3; i++)
          v[i] = horizon[t].x_tgt[i];
        for (int i = 0; i < 6; i++)
          v[3 + i] = mv[i] / mass;
      }
      else
        throw std::runtime_error("unknown stage reference or wrong size");
    }
    void set_reference_pose(int t, int foot, const double * p3) override
    {
      check_stage(t);
      if (foot < 0 || foot >= DC::NF)
        throw std::runtime_error("unknown end effector");
      ref_rot.set(t, foot, nullptr); // (a translation: identity rotation)
      fill_strided(buf.foot + ((size_t)t * DC::NF + foot) * 3, (size_t)H * DC::NF * 3, B, p3, 3);
    }
    void get_reference_pose(int t, int foot, int inst, double * p3) override
    {
      check_stage(t);
      if (foot < 0 || foot >= DC::NF || inst < 0 || inst >= B)
        throw std::runtime_error("unknown end effector or instance");
      get_linear(buf.foot + (((size_t)inst * H + t) * DC::NF + foot) * 3, 3, p3);
    }
    RefRotations ref_rot; // rotations of the foot reference placements: API state (smpc_model.h)
    void set_reference_rotation(int t, int foot, const double * R9) override
    {
      check_stage(t);
      if (foot < 0 || foot >= DC::NF)
        throw std::runtime_error("unknown end effector");
      ref_rot.set(t, foot, R9);
    }
    void get_reference_rotation(int t, int foot, double * R9) override
    {
      check_stage(t);
      if (foot < 0 || foot >= DC::NF)
        throw std::runtime_error("unknown end effector");
      ref_rot.get(t, foot, R9);
    }
    unsigned contact_mask(int t) const override
    {
      check_stage(t);
      return horizon[t].mask;
    }

    // state feedback front-end on measured states X [B][nq + nv] (host): host outputs, any may be null
    void update_internal_data(const double * X, double * feet, double * com, double * hg, double * cstate) override
    {
      const size_t nf = (size_t)B * DC::NF * 3, nc = (size_t)B * 3, nh = (size_t)B * 6, ns = (size_t)B * 9;
      double * st = staging((nf + nc + nh + ns) * sizeof(double));
      h2d(X_dev, X, (size_t)B * DK::NX * sizeof(double), stream);
      FrontendArgs<DK> fa;
      fa.b = fk;
      fa.X = X_dev;
      fa.feet = st;
      fa.com = st + nf;
      fa.hg = st + nf + nc;
      fa.cstate = st + nf + nc + nh;
      if constexpr (cent_is_full_dims<DK>::value)
        launch<FrontendArgs<DK>, frontend_full_body<DK>, 64, 1, 1>(B, stream, fa);
      else
        launch<FrontendArgs<DK>, frontend_body<DK>, 64, 1, 1>(B, stream, fa);
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
    // interpolated targets at `delay` after the last solve (host outputs, any may be null): x [B][9], xdot [B][9],
    // forces [B][NU]; with X_meas (measured multibody states [B][nq + nv]) also the Riccati feedback u [B][NU]
    // targets of a CentroidalID controller written into its device buffers; asynchronous on this engine's stream
    void interpolate_device_id(double delay, int knots, double * com, double * vcom, double * fp, double * fv, double * f) override
    {
      if (knots < 2 || knots > H + 1)
        throw std::runtime_error("interpolate: knots must be in [2, horizon + 1]");
      if (!(delay >= 0.0))
        throw std::runtime_error("interpolate: delay must be non-negative");
      set_device(device_id);
      CentInterpArgs<DC> ia;
      ia.b = buf;
      ia.head = head;
      ia.knots = knots;
      ia.delay = delay;
      ia.timestep = ms.timestep;
      ia.x_meas = nullptr;
      ia.x_out = nullptr;
      ia.xdot_out = nullptr;
      ia.f_out = f;
      ia.u_out = nullptr;
      ia.com_out = com;
      ia.vcom_out = vcom;
      ia.fp_out = fp;
      ia.fv_out = fv;
      ia.mass = mass;
      launch<CentInterpArgs<DC>, cent_interp_body<DC>, 64>(B, stream, ia);
    }
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
    event_t ev_handoff{};
    bool ev_handoff_valid = false;
    void interpolate(double delay, int knots, const double * X_meas, double * x_out, double * xdot_out, double * f_out, double * u_out) override
    {
      if (knots < 2 || knots > H + 1)
        throw std::runtime_error("interpolate: knots must be in [2, horizon + 1]");
      if (!(delay >= 0.0))
        throw std::runtime_error("interpolate: delay must be non-negative");
      const size_t n9 = (size_t)B * 9, nu = (size_t)B * DC::NU;
      double * st = staging((2 * n9 + 2 * nu) * sizeof(double));
      if (X_meas)
      {
        h2d(X_dev, X_meas, (size_t)B * DK::NX * sizeof(double), stream);
        launch_frontend(X_dev, true);
      }
      CentInterpArgs<DC> ia;
      ia.b = buf;
      ia.head = head;
      ia.knots = knots;
      ia.delay = delay;
      ia.timestep = ms.timestep;
      ia.x_meas = X_meas ? cstate_dev : nullptr;
      ia.x_out = st;
      ia.xdot_out = st + n9;
      ia.f_out = st + 2 * n9;
      ia.u_out = X_meas ? st + 2 * n9 + nu : nullptr;
      launch<CentInterpArgs<DC>, cent_interp_body<DC>, 64>(B, stream, ia);
      if (x_out)
        d2h(x_out, st, n9 * sizeof(double), stream);
      if (xdot_out)
        d2h(xdot_out, st + n9, n9 * sizeof(double), stream);
      if (f_out)
        d2h(f_out, st + 2 * n9, nu * sizeof(double), stream);
      if (u_out && X_meas)
        d2h(u_out, st + 2 * n9 + nu, nu * sizeof(double), stream);
      stream_sync(stream);
    }

    size_t state_io(StateIO & io) override
    {
      set_device(device_id);
      stream_sync(stream);
      io.tag(0x534d504343454e54LL, "kind (centroidal)");
      io.tag(B, "batch");
      io.tag(H, "horizon");
      io.tag(DC::NU, "nu");
      io.pod(head);
      io.pod(walking);
      io.host(velocity_base, sizeof(velocity_base));
      io.host(x_reference, sizeof(x_reference));
      io.host(com_ref_member, sizeof(com_ref_member));
      io.vec(horizon);
      io.vec(cycle);
      io.timer(timer);
      const size_t BR = (size_t)B * R;
      io.dev(buf.xs, BR * 9 * sizeof(double));
      io.dev(buf.us, BR * DC::NU * sizeof(double));
      io.dev(buf.vs, BR * DC::NC * sizeof(double));
      io.dev(buf.lams, BR * 9 * sizeof(double));
      io.dev(buf.ftraj, (size_t)B * DC::NF * 6 * sizeof(double));
      io.dev(buf.foot, (size_t)B * H * DC::NF * 3 * sizeof(double));
      io.dev(buf.vbase, (size_t)B * 6 * sizeof(double));
      io.dev(buf.vref, BR * 6 * sizeof(double));
      io.dev(buf.scal, (size_t)B * SC_N * sizeof(double));
      io.dev(buf.xdot01, (size_t)B * 18 * sizeof(double));
      if (io.mode == StateIO::LOAD)
        upload_stages();
      stream_sync(stream);
      return io.pos;
    }
    void iterate_host(const double * X) override
    {
      set_device(device_id);
      h2d(X_dev, X, (size_t)B * DK::NX * sizeof(double), stream);
      iterate_device(X_dev);
      stream_sync(stream);
    }
    void sync() override
    {
      set_device(device_id);
      stream_sync(stream);
    }

    double * stage_out = nullptr;
    size_t stage_out_bytes = 0;
    double * staging(size_t bytes)
    {
      set_device(device_id);
      if (bytes > stage_out_bytes)
      {
        stream_sync(stream);
        dev_free(stage_out);
        stage_out = (double *)dev_alloc(bytes);
        stage_out_bytes = bytes;
      }
      return stage_out;
    }
    void get_K(double * out, bool all) override
    {
      const int nt = all ? H : 1;
      const size_t n = (size_t)B * nt * DC::NU * 9;
      double * dev = staging(n * sizeof(double));
      CentGainsOutArgs<DC> ga;
      ga.b = buf;
      ga.out = dev;
      ga.all = all ? 1 : 0;
      launch<CentGainsOutArgs<DC>, cent_gains_out_body<DC>, 64>(B * nt, stream, ga);
      d2h(out, dev, n * sizeof(double), stream);
      stream_sync(stream);
    }
  };
} // namespace smpc
