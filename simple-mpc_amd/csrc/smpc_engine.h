// smpc_engine.h -- host side of the batched kinodynamics MPC engine: owns the device buffers, the
// shared (phase-aligned) gait state machine and the launch sequence of one control step.
//
// Mirrors, for a batch of B instances, the reference's MPC class:
//   MPC::MPC                   src/mpc.cpp:19-99      -> KinoEngine::KinoEngine (cold solve once, broadcast)
//   MPC::generateCycleHorizon  src/mpc.cpp:101-187    -> generate_cycle_horizon
//   MPC::iterate               src/mpc.cpp:189-218    -> iterate
//   MPC::recedeWithCycle       src/mpc.cpp:220-254    -> recede_host (+ ring head increment)
//   MPC::updateCycleTiming     src/mpc.cpp:256-276    -> GaitTimer::update_timing
//   MPC::switchToWalk/Stand    src/mpc.cpp:382-392
#pragma once
#ifndef SMPC_TRIAL_MINW
#define SMPC_TRIAL_MINW 3
#endif
#ifndef SMPC_LANE_MINW
#define SMPC_LANE_MINW 1
#endif
#ifndef SMPC_DERIV2_MINW
#define SMPC_DERIV2_MINW 2 // waves per SIMD the register allocation of deriv2_body is capped for (3: measured refusal)
#endif
#include "smpc_riccati_kino.h"
#include "smpc_kino_deriv2.h"
#include "smpc_solver_kernels.h"
#include "smpc_full_kernels.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

namespace smpc
{
  struct HostKinoSettings
  {
    double timestep;
    std::vector<double> w_x, w_u, w_frame, w_cent, w_centder, qmin, qmax;
    double gravity[3];
    int kinematics_limits;
    int terminal_constraint = 0; // createProblem(..., terminal_constraint)
    int force_cone = 0;          // friction-cone rows per foot in contact (3-D feet)
    int land_cstr = 0;           // height of a landing foot pinned to its contact pose
    double mu = 0.8;             // friction coefficient
  };
  struct HostMpcSettings
  {
    double swing_apex, support_force, TOL, mu_init, timestep;
    int max_iters, num_threads, T_fly, T_contact, T;
  };

  // CoM height at state x from the host copy of a device model (MPC::com0_ at the reference state, reference src/mpc.cpp:93)
  template <class M>
  double host_com_height(const M & m, const double * x)
  {
    constexpr int NJ = sizeof(m.mass) / sizeof(double);
    M3 Rj[NJ];
    V3 pj[NJ];
    double mt = 0.0, mz = 0.0;
    for (int j = 0; j < NJ; j++)
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
      const V3 cw = pj[j] + Rj[j] * ld3(m.com[j]);
      mt += m.mass[j];
      mz += m.mass[j] * cw.z;
    }
    return mz / mt;
  }

  // createTerminalConstraint(x0.head<3>()): tau = sqrt(x0_z / 9.81), reference = the base position until the first control step
  // (reference src/ocp-handler.cpp:133-136, src/kinodynamics.cpp:366-377); multipliers start at zero
  template <class D>
  void alloc_terminal_constraint(Buffers<D> & bf, const double * x0, double com_height, stream_t stream)
  {
    auto dalloc = [&](size_t n) {
      double * p = (double *)dev_alloc(n * sizeof(double));
      dev_zero(p, n * sizeof(double), stream);
      return p;
    };
    bf.CN = dalloc((size_t)bf.B * (3 * D::NDX + 3));
    bf.vN = dalloc((size_t)bf.B * 3);
    bf.vN_e = dalloc((size_t)bf.B * 3);
    bf.vN_b = dalloc((size_t)bf.B * 3);
    bf.dvN = dalloc((size_t)bf.B * 3);
    bf.dcm_ref = dalloc((size_t)bf.B * 3);
    bf.dcm_tau = std::sqrt(x0[2] / 9.81);
    bf.com0z = com_height;
    h2d(bf.dcm_ref, x0, 3 * sizeof(double), stream);
  }

  // integer gait bookkeeping (reference src/mpc.cpp:101-132, 220-276)
  struct GaitTimer
  {
    int H = 0, nf = 0;
    std::vector<std::vector<unsigned char>> states;
    std::vector<std::vector<int>> takeoff, land;
    void generate(const unsigned char * cs, int n, int nf_, int H_)
    {
      H = H_;
      nf = nf_;
      states.clear();
      const int reps = 1 + H / n; // original + m copies, m = H / n (integer division)
      for (int r = 0; r < reps; r++)
        for (int i = 0; i < n; i++)
          states.emplace_back(cs + (size_t)i * nf, cs + (size_t)(i + 1) * nf);
      takeoff.assign(nf, {});
      land.assign(nf, {});
      const int N = (int)states.size();
      for (int f = 0; f < nf; f++)
      {
        for (int i = 1; i < N; i++)
        {
          const bool now = states[i][f], prev = states[i - 1][f];
          if (!now && prev)
            takeoff[f].push_back(i + H);
          if (now && !prev)
            land[f].push_back(i + H);
        }
        if (states[N - 1][f] && !states[0][f])
          takeoff[f].push_back(N - 1 + H);
        if (!states[N - 1][f] && states[0][f])
          land[f].push_back(N - 1 + H);
      }
    }
    void update_timing(bool only_horizon)
    {
      for (int f = 0; f < nf; f++)
      {
        for (auto * v : {&land[f], &takeoff[f]})
        {
          for (int & t : *v)
            if (!only_horizon || t < H)
              t -= 1;
          if (!v->empty() && (*v)[0] < 0)
            v->erase(v->begin());
        }
      }
    }
    void recede_cycle()
    {
      std::rotate(states.begin(), states.begin() + 1, states.end());
      const int N = (int)states.size();
      for (int f = 0; f < nf; f++)
      {
        if (!states[N - 1][f] && states[N - 2][f])
          takeoff[f].push_back(N + H);
        if (states[N - 1][f] && !states[N - 2][f])
          land[f].push_back(N + H);
      }
      update_timing(false);
    }
  };

  enum KernelId
  {
    KID_RECEDE = 0,
    KID_DERIV,
    KID_RICCATI,
    KID_FORWARD,
    KID_TRIAL,
    KID_SELECT,
    KID_APPLY,
    KID_TREE,    // lane-per-problem tree pass (smpc_kino_lane.h) of the full-batch derivative launches
    KID_TREE_LS, // ... of the full-batch line-search launches (evaluation mode: heads only)
    KID_N
  };

  // checkpoint / resume of a handle (smpc_save_state / smpc_load_state): one pass over the state in a fixed order, in one
  // of three modes (count the bytes, copy out, copy in)
  struct StateIO
  {
    enum Mode
    {
      COUNT,
      SAVE,
      LOAD
    } mode;
    char * buf;
    size_t cap, pos = 0;
    stream_t st;
    StateIO(Mode m, void * b, size_t c, stream_t s) : mode(m), buf((char *)b), cap(c), st(s) {}
    void need(size_t n) const
    {
      if (mode != COUNT && pos + n > cap)
        throw std::runtime_error(mode == SAVE ? "state buffer too small" : "state buffer truncated");
    }
    void host(void * p, size_t n)
    {
      need(n);
      if (mode == SAVE)
        std::memcpy(buf + pos, p, n);
      else if (mode == LOAD)
        std::memcpy(p, buf + pos, n);
      pos += n;
    }
    void dev(void * p, size_t n)
    {
      need(n);
      if (mode == SAVE)
        d2h(buf + pos, p, n, st);
      else if (mode == LOAD)
        h2d(p, buf + pos, n, st);
      pos += n;
    }
    template <class T>
    void pod(T & v)
    {
      host(&v, sizeof(T));
    }
    // a value that must be the same in the handle and in the buffer (shape of the problem)
    void tag(long long v, const char * what)
    {
      long long w = v;
      pod(w);
      if (mode == LOAD && w != v)
        throw std::runtime_error(std::string("saved state does not match this handle: ") + what);
    }
    template <class T>
    void vec(std::vector<T> & v)
    {
      unsigned long long n = v.size();
      pod(n);
      if (mode == LOAD)
      {
        if (n * sizeof(T) > cap)
          throw std::runtime_error("state buffer corrupt");
        v.resize((size_t)n);
      }
      if (n)
        host(v.data(), (size_t)n * sizeof(T));
    }
    void timer(GaitTimer & t)
    {
      pod(t.H);
      pod(t.nf);
      unsigned long long ns = t.states.size();
      pod(ns);
      if (mode == LOAD)
        t.states.assign((size_t)ns, std::vector<unsigned char>());
      for (auto & s : t.states)
        vec(s);
      if (mode == LOAD)
      {
        t.takeoff.assign(t.nf, {});
        t.land.assign(t.nf, {});
      }
      for (int f = 0; f < t.nf; f++)
      {
        vec(t.takeoff[f]);
        vec(t.land[f]);
      }
    }
  };

  // kinematic tree, inertias and feet of the robot table -> device model (shared by the kinodynamics engine and the
  // front-end of the centroidal engine)
  template <class D>
  inline void fill_tree_model(const smpc_robot_model * rm, DevModel<D> & m)
  {
    if (rm->njoints != D::NJ || rm->nfeet != D::NF)
      throw std::runtime_error("robot shape (njoints, nfeet) does not match this kernel instantiation");
    int maxlev = 0;
    for (int j = 0; j < D::NJ; j++)
    {
      m.parent[j] = rm->parent[j];
      if (j > 0 && (rm->parent[j] < 0 || rm->parent[j] >= j))
        throw std::runtime_error("robot joints must be topologically ordered");
      m.jtype[j] = rm->jtype[j];
      m.level[j] = j == 0 ? 0 : m.level[rm->parent[j]] + 1;
      maxlev = std::max(maxlev, m.level[j]);
      m.anc[j] = (j == 0 ? 0u : m.anc[rm->parent[j]]) | (1u << j);
      if (j > 0)
        m.children[rm->parent[j]] |= 1u << j;
      for (int i = 0; i < 9; i++)
        m.jpR[j][i] = rm->jp_R[j][i];
      for (int i = 0; i < 3; i++)
      {
        m.jpp[j][i] = rm->jp_p[j][i];
        m.com[j][i] = rm->com[j][i];
      }
      m.mass[j] = rm->mass[j];
      for (int i = 0; i < 6; i++)
        m.inertia[j][i] = rm->inertia[j][i];
    }
    m.nlevels = maxlev + 1;
    for (int f = 0; f < D::NF; f++)
    {
      m.foot_joint[f] = rm->foot_joint[f];
      for (int i = 0; i < 3; i++)
      {
        m.foot_p[f][i] = rm->foot_p[f][i];
        m.foot_ref_p[f][i] = rm->foot_ref_p[f][i];
      }
    }
    m.total_mass = rm->total_mass;
  }

  // branch joints of the lane-per-problem evaluation (smpc_kino_lane.h): parents that are not the joint right before their child
  template <class D>
  inline void fill_lane_slots(const smpc_robot_model * rm, DevModel<D> & m)
  {
    int nslots = 0;
    bool ok = true;
    for (int j = 0; j < D::NJ; j++)
      m.par_slot[j] = m.save_slot[j] = -1;
    for (int j = 1; j < D::NJ; j++)
      if (rm->parent[j] != j - 1)
      {
        const int par = rm->parent[j];
        if (m.save_slot[par] < 0)
        {
          if (nslots == LANE_SLOTS)
          {
            ok = false;
            break;
          }
          m.save_slot[par] = nslots++;
        }
        m.par_slot[j] = m.save_slot[par];
      }
    m.lane_slots = ok ? std::max(nslots, 1) : 0;
  }

  template <class D>
  class KinoEngine
  {
  public:
    typedef Dims<D::NJ, D::NF> DD;
    Buffers<D> buf;
    int B, H, R, head = 0;
    int device_id = 0; // every entry point makes this the current device first: a process may hold handles on several GPUs
    HostMpcSettings ms;
    std::vector<StageShared<D>> horizon, cycle;
    StageShared<D> standing;
    GaitTimer timer;
    bool walking = true;
    double velocity_base[6] = {0, 0, 0, 0, 0, 0};
    std::vector<double> x_reference, x_model_ref;
    stream_t stream;
    // SMPC_STREAMS=2: the iterations of the two halves of the batch run on two streams, so that workgroups of the matrix-core bound
    // Riccati sweep of one half share the CUs with the VALU bound stage kernels of the other
    static constexpr int MAX_STREAMS = 4;
    stream_t streams[MAX_STREAMS] = {};  // streams[0] == stream
    stream_t cur{};      // stream of the launches being issued
    int n_streams = 1;
    int * und_lists[MAX_STREAMS] = {nullptr, nullptr, nullptr, nullptr}; // und_lists[0] == buf.und_list
    event_t ev_fork{}, ev_join[MAX_STREAMS] = {};
    double * X_dev = nullptr;
    double * stage_out = nullptr; // staging for linearised outputs
    size_t stage_out_bytes = 0;
    int cold_iters = 0;
    int riccati_nt = xcheck_env("SMPC_RICCATI_NT") ? std::atoi(xcheck_env("SMPC_RICCATI_NT")) : 128; // dense sweep: lanes per instance
    // SMPC_RICCATI=dense selects the model-independent sweep (A/B comparison and cross-check in the tests)
    bool structured_riccati = !(xcheck_env("SMPC_RICCATI") && std::string(xcheck_env("SMPC_RICCATI")) == "dense");
    std::vector<double> cold_trace; // [n][4] phi0, prim, dual, alpha
    // profiling
    bool profiling = false;
    static constexpr int LS_SLOTS = 64; // instance slots of the list-mode (backtracking) launches: 64 x (H+1) blocks when the list is empty
    bool speculative_ls = xcheck_env("SMPC_NO_SPECULATIVE_LS") == nullptr; // tentative full steps (run_iterations)
    bool early_exit_on_tol = false; // smpc_set_early_exit_on_tol: iterate() stops an instance's iterations once it is converged to TOL
    bool aux_launches = false; // true during the cold start: every launch uses the auxiliary kernel symbols
    // lane-per-problem stage evaluation (smpc_kino_lane.h) for problems without optional constraint blocks; SMPC_LANE_EVAL=0: the
    // wavefront-per-problem kernels throughout (A/B comparison)
    int lane_slots = 1;
    // The derivative pass starts from the lane-per-problem evaluation too (lane_tree_body + deriv2_body: 0.27 + 3.1 ms per launch at
    // B = 4096 against 4.3 - 4.7 ms for the one-kernel path); SMPC_LANE_DERIV=0: the one-kernel path (A/B comparison)
    bool lane_deriv = !(xcheck_env("SMPC_LANE_DERIV") && std::atoi(xcheck_env("SMPC_LANE_DERIV")) == 0);
    size_t handover_bytes = 0; // device memory of the lane hand-over (tiles + stream)
    bool lane_eval = !(xcheck_env("SMPC_LANE_EVAL") && std::atoi(xcheck_env("SMPC_LANE_EVAL")) == 0);
    bool lane_stream = !(xcheck_env("SMPC_LANE_STREAM") && std::atoi(xcheck_env("SMPC_LANE_STREAM")) == 0);
    bool stream_order_recorded = false;
    int foot_joint_h[D::NF] = {0}; // (host copy for deriv2_commit_code)
    double kernel_ms[KID_N] = {0};
    long kernel_calls[KID_N] = {0};
    std::vector<std::pair<int, std::pair<event_t, event_t>>> pending_events;
    static constexpr int TRIAL_MINW = SMPC_TRIAL_MINW; // waves per SIMD the trial kernel's register budget allows
    static constexpr int RICCATI_MINW = 2;             // the Riccati sweep is latency bound: 2 waves per SIMD (8 per CU, 19.8 KB LDS each)
    static constexpr double ARMIJO_C1 = 1e-4, REG_INIT = 1e-9, REG_MIN = 1e-10, REG_MAX = 1e9, REG_INC = 10.0, REG_DEC = 1.0 / 3.0, STALL_REL = 1e-9;

    KinoEngine(const smpc_robot_model * rm, const HostKinoSettings & ks, const HostMpcSettings & ms_, int batch, double gravity_arg, int device)
    : ms(ms_)
    {
      AllocScope ctor_scope; // (a throw below releases what was allocated so far: smpc_alloc_scope.h)
      if (rm->njoints != D::NJ || rm->nfeet != D::NF)
        throw std::runtime_error("robot shape (njoints, nfeet) does not match this kernel instantiation");
      if (batch <= 0)
        throw std::runtime_error("batch must be positive");
      if ((int)ks.w_x.size() != D::NDX * D::NDX || (int)ks.w_u.size() != D::NU * D::NU || (int)ks.w_frame.size() != 9 || (int)ks.w_cent.size() != 36
          || (int)ks.w_centder.size() != 36 || (int)ks.qmin.size() != D::NA || (int)ks.qmax.size() != D::NA)
        throw std::runtime_error("kinodynamics settings: weight / limit sizes do not match the robot");
      {
        // LDS layout the rigid-body kernel relies on: the velocity-product matrices run across the end of the evaluation part
        // into the start of the derivative part (KinoScratch: late block | WJl | JWJ)
        static KinoScratch<D, true> probe;
        if ((const char *)probe.WJl - (const char *)probe.cval != (std::ptrdiff_t)(KinoScratchEval<D>::LATE_DOUBLES * sizeof(double)))
          throw std::runtime_error("internal: KinoScratch layout is not contiguous across its two parts");
      }
      device_id = device;
      set_device(device);
      stream = stream_create();
      cur = stream;
      streams[0] = stream;
      if (std::getenv("SMPC_STREAMS"))
        n_streams = std::min(std::max(std::atoi(std::getenv("SMPC_STREAMS")), 1), (int)MAX_STREAMS);
      if (n_streams > 1)
      {
        ev_fork = event_create();
        for (int i = 1; i < n_streams; i++)
        {
          streams[i] = stream_create();
          ev_join[i] = event_create();
        }
      }
      B = batch;
      H = ms.T;
      R = H + 1;
      // ---- model table ----
      std::vector<DevModel<D>> hm(1);
      DevModel<D> & m = hm[0];
      std::memset(&m, 0, sizeof(m));
      fill_tree_model<D>(rm, m);
      fill_lane_slots<D>(rm, m);
      lane_slots = m.lane_slots;
      for (int f = 0; f < D::NF; f++)
        foot_joint_h[f] = m.foot_joint[f];
      m.dt = ks.timestep;
      for (int i = 0; i < 3; i++)
        m.gravity[i] = ks.gravity[i];
      std::copy(ks.w_x.begin(), ks.w_x.end(), m.w_x);
      std::copy(ks.w_u.begin(), ks.w_u.end(), m.w_u);
      for (int i = 0; i < D::NDX; i++)
        for (int j = 0; j < D::NDX; j++)
          m.w_xT[j * D::NDX + i] = m.w_x[i * D::NDX + j];
      for (int i = 0; i < D::NU; i++)
        for (int j = 0; j < D::NU; j++)
          m.w_uT[j * D::NU + i] = m.w_u[i * D::NU + j];
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
        m.w_diag = 0; // test hook: exercise the general path with diagonal data
      for (int i = 0; i < D::NDX; i++)
        m.wxd[i] = m.w_x[i * D::NDX + i];
      for (int i = 0; i < D::NU; i++)
        m.wud[i] = m.w_u[i * D::NU + i];
      std::copy(ks.w_frame.begin(), ks.w_frame.end(), m.w_frame);
      std::copy(ks.w_cent.begin(), ks.w_cent.end(), m.w_cent);
      std::copy(ks.w_centder.begin(), ks.w_centder.end(), m.w_centder);
      std::copy(ks.qmin.begin(), ks.qmin.end(), m.qmin);
      std::copy(ks.qmax.begin(), ks.qmax.end(), m.qmax);
      m.kinematics_limits = ks.kinematics_limits;
      m.mu = ms.mu_init;
      x_model_ref.assign(D::NX, 0.0);
      for (int i = 0; i < D::NQ; i++)
        x_model_ref[i] = rm->q_ref[i];
      x_reference = x_model_ref;
      for (int i = 0; i < D::NX; i++)
        m.x_term[i] = x_model_ref[i];
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
      buf.gains = dalloc(BH * (size_t)std::max((int)D::G_STRIDE, (int)GainsK<D>::STRIDE));
      if (lane_eval && m.lane_slots > 0 && !ks.terminal_constraint && !ks.force_cone && !ks.land_cstr)
      {
        // hand-over of the lane-per-problem evaluation.  With the stream (derivative pass: per-problem contiguous run, 688 doubles per
        // problem on Go2) the tiles hold the line-search heads only; without it (SMPC_LANE_STREAM=0, or no memory for it) every field.
        // A failed allocation falls back one level -- stream -> tiles -> the one-kernel evaluation -- instead of failing the handle.
        // (the flush addresses a problem's block by a 32-bit offset in doubles: 34 GB of stream -- B = 134 000 at H = 50)
        const size_t tiles = (((size_t)B + EV_LS - 1) / EV_LS) * (H + 1);
        auto try_alloc = [&](size_t doubles) -> double * {
          try
          {
            return dalloc(doubles);
          }
          catch (const std::exception &)
          {
            dev_clear_error();
            return nullptr;
          }
        };
        if (lane_deriv && lane_stream && (size_t)B * (H + 1) * EvStream<D>::STRIDE < ((size_t)1 << 32))
          buf.evd = try_alloc((size_t)B * (H + 1) * EvStream<D>::STRIDE);
        buf.ev_tile = ev_tile_doubles<D>(buf.evd != nullptr);
        buf.ev = try_alloc(tiles * buf.ev_tile);
        if (buf.ev == nullptr && buf.evd != nullptr)
        {
          // stream allocated, heads-only tiles not: give the stream back and try the full-field tiles (the second level of the fall-back:
          // stream -> tiles -> one-kernel stage evaluation)
          dev_free(buf.evd);
          buf.evd = nullptr;
          buf.ev_tile = ev_tile_doubles<D>(false);
          buf.ev = try_alloc(tiles * buf.ev_tile);
        }
        if (buf.evd != nullptr)
          buf.ev_order = (int *)dev_alloc((size_t)EvStream<D>::STRIDE * sizeof(int));
        handover_bytes = (buf.ev ? tiles * buf.ev_tile : 0) * sizeof(double) + (buf.evd ? (size_t)B * (H + 1) * EvStream<D>::STRIDE * sizeof(double) : 0);
      }
      buf.QN = dalloc((size_t)B * D::NDX * D::NDX);
      buf.qN = dalloc((size_t)B * D::NDX);
      buf.parts0 = dalloc((size_t)B * (H + 1) * 4);
      buf.partsT = dalloc((size_t)B * D::LS_N * (H + 1) * 2);
      buf.scal = dalloc((size_t)B * SC_N);
      buf.xdotT = dalloc((size_t)B * D::LS_N * 4 * D::NV);
      buf.xdot01 = dalloc((size_t)B * 4 * D::NV);
      buf.ls_sel = (int *)dev_alloc((size_t)B * sizeof(int));
      buf.und_list = (int *)dev_alloc((size_t)(B + 1) * sizeof(int));
      und_lists[0] = buf.und_list;
      for (int i = 1; i < n_streams; i++)
        und_lists[i] = (int *)dev_alloc((size_t)(B + 1) * sizeof(int));
      buf.stages = (StageShared<D> *)dev_alloc((size_t)H * sizeof(StageShared<D>));
      buf.model = (DevModel<D> *)dev_alloc(sizeof(DevModel<D>));
      X_dev = dalloc((size_t)B * D::NX);
      if (ks.terminal_constraint)
        alloc_terminal_constraint<D>(buf, x_model_ref.data(), host_com_height(m, x_model_ref.data()), stream);
      if (ks.force_cone)
      {
        if (!structured_riccati)
          throw std::runtime_error("force_cone needs the structured Riccati sweep (unset SMPC_RICCATI)");
        auto zalloc = [&](size_t n) {
          double * p = dalloc(n);
          dev_zero(p, n * sizeof(double), stream);
          return p;
        };
        buf.es = zalloc(BR * 2 * D::NF);
        buf.es_e = zalloc(BR * 2 * D::NF);
        buf.es_b = zalloc(BR * 2 * D::NF);
        buf.des = zalloc(BH * 2 * D::NF);
        buf.ek = zalloc(BH * 12 * D::NF);
        buf.cone_mu2 = ks.mu * ks.mu;
      }
      land_cstr = ks.land_cstr != 0;
      if (land_cstr)
      {
        if (!structured_riccati)
          throw std::runtime_error("land_cstr needs the structured Riccati sweep (unset SMPC_RICCATI)");
        auto zalloc = [&](size_t n) {
          double * p = dalloc(n);
          dev_zero(p, n * sizeof(double), stream);
          return p;
        };
        buf.ls = zalloc(BR * D::NF);
        buf.ls_e = zalloc(BR * D::NF);
        buf.ls_b = zalloc(BR * D::NF);
        buf.dls = zalloc(BH * D::NF);
        buf.lk = zalloc(BH * D::NF * (D::NV + 2));
      }
      if (std::getenv("SMPC_PHASE_PROFILE"))
        buf.dbg = dalloc(64);
      h2d(buf.model, hm.data(), sizeof(DevModel<D>), stream);
      stream_sync(stream);

      // ---- default problem (OCPHandler::createProblem, src/ocp-handler.cpp:96-137) ----
      StageShared<D> def;
      std::memset(&def, 0, sizeof(def));
      def.mask = (1u << D::NF) - 1u;
      for (int f = 0; f < D::NF; f++)
        def.u_ref[3 * f + 2] = -rm->total_mass * gravity_arg / (double)D::NF;
      for (int i = 0; i < D::NX; i++)
        def.x_tgt[i] = x_model_ref[i];
      horizon.assign(H, def);
      standing = def;
      {
        StageKernelArgs<D> sk;
        sk.b = buf;
        sk.head = 0;
        sk.j0 = sk.nj = sk.slots = 0;
        launch<StageKernelArgs<D>, lq_init_body<D>, 64>(B * H, stream, sk);
      }
      cold_solve(def);
      for (int f = 0; f < D::NF; f++)
        buf.land_z[f] = ref_foot_pos[f][2]; // contact poses of the cycle stages: the feet at the reference state (src/mpc.cpp:162)
      ref_rot.init(H, D::NF);
      ctor_scope.commit();
    }
    bool land_cstr = false;
    ~KinoEngine()
    {
      for (double * p : {buf.CN, buf.vN, buf.vN_e, buf.vN_b, buf.dvN, buf.dcm_ref, buf.es, buf.es_e, buf.es_b, buf.des, buf.ek, buf.ls, buf.ls_e, buf.ls_b, buf.dls, buf.lk})
        dev_free(p);
      for (double * p : {buf.xs_b, buf.us_b, buf.vs_b, buf.lams_b, buf.xs, buf.us, buf.vs, buf.lams, buf.vs_e, buf.lams_e, buf.dxs, buf.dus, buf.dvs, buf.dlams, buf.foot_ref, buf.ftraj, buf.vbase, buf.vref, buf.lq,
                         buf.gains, buf.ev, buf.QN, buf.qN, buf.parts0, buf.partsT, buf.scal, buf.xdotT, buf.xdot01, X_dev, stage_out})
        dev_free(p);
      dev_free(buf.ls_sel);
      dev_free(buf.und_list);
      dev_free(buf.evd);
      dev_free(buf.ev_order);
      if (ev_handoff_valid)
        event_destroy(ev_handoff);
      dev_free(sim_a);
      dev_free(sim_lam);
      dev_free(sim_mask);
      for (int i = 1; i < n_streams; i++)
      {
        dev_free(und_lists[i]);
        stream_destroy(streams[i]);
        event_destroy(ev_join[i]);
      }
      if (n_streams > 1)
        event_destroy(ev_fork);
      dev_free(buf.stages);
      dev_free(buf.model);
      stream_destroy(stream);
    }
    KinoEngine(const KinoEngine &) = delete;
    KinoEngine & operator=(const KinoEngine &) = delete;

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

    // aux: auxiliary launch (cold start on one instance, list-mode launch of the backtracking path): same code under a
    // second kernel symbol, so that profiler averages of the main symbol are those of full-batch launches
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
    void collect_profile()
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
      return sk;
    }
    // optional constraint blocks present: the kernels' EXT instantiations (the default ones carry none of that code)
    static bool has_ext(const Buffers<D> & b) { return b.es != nullptr || b.ls != nullptr || b.CN != nullptr; }
    void launch_deriv(const Buffers<D> & b, int slots = 0)
    {
      // list-mode launches (backtracking path, normally empty) are booked under "select" so that the per-kernel
      // averages of deriv / trial / apply stay those of full-batch launches
      if (b.ev != nullptr && lane_deriv)
      {
        // lane-per-problem evaluation, then the wavefront-per-problem derivative kernel that starts from it
        LaneKernelArgs<D> la;
        la.b = b;
        la.head = head;
        la.j0 = la.nj = 0;
        la.slots = slots;
        la.deriv = 1;
        la.order = nullptr;
        const int n = slots > 0 ? slots : b.B, kid = slots > 0 ? KID_SELECT : KID_DERIV, kid_tree = slots > 0 ? KID_SELECT : KID_TREE;
        if (b.evd != nullptr && !stream_order_recorded)
        {
          // once per handle: the tree kernel itself records the order in which it produces a problem's fields (one wavefront: the order
          // does not depend on the problem); the derivative kernel reads the fields back by that table
          LaneKernelArgs<D> lo = la;
          lo.b.B = b.B < 64 ? b.B : 64;
          lo.slots = 0;
          lo.order = b.ev_order;
          const bool prof = profiling;
          profiling = false;
          if (lane_slots == 1)
            timed_launch<LaneKernelArgs<D>, lane_tree_body<D, 1, true, true>, 64, SMPC_LANE_MINW>(KID_SELECT, 1, lo, true);
          else
            timed_launch<LaneKernelArgs<D>, lane_tree_body<D, 2, true, true>, 64, SMPC_LANE_MINW>(KID_SELECT, 1, lo, true);
          profiling = prof;
          // field ids -> commit codes of the derivative kernel (where each element of the stream goes in its scratch)
          std::vector<int> ord(EvStream<D>::STRIDE);
          d2h(ord.data(), b.ev_order, ord.size() * sizeof(int), cur);
          stream_sync(cur);
          for (int & v : ord)
            v = deriv2_commit_code<D>(foot_joint_h, v);
          h2d(b.ev_order, ord.data(), ord.size() * sizeof(int), cur);
          stream_sync(cur); // (ord is a local)
          stream_order_recorded = true;
        }
        const int gtree = (H + 1) * ((n + 63) / 64);
        if (b.evd != nullptr)
        {
          if (lane_slots == 1)
            timed_launch<LaneKernelArgs<D>, lane_tree_body<D, 1, true>, 64, SMPC_LANE_MINW>(kid_tree, gtree, la, slots > 0);
          else
            timed_launch<LaneKernelArgs<D>, lane_tree_body<D, 2, true>, 64, SMPC_LANE_MINW>(kid_tree, gtree, la, slots > 0);
        }
        else if (lane_slots == 1)
          timed_launch<LaneKernelArgs<D>, lane_tree_body<D, 1>, 64, SMPC_LANE_MINW>(kid_tree, gtree, la, slots > 0);
        else
          timed_launch<LaneKernelArgs<D>, lane_tree_body<D, 2>, 64, SMPC_LANE_MINW>(kid_tree, gtree, la, slots > 0);
        if (b.evd != nullptr)
          timed_launch<StageKernelArgs<D>, deriv2_body<D, true>, 64, SMPC_DERIV2_MINW>(kid, xcd_grid(n, H), stage_args(b, slots), slots > 0);
        else
          timed_launch<StageKernelArgs<D>, deriv2_body<D, false>, 64, SMPC_DERIV2_MINW>(kid, xcd_grid(n, H), stage_args(b, slots), slots > 0);
      }
      else if (has_ext(b) || !kCrossCheck) // (without the lane hand-over -- allocation refused -- the shipped library runs the instantiation it has)
        timed_launch<StageKernelArgs<D>, deriv_body<D, true>, 64, 2>(slots > 0 ? KID_SELECT : KID_DERIV, (slots > 0 ? slots : b.B) * (H + 1), stage_args(b, slots), slots > 0);
      else if constexpr (kCrossCheck)
        timed_launch<StageKernelArgs<D>, deriv_body<D, false>, 64, 2>(slots > 0 ? KID_SELECT : KID_DERIV, (slots > 0 ? slots : b.B) * (H + 1), stage_args(b, slots), slots > 0);
    }
    // line-search evaluation of the candidates sk.j0 .. sk.j0 + sk.nj - 1 (sk.slots > 0: for the compacted list of undecided instances)
    void launch_trial(const Buffers<D> & b, const StageKernelArgs<D> & sk, int kid, bool aux)
    {
      if (b.ev != nullptr)
      {
        LaneKernelArgs<D> la;
        la.b = b;
        la.head = sk.head;
        la.j0 = sk.j0;
        la.nj = sk.nj;
        la.slots = sk.slots;
        la.deriv = 0;
        la.order = nullptr;
        const int n = sk.slots > 0 ? sk.slots : b.B, kid_tree = kid == KID_TRIAL ? KID_TREE_LS : kid;
        if (lane_slots == 1)
          timed_launch<LaneKernelArgs<D>, lane_tree_body<D, 1>, 64, SMPC_LANE_MINW>(kid_tree, (H + 1) * ((n + 63) / 64), la, aux);
        else
          timed_launch<LaneKernelArgs<D>, lane_tree_body<D, 2>, 64, SMPC_LANE_MINW>(kid_tree, (H + 1) * ((n + 63) / 64), la, aux);
        timed_launch<StageKernelArgs<D>, trial_rows_body<D>, 64>(kid, xcd_grid(n, H), sk, aux);
      }
      else if (has_ext(b) || !kCrossCheck)
        timed_launch<StageKernelArgs<D>, trial_body<D, true>, 64, TRIAL_MINW>(kid, (sk.slots > 0 ? sk.slots : b.B) * (H + 1), sk, aux);
      else if constexpr (kCrossCheck)
        timed_launch<StageKernelArgs<D>, trial_body<D, false>, 64, TRIAL_MINW>(kid, (sk.slots > 0 ? sk.slots : b.B) * (H + 1), sk, aux);
    }
    // backward + forward sweep: Newton step and merit directional derivative
    void launch_sweeps(const Buffers<D> & b)
    {
      if (structured_riccati)
      {
        // kinodynamics-structured sweep, one wavefront per instance
        if (has_ext(b))
        {
          timed_launch<SolverArgs<D>, riccati_kino_body<D, true>, 64, RICCATI_MINW>(KID_RICCATI, b.B, solver_args(b));
          timed_launch<SolverArgs<D>, forward_kino_body<D, true>, 64>(KID_FORWARD, b.B, solver_args(b));
        }
        else
        {
          timed_launch<SolverArgs<D>, riccati_kino_body<D, false>, 64, RICCATI_MINW>(KID_RICCATI, b.B, solver_args(b));
          timed_launch<SolverArgs<D>, forward_kino_body<D, false>, 64>(KID_FORWARD, b.B, solver_args(b));
        }
      }
      else if constexpr (kCrossCheck)
      {
        // dense, model-independent sweep (cross-check of the structured one: SMPC_RICCATI=dense)
        switch (riccati_nt)
        {
        case 64:
          timed_launch<SolverArgs<D>, riccati_body<D, 64>, 64>(KID_RICCATI, b.B, solver_args(b));
          break;
        case 128:
          timed_launch<SolverArgs<D>, riccati_body<D, 128>, 128>(KID_RICCATI, b.B, solver_args(b));
          break;
        default:
          timed_launch<SolverArgs<D>, riccati_body<D, 256>, 256>(KID_RICCATI, b.B, solver_args(b));
        }
        timed_launch<SolverArgs<D>, forward_body<D>, 64>(KID_FORWARD, b.B, solver_args(b));
      }
      if (b.CN != nullptr)
        timed_launch<SolverArgs<D>, term_step_body<D>, 64>(KID_FORWARD, (b.B + 63) / 64, solver_args(b));
    }
    // backtracking candidates 2^-1 .. 2^-9 for the instances that are still undecided (compacted list)
    int launch_backtracking(const Buffers<D> & b)
    {
      const int slots = b.B < LS_SLOTS ? b.B : LS_SLOTS;
      StageKernelArgs<D> sk = stage_args(b, slots);
      sk.j0 = 1;
      sk.nj = D::LS_N - 1;
      timed_launch<SolverArgs<D>, compact_body<D>, 64>(KID_SELECT, 1, solver_args(b));
      return slots;
    }
    // line search with explicit trial evaluations: alpha = 1 for everybody, then the rest for the undecided
    void launch_line_search(const Buffers<D> & b)
    {
      StageKernelArgs<D> sk = stage_args(b);
      sk.j0 = 0;
      sk.nj = 1;
      launch_trial(b, sk, KID_TRIAL, false);
      timed_launch<SolverArgs<D>, select_body<D>, 64>(KID_SELECT, (b.B + 63) / 64, solver_args(b, 0, 1));
      const int slots = launch_backtracking(b);
      sk.slots = slots;
      sk.j0 = 1;
      sk.nj = D::LS_N - 1;
      launch_trial(b, sk, KID_SELECT, true);
      timed_launch<SolverArgs<D>, select_body<D>, 64>(KID_SELECT, (b.B + 63) / 64, solver_args(b, 1, D::LS_N - 1));
      timed_launch<SolverArgs<D>, apply_body<D>, 64>(KID_APPLY, b.B, solver_args(b));
    }
    // one ProxDDP iteration for the instances covered by b (b.B may be < B for the cold solve)
    void run_iteration(const Buffers<D> & b)
    {
      launch_deriv(b);
      launch_sweeps(b);
      launch_line_search(b);
    }
    // k ProxDDP iterations of one control step.  Iterations before the last take the full step TENTATIVELY and run the
    // next derivative pass at once: its merit IS the line-search value phi(1), so in the common case (Armijo accepts
    // alpha = 1) no separate trial evaluation is launched, and the result is the sequential algorithm's.  Instances
    // that reject alpha = 1 are restored, backtracked with explicit trial evaluations and re-derived (compacted list).
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
        speculative_step(b, it == k - 1);
    }
    // one iteration of the speculative scheme: sweeps, then either the explicit line search (last iteration) or the tentative
    // full step + next derivative pass + repair of the instances that rejected it
    void speculative_step(const Buffers<D> & b, bool last)
    {
      const int nb = (b.B + 63) / 64;
      {
        launch_sweeps(b);
        if (last)
        {
          launch_line_search(b);
          return;
        }
        SolverArgs<D> sa = solver_args(b);
        sa.mode = 1;
        timed_launch<SolverArgs<D>, apply_body<D>, 64>(KID_APPLY, b.B, sa);
        launch_deriv(b);
        timed_launch<SolverArgs<D>, spec_select_body<D>, 64>(KID_SELECT, nb, solver_args(b));
        // rejected instances (usually none: every launch below then exits at once)
        const int slots = launch_backtracking(b);
        sa = solver_args(b);
        sa.slots = slots;
        sa.mode = 2;
        timed_launch<SolverArgs<D>, apply_body<D>, 64>(KID_SELECT, slots, sa, true);
        StageKernelArgs<D> sk = stage_args(b, slots);
        sk.j0 = 1;
        sk.nj = D::LS_N - 1;
        launch_trial(b, sk, KID_SELECT, true);
        timed_launch<SolverArgs<D>, select_body<D>, 64>(KID_SELECT, nb, solver_args(b, 1, D::LS_N - 1));
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
      if (b.es != nullptr)
        d2d(b.es_e, b.es, (size_t)b.B * R * 2 * D::NF * sizeof(double), cur);
      if (b.ls != nullptr)
        d2d(b.ls_e, b.ls, (size_t)b.B * R * D::NF * sizeof(double), cur);
    }

    void upload_stages()
    {
      stage_ring.upload(buf.stages, horizon.data(), (size_t)H * sizeof(StageShared<D>), stream);
    }
    UploadRing stage_ring;

    // reference: src/mpc.cpp:72-91.  All instances share x0 = reference state: solve instance 0, broadcast.
    void cold_solve(const StageShared<D> & def)
    {
      std::vector<double> xs0((size_t)R * D::NX), us0((size_t)R * D::NU);
      for (int t = 0; t < R; t++)
      {
        std::copy(x_model_ref.begin(), x_model_ref.end(), xs0.begin() + (size_t)t * D::NX);
        std::copy(def.u_ref, def.u_ref + D::NU, us0.begin() + (size_t)t * D::NU);
      }
      head = 0;
      h2d(buf.xs, xs0.data(), xs0.size() * sizeof(double), stream);
      h2d(buf.us, us0.data(), us0.size() * sizeof(double), stream);
      std::vector<double> sc0(SC_N, 0.0);
      sc0[SC_PREG] = REG_INIT;
      h2d(buf.scal, sc0.data(), SC_N * sizeof(double), stream);
      upload_stages();
      // foot refs of the default problem are the identity placements: translation 0 (src/ocp-handler.cpp:116)
      dev_zero(buf.foot_ref, (size_t)H * D::NF * 3 * sizeof(double), stream);
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
        // stalled: predicted merit decrease below FP64 resolution (DESIGN.md "solver constants")
        if (std::fabs(sc[SC_DPHI0]) <= STALL_REL * std::fmax(1.0, std::fabs(sc[SC_PHI0])))
          break;
        if (sc[SC_DUAL] <= ms.TOL)
          copy_centres(b1);
      }
      aux_launches = false;
      // broadcast instance 0 to the whole batch
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
      if (buf.CN != nullptr)
      {
        bc(buf.vN, 3);
        bc(buf.dcm_ref, 3);
      }
      if (buf.es != nullptr)
        bc(buf.es, (size_t)R * 2 * D::NF);
      if (buf.ls != nullptr)
        bc(buf.ls, (size_t)R * D::NF);
      // swing start/end = reference foot positions (FootTrajectory ctor, src/foot-trajectory.cpp:20-39):
      // a reference-only recede call with land = -1 < T_fly keeps them, so initialise them here on the host
      std::vector<double> ft((size_t)D::NF * 6);
      host_foot_positions(x_model_ref.data(), ft.data());
      h2d(buf.ftraj, ft.data(), ft.size() * sizeof(double), stream);
      stream_sync(stream);
      bc(buf.ftraj, (size_t)D::NF * 6);
      stream_sync(stream);
      for (int f = 0; f < D::NF; f++)
        for (int i = 0; i < 3; i++)
          ref_foot_pos[f][i] = ft[f * 6 + i];
    }
    double ref_foot_pos[D::NF][3];

    // foot positions at the model reference state, [NF][6] = (start, end) both at the foot position.
    // Host restatement of FK limited to what the constructor needs (src/mpc.cpp:24-39).
    void host_foot_positions(const double * x, double * out)
    {
      std::vector<DevModel<D>> hm(1);
      d2h(hm.data(), buf.model, sizeof(DevModel<D>), stream);
      stream_sync(stream);
      const DevModel<D> & m = hm[0];
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

    void generate_cycle_horizon(const unsigned char * cs, int n)
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
            s.u_ref[3 * f + 2] = ms.support_force / (double)active;
          }
        s.land = s.mask & ~previous;
        previous = s.mask;
        for (int i = 0; i < D::NX; i++)
          s.x_tgt[i] = x_model_ref[i];
        cycle.push_back(s);
      }
    }
    // velocity commands live on the device, one per instance; the reference's single velocity_base_ is a broadcast
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
    void switch_to_walk(const double * v6)
    {
      walking = true;
      for (int i = 0; i < 6; i++)
        velocity_base[i] = v6[i];
      upload_velocity(v6, true);
    }
    void switch_to_stand()
    {
      walking = false;
      for (int i = 0; i < 6; i++)
        velocity_base[i] = 0.0;
      upload_velocity(velocity_base, true);
    }
    // one velocity command per instance, V: [B][6] (the walking state is unchanged, like assigning MPC::velocity_base_)
    void set_velocity_base_batched(const double * V)
    {
      for (int i = 0; i < 6; i++)
        velocity_base[i] = V[i];
      upload_velocity(V, false);
    }

    // One control step for the whole batch; Xd: device pointer [B][NX]
    void iterate_device(const double * Xd)
    {
      ref_rot.reset(); // (every control step rewrites every stage's reference pose with the identity rotation: src/mpc.cpp:303-309)
      if (cycle.empty())
        throw std::runtime_error("generateCycleHorizon must be called before iterate");
      // ---- recedeWithCycle (host, shared by the batch) ----
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
      // setReferenceState(H-1, x_reference_) ; setVelocityBase(H-1, velocity_base_)  (src/mpc.cpp:311-312)
      for (int i = 0; i < D::NX; i++)
        horizon[H - 1].x_tgt[i] = x_reference[i];
      for (int i = 0; i < 6; i++)
        horizon[H - 1].x_tgt[D::NQ + i] = velocity_base[i];
      upload_stages();
      head = head + 1 == R ? 0 : head + 1; // replaceStageCircular + cycleProblem as a ring advance
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
      if (n_streams > 1 && B >= 64 * n_streams && !has_ext(buf))
      {
        // the parts of the batch run their iterations on separate streams: the tail of one part's launch is filled by the next
        // launch of another part (every launch alone is a whole number of rounds of resident waves plus a partial one)
        Buffers<D> part[MAX_STREAMS];
        event_record(ev_fork, stream);
        for (int i = 0; i < n_streams; i++)
        {
          const int i0 = (int)((long long)B * i / n_streams), i1 = (int)((long long)B * (i + 1) / n_streams);
          part[i] = slice(buf, i0, i1 - i0, und_lists[i]);
          cur = streams[i];
          if (i > 0)
            stream_wait_event(streams[i], ev_fork);
          copy_centres(part[i]);
        }
        run_iterations_parts(part, ms.max_iters);
        for (int i = 1; i < n_streams; i++)
        {
          event_record(ev_join[i], streams[i]);
          stream_wait_event(stream, ev_join[i]);
        }
        cur = stream;
        return;
      }
      copy_centres(buf);
      run_iterations(buf, ms.max_iters);
    }
    // instances i0 .. i0 + n of every per-instance array
    Buffers<D> slice(const Buffers<D> & b, int i0, int n, int * und) const
    {
      Buffers<D> s = b;
      s.B = n;
      const size_t o = (size_t)i0, BRs = (size_t)R, Hs = (size_t)H;
      auto adv = [&](double *& p, size_t per) {
        if (p)
          p += o * per;
      };
      adv(s.xs, BRs * D::NX); adv(s.us, BRs * D::NU); adv(s.vs, BRs * D::NC); adv(s.lams, BRs * D::NDX);
      adv(s.vs_e, BRs * D::NC); adv(s.lams_e, BRs * D::NDX);
      adv(s.xs_b, BRs * D::NX); adv(s.us_b, BRs * D::NU); adv(s.vs_b, BRs * D::NC); adv(s.lams_b, BRs * D::NDX);
      adv(s.dxs, (Hs + 1) * D::NDX); adv(s.dus, Hs * D::NU); adv(s.dvs, Hs * D::NC); adv(s.dlams, Hs * D::NDX);
      adv(s.foot_ref, Hs * D::NF * 3); adv(s.ftraj, (size_t)D::NF * 6); adv(s.vbase, 6); adv(s.vref, BRs * 6);
      s.ev_inst0 = b.ev_inst0 + i0;
      adv(s.lq, Hs * D::LQ_STRIDE); adv(s.gains, Hs * (size_t)std::max((int)D::G_STRIDE, (int)GainsK<D>::STRIDE));
      adv(s.QN, (size_t)D::NDX * D::NDX); adv(s.qN, D::NDX);
      adv(s.parts0, (Hs + 1) * 4); adv(s.partsT, (size_t)D::LS_N * (Hs + 1) * 2); adv(s.scal, SC_N);
      adv(s.xdotT, (size_t)D::LS_N * 4 * D::NV); adv(s.xdot01, (size_t)4 * D::NV);
      s.ls_sel = b.ls_sel + i0;
      s.und_list = und;
      return s;
    }
    // run_iterations of the parts of the batch, their launches issued alternately so that every queue stays filled
    void run_iterations_parts(const Buffers<D> * part, int k)
    {
      auto on = [&](int i) -> const Buffers<D> & { cur = streams[i]; return part[i]; };
      if (!speculative_ls || k <= 1 || early_exit_on_tol) // (the convergence test belongs to the sequential scheme)
      {
        for (int it = 0; it < k; it++)
          for (int i = 0; i < n_streams; i++)
            run_iteration(on(i));
        return;
      }
      for (int i = 0; i < n_streams; i++)
      {
        const Buffers<D> & b = on(i);
        launch_deriv(b);
        timed_launch<SolverArgs<D>, merit0_body<D>, 64>(KID_SELECT, (b.B + 63) / 64, solver_args(b));
      }
      for (int it = 0; it < k; it++)
        for (int i = 0; i < n_streams; i++)
          speculative_step(on(i), it == k - 1);
    }
    // ---- per-stage references of the horizon: the OCPHandler setters / getters (reference src/kinodynamics.cpp:154-306,
    //      src/ocp-handler.cpp:58-81), broadcast over the batch.  The next iterate() overwrites the foot references of
    //      every stage and the state target of stage H-1, exactly like MPC::updateStepTrackerReferences does. ----
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
    // what: 0 = control target (nu), 1 = state target (nx)
    void set_stage_reference(int t, int what, const double * v, int n)
    {
      check_stage(t);
      if (what == 0)
      {
        if (n != D::NU)
          throw std::runtime_error("u_ref not of the right size");
        std::copy(v, v + n, horizon[t].u_ref);
      }
      else if (what == 1)
      {
        if (n != D::NX)
          throw std::runtime_error("x_ref not of the right size");
        std::copy(v, v + n, horizon[t].x_tgt);
        fill_strided(buf.vref + (size_t)ring_slot(head, t, R) * 6, (size_t)R * 6, B, v + D::NQ, 6); // velocity part is per instance
      }
      else
        throw std::runtime_error("unknown stage reference");
    }
    void get_stage_reference(int t, int what, double * v, int n)
    {
      check_stage(t);
      if (what == 0 && n == D::NU)
        std::copy(horizon[t].u_ref, horizon[t].u_ref + n, v);
      else if (what == 1 && n == D::NX)
      {
        std::copy(horizon[t].x_tgt, horizon[t].x_tgt + n, v);
        get_linear(buf.vref + (size_t)ring_slot(head, t, R) * 6, 6, v + D::NQ); // instance 0
      }
      else
        throw std::runtime_error("unknown stage reference or wrong size");
    }
    void set_reference_pose(int t, int foot, const double * p3)
    {
      check_stage(t);
      if (foot < 0 || foot >= D::NF)
        throw std::runtime_error("unknown end effector");
      ref_rot.set(t, foot, nullptr); // (a translation: identity rotation)
      fill_strided(buf.foot_ref + ((size_t)t * D::NF + foot) * 3, (size_t)H * D::NF * 3, B, p3, 3);
    }
    void get_reference_pose(int t, int foot, int inst, double * p3)
    {
      check_stage(t);
      if (foot < 0 || foot >= D::NF || inst < 0 || inst >= B)
        throw std::runtime_error("unknown end effector or instance");
      get_linear(buf.foot_ref + (((size_t)inst * H + t) * D::NF + foot) * 3, 3, p3);
    }
    RefRotations ref_rot; // rotations of the foot reference placements: API state (smpc_model.h)
    void set_reference_rotation(int t, int foot, const double * R9)
    {
      check_stage(t);
      if (foot < 0 || foot >= D::NF)
        throw std::runtime_error("unknown end effector");
      ref_rot.set(t, foot, R9);
    }
    void get_reference_rotation(int t, int foot, double * R9)
    {
      check_stage(t);
      if (foot < 0 || foot >= D::NF)
        throw std::runtime_error("unknown end effector");
      ref_rot.get(t, foot, R9);
    }
    unsigned contact_mask(int t) const
    {
      check_stage(t);
      return horizon[t].mask;
    }

    // Everything a later iterate() depends on: iterate, multipliers, swing trajectories, references, velocity commands, gait
    // bookkeeping.  Not included: the feedback gains and the LQ knots of the last solve (recomputed by the next iterate).
    size_t state_io(StateIO & io)
    {
      set_device(device_id);
      stream_sync(stream);
      io.tag(0x534d50434b494e4fLL, "kind (kinodynamics)");
      io.tag(B, "batch");
      io.tag(H, "horizon");
      io.tag(D::NX, "nx");
      io.tag(D::NU, "nu");
      io.tag(buf.CN != nullptr ? 1 : 0, "terminal constraint");
      io.tag(buf.es != nullptr ? 1 : 0, "friction-cone rows");
      io.tag(buf.ls != nullptr ? 1 : 0, "land rows");
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
      if (buf.CN != nullptr)
        io.dev(buf.vN, (size_t)B * 3 * sizeof(double));
      if (buf.es != nullptr)
        io.dev(buf.es, BR * 2 * D::NF * sizeof(double));
      if (buf.ls != nullptr)
        io.dev(buf.ls, BR * D::NF * sizeof(double));
      if (io.mode == StateIO::LOAD)
        upload_stages();
      stream_sync(stream);
      return io.pos;
    }
    void iterate_host(const double * X)
    {
      set_device(device_id);
      h2d(X_dev, X, (size_t)B * D::NX * sizeof(double), stream);
      iterate_device(X_dev);
      stream_sync(stream);
    }
    // the same without the final synchronisation: X must stay valid until sync() (one host thread can then keep several devices busy)
    void iterate_host_async(const double * X)
    {
      set_device(device_id);
      h2d(X_dev, X, (size_t)B * D::NX * sizeof(double), stream);
      iterate_device(X_dev);
    }
    void sync()
    {
      set_device(device_id);
      stream_sync(stream);
    }
    // What a controller consumes of a control step -- xs[1], us[0], K_0 -- of every instance as rows [x1 (NX) | u0 (NU) | K0 (NU x NDX)] of
    // `out`, `row_doubles` apart (the small return set of the sharded batch, gathered into ONE host buffer -- pinned by the
    // caller for full PCIe rate -- that several handles, one per device, fill side by side).  Asynchronous: sync() completes it.
    static constexpr int GATHER_ROW = D::NX + D::NU + D::NU * D::NDX;
    void gather_outputs_async(double * out, size_t row_doubles)
    {
      set_device(device_id);
      if (row_doubles < (size_t)GATHER_ROW)
        throw std::runtime_error("smpc_gather_outputs: the row stride is smaller than nx + nu + nu * ndx");
      const size_t dp = row_doubles * sizeof(double);
      const int s1 = ring_slot(head, 1, R), s0 = ring_slot(head, 0, R);
      d2h_2d(out, dp, buf.xs + (size_t)s1 * D::NX, (size_t)R * D::NX * sizeof(double), D::NX * sizeof(double), B, stream);
      d2h_2d(out + D::NX, dp, buf.us + (size_t)s0 * D::NU, (size_t)R * D::NU * sizeof(double), D::NU * sizeof(double), B, stream);
      const size_t n = (size_t)B * D::NU * D::NDX;
      double * dev = staging(n * sizeof(double));
      if (structured_riccati)
      {
        GainOutArgs<D> ga;
        ga.b = buf;
        ga.nt = 1;
        ga.out = dev;
        launch<GainOutArgs<D>, gains_out_body<D>, 64>(B, stream, ga);
        d2h_2d(out + D::NX + D::NU, dp, dev, (size_t)D::NU * D::NDX * sizeof(double), (size_t)D::NU * D::NDX * sizeof(double), B, stream);
      }
      else
        for (int i = 0; i < D::NU; i++) // dense sweep: rows of [K k] are NDX + 1 apart
          d2h_2d(out + D::NX + D::NU + (size_t)i * D::NDX, dp, buf.gains + D::G_K + (size_t)i * (D::NDX + 1), (size_t)H * D::G_STRIDE * sizeof(double),
                 D::NDX * sizeof(double), B, stream);
    }
    // the same rows packed into a DEVICE buffer (one kernel), for a caller that moves them itself: a collective towards the rank that
    // owns the controllers, or one copy into pinned memory from a side stream.  Asynchronous on the engine's stream.
    void gather_outputs_device(double * out_dev, size_t row_doubles)
    {
      set_device(device_id);
      if (row_doubles < (size_t)GATHER_ROW)
        throw std::runtime_error("smpc_gather_outputs_device: the row stride is smaller than nx + nu + nu * ndx");
      PackOutArgs<D> pa;
      pa.b = buf;
      pa.s1 = ring_slot(head, 1, R);
      pa.s0 = ring_slot(head, 0, R);
      pa.R = R;
      pa.g_off = structured_riccati ? GainsK<D>::G_W : D::G_K;
      pa.g_str = structured_riccati ? GainsK<D>::STRIDE : D::G_STRIDE;
      pa.g_tr = structured_riccati ? 1 : 0;
      pa.row = row_doubles;
      pa.out = out_dev;
      launch<PackOutArgs<D>, pack_outputs_body<D>, 64>(B, stream, pa);
    }
    // ... and into a buffer of ANOTHER device of the node: packed here, then one peer copy over xGMI on this engine's stream (the form
    // SURVEY 8e lists for a single process that drives all devices: every device's rows land in one buffer on the device -- or next to the
    // host thread -- that runs the controllers).  dst: [batch][GATHER_ROW] doubles, contiguous.
    void gather_outputs_peer(double * dst, int dst_device)
    {
      const size_t bytes = (size_t)B * GATHER_ROW * sizeof(double);
      double * dev = staging(bytes);
      gather_outputs_device(dev, GATHER_ROW);
      d2peer(dst, dst_device, dev, device_id, bytes, stream);
    }
    // xs[t] of every instance -> dense device buffer [B][NX], asynchronous on the engine's stream
    void gather_x_device(int t, double * out_dev)
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

    // u = u_interp - K_0 (x_interp (-) x_meas) at `delay` after the last solve, for measured states X [B][NX] (host)
    void riccati_feedback(double delay, const double * X, double * u_out)
    {
      if (!structured_riccati)
        throw std::runtime_error("riccati_feedback needs the structured Riccati sweep (unset SMPC_RICCATI)");
      if (!(delay >= 0.0))
        throw std::runtime_error("riccati_feedback: delay must be non-negative");
      const size_t nx = (size_t)B * D::NX, nu = (size_t)B * D::NU, nk = (size_t)B * D::NU * D::NDX;
      double * st = staging((nx + 2 * nu + nk) * sizeof(double));
      double *xi = st, *ui = st + nx, *uo = ui + nu, *k0 = uo + nu;
      h2d(X_dev, X, nx * sizeof(double), stream);
      InterpArgs<D> ia;
      ia.b = buf;
      ia.head = head;
      ia.knots = 2;
      ia.delay = delay;
      ia.timestep = ms.timestep;
      ia.x_out = xi;
      ia.acc_out = nullptr;
      ia.f_out = nullptr;
      ia.u_out = ui;
      launch<InterpArgs<D>, interp_body<D>, 64>(B, stream, ia);
      GainOutArgs<D> ga;
      ga.b = buf;
      ga.nt = 1;
      ga.out = k0;
      launch<GainOutArgs<D>, gains_out_body<D>, 64>(B, stream, ga);
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

    // state feedback front-end on measured states X [B][NX] (host): host outputs, any may be null
    void update_internal_data(const double * X, double * feet, double * com, double * hg, double * cstate)
    {
      const size_t nf = (size_t)B * D::NF * 3, nc = (size_t)B * 3, nh = (size_t)B * 6, ns = (size_t)B * 9;
      double * st = staging((nf + nc + nh + ns) * sizeof(double));
      h2d(X_dev, X, (size_t)B * D::NX * sizeof(double), stream);
      FrontendArgs<D> fa;
      fa.b = buf;
      fa.X = X_dev;
      fa.feet = feet ? st : nullptr;
      fa.com = com ? st + nf : nullptr;
      fa.hg = hg ? st + nf + nc : nullptr;
      fa.cstate = cstate ? st + nf + nc + nh : nullptr;
      launch<FrontendArgs<D>, frontend_body<D>, 64>(B, stream, fa);
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

    // constrained forward dynamics of the full-dynamics model for n states (host buffers; iters / kernel_ms may be null)
    void full_forward_dynamics(
      int n, const double * X, const double * tau, const unsigned * mask, const double * Kp, const double * Kd,
      double prox_accuracy, double prox_mu, int prox_max_iter, double * a, double * lam, int * iters, double * kernel_ms)
    {
      if (n < 1)
        throw std::runtime_error("full_forward_dynamics: n must be positive");
      constexpr int NV = D::NV, NX = D::NX, NCM = 3 * D::NF;
      // staging layout (doubles): X | tau | a | lam | mask (unsigned) | iters (int)
      const size_t oX = 0, oT = oX + (size_t)n * NX, oA = oT + (size_t)n * (NV - 6), oL = oA + (size_t)n * NV,
                   oM = oL + (size_t)n * NCM, oI = oM + ((size_t)n + 1) / 2, total = oI + ((size_t)n + 1) / 2;
      double * st = staging(total * sizeof(double));
      h2d(st + oX, X, (size_t)n * NX * sizeof(double), stream);
      h2d(st + oT, tau, (size_t)n * (NV - 6) * sizeof(double), stream);
      h2d(st + oM, mask, (size_t)n * sizeof(unsigned), stream);
      FullFdArgs<D> fa;
      fa.b = buf;
      fa.X = st + oX;
      fa.tau = st + oT;
      fa.mask = reinterpret_cast<const unsigned *>(st + oM);
      for (int i = 0; i < 3; i++)
      {
        fa.Kp[i] = Kp ? Kp[i] : 0.0;
        fa.Kd[i] = Kd ? Kd[i] : 0.0;
      }
      fa.prox_accuracy = prox_accuracy > 0 ? prox_accuracy : 1e-9; // ProximalSettings(1e-9, 1e-10, 10), src/fulldynamics.cpp:39
      fa.prox_mu = prox_mu > 0 ? prox_mu : 1e-10;
      fa.prox_max_iter = prox_max_iter > 0 ? prox_max_iter : 10;
      fa.a_out = st + oA;
      fa.lam_out = st + oL;
      fa.iters_out = reinterpret_cast<int *>(st + oI);
      stream_sync(stream);
      const auto t0 = std::chrono::steady_clock::now();
      launch<FullFdArgs<D>, full_fd_body<D>, 64, 2>(n, stream, fa); // 256 registers: 8 waves per CU with the 20.2 KB of LDS
      stream_sync(stream);
      if (kernel_ms)
        *kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      d2h(a, st + oA, (size_t)n * NV * sizeof(double), stream);
      d2h(lam, st + oL, (size_t)n * NCM * sizeof(double), stream);
      if (iters)
        d2h(iters, st + oI, (size_t)n * sizeof(int), stream);
      stream_sync(stream);
    }

    // One step of a simulated batch with states and torques resident in HBM (what the reference's examples do with a physics engine
    // between two controller ticks): constrained forward dynamics of the feet in `mask` (Baumgarte gains Kp, Kd; proximal settings of
    // record), then semi-implicit Euler over dt, X updated in place.  Asynchronous on this engine's stream.
    void sim_step_device(double * X_dev, const double * tau_dev, unsigned mask, const double * Kp, const double * Kd, double dt)
    {
      set_device(device_id);
      constexpr int NV = D::NV, NCM = 3 * D::NF;
      if (!sim_a)
      {
        sim_a = (double *)dev_alloc((size_t)B * NV * sizeof(double));
        sim_lam = (double *)dev_alloc((size_t)B * NCM * sizeof(double));
        sim_mask = (unsigned *)dev_alloc((size_t)B * sizeof(unsigned));
        sim_mask_value = ~0u;
      }
      if (mask != sim_mask_value)
      {
        std::vector<unsigned> m(B, mask);
        h2d(sim_mask, m.data(), m.size() * sizeof(unsigned), stream);
        stream_sync(stream); // (m goes out of scope)
        sim_mask_value = mask;
      }
      FullFdArgs<D> fa;
      fa.b = buf;
      fa.X = X_dev;
      fa.tau = tau_dev;
      fa.mask = sim_mask;
      for (int i = 0; i < 3; i++)
      {
        fa.Kp[i] = Kp ? Kp[i] : 0.0;
        fa.Kd[i] = Kd ? Kd[i] : 0.0;
      }
      fa.prox_accuracy = 1e-9; // ProximalSettings(1e-9, 1e-10, 10), src/fulldynamics.cpp:39
      fa.prox_mu = 1e-10;
      fa.prox_max_iter = 10;
      fa.a_out = sim_a;
      fa.lam_out = sim_lam;
      fa.iters_out = nullptr;
      launch<FullFdArgs<D>, full_fd_body<D>, 64, 2>(B, stream, fa);
      SimStepArgs<D> sa;
      sa.X = X_dev;
      sa.a = sim_a;
      sa.dt = dt;
      launch<SimStepArgs<D>, sim_integrate_body<D>, 64>(B, stream, sa);
    }
    // work issued on `other` from now on starts after what this engine's stream holds now
    void wait_stream(stream_t other)
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
    double *sim_a = nullptr, *sim_lam = nullptr;
    unsigned * sim_mask = nullptr;
    unsigned sim_mask_value = ~0u;
    // the same into device buffers (the inverse-dynamics engine's target buffers), asynchronous on this engine's stream
    void interpolate_device(double delay, int knots, double * x_dev, double * acc_dev, double * f_dev)
    {
      if (knots < 2 || knots > H + 1)
        throw std::runtime_error("interpolate: knots must be in [2, horizon + 1]");
      if (!(delay >= 0.0))
        throw std::runtime_error("interpolate: delay must be non-negative");
      set_device(device_id);
      InterpArgs<D> ia;
      ia.b = buf;
      ia.head = head;
      ia.knots = knots;
      ia.delay = delay;
      ia.timestep = ms.timestep;
      ia.x_out = x_dev;
      ia.acc_out = acc_dev;
      ia.f_out = f_dev;
      launch<InterpArgs<D>, interp_body<D>, 64>(B, stream, ia);
    }
    // interpolated whole-body targets at `delay` after the last solve; host outputs, any may be null
    void interpolate(double delay, int knots, double * x_out, double * acc_out, double * f_out)
    {
      if (knots < 2 || knots > H + 1)
        throw std::runtime_error("interpolate: knots must be in [2, horizon + 1]");
      if (!(delay >= 0.0))
        throw std::runtime_error("interpolate: delay must be non-negative");
      const size_t nx = (size_t)B * D::NX, na = (size_t)B * D::NV, nf = (size_t)B * 3 * D::NF;
      double * st = staging((nx + na + nf) * sizeof(double));
      InterpArgs<D> ia;
      ia.b = buf;
      ia.head = head;
      ia.knots = knots;
      ia.delay = delay;
      ia.timestep = ms.timestep;
      ia.x_out = x_out ? st : nullptr;
      ia.acc_out = acc_out ? st + nx : nullptr;
      ia.f_out = f_out ? st + nx + na : nullptr;
      launch<InterpArgs<D>, interp_body<D>, 64>(B, stream, ia);
      if (x_out)
        d2h(x_out, st, nx * sizeof(double), stream);
      if (acc_out)
        d2h(acc_out, st + nx, na * sizeof(double), stream);
      if (f_out)
        d2h(f_out, st + nx + na, nf * sizeof(double), stream);
      stream_sync(stream);
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
    // ring array [B][R][n] -> host linear [B][count][n] for t = 0..count-1
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
    // K_t of every stage [B][H][NU][NDX] (strided out of the gains block) or only K_0 [B][NU][NDX]
    void get_K(double * out, bool all)
    {
      stream_sync(stream);
      const int nt = all ? H : 1;
      if (structured_riccati)
      {
        // feedback is stored factored (W, L_R): K_t = -L_R^-T W_x, expanded on the device on request
        const size_t n = (size_t)B * nt * D::NU * D::NDX;
        double * dev = staging(n * sizeof(double));
        GainOutArgs<D> ga;
        ga.b = buf;
        ga.nt = nt;
        ga.out = dev;
        launch<GainOutArgs<D>, gains_out_body<D>, 64>(B * nt, stream, ga);
        d2h(out, dev, n * sizeof(double), stream);
        stream_sync(stream);
        return;
      }
      std::vector<double> row(D::NU * (D::NDX + 1));
      for (int b = 0; b < B; b++)
        for (int t = 0; t < nt; t++)
        {
          d2h(row.data(), buf.gains + ((size_t)b * H + t) * D::G_STRIDE + D::G_K, row.size() * sizeof(double), stream);
          stream_sync(stream);
          for (int i = 0; i < D::NU; i++)
            std::memcpy(out + (((size_t)b * nt + t) * D::NU + i) * D::NDX, row.data() + (size_t)i * (D::NDX + 1), D::NDX * sizeof(double));
        }
    }
  };
} // namespace smpc
