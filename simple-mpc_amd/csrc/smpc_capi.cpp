// smpc_capi.cpp -- extern "C" entry points of include/smpc.h over KinoEngine.  Compiled by hipcc for
// gfx950 (kernels are instantiated here); there is no CPU implementation behind these symbols.
#include "../../include/smpc.h"
#include "../../include/smpc_robots_builtin.h"
#include "smpc_cent_engine.h"
#include "smpc_full_engine.h"
#include "smpc_id.h"
#include <cstring>
#include <memory>
#include <string>

using namespace smpc;

typedef Dims<13, 4> DimsGo2; // free-flyer + 12 revolute joints, 4 point feet

typedef CentDims<4> CentGo2;
typedef CentEngine<DimsGo2, CentGo2> CentEngineGo2;
typedef CentDims<2, 6> CentTalos; // centroidal OCP of the Talos-class biped: 6-D feet, wrench cones (CentroidalOCP with force_size 6)
typedef FullDims<13, 4, 3> FullGo2;   // full dynamics: 12 joint torques, 3-D contacts
typedef FullDims<13, 4, 3, 5> FullGo2Cone; // the same with force_cone: 5 friction-pyramid rows per foot in contact
typedef FullDims<13, 4, 3, 0, 4> FullGo2Land; // land_cstr: 4 rows per landing foot
typedef FullDims<13, 4, 3, 5, 4> FullGo2ConeLand; // force_cone and land_cstr: the land rows behind the pyramid rows
typedef FullDims<23, 2, 6> FullTalos; // Talos-class humanoid: 22 joint torques, two 6-D feet with wrench cones
typedef FullDims<23, 2, 6, 0, 6> FullTalosLand; // land_cstr: 6 frame-velocity rows per landing foot
typedef FullDims<23, 2, 6, 0, 0, 1> KinoTalos;  // KINODYNAMICS OCP of the Talos-class biped: 6-D feet, wrench cones (KinodynamicsOCP with force_size 6)

struct smpc_handle
{
  std::unique_ptr<KinoEngine<DimsGo2>> eng;
  std::unique_ptr<CentEngineBase> cent; // centroidal handle (smpc_create_centroidal): eng is null
  std::unique_ptr<FullEngineBase> full; // full-dynamics handle (smpc_create_fulldynamics): eng and cent are null
};

namespace
{
  thread_local std::string g_err;
  int fail(int code, const std::string & msg)
  {
    g_err = msg;
    return code;
  }
  template <class F>
  int guarded(F && f)
  {
    try
    {
      f();
      return SMPC_OK;
    }
    catch (const std::exception & e)
    {
      return fail(SMPC_ERR_RUNTIME, e.what());
    }
  }
  HostMpcSettings host_mpc(const smpc_mpc_settings * mpc)
  {
    HostMpcSettings ms;
    ms.swing_apex = mpc->swing_apex;
    ms.support_force = mpc->support_force;
    ms.TOL = mpc->TOL;
    ms.mu_init = mpc->mu_init;
    ms.timestep = mpc->timestep;
    ms.max_iters = mpc->max_iters;
    ms.num_threads = mpc->num_threads;
    ms.T_fly = mpc->T_fly;
    ms.T_contact = mpc->T_contact;
    ms.T = mpc->T;
    return ms;
  }
  const char * KINO_ONLY = "this entry point needs a kinodynamics handle (smpc_create)";
} // namespace

extern "C"
{
  const smpc_robot_model * smpc_builtin_robot(const char * name)
  {
    if (!name)
      return nullptr;
    if (!std::strcmp(name, "go2_like"))
      return &SMPC_ROBOT_GO2_LIKE;
    if (!std::strcmp(name, "biped_like"))
      return &SMPC_ROBOT_BIPED_LIKE;
    if (!std::strcmp(name, "talos_like"))
      return &SMPC_ROBOT_TALOS_LIKE;
    return nullptr;
  }
  const char * smpc_last_error(void) { return g_err.c_str(); }
  int smpc_device_count(void) { return device_count(); }

  int smpc_create(
    const smpc_robot_model * robot, const smpc_kinodynamics_settings * ocp, const smpc_mpc_settings * mpc, int batch,
    double gravity_arg, int device_id, smpc_handle ** out)
  {
    if (!robot || !ocp || !mpc || !out)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (device_count() <= 0)
      return fail(SMPC_ERR_NO_DEVICE, "no HIP device visible: the MPC engine has no CPU path");
    if (ocp->force_size != 3 && ocp->force_size != 6)
      return fail(SMPC_ERR_INVALID, "force size in settings does not match reference force size");
    if (mpc->T < 2)
      return fail(SMPC_ERR_INVALID, "horizon must have at least 2 stages");
    if (ocp->force_size == 6)
    {
      // 6-D (flat) feet: the kinodynamics variant of the dense stage / solver kernels (FullDims<..., KIN = 1>, smpc_full_model.h)
      if (robot->njoints != KinoTalos::NJ || robot->nfeet != KinoTalos::NF)
        return fail(SMPC_ERR_INVALID, "robot shape (njoints, nfeet, force_size) does not match a built kernel instantiation");
      const int nv = robot->nv, ndx = 2 * nv, na = nv - 6, nu = na + 6 * robot->nfeet;
      HostFullSettings s;
      s.timestep = ocp->timestep;
      s.w_x.assign(ocp->w_x, ocp->w_x + (size_t)ndx * ndx);
      s.w_u.assign(ocp->w_u, ocp->w_u + (size_t)nu * nu);
      s.w_cent.assign(ocp->w_cent, ocp->w_cent + 36);
      s.w_centder.assign(ocp->w_centder, ocp->w_centder + 36);
      s.w_forces.assign(36, 0.0);
      s.w_frame.assign(ocp->w_frame, ocp->w_frame + 36);
      s.umin.assign(nu, 0.0); // (no torque box in this OCP)
      s.umax.assign(nu, 0.0);
      s.qmin.assign(ocp->qmin, ocp->qmin + na);
      s.qmax.assign(ocp->qmax, ocp->qmax + na);
      s.Kp.assign(6, 0.0);
      s.Kd.assign(6, 0.0);
      for (int i = 0; i < 3; i++)
        s.gravity[i] = ocp->gravity[i];
      s.mu = ocp->mu;
      s.Lfoot = ocp->Lfoot;
      s.Wfoot = ocp->Wfoot;
      s.force_size = 6;
      s.torque_limits = 0;
      s.kinematics_limits = ocp->kinematics_limits;
      s.force_cone = ocp->force_cone;
      s.land_cstr = 0; // (src/kinodynamics.cpp:134-146: land rows exist for 3-D feet only)
      s.terminal_constraint = ocp->terminal_constraint;
      for (int i = 0; i < na; i++)
        if (!(s.qmin[i] <= s.qmax[i]))
          return fail(SMPC_ERR_INVALID, "qmin must not exceed qmax (joint limits are indexed by actuated joint, 0 .. nv - 7)");
      auto sym = [](const std::vector<double> & w, int n) {
        for (int i = 0; i < n; i++)
          for (int j = 0; j < i; j++)
            if (std::fabs(w[(size_t)i * n + j] - w[(size_t)j * n + i]) > 1e-12 * (1.0 + std::fabs(w[(size_t)i * n + j])))
              return false;
        return true;
      };
      if (!sym(s.w_x, ndx) || !sym(s.w_u, nu) || !sym(s.w_cent, 6) || !sym(s.w_centder, 6) || !sym(s.w_frame, 6))
        return fail(SMPC_ERR_INVALID, "weight matrices must be symmetric");
      const HostMpcSettings ms6 = host_mpc(mpc);
      return guarded([&] {
        std::unique_ptr<smpc_handle> h(new smpc_handle());
#if !defined(SMPC_KINO_ONLY) || defined(SMPC_TALOS_TOO) // (experiment builds: -DSMPC_KINO_ONLY -DSMPC_TALOS_TOO = Go2 kinodynamics + the biped's two dense engines)
        h->full.reset(new FullEngine<KinoTalos>(robot, s, ms6, batch, gravity_arg, device_id));
#else
        throw std::runtime_error("SMPC_KINO_ONLY experiment build");
#endif
        *out = h.release();
      });
    }
    const int nv = robot->nv, ndx = 2 * nv, nu = nv - 6 + 3 * robot->nfeet;
    HostKinoSettings ks;
    ks.timestep = ocp->timestep;
    ks.w_x.assign(ocp->w_x, ocp->w_x + (size_t)ndx * ndx);
    ks.w_u.assign(ocp->w_u, ocp->w_u + (size_t)nu * nu);
    ks.w_frame.assign(ocp->w_frame, ocp->w_frame + 9);
    ks.w_cent.assign(ocp->w_cent, ocp->w_cent + 36);
    ks.w_centder.assign(ocp->w_centder, ocp->w_centder + 36);
    ks.qmin.assign(ocp->qmin, ocp->qmin + nv - 6);
    ks.qmax.assign(ocp->qmax, ocp->qmax + nv - 6);
    for (int i = 0; i < 3; i++)
      ks.gravity[i] = ocp->gravity[i];
    ks.kinematics_limits = ocp->kinematics_limits;
    ks.terminal_constraint = ocp->terminal_constraint;
    ks.force_cone = ocp->force_cone;
    ks.mu = ocp->mu;
    ks.land_cstr = ocp->land_cstr;
    for (int i = 0; i < nv - 6; i++)
      if (!(ks.qmin[i] <= ks.qmax[i]))
        return fail(SMPC_ERR_INVALID, "qmin must not exceed qmax (joint limits are indexed by actuated joint, 0 .. nv - 7)");
    // the weights enter the Gauss-Newton Hessian as they are: they must be symmetric
    for (int i = 0; i < ndx; i++)
      for (int j = 0; j < i; j++)
        if (std::fabs(ks.w_x[(size_t)i * ndx + j] - ks.w_x[(size_t)j * ndx + i]) > 1e-12 * (1.0 + std::fabs(ks.w_x[(size_t)i * ndx + j])))
          return fail(SMPC_ERR_INVALID, "w_x must be symmetric");
    for (int i = 0; i < nu; i++)
      for (int j = 0; j < i; j++)
        if (std::fabs(ks.w_u[(size_t)i * nu + j] - ks.w_u[(size_t)j * nu + i]) > 1e-12 * (1.0 + std::fabs(ks.w_u[(size_t)i * nu + j])))
          return fail(SMPC_ERR_INVALID, "w_u must be symmetric");
    HostMpcSettings ms;
    ms.swing_apex = mpc->swing_apex;
    ms.support_force = mpc->support_force;
    ms.TOL = mpc->TOL;
    ms.mu_init = mpc->mu_init;
    ms.timestep = mpc->timestep;
    ms.max_iters = mpc->max_iters;
    ms.num_threads = mpc->num_threads;
    ms.T_fly = mpc->T_fly;
    ms.T_contact = mpc->T_contact;
    ms.T = mpc->T;
    return guarded([&] {
      std::unique_ptr<smpc_handle> h(new smpc_handle());
#ifdef SMPC_CENT_ONLY
      throw std::runtime_error("SMPC_CENT_ONLY experiment build");
#else
      h->eng.reset(new KinoEngine<DimsGo2>(robot, ks, ms, batch, gravity_arg, device_id));
#endif
      *out = h.release();
    });
  }
  int smpc_create_centroidal(
    const smpc_robot_model * robot, const smpc_centroidal_settings * ocp, const smpc_mpc_settings * mpc, int batch, double gravity_arg,
    int device_id, smpc_handle ** out)
  {
    if (!robot || !ocp || !mpc || !out || !ocp->w_u || !ocp->w_com || !ocp->w_linear_mom || !ocp->w_angular_mom || !ocp->w_linear_acc
        || !ocp->w_angular_acc)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (device_count() <= 0)
      return fail(SMPC_ERR_NO_DEVICE, "no HIP device visible: the MPC engine has no CPU path");
    if (ocp->force_size != 3 && ocp->force_size != 6)
      return fail(SMPC_ERR_INVALID, "force size in settings does not match reference force size");
    const bool quad = ocp->force_size == 6;
    if (quad ? (robot->nfeet != CentTalos::NF || robot->njoints != FullTalos::NJ) : (robot->nfeet != CentGo2::NF || robot->njoints != DimsGo2::NJ))
      return fail(SMPC_ERR_INVALID, "robot shape (njoints, nfeet, force_size) does not match a built kernel instantiation");
    if (mpc->T < 2)
      return fail(SMPC_ERR_INVALID, "horizon must have at least 2 stages");
    const int nu = ocp->force_size * robot->nfeet;
    HostCentSettings cs;
    cs.timestep = ocp->timestep;
    cs.w_u.assign(ocp->w_u, ocp->w_u + (size_t)nu * nu);
    cs.w_com.assign(ocp->w_com, ocp->w_com + 9);
    cs.w_linear_mom.assign(ocp->w_linear_mom, ocp->w_linear_mom + 9);
    cs.w_angular_mom.assign(ocp->w_angular_mom, ocp->w_angular_mom + 9);
    cs.w_linear_acc.assign(ocp->w_linear_acc, ocp->w_linear_acc + 9);
    cs.w_angular_acc.assign(ocp->w_angular_acc, ocp->w_angular_acc + 9);
    for (int i = 0; i < 3; i++)
      cs.gravity[i] = ocp->gravity[i];
    cs.mu = ocp->mu;
    cs.Lfoot = ocp->Lfoot;
    cs.Wfoot = ocp->Wfoot;
    cs.force_size = ocp->force_size;
    auto sym = [](const std::vector<double> & w, int n) {
      for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
          if (std::fabs(w[(size_t)i * n + j] - w[(size_t)j * n + i]) > 1e-12 * (1.0 + std::fabs(w[(size_t)i * n + j])))
            return false;
      return true;
    };
    if (!sym(cs.w_u, nu) || !sym(cs.w_com, 3) || !sym(cs.w_linear_mom, 3) || !sym(cs.w_angular_mom, 3) || !sym(cs.w_linear_acc, 3)
        || !sym(cs.w_angular_acc, 3))
      return fail(SMPC_ERR_INVALID, "weight matrices must be symmetric");
    const HostMpcSettings ms = host_mpc(mpc);
    return guarded([&] {
      std::unique_ptr<smpc_handle> h(new smpc_handle());
#if !defined(SMPC_KINO_ONLY) || defined(SMPC_XCHECK_SUBSET) // (the cross-check HIP build: Go2 kinodynamics + both centroidal engines)
      if (quad)
        h->cent.reset(new CentEngine<FullTalos, CentTalos>(robot, cs, ms, batch, gravity_arg, device_id));
      else
        h->cent.reset(new CentEngineGo2(robot, cs, ms, batch, gravity_arg, device_id));
#elif defined(SMPC_CENT_ONLY)
      if (quad)
        throw std::runtime_error("SMPC_CENT_ONLY experiment build");
      h->cent.reset(new CentEngineGo2(robot, cs, ms, batch, gravity_arg, device_id));
#else
      (void)quad;
      throw std::runtime_error("SMPC_KINO_ONLY experiment build");
#endif
      *out = h.release();
    });
  }
  int smpc_create_fulldynamics(
    const smpc_robot_model * robot, const smpc_fulldynamics_settings * ocp, const smpc_mpc_settings * mpc, int batch, double gravity_arg,
    int device_id, smpc_handle ** out)
  {
    if (!robot || !ocp || !mpc || !out || !ocp->w_x || !ocp->w_u || !ocp->w_cent || !ocp->w_forces || !ocp->w_frame || !ocp->umin || !ocp->umax
        || !ocp->qmin || !ocp->qmax || !ocp->Kp_correction || !ocp->Kd_correction)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (device_count() <= 0)
      return fail(SMPC_ERR_NO_DEVICE, "no HIP device visible: the MPC engine has no CPU path");
    if (ocp->force_size != 3 && ocp->force_size != 6)
      return fail(SMPC_ERR_INVALID, "force size in settings does not match reference force size");
    if (mpc->T < 2)
      return fail(SMPC_ERR_INVALID, "horizon must have at least 2 stages");
    const int nv = robot->nv, ndx = 2 * nv, nu = nv - 6, fs = ocp->force_size;
    HostFullSettings s;
    s.timestep = ocp->timestep;
    s.w_x.assign(ocp->w_x, ocp->w_x + (size_t)ndx * ndx);
    s.w_u.assign(ocp->w_u, ocp->w_u + (size_t)nu * nu);
    s.w_cent.assign(ocp->w_cent, ocp->w_cent + 36);
    s.w_forces.assign(ocp->w_forces, ocp->w_forces + fs * fs);
    s.w_frame.assign(ocp->w_frame, ocp->w_frame + fs * fs);
    s.umin.assign(ocp->umin, ocp->umin + nu);
    s.umax.assign(ocp->umax, ocp->umax + nu);
    s.qmin.assign(ocp->qmin, ocp->qmin + nu);
    s.qmax.assign(ocp->qmax, ocp->qmax + nu);
    s.Kp.assign(ocp->Kp_correction, ocp->Kp_correction + fs);
    s.Kd.assign(ocp->Kd_correction, ocp->Kd_correction + fs);
    for (int i = 0; i < 3; i++)
      s.gravity[i] = ocp->gravity[i];
    s.mu = ocp->mu;
    s.Lfoot = ocp->Lfoot;
    s.Wfoot = ocp->Wfoot;
    s.force_size = fs;
    s.torque_limits = ocp->torque_limits;
    s.kinematics_limits = ocp->kinematics_limits;
    s.force_cone = ocp->force_cone;
    s.land_cstr = ocp->land_cstr;
    s.terminal_constraint = ocp->terminal_constraint;
    for (int i = 0; i < nu; i++)
      if (!(s.qmin[i] <= s.qmax[i]) || !(s.umin[i] <= s.umax[i]))
        return fail(SMPC_ERR_INVALID, "lower limits must not exceed upper limits (indexed by actuated joint, 0 .. nv - 7)");
    auto sym = [](const std::vector<double> & w, int n) {
      for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
          if (std::fabs(w[(size_t)i * n + j] - w[(size_t)j * n + i]) > 1e-12 * (1.0 + std::fabs(w[(size_t)i * n + j])))
            return false;
      return true;
    };
    if (!sym(s.w_x, ndx) || !sym(s.w_u, nu) || !sym(s.w_cent, 6) || !sym(s.w_forces, fs) || !sym(s.w_frame, fs))
      return fail(SMPC_ERR_INVALID, "weight matrices must be symmetric");
    const HostMpcSettings ms = host_mpc(mpc);
    return guarded([&] {
      std::unique_ptr<smpc_handle> h(new smpc_handle());
#if defined(SMPC_KINO_ONLY) && defined(SMPC_TALOS_TOO)
      if (robot->njoints == FullTalos::NJ && robot->nfeet == FullTalos::NF && fs == FullTalos::FS && !s.land_cstr)
        h->full.reset(new FullEngine<FullTalos>(robot, s, ms, batch, gravity_arg, device_id));
      else
        throw std::runtime_error("SMPC_KINO_ONLY experiment build");
#elif defined(SMPC_KINO_ONLY) && defined(SMPC_GO2FULL_TOO)
      if (robot->njoints == FullGo2::NJ && robot->nfeet == FullGo2::NF && fs == FullGo2::FS && !s.land_cstr && !s.force_cone)
        h->full.reset(new FullEngine<FullGo2>(robot, s, ms, batch, gravity_arg, device_id));
      else
        throw std::runtime_error("SMPC_KINO_ONLY experiment build");
#elif defined(SMPC_KINO_ONLY)
      throw std::runtime_error("SMPC_KINO_ONLY experiment build");
#else
      if (robot->njoints == FullGo2::NJ && robot->nfeet == FullGo2::NF && fs == FullGo2::FS && s.land_cstr && s.force_cone)
        h->full.reset(new FullEngine<FullGo2ConeLand>(robot, s, ms, batch, gravity_arg, device_id));
      else if (robot->njoints == FullGo2::NJ && robot->nfeet == FullGo2::NF && fs == FullGo2::FS && s.land_cstr)
        h->full.reset(new FullEngine<FullGo2Land>(robot, s, ms, batch, gravity_arg, device_id));
      else if (robot->njoints == FullTalos::NJ && robot->nfeet == FullTalos::NF && fs == FullTalos::FS && s.land_cstr)
        h->full.reset(new FullEngine<FullTalosLand>(robot, s, ms, batch, gravity_arg, device_id));
      else if (robot->njoints == FullGo2::NJ && robot->nfeet == FullGo2::NF && fs == FullGo2::FS && s.force_cone)
        h->full.reset(new FullEngine<FullGo2Cone>(robot, s, ms, batch, gravity_arg, device_id));
      else if (robot->njoints == FullGo2::NJ && robot->nfeet == FullGo2::NF && fs == FullGo2::FS)
        h->full.reset(new FullEngine<FullGo2>(robot, s, ms, batch, gravity_arg, device_id));
      else if (robot->njoints == FullTalos::NJ && robot->nfeet == FullTalos::NF && fs == FullTalos::FS)
        h->full.reset(new FullEngine<FullTalos>(robot, s, ms, batch, gravity_arg, device_id));
      else
        throw std::runtime_error("robot shape (njoints, nfeet, force_size) does not match a built kernel instantiation");
#endif
      *out = h.release();
    });
  }
  int smpc_get_contact_forces(smpc_handle * h, double * out)
  {
    if (!h || !out)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (!h->full)
      return fail(SMPC_ERR_INVALID, "smpc_get_contact_forces needs a full-dynamics handle (the other problems carry the forces in us)");
    return guarded([&] { h->full->get(9, out); });
  }
  int smpc_destroy(smpc_handle * h)
  {
    delete h;
    return SMPC_OK;
  }
  int smpc_get_dims(const smpc_handle * h, int * d)
  {
    if (!h || !d)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
    {
      for (int i = 0; i < 8; i++)
        d[i] = h->full->dims[i];
      return SMPC_OK;
    }
    if (h->cent)
    {
      d[0] = h->cent->nq_mb;
      d[1] = h->cent->nv_mb;
      d[2] = 9;
      d[3] = 9;
      d[4] = h->cent->nu;
      d[5] = h->cent->nc;
      d[6] = h->cent->nf;
      d[7] = h->cent->H;
      return SMPC_OK;
    }
    d[0] = DimsGo2::NQ;
    d[1] = DimsGo2::NV;
    d[2] = DimsGo2::NX;
    d[3] = DimsGo2::NDX;
    d[4] = DimsGo2::NU;
    d[5] = DimsGo2::NC;
    d[6] = DimsGo2::NF;
    d[7] = h->eng->H;
    return SMPC_OK;
  }
  int smpc_generate_cycle_horizon(smpc_handle * h, const uint8_t * cs, int n)
  {
    if (!h || !cs)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->generate_cycle_horizon(cs, n); });
    if (h->cent)
      return guarded([&] { h->cent->generate_cycle_horizon(cs, n); });
    return guarded([&] { h->eng->generate_cycle_horizon(cs, n); });
  }
  int smpc_switch_to_walk(smpc_handle * h, const double * v6)
  {
    if (!h || !v6)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->switch_to_walk(v6); });
    if (h->cent)
      return guarded([&] { h->cent->switch_to_walk(v6); });
    return guarded([&] { h->eng->switch_to_walk(v6); });
  }
  int smpc_switch_to_stand(smpc_handle * h)
  {
    if (!h)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->switch_to_stand(); });
    if (h->cent)
      return guarded([&] { h->cent->switch_to_stand(); });
    return guarded([&] { h->eng->switch_to_stand(); });
  }
  int smpc_set_velocity_base_batched(smpc_handle * h, const double * V)
  {
    if (!h || !V)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->set_velocity_base_batched(V); });
    if (h->cent)
      return guarded([&] { h->cent->set_velocity_base_batched(V); });
    return guarded([&] { h->eng->set_velocity_base_batched(V); });
  }
  int smpc_set_stage_reference(smpc_handle * h, int t, int what, const double * v, int n)
  {
    if (!h || !v)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->set_stage_reference(t, what, v, n); });
    if (h->cent)
      return guarded([&] { h->cent->set_stage_reference(t, what, v, n); });
    return guarded([&] { h->eng->set_stage_reference(t, what, v, n); });
  }
  int smpc_get_stage_reference(smpc_handle * h, int t, int what, double * v, int n)
  {
    if (!h || !v)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->get_stage_reference(t, what, v, n); });
    if (h->cent)
      return guarded([&] { h->cent->get_stage_reference(t, what, v, n); });
    return guarded([&] { h->eng->get_stage_reference(t, what, v, n); });
  }
  int smpc_set_reference_pose(smpc_handle * h, int t, int foot, const double * p3)
  {
    if (!h || !p3)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->set_reference_pose(t, foot, p3); });
    if (h->cent)
      return guarded([&] { h->cent->set_reference_pose(t, foot, p3); });
    return guarded([&] { h->eng->set_reference_pose(t, foot, p3); });
  }
  int smpc_get_reference_pose(smpc_handle * h, int t, int foot, int instance, double * p3)
  {
    if (!h || !p3)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->get_reference_pose(t, foot, instance, p3); });
    if (h->cent)
      return guarded([&] { h->cent->get_reference_pose(t, foot, instance, p3); });
    return guarded([&] { h->eng->get_reference_pose(t, foot, instance, p3); });
  }
  int smpc_set_reference_pose_se3(smpc_handle * h, int t, int foot, const double * p3, const double * R9)
  {
    if (!h || !p3 || !R9)
      return fail(SMPC_ERR_INVALID, "null argument");
    const int rc = smpc_set_reference_pose(h, t, foot, p3);
    if (rc != SMPC_OK)
      return rc;
    if (h->full)
      return guarded([&] { h->full->set_reference_rotation(t, foot, R9); });
    if (h->cent)
      return guarded([&] { h->cent->set_reference_rotation(t, foot, R9); });
    return guarded([&] { h->eng->set_reference_rotation(t, foot, R9); });
  }
  int smpc_get_reference_pose_se3(smpc_handle * h, int t, int foot, int instance, double * p3, double * R9)
  {
    if (!h || !p3 || !R9)
      return fail(SMPC_ERR_INVALID, "null argument");
    const int rc = smpc_get_reference_pose(h, t, foot, instance, p3);
    if (rc != SMPC_OK)
      return rc;
    if (h->full)
      return guarded([&] { h->full->get_reference_rotation(t, foot, R9); });
    if (h->cent)
      return guarded([&] { h->cent->get_reference_rotation(t, foot, R9); });
    return guarded([&] { h->eng->get_reference_rotation(t, foot, R9); });
  }
  int smpc_get_contact_state(smpc_handle * h, int t, uint8_t * out)
  {
    if (!h || !out)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] {
      const unsigned m = h->full ? h->full->contact_mask(t) : (h->cent ? h->cent->contact_mask(t) : h->eng->contact_mask(t));
      const int nf = h->full ? h->full->dims[6] : (h->cent ? h->cent->nf : DimsGo2::NF);
      for (int f = 0; f < nf; f++)
        out[f] = (m >> f) & 1u;
    });
  }
  int smpc_get_cycling_contact_state(smpc_handle * h, int t, uint8_t * out)
  {
    if (!h)
      return fail(SMPC_ERR_INVALID, "null argument");
    const GaitTimer & tm = h->full ? h->full->timer : (h->cent ? h->cent->timer : h->eng->timer);
    const int n = (int)tm.states.size();
    if (!out)
      return n;
    if (n == 0)
      return fail(SMPC_ERR_INVALID, "generateCycleHorizon has not been called");
    if (t < 0 || t >= n)
      return fail(SMPC_ERR_INVALID, "Stage index exceeds the cycle length");
    for (int f = 0; f < tm.nf; f++)
      out[f] = tm.states[t][f];
    return n;
  }
  int smpc_set_x_reference(smpc_handle * h, const double * x)
  {
    if (!h || !x)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      h->full->x_reference.assign(x, x + h->full->dims[2]);
    else if (h->cent)
      std::copy(x, x + 9, h->cent->x_reference);
    else
      h->eng->x_reference.assign(x, x + DimsGo2::NX);
    return SMPC_OK;
  }
  int smpc_iterate(smpc_handle * h, const double * X)
  {
    if (!h || !X)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->iterate_host(X); });
    if (h->cent)
      return guarded([&] { h->cent->iterate_host(X); });
    return guarded([&] { h->eng->iterate_host(X); });
  }
  int smpc_iterate_async(smpc_handle * h, const double * X)
  {
    if (!h || !X)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->iterate_host(X); }); // (synchronous on these handles)
    if (h->cent)
      return guarded([&] { h->cent->iterate_host(X); });
    return guarded([&] { h->eng->iterate_host_async(X); });
  }
  int smpc_gather_outputs(smpc_handle * h, double * out, size_t row_doubles)
  {
    if (!h || !out)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (!h->eng)
      return fail(SMPC_ERR_INVALID, KINO_ONLY);
    return guarded([&] { h->eng->gather_outputs_async(out, row_doubles); });
  }
  int smpc_gather_outputs_device(smpc_handle * h, double * out_device, size_t row_doubles)
  {
    if (!h || !out_device)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (!h->eng)
      return fail(SMPC_ERR_INVALID, KINO_ONLY);
    return guarded([&] { h->eng->gather_outputs_device(out_device, row_doubles); });
  }
  int smpc_gather_outputs_peer(smpc_handle * h, double * out_peer, int dst_device)
  {
    if (!h || !out_peer)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (!h->eng)
      return fail(SMPC_ERR_INVALID, KINO_ONLY);
    if (dst_device < 0 || dst_device >= device_count())
      return fail(SMPC_ERR_INVALID, "destination device out of range");
    return guarded([&] { h->eng->gather_outputs_peer(out_peer, dst_device); });
  }
  int smpc_iterate_device(smpc_handle * h, const double * Xd)
  {
    if (!h || !Xd)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->iterate_device(Xd); });
    if (h->cent)
      return guarded([&] { h->cent->iterate_device(Xd); });
    return guarded([&] { h->eng->iterate_device(Xd); });
  }
  int smpc_wait(smpc_handle * h)
  {
    if (!h)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->sync(); });
    if (h->cent)
      return guarded([&] { h->cent->sync(); });
    return guarded([&] { h->eng->sync(); });
  }
  void * smpc_get_stream(smpc_handle * h)
  {
    if (!h)
      return nullptr;
    return stream_native(h->full ? h->full->stream : (h->cent ? h->cent->stream : h->eng->stream));
  }
  static size_t state_pass(smpc_handle * h, StateIO::Mode mode, void * buf, size_t cap)
  {
    if (h->full)
    {
      StateIO io(mode, buf, cap, h->full->stream);
      return h->full->state_io(io);
    }
    if (h->cent)
    {
      StateIO io(mode, buf, cap, h->cent->stream);
      return h->cent->state_io(io);
    }
    StateIO io(mode, buf, cap, h->eng->stream);
    return h->eng->state_io(io);
  }
  int smpc_state_size(smpc_handle * h, size_t * bytes)
  {
    if (!h || !bytes)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] { *bytes = state_pass(h, StateIO::COUNT, nullptr, 0); });
  }
  int smpc_save_state(smpc_handle * h, void * buffer, size_t capacity, size_t * written)
  {
    if (!h || !buffer)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] {
      const size_t n = state_pass(h, StateIO::SAVE, buffer, capacity);
      if (written)
        *written = n;
    });
  }
  int smpc_load_state(smpc_handle * h, const void * buffer, size_t size)
  {
    if (!h || !buffer)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] { state_pass(h, StateIO::LOAD, const_cast<void *>(buffer), size); });
  }
  int smpc_get_x_device(smpc_handle * h, int t, double * out_device)
  {
    if (!h || !out_device)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->gather_x_device(t, out_device); });
    if (h->cent)
      return fail(SMPC_ERR_INVALID, KINO_ONLY);
    return guarded([&] { h->eng->gather_x_device(t, out_device); });
  }
  int smpc_get_xs(smpc_handle * h, double * out)
  {
    if (h->full)
      return guarded([&] { h->full->get(0, out); });
    if (h->cent)
      return guarded([&] { h->cent->get_ring(h->cent->bufs().xs, 9, h->cent->H + 1, out); });
    return guarded([&] { h->eng->get_ring(h->eng->buf.xs, DimsGo2::NX, h->eng->H + 1, out); });
  }
  int smpc_get_us(smpc_handle * h, double * out)
  {
    if (h->full)
      return guarded([&] { h->full->get(1, out); });
    if (h->cent)
      return guarded([&] { h->cent->get_ring(h->cent->bufs().us, h->cent->nu, h->cent->H, out); });
    return guarded([&] { h->eng->get_ring(h->eng->buf.us, DimsGo2::NU, h->eng->H, out); });
  }
  int smpc_get_vs(smpc_handle * h, double * out)
  {
    if (h->full)
      return guarded([&] { h->full->get(4, out); });
    if (h->cent)
      return guarded([&] { h->cent->get_ring(h->cent->bufs().vs, h->cent->nc, h->cent->H, out); });
    return guarded([&] { h->eng->get_ring(h->eng->buf.vs, DimsGo2::NC, h->eng->H, out); });
  }
  int smpc_debug_get_extra_multipliers(smpc_handle * h, int which, double * out)
  {
    if (!h || !out || h->full || h->cent)
      return fail(SMPC_ERR_INVALID, "kinodynamics handles only");
    double * src = which == 0 ? h->eng->buf.es : h->eng->buf.ls;
    if (!src)
      return fail(SMPC_ERR_INVALID, "the problem has no such rows");
    return guarded([&] { h->eng->get_ring(src, which == 0 ? 2 * DimsGo2::NF : DimsGo2::NF, h->eng->H, out); });
  }
  int smpc_get_lams(smpc_handle * h, double * out)
  {
    if (h->full)
      return guarded([&] { h->full->get(5, out); });
    // device arrays hold lambda_{t+1} at stage t; the API returns lams[0..H] with lams[0] = 0
    if (h->cent)
      return guarded([&] {
        auto & e = *h->cent;
        std::vector<double> tmp((size_t)e.B * e.H * 9);
        e.get_ring(e.bufs().lams, 9, e.H, tmp.data());
        for (int b = 0; b < e.B; b++)
        {
          double * o = out + (size_t)b * (e.H + 1) * 9;
          std::memset(o, 0, 9 * sizeof(double));
          std::memcpy(o + 9, tmp.data() + (size_t)b * e.H * 9, (size_t)e.H * 9 * sizeof(double));
        }
      });
    return guarded([&] {
      auto & e = *h->eng;
      std::vector<double> tmp((size_t)e.B * e.H * DimsGo2::NDX);
      e.get_ring(e.buf.lams, DimsGo2::NDX, e.H, tmp.data());
      for (int b = 0; b < e.B; b++)
      {
        double * o = out + (size_t)b * (e.H + 1) * DimsGo2::NDX;
        std::memset(o, 0, DimsGo2::NDX * sizeof(double));
        std::memcpy(o + DimsGo2::NDX, tmp.data() + (size_t)b * e.H * DimsGo2::NDX, (size_t)e.H * DimsGo2::NDX * sizeof(double));
      }
    });
  }
  int smpc_get_K0(smpc_handle * h, double * out)
  {
    if (h->full)
      return guarded([&] { h->full->get(2, out); });
    if (h->cent)
      return guarded([&] { h->cent->get_K(out, false); });
    return guarded([&] { h->eng->get_K(out, false); });
  }
  int smpc_get_Ks(smpc_handle * h, double * out)
  {
    if (h->full)
      return guarded([&] { h->full->get(3, out); });
    if (h->cent)
      return guarded([&] { h->cent->get_K(out, true); });
    return guarded([&] { h->eng->get_K(out, true); });
  }
  int smpc_get_state_derivative01(smpc_handle * h, double * out)
  {
    if (h->full)
      return guarded([&] { h->full->get(6, out); });
    if (h->cent)
      return guarded([&] { h->cent->get_linear(h->cent->bufs().xdot01, (size_t)h->cent->B * 18, out); });
    return guarded([&] { h->eng->get_linear(h->eng->buf.xdot01, (size_t)h->eng->B * 4 * DimsGo2::NV, out); });
  }
  int smpc_get_reference_poses(smpc_handle * h, double * out)
  {
    if (h->full)
      return guarded([&] { h->full->get(7, out); });
    if (h->cent)
      return guarded([&] { h->cent->get_linear(h->cent->bufs().foot, (size_t)h->cent->B * h->cent->H * h->cent->nf * 3, out); });
    return guarded([&] { h->eng->get_linear(h->eng->buf.foot_ref, (size_t)h->eng->B * h->eng->H * DimsGo2::NF * 3, out); });
  }
  int smpc_get_foot_timing(smpc_handle * h, int foot, int which, int * out, int cap)
  {
    const GaitTimer * tm = !h ? nullptr : (h->full ? &h->full->timer : (h->cent ? &h->cent->timer : &h->eng->timer));
    if (!tm || foot < 0 || foot >= tm->nf || tm->nf == 0)
    {
      fail(SMPC_ERR_INVALID, "invalid foot index or cycle horizon not generated");
      return SMPC_ERR_INVALID;
    }
    const std::vector<int> & v = which ? tm->land[foot] : tm->takeoff[foot];
    for (int i = 0; i < (int)v.size() && i < cap; i++)
      out[i] = v[i];
    return (int)v.size();
  }
  int smpc_get_info(smpc_handle * h, double * out)
  {
    if (h->full)
      return guarded([&] { h->full->get(8, out); });
    if (h->cent)
      return guarded([&] { h->cent->get_linear(h->cent->bufs().scal, (size_t)h->cent->B * SC_N, out); });
    return guarded([&] { h->eng->get_linear(h->eng->buf.scal, (size_t)h->eng->B * SC_N, out); });
  }
  int smpc_get_status(smpc_handle * h, int * out)
  {
    if (!h || !out)
      return fail(SMPC_ERR_INVALID, "null argument");
    const int B = h->full ? h->full->B : (h->cent ? h->cent->B : h->eng->B);
    std::vector<double> info((size_t)B * SC_N);
    const int rc = smpc_get_info(h, info.data());
    if (rc < 0)
      return rc;
    int bad = 0;
    for (int b = 0; b < B; b++)
    {
      const double * s = &info[(size_t)b * SC_N];
      int w = 0;
      for (int i = 0; i < 12; i++)
        if (!std::isfinite(s[i]))
          w |= SMPC_STATUS_NONFINITE;
      if (s[SC_LS_FAILED] != 0.0)
        w |= SMPC_STATUS_LS_FAILED;
      if (s[SC_PREG] >= 1e9)
        w |= SMPC_STATUS_REG_SATURATED;
      out[b] = w;
      bad += w != 0;
    }
    return bad;
  }
  int smpc_get_cold_trace(smpc_handle * h, double * out, int cap)
  {
    const int n = h->full ? h->full->cold_iters : (h->cent ? h->cent->cold_iters : h->eng->cold_iters);
    const std::vector<double> & tr = h->full ? h->full->cold_trace : (h->cent ? h->cent->cold_trace : h->eng->cold_trace);
    for (int i = 0; i < n && i < cap; i++)
      for (int k = 0; k < 4; k++)
        out[i * 4 + k] = tr[(size_t)i * 4 + k];
    return n;
  }
  // (size of what smpc_debug_get_lq returns: the row-major form of a kinodynamics knot)
  int smpc_lq_size(const smpc_handle * h)
  {
    typedef DimsGo2 D;
    return (h && h->full) ? h->full->lq_size() : (D::O_T - D::O_A) + D::NDX * D::NDX + D::NDX * D::NU + D::NU * D::NU + (D::O_vpd + D::NC - D::O_C);
  }
  int smpc_set_early_exit_on_tol(smpc_handle * h, int on)
  {
    if (!h)
      return fail(SMPC_ERR_INVALID, "null handle");
    if (h->cent)
      return fail(SMPC_ERR_INVALID, "smpc_set_early_exit_on_tol: kinodynamics and full-dynamics handles (the centroidal step is one fused kernel)");
    if (h->full)
      return guarded([&] { h->full->set_early_exit(on != 0); });
    h->eng->early_exit_on_tol = on != 0;
    return SMPC_OK;
  }
  int smpc_debug_get_lq(smpc_handle * h, int inst, int t, double * out)
  {
    if (h && h->full)
      return guarded([&] { h->full->debug_lq(inst, t, out); });
    if (h && h->cent)
      return fail(SMPC_ERR_INVALID, KINO_ONLY);
    if (!h || inst < 0 || inst >= h->eng->B || t < 0 || t >= h->eng->H)
      return fail(SMPC_ERR_INVALID, "Stage index exceeds stage vector size");
    return guarded([&] {
      // the device keeps [Q S; S^T R] as accumulator-layout tiles (Dims::O_T); returned in the documented row-major order
      // A | B | Q | S | R | C | q | r | f | d | lx | lu | lpd | vpd
      typedef DimsGo2 D;
      std::vector<double> raw(D::LQ_STRIDE);
      h->eng->get_linear(h->eng->buf.lq + ((size_t)inst * h->eng->H + t) * D::LQ_STRIDE, D::LQ_STRIDE, raw.data());
      constexpr int n = D::NDX, m = D::NU;
      double * o = out;
      std::copy(raw.begin() + D::O_A, raw.begin() + D::O_T, o); // A | B
      o += D::O_T - D::O_A;
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++)
          *o++ = raw[D::q_off(i, j)];
      for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++)
          *o++ = raw[D::s_off(i, j)];
      for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++)
          *o++ = raw[D::r_off(i, j)];
      std::copy(raw.begin() + D::O_C, raw.begin() + D::O_vpd + D::NC, o);
    });
  }
  int smpc_debug_get_steps(smpc_handle * h, double * dxs, double * dus)
  {
    if (h && h->full)
      return guarded([&] { h->full->debug_steps(dxs, dus); });
    if (h && h->cent)
      return guarded([&] {
        auto & e = *h->cent;
        e.get_linear(e.bufs().dxs, (size_t)e.B * (e.H + 1) * 9, dxs);
        e.get_linear(e.bufs().dus, (size_t)e.B * e.H * e.nu, dus);
      });
    return guarded([&] {
      auto & e = *h->eng;
      e.get_linear(e.buf.dxs, (size_t)e.B * (e.H + 1) * DimsGo2::NDX, dxs);
      e.get_linear(e.buf.dus, (size_t)e.B * e.H * DimsGo2::NU, dus);
    });
  }
  int smpc_debug_get_terminal(smpc_handle * h, int inst, double * QN, double * qN)
  {
    if (h && h->full)
      return guarded([&] { h->full->debug_terminal(inst, QN, qN); });
    if (h && h->cent)
      return fail(SMPC_ERR_INVALID, KINO_ONLY);
    if (!h || inst < 0 || inst >= h->eng->B)
      return fail(SMPC_ERR_INVALID, "instance index out of range");
    return guarded([&] {
      auto & e = *h->eng;
      e.get_linear(e.buf.QN + (size_t)inst * DimsGo2::NDX * DimsGo2::NDX, DimsGo2::NDX * DimsGo2::NDX, QN);
      e.get_linear(e.buf.qN + (size_t)inst * DimsGo2::NDX, DimsGo2::NDX, qN);
    });
  }
  int smpc_debug_get_phase_cycles(smpc_handle * h, double * out64)
  {
    if (h && h->full)
      return guarded([&] {
        if (!h->full->phase_cycles(out64))
          throw std::runtime_error("phase timers are off (set SMPC_PHASE_PROFILE=1 before smpc_create_fulldynamics)");
      });
    if (h && h->cent && h->cent->bufs().dbg)
      return guarded([&] { h->cent->get_linear(h->cent->bufs().dbg, 64, out64); });
    if (!h || h->cent || !h->eng->buf.dbg)
      return fail(SMPC_ERR_INVALID, "phase timers are off (set SMPC_PHASE_PROFILE=1 before smpc_create)");
    return guarded([&] { h->eng->get_linear(h->eng->buf.dbg, 64, out64); });
  }
  int smpc_set_profiling(smpc_handle * h, int en)
  {
    if (h->full)
      h->full->profiling = en != 0;
    else if (h->cent)
      h->cent->profiling = en != 0;
    else
      h->eng->profiling = en != 0;
    return SMPC_OK;
  }
  int smpc_kernel_time_slots(void) { return KID_N; }
  int smpc_get_kernel_times_n(smpc_handle * h, double * ms, long * calls, int n)
  {
    if (!h || !ms || !calls || n < 0)
      return fail(SMPC_ERR_INVALID, "null argument");
    const int m = n < KID_N ? n : (int)KID_N;
    if (h->full)
      return guarded([&] {
        h->full->collect_profile();
        for (int i = 0; i < m; i++)
        {
          ms[i] = h->full->kernel_ms[i];
          calls[i] = h->full->kernel_calls[i];
        }
      });
    if (h->cent)
      return guarded([&] {
        // centroidal handle: slot 0 = front-end kernel, slot 1 = the fused control-step kernel (6-D feet: recede, then deriv / riccati / forward /
        // line search in slots 2 .. 5)
        h->cent->collect_profile();
        for (int i = 0; i < m; i++)
        {
          ms[i] = i < CKID_N ? h->cent->kernel_ms[i] : 0.0;
          calls[i] = i < CKID_N ? h->cent->kernel_calls[i] : 0;
        }
      });
    return guarded([&] {
      h->eng->collect_profile();
      for (int i = 0; i < m; i++)
      {
        ms[i] = h->eng->kernel_ms[i];
        calls[i] = h->eng->kernel_calls[i];
      }
    });
  }
  // (kept for callers of the first form: it writes smpc_kernel_time_slots() entries -- use smpc_get_kernel_times_n with the capacity of your arrays)
  int smpc_get_kernel_times(smpc_handle * h, double * ms, long * calls) { return smpc_get_kernel_times_n(h, ms, calls, KID_N); }
  int smpc_reset_kernel_times(smpc_handle * h)
  {
    if (h->full)
      return guarded([&] {
        h->full->collect_profile();
        for (int i = 0; i < KID_N; i++)
        {
          h->full->kernel_ms[i] = 0;
          h->full->kernel_calls[i] = 0;
        }
      });
    if (h->cent)
      return guarded([&] {
        h->cent->collect_profile();
        for (int i = 0; i < CKID_N; i++)
        {
          h->cent->kernel_ms[i] = 0;
          h->cent->kernel_calls[i] = 0;
        }
      });
    return guarded([&] {
      h->eng->collect_profile();
      for (int i = 0; i < KID_N; i++)
      {
        h->eng->kernel_ms[i] = 0;
        h->eng->kernel_calls[i] = 0;
      }
    });
  }
  int smpc_update_internal_data(smpc_handle * h, const double * X, double * feet, double * com, double * hg, double * centroidal_state)
  {
    if (!h || !X)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->update_internal_data(X, feet, com, hg, centroidal_state); });
    if (h->cent)
      return guarded([&] { h->cent->update_internal_data(X, feet, com, hg, centroidal_state); });
    return guarded([&] { h->eng->update_internal_data(X, feet, com, hg, centroidal_state); });
  }
  int smpc_full_forward_dynamics(
    smpc_handle * h, int n, const double * X, const double * tau, const unsigned * contact_mask, const double * Kp,
    const double * Kd, double prox_accuracy, double prox_mu, int prox_max_iter, double * a_out, double * lambda_out,
    int * iters_out, double * kernel_ms)
  {
    if (!h || !X || !tau || !contact_mask || !a_out || !lambda_out)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full) // full-dynamics handle: its own robot, 3-D or 6-D contacts (Kp / Kd: force_size entries)
      return guarded([&] {
        h->full->full_forward_dynamics(n, X, tau, contact_mask, Kp, Kd, prox_accuracy, prox_mu, prox_max_iter, a_out, lambda_out, iters_out, kernel_ms);
      });
    if (!h->eng)
      return fail(SMPC_ERR_INVALID, "smpc_full_forward_dynamics needs a kinodynamics or a full-dynamics handle (they carry the multibody model)");
    return guarded([&] {
      h->eng->full_forward_dynamics(n, X, tau, contact_mask, Kp, Kd, prox_accuracy, prox_mu, prox_max_iter, a_out, lambda_out,
                                    iters_out, kernel_ms);
    });
  }
  int smpc_riccati_feedback(smpc_handle * h, double delay, const double * X_meas, double * u_out)
  {
    if (!h || !X_meas || !u_out)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full)
      return guarded([&] { h->full->riccati_feedback(delay, X_meas, u_out); });
    if (h->cent)
      return guarded([&] { h->cent->interpolate(delay, 2, X_meas, nullptr, nullptr, nullptr, u_out); });
    return guarded([&] { h->eng->riccati_feedback(delay, X_meas, u_out); });
  }
  int smpc_interpolate(smpc_handle * h, double delay, int knots, double * x_out, double * acc_out, double * force_out)
  {
    if (!h)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (h->full) // full-dynamics handle: force_out [B][nfeet][force_size] = interpolated MPC::getContactForces
      return guarded([&] { h->full->interpolate(delay, knots, x_out, acc_out, force_out); });
    if (h->cent) // centroidal handle: x_out [B][9], acc_out = state derivative [B][9], force_out [B][3 nfeet]
      return guarded([&] { h->cent->interpolate(delay, knots, nullptr, x_out, acc_out, force_out, nullptr); });
    return guarded([&] { h->eng->interpolate(delay, knots, x_out, acc_out, force_out); });
  }
  int smpc_interpolate_knots(int kind, double delay, double timestep, const double * knots, int n, int dim, double * out, int device_id)
  {
    if (!knots || !out || n < 1 || dim < 1 || kind < 0 || kind > 2 || !(timestep > 0.0) || !(delay >= 0.0))
      return fail(SMPC_ERR_INVALID, "invalid argument");
    // the robot is told by the size of the knots: the two topologies the engines are instantiated for (free-flyer + nv - 6 joints)
    const bool biped = (kind == 0 && dim == FullTalos::NX) || (kind == 1 && dim == FullTalos::NQ);
    if (kind == 0 && dim != DimsGo2::NX && !biped)
      return fail(SMPC_ERR_INVALID, "State is not of the right size");
    if (kind == 1 && dim != DimsGo2::NQ && !biped)
      return fail(SMPC_ERR_INVALID, "Configuration is not of the right size");
    if (device_count() <= 0)
      return fail(SMPC_ERR_NO_DEVICE, "no HIP device visible: the interpolator has no CPU path");
    return guarded([&] {
      set_device(device_id);
      stream_t st = stream_create();
      double * dk = (double *)dev_alloc(((size_t)n * dim + dim) * sizeof(double));
      h2d(dk, knots, (size_t)n * dim * sizeof(double), st);
      auto run = [&](auto dims) {
        typedef decltype(dims) DD;
        InterpKnotsArgs<DD> ia;
        ia.kind = kind;
        ia.n = n;
        ia.dim = dim;
        ia.delay = delay;
        ia.timestep = timestep;
        ia.knots = dk;
        ia.out = dk + (size_t)n * dim;
        launch<InterpKnotsArgs<DD>, interp_knots_body<DD>, 64>(1, st, ia);
      };
#ifndef SMPC_KINO_ONLY
      if (biped)
        run(FullTalos());
      else
#endif
        run(DimsGo2());
      d2h(out, dk + (size_t)n * dim, (size_t)dim * sizeof(double), st);
      stream_sync(st);
      dev_free(dk);
      stream_destroy(st);
    });
  }
  int smpc_friction_compensation(
    const double * dry, const double * viscous, int nu, const double * velocity, int velocity_size, double * torque, int torque_size,
    int batch, int device_id)
  {
    if (!dry || !viscous || !velocity || !torque || nu < 1 || batch < 1)
      return fail(SMPC_ERR_INVALID, "invalid argument");
    if (velocity_size != nu)
      return fail(SMPC_ERR_INVALID, "Velocity has wrong size");
    if (torque_size != nu)
      return fail(SMPC_ERR_INVALID, "Torque has wrong size");
    if (device_count() <= 0)
      return fail(SMPC_ERR_NO_DEVICE, "no HIP device visible: the friction compensation has no CPU path");
    return guarded([&] {
      set_device(device_id);
      stream_t st = stream_create();
      const size_t total = (size_t)batch * nu;
      double * d = (double *)dev_alloc((2 * (size_t)nu + 2 * total) * sizeof(double));
      h2d(d, dry, (size_t)nu * sizeof(double), st);
      h2d(d + nu, viscous, (size_t)nu * sizeof(double), st);
      h2d(d + 2 * nu, velocity, total * sizeof(double), st);
      h2d(d + 2 * nu + total, torque, total * sizeof(double), st);
      FrictionArgs fa;
      fa.dry = d;
      fa.viscous = d + nu;
      fa.velocity = d + 2 * nu;
      fa.torque = d + 2 * nu + total;
      fa.nu = nu;
      fa.total = total;
      launch<FrictionArgs, friction_body, 256>((int)((total + 255) / 256), st, fa);
      d2h(torque, fa.torque, total * sizeof(double), st);
      stream_sync(st);
      dev_free(d);
      stream_destroy(st);
    });
  }
  int smpc_centroidal_dynamics(
    double mass, const double * gravity, double timestep, int nfeet, const double * X, const double * U, const unsigned char * contact,
    const double * contact_pos, int batch, double * Xnext, double * A, double * B, int device_id)
  {
    if (!gravity || !X || !U || !contact || !contact_pos || !Xnext || nfeet < 1 || batch < 1 || !(mass > 0.0) || !(timestep > 0.0))
      return fail(SMPC_ERR_INVALID, "invalid argument");
    if (device_count() <= 0)
      return fail(SMPC_ERR_NO_DEVICE, "no HIP device visible: the centroidal dynamics has no CPU path");
    return guarded([&] {
      set_device(device_id);
      stream_t st = stream_create();
      const size_t nu = 3 * (size_t)nfeet, nb = (size_t)batch;
      const size_t n_in = nb * 9 + nb * nu + nb * nfeet * 3, n_out = nb * 9 + nb * 81 + nb * 9 * nu;
      double * d = (double *)dev_alloc((n_in + n_out) * sizeof(double) + nb * nfeet);
      double *dX = d, *dU = dX + nb * 9, *dP = dU + nb * nu, *dXn = dP + nb * nfeet * 3, *dA = dXn + nb * 9, *dB = dA + nb * 81;
      unsigned char * dC = reinterpret_cast<unsigned char *>(dB + nb * 9 * nu);
      h2d(dX, X, nb * 9 * sizeof(double), st);
      h2d(dU, U, nb * nu * sizeof(double), st);
      h2d(dP, contact_pos, nb * nfeet * 3 * sizeof(double), st);
      h2d(dC, contact, nb * nfeet, st);
      CentroidalArgs ca;
      ca.mass = mass;
      ca.dt = timestep;
      for (int i = 0; i < 3; i++)
        ca.g[i] = gravity[i];
      ca.nf = nfeet;
      ca.batch = batch;
      ca.X = dX;
      ca.U = dU;
      ca.pos = dP;
      ca.contact = dC;
      ca.Xn = dXn;
      ca.A = A ? dA : nullptr;
      ca.Bm = B ? dB : nullptr;
      launch<CentroidalArgs, centroidal_body, 64>((batch + 63) / 64, st, ca);
      d2h(Xnext, dXn, nb * 9 * sizeof(double), st);
      if (A)
        d2h(A, dA, nb * 81 * sizeof(double), st);
      if (B)
        d2h(B, dB, nb * 9 * nu * sizeof(double), st);
      stream_sync(st);
      dev_free(d);
      stream_destroy(st);
    });
  }

  // ---- whole-body inverse-dynamics QP (smpc_id.h) ----
  int smpc_id_create(const smpc_robot_model * robot, const smpc_id_settings * c, int batch, int device_id, smpc_id_handle ** out)
  {
    if (!robot || !c || !out)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (device_count() <= 0)
      return fail(SMPC_ERR_NO_DEVICE, "no HIP device visible: the inverse-dynamics engine has no CPU path");
    if (!c->effort_limit || !c->velocity_limit || !c->q_min || !c->q_max)
      return fail(SMPC_ERR_INVALID, "effort, velocity and position limits of the actuated joints are required");
    HostIdSettings hs;
    hs.dev.friction_coefficient = c->friction_coefficient;
    hs.dev.ratio_max = c->contact_weight_ratio_max;
    hs.dev.ratio_min = c->contact_weight_ratio_min;
    hs.dev.kp_base = c->kp_base;
    hs.dev.kp_posture = c->kp_posture;
    hs.dev.kp_contact = c->kp_contact;
    hs.dev.w_base = c->w_base;
    hs.dev.w_posture = c->w_posture;
    hs.dev.w_contact_motion = c->w_contact_motion;
    hs.dev.w_contact_force = c->w_contact_force;
    hs.dev.contact_motion_equality = c->contact_motion_equality;
    hs.dev.control_dt = c->control_dt;
    hs.dev.admm_iters = c->admm_iters > 0 ? c->admm_iters : 400;
    hs.dev.rho = c->admm_rho > 0 ? c->admm_rho : 0.1;
    hs.dev.sigma = c->admm_sigma > 0 ? c->admm_sigma : 1e-6;
    hs.dev.alpha = c->admm_alpha > 0 ? c->admm_alpha : 1.6;
    hs.dev.admm_tol = c->admm_tol == 0.0 ? 1e-7 : c->admm_tol;
    hs.dev.centroidal = c->centroidal != 0;
    hs.dev.pad_ = 0;
    hs.dev.base_as_coded = c->base_reference_as_coded != 0;
    hs.dev.tsid_bounds = c->tsid_joint_bounds != 0;
    hs.dev.kp_com = c->kp_com;
    hs.dev.kp_feet_tracking = c->kp_feet_tracking;
    hs.dev.w_com = c->w_com;
    hs.dev.w_feet_tracking = c->w_feet_tracking;
    if (c->centroidal && !(c->kp_com >= 0.0 && c->kp_feet_tracking >= 0.0))
      return fail(SMPC_ERR_INVALID, "task gains must not be negative");
    if (!(c->kp_base >= 0.0 && c->kp_posture >= 0.0 && c->kp_contact >= 0.0))
      return fail(SMPC_ERR_INVALID, "task gains must not be negative");
    const int na = robot->nv - 6;
    hs.tau_max.assign(c->effort_limit, c->effort_limit + na);
    hs.v_max.assign(c->velocity_limit, c->velocity_limit + na);
    hs.q_min.assign(c->q_min, c->q_min + na);
    hs.q_max.assign(c->q_max, c->q_max + na);
    const bool quad = c->force_size == 6;
    if (c->force_size != 0 && c->force_size != 3 && c->force_size != 6)
      return fail(SMPC_ERR_INVALID, "force size must be 3 (point feet) or 6 (flat feet)");
    if (quad)
    {
      if (!c->quad_contact_points)
        return fail(SMPC_ERR_INVALID, "flat feet need the four corners of every sole (quad_contact_points, [nfeet][4][3])");
      hs.quad_points.assign(c->quad_contact_points, c->quad_contact_points + (size_t)robot->nfeet * 12);
    }
    return guarded([&] {
#if defined(SMPC_KINO_ONLY) && !defined(SMPC_WITH_ID) // (tools/variant_build.sh <name> -DSMPC_WITH_ID: the ID engines too)
      throw std::runtime_error("SMPC_KINO_ONLY experiment build");
#endif
      if (!quad && robot->njoints == FullGo2::NJ && robot->nfeet == FullGo2::NF)
        *out = reinterpret_cast<smpc_id_handle *>(static_cast<IdEngineBase *>(new IdEngine<FullGo2>(robot, hs, batch, device_id)));
      else if (quad && robot->njoints == FullTalos::NJ && robot->nfeet == FullTalos::NF)
        *out = reinterpret_cast<smpc_id_handle *>(static_cast<IdEngineBase *>(new IdEngine<FullTalos>(robot, hs, batch, device_id)));
      else
        throw std::runtime_error("the inverse-dynamics engine is instantiated for 13 joints / 4 point feet and for 23 joints / 2 flat feet");
    });
  }
  void smpc_id_destroy(smpc_id_handle * h) { delete reinterpret_cast<IdEngineBase *>(h); }
  int smpc_id_set_target(smpc_id_handle * h, int instance, const double * q, const double * v, const double * a, const uint8_t * contact, const double * f)
  {
    if (!h || !q || !v || !a || !contact || !f)
      return fail(SMPC_ERR_INVALID, "null argument");
    IdEngineBase * e = reinterpret_cast<IdEngineBase *>(h);
    unsigned mask = 0;
    for (int k = 0; k < e->nf; k++)
      mask |= contact[k] ? (1u << k) : 0u;
    return guarded([&] { e->set_target(instance, q, v, a, mask, f); });
  }
  int smpc_id_set_targets(smpc_id_handle * h, const double * Q, const double * V, const double * A, const uint8_t * contact, const double * F)
  {
    if (!h || !Q || !V || !A || !contact || !F)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] { reinterpret_cast<IdEngineBase *>(h)->set_targets(Q, V, A, contact, F); });
  }
  int smpc_id_set_target_centroidal(smpc_id_handle * h, int instance, const double * com, const double * vcom, const double * feet_p,
                                    const double * feet_v, const uint8_t * contact, const double * f)
  {
    if (!h || !com || !vcom || !feet_p || !feet_v || !contact || !f)
      return fail(SMPC_ERR_INVALID, "null argument");
    IdEngineBase * e = reinterpret_cast<IdEngineBase *>(h);
    unsigned mask = 0;
    for (int k = 0; k < e->nf; k++)
      mask |= contact[k] ? (1u << k) : 0u;
    return guarded([&] { e->set_target_centroidal(instance, com, vcom, feet_p, feet_v, mask, f); });
  }
  int smpc_id_set_targets_centroidal(smpc_id_handle * h, const double * COM, const double * VCOM, const double * FEET_P, const double * FEET_V,
                                     const uint8_t * contact, const double * F)
  {
    if (!h || !COM || !VCOM || !FEET_P || !FEET_V || !contact || !F)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] { reinterpret_cast<IdEngineBase *>(h)->set_targets_centroidal(COM, VCOM, FEET_P, FEET_V, contact, F); });
  }
  int smpc_id_solve(smpc_id_handle * h, const double * X, double * tau, double * a, double * f, double * resid)
  {
    if (!h || !X || !tau)
      return fail(SMPC_ERR_INVALID, "null argument");
    IdEngineBase * e = reinterpret_cast<IdEngineBase *>(h);
    return guarded([&] {
      std::vector<double> ta, tf;
      if (!a)
        ta.resize((size_t)e->B * e->nv);
      if (!f)
        tf.resize((size_t)e->B * 3 * e->nf);
      e->solve(X, tau, a ? a : ta.data(), f ? f : tf.data(), resid);
    });
  }
  int smpc_id_solve_device(smpc_id_handle * h, const double * X_device, double * tau_device)
  {
    if (!h || !X_device)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] { reinterpret_cast<IdEngineBase *>(h)->solve_device(X_device, tau_device); });
  }
  int smpc_id_wait(smpc_id_handle * h)
  {
    if (!h)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] { reinterpret_cast<IdEngineBase *>(h)->wait(); });
  }
  int smpc_id_get_resid(smpc_id_handle * h, double * resid)
  {
    if (!h || !resid)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] { reinterpret_cast<IdEngineBase *>(h)->get_resid(resid); });
  }
  int smpc_id_reset(smpc_id_handle * h, int instance)
  {
    if (!h)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] { reinterpret_cast<IdEngineBase *>(h)->reset(instance); });
  }
  const double * smpc_id_get_tau_device(smpc_id_handle * h) { return h ? reinterpret_cast<IdEngineBase *>(h)->tau_device() : nullptr; }
  int smpc_id_set_targets_from_mpc(smpc_id_handle * id, smpc_handle * mpc, double delay, int knots)
  {
    if (!id || !mpc)
      return fail(SMPC_ERR_INVALID, "null argument");
    IdEngineBase * e = reinterpret_cast<IdEngineBase *>(id);
    if (mpc->cent)
    { // centroidal MPC -> CentroidalID (examples/talos_centroidal.py:218-243)
      double *com, *vcom, *fp, *fv, *x, *a, *f;
      e->centroidal_target_buffers(&com, &vcom, &fp, &fv);
      if (!com)
        return fail(SMPC_ERR_INVALID, "a centroidal MPC handle feeds a CentroidalID controller");
      if (e->B != mpc->cent->B || e->nf != mpc->cent->nf || e->nv != mpc->cent->nv_mb || e->nfw * e->nf != mpc->cent->nu)
        return fail(SMPC_ERR_INVALID, "the controller and the MPC must hold the same batch of the same robot");
      return guarded([&] {
        e->target_buffers(&x, &a, &f);
        e->set_mask_all(mpc->cent->contact_mask(0));
        const bool shared = e->solve_stream() == mpc->cent->stream; // (one in-order queue: nothing to order)
        if (!shared)
          e->wait();
        mpc->cent->interpolate_device_id(delay, knots, com, vcom, fp, fv, f);
        if (!shared)
          mpc->cent->wait_stream(e->solve_stream());
      });
    }
    {
      double *com, *vcom, *fp, *fv;
      e->centroidal_target_buffers(&com, &vcom, &fp, &fv);
      if (com)
        return fail(SMPC_ERR_INVALID, "a kinodynamics MPC handle feeds a KinodynamicsID controller");
    }
    if (mpc->full)
    { // kinodynamics OCP of a robot with flat feet (dense engine) or a full-dynamics OCP -> KinodynamicsID: states, accelerations, contact forces
      FullEngineBase & fe = *mpc->full;
      if (e->B != fe.B || e->nq != fe.dims[0] || e->nv != fe.dims[1] || e->nf != fe.dims[6] || e->nfw != fe.force_size)
        return fail(SMPC_ERR_INVALID, "the controller and the MPC must hold the same batch of the same robot");
      return guarded([&] {
        double *x, *a, *f;
        e->target_buffers(&x, &a, &f);
        e->set_mask_all(fe.contact_mask(0));
        const bool shared = e->solve_stream() == fe.stream;
        if (!shared)
          e->wait();
        fe.interpolate_device(delay, knots, x, a, f);
        if (!shared)
          fe.wait_stream(e->solve_stream());
      });
    }
    if (!mpc->eng)
      return fail(SMPC_ERR_INVALID, "smpc_id_set_targets_from_mpc needs an MPC handle");
    if (e->B != mpc->eng->B || e->nq != DimsGo2::NQ || e->nv != DimsGo2::NV)
      return fail(SMPC_ERR_INVALID, "the controller and the MPC must hold the same batch of the same robot");
    return guarded([&] {
      double *x, *a, *f;
      e->target_buffers(&x, &a, &f);
      e->set_mask_all(mpc->eng->contact_mask(0));
      const bool shared = e->solve_stream() == mpc->eng->stream; // (one in-order queue: nothing to order)
      if (!shared)
        e->wait(); // (the previous solve has read its targets)
      mpc->eng->interpolate_device(delay, knots, x, a, f);
      if (!shared)
        mpc->eng->wait_stream(e->solve_stream()); // the next solve starts after the targets are written
    });
  }
  int smpc_id_share_stream(smpc_id_handle * id, smpc_handle * mpc)
  {
    if (!id)
      return fail(SMPC_ERR_INVALID, "null argument");
    IdEngineBase * e = reinterpret_cast<IdEngineBase *>(id);
    if (!mpc)
      return guarded([&] { e->adopt_stream(e->solve_stream(), true); });
    if ((mpc->full ? mpc->full->device_id : (mpc->cent ? mpc->cent->device_id : mpc->eng->device_id)) != e->device())
      return fail(SMPC_ERR_INVALID, "smpc_id_share_stream: the controller and the MPC handle live on different devices");
    return guarded([&] { e->adopt_stream(mpc->full ? mpc->full->stream : (mpc->cent ? mpc->cent->stream : mpc->eng->stream), false); });
  }
  int smpc_sim_step_device(smpc_handle * h, double * X_device, const double * tau_device, const uint8_t * contact, const double * Kp, const double * Kd, double dt)
  {
    if (!h || !X_device || !tau_device || !contact)
      return fail(SMPC_ERR_INVALID, "null argument");
    if (!(dt > 0.0))
      return fail(SMPC_ERR_INVALID, "dt must be positive");
    if (h->full)
    { // full-dynamics handle: its own robot and contact model (Kp / Kd: force_size entries)
      unsigned mask = 0;
      for (int k = 0; k < h->full->dims[6]; k++)
        mask |= contact[k] ? (1u << k) : 0u;
      return guarded([&] { h->full->sim_step_device(X_device, tau_device, mask, Kp, Kd, dt); });
    }
    if (!h->eng)
      return fail(SMPC_ERR_INVALID, "smpc_sim_step_device needs a kinodynamics or a full-dynamics handle (they carry the multibody model)");
    unsigned mask = 0;
    for (int k = 0; k < DimsGo2::NF; k++)
      mask |= contact[k] ? (1u << k) : 0u;
    return guarded([&] { h->eng->sim_step_device(X_device, tau_device, mask, Kp, Kd, dt); });
  }
  double * smpc_id_get_x_device(smpc_id_handle * h) { return h ? reinterpret_cast<IdEngineBase *>(h)->x_device() : nullptr; }
  int smpc_id_debug_get(smpc_id_handle * h, int what, double * out)
  {
    if (!h || !out)
      return fail(SMPC_ERR_INVALID, "null argument");
    return guarded([&] { reinterpret_cast<IdEngineBase *>(h)->debug_get(what, out); });
  }
}
