// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_capi.cpp: extern "C" surface of the CPU oracle, loaded with ctypes by tests/ and by
// bench.py's cpu_baseline leg.  Nothing in simple-mpc_amd/ links or loads this library.
#include "../include/smpc_robots_builtin.h"
#include "orc_mpc_cent.hpp"
#include "orc_fulldyn.hpp"
#include "orc_id.hpp"
#include <chrono>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;
typedef BatchMPCT<FullModel, FullSettings> BatchMPCFull;
// createProblem(x0, T, force_size, gravity, terminal_constraint): the last argument, for the next orc_mpc_create / orc_fmpc_create
static bool g_terminal_constraint = false;

namespace
{
  Mat mat_from(const double * p, int r, int c)
  {
    Mat m(r, c);
    std::memcpy(m.a.data(), p, sizeof(double) * (size_t)r * c);
    return m;
  }
  void mat_to(const Mat & m, double * p) { std::memcpy(p, m.a.data(), sizeof(double) * m.a.size()); }
  void vec_to(const Vec & v, double * p) { std::memcpy(p, v.data(), sizeof(double) * v.size()); }

  StageRef make_ref(const KinoModel & md, unsigned mask, const double * u_ref, const double * x_tgt, const double * foot_ref)
  {
    StageRef r;
    r.mask = mask & 0xFFu;        // bits 0-7: feet in contact
    r.land = (mask >> 8) & 0xFFu; // bits 8-15: feet that land at this stage (land_cstr rows)
    r.u_ref.assign(u_ref, u_ref + md.nu);
    r.x_tgt.assign(x_tgt, x_tgt + md.nx);
    r.foot_ref.resize(md.nf);
    for (int f = 0; f < md.nf; f++)
      r.foot_ref[f] = v3(foot_ref[3 * f], foot_ref[3 * f + 1], foot_ref[3 * f + 2]);
    return r;
  }
  // orc_mpc_get / orc_cmpc_get. what: 0 xs [B][H+1][nx], 1 us [B][H][nu], 2 K0 [B][nu][ndx], 3 vs [B][H][nc], 4 lams [B][H+1][ndx],
  //       5 foot refs [B][H][nf][3], 6 info [B][12], 7 xdot [B][H][2nv], 8 Ks [B][H][nu][ndx]
  template <class MPC>
  void mpc_get_impl(MPC * m, int what, double * out)
  {
    const auto & md = m->md;
    const int H = m->H;
    for (int b = 0; b < m->B; b++)
    {
      const SolverState & S = m->sol[b];
      switch (what)
      {
      case 0:
        for (int t = 0; t <= H; t++)
          vec_to(S.xs[t], out + ((size_t)b * (H + 1) + t) * md.nx);
        break;
      case 1:
        for (int t = 0; t < H; t++)
          vec_to(S.us[t], out + ((size_t)b * H + t) * md.nu);
        break;
      case 2:
        if (!S.Ks.empty())
          mat_to(S.Ks[0], out + (size_t)b * md.nu * md.ndx);
        break;
      case 3:
        for (int t = 0; t < H; t++)
          vec_to(S.vs[t], out + ((size_t)b * H + t) * md.nc);
        break;
      case 4:
        for (int t = 0; t <= H; t++)
          vec_to(S.lams[t], out + ((size_t)b * (H + 1) + t) * md.ndx);
        break;
      case 5:
        for (int t = 0; t < H; t++)
          for (int f = 0; f < md.nf; f++)
            for (int i = 0; i < 3; i++)
              out[(((size_t)b * H + t) * md.nf + f) * 3 + i] = m->ocp[b].stages[t].foot_ref[f][i];
        break;
      case 6: {
        const IterInfo & I = m->last_info[b];
        double * o = out + (size_t)b * 12;
        o[0] = I.phi0;
        o[1] = I.dphi0;
        o[2] = I.alpha;
        o[3] = I.phi_new;
        o[4] = I.prim_infeas;
        o[5] = I.dual_infeas;
        o[6] = I.ls_failed;
        o[7] = S.preg;
        o[8] = I.prim_new;
        o[9] = I.cost;
        o[10] = I.cost_new;
        o[11] = I.ls_index;
        break;
      }
      case 7:
        for (int t = 0; t < H && t < (int)S.xdot.size(); t++)
          vec_to(S.xdot[t], out + ((size_t)b * H + t) * 2 * md.nv);
        break;
      case 8:
        for (int t = 0; t < H && t < (int)S.Ks.size(); t++)
          mat_to(S.Ks[t], out + ((size_t)b * H + t) * md.nu * md.ndx);
        break;
      case 9: // terminal constraint: multipliers (3) | reference (3) | tau
        for (int i = 0; i < 3; i++)
        {
          out[(size_t)b * 7 + i] = i < (int)S.vN.size() ? S.vN[i] : 0.0;
          out[(size_t)b * 7 + 3 + i] = m->ocp[b].dcm_ref[i];
        }
        out[(size_t)b * 7 + 6] = m->ocp[b].dcm_tau;
        break;
      }
    }
  }
  // LQ knot (b, t) of the last iteration, packed A,B,Q,S,R,C,q,r,f,d (row-major)
  template <class MPC>
  int mpc_get_knot_impl(MPC * m, int b, int t, double * out)
  {
    if (b < 0 || b >= m->B || t < 0 || t >= (int)m->last_knots[b].size())
      return -1;
    const Knot & k = m->last_knots[b][t];
    double * o = out;
    for (const Mat * M : {&k.A, &k.B, &k.Q, &k.S, &k.R, &k.C})
    {
      mat_to(*M, o);
      o += M->a.size();
    }
    for (const Vec * v : {&k.q, &k.r, &k.f, &k.d})
    {
      vec_to(*v, o);
      o += v->size();
    }
    return (int)(o - out);
  }
  // ProxDDP on an H-stage problem with per-stage references, run to convergence with the stopping rules of the cold solve (orc_mpc.hpp) --
  // for the checks of the converged answer against an independent NLP solver (tests/test_oracle_vs_scipy.py).
  // trace: [iter][6] = prim_infeas, dual_infeas, cost, phi0, alpha, ls_failed.  Returns the iteration count.
  template <class Model, class MakeRef>
  int generic_solve(
    const Model & md, const smpc_robot_model * rm, int H, MakeRef make, const double * x_tgt_term, const double * x0, const double * u0, int max_iter,
    double tol, double mu, double * trace, double * xs, double * us, double * vs, double * lams)
  {
    Rigid R(rm);
    OcpInstance o;
    for (int t = 0; t < H; t++)
      o.stages.push_back(make(t));
    o.x_tgt_term.assign(x_tgt_term, x_tgt_term + md.nx);
    SolverState S;
    S.xs.assign(H + 1, Vec(x0, x0 + md.nx));
    S.us.assign(H, Vec(u0, u0 + md.nu));
    S.vs.assign(H, Vec(md.nc, 0.0));
    S.lams.assign(H + 1, Vec(md.ndx, 0.0));
    ProxDDPT<Model> solver(md, mu);
    std::vector<Vec> vs_e = S.vs, lams_e = S.lams;
    // outer loop of the augmented Lagrangian: whenever the inner problem (fixed multiplier centres) has converged -- dual residual below the
    // inner tolerance, or the merit stalls at FP64 resolution -- the centres move to the current multipliers; the run ends when primal AND dual
    // residuals are below `tol`, or when two refreshes in a row bring nothing
    const double inner_tol = std::fmax(tol, 1e-7);
    int it = 0, idle_refresh = 0;
    for (; it < max_iter; it++)
    {
      const IterInfo info = solver.iterate(R, o, S, vs_e, lams_e);
      double * tr = trace + 6 * it;
      tr[0] = info.prim_infeas;
      tr[1] = info.dual_infeas;
      tr[2] = info.cost;
      tr[3] = info.phi0;
      tr[4] = info.alpha;
      tr[5] = info.ls_failed;
      if (std::fmax(info.prim_infeas, info.dual_infeas) <= tol)
      {
        it++;
        break;
      }
      const bool stall = std::fabs(info.dphi0) <= SolverConsts::STALL_REL * std::fmax(1.0, std::fabs(info.phi0));
      if (info.dual_infeas <= inner_tol || stall)
      {
        vs_e = S.vs;
        lams_e = S.lams;
        idle_refresh = stall ? idle_refresh + 1 : 0;
        if (idle_refresh >= 3)
        {
          it++;
          break;
        }
      }
      else
        idle_refresh = 0;
    }
    for (int t = 0; t <= H; t++)
      std::memcpy(xs + (size_t)t * md.nx, S.xs[t].data(), sizeof(double) * md.nx);
    for (int t = 0; t < H; t++)
    {
      std::memcpy(us + (size_t)t * md.nu, S.us[t].data(), sizeof(double) * md.nu);
      if (vs)
        std::memcpy(vs + (size_t)t * md.nc, S.vs[t].data(), sizeof(double) * md.nc);
    }
    if (lams)
      for (int t = 0; t <= H; t++)
        std::memcpy(lams + (size_t)t * md.ndx, S.lams[t].data(), sizeof(double) * md.ndx);
    return it;
  }
  // kind of every constraint row of a stage (0 absent, 1 equality, 2 box [lo, hi], 3 <= 0) and its bounds
  template <class Model>
  void generic_row_kinds(const Model & md, const StageRef & r, int * kind, double * lo, double * hi)
  {
    for (int i = 0; i < md.nc; i++)
    {
      kind[i] = md.row_kind(r, i);
      lo[i] = md.row_lo_v(i);
      hi[i] = md.row_hi_v(i);
    }
  }
} // namespace

extern "C"
{
  const smpc_robot_model * orc_builtin_robot(const char * name)
  {
    if (!std::strcmp(name, "go2_like"))
      return &SMPC_ROBOT_GO2_LIKE;
    if (!std::strcmp(name, "biped_like"))
      return &SMPC_ROBOT_BIPED_LIKE;
    if (!std::strcmp(name, "talos_like"))
      return &SMPC_ROBOT_TALOS_LIKE;
    return nullptr;
  }
  int orc_robot_dims(const smpc_robot_model * m, int * out) // nq nv nfeet njoints
  {
    out[0] = m->nq;
    out[1] = m->nv;
    out[2] = m->nfeet;
    out[3] = m->njoints;
    return 0;
  }
  void orc_robot_info(const smpc_robot_model * m, double * q_ref, double * q_lo, double * q_hi, double * mass)
  {
    for (int i = 0; i < m->nq; i++)
      q_ref[i] = m->q_ref[i];
    for (int i = 0; i < m->nv - 6; i++)
    {
      q_lo[i] = m->q_lo[i];
      q_hi[i] = m->q_hi[i];
    }
    *mass = m->total_mass;
  }

  // ---- Lie group helpers ----
  void orc_x_integrate(const smpc_robot_model * m, const double * x, const double * dx, double * out)
  {
    x_integrate(m->nq, m->nv, x, dx, out);
  }
  void orc_x_difference(const smpc_robot_model * m, const double * x0, const double * x1, double * out)
  {
    x_difference(m->nq, m->nv, x0, x1, out);
  }
  void orc_exp6(const double * nu, double * R9, double * p3)
  {
    SE3 M = exp6(nu);
    std::memcpy(R9, M.R.m, 9 * sizeof(double));
    std::memcpy(p3, M.p.x, 3 * sizeof(double));
  }
  void orc_log6(const double * R9, const double * p3, double * nu)
  {
    SE3 M;
    std::memcpy(M.R.m, R9, 9 * sizeof(double));
    std::memcpy(M.p.x, p3, 3 * sizeof(double));
    log6(M, nu);
  }
  void orc_Jexp6(const double * nu, double * J36) { mat_to(Jexp6(nu), J36); }
  void orc_Jlog6_of_exp(const double * nu, double * J36) { mat_to(Jlog6(exp6(nu)), J36); }

  // ---- kinodynamics model ----
  void * orc_kino_create(
    const smpc_robot_model * m, double dt, const double * w_x, const double * w_u, const double * w_frame,
    const double * w_cent, const double * w_centder, const double * qmin, const double * qmax, const double * gravity,
    int kinematics_limits)
  {
    KinoSettings s;
    const int ndx = 2 * m->nv, nu = m->nv - 6 + 3 * m->nfeet;
    s.timestep = dt;
    s.w_x = mat_from(w_x, ndx, ndx);
    s.w_u = mat_from(w_u, nu, nu);
    s.w_frame = mat_from(w_frame, 3, 3);
    s.w_cent = mat_from(w_cent, 6, 6);
    s.w_centder = mat_from(w_centder, 6, 6);
    s.qmin.assign(qmin, qmin + m->nv - 6);
    s.qmax.assign(qmax, qmax + m->nv - 6);
    for (int i = 0; i < 3; i++)
      s.gravity[i] = gravity[i];
    s.kinematics_limits = kinematics_limits != 0;
    return new KinoModel(m, s);
  }
  // force_cone rows of the kinodynamics stage (CentroidalFrictionConeResidual per foot in contact) and the friction coefficient
  void orc_kino_set_force_cone(void * h, int on, double mu)
  {
    KinoModel * md = (KinoModel *)h;
    md->s.force_cone = on != 0;
    md->s.mu = mu;
    md->configure();
  }
  void orc_kino_set_land_cstr(void * h, int on)
  {
    KinoModel * md = (KinoModel *)h;
    md->s.land_cstr = on != 0;
    md->configure();
  }
  // 6-D feet (force_size 6): control = [(f, tau) per foot ; joint accelerations], 6 x 6 foot-pose weight, sole half-length / half-width of
  // the wrench cone (reference KinodynamicsSettings force_size / Lfoot / Wfoot, include/simple-mpc/kinodynamics.hpp:24-51)
  void orc_kino_set_force_size(void * h, int fs, const double * w_u, const double * w_frame, double mu, double Lfoot, double Wfoot)
  {
    KinoModel * md = (KinoModel *)h;
    md->fs = fs;
    md->s.force_size = fs;
    md->nu = md->nv - 6 + fs * md->nf;
    md->s.w_u = mat_from(w_u, md->nu, md->nu);
    md->s.w_frame = mat_from(w_frame, fs, fs);
    md->s.mu = mu;
    md->s.Lfoot = Lfoot;
    md->s.Wfoot = Wfoot;
    md->Acone = wrench_cone_matrix(mu, Lfoot, Wfoot);
    md->configure();
  }
  void orc_set_fold_u_rows(int on) { fold_u_rows() = on != 0; }
  void orc_kino_destroy(void * h) { delete (KinoModel *)h; }
  void orc_kino_dims(void * h, int * out) // nx ndx nu nc nf
  {
    KinoModel * md = (KinoModel *)h;
    out[0] = md->nx;
    out[1] = md->ndx;
    out[2] = md->nu;
    out[3] = md->nc;
    out[4] = md->nf;
  }
  void orc_kino_eval(
    void * h, unsigned mask, const double * u_ref, const double * x_tgt, const double * foot_ref, const double * x,
    const double * u, double * xnext, double * xdot, double * cost, double * c)
  {
    KinoModel * md = (KinoModel *)h;
    Rigid R(md->M);
    StageRef r = make_ref(*md, mask, u_ref, x_tgt, foot_ref);
    StageEval o;
    md->eval(R, r, x, u, o);
    vec_to(o.xnext, xnext);
    vec_to(o.xdot, xdot);
    *cost = o.cost;
    vec_to(o.c, c);
  }
  void orc_kino_deriv(
    void * h, unsigned mask, const double * u_ref, const double * x_tgt, const double * foot_ref, const double * x,
    const double * u, double * A, double * B, double * lx, double * lu, double * Lxx, double * Lxu, double * Luu,
    double * Cx, double * Cu)
  {
    KinoModel * md = (KinoModel *)h;
    Rigid R(md->M);
    StageRef r = make_ref(*md, mask, u_ref, x_tgt, foot_ref);
    StageDer o;
    md->deriv(R, r, x, u, o);
    mat_to(o.A, A);
    mat_to(o.B, B);
    vec_to(o.lx, lx);
    vec_to(o.lu, lu);
    mat_to(o.Lxx, Lxx);
    mat_to(o.Lxu, Lxu);
    mat_to(o.Luu, Luu);
    mat_to(o.Cx, Cx);
    mat_to(o.Cu, Cu);
  }
  // DCM terminal constraint value c (3) and Jacobian C (3 x ndx)
  void orc_kino_term_cstr(void * h, const double * x, const double * ref, double tau, double * c, double * C)
  {
    KinoModel * md = (KinoModel *)h;
    Rigid R(md->M);
    Mat Cm;
    md->term_cstr(R, x, ref, tau, c, &Cm);
    mat_to(Cm, C);
  }
  void orc_kino_term(void * h, const double * x_tgt, const double * x, double * cost, double * lx, double * Lxx)
  {
    KinoModel * md = (KinoModel *)h;
    Rigid R(md->M);
    Vec xt(x_tgt, x_tgt + md->nx);
    *cost = md->term_eval(R, xt, x);
    Vec l;
    Mat L;
    md->term_deriv(R, xt, x, l, L);
    vec_to(l, lx);
    mat_to(L, Lxx);
  }
  // centroidal quantities for physics identities: hg(6), Ag(6 x nv), dAg*v (6), com(3), feet(nf*3)
  void orc_centroidal(
    const smpc_robot_model * m, const double * x, double * hg, double * Ag, double * dAgv, double * com, double * feet)
  {
    Rigid R(m);
    R.fk(x);
    R.velocities(x + m->nq);
    R.forces(x + m->nq, nullptr);
    vec_to(sv_vec(R.hg()), hg);
    mat_to(R.Ag(), Ag);
    vec_to(sv_vec(R.dAg_v()), dAgv);
    for (int i = 0; i < 3; i++)
      com[i] = R.com[i];
    for (int f = 0; f < m->nfeet; f++)
      for (int i = 0; i < 3; i++)
        feet[3 * f + i] = R.foot_p[f][i];
  }

  // ---- constrained forward dynamics of the full-dynamics model (orc_full.hpp) ----
  // x = [q; v], tau (nv - 6), contact mask, Kp / Kd (3) -> a (nv), lam (3 per foot in contact, contact frame),
  // M (nv x nv), nle (nv), J (3 nc x nv), gamma (3 nc), tau_rnea (nv) = RNEA(q, v, a);  returns the proximal iteration count
  int orc_full_forward_dynamics(
    const smpc_robot_model * m, const double * x, const double * tau, unsigned mask, int fs, const double * Kp, const double * Kd,
    double * a, double * lam, double * Mq, double * nle, double * J, double * gamma, double * tau_rnea)
  {
    ConstraintDynamics cd(m);
    cd.fs = fs;
    for (int i = 0; i < fs; i++)
    {
      cd.Kp[i] = Kp[i];
      cd.Kd[i] = Kd[i];
    }
    cd.compute(x, x + m->nq, tau, mask);
    vec_to(cd.a, a);
    vec_to(cd.lam, lam);
    mat_to(cd.Mq, Mq);
    vec_to(cd.nle, nle);
    mat_to(cd.Jc, J);
    vec_to(cd.gamma, gamma);
    vec_to(cd.rnea(x + m->nq, cd.a.data()), tau_rnea);
    return cd.prox_iters;
  }
  // same inputs + proximal accuracy / iteration cap (<= 0: the reference's settings) -> a, lam and their derivatives wrt
  // the tangent of q, v and tau (row-major; lam rows = 3 per foot in contact), and the RNEA partials at the solution
  int orc_full_dynamics_derivatives(
    const smpc_robot_model * m, const double * x, const double * tau, unsigned mask, int fs, const double * Kp, const double * Kd,
    double prox_accuracy, int prox_max_iter, double * a, double * lam, double * da_dq, double * da_dv, double * da_dtau,
    double * dlam_dq, double * dlam_dv, double * dlam_dtau, double * dtau_dq, double * dtau_dv)
  {
    ConstraintDynamics cd(m);
    cd.fs = fs;
    for (int i = 0; i < fs; i++)
    {
      cd.Kp[i] = Kp[i];
      cd.Kd[i] = Kd[i];
    }
    if (prox_accuracy > 0)
      cd.prox_accuracy = prox_accuracy;
    if (prox_max_iter > 0)
      cd.prox_max_iter = prox_max_iter;
    cd.compute(x, x + m->nq, tau, mask);
    cd.derivatives(x + m->nq);
    vec_to(cd.a, a);
    vec_to(cd.lam, lam);
    mat_to(cd.da_dq, da_dq);
    mat_to(cd.da_dv, da_dv);
    mat_to(cd.da_dtau, da_dtau);
    mat_to(cd.dlam_dq, dlam_dq);
    mat_to(cd.dlam_dv, dlam_dv);
    mat_to(cd.dlam_dtau, dlam_dtau);
    mat_to(cd.dtau_dq, dtau_dq);
    mat_to(cd.dtau_dv, dtau_dv);
    return cd.prox_iters;
  }
  // ---- full-dynamics stage model (orc_fulldyn.hpp) ----
  void * orc_full_create(
    const smpc_robot_model * m, double dt, const double * w_x, const double * w_u, const double * w_cent, const double * w_forces,
    const double * w_frame, const double * gravity, const double * Kp, const double * Kd, const double * umin,
    const double * umax, const double * qmin, const double * qmax, int torque_limits, int kinematics_limits, int force_size,
    int force_cone, double mu, double Lfoot, double Wfoot)
  {
    FullSettings s;
    const int ndx = 2 * m->nv, nu = m->nv - 6, fs = force_size;
    s.timestep = dt;
    s.force_size = fs;
    s.force_cone = force_cone != 0;
    s.mu = mu;
    s.Lfoot = Lfoot;
    s.Wfoot = Wfoot;
    s.w_x = mat_from(w_x, ndx, ndx);
    s.w_u = mat_from(w_u, nu, nu);
    s.w_cent = mat_from(w_cent, 6, 6);
    s.w_forces = mat_from(w_forces, fs, fs);
    s.w_frame = mat_from(w_frame, fs, fs);
    for (int i = 0; i < 3; i++)
      s.gravity[i] = gravity[i];
    for (int i = 0; i < fs; i++)
    {
      s.Kp[i] = Kp[i];
      s.Kd[i] = Kd[i];
    }
    s.umin.assign(umin, umin + nu);
    s.umax.assign(umax, umax + nu);
    s.qmin.assign(qmin, qmin + nu);
    s.qmax.assign(qmax, qmax + nu);
    s.torque_limits = torque_limits != 0;
    s.kinematics_limits = kinematics_limits != 0;
    return new FullModel(m, s);
  }
  void orc_full_destroy(void * h) { delete (FullModel *)h; }
  // land_cstr rows of the full-dynamics stage (frame velocity / height of a landing foot): rebuilds the model with the switch
  void orc_full_set_land_cstr(void * h, int on)
  {
    FullModel * md = (FullModel *)h;
    FullSettings s = md->s;
    s.land_cstr = on != 0;
    *md = FullModel(md->M, s);
  }
  void orc_full_dims(void * h, int * out) // nx ndx nu nc nf
  {
    FullModel * md = (FullModel *)h;
    out[0] = md->nx;
    out[1] = md->ndx;
    out[2] = md->nu;
    out[3] = md->nc;
    out[4] = md->nf;
  }
  static StageRef full_ref(const FullModel & md, unsigned mask, const double * u_ref, const double * x_tgt, const double * foot_ref)
  {
    StageRef r;
    r.mask = mask & 0xFFu;        // bits 0-7: feet in contact ; bits 8-15: feet that land at this stage (as make_ref)
    r.land = (mask >> 8) & 0xFFu;
    r.u_ref.assign(u_ref, u_ref + md.n_uref()); // [control reference ; force reference per foot]
    r.x_tgt.assign(x_tgt, x_tgt + md.nx);
    r.foot_ref.resize(md.nf);
    for (int f = 0; f < md.nf; f++)
      r.foot_ref[f] = v3(foot_ref[3 * f], foot_ref[3 * f + 1], foot_ref[3 * f + 2]);
    return r;
  }
  void orc_full_eval(
    void * h, unsigned mask, const double * u_ref, const double * x_tgt, const double * foot_ref, const double * x,
    const double * u, double * xnext, double * xdot, double * cost, double * c)
  {
    FullModel * md = (FullModel *)h;
    Rigid R(md->M);
    StageEval o;
    md->eval(R, full_ref(*md, mask, u_ref, x_tgt, foot_ref), x, u, o);
    vec_to(o.xnext, xnext);
    vec_to(o.xdot, xdot);
    *cost = o.cost;
    vec_to(o.c, c);
  }
  void orc_full_deriv(
    void * h, unsigned mask, const double * u_ref, const double * x_tgt, const double * foot_ref, const double * x,
    const double * u, double * A, double * B, double * lx, double * lu, double * Lxx, double * Lxu, double * Luu,
    double * Cx, double * Cu)
  {
    FullModel * md = (FullModel *)h;
    Rigid R(md->M);
    StageDer o;
    md->deriv(R, full_ref(*md, mask, u_ref, x_tgt, foot_ref), x, u, o);
    mat_to(o.A, A);
    mat_to(o.B, B);
    vec_to(o.lx, lx);
    vec_to(o.lu, lu);
    mat_to(o.Lxx, Lxx);
    mat_to(o.Lxu, Lxu);
    mat_to(o.Luu, Luu);
    mat_to(o.Cx, Cx);
    mat_to(o.Cu, Cu);
  }
  // terminal cost of the full-dynamics OCP (state + 10 x centroidal, orc_fulldyn.hpp: term_eval) -- what tests/test_oracle_vs_scipy.py adds to
  // the stage costs on the side of the independent solver
  double orc_full_term(void * h, const double * x_tgt, const double * x)
  {
    FullModel * md = (FullModel *)h;
    Rigid R(md->M);
    Vec xt(x_tgt, x_tgt + md->nx);
    return md->term_eval(R, xt, x);
  }
  // ProxDDP on an H-stage problem with the same references at every stage (masks per stage), from the constant guess
  // (x0, u0): up to max_iter iterations with the stopping rules of the cold solve (orc_mpc.hpp).  trace: [iter][6] =
  // prim_infeas, dual_infeas, cost, phi0, alpha, ls_failed.  Returns the iteration count.
  int orc_full_solve(
    void * h, int H, const unsigned * masks, const double * u_ref, const double * x_tgt, const double * foot_ref,
    const double * x0, const double * u0, int max_iter, double tol, double mu, double * trace, double * xs, double * us)
  {
    FullModel * md = (FullModel *)h;
    Rigid R(md->M);
    OcpInstance o;
    for (int t = 0; t < H; t++)
      o.stages.push_back(full_ref(*md, masks[t], u_ref, x_tgt, foot_ref));
    o.x_tgt_term.assign(x_tgt, x_tgt + md->nx);
    SolverState S;
    S.xs.assign(H + 1, Vec(x0, x0 + md->nx));
    S.us.assign(H, Vec(u0, u0 + md->nu));
    S.vs.assign(H, Vec(md->nc, 0.0));
    S.lams.assign(H + 1, Vec(md->ndx, 0.0));
    ProxDDPT<FullModel> solver(*md, mu);
    std::vector<Vec> vs_e = S.vs, lams_e = S.lams;
    int it = 0;
    for (; it < max_iter; it++)
    {
      const IterInfo info = solver.iterate(R, o, S, vs_e, lams_e);
      double * tr = trace + 6 * it;
      tr[0] = info.prim_infeas;
      tr[1] = info.dual_infeas;
      tr[2] = info.cost;
      tr[3] = info.phi0;
      tr[4] = info.alpha;
      tr[5] = info.ls_failed;
      if (std::fmax(info.prim_infeas, info.dual_infeas) <= tol)
      {
        it++;
        break;
      }
      if (std::fabs(info.dphi0) <= SolverConsts::STALL_REL * std::fmax(1.0, std::fabs(info.phi0)))
      {
        it++;
        break;
      }
      if (info.dual_infeas <= tol)
      {
        vs_e = S.vs;
        lams_e = S.lams;
      }
    }
    for (int t = 0; t <= H; t++)
      std::memcpy(xs + (size_t)t * md->nx, S.xs[t].data(), sizeof(double) * md->nx);
    for (int t = 0; t < H; t++)
      std::memcpy(us + (size_t)t * md->nu, S.us[t].data(), sizeof(double) * md->nu);
    return it;
  }
  // RNEA(q, v, a) with the gravity field (for the finite-difference checks of the partials)
  void orc_full_rnea(const smpc_robot_model * m, const double * x, const double * a, double * tau)
  {
    ConstraintDynamics cd(m);
    cd.R.fk(x);
    cd.R.velocities(x + m->nq);
    vec_to(cd.rnea(x + m->nq, a), tau);
  }

  // ---- proximal Riccati on packed knots (row-major, stage-major) ----
  void orc_riccati(
    int H, int ndx, int nu, int nc, double mu, const double * Q, const double * S, const double * Rm, const double * q,
    const double * r, const double * A, const double * B, const double * f, const double * C, const double * D,
    const double * d, const double * QN, const double * qN, double * dxs, double * dus, double * dvs, double * dlams,
    double * Ks)
  {
    std::vector<Knot> kn(H);
    for (int t = 0; t < H; t++)
    {
      kn[t].Q = mat_from(Q + (size_t)t * ndx * ndx, ndx, ndx);
      kn[t].S = mat_from(S + (size_t)t * ndx * nu, ndx, nu);
      kn[t].R = mat_from(Rm + (size_t)t * nu * nu, nu, nu);
      kn[t].A = mat_from(A + (size_t)t * ndx * ndx, ndx, ndx);
      kn[t].B = mat_from(B + (size_t)t * ndx * nu, ndx, nu);
      kn[t].C = mat_from(C + (size_t)t * nc * ndx, nc, ndx);
      kn[t].D = mat_from(D + (size_t)t * nc * nu, nc, nu);
      kn[t].q.assign(q + (size_t)t * ndx, q + (size_t)(t + 1) * ndx);
      kn[t].r.assign(r + (size_t)t * nu, r + (size_t)(t + 1) * nu);
      kn[t].f.assign(f + (size_t)t * ndx, f + (size_t)(t + 1) * ndx);
      kn[t].d.assign(d + (size_t)t * nc, d + (size_t)(t + 1) * nc);
    }
    Mat QNm = mat_from(QN, ndx, ndx);
    Vec qNv(qN, qN + ndx);
    std::vector<Vec> dx, du, dv, dl;
    std::vector<Mat> K;
    prox_riccati(kn, QNm, qNv, mu, dx, du, dv, dl, &K);
    for (int t = 0; t <= H; t++)
    {
      vec_to(dx[t], dxs + (size_t)t * ndx);
      vec_to(dl[t], dlams + (size_t)t * ndx);
    }
    for (int t = 0; t < H; t++)
    {
      vec_to(du[t], dus + (size_t)t * nu);
      vec_to(dv[t], dvs + (size_t)t * nc);
      mat_to(K[t], Ks + (size_t)t * nu * ndx);
    }
  }

  // ---- gait timer (integer KATs of tests/mpc.cpp:78-90) ----
  void * orc_timer_create(const unsigned char * cs, int n, int nf, int H)
  {
    CycleTimer * t = new CycleTimer();
    std::vector<std::vector<char>> v(n, std::vector<char>(nf));
    for (int i = 0; i < n; i++)
      for (int f = 0; f < nf; f++)
        v[i][f] = cs[i * nf + f];
    t->generate(v, H, nf);
    return t;
  }
  void orc_timer_destroy(void * h) { delete (CycleTimer *)h; }
  void orc_timer_recede(void * h) { ((CycleTimer *)h)->recede_cycle(); }
  int orc_timer_get(void * h, int foot, int which, int * out, int cap) // which 0 takeoff 1 land
  {
    CycleTimer * t = (CycleTimer *)h;
    const std::vector<int> & v = which ? t->land[foot] : t->takeoff[foot];
    for (int i = 0; i < (int)v.size() && i < cap; i++)
      out[i] = v[i];
    return (int)v.size();
  }
  void orc_bezier8(const double * p0, const double * p1, double apex, float t, double * out)
  {
    V3 r = bezier8(v3(p0[0], p0[1], p0[2]), v3(p1[0], p1[1], p1[2]), apex, t);
    for (int i = 0; i < 3; i++)
      out[i] = r[i];
  }

  // ---- batched MPC ----
  struct orc_mpc_settings
  {
    double swing_apex, support_force, TOL, mu_init, timestep;
    int max_iters, num_threads, T_fly, T_contact, T;
  };
  void * orc_mpc_create(void * kino, const orc_mpc_settings * s, int B, double gravity_arg)
  {
    KinoModel * md = (KinoModel *)kino;
    MPCSettings ms;
    ms.swing_apex = s->swing_apex;
    ms.support_force = s->support_force;
    ms.TOL = s->TOL;
    ms.mu_init = s->mu_init;
    ms.timestep = s->timestep;
    ms.max_iters = s->max_iters;
    ms.num_threads = s->num_threads;
    ms.T_fly = s->T_fly;
    ms.T_contact = s->T_contact;
    ms.T = s->T;
#ifdef _OPENMP
    if (s->num_threads > 0)
      omp_set_num_threads(s->num_threads);
#endif
    return new BatchMPC(md->M, md->s, ms, s->T, B, gravity_arg, g_terminal_constraint);
  }
  void orc_set_terminal_constraint(int on) { g_terminal_constraint = on != 0; }
  void orc_mpc_destroy(void * h) { delete (BatchMPC *)h; }
  void orc_mpc_generate_cycle(void * h, const unsigned char * cs, int n)
  {
    BatchMPC * m = (BatchMPC *)h;
    std::vector<std::vector<char>> v(n, std::vector<char>(m->nf));
    for (int i = 0; i < n; i++)
      for (int f = 0; f < m->nf; f++)
        v[i][f] = cs[i * m->nf + f];
    m->generateCycleHorizon(v);
  }
  void orc_mpc_switch_to_walk(void * h, const double * v6) { ((BatchMPC *)h)->switchToWalk(v6); }
  void orc_mpc_switch_to_stand(void * h) { ((BatchMPC *)h)->switchToStand(); }
  void orc_mpc_set_velocity_batched(void * h, const double * V) { ((BatchMPC *)h)->setVelocityBaseBatched(V); }
  // OCPHandler per-stage setters on the shared horizon (what: 0 control target, 1 state target; broadcast over the batch)
  void orc_mpc_set_stage_reference(void * h, int t, int what, const double * v)
  {
    BatchMPC * m = (BatchMPC *)h;
    if (what == 0)
      m->horizon[t].u_ref.assign(v, v + m->md.nu);
    else
    {
      m->horizon[t].x_tgt.assign(v, v + m->md.nx);
      for (int b = 0; b < m->B; b++)
        for (int i = 0; i < 6; i++)
          m->vref[b][t][i] = v[m->md.nq + i];
    }
  }
  void orc_cmpc_set_stage_reference(void * h, int t, int what, const double * v)
  {
    BatchMPCCent * m = (BatchMPCCent *)h;
    if (what == 0)
      m->horizon[t].u_ref.assign(v, v + m->md.nu);
    else
    {
      for (int i = 0; i < 3; i++)
      {
        m->horizon[t].x_tgt[i] = v[i];
        m->com_ref_member[i] = v[i];
      }
      for (int b = 0; b < m->B; b++)
        for (int i = 0; i < 6; i++)
          m->vref[b][t][i] = m->md.mass * v[3 + i];
    }
  }
  void orc_mpc_set_x_reference(void * h, const double * x)
  {
    BatchMPC * m = (BatchMPC *)h;
    m->x_reference.assign(x, x + m->md.nx);
  }
  void orc_mpc_set_early_exit(void * h, int on) { ((BatchMPC *)h)->early_exit_on_tol = on != 0; }
  double orc_mpc_iterate(void * h, const double * X)
  {
    auto t0 = std::chrono::steady_clock::now();
    ((BatchMPC *)h)->iterate(X);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  void orc_mpc_get(void * h, int what, double * out) { mpc_get_impl((BatchMPC *)h, what, out); }
  void orc_mpc_keep_knots(void * h, int on)
  {
    BatchMPC * m = (BatchMPC *)h;
    m->keep_knots = on != 0;
    m->last_knots.assign(m->B, std::vector<Knot>());
  }
  int orc_mpc_get_knot(void * h, int b, int t, double * out) { return mpc_get_knot_impl((BatchMPC *)h, b, t, out); }
  int orc_mpc_cold_iters(void * h) { return (int)((BatchMPC *)h)->cold_trace.size(); }
  void orc_mpc_cold_trace(void * h, double * out) // [n][4]: phi0, prim, dual, alpha
  {
    BatchMPC * m = (BatchMPC *)h;
    for (size_t i = 0; i < m->cold_trace.size(); i++)
    {
      out[4 * i] = m->cold_trace[i].phi0;
      out[4 * i + 1] = m->cold_trace[i].prim_infeas;
      out[4 * i + 2] = m->cold_trace[i].dual_infeas;
      out[4 * i + 3] = m->cold_trace[i].alpha;
    }
  }
  int orc_mpc_timing(void * h, int foot, int which, int * out, int cap)
  {
    return orc_timer_get(&((BatchMPC *)h)->timer, foot, which, out, cap);
  }

  // ---- the same host state machine on the full-dynamics stage model (orc_fulldyn.hpp) ----
  void * orc_fmpc_create(void * full, const orc_mpc_settings * s, int B, double gravity_arg)
  {
    FullModel * md = (FullModel *)full;
    MPCSettings ms;
    ms.swing_apex = s->swing_apex;
    ms.support_force = s->support_force;
    ms.TOL = s->TOL;
    ms.mu_init = s->mu_init;
    ms.timestep = s->timestep;
    ms.max_iters = s->max_iters;
    ms.num_threads = s->num_threads;
    ms.T_fly = s->T_fly;
    ms.T_contact = s->T_contact;
    ms.T = s->T;
#ifdef _OPENMP
    if (s->num_threads > 0)
      omp_set_num_threads(s->num_threads);
#endif
    return new BatchMPCFull(md->M, md->s, ms, s->T, B, gravity_arg, g_terminal_constraint);
  }
  void orc_fmpc_destroy(void * h) { delete (BatchMPCFull *)h; }
  void orc_fmpc_generate_cycle(void * h, const unsigned char * cs, int n)
  {
    BatchMPCFull * m = (BatchMPCFull *)h;
    std::vector<std::vector<char>> v(n, std::vector<char>(m->nf));
    for (int i = 0; i < n; i++)
      for (int f = 0; f < m->nf; f++)
        v[i][f] = cs[i * m->nf + f];
    m->generateCycleHorizon(v);
  }
  void orc_fmpc_switch_to_walk(void * h, const double * v6) { ((BatchMPCFull *)h)->switchToWalk(v6); }
  void orc_fmpc_switch_to_stand(void * h) { ((BatchMPCFull *)h)->switchToStand(); }
  void orc_fmpc_set_velocity_batched(void * h, const double * V) { ((BatchMPCFull *)h)->setVelocityBaseBatched(V); }
  // OCPHandler per-stage setters on the shared horizon (what: 0 control target, 1 state target; broadcast over the batch)
  void orc_fmpc_set_x_reference(void * h, const double * x)
  {
    BatchMPCFull * m = (BatchMPCFull *)h;
    m->x_reference.assign(x, x + m->md.nx);
  }
  double orc_fmpc_iterate(void * h, const double * X)
  {
    auto t0 = std::chrono::steady_clock::now();
    ((BatchMPCFull *)h)->iterate(X);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  void orc_fmpc_get(void * h, int what, double * out) { mpc_get_impl((BatchMPCFull *)h, what, out); }
  void orc_fmpc_keep_knots(void * h, int on)
  {
    BatchMPCFull * m = (BatchMPCFull *)h;
    m->keep_knots = on != 0;
    m->last_knots.assign(m->B, std::vector<Knot>());
  }
  int orc_fmpc_get_knot(void * h, int b, int t, double * out) { return mpc_get_knot_impl((BatchMPCFull *)h, b, t, out); }
  int orc_fmpc_cold_iters(void * h) { return (int)((BatchMPCFull *)h)->cold_trace.size(); }
  void orc_fmpc_cold_trace(void * h, double * out) // [n][4]: phi0, prim, dual, alpha
  {
    BatchMPCFull * m = (BatchMPCFull *)h;
    for (size_t i = 0; i < m->cold_trace.size(); i++)
    {
      out[4 * i] = m->cold_trace[i].phi0;
      out[4 * i + 1] = m->cold_trace[i].prim_infeas;
      out[4 * i + 2] = m->cold_trace[i].dual_infeas;
      out[4 * i + 3] = m->cold_trace[i].alpha;
    }
  }
  int orc_fmpc_timing(void * h, int foot, int which, int * out, int cap)
  {
    return orc_timer_get(&((BatchMPCFull *)h)->timer, foot, which, out, cap);
  }
  // single ProxDDP iteration on explicit inputs, for stage-by-stage GPU parity:
  // returns the LQ knots of iteration 1 for instance b (packed like orc_riccati inputs)
  // Interpolator (reference src/interpolator.cpp:5-78), floating-base robot with nq = nv + 1.
  //   kind 0: interpolateState (knots of size nq + nv), 1: interpolateConfiguration (nq), 2: interpolateLinear (dim)
  // step = (size_t)(delay / timestep); beyond the last interval -> last knot; else knot[step] towards knot[step+1]
  // by progress = (delay - step * timestep) / timestep: the configuration on the manifold
  // (integrate(q0, progress * difference(q0, q1)), what pinocchio::interpolate does), everything else linearly.
  void orc_interpolate(int kind, int nv, double delay, double timestep, const double * knots, int n, int dim, double * out)
  {
    const size_t step = (size_t)(delay / timestep);
    const double s = (delay - (double)step * timestep) / timestep;
    if (step >= (size_t)n - 1)
    {
      for (int i = 0; i < dim; i++)
        out[i] = knots[(size_t)(n - 1) * dim + i];
      return;
    }
    const double * k0 = knots + step * dim;
    const double * k1 = knots + (step + 1) * dim;
    if (kind == 2)
    {
      for (int i = 0; i < dim; i++)
        out[i] = k1[i] * s + k0[i] * (1.0 - s);
      return;
    }
    const int nq = nv + 1;
    std::vector<double> x0(nq + nv, 0.0), x1(nq + nv, 0.0), d(2 * nv), xo(nq + nv);
    std::copy(k0, k0 + nq, x0.begin());
    std::copy(k1, k1 + nq, x1.begin());
    x_difference(nq, nv, x0.data(), x1.data(), d.data());
    for (int i = 0; i < nv; i++)
    {
      d[i] *= s;
      d[nv + i] = 0.0;
    }
    x_integrate(nq, nv, x0.data(), d.data(), xo.data());
    for (int i = 0; i < nq; i++)
      out[i] = xo[i];
    if (kind == 0)
      for (int i = 0; i < nv; i++)
        out[nq + i] = k1[nq + i] * s + k0[nq + i] * (1.0 - s);
  }
  // FrictionCompensation::computeFriction (reference src/friction-compensation.cpp:22-37)
  void orc_friction(const double * dry, const double * viscous, int nu, const double * velocity, double * torque, int batch)
  {
    for (int b = 0; b < batch; b++)
      for (int j = 0; j < nu; j++)
      {
        const double v = velocity[(size_t)b * nu + j];
        torque[(size_t)b * nu + j] += viscous[j] * v + dry[j] * (double)((v > 0) - (v < 0));
      }
  }
  // CentroidalFwdDynamics + IntegratorEuler (reference src/centroidal-dynamics.cpp:79-81), 3-D forces:
  //   x = [c; h; L],  xdot = [h / m ; m g + sum_contact f_i ; sum_contact (p_i - c) x f_i],  x+ = x + dt xdot
  //   A = I + dt d(xdot)/dx,  B = dt d(xdot)/du   (row-major 9x9, 9x(3 nf); columns of feet in the air are zero)
  void orc_centroidal_dynamics(
    double mass, const double * gravity, double dt, int nf, const double * x, const double * u, const unsigned char * contact,
    const double * pos, double * xnext, double * A, double * B)
  {
    const int nu = 3 * nf;
    double xd[9];
    for (int i = 0; i < 3; i++)
    {
      xd[i] = x[3 + i] / mass;
      xd[3 + i] = mass * gravity[i];
      xd[6 + i] = 0.0;
    }
    for (int i = 0; i < 81; i++)
      A[i] = 0.0;
    for (int i = 0; i < 9 * nu; i++)
      B[i] = 0.0;
    for (int i = 0; i < 9; i++)
      A[i * 9 + i] = 1.0;
    for (int i = 0; i < 3; i++)
      A[i * 9 + 3 + i] = dt / mass;
    for (int f = 0; f < nf; f++)
    {
      if (!contact[f])
        continue;
      const double * F = u + 3 * f;
      const double r[3] = {pos[3 * f] - x[0], pos[3 * f + 1] - x[1], pos[3 * f + 2] - x[2]};
      for (int i = 0; i < 3; i++)
        xd[3 + i] += F[i];
      xd[6] += r[1] * F[2] - r[2] * F[1];
      xd[7] += r[2] * F[0] - r[0] * F[2];
      xd[8] += r[0] * F[1] - r[1] * F[0];
      // d(r x F)/dc = -d(r x F)/dr = [F]x ;  d(r x F)/dF = [r]x
      const double Fx[9] = {0, -F[2], F[1], F[2], 0, -F[0], -F[1], F[0], 0};
      const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
        {
          A[(6 + i) * 9 + j] += dt * Fx[i * 3 + j];
          B[(6 + i) * nu + 3 * f + j] = dt * rx[i * 3 + j];
        }
      for (int i = 0; i < 3; i++)
        B[(3 + i) * nu + 3 * f + i] = dt;
    }
    for (int i = 0; i < 9; i++)
      xnext[i] = x[i] + dt * xd[i];
  }

  // ---- centroidal OCP + MPC (orc_cent.hpp, orc_mpc_cent.hpp) ----
  void * orc_cent_create(
    const smpc_robot_model * m, double dt, const double * w_u, const double * w_com, const double * w_linear_mom,
    const double * w_angular_mom, const double * w_linear_acc, const double * w_angular_acc, const double * gravity, double mu)
  {
    CentSettings s;
    const int nu = 3 * m->nfeet;
    s.timestep = dt;
    s.w_u = mat_from(w_u, nu, nu);
    s.w_com = mat_from(w_com, 3, 3);
    s.w_linear_mom = mat_from(w_linear_mom, 3, 3);
    s.w_angular_mom = mat_from(w_angular_mom, 3, 3);
    s.w_linear_acc = mat_from(w_linear_acc, 3, 3);
    s.w_angular_acc = mat_from(w_angular_acc, 3, 3);
    for (int i = 0; i < 3; i++)
      s.gravity[i] = gravity[i];
    s.mu = mu;
    return new CentModel(m, s);
  }
  // 6-D feet: u = [(f, tau) per foot], wrench cones of a 2 Lfoot x 2 Wfoot sole (reference CentroidalSettings force_size / Lfoot / Wfoot,
  // include/simple-mpc/centroidal-dynamics.hpp:27-43).  Returns a new model; the old one stays valid.
  void * orc_cent_create6(
    const smpc_robot_model * m, double dt, const double * w_u, const double * w_com, const double * w_linear_mom,
    const double * w_angular_mom, const double * w_linear_acc, const double * w_angular_acc, const double * gravity, double mu, double Lfoot,
    double Wfoot)
  {
    CentSettings s;
    const int nu = 6 * m->nfeet;
    s.timestep = dt;
    s.force_size = 6;
    s.w_u = mat_from(w_u, nu, nu);
    s.w_com = mat_from(w_com, 3, 3);
    s.w_linear_mom = mat_from(w_linear_mom, 3, 3);
    s.w_angular_mom = mat_from(w_angular_mom, 3, 3);
    s.w_linear_acc = mat_from(w_linear_acc, 3, 3);
    s.w_angular_acc = mat_from(w_angular_acc, 3, 3);
    for (int i = 0; i < 3; i++)
      s.gravity[i] = gravity[i];
    s.mu = mu;
    s.Lfoot = Lfoot;
    s.Wfoot = Wfoot;
    return new CentModel(m, s);
  }
  void orc_cent_destroy(void * h) { delete (CentModel *)h; }
  static StageRef cent_ref(const CentModel & md, unsigned mask, const double * u_ref, const double * x_tgt, const double * pos)
  {
    StageRef r;
    r.mask = mask;
    r.u_ref.assign(u_ref, u_ref + md.nu);
    r.x_tgt.assign(x_tgt, x_tgt + 9);
    r.foot_ref.resize(md.nf);
    for (int f = 0; f < md.nf; f++)
      r.foot_ref[f] = v3(pos[3 * f], pos[3 * f + 1], pos[3 * f + 2]);
    return r;
  }
  void orc_cent_eval(
    void * h, unsigned mask, const double * u_ref, const double * x_tgt, const double * pos, const double * x, const double * u,
    double * xnext, double * xdot, double * cost, double * c)
  {
    CentModel * md = (CentModel *)h;
    Rigid scratch(md->M);
    StageEval o;
    md->eval(scratch, cent_ref(*md, mask, u_ref, x_tgt, pos), x, u, o);
    vec_to(o.xnext, xnext);
    vec_to(o.xdot, xdot);
    *cost = o.cost;
    vec_to(o.c, c);
  }
  void orc_cent_deriv(
    void * h, unsigned mask, const double * u_ref, const double * x_tgt, const double * pos, const double * x, const double * u,
    double * A, double * B, double * lx, double * lu, double * Lxx, double * Lxu, double * Luu, double * Cx, double * Cu)
  {
    CentModel * md = (CentModel *)h;
    Rigid scratch(md->M);
    StageDer o;
    md->deriv(scratch, cent_ref(*md, mask, u_ref, x_tgt, pos), x, u, o);
    mat_to(o.A, A);
    mat_to(o.B, B);
    vec_to(o.lx, lx);
    vec_to(o.lu, lu);
    mat_to(o.Lxx, Lxx);
    mat_to(o.Lxu, Lxu);
    mat_to(o.Luu, Luu);
    mat_to(o.Cx, Cx);
    mat_to(o.Cu, Cu);
  }
  void orc_cent_term(void * h, const double * x, double * cost, double * lx, double * Lxx)
  {
    CentModel * md = (CentModel *)h;
    Rigid scratch(md->M);
    Vec dummy, g;
    Mat Hm;
    *cost = md->term_eval(scratch, dummy, x);
    md->term_deriv(scratch, dummy, x, g, Hm);
    vec_to(g, lx);
    mat_to(Hm, Lxx);
  }
  void * orc_cmpc_create(void * cent, const smpc_robot_model * robot, const orc_mpc_settings * s, int B, double gravity_arg)
  {
    CentModel * md = (CentModel *)cent;
    MPCSettings ms;
    ms.swing_apex = s->swing_apex;
    ms.support_force = s->support_force;
    ms.TOL = s->TOL;
    ms.mu_init = s->mu_init;
    ms.timestep = s->timestep;
    ms.max_iters = s->max_iters;
    ms.num_threads = s->num_threads;
    ms.T_fly = s->T_fly;
    ms.T_contact = s->T_contact;
    ms.T = s->T;
#ifdef _OPENMP
    if (s->num_threads > 0)
      omp_set_num_threads(s->num_threads);
#endif
    return new BatchMPCCent(robot, md->s, ms, s->T, B, gravity_arg);
  }
  void orc_cmpc_destroy(void * h) { delete (BatchMPCCent *)h; }
  void orc_cmpc_generate_cycle(void * h, const unsigned char * cs, int n)
  {
    BatchMPCCent * m = (BatchMPCCent *)h;
    std::vector<std::vector<char>> v(n, std::vector<char>(m->nf));
    for (int i = 0; i < n; i++)
      for (int f = 0; f < m->nf; f++)
        v[i][f] = cs[i * m->nf + f];
    m->generateCycleHorizon(v);
  }
  void orc_cmpc_switch_to_walk(void * h, const double * v6) { ((BatchMPCCent *)h)->switchToWalk(v6); }
  void orc_cmpc_switch_to_stand(void * h) { ((BatchMPCCent *)h)->switchToStand(); }
  void orc_cmpc_set_velocity_batched(void * h, const double * V) { ((BatchMPCCent *)h)->setVelocityBaseBatched(V); }
  void orc_cmpc_set_x_reference(void * h, const double * x9) { ((BatchMPCCent *)h)->x_reference.assign(x9, x9 + 9); }
  double orc_cmpc_iterate(void * h, const double * X)
  {
    auto t0 = std::chrono::steady_clock::now();
    ((BatchMPCCent *)h)->iterate(X);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  // same `what` codes as orc_mpc_get; 7 = xdot [B][H][18] (2 nv with nv = 9: only the first 9 entries are used)
  void orc_cmpc_get(void * h, int what, double * out) { mpc_get_impl((BatchMPCCent *)h, what, out); }
  void orc_cmpc_keep_knots(void * h, int on)
  {
    BatchMPCCent * m = (BatchMPCCent *)h;
    m->keep_knots = on != 0;
    m->last_knots.assign(m->B, std::vector<Knot>());
  }
  int orc_cmpc_get_knot(void * h, int b, int t, double * out) { return mpc_get_knot_impl((BatchMPCCent *)h, b, t, out); }
  int orc_cmpc_cold_iters(void * h) { return (int)((BatchMPCCent *)h)->cold_trace.size(); }
  void orc_cmpc_cold_trace(void * h, double * out)
  {
    BatchMPCCent * m = (BatchMPCCent *)h;
    for (size_t i = 0; i < m->cold_trace.size(); i++)
    {
      out[4 * i] = m->cold_trace[i].phi0;
      out[4 * i + 1] = m->cold_trace[i].prim_infeas;
      out[4 * i + 2] = m->cold_trace[i].dual_infeas;
      out[4 * i + 3] = m->cold_trace[i].alpha;
    }
  }
  int orc_cmpc_timing(void * h, int foot, int which, int * out, int cap)
  {
    return orc_timer_get(&((BatchMPCCent *)h)->timer, foot, which, out, cap);
  }
  // FLOP instrumentation (only in the -DORC_COUNT_FLOPS build made by tools/count_flops.py; -1 otherwise)
  void orc_flops_reset()
  {
#ifdef ORC_COUNT_FLOPS
    flop_counter = 0.0;
#endif
  }
  double orc_flops_get()
  {
#ifdef ORC_COUNT_FLOPS
    return flop_counter;
#else
    return -1.0;
#endif
  }
  // number of OpenMP threads of the batched loops (the host may be under a CPU quota smaller than its hardware thread count)
  void orc_set_num_threads(int n)
  {
    if (n > 0)
      omp_set_num_threads(n);
  }
  int orc_num_threads()
  {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
  }

  // ---- whole-body inverse-dynamics QP (orc_id.hpp): KinodynamicsID for a batch ----
  struct orc_id_settings
  {
    double friction_coefficient, contact_weight_ratio_max, contact_weight_ratio_min;
    double kp_base, kp_posture, kp_contact;
    double w_base, w_posture, w_contact_motion, w_contact_force;
    int contact_motion_equality;
    double control_dt;
    const double *tau_max, *v_max, *q_min, *q_max;
    int admm_iters;
    double rho, sigma, alpha, admm_tol;
    int centroidal;
    double kp_com, kp_feet_tracking, w_com, w_feet_tracking;
    int base_reference_as_coded, tsid_joint_bounds;
    int force_size;              // 3: point feet ; 6: flat feet (tsid Contact6d)
    const double * quad_points;  // force_size 6: [nf][4][3] corners of the soles in their foot frames
  };
  static IDSettings id_settings_from(const smpc_robot_model * m, const orc_id_settings * c)
  {
    IDSettings s;
    s.friction_coefficient = c->friction_coefficient;
    s.contact_weight_ratio_max = c->contact_weight_ratio_max;
    s.contact_weight_ratio_min = c->contact_weight_ratio_min;
    s.kp_base = c->kp_base;
    s.kp_posture = c->kp_posture;
    s.kp_contact = c->kp_contact;
    s.w_base = c->w_base;
    s.w_posture = c->w_posture;
    s.w_contact_motion = c->w_contact_motion;
    s.w_contact_force = c->w_contact_force;
    s.contact_motion_equality = c->contact_motion_equality != 0;
    s.control_dt = c->control_dt;
    const int na = m->nv - 6;
    s.tau_max.assign(c->tau_max, c->tau_max + na);
    s.v_max.assign(c->v_max, c->v_max + na);
    s.q_min.assign(c->q_min, c->q_min + na);
    s.q_max.assign(c->q_max, c->q_max + na);
    s.admm_iters = c->admm_iters;
    s.rho = c->rho;
    s.sigma = c->sigma;
    s.alpha = c->alpha;
    s.admm_tol = c->admm_tol;
    s.centroidal = c->centroidal != 0;
    s.kp_com = c->kp_com;
    s.kp_feet_tracking = c->kp_feet_tracking;
    s.w_com = c->w_com;
    s.w_feet_tracking = c->w_feet_tracking;
    s.base_reference_as_coded = c->base_reference_as_coded != 0;
    s.tsid_joint_bounds = c->tsid_joint_bounds != 0;
    s.force_size = c->force_size == 6 ? 6 : 3;
    if (s.force_size == 6)
      s.quad_points.assign(c->quad_points, c->quad_points + (size_t)m->nfeet * 12);
    return s;
  }
  void * orc_id_create(const smpc_robot_model * m, const orc_id_settings * c, int B) { return new BatchKinoID(m, id_settings_from(m, c), B); }
  void orc_id_destroy(void * h) { delete (BatchKinoID *)h; }
  // target of instance b (b < 0: every instance): q (nq), v (nv), a (nv), contact mask, f (3 nf)
  void orc_id_set_target(void * h, int b, const double * q, const double * v, const double * a, unsigned mask, const double * f)
  {
    BatchKinoID * k = (BatchKinoID *)h;
    IDTarget t;
    t.q.assign(q, q + k->M->nq);
    t.v.assign(v, v + k->M->nv);
    t.a.assign(a, a + k->M->nv);
    t.mask = mask;
    t.f.assign(f, f + k->s.nfw() * k->M->nfeet);
    for (int i = 0; i < k->B; i++)
      if (b < 0 || b == i)
      {
        t.com = k->tgt[i].com;
        t.vcom = k->tgt[i].vcom;
        t.feet_p = k->tgt[i].feet_p;
        t.feet_v = k->tgt[i].feet_v;
        k->tgt[i] = t;
      }
  }
  // CentroidalID::setTarget (centroidal-id.cpp:87-145): CoM position / velocity, foot positions / velocities, contacts, forces; the
  // posture / base targets become the reference state
  void orc_id_set_target_centroidal(void * h, int b, const double * com, const double * vcom, const double * feet_p, const double * feet_v, unsigned mask, const double * f)
  {
    BatchKinoID * k = (BatchKinoID *)h;
    const int nf = k->M->nfeet;
    IDTarget t;
    t.q.assign(k->M->q_ref, k->M->q_ref + k->M->nq);
    t.v.assign(k->M->nv, 0.0);
    t.a.assign(k->M->nv, 0.0);
    t.mask = mask;
    t.f.assign(f, f + k->s.nfw() * nf);
    t.com.assign(com, com + 3);
    t.vcom.assign(vcom, vcom + 3);
    t.feet_p.assign(feet_p, feet_p + 3 * nf);
    t.feet_v.assign(feet_v, feet_v + 3 * nf);
    for (int i = 0; i < k->B; i++)
      if (b < 0 || b == i)
        k->tgt[i] = t;
  }
  void orc_id_solve(void * h, const double * X, double * tau, double * a, double * f, double * resid)
  {
    BatchKinoID * k = (BatchKinoID *)h;
    k->solve(X, tau, a, f);
    for (int b = 0; b < k->B; b++)
      resid[b] = k->resid[b];
  }
  // pieces, for the tests of the kernels: M (nv x nv), nle (nv), J (3 nf x nv), Jdv (3 nf), vfoot (3 nf)
  // (orc_id_quantities6: the LOCAL 6-D rows of flat feet, J (6 nf x nv), Jdv, vfoot (6 nf))
  void orc_id_quantities6(const smpc_robot_model * m, const double * x, double * Mq, double * nle, double * J, double * Jdv, double * vfoot)
  {
    IDQuantities Q;
    id_quantities(m, x, Q, 6);
    mat_to(Q.M, Mq);
    vec_to(Q.nle, nle);
    mat_to(Q.J, J);
    vec_to(Q.Jdv, Jdv);
    vec_to(Q.vfoot, vfoot);
  }
  void orc_id_quantities(const smpc_robot_model * m, const double * x, double * Mq, double * nle, double * J, double * Jdv, double * vfoot)
  {
    IDQuantities Q;
    id_quantities(m, x, Q);
    mat_to(Q.M, Mq);
    vec_to(Q.nle, nle);
    mat_to(Q.J, J);
    vec_to(Q.Jdv, Jdv);
    vec_to(Q.vfoot, vfoot);
  }
  // QP of instance b at state x: H (n x n), g (n), C (m x n), l, u (m); returns m
  int orc_id_qp(void * h, int b, const double * x, double * H, double * g, double * C, double * l, double * u)
  {
    BatchKinoID * k = (BatchKinoID *)h;
    IDQuantities Q;
    id_quantities(k->M, x, Q, k->s.force_size);
    QP qp;
    id_assemble(k->M, k->s, k->tgt[b], x, Q, qp);
    mat_to(qp.H, H);
    vec_to(qp.g, g);
    mat_to(qp.C, C);
    vec_to(qp.l, l);
    vec_to(qp.u, u);
    return qp.m;
  }

  // ---- converged solves and constraint typing of the three stage models (tests/test_oracle_vs_scipy.py) ----
  // masks [H]; u_ref [H][nu], x_tgt [H][nx], foot_ref [H][nf*3] per stage; x_tgt_term [nx]
  int orc_kino_solve(
    void * h, int H, const unsigned * masks, const double * u_ref, const double * x_tgt, const double * foot_ref, const double * x_tgt_term,
    const double * x0, const double * u0, int max_iter, double tol, double mu, double * trace, double * xs, double * us, double * vs, double * lams)
  {
    KinoModel * md = (KinoModel *)h;
    return generic_solve(*md, md->M, H, [&](int t) { return make_ref(*md, masks[t], u_ref + (size_t)t * md->nu, x_tgt + (size_t)t * md->nx, foot_ref + (size_t)t * md->nf * 3); },
                         x_tgt_term, x0, u0, max_iter, tol, mu, trace, xs, us, vs, lams);
  }
  void orc_kino_row_kinds(void * h, unsigned mask, int * kind, double * lo, double * hi)
  {
    KinoModel * md = (KinoModel *)h;
    std::vector<double> z(md->nx + md->nu + md->nf * 3, 0.0);
    generic_row_kinds(*md, make_ref(*md, mask, z.data(), z.data(), z.data()), kind, lo, hi);
  }
  int orc_cent_solve(
    void * h, int H, const unsigned * masks, const double * u_ref, const double * x_tgt, const double * pos, const double * x0, const double * u0,
    int max_iter, double tol, double mu, double * trace, double * xs, double * us, double * vs, double * lams)
  {
    CentModel * md = (CentModel *)h;
    std::vector<double> zt(9, 0.0);
    return generic_solve(*md, md->M, H, [&](int t) { return cent_ref(*md, masks[t], u_ref + (size_t)t * md->nu, x_tgt + (size_t)t * 9, pos + (size_t)t * md->nf * 3); },
                         zt.data(), x0, u0, max_iter, tol, mu, trace, xs, us, vs, lams);
  }
  void orc_cent_row_kinds(void * h, unsigned mask, int * kind, double * lo, double * hi)
  {
    CentModel * md = (CentModel *)h;
    std::vector<double> z(64, 0.0);
    generic_row_kinds(*md, cent_ref(*md, mask, z.data(), z.data(), z.data()), kind, lo, hi);
  }
  void orc_full_row_kinds(void * h, unsigned mask, int * kind, double * lo, double * hi)
  {
    FullModel * md = (FullModel *)h;
    std::vector<double> z(md->nx + md->nu + md->nf * 3 + 64, 0.0);
    generic_row_kinds(*md, full_ref(*md, mask, z.data(), z.data(), z.data()), kind, lo, hi);
  }
}
