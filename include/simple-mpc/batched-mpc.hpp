// simple-mpc/batched-mpc.hpp -- header-only C++ host mirror of the reference's MPC class over the C ABI
// of smpc.h.  Same namespace, verb and settings-field names as the reference
// (include/simple-mpc/mpc.hpp:29-49,55-197; include/simple-mpc/kinodynamics.hpp:24-51), Eigen types replaced by
// std::vector<double> (Eigen is not available in this build).  Errors are rethrown as std::runtime_error, as in the
// reference (src/kinodynamics.cpp:175,235,279; src/ocp-handler.cpp:28,32,76).
#pragma once
#include "../smpc.h"
#include <map>
#include <stdexcept>
#include <string>
#include <memory>
#include <vector>

namespace simple_mpc
{
  struct KinodynamicsSettings // reference include/simple-mpc/kinodynamics.hpp:24-51
  {
    double timestep = 0.01;
    std::vector<double> w_x, w_u, w_frame, w_cent, w_centder; // dense row-major matrices
    std::vector<double> qmin, qmax;
    double gravity[3] = {0, 0, -9.81};
    double mu = 0.8, Lfoot = 0.01, Wfoot = 0.01;
    int force_size = 3;
    bool kinematics_limits = true, force_cone = false, land_cstr = false;
    bool terminal_constraint = false; // createProblem's last argument (reference src/ocp-handler.cpp:96-137)
  };

  struct CentroidalSettings // reference include/simple-mpc/centroidal-dynamics.hpp:27-43
  {
    double timestep = 0.01;
    std::vector<double> w_u;                                                         // nu x nu
    std::vector<double> w_com, w_linear_mom, w_angular_mom, w_linear_acc, w_angular_acc; // 3 x 3
    double gravity[3] = {0, 0, -9.81};
    double mu = 0.8, Lfoot = 0.1, Wfoot = 0.075;
    int force_size = 3;
  };

  struct FullDynamicsSettings // reference include/simple-mpc/fulldynamics.hpp:28-65
  {
    double timestep = 0.01;
    std::vector<double> w_x, w_u, w_cent, w_forces, w_frame; // dense row-major: ndx^2, nu^2 (nu = nv - 6), 36, force_size^2 twice
    std::vector<double> umin, umax, qmin, qmax;               // nu each
    std::vector<double> Kp_correction, Kd_correction;         // force_size each
    double gravity[3] = {0, 0, -9.81};
    double mu = 0.8, Lfoot = 0.1, Wfoot = 0.075;
    int force_size = 6;
    bool torque_limits = true, kinematics_limits = true, force_cone = true, land_cstr = false;
    bool terminal_constraint = false; // createProblem's last argument
  };

  struct MPCSettings // reference include/simple-mpc/mpc.hpp:29-49
  {
    double swing_apex = 0.15;
    double support_force = 1000;
    double TOL = 1e-4;
    double mu_init = 1e-8;
    std::size_t max_iters = 1;
    std::size_t num_threads = 2;
    int T_fly = 80;
    int T_contact = 20;
    std::size_t T = 100;
    double timestep = 0.01;
    // extension: SolverProxDDP's convergence test inside iterate (the reference passes TOL to its solver, src/mpc.cpp:43,212); off: exactly
    // max_iters iterations per control step (the metric of record)
    bool early_exit_on_tol = false;
  };

  class BatchedMPC
  {
    smpc_handle * h_ = nullptr;
    int dims_[8] = {0};
    int batch_ = 0;
    int force_size_ = 3;
    std::vector<std::string> ee_names_;
    static void check(int code)
    {
      if (code < 0)
        throw std::runtime_error(smpc_last_error());
    }

  public:
    MPCSettings settings_;
    std::vector<double> xs_, us_, K0_; // [B][H+1][nx], [B][H][nu], [B][nu][ndx] after iterate()

    // KinodynamicsOCP(settings, model) + createProblem(x_ref, T, force_size, gravity, false) + MPC(settings, ocp)
    BatchedMPC(const smpc_robot_model * robot, const KinodynamicsSettings & ocp, const MPCSettings & settings, int batch,
               double gravity_arg = -9.81, int device_id = 0)
    : batch_(batch), settings_(settings)
    {
      smpc_kinodynamics_settings ks{};
      ks.timestep = ocp.timestep;
      ks.w_x = ocp.w_x.data();
      ks.w_u = ocp.w_u.data();
      ks.w_frame = ocp.w_frame.data();
      ks.w_cent = ocp.w_cent.data();
      ks.w_centder = ocp.w_centder.data();
      ks.qmin = ocp.qmin.data();
      ks.qmax = ocp.qmax.data();
      for (int i = 0; i < 3; i++)
        ks.gravity[i] = ocp.gravity[i];
      ks.mu = ocp.mu;
      ks.Lfoot = ocp.Lfoot;
      ks.Wfoot = ocp.Wfoot;
      ks.force_size = ocp.force_size;
      ks.kinematics_limits = ocp.kinematics_limits;
      ks.force_cone = ocp.force_cone;
      ks.land_cstr = ocp.land_cstr;
      ks.terminal_constraint = ocp.terminal_constraint;
      smpc_mpc_settings ms = c_settings(settings);
      check(smpc_create(robot, &ks, &ms, batch, gravity_arg, device_id, &h_));
      if (settings.early_exit_on_tol)
        check(smpc_set_early_exit_on_tol(h_, 1));
      finish(robot);
    }
    // CentroidalOCP(settings, model) + createProblem(getCentroidalState(), T, force_size, gravity, false) + MPC(settings, ocp)
    // (reference src/centroidal-dynamics.cpp:27-37; tests/mpc.cpp:172-258).  iterate() still takes the measured multibody
    // states [B][nq + nv]; xs_ holds centroidal states [B][H+1][9].
    BatchedMPC(const smpc_robot_model * robot, const CentroidalSettings & ocp, const MPCSettings & settings, int batch,
               double gravity_arg = -9.81, int device_id = 0)
    : batch_(batch), settings_(settings)
    {
      smpc_centroidal_settings cs{};
      cs.timestep = ocp.timestep;
      cs.w_u = ocp.w_u.data();
      cs.w_com = ocp.w_com.data();
      cs.w_linear_mom = ocp.w_linear_mom.data();
      cs.w_angular_mom = ocp.w_angular_mom.data();
      cs.w_linear_acc = ocp.w_linear_acc.data();
      cs.w_angular_acc = ocp.w_angular_acc.data();
      for (int i = 0; i < 3; i++)
        cs.gravity[i] = ocp.gravity[i];
      cs.mu = ocp.mu;
      cs.Lfoot = ocp.Lfoot;
      cs.Wfoot = ocp.Wfoot;
      cs.force_size = ocp.force_size;
      if ((int)ocp.w_u.size() != ocp.force_size * robot->nfeet * ocp.force_size * robot->nfeet || ocp.w_com.size() != 9 || ocp.w_linear_mom.size() != 9
          || ocp.w_angular_mom.size() != 9 || ocp.w_linear_acc.size() != 9 || ocp.w_angular_acc.size() != 9)
        throw std::runtime_error("centroidal settings: weight sizes do not match the robot");
      smpc_mpc_settings ms = c_settings(settings);
      check(smpc_create_centroidal(robot, &cs, &ms, batch, gravity_arg, device_id, &h_));
      finish(robot);
    }
    // FullDynamicsOCP(settings, model) + createProblem(x_ref, T, force_size, gravity, false) + MPC(settings, ocp)
    // (reference src/fulldynamics.cpp:30-76; benchmark/talos.cpp).  us_ holds the nv - 6 joint torques.
    BatchedMPC(const smpc_robot_model * robot, const FullDynamicsSettings & ocp, const MPCSettings & settings, int batch,
               double gravity_arg = -9.81, int device_id = 0)
    : batch_(batch), settings_(settings)
    {
      const size_t nu = (size_t)robot->nv - 6, ndx = 2 * (size_t)robot->nv, fs = (size_t)ocp.force_size;
      if (ocp.w_x.size() != ndx * ndx || ocp.w_u.size() != nu * nu || ocp.w_cent.size() != 36 || ocp.w_forces.size() != fs * fs
          || ocp.w_frame.size() != fs * fs || ocp.umin.size() != nu || ocp.umax.size() != nu || ocp.qmin.size() != nu
          || ocp.qmax.size() != nu || ocp.Kp_correction.size() != fs || ocp.Kd_correction.size() != fs)
        throw std::runtime_error("full-dynamics settings: sizes do not match the robot");
      smpc_fulldynamics_settings fsn{};
      fsn.timestep = ocp.timestep;
      fsn.w_x = ocp.w_x.data();
      fsn.w_u = ocp.w_u.data();
      fsn.w_cent = ocp.w_cent.data();
      fsn.w_forces = ocp.w_forces.data();
      fsn.w_frame = ocp.w_frame.data();
      fsn.umin = ocp.umin.data();
      fsn.umax = ocp.umax.data();
      fsn.qmin = ocp.qmin.data();
      fsn.qmax = ocp.qmax.data();
      fsn.Kp_correction = ocp.Kp_correction.data();
      fsn.Kd_correction = ocp.Kd_correction.data();
      for (int i = 0; i < 3; i++)
        fsn.gravity[i] = ocp.gravity[i];
      fsn.mu = ocp.mu;
      fsn.Lfoot = ocp.Lfoot;
      fsn.Wfoot = ocp.Wfoot;
      fsn.force_size = ocp.force_size;
      fsn.torque_limits = ocp.torque_limits;
      fsn.kinematics_limits = ocp.kinematics_limits;
      fsn.force_cone = ocp.force_cone;
      fsn.land_cstr = ocp.land_cstr;
      fsn.terminal_constraint = ocp.terminal_constraint;
      force_size_ = ocp.force_size;
      smpc_mpc_settings ms = c_settings(settings);
      check(smpc_create_fulldynamics(robot, &fsn, &ms, batch, gravity_arg, device_id, &h_));
      if (settings.early_exit_on_tol)
        check(smpc_set_early_exit_on_tol(h_, 1));
      finish(robot);
    }
    ~BatchedMPC() { smpc_destroy(h_); }
    BatchedMPC(const BatchedMPC &) = delete;
    BatchedMPC & operator=(const BatchedMPC &) = delete;

    int nx() const { return dims_[2]; }             // problem state (centroidal handle: 9)
    int nx_in() const { return dims_[0] + dims_[1]; } // measured multibody state nq + nv
    int ndx() const { return dims_[3]; }
    int nu() const { return dims_[4]; }
    int horizon() const { return dims_[7]; }

    // reference src/mpc.cpp:101-187
    void generateCycleHorizon(const std::vector<std::map<std::string, bool>> & contact_states)
    {
      std::vector<uint8_t> cs;
      for (auto & st : contact_states)
        for (auto & n : ee_names_)
          cs.push_back(st.at(n) ? 1 : 0);
      check(smpc_generate_cycle_horizon(h_, cs.data(), (int)contact_states.size()));
    }
    // reference src/mpc.cpp:189-218; X is [B][nx]
    void iterate(const std::vector<double> & X)
    {
      if ((int)X.size() != batch_ * nx_in())
        throw std::runtime_error("X must hold batch * (nq + nv) values");
      check(smpc_iterate(h_, X.data()));
      xs_.resize((size_t)batch_ * (horizon() + 1) * nx());
      us_.resize((size_t)batch_ * horizon() * nu());
      K0_.resize((size_t)batch_ * nu() * ndx());
      check(smpc_get_xs(h_, xs_.data()));
      check(smpc_get_us(h_, us_.data()));
      check(smpc_get_K0(h_, K0_.data()));
    }
    // the pieces of iterate() for callers that drive several handles (BatchedMPCGroup): launch without waiting, the small return set
    // [x1 | u0 | K0] of every instance as rows of a caller-owned (pinned) buffer, wait
    void iterateAsync(const double * X) { check(smpc_iterate_async(h_, X)); }
    void gatherOutputs(double * out, std::size_t row_doubles) { check(smpc_gather_outputs(h_, out, row_doubles)); }
    // ... packed into a device buffer of this handle's device ([batch][row_doubles]), or into a buffer on another device of the node (peer
    // copy over xGMI; rows contiguous: [batch][gatherRow()]) -- both asynchronous on the handle's stream
    void gatherOutputsDevice(double * out_device, std::size_t row_doubles) { check(smpc_gather_outputs_device(h_, out_device, row_doubles)); }
    void gatherOutputsPeer(double * out_peer, int dst_device) { check(smpc_gather_outputs_peer(h_, out_peer, dst_device)); }
    void wait() { check(smpc_wait(h_)); }
    int gatherRow() const { return nx() + nu() + nu() * ndx(); }
    // MPC::getContactForces for every stage (reference src/mpc.cpp:354-380): [B][H][nfeet][force_size]; full-dynamics handles only
    std::vector<double> getContactForces()
    {
      std::vector<double> f((size_t)batch_ * horizon() * ee_names_.size() * force_size_);
      check(smpc_get_contact_forces(h_, f.data()));
      return f;
    }
    // per-instance status words of the last control step (smpc_get_status: SMPC_STATUS_* bits, 0 = healthy); returns how many are not
    int status(std::vector<int> & words)
    {
      words.assign((size_t)batch_, 0);
      const int rc = smpc_get_status(h_, words.data());
      check(rc);
      return rc;
    }
    void switchToWalk(const double * velocity_base6) { check(smpc_switch_to_walk(h_, velocity_base6)); }
    void switchToStand() { check(smpc_switch_to_stand(h_)); }
    // one velocity command per instance, V: [B][6] (the reference's MPC::velocity_base_, one per robot of the batch)
    void setVelocityBaseBatched(const std::vector<double> & V)
    {
      if ((int)V.size() != batch_ * 6)
        throw std::runtime_error("velocity_base size should be batch * 6");
      check(smpc_set_velocity_base_batched(h_, V.data()));
    }
    // OCPHandler per-stage setters / getters (reference include/simple-mpc/ocp-handler.hpp:66-127), broadcast over the batch
    void setReferenceControl(std::size_t t, const std::vector<double> & u_ref) { check(smpc_set_stage_reference(h_, (int)t, 0, u_ref.data(), (int)u_ref.size())); }
    std::vector<double> getReferenceControl(std::size_t t)
    {
      std::vector<double> u(nu());
      check(smpc_get_stage_reference(h_, (int)t, 0, u.data(), nu()));
      return u;
    }
    void setReferenceState(std::size_t t, const std::vector<double> & x_ref) { check(smpc_set_stage_reference(h_, (int)t, 1, x_ref.data(), (int)x_ref.size())); }
    std::vector<double> getReferenceState(std::size_t t)
    {
      std::vector<double> x(nx());
      check(smpc_get_stage_reference(h_, (int)t, 1, x.data(), nx()));
      return x;
    }
    void setReferencePose(std::size_t t, const std::string & ee_name, const double * translation3) { check(smpc_set_reference_pose(h_, (int)t, foot(ee_name), translation3)); }
    std::vector<double> getReferencePose(std::size_t t, const std::string & ee_name)
    {
      std::vector<double> p(3);
      check(smpc_get_reference_pose(h_, (int)t, foot(ee_name), 0, p.data()));
      return p;
    }
    // ... with the rotation of the SE3 (row-major 3 x 3): kept and returned (reference tests/problem.cpp:157-160) until the next iterate, which
    // rewrites every stage's pose with the identity rotation (src/mpc.cpp:303-309)
    struct Placement
    {
      double translation[3];
      double rotation[9];
    };
    void setReferencePose(std::size_t t, const std::string & ee_name, const Placement & M)
    {
      check(smpc_set_reference_pose_se3(h_, (int)t, foot(ee_name), M.translation, M.rotation));
    }
    Placement getReferencePlacement(std::size_t t, const std::string & ee_name)
    {
      Placement M;
      check(smpc_get_reference_pose_se3(h_, (int)t, foot(ee_name), 0, M.translation, M.rotation));
      return M;
    }
    std::vector<bool> getContactState(std::size_t t)
    {
      std::vector<uint8_t> c(ee_names_.size());
      check(smpc_get_contact_state(h_, (int)t, c.data()));
      return std::vector<bool>(c.begin(), c.end());
    }
    // checkpoint / resume of the whole batch
    std::vector<unsigned char> saveState()
    {
      size_t n = 0, w = 0;
      check(smpc_state_size(h_, &n));
      std::vector<unsigned char> buf(n);
      check(smpc_save_state(h_, buf.data(), n, &w));
      buf.resize(w);
      return buf;
    }
    void loadState(const std::vector<unsigned char> & buf) { check(smpc_load_state(h_, buf.data(), buf.size())); }
    std::vector<int> getFootTakeoffCycle(const std::string & ee) { return timing(ee, 0); }
    std::vector<int> getFootLandCycle(const std::string & ee) { return timing(ee, 1); }
    smpc_handle * handle() { return h_; }

    // reference src/mpc.cpp:346-352, t = 0 or 1: [B][dim] with dim = 2 nv (kinodynamics) or 9 (centroidal)
    std::vector<double> getStateDerivative(int t)
    {
      if (t != 0 && t != 1)
        throw std::runtime_error("state derivative is retained for t = 0, 1 only");
      const int dim = nx() == 9 ? 9 : 2 * dims_[1];
      std::vector<double> all((size_t)batch_ * 2 * dim), out((size_t)batch_ * dim);
      check(smpc_get_state_derivative01(h_, all.data()));
      for (int b = 0; b < batch_; b++)
        for (int i = 0; i < dim; i++)
          out[(size_t)b * dim + i] = all[((size_t)b * 2 + t) * dim + i];
      return out;
    }

  private:
    static smpc_mpc_settings c_settings(const MPCSettings & settings)
    {
      smpc_mpc_settings ms{};
      ms.swing_apex = settings.swing_apex;
      ms.support_force = settings.support_force;
      ms.TOL = settings.TOL;
      ms.mu_init = settings.mu_init;
      ms.max_iters = (int)settings.max_iters;
      ms.num_threads = (int)settings.num_threads;
      ms.T_fly = settings.T_fly;
      ms.T_contact = settings.T_contact;
      ms.T = (int)settings.T;
      ms.timestep = settings.timestep;
      return ms;
    }
    void finish(const smpc_robot_model * robot)
    {
      check(smpc_get_dims(h_, dims_));
      for (int f = 0; f < robot->nfeet; f++)
        ee_names_.push_back(robot->foot_name[f]);
    }
    int foot(const std::string & ee) const
    {
      for (size_t f = 0; f < ee_names_.size(); f++)
        if (ee_names_[f] == ee)
          return (int)f;
      throw std::runtime_error("unknown end effector " + ee);
    }
    std::vector<int> timing(const std::string & ee, int which)
    {
      for (size_t f = 0; f < ee_names_.size(); f++)
        if (ee_names_[f] == ee)
        {
          std::vector<int> v(256);
          const int n = smpc_get_foot_timing(h_, (int)f, which, v.data(), 256);
          check(n);
          v.resize(n);
          return v;
        }
      throw std::runtime_error("unknown end effector " + ee);
    }
  };

  // The batch sharded over the GPUs of one node: one BatchedMPC per device, contiguous blocks of instances, no collective on
  // the solve path; one control step = every device launched back to back from this thread, then one pass collecting what the
  // controllers consume -- xs[1], us[0], K_0 -- into ONE host buffer (rows [x1 | u0 | K0]; pass pinned memory for full PCIe rate).
  // Kinodynamics OCP.  Every handle runs the same gait, commands and settings.
  class BatchedMPCGroup
  {
    std::vector<std::unique_ptr<BatchedMPC>> parts_;
    std::vector<int> first_; // first instance of every part
    int batch_ = 0;

  public:
    BatchedMPCGroup(const smpc_robot_model * robot, const KinodynamicsSettings & ocp, const MPCSettings & settings, int batch,
                    const std::vector<int> & device_ids, double gravity_arg = -9.81)
    : batch_(batch)
    {
      const int n = (int)device_ids.size();
      if (n <= 0 || batch < n)
        throw std::runtime_error("BatchedMPCGroup: at least one device and one instance per device");
      for (int i = 0; i < n; i++)
      {
        const int i0 = (int)((long long)batch * i / n), i1 = (int)((long long)batch * (i + 1) / n);
        first_.push_back(i0);
        parts_.emplace_back(new BatchedMPC(robot, ocp, settings, i1 - i0, gravity_arg, device_ids[i]));
      }
      first_.push_back(batch);
    }
    int parts() const { return (int)parts_.size(); }
    int batch() const { return batch_; }
    BatchedMPC & part(int i) { return *parts_[i]; }
    int firstInstance(int i) const { return first_[i]; }
    int gatherRow() const { return parts_[0]->gatherRow(); }
    void generateCycleHorizon(const std::vector<std::map<std::string, bool>> & contact_states)
    {
      for (auto & p : parts_)
        p->generateCycleHorizon(contact_states);
    }
    void switchToWalk(const double * v6)
    {
      for (auto & p : parts_)
        p->switchToWalk(v6);
    }
    void switchToStand()
    {
      for (auto & p : parts_)
        p->switchToStand();
    }
    // X: [batch][nx] measured states; out: [batch][gatherRow()] rows [x1 | u0 | K0] (both caller-owned; they must stay valid until this
    // returns).  All devices work concurrently; returns when every part is complete.
    void iterate(const double * X, double * out)
    {
      const int nxin = parts_[0]->nx_in(), row = gatherRow();
      for (int i = 0; i < parts(); i++)
        parts_[i]->iterateAsync(X + (std::size_t)first_[i] * nxin);
      for (int i = 0; i < parts(); i++)
        parts_[i]->gatherOutputs(out + (std::size_t)first_[i] * row, (std::size_t)row);
      for (auto & p : parts_)
        p->wait();
    }
  };
} // namespace simple_mpc
