// simple-mpc/batched-id.hpp -- header-only C++ host mirror of the reference's whole-body inverse-dynamics controllers over the C ABI of
// smpc.h: KinodynamicsID (include/simple-mpc/inverse-dynamics/kinodynamics-id.hpp:17-92) and CentroidalID
// (include/simple-mpc/inverse-dynamics/centroidal-id.hpp), one controller per robot of a batch.  Same verbs and settings-field names as
// the reference; Eigen types replaced by std::vector<double>, pinocchio::SE3 / Motion arguments by translations / linear velocities.
// Point feet (tsid ContactPoint, 3 force components per foot) by default; a robot with flat feet (tsid Contact6d: addQuadFoot in the
// reference, kinodynamics-id.cpp:41-49) passes force_size = 6 and the corners of its soles -- forces are then wrenches, 6 per foot.
// The reference reads effort / velocity limits from its pinocchio model: here they are constructor arguments, the position
// limits come from the robot table.  Errors are rethrown as std::runtime_error.
#pragma once
#include "../smpc.h"
#include <stdexcept>
#include <string>
#include <vector>

namespace simple_mpc
{
  struct KinodynamicsIDSettings // reference kinodynamics-id.hpp:24-50
  {
    double friction_coefficient = 0.6;
    double contact_weight_ratio_max = 10.0;
    double contact_weight_ratio_min = 0.01;
    double kp_base = 0., kp_posture = 0., kp_contact = 0.;
    double w_base = -1., w_posture = -1., w_contact_motion = -1., w_contact_force = -1.; // <= 0: task disabled
    bool contact_motion_equality = false;
    // extensions (see include/smpc.h): the base task literally as the reference codes it (kinodynamics-id.cpp:222-223); TSID's
    // TaskJointPosVelAccBounds in full
    bool base_reference_as_coded = false;
    bool tsid_joint_bounds = false;
    // contact model: 3 = point feet; 6 = flat feet with quad_contact_points [nfeet][4][3], the corners of every sole in its foot frame
    // (RobotModelHandler::getQuadFootContactPoints of the reference)
    int force_size = 3;
    std::vector<double> quad_contact_points;
  };
  struct CentroidalIDSettings : KinodynamicsIDSettings // reference centroidal-id.hpp
  {
    double kp_com = 0., kp_feet_tracking = 0.;
    double w_com = -1., w_feet_tracking = -1.;
  };

  class BatchedIDBase
  {
  public:
    ~BatchedIDBase()
    {
      if (h_)
        smpc_id_destroy(h_);
    }
    BatchedIDBase(const BatchedIDBase &) = delete;
    BatchedIDBase & operator=(const BatchedIDBase &) = delete;
    int batch() const { return batch_; }
    // solve(t, q_meas, v_meas, tau_res) for the batch: q_meas [B][nq], v_meas [B][nv] -> tau_res [B][nv - 6]  (kinodynamics-id.cpp:185-237)
    void solve(double /*t*/, const std::vector<double> & q_meas, const std::vector<double> & v_meas, std::vector<double> & tau_res)
    {
      if (q_meas.size() != (size_t)batch_ * nq_ || v_meas.size() != (size_t)batch_ * nv_)
        throw std::runtime_error("q_meas / v_meas must hold one configuration / velocity per robot");
      std::vector<double> X((size_t)batch_ * (nq_ + nv_));
      for (int b = 0; b < batch_; b++)
      {
        std::copy(q_meas.begin() + (size_t)b * nq_, q_meas.begin() + (size_t)(b + 1) * nq_, X.begin() + (size_t)b * (nq_ + nv_));
        std::copy(v_meas.begin() + (size_t)b * nv_, v_meas.begin() + (size_t)(b + 1) * nv_, X.begin() + (size_t)b * (nq_ + nv_) + nq_);
      }
      tau_res.resize((size_t)batch_ * (nv_ - 6));
      a_.resize((size_t)batch_ * nv_);
      f_.resize((size_t)batch_ * fs_ * nf_);
      resid_.resize(batch_);
      check(smpc_id_solve(h_, X.data(), tau_res.data(), a_.data(), f_.data(), resid_.data()));
    }
    void getAccelerations(std::vector<double> & a) const { a = a_; }          // [B][nv] of the last solve
    const std::vector<double> & getContactForces() const { return f_; }       // [B][nfeet][3] world frame (point feet) / [B][nfeet][6] wrenches in the foot frames (flat feet)
    const std::vector<double> & residuals() const { return resid_; }          // [B] the larger of the QP's primal / dual residuals
    smpc_id_handle * handle() { return h_; }

  protected:
    BatchedIDBase() = default;
    void create(const smpc_robot_model * robot, double control_dt, const CentroidalIDSettings & s, bool centroidal, const std::vector<double> & effort_limit,
                const std::vector<double> & velocity_limit, int batch, int device_id)
    {
      nq_ = robot->nq;
      nv_ = robot->nv;
      nf_ = robot->nfeet;
      batch_ = batch;
      if ((int)effort_limit.size() != nv_ - 6 || (int)velocity_limit.size() != nv_ - 6)
        throw std::runtime_error("effort_limit and velocity_limit must have nv - 6 entries");
      smpc_id_settings c{};
      c.friction_coefficient = s.friction_coefficient;
      c.contact_weight_ratio_max = s.contact_weight_ratio_max;
      c.contact_weight_ratio_min = s.contact_weight_ratio_min;
      c.kp_base = s.kp_base;
      c.kp_posture = s.kp_posture;
      c.kp_contact = s.kp_contact;
      c.w_base = s.w_base;
      c.w_posture = s.w_posture;
      c.w_contact_motion = s.w_contact_motion;
      c.w_contact_force = s.w_contact_force;
      c.contact_motion_equality = s.contact_motion_equality ? 1 : 0;
      c.control_dt = control_dt;
      c.effort_limit = effort_limit.data();
      c.velocity_limit = velocity_limit.data();
      c.q_min = robot->q_lo; // lower / upperPositionLimit of the actuated joints (the table is indexed by actuated joint)
      c.q_max = robot->q_hi;
      c.centroidal = centroidal ? 1 : 0;
      c.kp_com = s.kp_com;
      c.kp_feet_tracking = s.kp_feet_tracking;
      c.w_com = s.w_com;
      c.w_feet_tracking = s.w_feet_tracking;
      c.base_reference_as_coded = s.base_reference_as_coded ? 1 : 0;
      c.tsid_joint_bounds = s.tsid_joint_bounds ? 1 : 0;
      fs_ = s.force_size;
      if (fs_ != 3 && fs_ != 6)
        throw std::runtime_error("force_size must be 3 (point feet) or 6 (flat feet)");
      if (fs_ == 6 && (int)s.quad_contact_points.size() != nf_ * 12)
        throw std::runtime_error("flat feet: quad_contact_points must hold the four corners of every sole ([nfeet][4][3])");
      c.force_size = fs_;
      c.quad_contact_points = fs_ == 6 ? s.quad_contact_points.data() : nullptr;
      check(smpc_id_create(robot, &c, batch, device_id, &h_));
    }
    static void check(int rc)
    {
      if (rc < 0)
        throw std::runtime_error(smpc_last_error());
    }
    std::vector<uint8_t> flags(const std::vector<bool> & contact_state_target) const
    {
      if ((int)contact_state_target.size() != nf_)
        throw std::runtime_error("contact_state_target must have one entry per foot");
      return std::vector<uint8_t>(contact_state_target.begin(), contact_state_target.end());
    }
    smpc_id_handle * h_ = nullptr;
    int batch_ = 0, nq_ = 0, nv_ = 0, nf_ = 0, fs_ = 3;
    std::vector<double> a_, f_, resid_;
  };

  class BatchedKinodynamicsID : public BatchedIDBase
  {
  public:
    typedef KinodynamicsIDSettings Settings;
    BatchedKinodynamicsID(const smpc_robot_model * robot, double control_dt, const Settings & settings, const std::vector<double> & effort_limit,
                          const std::vector<double> & velocity_limit, int batch = 1, int device_id = 0)
    {
      CentroidalIDSettings s;
      static_cast<KinodynamicsIDSettings &>(s) = settings;
      create(robot, control_dt, s, false, effort_limit, velocity_limit, batch, device_id);
    }
    // setTarget(q_target, v_target, a_target, contact_state_target, f_target) (kinodynamics-id.cpp:120-183): one robot, or every robot
    // (instance < 0); f_target 3 per foot, world frame (point feet) / the contact wrench, 6 per foot (flat feet)
    void setTarget(const std::vector<double> & q_target, const std::vector<double> & v_target, const std::vector<double> & a_target,
                   const std::vector<bool> & contact_state_target, const std::vector<double> & f_target, int instance = -1)
    {
      if ((int)q_target.size() != nq_ || (int)v_target.size() != nv_ || (int)a_target.size() != nv_ || (int)f_target.size() != fs_ * nf_)
        throw std::runtime_error("setTarget: q (nq), v (nv), a (nv), f (force_size per foot)");
      const std::vector<uint8_t> c = flags(contact_state_target);
      check(smpc_id_set_target(h_, instance, q_target.data(), v_target.data(), a_target.data(), c.data(), f_target.data()));
    }
  };

  class BatchedCentroidalID : public BatchedIDBase
  {
  public:
    typedef CentroidalIDSettings Settings;
    BatchedCentroidalID(const smpc_robot_model * robot, double control_dt, const Settings & settings, const std::vector<double> & effort_limit,
                        const std::vector<double> & velocity_limit, int batch = 1, int device_id = 0)
    {
      create(robot, control_dt, settings, true, effort_limit, velocity_limit, batch, device_id);
    }
    // setTarget(com_position, com_velocity, feet_pose_vec, feet_velocity_vec, contact_state_target, f_target) (centroidal-id.cpp:86-147):
    // feet positions / linear velocities 3 per foot, world frame
    void setTarget(const std::vector<double> & com_position, const std::vector<double> & com_velocity, const std::vector<double> & feet_position,
                   const std::vector<double> & feet_velocity, const std::vector<bool> & contact_state_target, const std::vector<double> & f_target,
                   int instance = -1)
    {
      if (com_position.size() != 3 || com_velocity.size() != 3 || (int)feet_position.size() != 3 * nf_ || (int)feet_velocity.size() != 3 * nf_ ||
          (int)f_target.size() != fs_ * nf_)
        throw std::runtime_error("setTarget: com (3), com velocity (3), feet positions / velocities (3 per foot), forces (force_size per foot)");
      const std::vector<uint8_t> c = flags(contact_state_target);
      check(smpc_id_set_target_centroidal(h_, instance, com_position.data(), com_velocity.data(), feet_position.data(), feet_velocity.data(), c.data(),
                                          f_target.data()));
    }
  };
} // namespace simple_mpc
