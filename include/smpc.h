/*
 * smpc.h -- C ABI of the MI355X batched locomotion-MPC engine (libsmpc_hip.so).
 *
 * Drop-in boundary for the hot path of Simple-Robotics/simple-mpc: `MPC::iterate()` and what it drives
 * (reference src/mpc.cpp:189-218).  Plain pointers and sizes only; all arrays are row-major,
 * instance-major ([B][...]), IEEE double.  Host C++ (simple-mpc_amd/csrc/simple_mpc.hpp) and the
 * Python module `simple_mpc` bind these symbols; INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Every function returns 0 on success and a negative code on failure; smpc_last_error() then holds the
 * message (the reference throws std::runtime_error at the same places: src/kinodynamics.cpp:175,235,
 * 279,299; src/ocp-handler.cpp:28,32,76).
 *
 * Each entry point cites the reference interface it replaces.
 */
#ifndef SMPC_H
#define SMPC_H
#include "smpc_robot.h"
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMPC_OK 0
#define SMPC_ERR_INVALID (-1)
#define SMPC_ERR_RUNTIME (-2)
#define SMPC_ERR_NO_DEVICE (-3)

typedef struct smpc_handle smpc_handle;

/* KinodynamicsSettings: reference include/simple-mpc/kinodynamics.hpp:24-51 (same field names).
 * Matrices are dense row-major: w_x (ndx x ndx), w_u (nu x nu), w_frame (3x3), w_cent, w_centder (6x6). */
typedef struct smpc_kinodynamics_settings
{
  double timestep;
  const double * w_x;
  const double * w_u;
  const double * w_frame;
  const double * w_cent;
  const double * w_centder;
  const double * qmin;
  const double * qmax;
  double gravity[3];
  double mu;
  double Lfoot;
  double Wfoot;
  int force_size;
  int kinematics_limits;
  int force_cone;
  int land_cstr;
  /* createProblem(x0, T, force_size, gravity, terminal_constraint): the last argument (reference src/ocp-handler.cpp:96-137).
   * Non-zero adds the DCM equality com + tau vcom = com_ref at the terminal node (src/kinodynamics.cpp:366-388). */
  int terminal_constraint;
} smpc_kinodynamics_settings;

/* CentroidalSettings: reference include/simple-mpc/centroidal-dynamics.hpp:27-43 (same field names).
 * w_u (nu x nu, nu = force_size * nfeet) dense row-major; the other weights are 3 x 3. */
typedef struct smpc_centroidal_settings
{
  double timestep;
  const double * w_u;
  const double * w_com;
  const double * w_linear_mom;
  const double * w_angular_mom;
  const double * w_linear_acc;
  const double * w_angular_acc;
  double gravity[3];
  double mu;
  double Lfoot;
  double Wfoot;
  int force_size;
} smpc_centroidal_settings;

/* FullDynamicsSettings: reference include/simple-mpc/fulldynamics.hpp:28-65 (same field names).
 * w_x (ndx x ndx), w_u (nu x nu, nu = nv - 6), w_cent (6 x 6), w_forces, w_frame (force_size x force_size), dense row-major;
 * umin / umax (nu), qmin / qmax (nv - 6), Kp_correction / Kd_correction (force_size). */
typedef struct smpc_fulldynamics_settings
{
  double timestep;
  const double * w_x;
  const double * w_u;
  const double * w_cent;
  const double * w_forces;
  const double * w_frame;
  const double * umin;
  const double * umax;
  const double * qmin;
  const double * qmax;
  const double * Kp_correction;
  const double * Kd_correction;
  double gravity[3];
  double mu;
  double Lfoot;
  double Wfoot;
  int force_size;
  int torque_limits;
  int kinematics_limits;
  int force_cone;
  int land_cstr;
  int terminal_constraint; /* createProblem's last argument: DCM terminal equality (reference src/fulldynamics.cpp:433-455) */
} smpc_fulldynamics_settings;

/* MPCSettings: reference include/simple-mpc/mpc.hpp:29-49 (same field names). */
typedef struct smpc_mpc_settings
{
  double swing_apex;
  double support_force;
  double TOL;
  double mu_init;
  int max_iters;
  int num_threads; /* accepted for API compatibility; the GPU path ignores it */
  int T_fly;
  int T_contact;
  int T;
  double timestep;
} smpc_mpc_settings;

/* Built-in robot tables ("go2_like", "biped_like"); NULL if unknown.
 * Replaces RobotModelHandler(model, "standing", base) + addPointFoot
 * (reference src/robot-handler.cpp:12-60; examples/go2_kinodynamics.py:23-27). */
const smpc_robot_model * smpc_builtin_robot(const char * name);

const char * smpc_last_error(void);
/* SolverProxDDP's convergence test inside iterate (reference src/mpc.cpp:43 hands TOL to the solver, :212 runs it: `run` returns as soon
 * as the primal and dual infeasibilities are below TOL).  Off by default -- the metric of record is "at fixed ProxDDP iterations"; on: an
 * instance that is converged at the start of an iteration takes no further step in this control step (its iterate, multipliers and
 * regularisation stay; the feedback gains are those of the sweep at the converged point).  Kinodynamics and full-dynamics handles. */
int smpc_set_early_exit_on_tol(smpc_handle * h, int on);
/* number of HIP devices visible (0 => every compute entry point fails loudly) */
int smpc_device_count(void);

/* KinodynamicsOCP(settings, model) + createProblem(x_ref, T, force_size, gravity, false) + MPC(settings, ocp):
 * reference src/kinodynamics.cpp:29-38, src/ocp-handler.cpp:96-137, src/mpc.cpp:19-99.
 * `gravity_arg` is the 4th argument of createProblem (its sign convention differs between callers,
 * SURVEY App. C.8).  Performs the cold solve (<= 100 ProxDDP iterations). */
int smpc_create(
  const smpc_robot_model * robot, const smpc_kinodynamics_settings * ocp, const smpc_mpc_settings * mpc, int batch,
  double gravity_arg, int device_id, smpc_handle ** out);
/* CentroidalOCP(settings, model) + createProblem(getCentroidalState(), T, force_size, gravity, false) + MPC(settings, ocp):
 * reference src/centroidal-dynamics.cpp:27-37, src/ocp-handler.cpp:96-137, src/mpc.cpp:19-99 (SURVEY 8a rows a6, a8, a9;
 * BASELINE config "Go2 centroidal (9-dim state), H = 50").  The handle is used through the same entry points as a
 * kinodynamics handle: smpc_iterate still takes the measured MULTIBODY states [B][nq + nv] (the reference reduces them with
 * getCentroidalState, src/mpc.cpp:200); the problem state is [com; h_lin; h_ang], so xs is [B][H+1][9], us [B][H][3 nfeet],
 * K0 [B][3 nfeet][9], vs [B][H][2 nfeet] (friction-cone rows), smpc_get_state_derivative01 [B][2][9],
 * smpc_get_reference_poses the contact positions [B][H][nfeet][3], smpc_set_x_reference takes 9 doubles.
 * smpc_get_dims reports nq, nv of the robot and nx = ndx = 9.  Entry points that are specific to the kinodynamics
 * problem (debug_get_lq, debug_get_terminal, get_x_device) fail with SMPC_ERR_INVALID on such a handle. */
int smpc_create_centroidal(
  const smpc_robot_model * robot, const smpc_centroidal_settings * ocp, const smpc_mpc_settings * mpc, int batch,
  double gravity_arg, int device_id, smpc_handle ** out);
/* FullDynamicsOCP(settings, model) + createProblem(x_ref, T, force_size, gravity, false) + MPC(settings, ocp):
 * reference src/fulldynamics.cpp:30-76 (contact models, ProximalSettings(1e-9, 1e-10, 10)), :78-214 (createStage),
 * :418-455 (terminal cost), src/mpc.cpp:19-99 (BASELINE's full-dynamics configuration).  State (q, v),
 * control = the nv - 6 joint torques; nc = nu (torque box) + nv - 6 (joint box) [+ cone rows of 6-D feet].  Used through the
 * same entry points as a kinodynamics handle; smpc_set_stage_reference takes what = 2 for the contact-force references
 * (setReferenceForces, src/fulldynamics.cpp:258-300) and smpc_get_contact_forces reads MPC::getContactForces. */
int smpc_create_fulldynamics(
  const smpc_robot_model * robot, const smpc_fulldynamics_settings * ocp, const smpc_mpc_settings * mpc, int batch,
  double gravity_arg, int device_id, smpc_handle ** out);
/* MPC::getContactForces(t) for every stage (reference src/mpc.cpp:354-380): contact forces of the constrained dynamics at the
 * solution, out [B][H][nfeet][force_size], zero for feet that are not in contact at that stage.  Full-dynamics handles only. */
int smpc_get_contact_forces(smpc_handle * h, double * out);
int smpc_destroy(smpc_handle * h);

/* dims[0..7] = nq, nv, nx, ndx, nu, nc, nfeet, H */
int smpc_get_dims(const smpc_handle * h, int * dims);

/* MPC::generateCycleHorizon (reference src/mpc.cpp:101-187). contact_states: [n_states][nfeet], 0/1. */
int smpc_generate_cycle_horizon(smpc_handle * h, const uint8_t * contact_states, int n_states);
/* MPC::switchToWalk / switchToStand (reference src/mpc.cpp:382-392) */
int smpc_switch_to_walk(smpc_handle * h, const double * velocity_base6);
int smpc_switch_to_stand(smpc_handle * h);
/* One velocity command per instance, V: [B][6] (MPC::velocity_base_ is a public member the callers of the reference assign
 * directly, bindings/expose-mpc.cpp:86; a batch of robots is not steered by one joystick).  The walking / standing state is
 * unchanged.  smpc_switch_to_walk / smpc_switch_to_stand broadcast one command to every instance. */
int smpc_set_velocity_base_batched(smpc_handle * h, const double * V);
/* Per-stage references of the horizon -- the OCPHandler setters / getters (reference include/simple-mpc/ocp-handler.hpp:
 * 66-127; src/kinodynamics.cpp:154-306, src/centroidal-dynamics.cpp:120-304, src/ocp-handler.cpp:58-81), broadcast over the
 * batch; t >= horizon is the reference's "Stage index exceeds stage vector size" error.
 *   what = 0: control target (setReferenceControl; setReferenceForce(s) are segments of it), n = nu
 *   what = 1: reference state as setReferenceState / getReferenceState define it, n = nx (kinodynamics: the state_cost
 *             target; centroidal: [com_ref; v_lin; v_ang], stored as momenta m v like the reference does)
 * setPoseBase / setVelocityBase are read-modify-write of what = 1, as in the reference.  Getters return instance 0.
 * Like in the reference, the next smpc_iterate overwrites the foot references of every stage and the state target of the
 * last stage (MPC::updateStepTrackerReferences, src/mpc.cpp:278-313). */
int smpc_set_stage_reference(smpc_handle * h, int t, int what, const double * v, int n);
int smpc_get_stage_reference(smpc_handle * h, int t, int what, double * v, int n);
/* setReferencePose / getReferencePose (translation; src/kinodynamics.cpp:154-169, src/centroidal-dynamics.cpp:151-188) */
int smpc_set_reference_pose(smpc_handle * h, int t, int foot, const double * p3);
int smpc_get_reference_pose(smpc_handle * h, int t, int foot, int instance, double * p3);
/* ... with the rotation of the SE3 (R9: row-major 3 x 3; src/kinodynamics.cpp:154-170, reference tests/problem.cpp:157-160: what was set is
 * what is returned).  MPC::iterate overwrites every stage's pose with (identity rotation, Bezier position) before it solves
 * (src/mpc.cpp:303-309): so does smpc_iterate -- no solve behind this boundary ever evaluates another rotation (the stage kernels use
 * M_ref = (I, p)); a rotation set here lives until the next control step, exactly as in the reference.  smpc_set_reference_pose (translation
 * only) sets the identity rotation. */
int smpc_set_reference_pose_se3(smpc_handle * h, int t, int foot, const double * p3, const double * R9);
int smpc_get_reference_pose_se3(smpc_handle * h, int t, int foot, int instance, double * p3, double * R9);
/* getContactState(t): contact flag per foot (src/kinodynamics.cpp:343-349); getContactSupport = their sum */
int smpc_get_contact_state(smpc_handle * h, int t, uint8_t * out_nfeet);
/* MPC::getCyclingContactState(t, ee_name) for every foot: entry t of the (rotating) contact sequence the cycle horizon was generated
 * from (reference include/simple-mpc/mpc.hpp:144-147, src/mpc.cpp:103-110,230); returns the sequence length, or the length alone
 * when out is NULL.  Error before generateCycleHorizon or for t outside the sequence. */
int smpc_get_cycling_contact_state(smpc_handle * h, int t, uint8_t * out_nfeet);
/* MPC::x_reference_ (public member, reference include/simple-mpc/mpc.hpp:191) */
int smpc_set_x_reference(smpc_handle * h, const double * x_ref);

/* MPC::iterate for the whole batch (reference src/mpc.cpp:189-218). X: [B][nx] measured states (host).
 * Synchronous: returns when the solve is complete. */
int smpc_iterate(smpc_handle * h, const double * X);
/* Same with X already resident in HBM; asynchronous on the handle's stream (pair with smpc_wait). */
int smpc_iterate_device(smpc_handle * h, const double * X_device);
int smpc_wait(smpc_handle * h);
/* The hipStream_t (as void *) every launch of this handle is issued on.  A caller that produces the measured states on the device -- a
 * batched simulator, torch through torch.cuda.ExternalStream -- enqueues its own kernels there: the closed loop
 * iterate_device -> get_x_device -> (caller's kernels) -> iterate_device is then one in-order queue with no host-side wait between control
 * steps.  The stream belongs to the handle.  NULL in the CPU test build. */
void * smpc_get_stream(smpc_handle * h);
/* smpc_iterate without the final synchronisation (X must stay valid until smpc_wait): one host thread keeps several handles -- one per
 * device, each with its share of the batch -- busy at once (include/simple-mpc/batched-mpc.hpp BatchedMPCGroup). */
int smpc_iterate_async(smpc_handle * h, const double * X);
/* The small return set of a control step -- xs[1], us[0], K_0 (what MPC::iterate's caller consumes: reference examples/go2_kinodynamics.py
 * :254-292) -- of every instance of this handle as rows [x1 (nx) | u0 (nu) | K0 (nu x ndx, row-major)] of `out`, `row_doubles` (>= nx + nu
 * + nu ndx) doubles apart.  `out` is typically one slice of a single pinned host buffer that the handles of all devices fill side by
 * side ("async D2H into one pinned host buffer").  Asynchronous on the handle's stream: smpc_wait completes it.
 * Kinodynamics handles. */
int smpc_gather_outputs(smpc_handle * h, double * out, size_t row_doubles);
/* The same rows packed into a DEVICE buffer [batch][row_doubles] by one kernel on the handle's stream, for a caller that moves them itself:
 * a collective towards the process that owns the controllers (one process per device: torch.distributed gather over RCCL), a peer copy, or
 * one copy into pinned memory from a side stream so that the transfer overlaps the next control step.  Kinodynamics handles. */
int smpc_gather_outputs_device(smpc_handle * h, double * out_device, size_t row_doubles);
/* ... and into a buffer on ANOTHER device of the node (hipMemcpyPeerAsync over xGMI on the handle's stream, after the pack kernel): one
 * process, one handle per device, every device's rows gathered on the device that runs the controllers.  `out_peer` points to
 * [batch][nx + nu + nu ndx] doubles on device `dst_device`, rows contiguous.  Kinodynamics handles. */
int smpc_gather_outputs_peer(smpc_handle * h, double * out_peer, int dst_device);
/* Checkpoint / resume (the reference has none; a batched simulator needs it to roll back or migrate a batch).
 * The state is everything a later smpc_iterate depends on: iterate, multipliers, swing trajectories, references, velocity
 * commands, gait bookkeeping -- not the feedback gains of the last solve (the next iterate recomputes them).
 *   smpc_state_size   bytes needed for this handle (it grows with the gait cycle: call it after generateCycleHorizon)
 *   smpc_save_state   writes at most `capacity` bytes to `buffer` (host), returns the number written through *written
 *   smpc_load_state   restores; the buffer must come from a handle of the same kind, batch, horizon and robot
 * After smpc_load_state the handle continues bit-identically to the handle the state was saved from. */
int smpc_state_size(smpc_handle * h, size_t * bytes);
int smpc_save_state(smpc_handle * h, void * buffer, size_t capacity, size_t * written);
int smpc_load_state(smpc_handle * h, const void * buffer, size_t size);
/* xs_[t] of every instance into a dense device buffer [B][nx]; asynchronous on the handle's stream.
 * Lets a closed loop keep the measured states resident in HBM (x_meas = xs[1] + noise). */
int smpc_get_x_device(smpc_handle * h, int t, double * out_device);

/* MPC::xs_ / us_ / Ks_ (reference include/simple-mpc/mpc.hpp:187-189; filled at src/mpc.cpp:215-217).
 * out: xs [B][H+1][nx], us [B][H][nu], K0 [B][nu][ndx], Ks [B][H][nu][ndx] */
int smpc_get_xs(smpc_handle * h, double * out);
int smpc_get_us(smpc_handle * h, double * out);
int smpc_get_K0(smpc_handle * h, double * out);
int smpc_get_Ks(smpc_handle * h, double * out);
/* solver multipliers (results_.vs / results_.lams): vs [B][H][nc], lams [B][H+1][ndx] (lams[0] = 0) */
int smpc_get_vs(smpc_handle * h, double * out);
/* (tests) multipliers of the optional rows of a kinodynamics handle: which = 0 friction-cone rows [B][H][2 nfeet],
 * 1 land rows [B][H][nfeet] */
int smpc_debug_get_extra_multipliers(smpc_handle * h, int which, double * out);
int smpc_get_lams(smpc_handle * h, double * out);
/* MPC::getStateDerivative(t) for t = 0,1 (reference src/mpc.cpp:346-352). out: [B][2][2 nv] */
int smpc_get_state_derivative01(smpc_handle * h, double * out);
/* MPC::getReferencePose(t, foot).translation() for all t, feet (reference src/mpc.cpp:336-339). out: [B][H][nfeet][3] */
int smpc_get_reference_poses(smpc_handle * h, double * out);
/* MPC::foot_takeoff_times_ / foot_land_times_ (reference include/simple-mpc/mpc.hpp:184-185). which: 0 takeoff, 1 land.
 * Returns the number of entries (<= cap copied). */
int smpc_get_foot_timing(smpc_handle * h, int foot, int which, int * out, int cap);
/* per-instance solver scalars of the last iteration, [B][16]:
 * phi0, dphi0, alpha, phi_new, prim_infeas, dual_infeas, ls_failed, preg, prim_new, cost, cost_new, ls_index */
int smpc_get_info(smpc_handle * h, double * out);
/* Per-instance status word of the last control step (the reference ignores the solver's return value, src/mpc.cpp:212; a batch needs to
 * know which of its members went wrong), out [B]: bit 0 = a non-finite number among the solver scalars (the instance's trajectory
 * must not be used), bit 1 = the last line search failed (the smallest step was taken), bit 2 = the primal regularisation has reached
 * its upper limit.  Returns the number of instances with a non-zero word, or a negative error code. */
#define SMPC_STATUS_NONFINITE 1
#define SMPC_STATUS_LS_FAILED 2
#define SMPC_STATUS_REG_SATURATED 4
int smpc_get_status(smpc_handle * h, int * out);
/* cold-solve trace of the constructor: returns n iterations; out [n][4] = phi0, prim, dual, alpha (cap rows) */
int smpc_get_cold_trace(smpc_handle * h, double * out, int cap);

/* test / debug access to one LQ knot (stage t of instance inst) as assembled by the stage kernel in the
 * LAST iteration: out must hold smpc_lq_size() doubles, laid out A,B,Q,S,R,C,q,r,f,d,lx,lu,lpd,vpd */
int smpc_lq_size(const smpc_handle * h);
int smpc_debug_get_lq(smpc_handle * h, int inst, int t, double * out);
/* last line-search steps: dxs [B][H+1][ndx], dus [B][H][nu] */
int smpc_debug_get_steps(smpc_handle * h, double * dxs, double * dus);
/* terminal node of the last iteration for one instance: QN [ndx][ndx], qN [ndx] */
int smpc_debug_get_terminal(smpc_handle * h, int inst, double * QN, double * qN);

/* in-kernel phase timers of the Riccati sweep (shader cycles, block 0 only); needs SMPC_PHASE_PROFILE=1
 * in the environment at smpc_create time. out: 64 doubles */
int smpc_debug_get_phase_cycles(smpc_handle * h, double * out64);

/* profiling: when enabled every kernel launch is bracketed by HIP events on the handle's stream.
 * smpc_get_kernel_times: ms[9], calls[9] for recede, deriv, riccati, forward, trial, select, apply, tree, tree_ls (the lane-per-problem tree
 * pass that precedes the derivative resp. the line-search kernel of a kinodynamics handle). */
int smpc_set_profiling(smpc_handle * h, int enabled);
/* number of kernel slots this library reports (9 today; grows when kernels are added) */
int smpc_kernel_time_slots(void);
/* the first min(n, smpc_kernel_time_slots()) slots into ms[n], calls[n].  Centroidal handles: frontend, step (6-D feet: recede), deriv, riccati,
 * forward, line search, and for 6-D feet the candidate kernel of the line search in the slot behind them. */
int smpc_get_kernel_times_n(smpc_handle * h, double * ms, long * calls, int n);
/* the same without a capacity: writes smpc_kernel_time_slots() entries -- size the arrays with that call, or use the _n form */
int smpc_get_kernel_times(smpc_handle * h, double * ms, long * calls);
int smpc_reset_kernel_times(smpc_handle * h);

/* ---- interpolation between MPC knots (replaces the Interpolator class,
 *      reference include/simple-mpc/interpolator.hpp, src/interpolator.cpp:5-78) ----
 * smpc_interpolate: batched targets for the whole-body controller that follows the MPC, from the solution held by the
 *   handle, as the reference examples compute them (examples/go2_kinodynamics.py:276-284):
 *     x_out     [B][nx]    interpolateState over xs[0 .. knots-1] (configuration on the manifold, velocity linear)
 *     acc_out   [B][nv]    interpolateLinear over getStateDerivative(t)[nv:] with the joint part replaced by
 *                          us[t][3 nf:], t = 0, 1
 *     force_out [B][3 nf]  interpolateLinear over us[t][: 3 nf], t = 0, 1
 *   delay >= 0 (seconds after the last smpc_iterate); beyond the last interval the last knot is returned, like the
 *   reference.  Any output pointer may be NULL.  Host buffers.
 *   Centroidal handle (examples/talos_centroidal.py: interpolateLinear over states, state derivatives and forces):
 *     x_out [B][9], acc_out [B][9] = interpolated getStateDerivative, force_out [B][3 nf].
 * smpc_interpolate_knots: the Interpolator methods on explicit host knot lists [n][dim]:
 *     kind 0 interpolateState (dim = nq + nv), 1 interpolateConfiguration (dim = nq), 2 interpolateLinear (any dim).
 *   Errors mirror the reference's assertions ("State is not of the right size"). */
int smpc_interpolate(smpc_handle * h, double delay, int knots, double * x_out, double * acc_out, double * force_out);

/* ---- full-dynamics model, first block: constrained forward dynamics of n states -- replaces
 *      pinocchio::constraintDynamics as called by Aligator's MultibodyConstraintFwdDynamics for the contacts
 *      FullDynamicsOCP builds (reference src/fulldynamics.cpp:39,50-75,139): CONTACT_3D, LOCAL frame, Baumgarte corrector
 *      Kp / Kd [3] (NULL = 0), actuation [0; I].  X [n][nq + nv], tau [n][nv - 6], contact_mask [n] (bit f = foot f in
 *      contact), all host.  Outputs (host): a_out [n][nv]; lambda_out [n][3 nfeet] contact forces ON the robot in the contact
 *      frames, the feet in contact first (in foot order), remaining entries 0; iters_out [n] proximal iterations (may be
 *      NULL); kernel_ms: wall time of the launch (may be NULL).  prox_accuracy / prox_mu / prox_max_iter <= 0 select the
 *      reference's ProximalSettings(1e-9, 1e-10, 10).  n need not be the handle's batch size; the handle supplies the
 *      robot table and gravity.  Kinodynamics handles (the quadruped, CONTACT_3D) and -- round 4 -- full-dynamics handles of either
 *      robot: the contact model is the handle's (3-D LOCAL point feet, or 6-D LOCAL_WORLD_ALIGNED flat feet with Kp / Kd [6] and
 *      lambda_out [n][6 nfeet] contact wrenches). */
int smpc_full_forward_dynamics(
  smpc_handle * h, int n, const double * X, const double * tau, const unsigned * contact_mask, const double * Kp,
  const double * Kd, double prox_accuracy, double prox_mu, int prox_max_iter, double * a_out, double * lambda_out,
  int * iters_out, double * kernel_ms);

/* ---- state feedback front-end (replaces RobotDataHandler::updateInternalData(x, false) and
 *      getCentroidalState, reference src/robot-handler.cpp:106-127,142-149) for a batch of measured multibody states
 *      X [B][nx] (host): feet [B][nf][3] foot positions (world), com [B][3], hg [B][6] centroidal momentum
 *      [linear; angular about the CoM], centroidal_state [B][9] = [com; h_lin; h_ang].  Any output may be NULL.
 *      Every kind of handle: the kinodynamics, centroidal and full-dynamics OCPs of the quadruped and of the biped. */
int smpc_update_internal_data(smpc_handle * h, const double * X, double * feet, double * com, double * hg, double * centroidal_state);
/* Riccati feedback application between MPC knots (reference examples/go2_fulldynamics.py:271-285):
 *   u_out[b] = interpolateLinear(us)[b] - Ks[0][b] * difference(X_meas[b], interpolateState(xs)[b])
 * X_meas [B][nx], u_out [B][nu] (host).  Centroidal handle: X_meas are still the measured multibody states [B][nq + nv];
 * the feedback acts on their centroidal state, u_out[b] = f(d) - K0 (x(d) - getCentroidalState(X_meas[b])). */
int smpc_riccati_feedback(smpc_handle * h, double delay, const double * X_meas, double * u_out);
int smpc_interpolate_knots(int kind, double delay, double timestep, const double * knots, int n, int dim, double * out, int device_id);

/* ---- friction compensation (replaces FrictionCompensation::computeFriction, reference
 *      src/friction-compensation.cpp:22-37): torque[b][j] += viscous[j] * velocity[b][j] + dry[j] * sign(velocity[b][j])
 *      for a batch of joint velocity / torque vectors of size nu (host buffers, torque in / out).  velocity_size /
 *      torque_size are the per-instance vector lengths the caller holds: a mismatch with nu is the reference's
 *      "Velocity has wrong size" / "Torque has wrong size" error. */
int smpc_friction_compensation(
  const double * dry, const double * viscous, int nu, const double * velocity, int velocity_size, double * torque, int torque_size,
  int batch, int device_id);

/* ---- centroidal model (replaces CentroidalFwdDynamics + IntegratorEuler as composed in
 *      reference src/centroidal-dynamics.cpp:79-81; the centroidal state itself comes from smpc_update_internal_data) ----
 * Batched forward step with derivatives, 3-D contact forces:
 *     x = [c; h; L] (CoM, linear momentum, angular momentum about the CoM),  u = [f_1 .. f_nfeet]
 *     xdot = [h / m ;  m g + sum_{contact} f_i ;  sum_{contact} (p_i - c) x f_i],   Xnext = x + timestep * xdot
 *     A = d Xnext / dx  [B][9][9],   B = d Xnext / du  [B][9][3 nfeet]   (row-major; columns of feet in the air are zero)
 * X [B][9], U [B][3 nfeet], contact [B][nfeet] (0 / 1), contact_pos [B][nfeet][3]; host buffers; A and B may be NULL. */
int smpc_centroidal_dynamics(
  double mass, const double * gravity, double timestep, int nfeet, const double * X, const double * U, const unsigned char * contact,
  const double * contact_pos, int batch, double * Xnext, double * A, double * B, int device_id);

/* ---- whole-body inverse-dynamics QP: KinodynamicsID (reference include/simple-mpc/inverse-dynamics/kinodynamics-id.hpp:17-92,
 *      src/inverse-dynamics/kinodynamics-id.cpp:7-237), batched: one QP per robot and control tick, 3-D point feet.
 *      Field names of KinodynamicsID::Settings; the limits the reference reads from the pinocchio model (effortLimit, velocityLimit,
 *      lower / upperPositionLimit of the actuated joints) are passed explicitly; admm_* are the solver's own (0 = defaults). ---- */
typedef struct smpc_id_settings
{
  double friction_coefficient;
  double contact_weight_ratio_max;
  double contact_weight_ratio_min;
  double kp_base, kp_posture, kp_contact;
  double w_base, w_posture, w_contact_motion, w_contact_force; /* <= 0: task disabled, as in the reference */
  int contact_motion_equality;
  double control_dt;
  const double * effort_limit;   /* nv - 6 */
  const double * velocity_limit; /* nv - 6 */
  const double * q_min;          /* nv - 6 */
  const double * q_max;          /* nv - 6 */
  int admm_iters;                /* cap on the ADMM iterations per solve (default 400; residuals are checked every 20 iterations and the loop
                                    stops below 1e-7; warm-started from the previous tick) */
  double admm_rho, admm_sigma, admm_alpha; /* defaults 0.1, 1e-6, 1.6 */
  double admm_tol;               /* 0: default 1e-7 ; < 0: never stop early (exactly admm_iters iterations) */
  /* CentroidalID::Settings (include/simple-mpc/inverse-dynamics/centroidal-id.hpp): centroidal != 0 selects that controller -- the base
   * task keeps its orientation rows, a centre-of-mass task and a position-tracking task per foot out of contact are added; the posture
   * and base targets stay at the reference state (centroidal-id.cpp:60-84) */
  int centroidal;
  double kp_com, kp_feet_tracking;
  double w_com, w_feet_tracking; /* <= 0: task disabled */
  /* The reference AS CODED (src/inverse-dynamics/kinodynamics-id.cpp:222-223: setDerivative is called twice, so the base acceleration
   * target becomes the velocity reference of the base task and its acceleration reference stays zero).  0 (default): velocity and
   * acceleration references from the respective targets -- with zero base targets, as in the reference's tests, both coincide. */
  int base_reference_as_coded;
  /* TSID's TaskJointPosVelAccBounds in full (time step 2 control_dt, braking-distance position bounds, viability bounds with the default
   * acceleration limit; setImposeBounds(true, true, true, false) of kinodynamics-id.cpp:80-88).  0 (default): position / velocity limits
   * as acceleration bounds over one control period. */
  int tsid_joint_bounds;
  /* Robots with flat (QUAD) feet -- RobotModelHandler::addQuadFoot, tsid::contacts::Contact6d (kinodynamics-id.cpp:41-49, 163-167, 204-208):
   * force_size 6 and the four corners of every sole in its foot frame, quad_contact_points [nfeet][4][3] (getQuadFootContactPoints).  The QP then
   * holds the forces at the corners (12 per foot, foot frame), the 6-D LOCAL contact motion, a friction pyramid per corner and the bound on the
   * total normal force; force targets and reported contact forces are 6-D wrenches per foot (foot frame).  0 or 3: point feet. */
  int force_size;
  const double * quad_contact_points;
} smpc_id_settings;
typedef struct smpc_id_handle smpc_id_handle;
/* KinodynamicsID(model_handler, control_dt, settings): the default target is the reference state, every foot in contact with an equal
 * share of the weight (kinodynamics-id.cpp:96-112). */
int smpc_id_create(const smpc_robot_model * robot, const smpc_id_settings * settings, int batch, int device_id, smpc_id_handle ** out);
void smpc_id_destroy(smpc_id_handle * h);
/* setTarget(q, v, a, contact_state, f) (kinodynamics-id.cpp:120-183) of one instance, or of every instance (instance < 0):
 * q (nq), v (nv), a (nv), contact flag per foot, f (3 per foot, world frame; flat feet: the 6-D wrench per foot, foot frame -- wherever
 * this section says "3 per foot" / "3 nfeet" for a force, a flat-foot handle takes 6) */
int smpc_id_set_target(smpc_id_handle * h, int instance, const double * q, const double * v, const double * a, const uint8_t * contact, const double * f);
/* one target per robot of the batch: Q [B][nq], V [B][nv], A [B][nv], contact [B][nfeet], F [B][3 nfeet] */
int smpc_id_set_targets(smpc_id_handle * h, const double * Q, const double * V, const double * A, const uint8_t * contact, const double * F);
/* CentroidalID::setTarget(com_position, com_velocity, feet_pose_vec, feet_velocity_vec, contact_state_target, f_target)
 * (centroidal-id.cpp:86-147) of one instance or of every instance (instance < 0): com, vcom (3), feet_p, feet_v (3 per foot, world
 * frame: the translations / linear velocities of the reference's SE3 / Motion arguments), contact flag per foot, f (3 per foot).
 * SMPC_ERR_INVALID on a KinodynamicsID handle. */
int smpc_id_set_target_centroidal(smpc_id_handle * h, int instance, const double * com, const double * vcom, const double * feet_p,
                                  const double * feet_v, const uint8_t * contact, const double * f);
/* one target per robot: COM, VCOM [B][3], FEET_P, FEET_V [B][3 nfeet], contact [B][nfeet], F [B][3 nfeet] */
int smpc_id_set_targets_centroidal(smpc_id_handle * h, const double * COM, const double * VCOM, const double * FEET_P, const double * FEET_V,
                                   const uint8_t * contact, const double * F);
/* solve(t, q_meas, v_meas, tau) + getAccelerations for the batch (kinodynamics-id.cpp:185-237): X [B][nq + nv] (host) ->
 * tau [B][nv - 6], a [B][nv] (may be NULL), f [B][3 nfeet] contact forces of the solution (may be NULL), resid [B] the larger of the
 * QP's primal / dual residuals (may be NULL) */
int smpc_id_solve(smpc_id_handle * h, const double * X, double * tau, double * a, double * f, double * resid);
/* Same with the states resident in HBM: X_device [B][nq + nv]; asynchronous on the handle's stream (pair with smpc_id_wait).  tau_device
 * [B][nv - 6] may be NULL: the torques then stay in the handle's own buffer, smpc_id_get_tau_device. */
int smpc_id_solve_device(smpc_id_handle * h, const double * X_device, double * tau_device);
int smpc_id_wait(smpc_id_handle * h);
/* residuals of the last solve, resid [B] (the larger of the QP's primal / dual residuals; not finite: the solve of that robot failed,
 * its warm start was dropped and its next solve starts from scratch); joins the handle's stream */
int smpc_id_get_resid(smpc_id_handle * h, double * resid);
/* forget the warm start (ADMM iterate, step-size parameter) of one robot, or of every robot (instance < 0) */
int smpc_id_reset(smpc_id_handle * h, int instance);
const double * smpc_id_get_tau_device(smpc_id_handle * h);
/* the handle's own state buffer [B][nq + nv] in HBM (smpc_id_solve copies the host states there; a simulator may keep its states in it) */
double * smpc_id_get_x_device(smpc_id_handle * h);
/* The 1 kHz loop of the reference's examples (examples/go2_kinodynamics.py:264-300) without host round trips, for a kinodynamics MPC handle
 * and a KinodynamicsID handle of the same batch on the same device:
 *   smpc_id_set_targets_from_mpc  interpolateState / interpolateLinear of the MPC's solution at `delay` seconds after its last iterate
 *                                 (as smpc_interpolate, `knots` as there) written straight into the controller's target buffers; the
 *                                 contact flags are those of the MPC's stage 0.  Ordered after the MPC's work, before the next solve.
 *                                 A centroidal MPC handle feeds a CentroidalID handle the same way (examples/talos_centroidal.py:218-243):
 *                                 centre of mass, its velocity (momentum / mass), the foot references between stages 0 and 1 and
 *                                 their velocities, the interpolated forces.
 *   smpc_sim_step_device          one step of a simulated batch: constrained forward dynamics of the feet in contact (flags per foot;
 *                                 Baumgarte gains Kp, Kd [3], NULL = 0; ProximalSettings of record) under torques tau_device, then
 *                                 semi-implicit Euler over dt; X_device [B][nq + nv] is updated in place.  Asynchronous on the MPC
 *                                 handle's stream (smpc_wait joins).  Kinodynamics handles of the quadruped, and full-dynamics handles of
 *                                 either robot (their own contact model: Kp, Kd of force_size entries).
 * Round 4: the MPC handle of smpc_id_set_targets_from_mpc / smpc_id_share_stream may be of any kind -- the kinodynamics OCP of the biped
 * with flat feet and the full-dynamics OCPs feed a KinodynamicsID controller of the same robot (states, accelerations, contact forces /
 * wrenches of stage 0 .. 1 interpolated on the device). */
int smpc_id_set_targets_from_mpc(smpc_id_handle * id, smpc_handle * mpc, double delay, int knots);
/* From now on the controller issues its work on the MPC handle's stream (kinodynamics or centroidal handle; NULL: back to its own):
 * MPC step, targets, QP solves and simulator steps then form one in-order queue -- smpc_id_wait / smpc_wait are needed only before the
 * host reads a result, not between the legs of a tick.  The stream belongs to the MPC handle: go back (NULL) before that handle is
 * destroyed. */
int smpc_id_share_stream(smpc_id_handle * id, smpc_handle * mpc);
int smpc_sim_step_device(smpc_handle * h, double * X_device, const double * tau_device, const uint8_t * contact, const double * Kp, const double * Kd, double dt);
/* (tests) intermediate results of the last solve, padded layouts of simple-mpc_amd/csrc/smpc_id.h: what = 0 M, 1 nle, 2 J, 3 dJ v, 4 foot
 * velocities, 5 H [32][32], 6 g [32], 7 C [80][32], 8 l [80], 9 u [80], 10 centre of mass [3], 11 foot positions [3 nfeet], 12 torques [nv - 6]; every one [B][...] */
int smpc_id_debug_get(smpc_id_handle * h, int what, double * out);

#ifdef __cplusplus
}
#endif
#endif
