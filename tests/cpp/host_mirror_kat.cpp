// C++ host-side test of the drop-in boundary, written like the reference's own tests/mpc.cpp:95-190 (mpc_kinodynamics):
// build the settings structs, construct the MPC through the header mirror include/simple-mpc/batched-mpc.hpp (which only
// speaks the C ABI of include/smpc.h), generate the cycle horizon of the reference test, check the container sizes
// (tests/mpc.cpp:43-44) and the foot-timing known answers (tests/mpc.cpp:78-81, :87-90), iterate, check the outputs.
// Linked against libsmpc_hip.so on the GPU tier and against the sequential-lane test build on the CPU tier.
#include "simple-mpc/batched-id.hpp"
#include "simple-mpc/batched-mpc.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>

#define CHECK(cond)                                                                                                    \
  do                                                                                                                   \
  {                                                                                                                    \
    if (!(cond))                                                                                                       \
    {                                                                                                                  \
      std::fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond);                                          \
      return 1;                                                                                                        \
    }                                                                                                                  \
  } while (0)

using namespace simple_mpc;

int main()
{
  const smpc_robot_model * robot = smpc_builtin_robot("go2_like");
  CHECK(robot != nullptr);
  const int nv = robot->nv, nq = robot->nq, nf = robot->nfeet;
  CHECK(nv == 18 && nq == 19 && nf == 4);

  // KinodynamicsSettings of the reference example (examples/go2_kinodynamics.py:42-85)
  KinodynamicsSettings ks;
  const int ndx = 2 * nv, nu = nv - 6 + 3 * nf;
  const double wbp[6] = {0, 0, 100, 10, 10, 0};
  ks.w_x.assign((size_t)ndx * ndx, 0.0);
  for (int i = 0; i < 6; i++)
    ks.w_x[(size_t)i * ndx + i] = wbp[i];
  for (int i = 6; i < nv; i++)
    ks.w_x[(size_t)i * ndx + i] = 1.0;
  for (int i = 0; i < 6; i++)
    ks.w_x[(size_t)(nv + i) * ndx + nv + i] = 10.0;
  for (int i = 6; i < nv; i++)
    ks.w_x[(size_t)(nv + i) * ndx + nv + i] = 0.1;
  ks.w_u.assign((size_t)nu * nu, 0.0);
  for (int i = 0; i < nu; i++)
    ks.w_u[(size_t)i * nu + i] = i < 3 * nf ? 0.01 : 1e-5;
  ks.w_frame = {2000, 0, 0, 0, 2000, 0, 0, 0, 2000};
  const double wc[6] = {0, 0, 1, 0.1, 0.1, 10}, wcd[6] = {0, 0, 0, 0.1, 0.1, 0.1};
  ks.w_cent.assign(36, 0.0);
  ks.w_centder.assign(36, 0.0);
  for (int i = 0; i < 6; i++)
  {
    ks.w_cent[i * 7] = wc[i];
    ks.w_centder[i * 7] = wcd[i];
  }
  ks.qmin.assign(robot->q_lo, robot->q_lo + (nv - 6)); // joint limits, index = v index - 6
  ks.qmax.assign(robot->q_hi, robot->q_hi + (nv - 6));
  ks.kinematics_limits = true;
  ks.force_size = 3;

  MPCSettings ms; // reference tests/mpc.cpp:112-125
  ms.max_iters = 1;
  ms.support_force = robot->total_mass * 9.81;
  ms.TOL = 1e-6;
  ms.mu_init = 1e-8;
  ms.num_threads = 8;
  ms.swing_apex = 0.1;
  ms.T_fly = 80;
  ms.T_contact = 20;
  ms.T = 100;
  ms.timestep = 0.01;

  BatchedMPC mpc(robot, ks, ms, /*batch=*/2);
  CHECK(mpc.horizon() == 100 && mpc.nx() == nq + nv && mpc.nu() == nu);

  // gait of the reference test (tests/mpc.cpp:130-165): feet 0 / 1 carry the left / right patterns, feet 2 / 3 mirror them
  std::vector<std::map<std::string, bool>> contact_states;
  auto push = [&](int n, bool left, bool right) {
    for (int i = 0; i < n; i++)
      contact_states.push_back({{robot->foot_name[0], left}, {robot->foot_name[1], right}, {robot->foot_name[2], left}, {robot->foot_name[3], right}});
  };
  push(10, true, true);
  push(50, true, false);
  push(10, true, true);
  push(50, false, true);
  mpc.generateCycleHorizon(contact_states);
  CHECK(mpc.getFootTakeoffCycle(robot->foot_name[0])[0] == 170); // tests/mpc.cpp:78-81
  CHECK(mpc.getFootTakeoffCycle(robot->foot_name[1])[0] == 110);
  CHECK(mpc.getFootLandCycle(robot->foot_name[0])[0] == 219);
  CHECK(mpc.getFootLandCycle(robot->foot_name[1])[0] == 160);

  std::vector<double> X((size_t)2 * mpc.nx(), 0.0);
  for (int b = 0; b < 2; b++)
    for (int i = 0; i < nq; i++)
      X[(size_t)b * mpc.nx() + i] = robot->q_ref[i];
  mpc.iterate(X);
  {
    // standing at the reference posture, all feet in contact: the contact forces carry the weight
    double fz = 0.0;
    for (int f = 0; f < nf; f++)
      fz += mpc.us_[3 * f + 2];
    CHECK(std::fabs(fz - robot->total_mass * 9.81) < 0.10 * robot->total_mass * 9.81); // (H = 100 cold start stalls at 6 %)
  }
  for (int it = 1; it < 10; it++)
    mpc.iterate(X);
  CHECK((int)mpc.xs_.size() == 2 * (100 + 1) * mpc.nx()); // xs_.size() == T + 1 per instance (tests/mpc.cpp:43)
  CHECK((int)mpc.us_.size() == 2 * 100 * mpc.nu());       // us_.size() == T     (tests/mpc.cpp:44)
  CHECK(mpc.getFootTakeoffCycle(robot->foot_name[0])[0] == 160); // tests/mpc.cpp:87-90
  CHECK(mpc.getFootTakeoffCycle(robot->foot_name[1])[0] == 100);
  CHECK(mpc.getFootLandCycle(robot->foot_name[0])[0] == 209);
  CHECK(mpc.getFootLandCycle(robot->foot_name[1])[0] == 150);
  for (double v : mpc.xs_)
    CHECK(std::isfinite(v));
  {
    std::vector<int> words;
    CHECK(mpc.status(words) == 0 && words.size() == 2 && words[0] == 0 && words[1] == 0);
  }
  // both instances got the same measured state: identical solutions
  const size_t half = mpc.xs_.size() / 2;
  for (size_t i = 0; i < half; i++)
    CHECK(mpc.xs_[i] == mpc.xs_[half + i]);
  {
    // the batch sharded over two handles ("devices": both may be device 0) + the gathered return set [x1 | u0 | K0] in ONE host buffer:
    // the same rows as a single handle's outputs, instance by instance (no collective on the solve path)
    MPCSettings gs = ms;
    gs.T = 12;
    BatchedMPCGroup grp(robot, ks, gs, /*batch=*/3, std::vector<int>{0, 0});
    BatchedMPC one(robot, ks, gs, 3);
    grp.generateCycleHorizon(contact_states);
    one.generateCycleHorizon(contact_states);
    CHECK(grp.parts() == 2 && grp.firstInstance(1) == 1 && grp.gatherRow() == one.nx() + one.nu() + one.nu() * one.ndx());
    std::vector<double> X3((size_t)3 * one.nx(), 0.0);
    for (int b = 0; b < 3; b++)
    {
      for (int i = 0; i < nq; i++)
        X3[(size_t)b * one.nx() + i] = robot->q_ref[i];
      X3[(size_t)b * one.nx() + 2] += 0.004 * b; // three different measured heights
    }
    std::vector<double> rows((size_t)3 * grp.gatherRow(), -1.0);
    for (int it = 0; it < 3; it++)
    {
      grp.iterate(X3.data(), rows.data());
      one.iterate(X3);
    }
    const int row = grp.gatherRow(), nx1 = one.nx(), nu1 = one.nu(), H1 = one.horizon();
    for (int b = 0; b < 3; b++)
    {
      for (int i = 0; i < nx1; i++)
        CHECK(rows[(size_t)b * row + i] == one.xs_[((size_t)b * (H1 + 1) + 1) * nx1 + i]);
      for (int i = 0; i < nu1; i++)
        CHECK(rows[(size_t)b * row + nx1 + i] == one.us_[(size_t)b * H1 * nu1 + i]);
      for (int i = 0; i < nu1 * one.ndx(); i++)
        CHECK(rows[(size_t)b * row + nx1 + nu1 + i] == one.K0_[(size_t)b * nu1 * one.ndx() + i]);
    }
  }
  // errors surface as std::runtime_error, as in the reference
  bool threw = false;
  try
  {
    mpc.iterate(std::vector<double>(3, 0.0));
  }
  catch (const std::runtime_error &)
  {
    threw = true;
  }
  CHECK(threw);
  {
    // checkpoint / resume through the header mirror: the restored handle repeats the step bit for bit
    const std::vector<unsigned char> ck = mpc.saveState();
    mpc.iterate(X);
    const std::vector<double> xs_a = mpc.xs_;
    mpc.loadState(ck);
    mpc.iterate(X);
    CHECK(xs_a == mpc.xs_);
    // setter / getter round trips of tests/problem.cpp:157-195 through the mirror
    const double pl[3] = {1, 0, 2};
    mpc.setReferencePose(4, robot->foot_name[0], pl);
    CHECK(mpc.getReferencePose(4, robot->foot_name[0])[2] == 2.0);
    std::vector<double> ur = mpc.getReferenceControl(3);
    ur[1] = 1.0;
    mpc.setReferenceControl(3, ur);
    CHECK(mpc.getReferenceControl(3)[1] == 1.0);
    CHECK(mpc.getContactState(2).size() == 4 && mpc.getContactState(2)[0]);
    bool threw3 = false;
    try
    {
      mpc.getReferenceState(100);
    }
    catch (const std::runtime_error &)
    {
      threw3 = true;
    }
    CHECK(threw3);
    std::vector<double> V(12, 0.0);
    V[0] = 0.3;
    V[6 + 5] = 0.4;
    mpc.setVelocityBaseBatched(V);
    mpc.iterate(X);
    bool differ = false;
    const size_t hh = mpc.xs_.size() / 2;
    for (size_t i = 0; i < hh; i++)
      differ = differ || mpc.xs_[i] != mpc.xs_[hh + i];
    CHECK(differ); // two commands, two plans
  }
  // ---- centroidal OCP, written like the reference's mpc_centroidal test (tests/mpc.cpp:172-258) ----
  {
    CentroidalSettings cs; // tests/test_utils.cpp:194-218 with 3-D forces
    const int nuc = 3 * nf;
    cs.w_u.assign((size_t)nuc * nuc, 0.0);
    for (int i = 0; i < nuc; i++)
      cs.w_u[(size_t)i * nuc + i] = 1.0;
    cs.w_com.assign(9, 0.0);
    cs.w_linear_mom = {0.01, 0, 0, 0, 0.01, 0, 0, 0, 100};
    cs.w_angular_mom = {0.1, 0, 0, 0, 0.1, 0, 0, 0, 1000};
    cs.w_linear_acc = {0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01};
    cs.w_angular_acc = {0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01};
    cs.force_size = 3;
    BatchedMPC cmpc(robot, cs, ms, /*batch=*/2);
    CHECK(cmpc.horizon() == 100 && cmpc.nx() == 9 && cmpc.nu() == nuc && cmpc.nx_in() == nq + nv);
    cmpc.generateCycleHorizon(contact_states);
    CHECK(cmpc.getFootTakeoffCycle(robot->foot_name[0])[0] == 170);
    for (int it = 0; it < 10; it++)
      cmpc.iterate(X); // the multibody state, as in the reference test (x_multibody)
    CHECK((int)cmpc.xs_.size() == 2 * 101 * 9 && (int)cmpc.us_.size() == 2 * 100 * nuc);
    CHECK(cmpc.getFootTakeoffCycle(robot->foot_name[0])[0] == 160);
    const std::vector<double> xdot = cmpc.getStateDerivative(0); // tests/mpc.cpp:257
    CHECK((int)xdot.size() == 2 * 9);
    for (double v : cmpc.xs_)
      CHECK(std::isfinite(v));
    for (double v : xdot)
      CHECK(std::isfinite(v));
    // xdot = [h / m ; m g + sum f ; ...]: linear-momentum rate consistent with the first control
    double fz = 0.0;
    for (int f = 0; f < nf; f++)
      fz += cmpc.us_[3 * f + 2];
    CHECK(std::fabs(xdot[5] - (fz - robot->total_mass * 9.81)) < 1e-9 * (1.0 + std::fabs(fz)));
    bool threw2 = false;
    try
    {
      CentroidalSettings bad = cs;
      bad.w_com.assign(4, 0.0);
      BatchedMPC nope(robot, bad, ms, 1);
    }
    catch (const std::runtime_error &)
    {
      threw2 = true;
    }
    CHECK(threw2);
  }
  // ---- full-dynamics OCP, written like the reference's mpc_fulldynamics test (tests/mpc.cpp:20-91) with the Go2 settings of
  // examples/go2_fulldynamics.py:42-77; a short horizon keeps the CPU build of the kernels quick ----
  {
    FullDynamicsSettings fs;
    const int nuf = nv - 6;
    fs.w_x.assign((size_t)ndx * ndx, 0.0);
    for (int i = 6; i < nv; i++)
      fs.w_x[(size_t)i * ndx + i] = 1.0;
    for (int i = nv; i < nv + 6; i++)
      fs.w_x[(size_t)i * ndx + i] = 10.0;
    for (int i = nv + 6; i < ndx; i++)
      fs.w_x[(size_t)i * ndx + i] = 0.1;
    fs.w_u.assign((size_t)nuf * nuf, 0.0);
    for (int i = 0; i < nuf; i++)
      fs.w_u[(size_t)i * nuf + i] = 1e-4;
    fs.w_cent.assign(36, 0.0);
    fs.w_cent[0] = fs.w_cent[7] = 0.04;
    fs.w_forces = {1e-4, 0, 0, 0, 1e-4, 0, 0, 0, 1e-4};
    fs.w_frame = {1000, 0, 0, 0, 1000, 0, 0, 0, 1000};
    fs.umin.assign(nuf, -40.0);
    fs.umax.assign(nuf, 40.0);
    fs.qmin.assign(robot->q_lo, robot->q_lo + nuf);
    fs.qmax.assign(robot->q_hi, robot->q_hi + nuf);
    fs.Kp_correction.assign(3, 0.0);
    fs.Kd_correction.assign(3, 0.0);
    fs.force_size = 3;
    fs.mu = 0.8;
    fs.Lfoot = fs.Wfoot = 0.01;
    fs.force_cone = false;
    MPCSettings fms = ms;
    fms.T = 12;
    fms.T_fly = 6;
    fms.T_contact = 2;
    BatchedMPC fmpc(robot, fs, fms, /*batch=*/2);
    CHECK(fmpc.horizon() == 12 && fmpc.nx() == nq + nv && fmpc.nu() == nuf);
    std::vector<std::map<std::string, bool>> cyc;
    for (int i = 0; i < 16; i++)
    {
      std::map<std::string, bool> st;
      for (int f = 0; f < nf; f++)
        st[robot->foot_name[f]] = (i < 2 || (i >= 8 && i < 10)) ? true : ((i < 8) == (f == 0 || f == 3));
      cyc.push_back(st);
    }
    fmpc.generateCycleHorizon(cyc);
    for (int it = 0; it < 4; it++)
      fmpc.iterate(X);
    CHECK((int)fmpc.xs_.size() == 2 * 13 * (nq + nv) && (int)fmpc.us_.size() == 2 * 12 * nuf);
    for (double v : fmpc.xs_)
      CHECK(std::isfinite(v));
    const std::vector<double> lam = fmpc.getContactForces(); // [B][H][nfeet][3]
    CHECK((int)lam.size() == 2 * 12 * nf * 3);
    double fz0 = 0.0;
    for (int f = 0; f < nf; f++)
      fz0 += lam[3 * f + 2];
    CHECK(fz0 > 0.5 * robot->total_mass * 9.81 && fz0 < 1.5 * robot->total_mass * 9.81); // the stance feet carry the robot
    {
      // every option of FullDynamicsSettings is built: land_cstr and force_cone together, across the first touch-down of the cycle
      FullDynamicsSettings all = fs;
      all.land_cstr = true;
      all.force_cone = true;
      BatchedMPC lmpc(robot, all, fms, 1);
      lmpc.generateCycleHorizon(cyc);
      const std::vector<double> X1(X.begin(), X.begin() + nq + nv);
      for (int it = 0; it < 10; it++)
        lmpc.iterate(X1);
      for (double v : lmpc.xs_)
        CHECK(std::isfinite(v));
    }
    bool threw4 = false;
    try
    {
      FullDynamicsSettings bad = fs;
      bad.force_size = 6; // point feet carry 3-D forces: rejected, not approximated
      bad.Kp_correction.assign(6, 0.0);
      bad.Kd_correction.assign(6, 0.0);
      BatchedMPC nope(robot, bad, fms, 1);
    }
    catch (const std::runtime_error &)
    {
      threw4 = true;
    }
    CHECK(threw4);
  }
  {
    // ---- the inverse-dynamics controllers, as the reference's tests/inverse-dynamics/kinodynamics-id.cpp:100-143 uses them:
    //      posture task only, no contact -> the joints hold still, the base falls freely ; then all tasks, feet on the ground ----
    const std::vector<double> effort = {23.7, 23.7, 45.43, 23.7, 23.7, 45.43, 23.7, 23.7, 45.43, 23.7, 23.7, 45.43};
    const std::vector<double> vmax = {30.1, 30.1, 15.7, 30.1, 30.1, 15.7, 30.1, 30.1, 15.7, 30.1, 30.1, 15.7};
    KinodynamicsIDSettings s;
    s.kp_posture = 20.0;
    s.w_posture = 1.0;
    BatchedKinodynamicsID kid(robot, 1e-3, s, effort, vmax, /*batch=*/2);
    std::vector<double> q(robot->q_ref, robot->q_ref + nq), z(nv, 0.0), f0(3 * nf, 0.0);
    kid.setTarget(q, z, z, std::vector<bool>(nf, false), f0);
    std::vector<double> Q(2 * nq), V(2 * nv, 0.0), tau, acc;
    for (int b = 0; b < 2; b++)
      std::copy(q.begin(), q.end(), Q.begin() + b * nq);
    kid.solve(0.0, Q, V, tau);
    kid.getAccelerations(acc);
    CHECK((int)tau.size() == 2 * (nv - 6) && (int)acc.size() == 2 * nv);
    CHECK(std::fabs(acc[2] + 9.81) < 1e-6 && std::fabs(acc[nv + 2] + 9.81) < 1e-6); // free fall of both bases
    for (int j = 6; j < nv; j++)
      CHECK(std::fabs(acc[j]) < 1e-5); // posture held
    KinodynamicsIDSettings s2;
    s2.kp_base = 10.0;
    s2.kp_posture = 1.0;
    s2.kp_contact = 10.0;
    s2.w_base = 10.0;
    s2.w_posture = 0.1;
    s2.w_contact_force = 1e-3;
    s2.w_contact_motion = 1.0;
    BatchedKinodynamicsID kid2(robot, 1e-3, s2, effort, vmax, 2);
    kid2.solve(0.0, Q, V, tau); // default target: the reference state, every foot in contact, equal shares of the weight
    double fz = 0.0;
    for (int k = 0; k < nf; k++)
      fz += kid2.getContactForces()[3 * k + 2];
    CHECK(std::fabs(fz - robot->total_mass * 9.81) < 0.05 * robot->total_mass * 9.81); // the feet carry the robot standing still
    for (size_t i = 0; i < tau.size(); i++)
      CHECK(std::fabs(tau[i]) <= effort[i % (nv - 6)] + 1e-6);
    CHECK(kid2.residuals()[0] < 1e-4);
    // CentroidalID: default targets (CoM and feet of the reference state), then a foot in the air with a target 5 cm up
    CentroidalIDSettings cs;
    static_cast<KinodynamicsIDSettings &>(cs) = s2;
    cs.kp_com = 7.0;
    cs.w_com = 10.0;
    cs.kp_feet_tracking = 5.0;
    cs.w_feet_tracking = 100.0;
    BatchedCentroidalID cid(robot, 1e-3, cs, effort, vmax, 2);
    cid.solve(0.0, Q, V, tau);
    CHECK(cid.residuals()[0] < 1e-4);
    std::vector<double> com(3), fp(3 * nf);
    {
      std::vector<double> dbg(2 * 3), dfp(2 * 3 * nf);
      CHECK(smpc_id_debug_get(cid.handle(), 10, dbg.data()) == 0 && smpc_id_debug_get(cid.handle(), 11, dfp.data()) == 0);
      std::copy(dbg.begin(), dbg.begin() + 3, com.begin());
      std::copy(dfp.begin(), dfp.begin() + 3 * nf, fp.begin());
    }
    fp[3 * 2 + 2] += 0.05;
    std::vector<bool> contact(nf, true);
    contact[2] = false;
    std::vector<double> ft(3 * nf, 0.0);
    for (int k = 0; k < nf; k++)
      ft[3 * k + 2] = k == 2 ? 0.0 : robot->total_mass * 9.81 / 3.0;
    cid.setTarget(com, std::vector<double>(3, 0.0), fp, std::vector<double>(3 * nf, 0.0), contact, ft, /*instance=*/1);
    cid.solve(0.0, Q, V, tau);
    cid.getAccelerations(acc);
    CHECK(std::fabs(cid.getContactForces()[3 * nf + 3 * 2 + 2]) < 1e-6); // robot 1: the foot in the air carries nothing
    CHECK(std::fabs(cid.getContactForces()[3 * 2 + 2]) > 1.0);          // robot 0 still stands on it
    bool threw5 = false;
    try
    {
      kid2.setTarget(q, z, z, std::vector<bool>(nf - 1, true), f0);
    }
    catch (const std::runtime_error &)
    {
      threw5 = true;
    }
    CHECK(threw5);
  }
  {
    // ---- a biped with flat feet (the Talos configuration of the reference's own MPC tests, tests/mpc.cpp:97-170 / test_utils.cpp:147-197):
    //      kinodynamics OCP with 6-D feet behind the same BatchedMPC, then the flat-foot inverse dynamics (tsid Contact6d) ----
    const smpc_robot_model * talos = smpc_builtin_robot("talos_like");
    CHECK(talos != nullptr);
    const int tnv = talos->nv, tnq = talos->nq, tnf = talos->nfeet;
    CHECK(tnv == 28 && tnq == 29 && tnf == 2);
    KinodynamicsSettings tk;
    const int tndx = 2 * tnv, tnu = tnv - 6 + 6 * tnf;
    const double wq[28] = {0, 0, 1000, 1000, 1000, 1000, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 1, 1000, 1, 1, 10, 10, 1, 1, 10, 10};
    const double wv[28] = {10, 10, 10, 10, 10, 10, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0.1, 100, 10, 10, 10, 10, 10, 10, 10, 10};
    tk.w_x.assign((size_t)tndx * tndx, 0.0);
    for (int i = 0; i < tnv; i++)
    {
      tk.w_x[(size_t)i * tndx + i] = 10.0 * wq[i];
      tk.w_x[(size_t)(tnv + i) * tndx + tnv + i] = 10.0 * wv[i];
    }
    tk.w_u.assign((size_t)tnu * tnu, 0.0);
    const double wf[6] = {0.001, 0.001, 0.01, 0.1, 0.1, 0.1};
    for (int i = 0; i < tnu; i++)
      tk.w_u[(size_t)i * tnu + i] = i < 6 * tnf ? wf[i % 6] : 1e-4;
    tk.w_frame.assign(36, 0.0);
    for (int i = 0; i < 6; i++)
      tk.w_frame[i * 7] = 100000.0;
    const double twc[6] = {0, 0, 1, 0.1, 0.1, 10}, twcd[6] = {0, 0, 0, 0.1, 0.1, 0.1};
    tk.w_cent.assign(36, 0.0);
    tk.w_centder.assign(36, 0.0);
    for (int i = 0; i < 6; i++)
    {
      tk.w_cent[i * 7] = twc[i];
      tk.w_centder[i * 7] = twcd[i];
    }
    tk.qmin.assign(talos->q_lo, talos->q_lo + (tnv - 6));
    tk.qmax.assign(talos->q_hi, talos->q_hi + (tnv - 6));
    tk.kinematics_limits = true;
    tk.force_cone = true;
    tk.force_size = 6;
    tk.mu = 0.8;
    tk.Lfoot = 0.1;
    tk.Wfoot = 0.075;
    MPCSettings tm;
    tm.max_iters = 1;
    tm.support_force = talos->total_mass * 9.81;
    tm.TOL = 1e-4;
    tm.mu_init = 1e-8;
    tm.swing_apex = 0.15;
    tm.T_fly = 80;
    tm.T_contact = 20;
    tm.T = 30;
    tm.timestep = 0.01;
    BatchedMPC tmpc(talos, tk, tm, /*batch=*/2);
    CHECK(tmpc.horizon() == 30 && tmpc.nx() == tnq + tnv && tmpc.nu() == tnu);
    std::vector<std::map<std::string, bool>> walk;
    auto tpush = [&](int n, bool left, bool right) {
      for (int i = 0; i < n; i++)
        walk.push_back({{talos->foot_name[0], left}, {talos->foot_name[1], right}});
    };
    tpush(20, true, true);
    tpush(80, true, false);
    tpush(20, true, true);
    tpush(80, false, true);
    tmpc.generateCycleHorizon(walk);
    std::vector<double> TX((size_t)2 * tmpc.nx(), 0.0);
    for (int b = 0; b < 2; b++)
      for (int i = 0; i < tnq; i++)
        TX[(size_t)b * tmpc.nx() + i] = talos->q_ref[i];
    for (int it = 0; it < 3; it++)
      tmpc.iterate(TX);
    double tfz = 0.0; // controls = [wrench of the left foot (6) | wrench of the right foot (6) | joint accelerations]
    for (int f = 0; f < tnf; f++)
      tfz += tmpc.us_[6 * f + 2];
    CHECK(std::fabs(tfz - talos->total_mass * 9.81) < 0.10 * talos->total_mass * 9.81);
    for (size_t i = 0; i < tmpc.xs_.size(); i++)
      CHECK(std::isfinite(tmpc.xs_[i]));

    const double eff[22] = {100, 160, 160, 300, 160, 100, 100, 160, 160, 300, 160, 100, 200, 200, 44, 44, 22, 22, 44, 44, 22, 22};
    const double vm[22] = {3.87, 5.86, 5.86, 7.0, 5.86, 4.8, 3.87, 5.86, 5.86, 7.0, 5.86, 4.8, 5.4, 5.4, 2.7, 3.66, 4.58, 4.58, 2.7, 3.66, 4.58, 4.58};
    KinodynamicsIDSettings ts; // gains of the reference's contactQuad tests (tests/inverse-dynamics/kinodynamics-id.cpp:192-204)
    ts.kp_base = 1.0;
    ts.kp_posture = 1.0;
    ts.kp_contact = 10.0;
    ts.w_base = 1.0;
    ts.w_posture = 0.05;
    ts.w_contact_motion = 10.0;
    ts.w_contact_force = 1.0;
    ts.force_size = 6;
    const double quad[12] = {0.1, 0.075, 0, -0.1, 0.075, 0, -0.1, -0.075, 0, 0.1, -0.075, 0};
    for (int f = 0; f < tnf; f++)
      ts.quad_contact_points.insert(ts.quad_contact_points.end(), quad, quad + 12);
    BatchedKinodynamicsID tid(talos, 1e-3, ts, std::vector<double>(eff, eff + 22), std::vector<double>(vm, vm + 22), 2);
    std::vector<double> TQ(2 * tnq), TV(2 * tnv, 0.0), ttau;
    for (int b = 0; b < 2; b++)
      std::copy(talos->q_ref, talos->q_ref + tnq, TQ.begin() + b * tnq);
    tid.solve(0.0, TQ, TV, ttau); // default target: the reference state, both feet in contact
    CHECK((int)ttau.size() == 2 * (tnv - 6) && (int)tid.getContactForces().size() == 2 * 6 * tnf);
    double wz = 0.0; // the wrenches T f of the corner forces, foot frames (soles flat on the ground: z = vertical)
    for (int f = 0; f < tnf; f++)
      wz += tid.getContactForces()[6 * f + 2];
    CHECK(std::fabs(wz - talos->total_mass * 9.81) < 0.05 * talos->total_mass * 9.81);
    for (size_t i = 0; i < ttau.size(); i++)
      CHECK(std::fabs(ttau[i]) <= eff[i % 22] + 1e-6);
    bool threw6 = false;
    try
    {
      KinodynamicsIDSettings bad = ts;
      bad.quad_contact_points.clear();
      BatchedKinodynamicsID nope(talos, 1e-3, bad, std::vector<double>(eff, eff + 22), std::vector<double>(vm, vm + 22), 1);
    }
    catch (const std::runtime_error &)
    {
      threw6 = true;
    }
    CHECK(threw6);
  }
  std::puts("host mirror KAT: OK");
  return 0;
}
