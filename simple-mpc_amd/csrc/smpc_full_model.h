// smpc_full_model.h -- compile-time dimensions, device model table and LQ knot layout of the FULL-DYNAMICS OCP
// (reference FullDynamicsOCP: src/fulldynamics.cpp:30-214, settings include/simple-mpc/fulldynamics.hpp:28-65).
//
//   state   x = (q, v)                 control  u = joint torques (nu = nv - 6, actuation [0; I], src/fulldynamics.cpp:35-37)
//   dynamics  MultibodyConstraintFwdDynamics (one rigid contact per foot in contact) + IntegratorSemiImplEuler
//   constraint rows per stage (NC):  [ torque box (nu) | joint box (nv - 6) | cone rows on the contact forces (NCONE) ]
//     - the two box blocks are unit selectors on u / x: they enter the stage KKT as diagonal terms and are never stored
//       as matrices
//     - the cone rows (force_cone: 17 wrench-cone rows per 6-D foot, src/fulldynamics.cpp:163-173; 5 friction-pyramid rows per 3-D
//       foot, :185-190) are dense in x and u
//       (through the contact-force derivatives): blocks Cd (NCONE x NDX) and Dd (NCONE x NU) of the knot
// FS = force size: 3 (point feet, CONTACT_3D LOCAL) or 6 (quad feet, CONTACT_6D LOCAL_WORLD_ALIGNED), src/fulldynamics.cpp:56-75.
//
// The model-independent kernels of smpc_solver_kernels.h (line search bookkeeping, apply, recede, gather) are instantiated
// on FullDims unchanged: DevModel / StageShared are specialised here with the member names those kernels use.
#pragma once
#include "smpc_model.h"

namespace smpc
{
  // CN_: cone rows per 3-D foot (0: none -- the instantiation of the Go2 example of record, force_cone = false; 5: the friction
  // pyramid of MultibodyFrictionConeResidual, src/fulldynamics.cpp:185-190); 6-D feet always carry their 17 wrench-cone rows
  // LN_: land_cstr rows per foot (0: none; 6: the LOCAL_WORLD_ALIGNED frame velocity of a landing 6-D foot, src/fulldynamics.cpp:175-181; 4: its 3
  // linear rows and the height of the contact pose for a 3-D foot, src/fulldynamics.cpp:191-210)
  // KIN_: 0 = full dynamics (above); 1 = the KINODYNAMICS OCP of a robot with 6-D feet on the same stage / solver kernels (reference
  // KinodynamicsOCP with force_size == 6, src/kinodynamics.cpp:40-152: examples/talos_kinodynamics.py, tests/test_utils.cpp:147-197):
  //   control u = [contact wrench (f, tau) per foot (NCM) ; joint accelerations (NA)], the base acceleration follows from the six unactuated
  //   rows of the equations of motion (= Aligator's KinodynamicsFwdDynamics in another frame: Ag a + dAg v = hdot(u, q));
  //   costs: state, control, centroidal momentum, its derivative hdot(u, q) (w_centder), foot placement;
  //   constraint rows: [ (no torque box) | joint box (NA) | CentroidalWrenchConeResidual: 17 constant rows on u per foot in contact (NCONE,
  //   dense block Dd, pivoted explicitly) | 6-row LOCAL frame velocity of every foot in contact (NVEL: equality rows on the state only --
  //   they are FOLDED by the stage kernel, Q += Cv^T Cv / mu, q += Cv^T d / mu, as the structured kinodynamics sweep does; the rows are kept
  //   in the knot for the multiplier step of the forward sweep) ]
  template <int NJ_, int NF_, int FS_, int CN_ = 0, int LN_ = 0, int KIN_ = 0>
  struct FullDims
  {
    static_assert(LN_ == 0 || LN_ == (FS_ == 6 ? 6 : 4), "land_cstr rows per foot");
    static_assert(KIN_ == 0 || (FS_ == 6 && LN_ == 0), "kinodynamics variant: 6-D feet (point feet run on KinoEngine), no land rows (src/kinodynamics.cpp:134)");
    static constexpr bool KINO = KIN_ != 0;
    // The two widest derivative blocks of a stage (R1, JT: FullDerivWide, smpc_full_stage.h) in a per-block slice of device memory instead of
    // LDS: where that buys a third resident block per CU (the biped: full dynamics 73.9 KB -> 53.0 KB, kinodynamics variant 68.7 KB -> 53.3 KB;
    // the quadruped already runs four -- with its blocks in device memory five fit (31.6 KB), two of them share a SIMD and the launch goes
    // 11.9 -> 13.0 ms: measured, not taken)
    static constexpr bool WIDE_DEV = NJ_ > 16;
    static constexpr int NJ = NJ_;     // joints incl. free-flyer
    static constexpr int NF = NF_;     // feet
    static constexpr int FS = FS_;     // contact force size
    static constexpr int NV = NJ_ + 5;
    static constexpr int NQ = NJ_ + 6;
    static constexpr int NX = NQ + NV;
    static constexpr int NDX = 2 * NV;
    static constexpr int NA = NV - 6;
    static constexpr int NU = KINO ? NA + FS_ * NF_ : NA;
    static constexpr int PF = FS_;                            // size of a foot-pose residual: translation (3) or log6 placement (6)
    static constexpr int NCM = FS_ * NF_;                     // contact rows when every foot is in contact
    static constexpr int NCONE1 = FS_ == 6 ? 17 : CN_;        // cone rows per foot (wrench cone of 6-D feet / friction pyramid of 3-D feet)
    static constexpr int NCONE = NCONE1 * NF_;
    static constexpr int NLAND1 = LN_;                        // land_cstr rows per foot (equality rows on the state)
    static constexpr int NLAND = NLAND1 * NF_;
    static constexpr int NCD = NCONE + NLAND;                 // dense rows of the knot: cone rows, then land rows
    static constexpr int NVEL = KINO ? FS_ * NF_ : 0;         // kinodynamics: frame-velocity rows (state only, folded), behind the dense rows
    static constexpr int NC = NU + NA + NCD + NVEL;
    static constexpr int NXU = NDX + NU;
    // LQ knot block (doubles), one per (instance, stage)
    static constexpr int O_A = 0;
    static constexpr int O_B = O_A + NDX * NDX;
    static constexpr int O_Q = O_B + NDX * NU;
    static constexpr int O_S = O_Q + NDX * NDX;
    static constexpr int O_R = O_S + NDX * NU;
    static constexpr int O_C = O_R + NU * NU;       // Cd: dense rows (cone | land), NCD x NDX (active rows, else zero)
    static constexpr int O_D = O_C + NCD * NDX;     // Dd: NCD x NU
    static constexpr int O_V = O_D + NCD * NU;      // Cv: NVEL x NDX (kinodynamics: frame-velocity rows of the feet in contact, else zero)
    static constexpr int O_q = O_V + NVEL * NDX;
    static constexpr int O_r = O_q + NDX;
    static constexpr int O_f = O_r + NU;
    static constexpr int O_d = O_f + NDX;           // mu (nu+ - nu), all NC rows
    static constexpr int O_lx = O_d + NC;
    static constexpr int O_lu = O_lx + NDX;
    static constexpr int O_lpd = O_lu + NU;
    static constexpr int O_vpd = O_lpd + NDX;       // active ? 2 nu+ - nu : 0, all NC rows
    static constexpr int O_act = O_vpd + NC;        // 1.0 / 0.0 activity of all NC rows (box rows, then the dense cone rows)
    // 1.0 while the dense cone rows [Cd | Dd] of this block hold a nonzero row (fdyn_deriv_body: a stage without an active cone row writes its rows --
    // zeros -- only over a block that is not zero already; fresh allocations are zero-filled)
    static constexpr int O_cdirty = O_act + NC;
    static constexpr int LQ_STRIDE = ((O_cdirty + 1 + 7) / 8) * 8;
    // gains block per (instance, stage)
    static constexpr int G_K = 0;                          // [K k]  NU x (NDX+1)
    static constexpr int G_Z = G_K + NU * (NDX + 1);       // [Z z]  NCD x (NDX+1)  (multiplier feedback of the dense rows)
    // INVARIANT of G_Z: rows of INACTIVE dense constraint rows are not written by the light grid of riccati_dense_body (stale values of an
    // earlier iteration / stage stay there); every reader -- forward_full_body, cent6_forward_body -- forms d / mu itself for a row whose
    // activity flag lq[O_act + ...] is 0 and never reads [Z z] of it.  A new reader of G_Z must test the same flag.
    // P~: upper triangle packed row by row (round 6; the forward sweep is bound by the bytes it reads, half of P~ is 18 % of them): entry
    // (i, j), i <= j, at pt_off(i, j); the sweeps that write it and forward_full_body go through pt_row
    static constexpr bool PT_PACKED = true;
    static constexpr int G_Pt = G_Z + NCD * (NDX + 1);
    SMPC_HD static constexpr int pt_row(int i) { return G_Pt + i * NDX - i * (i - 1) / 2 - i; } // + j = entry (i, j), j >= i
    static constexpr int G_pn = G_Pt + NDX * (NDX + 1) / 2; // p_{t+1}
    static constexpr int G_STRIDE = ((G_pn + NDX + 7) / 8) * 8;
    static constexpr int LS_N = 10;
    static_assert(NJ_ <= 32, "ancestor bit sets");
  };

  // the part of the device model every phase reads: copied into LDS once per block
  template <class D>
  struct FullHead
  {
    double total_mass, dt, gravity[3], mu; // mu: ProxDDP penalty (mu_init)
    // FullDynamicsSettings (include/simple-mpc/fulldynamics.hpp:28-65)
    double w_cent[36], w_forces[D::FS * D::FS], w_frame[D::FS * D::FS];
    double w_centder[36]; // kinodynamics variant: weight of the centroidal_derivative_cost (src/kinodynamics.cpp:63-64)
    double Kp[D::FS], Kd[D::FS];
    double umin[D::NU], umax[D::NU], qmin[D::NA], qmax[D::NA];
    double fric_mu, Lfoot, Wfoot;
    double land_z[D::NF]; // land_cstr: heights of the contact poses the cycle stages are created with (the feet at the reference state, src/mpc.cpp:162)
    double prox_accuracy, prox_mu; // ProximalSettings(1e-9, 1e-10, 10), src/fulldynamics.cpp:39
    int prox_max_iter, torque_limits, kinematics_limits, force_cone, w_diag, nlevels, land_cstr, pad0_;
    int parent[D::NJ], jtype[D::NJ], level[D::NJ];
    unsigned anc[D::NJ], children[D::NJ]; // bit a of anc[j]: joint a is j or one of its ancestors
    int foot_joint[D::NF];
    int pad_[2 + (6 + 5 * D::NJ + D::NF) % 2];
    double wxd[D::NDX], wud[D::NU]; // diagonals of w_x, w_u
  };
  template <class D>
  struct FullDevModel : FullHead<D>
  {
    double jpR[D::NJ][9];
    double jpp[D::NJ][3];
    double mass[D::NJ];
    double com[D::NJ][3];
    double inertia[D::NJ][6];
    double foot_p[D::NF][3];
    double foot_ref_p[D::NF][3];
    double w_x[D::NDX * D::NDX];
    double w_u[D::NU * D::NU];
    double x_term[D::NX];
  };
  template <int NJ_, int NF_, int FS_, int CN_, int LN_, int KIN_>
  struct DevModel<FullDims<NJ_, NF_, FS_, CN_, LN_, KIN_>> : FullDevModel<FullDims<NJ_, NF_, FS_, CN_, LN_, KIN_>>
  {
  };

  // stage descriptor shared by the phase-aligned batch
  template <int NJ_, int NF_, int FS_, int CN_, int LN_, int KIN_>
  struct StageShared<FullDims<NJ_, NF_, FS_, CN_, LN_, KIN_>>
  {
    typedef FullDims<NJ_, NF_, FS_, CN_, LN_, KIN_> D;
    unsigned mask;
    unsigned land; // bit per foot: the foot lands at this stage of the cycle (land_cstr rows; reference src/mpc.cpp:167-178)
    double u_ref[D::NU];   // control reference (zero in the reference's stages, src/fulldynamics.cpp:89)
    double f_ref[D::NCM];  // contact-force reference per foot (src/fulldynamics.cpp:122-137)
    double x_tgt[D::NX];
  };
} // namespace smpc
