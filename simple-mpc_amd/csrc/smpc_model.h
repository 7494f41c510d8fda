// smpc_model.h -- compile-time problem dimensions, the device-resident model table and the HBM layout
// of every per-instance buffer of the batched kinodynamics MPC engine.
//
// Batch layout: instance-major.  Every per-instance array is [B][...] with the instance's data
// contiguous, so that the lanes of the workgroup that owns an (instance, stage) read and write
// contiguous, coalesced runs.  Stage-indexed arrays are rings of R = H+1 slots addressed through
// slot(t) = (head + t) % R: receding the horizon (reference: replaceStageCircular + cycleProblem,
// src/mpc.cpp:225-226) is a head increment, not a memmove.
#pragma once
#include <vector>
#include "../../include/smpc_robot.h"
#include <smpc_backend.h>

namespace smpc
{
  template <int NJ_, int NF_>
  struct Dims
  {
    static constexpr int NJ = NJ_;         // joints incl. free-flyer
    static constexpr int NF = NF_;         // feet (3-D point contacts)
    static constexpr int NV = NJ_ + 5;     // 6 + (NJ-1)
    static constexpr int NQ = NJ_ + 6;     // 7 + (NJ-1)
    static constexpr int NX = NQ + NV;     // reference: nq + nv
    static constexpr int NDX = 2 * NV;     // src/ocp-handler.cpp:17
    static constexpr int NA = NV - 6;      // actuated joints
    static constexpr int NU = NA + 3 * NF; // src/kinodynamics.cpp:34
    static constexpr int NC = NA + 3 * NF; // joint box rows + 3 velocity rows per foot
    // LQ knot block (doubles), one per (instance, stage)
    static constexpr int O_A = 0;
    static constexpr int O_B = O_A + NDX * NDX;
    // [Q S; S^T R] as the UPPER 16 x 16 TILES of the (x | u) grid in the accumulator layout of v_mfma_f64_16x16x4 -- tile (I, J), I <= J, element v
    // of lane l holds entry (16 I + (l >> 4) + 4 v, 16 J + (l & 15)), at O_T + ((tile * 4 + v) * 64 + l): the derivative pass stores its accumulator
    // registers as they are (512-byte runs) and the Riccati sweep loads its own (no per-element address arithmetic on either side).  Diagonal tiles
    // hold both triangles (the writer mirrors the upper one).  Readers that want single entries: q_off / s_off / r_off.
    static constexpr int NXU = NDX + NU;
    static constexpr int NTT = (NXU + 15) / 16;                       // tile rows / columns of the grid
    static constexpr int O_T = O_B + NDX * NU;
    static constexpr int N_T = NTT * (NTT + 1) / 2 * 256;
    SMPC_HD static constexpr int t_off(int r, int c)                  // entry (r, c) of the grid, r / 16 <= c / 16
    {
      return O_T + (((r / 16) * NTT - (r / 16) * (r / 16 - 1) / 2 + (c / 16 - r / 16)) * 4 + (r % 16) / 4) * 64 + ((r % 16) % 4) * 16 + c % 16;
    }
    SMPC_HD static constexpr int q_off(int i, int j) { return i <= j ? t_off(i, j) : t_off(j, i); }      // Q(i, j)
    SMPC_HD static constexpr int s_off(int i, int j) { return t_off(i, NDX + j); }                        // S(i, j)
    SMPC_HD static constexpr int r_off(int i, int j) { return i <= j ? t_off(NDX + i, NDX + j) : t_off(NDX + j, NDX + i); } // R(i, j)
    static constexpr int O_C = O_T + N_T;
    static constexpr int O_q = O_C + NC * NDX;
    static constexpr int O_r = O_q + NDX;
    static constexpr int O_f = O_r + NU;
    static constexpr int O_d = O_f + NDX;
    static constexpr int O_lx = O_d + NC;   // cost gradient without multipliers (for dphi)
    static constexpr int O_lu = O_lx + NDX;
    static constexpr int O_lpd = O_lu + NU; // 2 lam+ - lam of this stage's dynamics
    static constexpr int O_vpd = O_lpd + NDX;
    static constexpr int LQ_STRIDE = ((O_vpd + NC + 7) / 8) * 8;
    // gains block per (instance, stage): Kk = [K k] (NU x (NDX+1)), Pt (NDX x NDX), pnext (NDX)
    static constexpr int G_K = 0;
    static constexpr int G_Pt = G_K + NU * (NDX + 1);
    static constexpr int G_pn = G_Pt + NDX * NDX;
    static constexpr int G_STRIDE = ((G_pn + NDX + 7) / 8) * 8;
    static constexpr int LS_N = 10; // line-search candidates 2^0 .. 2^-9
  };

  // Model + settings, device resident (one copy per handle).
  // Model constants.  The "small" part (tree tables, small weights, scalars: 1.9 KB) is copied into LDS by every
  // rigid-body kernel block at start, so that no phase of the tree algorithms waits on a global load; the large
  // weight matrices stay in global memory (L2-resident, read in the assembly phases only).
  template <class D>
  struct DevModelSmall
  {
    double total_mass;
    // KinodynamicsSettings (include/simple-mpc/kinodynamics.hpp:24-51)
    double dt;
    double gravity[3];
    double w_cent[36];
    double w_centder[36];
    double qmin[D::NA];
    double qmax[D::NA];
    double wxd[D::NDX], wud[D::NU]; // diagonals of w_x, w_u (used when w_diag: both weight matrices are diagonal)
    // solver
    double mu;
    int parent[D::NJ];
    int jtype[D::NJ];
    int level[D::NJ];
    unsigned anc[D::NJ];      // bit a set <=> joint a is an ancestor-or-self of joint j
    unsigned children[D::NJ]; // bit c set <=> parent[c] == j
    int foot_joint[D::NF];
    int nlevels;
    int kinematics_limits;
    int w_diag; // 1: w_x and w_u are diagonal (the usual case) -> no large-weight traffic in the kernels
    int pad_[1 + (5 * D::NJ + D::NF) % 2]; // keeps sizeof a multiple of 8
  };
  template <class D>
  struct DevModel : DevModelSmall<D>
  {
    // per-joint geometry / inertia: each lane keeps its joint's 24 constants in registers (loaded once per block)
    double jpR[D::NJ][9];
    double jpp[D::NJ][3];
    double mass[D::NJ];
    double com[D::NJ][3];
    double inertia[D::NJ][6];
    double foot_p[D::NF][3];
    double foot_ref_p[D::NF][3];
    double w_frame[9]; // (read from global memory by the two phases that use it: 72 B of LDS decide the 8th resident wave)
    double w_x[D::NDX * D::NDX];
    double w_u[D::NU * D::NU];
    double w_xT[D::NDX * D::NDX]; // transposes: lane = row matvecs read them with coalesced loads
    double w_uT[D::NU * D::NU];
    // terminal state_cost target (model reference state)
    double x_term[D::NX];
    // lane-per-problem evaluation (smpc_kino_lane.h): the kinematic state of a joint's parent is either that of the joint before it
    // (par_slot < 0) or one of LANE_SLOTS saved branch joints (par_slot = slot the parent was saved into: save_slot of that joint)
    int par_slot[D::NJ];
    int save_slot[D::NJ];
    int lane_slots; // saved branch joints the tree needs (the lane kernel is instantiated for 1 and 2; more: 0 = not available)
  };
  constexpr int LANE_SLOTS = 2;

  // stage descriptors shared by all instances (phase-aligned batch), uploaded per control step
  template <class D>
  struct StageShared
  {
    unsigned mask;
    unsigned land; // bit per foot: the foot lands at this stage of the cycle (land_cstr rows; reference src/mpc.cpp:167-178)
    double u_ref[D::NU];
    double x_tgt[D::NX];
  };

  template <class D>
  struct Buffers
  {
    int B = 0, H = 0, R = 0; // batch, horizon, ring length H+1
    // ring state, [B][R][.]
    double *xs = nullptr, *us = nullptr, *vs = nullptr, *lams = nullptr; // lams[slot(t)] = lambda_{t+1}
    double *vs_e = nullptr, *lams_e = nullptr;                           // AL centres
    double *xs_b = nullptr, *us_b = nullptr, *vs_b = nullptr, *lams_b = nullptr; // iterate before a tentative full step
    // steps, [B][H(+1)][.] (linear in t)
    double *dxs = nullptr, *dus = nullptr, *dvs = nullptr, *dlams = nullptr;
    // per-instance references, [B][H][NF*3]
    double * foot_ref = nullptr;
    double * ftraj = nullptr; // [B][NF][6] swing start / end
    // velocity command per instance (MPC::velocity_base_, [B][6]) and the velocity part of the state_cost target of every
    // stage in the horizon (ring [B][R][6]: what setVelocityBase wrote when the stage entered, src/mpc.cpp:312)
    double *vbase = nullptr, *vref = nullptr;
    // LQ + gains
    double *lq = nullptr, *gains = nullptr;
    // lane-per-problem stage evaluation (smpc_kino_lane.h): one block per (instance, stage), [B][H+1][EvLayout::STRIDE]; the
    // working set of the evaluating lane and the hand-over to the derivative kernel.  nullptr: the handle does not use it.
    double * ev = nullptr;
    int ev_inst0 = 0; // first instance of this view of the batch (the blocks are tiled over the whole batch)
    size_t ev_tile = 0; // doubles from one tile of `ev` to the next (ev_tile_doubles: all fields, or the heads only when the stream carries the rest)
    // derivative pass: the hand-over of a problem as one contiguous run [B][H+1][EvStream::STRIDE] in the order the tree kernel produces its
    // fields, and that order (ev_order[position] = field id of EvLayout, -1: padding; recorded once by the kernel itself).  nullptr: tiles.
    double * evd = nullptr;
    int * ev_order = nullptr;
    double *QN = nullptr, *qN = nullptr; // [B][NDX*NDX], [B][NDX]
    // terminal equality constraint com + tau vcom = dcm_ref (DCMPositionResidual; createProblem(..., terminal_constraint = true),
    // reference src/ocp-handler.cpp:133-136, src/kinodynamics.cpp:366-388).  CN == nullptr: the problem has none.
    double * CN = nullptr;      // [B][3 NDX + 3]: Jacobian rows | mu (v+ - v) of the current point
    double *vN = nullptr, *vN_e = nullptr, *vN_b = nullptr, *dvN = nullptr; // [B][3] multipliers, AL centres, backup, step
    double * dcm_ref = nullptr; // [B][3]
    double dcm_tau = 0.0, com0z = 0.0;
    // friction-cone rows of the kinodynamics stage (force_cone; CentroidalFrictionConeResidual per foot in contact, reference
    // src/kinodynamics.cpp:124-129): two rows per foot on the force part of u.  es == nullptr: the problem has none.
    //   es / es_e / es_b [B][R][2 NF] multipliers, centres, backup ; des [B][H][2 NF] step ;
    //   ek [B][H][12 NF]: Jacobian rows of the active rows (2 NF x 3) | d = mu (nu+ - nu) | active ? 2 nu+ - nu : 0 | activity
    double *es = nullptr, *es_e = nullptr, *es_b = nullptr, *des = nullptr, *ek = nullptr;
    double cone_mu2 = 0.0; // friction coefficient squared
    // land_cstr rows of the kinodynamics stage: the height of a foot that lands at the stage is pinned to its contact pose
    // (FrameTranslationResidual sliced to z, EqualityConstraint; reference src/kinodynamics.cpp:134-146).  ls == nullptr: none.
    //   ls / ls_e / ls_b [B][R][NF], dls [B][H][NF], lk [B][H][NF (NV + 2)]: rows d p_z / dq (NF x NV) | d | 2 nu+ - nu
    double *ls = nullptr, *ls_e = nullptr, *ls_b = nullptr, *dls = nullptr, *lk = nullptr;
    double land_z[D::NF] = {0}; // contact-pose heights: the feet at the reference state (src/mpc.cpp:162)
    // merit bookkeeping
    double * parts0 = nullptr;   // [B][H+1][4] phi, cost, prim, dual at the current point
    double * partsT = nullptr;   // [B][LS_N][H+1][2] phi, prim at trial points
    double * scal = nullptr;     // [B][16] per-instance scalars, see SC_*
    double * xdotT = nullptr;    // [B][LS_N][2][2NV] trial xdot for t = 0,1
    double * xdot01 = nullptr;   // [B][2][2NV]
    // full-dynamics handles: contact forces of every stage (MPC::getContactForces, reference src/mpc.cpp:354-380) at the trial
    // points [B][LS_N][H][FS NF] and at the accepted point [B][H][FS NF]; null on the other handles
    double *forcesT = nullptr, *forces = nullptr;
    int nforce = 0; // FS * NF
    int * ls_sel = nullptr;      // [B] selected candidate, -1 = undecided
    int * und_list = nullptr;    // [B + 1] compacted indices of the undecided instances; und_list[B] = their count
    double * dbg = nullptr;      // [64] optional in-kernel phase timers (null = off)
    StageShared<D> * stages = nullptr; // [H] linear in t
    DevModel<D> * model = nullptr;
  };
  enum
  {
    SC_PHI0 = 0,
    SC_DPHI0 = 1,
    SC_ALPHA = 2,
    SC_PHI_NEW = 3,
    SC_PRIM = 4,
    SC_DUAL = 5,
    SC_LS_FAILED = 6,
    SC_PREG = 7,
    SC_PRIM_NEW = 8,
    SC_COST = 9,
    SC_COST_NEW = 10,
    SC_LS_INDEX = 11,
    SC_PREG_OLD = 12, // regularisation before a tentative full step
    SC_N = 16
  };

  // optional in-kernel phase timer: accumulates shader cycles since the previous tick into dbg[slot]
  // Rotations of the foot reference placements, [H][nf] row-major 3 x 3 (host side).  OCPHandler::setReferencePose takes a full SE3 and
  // getReferencePose returns it (reference src/kinodynamics.cpp:154-170, tests/problem.cpp:157-160); MPC::iterate overwrites every stage's
  // pose with (identity rotation, Bezier position) before it solves (src/mpc.cpp:303-309), so no solve of this boundary ever sees another
  // rotation: the stage kernels evaluate M_ref = (I, p), this table carries what was set until the next iterate resets it -- as the reference does.
  struct RefRotations
  {
    std::vector<double> R;
    int H = 0, nf = 0;
    bool any = false; // some entry is not the identity
    void init(int H_, int nf_)
    {
      H = H_;
      nf = nf_;
      R.assign((size_t)H * nf * 9, 0.0);
      for (size_t i = 0; i < (size_t)H * nf; i++)
        R[i * 9] = R[i * 9 + 4] = R[i * 9 + 8] = 1.0;
      any = false;
    }
    void set(int t, int f, const double * R9)
    {
      double * r = &R[((size_t)t * nf + f) * 9];
      if (R9 == nullptr)
      {
        for (int i = 0; i < 9; i++)
          r[i] = (i % 4 == 0) ? 1.0 : 0.0;
        return;
      }
      for (int i = 0; i < 9; i++)
      {
        r[i] = R9[i];
        if (R9[i] != ((i % 4 == 0) ? 1.0 : 0.0))
          any = true;
      }
    }
    void get(int t, int f, double * R9) const
    {
      const double * r = &R[((size_t)t * nf + f) * 9];
      for (int i = 0; i < 9; i++)
        R9[i] = r[i];
    }
    void reset() // (every control step: src/mpc.cpp:303-309)
    {
      if (any)
        init(H, nf);
    }
  };
  // Cross-check builds (-DSMPC_CROSSCHECK: the sequential-lane test build, and libsmpc_hip_xcheck.so of the GPU tests) carry the alternative
  // paths of the engines -- the dense / VALU sweeps, the one-kernel stage and centroidal forms, the sequential line search -- and the environment
  // switches that select them.  The shipped library has neither: xcheck_env() is nullptr there and the launches sit under `if constexpr`.
#ifdef SMPC_CROSSCHECK
  constexpr bool kCrossCheck = true;
  inline const char * xcheck_env(const char * name) { return std::getenv(name); }
#else
  constexpr bool kCrossCheck = false;
  inline const char * xcheck_env(const char *) { return nullptr; }
#endif
  SMPC_DEV void prof_tick(double * dbg, int slot, long long & tprev)
  {
    if (!dbg)
      return;
    const long long now = SMPC_CLOCK();
    SMPC_LANES(64)
    if (lane == 0)
      dbg[slot] += (double)(now - tprev);
    SMPC_LANES_END
    tprev = SMPC_CLOCK();
  }

  SMPC_HD int ring_slot(int head, int t, int R)
  {
    int s = head + t;
    return s >= R ? s - R : s;
  }
} // namespace smpc
