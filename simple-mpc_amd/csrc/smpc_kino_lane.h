// smpc_kino_lane.h -- the serial part of a kinodynamics stage evaluation with one LANE per (instance, stage) problem.
//
// The evaluation x+ = f(x, u) of one stage (HOT(1), HOT(6); reference: what TrajOptProblem::evaluate does per stage inside
// SolverProxDDP::run, src/mpc.cpp:212, for the stage built by src/kinodynamics.cpp:40-152) starts with a long chain of small scalar
// steps -- forward kinematics down the tree, momentum totals, a 6 x 6 solve, SE(3) exp / log: ~5 kFLOP with almost no parallelism
// inside one problem.  Run by a wavefront per problem (smpc_kino_stage.h) that chain costs ~45 k cycles of mostly idle lanes and LDS
// round trips; run by one lane per problem, 64 problems per wavefront, every VALU instruction does 64 problems' work.
//
//   lane_tree_body  (this file, lane = problem)   staging of the 64 problems' inputs through LDS (coalesced loads, transposed) ->
//                   one pass down the tree in registers -> root totals, base acceleration, x+ on SE(3), the base rows of the
//                   defect and of the state residual -> the problem's HEAD (54 doubles) and, for the derivative pass, its per-joint
//                   kinematics / inertias / SE(3) Jacobians (EvLayout), written as coalesced 512-byte runs
//   trial_rows_body (this file, wavefront = problem)   everything element-wise of a line-search candidate: defect rows, residuals,
//                   costs, constraint rows, AL multipliers, merit partials (lane = row), from the HEAD + the iterate
//   deriv2_body     (smpc_kino_deriv2.h, wavefront = problem)   the same rows + composites, derivative columns, [A | B], Gauss-Newton
//                   products on the matrix cores, from the HEAD + the per-joint block
//
// A joint's kinematic state needs its parent's, which is either the joint before it or one of NSLOT saved branch joints (the model
// table says which: DevModel::par_slot / save_slot).  The evaluation needs no composite quantity per joint -- only the totals at
// the root (momentum, its rate including the joint accelerations) -- so there is no leaf -> root pass.
// Lanes of a wavefront share the stage index t (hence the contact mask: branches on it are scalar), consecutive lanes are
// consecutive instances.  Outputs: line-search evaluations go to field-major tiles of the wavefront's 64 problems (EvLayout: coalesced
// stores, one strided load per reader); the derivative pass goes to a per-problem contiguous stream in production order (EvStream,
// ev_stream_flush: parked per lane in LDS, flushed by a wave-wide transposition; the reader takes it with coalesced loads and commits it
// by the order table the kernel records itself).
#pragma once
#include "smpc_kino_kernels.h"
#include <type_traits>

namespace smpc
{
  template <class D>
  struct EvLayout
  {
    static constexpr int NJ = D::NJ, NV = D::NV, NX = D::NX, NDX = D::NDX, NU = D::NU, NC = D::NC, NF = D::NF;
    // ---- HEAD: what the row kernels need of a problem, one per line-search candidate (slot 0: the derivative pass) ----
    static constexpr int H_ab = 0;                   // 6   base acceleration
    static constexpr int H_eb = H_ab + 6;            // 6   base rows of the defect x+ (-) x_{t+1} (SE(3) log)
    static constexpr int H_rb = H_eb + 6;            // 6   base rows of the state residual x (-) x_tgt
    static constexpr int H_fv = H_rb + 6;            // [NF][3] LOCAL-frame velocity of the foot points
    static constexpr int H_footp = H_fv + 3 * NF;    // [NF][3] foot positions (world)
    static constexpr int H_hg = H_footp + 3 * NF;    // 6   centroidal momentum
    static constexpr int H_hd = H_hg + 6;            // 6   its target rate from gravity and the contact forces
    static constexpr int HEAD = H_hd + 6;
    static_assert(HEAD <= 64, "the head is read with one load per lane");
    static constexpr int O_head = 0;                 // [LS_N][HEAD]
    // ---- derivative pass only ----
    static constexpr int O_S = O_head + D::LS_N * HEAD; // [NV][6]  motion columns (world frame, at the origin)
    static constexpr int O_vel = O_S + NV * 6;       // [NJ][6]
    static constexpr int O_acc = O_vel + NJ * 6;     // [NJ][6]  accelerations without the base's own contribution (O_dab is added by the reader)
    static constexpr int O_I = O_acc + NJ * 6;       // [NJ][10] world inertias of the bodies
    static constexpr int O_oRf = O_I + NJ * 10;      // [NF][9]  rotation of each foot's joint
    static constexpr int O_com = O_oRf + NF * 9;     // 3
    static constexpr int O_dab = O_com + 3;          // 6  spatial acceleration of the base: sum_{k < 6} a_k S_k
    static constexpr int O_Agbi = O_dab + 6;         // [6][6]
    static constexpr int O_Je3 = O_Agbi + 36;        // 9
    static constexpr int O_JeQ = O_Je3 + 9;          // 9
    static constexpr int O_Jq = O_JeQ + 9;           // 36
    static constexpr int O_Jl = O_Jq + 36;           // 36
    static constexpr int STRIDE = ((O_Jl + 36 + 7) / 8) * 8;
    static constexpr int N_DERIV = O_Jl + 36 - O_S;  // doubles the derivative kernel reads beyond the head
  };

  template <class D>
  struct LaneKernelArgs
  {
    Buffers<D> b;
    int head;
    int j0, nj; // TRIAL: candidate range
    int slots;  // 0: lane per (instance, stage) of the batch; > 0: walk the compacted list of undecided instances
    int deriv;  // 1: DERIV mode
    int * order; // DERIV mode, set-up launch only: record the field id of every position of the problem's stream (Buffers::ev_order)
  };

  // Derivative pass: the tree kernel's fields of a problem (head of candidate 0 + the derivative part) as ONE contiguous run, in production
  // order -- the wavefront that reads a problem then takes it with ten fully coalesced loads instead of 619 doubles at a stride of 512
  // bytes (16 x the L2 -> L1 line traffic: 0.28 ms of the derivative kernel's 3.1).  The tree kernel parks EV_CH fields per lane in LDS
  // rows it no longer needs (the base part of its staged inputs) and flushes them transposed: every store instruction writes four
  // problems' 128-byte runs.
  constexpr int EV_CH = 16;
  template <class D>
  struct EvStream
  {
    static constexpr int N = EvLayout<D>::HEAD + EvLayout<D>::N_DERIV;
    // the kernel pads its groups of fields to whole flushes, so that every store site knows its place in the parking slot at compile time
    // (one LDS store with an immediate offset; the flush calls sit at fixed places): base joint 58 -> 64, every other joint 28 -> 32, a
    // foot's 15 -> 16, the tail (totals, base acceleration, SE(3) Jacobians: 165) -> 176
    static constexpr int G_BASE = 64, G_JOINT = 32, G_FOOT = 16, G_TAIL = 176;
    static constexpr int STRIDE = G_BASE + (D::NJ - 1) * G_JOINT + D::NF * G_FOOT + G_TAIL;
    static_assert(STRIDE >= N && STRIDE % EV_CH == 0, "padded stream");
    static constexpr int NLOAD = (STRIDE + 63) / 64;
  };
  // Field i of lane l of tile w lives at ev[(w * STRIDE + i) * 64 + l].
  constexpr int EV_LS = 64;
  struct LaneBlk
  {
    double * p;
    SMPC_HD double & operator[](int i) const { return p[(size_t)i * EV_LS]; }
  };
  // doubles from one tile to the next: the rows of a tile plus an odd number of 128-byte lines, so that the same field of different
  // tiles (what the wavefronts of a launch write at about the same time) does not map to the same memory channels
  // (heads_only: the tile of a handle whose derivative fields travel in the stream -- the [LS_N][HEAD] heads, which come first, are all it holds)
  template <class D>
  SMPC_HD constexpr size_t ev_tile_doubles(bool heads_only = false)
  {
    return (size_t)(heads_only ? ((EvLayout<D>::O_S + 7) / 8) * 8 : EvLayout<D>::STRIDE) * EV_LS + 9 * 16;
  }
  template <class D>
  SMPC_HD LaneBlk lane_block(const Buffers<D> & b, int inst, int t)
  {
    // tile = (group of 64 consecutive instances, stage): exactly the 64 problems one wavefront of lane_tree_body evaluates
    const size_t gi = (size_t)(b.ev_inst0 + inst);
    return LaneBlk{b.ev + ((gi / EV_LS) * (b.H + 1) + t) * b.ev_tile + gi % EV_LS};
  }
  // Block index -> problem index of the wavefront-per-problem kernels that READ the tiles (one field per lane, 512 bytes apart: a
  // 64-byte sector holds the same field of 8 neighbouring problems).  The hardware deals consecutive workgroups round-robin to the 8
  // XCDs, each with its own L2: with the identity map the 8 readers of a sector sit on 8 different L2s and every sector crosses the
  // fabric 8 times.  Here XCD x takes tiles x, x + 8, ... and walks the 64 problems of a tile with consecutive workgroups, so a sector
  // is fetched once and its other 7 readers hit in L2.  (n is rounded up to whole groups of 8 tiles by the launch; p >= n: idle block)
  constexpr int EV_XCDS = 8;
  // block -> (instance slot, stage) for a launch over n instance slots and H + 1 stages: tile = block's (q / 64) * 8 + XCD, the tile's
  // (instance group, stage) = (tile / (H + 1), tile % (H + 1)), slot = group * 64 + q % 64
  SMPC_HD void xcd_problem(int block, int H, int & slot, int & t)
  {
    const int q = block / EV_XCDS, x = block % EV_XCDS;
    const int tile = (q / EV_LS) * EV_XCDS + x;
    slot = (tile / (H + 1)) * EV_LS + q % EV_LS;
    t = tile % (H + 1);
  }
  // blocks of such a launch: whole groups of 8 tiles covering ceil(n / 64) * (H + 1) tiles (slots >= n: idle blocks)
  SMPC_HD int xcd_grid(int n, int H)
  {
    const int tiles = ((n + EV_LS - 1) / EV_LS) * (H + 1);
    return ((tiles + EV_XCDS - 1) / EV_XCDS) * EV_XCDS * EV_LS;
  }
  SMPC_HD void st3(LaneBlk b, int o, V3 v)
  {
    b[o] = v.x;
    b[o + 1] = v.y;
    b[o + 2] = v.z;
  }
  SMPC_HD void stm3(LaneBlk b, int o, const M3 & m)
  {
    b[o] = m.a00;
    b[o + 1] = m.a01;
    b[o + 2] = m.a02;
    b[o + 3] = m.a10;
    b[o + 4] = m.a11;
    b[o + 5] = m.a12;
    b[o + 6] = m.a20;
    b[o + 7] = m.a21;
    b[o + 8] = m.a22;
  }
  SMPC_HD void stsv(LaneBlk b, int o, const SV & s)
  {
    st3(b, o, s.l);
    st3(b, o + 3, s.a);
  }
  SMPC_HD void stsi(LaneBlk b, int o, const SI & I)
  {
    b[o] = I.m;
    b[o + 1] = I.mc.x;
    b[o + 2] = I.mc.y;
    b[o + 3] = I.mc.z;
    b[o + 4] = I.jxx;
    b[o + 5] = I.jxy;
    b[o + 6] = I.jxz;
    b[o + 7] = I.jyy;
    b[o + 8] = I.jyz;
    b[o + 9] = I.jzz;
  }
  SMPC_HD V3 m3_col(const M3 & R, int c)
  {
    return c == 0 ? mk3(R.a00, R.a10, R.a20) : (c == 1 ? mk3(R.a01, R.a11, R.a21) : mk3(R.a02, R.a12, R.a22));
  }

  // base part (7 doubles) of x (+) alpha dx: the formulas of trial_one / apply_body (the trial point of a candidate must be the point
  // apply_body stores, bit for bit); the vector part is x[i] + alpha dx[i - 1] for every i >= 7 (nq = nv + 1)
  SMPC_HD void lane_base_point(const double * x, const double * dx, bool step, double alpha, double * xb)
  {
    if (!step)
    {
#pragma unroll
      for (int i = 0; i < 7; i++)
        xb[i] = x[i];
      return;
    }
    const V3 dv = alpha * ld3(dx), dw = alpha * ld3(dx + 3);
    const Quat q0{x[3], x[4], x[5], x[6]};
    const SE3 E = exp6(dv, dw);
    const V3 p = ld3(x) + quat_to_R(q0) * E.p;
    Quat qn = quat_mul(q0, quat_exp(dw));
    const double n = 1.0 / sqrt(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
    xb[0] = p.x;
    xb[1] = p.y;
    xb[2] = p.z;
    xb[3] = qn.x * n;
    xb[4] = qn.y * n;
    xb[5] = qn.z * n;
    xb[6] = qn.w * n;
  }

  // kinematic state of a joint: placement, spatial velocity, spatial acceleration (bias + joint accelerations; the base's own
  // acceleration is not in it)
  struct LaneJoint
  {
    M3 R;
    V3 p;
    SV v, a;
  };

  // staged inputs of the 64 problems of a wavefront: field f of problem p at stage[f * LANE_PAD + p]
  constexpr int LANE_PAD = 65; // (odd stride: the transposing writes of the staging loop are conflict-free)
  template <class D>
  struct LaneStage
  {
    static constexpr int F_xb = 0;            // 7  base of x (raw)
    static constexpr int F_dxb = F_xb + 7;    // 6  base of dx (raw)
    static constexpr int F_vb = F_dxb + 6;    // 6  base velocities of the evaluation point
    static constexpr int F_q = F_vb + 6;      // NJ - 1 joint angles of the evaluation point
    static constexpr int F_vj = F_q + D::NJ - 1; // NV - 6 joint velocities
    static constexpr int F_u = F_vj + D::NV - 6; // NU controls of the evaluation point
    static constexpr int N = F_u + D::NU;
    // rows F_xb .. F_vb + 5 are read into registers before a lane produces its first output: in DERIV mode with the stream hand-over that
    // memory (FREE rows of LANE_PAD doubles) becomes the lanes' field-parking area
    static constexpr int FREE = F_q;
  };

  // Model constants the tree pass reads joint by joint, copied into LDS once per block: the hand-over stores of joint j - 1 and the
  // (scalar) loads of joint j's constants would otherwise be ordered through memory -- the compiler cannot know that the model table
  // and the hand-over blocks never alias and waits for every outstanding store before each of those loads
  template <class D>
  struct LaneModel
  {
    double jpR[D::NJ][9], jpp[D::NJ][3], mass[D::NJ], com[D::NJ][3], inertia[D::NJ][6], foot_p[D::NF][3];
    int jtype[D::NJ], par_slot[D::NJ], save_slot[D::NJ], foot_joint[D::NF];
  };

  // Stream hand-over, device side.  Every lane parks the fields it produces in its own EV_CH (+1 pad) doubles of the parking area
  // (lane-major: park[lane * EV_PP + k]); when EV_CH are there the wavefront flushes them transposed: lane -> (problem q * 8 + lane / 8,
  // fields 2 (lane % 8), + 1), i.e. eight problems' 128-byte runs per store instruction.  Not inlined: reached from every store site.
  // INVARIANT: nothing in lane_tree_body reads `evd` back -- the stores are untracked (store2_nowait); the reader is the next kernel.
  constexpr int EV_PP = EV_CH + 1;
  template <class D>
  SMPC_DEV_NOINLINE void ev_stream_flush(const double * park, const unsigned * poff, double * evd, int np, int c0, int lane)
  {
    constexpr int NT = 64, PB = NT / (EV_CH / 2); // problems per store instruction
    // (straight-line: all LDS reads, one wait, all stores.  A lane whose problem does not exist -- partial wavefront -- repeats the last
    //  problem's store: the same bytes to the same address)
    const int f = 2 * (lane % (EV_CH / 2));
    constexpr int NB = 4; // problems-per-lane of one batch (two batches: the function's register footprint is the caller's to keep free)
#pragma unroll
    for (int q0 = 0; q0 < NT / PB; q0 += NB)
    {
      double v0[NB], v1[NB];
      unsigned po[NB];
#pragma unroll
      for (int q = 0; q < NB; q++)
      {
        const int p0 = (q0 + q) * PB + lane / (EV_CH / 2), p = p0 < np ? p0 : np - 1;
        v0[q] = park[p * EV_PP + f];
        v1[q] = park[p * EV_PP + f + 1];
        po[q] = poff[p];
      }
#pragma unroll
      for (int q = 0; q < NB; q++)
        store2_nowait(evd + (size_t)po[q] + c0 + f, v0[q], v1[q]);
    }
  }

  // =============================================================================================
  // lane_tree_body: lanes of a block share the stage t; lane <-> instance (full-batch launch) or entry of the compacted list of
  // undecided instances (slots > 0).  grid = (H + 1) * ceil(n / 64), n = B or slots.
  // =============================================================================================
  // STREAM (DERIV mode only): the derivative pass's hand-over as a per-problem contiguous stream (EvStream) instead of the tile
  // RECORD: the set-up launch that writes the order of the stream's fields (LaneKernelArgs::order)
  template <class D, int NSLOT, bool STREAM = false, bool RECORD = false>
  SMPC_DEV void lane_tree_body(const LaneKernelArgs<D> & ka, int block)
  {
    typedef EvLayout<D> L;
    typedef LaneStage<D> ST;
    constexpr int NT = 64, NX = D::NX, NU = D::NU, NDX = D::NDX, NF = D::NF, NV = D::NV, NJ = D::NJ, NQ = D::NQ;
    static_assert(NQ == NV + 1 && NX <= NT, "x[i] <-> dx[i - 1] beyond the base; one element per lane");
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const int t = block % (H + 1), g = block / (H + 1);
    const bool term = t == H;
    const bool deriv = STREAM || ka.deriv != 0; // (STREAM: the derivative-mode instantiation -- a constant there)
    const DevModel<D> & mg = *b.model;
    const unsigned mask = term ? 0u : b.stages[t].mask;
    const double * x_tgt = term ? mg.x_term : b.stages[t].x_tgt;
    const int st = ring_slot(ka.head, t, R), sn = ring_slot(ka.head, term ? t : t + 1, R);
    const int count = ka.slots > 0 ? b.und_list[b.B] : b.B;
    const int stride = ka.slots > 0 ? ka.slots : b.B; // entries per sweep of the grid
    const double dt = mg.dt;
    typedef EvStream<D> ES;
    SMPC_LDS(double, stg, ST::N * LANE_PAD);
    SMPC_LDS(unsigned, poff, NT); // offset of every lane's problem in the stream buffer, in doubles (the flush addresses other lanes' blocks)
    SMPC_LDS(LaneModel<D>, lms, 1);
    constexpr bool stream = STREAM; // (DERIV mode with Buffers::evd)
    static_assert(NT * EV_PP <= ST::FREE * LANE_PAD, "the parking area fits the staged rows that are read before the first field is produced");
    LaneModel<D> & lm = lms[0];
    SMPC_LANES(NT)
    {
      for (int i = lane; i < NJ * 9; i += NT)
        lm.jpR[i / 9][i % 9] = mg.jpR[i / 9][i % 9];
      for (int i = lane; i < NJ * 6; i += NT)
        lm.inertia[i / 6][i % 6] = mg.inertia[i / 6][i % 6];
      for (int i = lane; i < NJ * 3; i += NT)
      {
        lm.jpp[i / 3][i % 3] = mg.jpp[i / 3][i % 3];
        lm.com[i / 3][i % 3] = mg.com[i / 3][i % 3];
      }
      if (lane < NJ)
      {
        lm.mass[lane] = mg.mass[lane];
        lm.jtype[lane] = mg.jtype[lane];
        lm.par_slot[lane] = mg.par_slot[lane];
        lm.save_slot[lane] = mg.save_slot[lane];
      }
      if (lane < NF * 3)
        lm.foot_p[lane / 3][lane % 3] = mg.foot_p[lane / 3][lane % 3];
      if (lane < NF)
        lm.foot_joint[lane] = mg.foot_joint[lane];
    }
    SMPC_LANES_END_WAVE
    const double total_mass = mg.total_mass;
    const V3 gravity = ld3(mg.gravity);
    int spos = 0; // position in a problem's stream: the same in every lane (declared here, in uniform control flow, so that it lives in a scalar register)
    for (int base = g * NT; base < count; base += stride)
    {
      const int np = count - base < NT ? count - base : NT; // problems of this sweep
      for (int jj = 0; jj < (deriv ? 1 : ka.nj); jj++)
      {
        const int cand = deriv ? 0 : ka.j0 + jj;
        double alpha = 1.0;
        for (int i = 0; i < cand; i++)
          alpha *= 0.5;
        // ---- staging: lane = element, loop over the problems; coalesced runs in, transposed into LDS.  The loads of SB problems
        //      are issued back to back (indices clamped, one wait), then committed: a rolled loop would pay the memory latency
        //      once per problem ----
        SMPC_LANES(NT)
        {
          constexpr int SB = STREAM ? 32 : 16; // problems per batch of loads (DERIV mode fetches no steps: twice the batch in the same registers)
          const int lx = lane < NX ? lane : 0, ldx = (lane >= 7 && lane < NX) ? lane - 1 : (lane < 6 ? lane : 0), lu = lane < NU ? lane : 0;
          for (int p0 = 0; p0 < np; p0 += SB)
          {
            double vx[SB], vdx[SB], vu[SB], vdu[SB];
#pragma unroll
            for (int q = 0; q < SB; q++)
            {
              const int p = p0 + q < np ? p0 + q : np - 1;
              const int inst = ka.slots > 0 ? b.und_list[base + p] : base + p;
              const size_t ib = (size_t)inst * R;
              vx[q] = b.xs[(ib + st) * NX + lx];
              vu[q] = b.us[(ib + st) * NU + lu];
              vdx[q] = (STREAM || deriv) ? 0.0 : b.dxs[((size_t)inst * (H + 1) + t) * NDX + ldx];
              vdu[q] = (STREAM || deriv) ? 0.0 : b.dus[((size_t)inst * H + (term ? 0 : t)) * NU + lu];
            }
#pragma unroll
            for (int q = 0; q < SB; q++)
            {
              const int p = p0 + q;
              if (p < np)
              {
                if (lane < NX)
                {
                  // x = [base 7 | joint angles NJ-1 | base velocities 6 | joint velocities]
                  const int f = lane < 7 ? ST::F_xb + lane : (lane < 7 + NJ - 1 ? ST::F_q + lane - 7 : (lane < 7 + NJ - 1 + 6 ? ST::F_vb + lane - (7 + NJ - 1) : ST::F_vj + lane - (7 + NJ - 1 + 6)));
                  stg[f * LANE_PAD + p] = (!deriv && lane >= 7) ? vx[q] + alpha * vdx[q] : vx[q];
                }
                if (!deriv && lane < 6)
                  stg[(ST::F_dxb + lane) * LANE_PAD + p] = vdx[q];
                if (lane < NU)
                  stg[(ST::F_u + lane) * LANE_PAD + p] = term ? 0.0 : (deriv ? vu[q] : vu[q] + alpha * vdu[q]);
              }
            }
          }
        }
        SMPC_LANES_END_WAVE
        // ---- lane = problem ----
        SMPC_LANES(NT)
        if (lane < np || (stream && SMPC_LOCKSTEP)) // (stream: the flush is a wave-wide transposition -- idle lanes run along on a neighbour's instance and store nothing)
        {
          const int pl = lane < np ? lane : np - 1;
          const int inst = ka.slots > 0 ? b.und_list[base + pl] : base + pl;
          if (deriv || ka.slots > 0 || b.ls_sel[inst] < 0) // (full-batch trial launch: skip the instances that already accepted a candidate)
          {
            const LaneBlk blk = lane_block<D>(b, inst, t);
            const int hd0 = L::O_head + cand * L::HEAD;
            auto SG = [&](int f) { return stg[f * LANE_PAD + lane]; };
            // ---- outputs: tile (field-major, EVAL mode) or stream (DERIV mode: EV_CH fields parked in this lane's column of the
            //      rows that staged the base inputs, then flushed transposed; `spos` is the same in every lane) ----
            spos = 0; // (position of the current flush group in the problem's stream)
            int kpos = 0; // place of the next field in this lane's parking slot: a compile-time constant at every store site (padded groups)
            const size_t pbase = ((size_t)(b.ev_inst0 + inst) * (H + 1) + t) * ES::STRIDE; // (< 2^32 doubles: 34 GB of stream)
            double * const park = stg + lane * EV_PP;
            if (stream)
              poff[lane] = (unsigned)pbase;
            auto put = [&](int id, double v) {
              if constexpr (SMPC_LOCKSTEP)
                park[kpos] = v;
              else
                b.evd[pbase + spos + kpos] = v; // (lanes one after the other: straight into the lane's own block)
              if constexpr (RECORD)
                ka.order[spos + kpos] = id;
              kpos++;
              if (kpos == EV_CH)
              {
                if constexpr (SMPC_LOCKSTEP)
                  ev_stream_flush<D>(stg, poff, b.evd, np, spos, lane);
                spos += EV_CH;
                kpos = 0;
              }
            };
            auto pad = [&](auto n_) { // n_: std::integral_constant -- n fields of padding
              if (stream)
              {
#pragma unroll
                for (int i = 0; i < decltype(n_)::value; i++)
                  put(-1, 0.0);
              }
            };
            auto O1 = [&](int off, double v) {
              if (stream)
                put(off, v);
              else
                blk[off] = v;
            };
            auto O3 = [&](int off, V3 v) {
              O1(off, v.x);
              O1(off + 1, v.y);
              O1(off + 2, v.z);
            };
            auto OSV = [&](int off, const SV & sv_) {
              O3(off, sv_.l);
              O3(off + 3, sv_.a);
            };
            auto OM3 = [&](int off, const M3 & m) {
              O1(off, m.a00);
              O1(off + 1, m.a01);
              O1(off + 2, m.a02);
              O1(off + 3, m.a10);
              O1(off + 4, m.a11);
              O1(off + 5, m.a12);
              O1(off + 6, m.a20);
              O1(off + 7, m.a21);
              O1(off + 8, m.a22);
            };
            auto OSI = [&](int off, const SI & I_) {
              O1(off, I_.m);
              O3(off + 1, I_.mc);
              O1(off + 4, I_.jxx);
              O1(off + 5, I_.jxy);
              O1(off + 6, I_.jxz);
              O1(off + 7, I_.jyy);
              O1(off + 8, I_.jyz);
              O1(off + 9, I_.jzz);
            };
            double xb[7];
            {
              double xr[7], dxr[6];
#pragma unroll
              for (int i = 0; i < 7; i++)
                xr[i] = SG(ST::F_xb + i);
#pragma unroll
              for (int i = 0; i < 6; i++)
                dxr[i] = deriv ? 0.0 : SG(ST::F_dxb + i);
              lane_base_point(xr, dxr, !deriv, alpha, xb);
            }
            double vb[6];
#pragma unroll
            for (int k = 0; k < 6; k++)
              vb[k] = SG(ST::F_vb + k);
            // ---- root -> leaf, one pass: placements, motion columns, velocities, accelerations, body inertias; totals at the
            //      root.  The loop stays rolled: j is uniform, the model constants of a joint are scalar loads of that iteration ----
            LaneJoint cur, slot[NSLOT];
            SI Itot;
            SV htot, Ftot;
            const M3 R0 = quat_to_R(Quat{xb[3], xb[4], xb[5], xb[6]});
            const V3 p0 = mk3(xb[0], xb[1], xb[2]);
            V3 fsum = mk3(0, 0, 0), msum = mk3(0, 0, 0); // sum of the contact forces, sum of p_f x F_f
#pragma unroll 1
            for (int j = 0; j < NJ; j++)
            {
              if (j == 0)
              {
                cur.R = R0;
                cur.p = p0;
                cur.v = sv0();
#pragma unroll
                for (int k = 0; k < 6; k++)
                {
                  const V3 ax = m3_col(R0, k % 3);
                  const SV sk = k < 3 ? SV{ax, mk3(0, 0, 0)} : SV{cross(p0, ax), ax};
                  cur.v = cur.v + vb[k] * sk;
                  if (deriv)
                    OSV(L::O_S + k * 6, sk);
                }
                pad(std::integral_constant<int, 2>()); // (36 + 2: the same place in the slot as after a joint's one column)
                cur.a = sv0();
              }
              else
              {
                const int ps = lm.par_slot[j]; // (uniform: scalar branches)
#pragma unroll
                for (int s = 0; s < NSLOT; s++)
                  if (ps == s)
                    cur = slot[s];
                double sn_, cs_;
                sincos(SG(ST::F_q + j - 1), &sn_, &cs_);
                const int jt = lm.jtype[j];
                const M3 Rq = jt == 1 ? M3{1, 0, 0, 0, cs_, -sn_, 0, sn_, cs_}
                                      : (jt == 2 ? M3{cs_, 0, sn_, 0, 1, 0, -sn_, 0, cs_} : M3{cs_, -sn_, 0, sn_, cs_, 0, 0, 0, 1});
                const M3 Rj = cur.R * (ldm3(lm.jpR[j]) * Rq);
                const V3 pj = cur.p + cur.R * ld3(lm.jpp[j]);
                const V3 ax = m3_col(Rj, jt - 1);
                const SV sk = SV{cross(pj, ax), ax};
                const double qd = SG(ST::F_vj + j - 1), aj = SG(ST::F_u + 3 * NF + j - 1);
                const SV vp = cur.v;
                cur.R = Rj;
                cur.p = pj;
                cur.v = vp + qd * sk;
                cur.a = cur.a + qd * crm(vp, sk) + aj * sk;
                if (deriv)
                  OSV(L::O_S + (j + 5) * 6, sk);
              }
              {
                const int ss = lm.save_slot[j];
#pragma unroll
                for (int s = 0; s < NSLOT; s++)
                  if (ss == s)
                    slot[s] = cur;
              }
              // world inertia about the origin, momentum, net force
              {
                const double m = lm.mass[j];
                const V3 c = cur.R * ld3(lm.com[j]) + cur.p;
                const double * il = lm.inertia[j];
                const M3 Il = M3{il[0], il[1], il[3], il[1], il[2], il[4], il[3], il[4], il[5]};
                const M3 Iw = cur.R * Il * transpose(cur.R);
                const double cc = dot(c, c);
                SI I;
                I.m = m;
                I.mc = m * c;
                I.jxx = Iw.a00 + m * (cc - c.x * c.x);
                I.jxy = Iw.a01 - m * c.x * c.y;
                I.jxz = Iw.a02 - m * c.x * c.z;
                I.jyy = Iw.a11 + m * (cc - c.y * c.y);
                I.jyz = Iw.a12 - m * c.y * c.z;
                I.jzz = Iw.a22 + m * (cc - c.z * c.z);
                const SV h = I * cur.v;
                const SV F = I * cur.a + crf(cur.v, h);
                if (j == 0)
                {
                  Itot = I;
                  htot = h;
                  Ftot = F;
                }
                else
                {
                  Itot = Itot + I;
                  htot = htot + h;
                  Ftot = Ftot + F;
                }
                if (deriv)
                {
                  OSV(L::O_vel + j * 6, cur.v);
                  OSV(L::O_acc + j * 6, cur.a);
                  OSI(L::O_I + j * 10, I);
                }
                pad(std::integral_constant<int, 4>()); // (6 + 22 + 4 = 32, base 38 + 22 + 4 = 64)
              }
#pragma unroll 1
              for (int f = 0; f < NF; f++)
                if (j == lm.foot_joint[f])
                {
                  const V3 fp = cur.R * ld3(lm.foot_p[f]) + cur.p;
                  O3(hd0 + L::H_footp + f * 3, fp);
                  // LOCAL-frame velocity of the foot point (the contact rows)
                  O3(hd0 + L::H_fv + f * 3, tmul(cur.R, cur.v.l + cross(cur.v.a, fp)));
                  if ((mask >> f) & 1u)
                  {
                    const V3 Ff = mk3(SG(ST::F_u + 3 * f), SG(ST::F_u + 3 * f + 1), SG(ST::F_u + 3 * f + 2));
                    fsum = fsum + Ff;
                    msum = msum + cross(fp, Ff);
                  }
                  if (deriv)
                    OM3(L::O_oRf + f * 9, cur.R);
                  pad(std::integral_constant<int, 1>()); // (15 + 1)
                }
            }
            // ---- centre of mass, centroidal momentum, its rate without the base acceleration, the target rate ----
            const double im = 1.0 / Itot.m;
            const V3 com = im * Itot.mc;
            double rhs[6];
            {
              const V3 hga = htot.a - cross(com, htot.l);
              O3(hd0 + L::H_hg, htot.l);
              O3(hd0 + L::H_hg + 3, hga);
              const V3 fl = total_mass * gravity + fsum;
              const V3 fa = msum - cross(com, fsum); // sum (p_f - c) x F_f
              O3(hd0 + L::H_hd, fl);
              O3(hd0 + L::H_hd + 3, fa);
              const V3 ba = Ftot.a - cross(com, Ftot.l);
              rhs[0] = fl.x - Ftot.l.x;
              rhs[1] = fl.y - Ftot.l.y;
              rhs[2] = fl.z - Ftot.l.z;
              rhs[3] = fa.x - ba.x;
              rhs[4] = fa.y - ba.y;
              rhs[5] = fa.z - ba.z;
            }
            // ---- base acceleration: Agbi = X0^-1 M1, M1 = Ic0^-1 T(c)^-1 in closed form (smpc_kino_stage.h) ----
            double ab[6];
            {
              const double cc = dot(com, com);
              const double jxx = Itot.jxx - Itot.m * (cc - com.x * com.x), jyy = Itot.jyy - Itot.m * (cc - com.y * com.y),
                           jzz = Itot.jzz - Itot.m * (cc - com.z * com.z);
              const double jxy = Itot.jxy + Itot.m * com.x * com.y, jxz = Itot.jxz + Itot.m * com.x * com.z,
                           jyz = Itot.jyz + Itot.m * com.y * com.z;
              const double a00 = jyy * jzz - jyz * jyz, a01 = jxz * jyz - jxy * jzz, a02 = jxy * jyz - jxz * jyy;
              const double a11 = jxx * jzz - jxz * jxz, a12 = jxy * jxz - jxx * jyz, a22 = jxx * jyy - jxy * jxy;
              const double idet = 1.0 / (jxx * a00 + jxy * a01 + jxz * a02);
              const M3 Ji = M3{a00 * idet, a01 * idet, a02 * idet, a01 * idet, a11 * idet, a12 * idet, a02 * idet, a12 * idet, a22 * idet};
              double Agbi[36];
#pragma unroll
              for (int c = 0; c < 6; c++)
              {
                V3 ml, ma;
                if (c < 3)
                {
                  ml = mk3(c == 0 ? im : 0.0, c == 1 ? im : 0.0, c == 2 ? im : 0.0);
                  ma = mk3(0, 0, 0);
                }
                else
                {
                  ma = m3_col(Ji, c - 3);
                  ml = cross(com, ma);
                }
                const V3 top = tmul(R0, ml - cross(p0, ma)), bot = tmul(R0, ma);
                Agbi[0 * 6 + c] = top.x;
                Agbi[1 * 6 + c] = top.y;
                Agbi[2 * 6 + c] = top.z;
                Agbi[3 * 6 + c] = bot.x;
                Agbi[4 * 6 + c] = bot.y;
                Agbi[5 * 6 + c] = bot.z;
              }
#pragma unroll
              for (int r = 0; r < 6; r++)
              {
                double s = 0.0;
#pragma unroll
                for (int m = 0; m < 6; m++)
                  s += Agbi[r * 6 + m] * rhs[m];
                ab[r] = term ? 0.0 : s;
                O1(hd0 + L::H_ab + r, ab[r]);
              }
              if (deriv)
              {
#pragma unroll
                for (int i = 0; i < 36; i++)
                  O1(L::O_Agbi + i, Agbi[i]);
                O3(L::O_com, com);
                SV dab = sv0();
#pragma unroll
                for (int k = 0; k < 6; k++)
                {
                  const V3 ax = m3_col(R0, k % 3);
                  dab = dab + ab[k] * (k < 3 ? SV{ax, mk3(0, 0, 0)} : SV{cross(p0, ax), ax});
                }
                OSV(L::O_dab, dab);
              }
            }
            // ---- x+ = x (+) [dt (v + dt a); dt a]: base on SE(3); base rows of the defect e = x+ (-) x_{t+1} ----
            {
              const V3 vec = mk3(dt * (vb[0] + dt * ab[0]), dt * (vb[1] + dt * ab[1]), dt * (vb[2] + dt * ab[2]));
              const V3 w = mk3(dt * (vb[3] + dt * ab[3]), dt * (vb[4] + dt * ab[4]), dt * (vb[5] + dt * ab[5]));
              const double tw = sqrt(dot(w, w));
              const M3 W = skew(w);
              const M3 W2 = W * W;
              const SE3Coef kf = se3_coef(tw);
              const V3 out = vec + kf.B * (W * vec) + kf.C * (W2 * vec);
              const V3 pn = p0 + R0 * out;
              const double qs = 0.5 * kf.sinch;
              Quat qn = quat_mul(Quat{xb[3], xb[4], xb[5], xb[6]}, Quat{qs * w.x, qs * w.y, qs * w.z, kf.ch});
              const double n = 1.0 / sqrt(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
              qn = Quat{qn.x * n, qn.y * n, qn.z * n, qn.w * n};
              if (deriv)
              {
                const M3 Q = se3_Q(-1.0 * vec, -1.0 * w, kf);
                const M3 J = m3_id() + (-kf.B) * W + kf.C * W2; // Jexp3(w)
                OM3(L::O_Je3, J);
                OM3(L::O_JeQ, Q);
                // action matrix of exp6(nu)^-1 = [[R^T, -R^T [p]x],[0, R^T]]
                const M3 Rt = transpose(m3_id() + kf.sinc * W + kf.B * W2);
                const M3 Xm = (-1.0) * (Rt * skew(out));
                const double rt[9] = {Rt.a00, Rt.a01, Rt.a02, Rt.a10, Rt.a11, Rt.a12, Rt.a20, Rt.a21, Rt.a22};
                const double xx[9] = {Xm.a00, Xm.a01, Xm.a02, Xm.a10, Xm.a11, Xm.a12, Xm.a20, Xm.a21, Xm.a22};
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                  for (int jx = 0; jx < 3; jx++)
                  {
                    O1(L::O_Jq + i * 6 + jx, rt[i * 3 + jx]);
                    O1(L::O_Jq + (i + 3) * 6 + jx + 3, rt[i * 3 + jx]);
                    O1(L::O_Jq + i * 6 + jx + 3, xx[i * 3 + jx]);
                    O1(L::O_Jq + (i + 3) * 6 + jx, 0.0);
                  }
              }
              if (!term)
              {
                // x_{t+1} at the evaluation point: its base straight from the iterate (13 loads of this lane's own instance)
                const double * xng = b.xs + ((size_t)inst * R + sn) * NX;
                const double * dxn = b.dxs + ((size_t)inst * (H + 1) + t + 1) * NDX;
                double xr[7], dxr[6], xnp[7];
#pragma unroll
                for (int i = 0; i < 7; i++)
                  xr[i] = xng[i];
#pragma unroll
                for (int i = 0; i < 6; i++)
                  dxr[i] = deriv ? 0.0 : dxn[i];
                lane_base_point(xr, dxr, !deriv, alpha, xnp);
                const SE3 Mb{quat_to_R(Quat{xnp[3], xnp[4], xnp[5], xnp[6]}), mk3(xnp[0], xnp[1], xnp[2])};
                const SE3 Ma{quat_to_R(qn), pn};
                V3 ev, ew;
                log6(se3_mul(se3_inv(Mb), Ma), ev, ew);
                O3(hd0 + L::H_eb, ev);
                O3(hd0 + L::H_eb + 3, ew);
              }
              else if (stream)
              { // (the stream has the same fields at every node: the terminal node has no defect)
                O3(hd0 + L::H_eb, mk3(0, 0, 0));
                O3(hd0 + L::H_eb + 3, mk3(0, 0, 0));
              }
            }
            // ---- base rows of the state residual rx = x (-) x_tgt (SE(3) log), Jlog6 for the derivative pass ----
            {
              double xt[7];
#pragma unroll
              for (int i = 0; i < 7; i++)
                xt[i] = x_tgt[i];
              const SE3 Mt{quat_to_R(Quat{xt[3], xt[4], xt[5], xt[6]}), mk3(xt[0], xt[1], xt[2])};
              const SE3 M = se3_mul(se3_inv(Mt), SE3{R0, p0});
              const V3 w = log3(M.R);
              const double tw = sqrt(dot(w, w));
              const M3 W = skew(w);
              const M3 W2 = W * W;
              const SE3Coef kf = se3_coef(tw);
              const V3 out = M.p + (-0.5) * (W * M.p) + kf.D * (W2 * M.p);
              O3(hd0 + L::H_rb, out);
              O3(hd0 + L::H_rb + 3, w);
              if (deriv)
              {
                const M3 Q = se3_Q(-1.0 * out, -1.0 * w, kf);
                const M3 J = m3_id() + 0.5 * W + kf.D * W2; // Jlog3(w)
                const M3 Xm = (-1.0) * (J * Q * J);
                const double ji[9] = {J.a00, J.a01, J.a02, J.a10, J.a11, J.a12, J.a20, J.a21, J.a22};
                const double xx[9] = {Xm.a00, Xm.a01, Xm.a02, Xm.a10, Xm.a11, Xm.a12, Xm.a20, Xm.a21, Xm.a22};
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                  for (int jx = 0; jx < 3; jx++)
                  {
                    O1(L::O_Jl + i * 6 + jx, ji[i * 3 + jx]);
                    O1(L::O_Jl + (i + 3) * 6 + jx + 3, ji[i * 3 + jx]);
                    O1(L::O_Jl + i * 6 + jx + 3, xx[i * 3 + jx]);
                    O1(L::O_Jl + (i + 3) * 6 + jx, 0.0);
                  }
              }
            }
            pad(std::integral_constant<int, ES::G_TAIL - 165>()); // (the tail's 165 fields)
          }
        }
        SMPC_LANES_END_WAVE
      }
    }
  }

  // =============================================================================================
  // trial_rows_body: grid = n * (H + 1), n = B or slots; one wavefront per (instance, stage), lane = row.  Everything element-wise
  // of the line-search candidates j0 .. j0 + nj - 1 at the trial point x (+) alpha dx, u + alpha du, lam + alpha dlam, nu + alpha dnu
  // (HOT(6); reference RolloutType::LINEAR, src/mpc.cpp:44): defect, residuals, costs, constraint rows, AL multipliers -> merit
  // partials partsT, xdotT.  The serial part of the evaluation comes from lane_tree_body's HEAD of the candidate.
  // =============================================================================================
  // LDS scratch of the row phases: plain double pointers into a block of RowsScratch<D>::N doubles (the derivative kernel places it
  // inside arrays of its own scratch that are idle at that time: no second struct type may name that memory)
  template <class D>
  struct RowsScratch
  {
    double *px, *pxn, *pu; // NX, NX, NU: the evaluation point, x_{t+1}, controls
    double * head;         // 64
    double *rx, *ru;       // NDX, NU (general weight matrices: the residuals for the row products)
    double *red, *red8;    // 3 * 64, 3 * 8
    double *xt, *uref, *fref; // NX, NU, 3 NF: state target (its base-velocity part per instance), control reference, foot references
    static constexpr int N = 2 * D::NX + D::NU + 64 + D::NDX + D::NU + 3 * 64 + 3 * 8 + D::NX + D::NU + 3 * D::NF;
    SMPC_HD explicit RowsScratch(double * base)
    {
      px = base;
      pxn = px + D::NX;
      pu = pxn + D::NX;
      head = pu + D::NU;
      rx = head + 64;
      ru = rx + D::NDX;
      red = ru + D::NU;
      red8 = red + 3 * 64;
      xt = red8 + 3 * 8;
      uref = xt + D::NX;
      fref = uref + D::NU;
    }
  };
  template <class D>
  SMPC_DEV void trial_rows_one(const StageKernelArgs<D> & ka, int inst, int t, int j);

  template <class D>
  SMPC_DEV void trial_rows_body(const StageKernelArgs<D> & ka, int block)
  {
    const int H = ka.b.H;
    int slot, t;
    xcd_problem(block, H, slot, t); // (XCD-aware: see xcd_problem)
    if (t > H || slot >= (ka.slots > 0 ? ka.slots : ka.b.B))
      return;
    const int count = ka.slots > 0 ? (slot < ka.slots ? ka.b.und_list[ka.b.B] : 0) : (slot < ka.b.B ? slot + 1 : 0); // (padding blocks: idle)
    const int stride = ka.slots > 0 ? ka.slots : ka.b.B;
    for (int m = slot; m < count; m += stride)
    {
      const int inst = ka.slots > 0 ? ka.b.und_list[m] : m;
      if (ka.slots == 0 && ka.b.ls_sel[inst] >= 0)
        break; // already accepted an earlier candidate (uniform across the workgroup)
      for (int jj = 0; jj < ka.nj; jj++)
        trial_rows_one<D>(ka, inst, t, ka.j0 + jj);
    }
  }

  // Rows of one evaluation point, shared by the trial kernel and the derivative kernel.  Inputs in LDS (sc.px, sc.pxn, sc.pu, sc.head)
  // and per lane (plam, lam_e of dynamics row `lane`; pnu, nu_e of constraint row `lane`).  Outputs per lane: the dynamics row's
  // multiplier estimate lamp, the constraint row's vplus / act, the weighted residuals (see the lane map below), and the wave
  // totals red[0..2] = cost, penalty, primal infeasibility (every lane).
  //   lane map of the weighted residuals wres:  0 .. NDX-1 state (W r)_i | NDX .. NDX+5 W hg | NDX+6 .. NDX+11 W hd | NDX+12 .. +3NF W rf
  //   wru (lane < NU): (W_u r_u)_i
  template <class D, bool OUT, class Sc>
  SMPC_DEV void kino_rows(const Sc & sc, const DevModelSmall<D> & md, const DevModel<D> & mg, const double * wframe, unsigned mask, bool term, SMPC_PL_REF(double, plam_, 64), SMPC_PL_REF(double, lam_e_, 64), SMPC_PL_REF(double, pnu_, 64),
                          SMPC_PL_REF(double, nu_e_, 64), SMPC_PL_REF(double, lamp_, 64), SMPC_PL_REF(double, vplus_, 64), SMPC_PL_REF(int, act_, 64),
                          SMPC_PL_REF(double, wres_, 64), SMPC_PL_REF(double, wru_, 64), double * red);

  template <class D>
  SMPC_DEV void trial_rows_one(const StageKernelArgs<D> & ka, int inst, int t, int j)
  {
    typedef EvLayout<D> L;
    constexpr int NT = 64;
    constexpr int NV = D::NV, NX = D::NX, NDX = D::NDX, NU = D::NU, NC = D::NC, NF = D::NF;
    const Buffers<D> & b = ka.b;
    const int H = b.H, R = b.R;
    const bool term = t == H;
    const DevModel<D> & mg = *b.model;
    SMPC_LDS(double, rows_lds, RowsScratch<D>::N);
    RowsScratch<D> sc(rows_lds);
    const int st = ring_slot(ka.head, t, R), sn = ring_slot(ka.head, term ? t : t + 1, R);
    const size_t ib = (size_t)inst * R;
    double alpha = 1.0;
    for (int i = 0; i < j; i++)
      alpha *= 0.5;
    const unsigned mask = term ? 0u : b.stages[t].mask;
    const double * x_tgt = term ? mg.x_term : b.stages[t].x_tgt;
    const double * u_ref = term ? nullptr : b.stages[t].u_ref;
    const double * vref = term ? nullptr : b.vref + (ib + st) * 6;
    const double * foot_ref = term ? nullptr : b.foot_ref + ((size_t)inst * H + t) * NF * 3;
    const double * dx = b.dxs + ((size_t)inst * (H + 1) + t) * NDX;
    const size_t lt = (size_t)inst * H + (term ? 0 : t);
    const LaneBlk blk = lane_block<D>(b, inst, t);
    SMPC_PL(double, plam, NT);
    SMPC_PL(double, lame, NT);
    SMPC_PL(double, pnu, NT);
    SMPC_PL(double, nue, NT);
    SMPC_LANES(NT)
    {
      // the trial point (vector parts; the base entries are not used: the HEAD carries what depends on them)
      static_assert(NX <= NT && NU <= NT && NDX <= NT && NC <= NT, "one element per lane");
      const double vx = b.xs[(ib + st) * NX + (lane < NX ? lane : 0)];
      const double vxn = b.xs[(ib + sn) * NX + (lane < NX ? lane : 0)];
      const double vdx = dx[(lane >= 1 && lane <= NDX) ? lane - 1 : 0];
      const double vdxn = dx[(term ? 0 : NDX) + ((lane >= 1 && lane <= NDX) ? lane - 1 : 0)];
      const double vu = b.us[(ib + st) * NU + (lane < NU ? lane : 0)], vdu = b.dus[lt * NU + (lane < NU ? lane : 0)];
      const double vl = b.lams[(ib + st) * NDX + (lane < NDX ? lane : 0)], vdl = b.dlams[lt * NDX + (lane < NDX ? lane : 0)];
      const double vn = b.vs[(ib + st) * NC + (lane < NC ? lane : 0)], vdn = b.dvs[lt * NC + (lane < NC ? lane : 0)];
      const double vle = b.lams_e[(ib + st) * NDX + (lane < NDX ? lane : 0)];
      const double vne = b.vs_e[(ib + st) * NC + (lane < NC ? lane : 0)];
      const double vh = blk[L::O_head + j * L::HEAD + (lane < L::HEAD ? lane : 0)];
      // state target: shared pose part, per-instance base-velocity part (address select, one load); control and foot references
      const double vxt = *((vref != nullptr && lane >= D::NQ && lane < D::NQ + 6) ? vref + (lane - D::NQ) : x_tgt + (lane < NX ? lane : 0));
      const double vur = term ? 0.0 : u_ref[lane < NU ? lane : 0];
      const double vfr = term ? 0.0 : foot_ref[lane < NF * 3 ? lane : 0];
      if (lane < NX)
      {
        sc.px[lane] = vx + alpha * vdx;
        sc.pxn[lane] = vxn + alpha * vdxn;
        sc.xt[lane] = vxt;
      }
      if (lane < NU)
        sc.uref[lane] = vur;
      if (lane < NF * 3)
        sc.fref[lane] = vfr;
      if (lane < NU)
        sc.pu[lane] = term ? 0.0 : vu + alpha * vdu;
      sc.head[lane] = vh;
      SMPC_PLV(plam) = term ? 0.0 : vl + alpha * vdl;
      SMPC_PLV(lame) = vle;
      SMPC_PLV(pnu) = term ? 0.0 : vn + alpha * vdn;
      SMPC_PLV(nue) = vne;
    }
    SMPC_LANES_END_WAVE
    double red[3];
    SMPC_PL(double, dmy, NT);
    SMPC_PL(int, dmyi, NT);
    kino_rows<D, false>(sc, mg, mg, mg.w_frame, mask, term, plam, lame, pnu, nue, dmy, dmy, dmyi, dmy, dmy, red);
    double * parts = b.partsT + (((size_t)inst * D::LS_N + j) * (H + 1) + t) * 2;
    SMPC_LANES(NT)
    {
      if (lane == 0)
      {
        parts[0] = term ? red[0] : red[0] + red[1];
        parts[1] = term ? 0.0 : red[2];
      }
      if (t < 2 && lane < NV)
      {
        double * xd = b.xdotT + (((size_t)inst * D::LS_N + j) * 2 + t) * 2 * NV;
        xd[lane] = sc.px[D::NQ + lane];
        xd[NV + lane] = lane < 6 ? sc.head[L::H_ab + lane] : sc.pu[3 * NF + lane - 6];
      }
    }
    SMPC_LANES_END_WAVE
  }

  template <class D, bool OUT, class Sc>
  SMPC_DEV void kino_rows(const Sc & sc, const DevModelSmall<D> & md, const DevModel<D> & mg, const double * wframe, unsigned mask, bool term, SMPC_PL_REF(double, plam_, 64), SMPC_PL_REF(double, lam_e_, 64), SMPC_PL_REF(double, pnu_, 64),
                          SMPC_PL_REF(double, nu_e_, 64), SMPC_PL_REF(double, lamp_, 64), SMPC_PL_REF(double, vplus_, 64), SMPC_PL_REF(int, act_, 64),
                          SMPC_PL_REF(double, wres_, 64), SMPC_PL_REF(double, wru_, 64), double * red)
  {
    typedef EvLayout<D> L;
    constexpr int NT = 64;
    constexpr int NV = D::NV, NQ = D::NQ, NDX = D::NDX, NU = D::NU, NC = D::NC, NF = D::NF, NA = D::NA;
    static_assert(NDX + 12 + 3 * NF <= NT && NDX + NC <= NT + NC, "one residual term per lane");
    const double dt = md.dt, mu = md.mu;
    const bool wdiag = md.w_diag != 0;
    if (!wdiag)
    {
      // general weight matrices: residuals through LDS for the row products
      SMPC_LANES(NT)
      {
        if (lane < NDX)
          sc.rx[lane] = lane < 6 ? sc.head[L::H_rb + lane]
                                 : sc.px[lane + 1] - sc.xt[lane + 1];
        if (!term && lane < NU)
          sc.ru[lane] = sc.pu[lane] - sc.uref[lane];
      }
      SMPC_LANES_END_WAVE
    }
    SMPC_LANES(NT)
    {
      double cost = 0.0, pen = 0.0, prim = 0.0;
      // ---- dynamics row `lane`: defect, multiplier estimate, penalty ----
      if (!term && lane < NDX)
      {
        const int i = lane;
        double e;
        if (i < 6)
          e = sc.head[L::H_eb + i];
        else if (i < NV)
          e = (sc.px[i + 1] + dt * (sc.px[NQ + i] + dt * sc.pu[3 * NF + i - 6])) - sc.pxn[i + 1];
        else
        {
          const int k = i - NV;
          const double ak = k < 6 ? sc.head[L::H_ab + k] : sc.pu[3 * NF + k - 6];
          e = (sc.px[NQ + k] + dt * ak) - sc.pxn[NQ + k];
        }
        const double lp = SMPC_PLV(lam_e_) + e / mu, dl = lp - SMPC_PLV(plam_);
        pen = 0.5 * mu * (lp * lp + dl * dl);
        prim = fabs(e);
        if constexpr (OUT)
          SMPC_PLV(lamp_) = lp;
      }
      // ---- residual terms: state | momentum | momentum rate | foot positions ----
      double wres = 0.0;
      if (lane < NDX)
      {
        const int i = lane;
        const double r = i < 6 ? sc.head[L::H_rb + i]
                               : sc.px[i + 1] - sc.xt[i + 1];
        if (wdiag)
          wres = md.wxd[i] * r;
        else
        {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < NDX; k++)
            s += mg.w_xT[k * NDX + i] * sc.rx[k];
          wres = s;
        }
        cost = r * wres;
      }
      else if (lane < NDX + 6)
      {
        const int i = lane - NDX;
        const double sc10 = term ? 10.0 : 1.0; // terminal: 10 * w_cent (src/kinodynamics.cpp:361)
        double s = 0.0;
        for (int k = 0; k < 6; k++)
          s += sc10 * md.w_cent[i * 6 + k] * sc.head[L::H_hg + k];
        wres = s;
        cost = sc.head[L::H_hg + i] * s;
      }
      else if (!term && lane < NDX + 12)
      {
        const int i = lane - NDX - 6;
        double s = 0.0;
        for (int k = 0; k < 6; k++)
          s += md.w_centder[i * 6 + k] * sc.head[L::H_hd + k];
        wres = s;
        cost = sc.head[L::H_hd + i] * s;
      }
      else if (!term && lane < NDX + 12 + 3 * NF)
      {
        const int i = lane - NDX - 12, f = i / 3, r = i % 3;
        double s = 0.0;
        for (int k = 0; k < 3; k++)
          s += wframe[r * 3 + k] * (sc.head[L::H_footp + f * 3 + k] - sc.fref[f * 3 + k]);
        wres = s;
        cost = (sc.head[L::H_footp + i] - sc.fref[i]) * s;
      }
      if constexpr (OUT)
        SMPC_PLV(wres_) = wres;
      // ---- control residual (lane < NU) ----
      if (!term && lane < NU)
      {
        const double r = sc.pu[lane] - sc.uref[lane];
        double wr;
        if (wdiag)
          wr = md.wud[lane] * r;
        else
        {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < NU; k++)
            s += mg.w_uT[k * NU + lane] * sc.ru[k];
          wr = s;
        }
        cost += r * wr;
        if constexpr (OUT)
          SMPC_PLV(wru_) = wr;
      }
      // ---- constraint row `lane`: joint box (kinematics_limits) | LOCAL velocity of the feet in contact ----
      if (!term && lane < NC)
      {
        const int i = lane;
        int kind; // 0 absent 1 equality 2 box
        double c = 0.0;
        if (i < NA)
        {
          kind = md.kinematics_limits ? 2 : 0;
          c = md.kinematics_limits ? sc.px[7 + i] : 0.0;
        }
        else
        {
          kind = ((mask >> ((i - NA) / 3)) & 1u) ? 1 : 0;
          c = kind ? sc.head[L::H_fv + i - NA] : 0.0;
        }
        double vp = 0.0;
        int act = 0;
        if (kind != 0)
        {
          const double z = c + mu * SMPC_PLV(nu_e_);
          double proj = 0.0;
          if (kind == 2)
            proj = fmin(fmax(z, md.qmin[i < NA ? i : 0]), md.qmax[i < NA ? i : 0]);
          vp = (z - proj) / mu;
          act = (z != proj) || kind == 1;
        }
        const double dv = vp - SMPC_PLV(pnu_);
        pen += 0.5 * mu * (vp * vp + dv * dv);
        if (kind == 2)
          prim = fmax(prim, fmax(fmax(c - md.qmax[i < NA ? i : 0], md.qmin[i < NA ? i : 0] - c), 0.0));
        else if (kind == 1)
          prim = fmax(prim, fabs(c));
        if constexpr (OUT)
        {
          SMPC_PLV(vplus_) = vp;
          SMPC_PLV(act_) = act;
        }
      }
      sc.red[lane] = cost;
      sc.red[64 + lane] = pen;
      sc.red[128 + lane] = prim;
    }
    SMPC_LANES_END_WAVE
    // fixed-order reductions (bitwise reproducible): 64 -> 8 -> 1
    SMPC_LANES(NT)
    if (lane < 24)
    {
      const int q = lane / 8, g8 = lane % 8;
      double s = 0.0;
#pragma unroll
      for (int i = 0; i < 8; i++)
        s = q == 2 ? fmax(s, sc.red[128 + g8 * 8 + i]) : s + sc.red[q * 64 + g8 * 8 + i];
      sc.red8[lane] = s;
    }
    SMPC_LANES_END_WAVE
    {
      double c = 0.0, p = 0.0, m = 0.0;
      for (int i = 0; i < 8; i++)
      {
        c += sc.red8[i];
        p += sc.red8[8 + i];
        m = fmax(m, sc.red8[16 + i]);
      }
      red[0] = 0.5 * c;
      red[1] = p;
      red[2] = m;
    }
  }
} // namespace smpc
