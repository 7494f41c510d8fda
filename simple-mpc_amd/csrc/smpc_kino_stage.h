// smpc_kino_stage.h -- the per-(instance, stage) kinodynamics kernel body.
//
// One 64-lane wavefront owns one (instance, stage) pair.  The stage's state/control block and all
// rigid-body intermediates are staged in LDS; the tree algorithms run level-synchronously with one
// lane per joint / per dof column, the dense assembly (A, B, Gauss-Newton Hessians, constraint rows)
// runs with all 64 lanes, and the LQ knot is written to HBM as contiguous coalesced runs.
//
// What it computes is what the reference obtains, per stage, from Aligator/Pinocchio inside
// SolverProxDDP::run (reference src/mpc.cpp:212) for the stage built by
// KinodynamicsOCP::createStage (reference src/kinodynamics.cpp:40-152):
//   HOT(1) evaluate   : x+ = f(x,u) (KinodynamicsFwdDynamics + IntegratorSemiImplEuler), costs, constraints
//   HOT(2) derivatives: A, B, cost gradient + Gauss-Newton Hessian, constraint Jacobians
//   HOT(3) LQ assembly: AL multipliers / active set, knot (Q,S,R,q,r,A,B,f,C,d)
// World-frame spatial formulation; derivation in DESIGN.md "Rigid-body derivatives".
#pragma once
#include <cstddef>
#include "smpc_math.h"
#include "smpc_model.h"
#include <type_traits>

namespace smpc
{
  // LDS scratch of one (instance, stage) wavefront.  The evaluation-only part (trial points of the line
  // search) is 13.3 KB; the derivative part adds 7 KB (arrays with disjoint lifetimes share storage, see the overlays
  // below) and is only instantiated by the derivative kernel: 20.4 KB = 8 resident waves per CU.
  template <class D>
  struct KinoScratchEval
  {
    DevModelSmall<D> ml; // model constants (copied from global memory once per block)
    double x[D::NX], u[D::NU];
    // stage inputs fetched with the block's other global loads, so that no later phase waits on global memory
    double in_x_tgt[D::NX], in_u_ref[D::NU], in_foot_ref[D::NF * 3]; // x target, u reference, foot refs
    // tree block A (contiguous, 667 doubles): dead once the derivative columns and the constraint values are formed;
    // reused -- in this order of time -- by the wave reductions of the cost / multiplier phases and by the
    // weighted-Jacobian tables of the assembly phases (accessors below)
    double oR[D::NJ * 9], S[D::NV * 6], vel[D::NJ * 6], acc[D::NJ * 6], Ic[D::NJ * 10], hc[D::NJ * 6], Fc[D::NJ * 6];
    SMPC_HD double * part() { return oR; }        // [64]
    SMPC_HD double * part2() { return oR + 64; }  // [64]
    SMPC_HD double * part8() { return oR + 128; } // [16]
    double footp[D::NF * 3];
    double com[3];
    double Ag[6 * D::NV];
    double b0[6], hd[6], hg[6];
    double Agbi[36];
    double a[D::NV];
    double xnext[D::NX], e[D::NDX];
    // ping-pong buffers of the 6x6 Gauss-Jordan: they live in xnext | e, which are written only after Agbi is formed
    SMPC_HD double * gjA_() { return xnext; }
    SMPC_HD double * gjB_() { return xnext + 36; }
    static_assert(D::NX + D::NDX >= 72, "Gauss-Jordan scratch inside xnext | e");
    double red[4];
    double rx[D::NDX];                                          // state residual (its base block comes out of the SE(3) pair, early)
    double lam_next[D::NDX], nu[D::NC];                         // multipliers (block inputs)
    // ---- "late block": written only after the derivative columns are formed (constraint values, residuals, weighted
    //      residuals, multiplier estimates).  Until then the derivative kernel keeps the first 216 doubles of the per-body
    //      velocity-product matrices here (they continue into WJl | JWJ of the derivative part, which follows directly) ----
    double cval[D::NC];
    double Wrx[D::NDX], ru[D::NU], Wru[D::NU], Whg[6], Whd[6], rf[D::NF * 3], Wrf[D::NF * 3];
    double vplus[D::NC], lamp[D::NDX];
    int act[D::NC];
    static constexpr int LATE_DOUBLES = 2 * D::NC + 2 * D::NDX + 2 * D::NU + 12 + 6 * D::NF + D::NC / 2;
  };
  template <class D>
  struct KinoScratchDerivPart
  {
    static_assert(3 * D::NV >= D::NJ * 3 && D::NF * 3 * D::NV >= D::NJ * 10, "op_() / I_() overlays");
    // state-cost tables of the base block, first members: they continue the late block of the evaluation part
    double WJl[D::NDX * 6];           // w_x[:,0:6] * Jl ;  (Jl^T w_x[0:6,:])(i, k) = WJl[k][i] since w_x is symmetric
    double JWJ[36];                   // Jl^T w_x[0:6,0:6] Jl
    double dh_dq[6 * D::NV], dhd_dq[6 * D::NV], dhd_dv[6 * D::NV]; // dhd_*: overwritten in place by ab_dq / ab_dv
    double Jfoot[D::NF * 3 * D::NV];
    double dtgt[3 * D::NV];
    double cn[D::NDX]; // C_x^T nu of the contact rows (formed with the constraint Jacobian columns)
    double Je3[9], JeQ[9], Jq[36], Jl[36];
  };
  template <class D>
  struct KinoScratchNoDeriv
  {
    double xn1[D::NX]; // trial point of the next state (line search only)
    double wfr[9];        // w_frame (the derivative kernel keeps it in the dead acceleration vector, see wframe_())
    double op[D::NJ * 3]; // joint positions (the derivative kernel overlays them, see op_(); the world-frame body inertias are
                          // only read by the derivative kernel)
  };
  template <class D, bool DERIV>
  struct KinoScratch : KinoScratchEval<D>, std::conditional<DERIV, KinoScratchDerivPart<D>, KinoScratchNoDeriv<D>>::type
  {
    // joint positions (read until the base integration) and world-frame body inertias (read until the velocity-product
    // matrices are formed): in the derivative kernel they live in dtgt / Jfoot, which the derivative columns write later
    SMPC_HD double * op_()
    {
      if constexpr (DERIV)
        return this->dtgt;
      else
        return this->op;
    }
    // foot-position weight (3 x 3): 72 B of LDS decide the derivative kernel's 8th resident wave, so there it is parked in
    // the joint-acceleration vector `a` once the net-force update has consumed it (the block's load phase fetches it
    // into spare lanes of the x_{t+1} register)
    SMPC_HD double * wframe_()
    {
      if constexpr (DERIV)
        return this->a;
      else
        return this->wfr;
    }
    SMPC_HD double * I_()
    {
      static_assert(DERIV, "body inertias are kept by the derivative kernel only");
      return this->Jfoot;
    }
    // tables living in tree block A (written by the table phase, after the last reader of the tree data)
    SMPC_HD double * WJc() { return this->oR; }       // [6][NDX]     w_cent * [dh_dq | Ag]
    SMPC_HD double * WD() { return this->oR + 216; }  // [6][NV]      w_centder[:,3:6] * dtgt
    SMPC_HD double * WJu() { return this->oR + 324; } // [6][3 NF]    w_centder * Ju (force columns)
    SMPC_HD double * WJf() { return this->oR + 396; } // [3 NF][NV]   w_frame * Jfoot
    SMPC_HD double * aS() { return this->dhd_dq; }  // [NV][6] temporaries of the net-force update: dhd_dq | dhd_dv are
    SMPC_HD double * IaS() { return this->dhd_dv; } // [NV][6] written afterwards, by the derivative columns
    SMPC_HD double * ab_du() { return this->WJl; }  // [6][NU] lives in the block of the state-cost tables, which are
                                                     //         formed after the [A|B] assembly has consumed it
    SMPC_HD double * ab_dq() { return this->dhd_dq; }      // [6][NV] in place
    SMPC_HD double * ab_dv() { return this->dhd_dv; }      // [6][NV] in place
    static_assert(6 * D::NDX == 216 && 6 * D::NV == 108 && 18 * D::NF == 72 && 396 + 3 * D::NF * D::NV <= D::NJ * 9 + D::NV * 6 + D::NJ * 40,
                  "table layout inside tree block A");
  };

  // inputs describing one stage evaluation
  template <class D>
  struct StageIn
  {
    const DevModel<D> * md;
    unsigned mask;
    const double * u_ref;    // NU (global)   -- these three are read once, by the caller's load phase, into the
    const double * x_tgt;    // NX (global)      scratch copies in_u_ref / in_x_tgt / in_foot_ref that the stage
    const double * foot_ref; // NF*3 (global)    functions use
    double * C_rows = nullptr; // derivative pass: contact rows of the knot's C block (global, [3 NF][NDX])
    bool terminal;
    double * prof = nullptr; // optional phase timers (null = off)
    long long * tprev = nullptr;
  };

  // copy the small model block into the block's LDS scratch (call inside the block's load phase)
  template <class D, int NT, class Scratch>
  SMPC_DEV void lanes_load_model(Scratch & sc, const DevModel<D> * gm, int lane)
  {
    constexpr int N = (int)(sizeof(DevModelSmall<D>) / sizeof(double)), PER = (N + NT - 1) / NT;
    static_assert(sizeof(DevModelSmall<D>) % sizeof(double) == 0, "LDS copy is done in doubles");
    const alias_double * src = reinterpret_cast<const alias_double *>(static_cast<const DevModelSmall<D> *>(gm));
    alias_double * dst = reinterpret_cast<alias_double *>(&sc.ml);
    double r[PER]; // all loads in flight before the first store (a rolled loop would serialise the latencies)
#pragma unroll
    for (int n = 0; n < PER; n++)
      r[n] = src[lane + n * NT < N ? lane + n * NT : 0];
#pragma unroll
    for (int n = 0; n < PER; n++)
      if (lane + n * NT < N)
        dst[lane + n * NT] = r[n];
  }

  // the same in two halves, for load phases that issue further (slower) loads between them: the commit then waits for the model's loads only
  template <class D, int NT>
  struct ModelLoad
  {
    static constexpr int N = (int)(sizeof(DevModelSmall<D>) / sizeof(double)), PER = (N + NT - 1) / NT;
    double r[PER];
    SMPC_DEV void issue(const DevModel<D> * gm, int lane)
    {
      const alias_double * src = reinterpret_cast<const alias_double *>(static_cast<const DevModelSmall<D> *>(gm));
#pragma unroll
      for (int n = 0; n < PER; n++)
        r[n] = src[lane + n * NT < N ? lane + n * NT : 0];
    }
    template <class Scratch>
    SMPC_DEV void commit(Scratch & sc, int lane) const
    {
      alias_double * dst = reinterpret_cast<alias_double *>(&sc.ml);
#pragma unroll
      for (int n = 0; n < PER; n++)
        if (lane + n * NT < N)
          dst[lane + n * NT] = r[n];
    }
  };

  // SE(3) work of a stage on two lanes in lockstep.  Lane 0 integrates the base (exp side, nu = dt (v + dt a)): x+ and,
  // with derivatives, Jexp6(nu) and the action matrix of exp6(nu)^-1.  Lane 1 forms the base block of the state
  // residual (log side, nu = log6(M_tgt^-1 M)) and, with derivatives, Jlog6(nu).  Both need W = [w]x, W^2, the same
  // coefficient functions of |w| and the Q block of the SE(3) Jacobian: those run once, as the same instructions,
  // for the two lanes (on a SIMD machine two different single-lane branches cost the sum of both).
  template <class D, bool DERIV>
  SMPC_DEV void kino_se3_pair(KinoScratch<D, DERIV> & sc, const StageIn<D> & in, const double * vq, double dt, int lane)
  {
    const bool ex = lane == 0;
    V3 vec, w; // vec: exp side dv ; log side translation of M_tgt^-1 M
    if (ex)
    {
      vec = mk3(dt * (vq[0] + dt * sc.a[0]), dt * (vq[1] + dt * sc.a[1]), dt * (vq[2] + dt * sc.a[2]));
      w = mk3(dt * (vq[3] + dt * sc.a[3]), dt * (vq[4] + dt * sc.a[4]), dt * (vq[5] + dt * sc.a[5]));
    }
    else
    {
      const double * xt = sc.in_x_tgt; // (LDS copy made by the block's load phase)
      const SE3 Mt{quat_to_R(Quat{xt[3], xt[4], xt[5], xt[6]}), ld3(xt)};
      const SE3 M = se3_mul(se3_inv(Mt), SE3{ldm3(&sc.oR[0]), ld3(&sc.op_()[0])});
      w = log3(M.R);
      vec = M.p;
    }
    const double t = sqrt(dot(w, w));
    const M3 W = skew(w);
    const M3 W2 = W * W;
    const SE3Coef kf = se3_coef(t);
    const double cB = kf.B, cC = kf.C, cD = kf.D;
    // exp: E.p = (I + B W + C W^2) dv ;  log: v = (I - W/2 + D W^2) p
    const double c1 = ex ? cB : -0.5, c2 = ex ? cC : cD;
    const V3 out = vec + c1 * (W * vec) + c2 * (W2 * vec);
    const V3 v = ex ? vec : out;
    M3 J = m3_id(), Q = m3_id();
    if constexpr (DERIV)
    {
      Q = se3_Q(-1.0 * v, -1.0 * w, kf);
      J = m3_id() + (ex ? -cB : 0.5) * W + c2 * W2; // Jexp3(w) | Jlog3(w)
    }
    if (ex)
    {
      const M3 R0 = ldm3(&sc.oR[0]);
      st3(&sc.xnext[0], ld3(&sc.op_()[0]) + R0 * out);
      const double qs = 0.5 * kf.sinch;
      Quat qn = quat_mul(Quat{sc.x[3], sc.x[4], sc.x[5], sc.x[6]}, Quat{qs * w.x, qs * w.y, qs * w.z, kf.ch});
      const double n = 1.0 / sqrt(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
      sc.xnext[3] = qn.x * n;
      sc.xnext[4] = qn.y * n;
      sc.xnext[5] = qn.z * n;
      sc.xnext[6] = qn.w * n;
      if constexpr (DERIV)
      {
        stm3(sc.Je3, J);
        stm3(sc.JeQ, Q);
        // action matrix of exp6(nu)^-1 = [[R^T, -R^T [p]x],[0, R^T]]
        const M3 Rt = transpose(m3_id() + kf.sinc * W + cB * W2);
        const M3 X = (-1.0) * (Rt * skew(out));
        double * Jq = sc.Jq;
        const double rt[9] = {Rt.a00, Rt.a01, Rt.a02, Rt.a10, Rt.a11, Rt.a12, Rt.a20, Rt.a21, Rt.a22};
        const double xx[9] = {X.a00, X.a01, X.a02, X.a10, X.a11, X.a12, X.a20, X.a21, X.a22};
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
          {
            Jq[i * 6 + j] = rt[i * 3 + j];
            Jq[(i + 3) * 6 + j + 3] = rt[i * 3 + j];
            Jq[i * 6 + j + 3] = xx[i * 3 + j];
            Jq[(i + 3) * 6 + j] = 0.0;
          }
      }
    }
    else
    {
      st3(&sc.rx[0], out);
      st3(&sc.rx[3], w);
      if constexpr (DERIV)
      {
        const M3 X = (-1.0) * (J * Q * J);
        const double ji[9] = {J.a00, J.a01, J.a02, J.a10, J.a11, J.a12, J.a20, J.a21, J.a22};
        const double xx[9] = {X.a00, X.a01, X.a02, X.a10, X.a11, X.a12, X.a20, X.a21, X.a22};
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++)
          {
            sc.Jl[i * 6 + j] = ji[i * 3 + j];
            sc.Jl[(i + 3) * 6 + j + 3] = ji[i * 3 + j];
            sc.Jl[i * 6 + j + 3] = xx[i * 3 + j];
            sc.Jl[(i + 3) * 6 + j] = 0.0;
          }
      }
    }
  }

  // Derivative columns of the momentum, its rate and the foot points (lane = dof), then the base-acceleration derivatives
  // Agbi * [dtgt - dhd_dq | -dhd_dv | G_f | -Ag_j] (lane = column).  Inputs in the scratch: S, vel, acc and Fc for the solved
  // accelerations, Ic, hc, the composite velocity-product matrices Bm, com, footp, oR of the foot joints, Ag, Agbi, u, nu.
  template <class D>
  SMPC_DEV void kino_deriv_columns(KinoScratch<D, true> & sc, const StageIn<D> & in, double * Bm)
  {
    constexpr int NT = 64;
    constexpr int NV = D::NV, NF = D::NF;
    const DevModelSmall<D> & md = sc.ml;
    // ---- derivative columns: lane k < NV ----
    SMPC_LANES(NT)
    if (lane < NV)
    {
      const int k = lane;
      const int i = k < 6 ? 0 : k - 5;
      const int lam = md.parent[i];
      const SV s = ldsv(&sc.S[k * 6]);
      const SI Ici = ldsi(&sc.Ic[i * 10]);
      const SV vi = ldsv(&sc.vel[i * 6]);
      SV d = sv0(), Ak = sv0();
      if (lam >= 0)
      {
        const SV vl = ldsv(&sc.vel[lam * 6]);
        d = crm(vl, s);
        Ak = crm(ldsv(&sc.acc[lam * 6]), s) + crm(vl, d);
      }
      // sum over subtree bodies of  v_l x* (I_l y) - I_l (v_l x y)  for y = S_k and y = d_k: composite B_i y
      SV BS, Bd;
      {
        const double * Bc = &Bm[i * 36];
        const double sv[6] = {s.l.x, s.l.y, s.l.z, s.a.x, s.a.y, s.a.z};
        const double dv[6] = {d.l.x, d.l.y, d.l.z, d.a.x, d.a.y, d.a.z};
        double o1[6], o2[6];
#pragma unroll
        for (int r = 0; r < 6; r++)
        {
          double a1 = 0.0, a2 = 0.0;
#pragma unroll
          for (int m = 0; m < 6; m++)
          {
            const double bv = Bc[r * 6 + m];
            a1 += bv * sv[m];
            a2 += bv * dv[m];
          }
          o1[r] = a1;
          o2[r] = a2;
        }
        BS = SV{mk3(o1[0], o1[1], o1[2]), mk3(o1[3], o1[4], o1[5])};
        Bd = SV{mk3(o2[0], o2[1], o2[2]), mk3(o2[3], o2[4], o2[5])};
      }
      const SV hci = ldsv(&sc.hc[i * 6]);
      BS = BS + crf(s, hci);
      Bd = Bd + crf(d, hci);
      const SV dh = crf(s, hci) + Ici * d;
      const SV dF = crf(s, ldsv(&sc.Fc[i * 6])) + Ici * Ak + Bd;
      const SV dFv = BS + Ici * (crm(vi, s) + d);
      const V3 com = ld3(sc.com);
      const double im = 1.0 / md.total_mass;
      const V3 jc = im * (Ici * s).l;
      const SV h0 = ldsv(&sc.hc[0]);
      const SV F0 = ldsv(&sc.Fc[0]);
      const V3 dha = dh.a - cross(com, dh.l) - cross(jc, h0.l);
      const V3 dFa = dF.a - cross(com, dF.l) - cross(jc, F0.l);
      const V3 dFva = dFv.a - cross(com, dFv.l);
      sc.dh_dq[0 * NV + k] = dh.l.x;
      sc.dh_dq[1 * NV + k] = dh.l.y;
      sc.dh_dq[2 * NV + k] = dh.l.z;
      sc.dh_dq[3 * NV + k] = dha.x;
      sc.dh_dq[4 * NV + k] = dha.y;
      sc.dh_dq[5 * NV + k] = dha.z;
      sc.dhd_dq[0 * NV + k] = dF.l.x;
      sc.dhd_dq[1 * NV + k] = dF.l.y;
      sc.dhd_dq[2 * NV + k] = dF.l.z;
      sc.dhd_dq[3 * NV + k] = dFa.x;
      sc.dhd_dq[4 * NV + k] = dFa.y;
      sc.dhd_dq[5 * NV + k] = dFa.z;
      sc.dhd_dv[0 * NV + k] = dFv.l.x;
      sc.dhd_dv[1 * NV + k] = dFv.l.y;
      sc.dhd_dv[2 * NV + k] = dFv.l.z;
      sc.dhd_dv[3 * NV + k] = dFva.x;
      sc.dhd_dv[4 * NV + k] = dFva.y;
      sc.dhd_dv[5 * NV + k] = dFva.z;
      // feet: linear Jacobian columns, d hdot_tgt / dq, and the local-velocity constraint Jacobian columns: these
      // go straight into the contact rows of the knot's C block (zero rows for feet in the air) and into C_x^T nu
      V3 dt_ang = mk3(0, 0, 0);
      double cnq = 0.0, cnv = 0.0;
      for (int f = 0; f < NF; f++)
      {
        const int lj = md.foot_joint[f];
        const bool anc = (md.anc[lj] >> i) & 1u;
        const bool contact = (in.mask >> f) & 1u;
        const V3 pf = ld3(&sc.footp[f * 3]);
        V3 jf = mk3(0, 0, 0), cq = mk3(0, 0, 0), cv = mk3(0, 0, 0);
        if (anc)
        {
          jf = s.l + cross(s.a, pf);
          if (contact)
          {
            const M3 Rl = ldm3(&sc.oR[lj * 9]);
            cv = tmul(Rl, jf);
            if (lam >= 0)
              cq = tmul(Rl, d.l + cross(d.a, pf));
          }
        }
        sc.Jfoot[(f * 3 + 0) * NV + k] = jf.x;
        sc.Jfoot[(f * 3 + 1) * NV + k] = jf.y;
        sc.Jfoot[(f * 3 + 2) * NV + k] = jf.z;
        if (in.C_rows != nullptr)
        {
          double * cr = in.C_rows + (size_t)(3 * f) * D::NDX;
          cr[0 * D::NDX + k] = cq.x;
          cr[1 * D::NDX + k] = cq.y;
          cr[2 * D::NDX + k] = cq.z;
          cr[0 * D::NDX + NV + k] = cv.x;
          cr[1 * D::NDX + NV + k] = cv.y;
          cr[2 * D::NDX + NV + k] = cv.z;
        }
        const double * nuf = &sc.nu[D::NA + 3 * f];
        cnq += cq.x * nuf[0] + cq.y * nuf[1] + cq.z * nuf[2];
        cnv += cv.x * nuf[0] + cv.y * nuf[1] + cv.z * nuf[2];
        if (contact)
          dt_ang = dt_ang + cross(jf - jc, ld3(&sc.u[3 * f]));
      }
      sc.cn[k] = cnq;
      sc.cn[NV + k] = cnv;
      sc.dtgt[0 * NV + k] = dt_ang.x;
      sc.dtgt[1 * NV + k] = dt_ang.y;
      sc.dtgt[2 * NV + k] = dt_ang.z;
    if (in.prof) prof_tick(in.prof, 27, *in.tprev);
    }
    SMPC_LANES_END_WAVE
    // ---- base-acceleration derivatives: Agbi * [dtgt - dhd_dq | -dhd_dv | G_f | -Ag_j]; lane = column ----
    SMPC_LANES(NT)
    {
      constexpr int NCOL = 2 * NV + D::NU;
      static_assert(NCOL <= NT - 2, "one lane per column, two spare lanes");
      if (lane < NCOL)
      {
        const int c = lane;
        double rhs[6] = {0, 0, 0, 0, 0, 0};
        double * dst;
        int ld;
        if (c < NV)
        {
#pragma unroll
          for (int m = 0; m < 6; m++)
            rhs[m] = (m >= 3 ? sc.dtgt[(m - 3) * NV + c] : 0.0) - sc.dhd_dq[m * NV + c];
          dst = &sc.ab_dq()[c];
          ld = NV;
        }
        else if (c < 2 * NV)
        {
          const int k = c - NV;
#pragma unroll
          for (int m = 0; m < 6; m++)
            rhs[m] = -sc.dhd_dv[m * NV + k];
          dst = &sc.ab_dv()[k];
          ld = NV;
        }
        else
        {
          const int k = c - 2 * NV;
          if (k < 3 * NF)
          {
            const int f = k / 3, jj = k % 3;
            if ((in.mask >> f) & 1u)
            {
              const V3 rr = ld3(&sc.footp[f * 3]) - ld3(sc.com);
              const V3 e = mk3(jj == 0, jj == 1, jj == 2);
              const V3 xc = cross(rr, e); // column jj of [rr]x
              rhs[0] = e.x;
              rhs[1] = e.y;
              rhs[2] = e.z;
              rhs[3] = xc.x;
              rhs[4] = xc.y;
              rhs[5] = xc.z;
            }
          }
          else
          {
            const int kk = k - 3 * NF + 6;
#pragma unroll
            for (int m = 0; m < 6; m++)
              rhs[m] = -sc.Ag[m * NV + kk];
          }
          dst = &sc.ab_du()[k];
          ld = D::NU;
        }
#pragma unroll
        for (int r = 0; r < 6; r++)
        {
          double acc = 0.0;
#pragma unroll
          for (int m = 0; m < 6; m++)
            acc += sc.Agbi[r * 6 + m] * rhs[m];
          dst[r * ld] = acc;
        }
      }
    }
    SMPC_LANES_END_WAVE
    if (in.prof) prof_tick(in.prof, 28, *in.tprev);
  }

  // ---------------------------------------------------------------------------------------------
  // Tree + dynamics phases.  On return (all lanes synchronised) the scratch holds: kinematics,
  // composite quantities, Ag, hg, b0, hd, Agbi, a, xnext.  If DERIV, also acc/Fc for the solved
  // acceleration and every derivative column.
  // ---------------------------------------------------------------------------------------------
  // KIN_ONLY: stop after the kinematics / centroidal quantities (state front-end: no controls needed)
  template <class D, bool DERIV, bool KIN_ONLY = false>
  SMPC_DEV void kino_tree_phases(KinoScratch<D, DERIV> & sc, const StageIn<D> & in)
  {
    constexpr int NT = 64;
    constexpr int NJ = D::NJ, NV = D::NV, NQ = D::NQ, NF = D::NF;
    const DevModelSmall<D> & md = sc.ml;
    const DevModel<D> & mg = *in.md;
    (void)mg;
    const int nlev = md.nlevels;

    // ---- root -> leaf pass.  A SIMD pass costs the same with 1 or 64 active lanes, so only what really depends on
    //      the parent runs level by level (placement, motion column, velocity, bias acceleration: ~80 flops);
    //      the joint-local part before and the inertia / momentum / force part after run once for all joints ----
    const double * vq = &sc.x[NQ];
    SMPC_PLA(double, rl, NT, 9); // jpR * Rq of this lane's joint
    // geometry / inertia constants of this lane's joint: global -> registers once (the loads fly during the sincos)
    SMPC_PLA(double, jg, NT, 24); // jpR 0..8 | jpp 9..11 | mass 12 | com 13..15 | inertia 16..21
    SMPC_LANES(NT)
    {
      const int j = lane < NJ ? lane : 0;
#pragma unroll
      for (int i = 0; i < 9; i++)
        SMPC_PLV(jg)[i] = mg.jpR[j][i];
#pragma unroll
      for (int i = 0; i < 3; i++)
      {
        SMPC_PLV(jg)[9 + i] = mg.jpp[j][i];
        SMPC_PLV(jg)[13 + i] = mg.com[j][i];
      }
      SMPC_PLV(jg)[12] = mg.mass[j];
#pragma unroll
      for (int i = 0; i < 6; i++)
        SMPC_PLV(jg)[16 + i] = mg.inertia[j][i];
      SMPC_PLV(jg)[22] = 0.0;
      SMPC_PLV(jg)[23] = 0.0;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < NJ)
    {
      const int j = lane;
      if (j == 0)
      {
        const M3 R = quat_to_R(Quat{sc.x[3], sc.x[4], sc.x[5], sc.x[6]});
        const V3 p = ld3(sc.x);
        SV v = sv0();
        for (int k = 0; k < 6; k++)
        {
          const int col = k % 3;
          const V3 ax = col == 0 ? mk3(R.a00, R.a10, R.a20) : (col == 1 ? mk3(R.a01, R.a11, R.a21) : mk3(R.a02, R.a12, R.a22));
          const SV sk = k < 3 ? SV{ax, mk3(0, 0, 0)} : SV{cross(p, ax), ax};
          stsv(&sc.S[k * 6], sk);
          v = v + vq[k] * sk;
        }
        stm3(&sc.oR[0], R);
        st3(&sc.op_()[0], p);
        stsv(&sc.vel[0], v);
        stsv(&sc.acc[0], sv0());
      }
      else
      {
        const double ang = sc.x[6 + j];
        double s, c;
        sincos(ang, &s, &c); // one argument reduction for both
        const int jt = md.jtype[j];
        const M3 Rq = jt == 1 ? M3{1, 0, 0, 0, c, -s, 0, s, c} : (jt == 2 ? M3{c, 0, s, 0, 1, 0, -s, 0, c} : M3{c, -s, 0, s, c, 0, 0, 0, 1});
        const M3 Rl = ldm3(&SMPC_PLV(jg)[0]) * Rq;
        stm3(SMPC_PLV(rl), Rl);
      }
    }
    SMPC_LANES_END_WAVE
    for (int lvl = 1; lvl < nlev; lvl++)
    {
      SMPC_LANES(NT)
      if (lane > 0 && lane < NJ && md.level[lane] == lvl)
      {
        const int j = lane;
        const int par = md.parent[j];
        const M3 Rp = ldm3(&sc.oR[par * 9]);
        const M3 R = Rp * ldm3(SMPC_PLV(rl));
        const V3 p = ld3(&sc.op_()[par * 3]) + Rp * ld3(&SMPC_PLV(jg)[9]);
        const int col = md.jtype[j] - 1;
        const V3 ax = col == 0 ? mk3(R.a00, R.a10, R.a20) : (col == 1 ? mk3(R.a01, R.a11, R.a21) : mk3(R.a02, R.a12, R.a22));
        const SV sk = SV{cross(p, ax), ax};
        const SV vp = ldsv(&sc.vel[par * 6]);
        const double qd = vq[j + 5];
        stm3(&sc.oR[j * 9], R);
        st3(&sc.op_()[j * 3], p);
        stsv(&sc.S[(j + 5) * 6], sk);
        stsv(&sc.vel[j * 6], vp + qd * sk);
        stsv(&sc.acc[j * 6], ldsv(&sc.acc[par * 6]) + qd * crm(vp, sk));
      }
      SMPC_LANES_END_WAVE
    }
    SMPC_LANES(NT)
    if (lane < NJ)
    {
      const int j = lane;
      const M3 R = ldm3(&sc.oR[j * 9]);
      const V3 p = ld3(&sc.op_()[j * 3]);
      const SV v = ldsv(&sc.vel[j * 6]), a = ldsv(&sc.acc[j * 6]);
      // world inertia about the origin
      const double m = SMPC_PLV(jg)[12];
      const V3 c = R * ld3(&SMPC_PLV(jg)[13]) + p;
      const double * il = &SMPC_PLV(jg)[16];
      const M3 Il = M3{il[0], il[1], il[3], il[1], il[2], il[4], il[3], il[4], il[5]};
      const M3 Iw = R * Il * transpose(R);
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
      if constexpr (DERIV)
        stsi(&sc.I_()[j * 10], I);
      stsi(&sc.Ic[j * 10], I);
      const SV h = I * v;
      stsv(&sc.hc[j * 6], h);
      stsv(&sc.Fc[j * 6], I * a + crf(v, h));
    }
    else if (lane >= 32 && lane < 32 + NF)
    {
      const int f = lane - 32, j = md.foot_joint[f];
      st3(&sc.footp[f * 3], ldm3(&sc.oR[j * 9]) * ld3(mg.foot_p[f]) + ld3(&sc.op_()[j * 3]));
    }
    SMPC_LANES_END_WAVE
    if (in.prof) prof_tick(in.prof, 16, *in.tprev);
    // ---- per-body "velocity product" matrices  B_l y = v_l x* (I_l y) - I_l (v_l x y)  (lane = (body, column));
    //      their subtree sums give the sums over subtree bodies in the derivative columns as two 6x6 products.
    //      They live in the block the weighted-Jacobian tables take over later. ----
    double * Bm = nullptr;
    if constexpr (DERIV)
    {
      // [late block | WJl | JWJ] is one contiguous run: the evaluation part ends with the late block, the derivative part
      // starts with WJl
      typedef KinoScratchEval<D> EV;
      static_assert(EV::LATE_DOUBLES + D::NDX * 6 + 36 >= NJ * 36, "B matrices overlay late block | WJl | JWJ");
      static_assert(offsetof(EV, cval) + EV::LATE_DOUBLES * sizeof(double) == sizeof(EV) && D::NC % 2 == 0, "late block must end the evaluation part");
      static_assert(offsetof(KinoScratchDerivPart<D>, WJl) == 0 && offsetof(KinoScratchDerivPart<D>, JWJ) == D::NDX * 6 * sizeof(double),
                    "tables must start the derivative part");
      Bm = sc.cval;
      SMPC_LANES(NT)
      for (int idx = lane; idx < NJ * 6; idx += NT)
      {
        const int l = idx / 6, m = idx % 6;
        const SI Il = ldsi(&sc.I_()[l * 10]);
        const SV vl = ldsv(&sc.vel[l * 6]);
        const V3 e = mk3(m % 3 == 0, m % 3 == 1, m % 3 == 2), z = mk3(0, 0, 0);
        const SV y = m < 3 ? SV{e, z} : SV{z, e};
        const SV col = crf(vl, Il * y) - Il * crm(vl, y);
        double * dst = &Bm[l * 36 + m];
        dst[0] = col.l.x;
        dst[6] = col.l.y;
        dst[12] = col.l.z;
        dst[18] = col.a.x;
        dst[24] = col.a.y;
        dst[30] = col.a.z;
      }
      SMPC_LANES_END_WAVE
    }
    // ---- composites, leaf -> root: lane = one scalar of (Ic | hc | Fc | B), walking the joints in reverse
    //      topological order (parent < child) ----
    SMPC_LANES(NT)
    if (lane < 22 + (DERIV ? 36 : 0))
    {
      double * base = lane < 10 ? sc.Ic : (lane < 16 ? sc.hc : (lane < 22 ? sc.Fc : Bm));
      const int stride = lane < 10 ? 10 : (lane < 22 ? 6 : 36);
      const int e = lane < 10 ? lane : (lane < 16 ? lane - 10 : (lane < 22 ? lane - 16 : lane - 22));
      for (int j = NJ - 1; j >= 1; j--)
      {
        const int par = md.parent[j];
        base[par * stride + e] += base[j * stride + e];
      }
    }
    SMPC_LANES_END_WAVE
    if (in.prof) prof_tick(in.prof, 19, *in.tprev);
    // ---- CoM, centroidal map columns, hg, b0, hdot target, 6x6 inertia for the base solve ----
    SMPC_LANES(NT)
    {
      const SI I0 = ldsi(&sc.Ic[0]);
      const V3 com = (1.0 / I0.m) * I0.mc;
      if (lane < NV)
      {
        const int k = lane;
        const int j = k < 6 ? 0 : k - 5;
        const SV c = ldsi(&sc.Ic[j * 10]) * ldsv(&sc.S[k * 6]);
        const V3 ang = c.a - cross(com, c.l);
        sc.Ag[0 * NV + k] = c.l.x;
        sc.Ag[1 * NV + k] = c.l.y;
        sc.Ag[2 * NV + k] = c.l.z;
        sc.Ag[3 * NV + k] = ang.x;
        sc.Ag[4 * NV + k] = ang.y;
        sc.Ag[5 * NV + k] = ang.z;
      }
      else if (lane == 32)
      {
        st3(sc.com, com);
        const SV h0 = ldsv(&sc.hc[0]);
        st3(&sc.hg[0], h0.l);
        st3(&sc.hg[3], h0.a - cross(com, h0.l));
        const SV f0 = ldsv(&sc.Fc[0]);
        st3(&sc.b0[0], f0.l);
        st3(&sc.b0[3], f0.a - cross(com, f0.l));
      }
      else if (lane == 33)
      {
        V3 fl = md.total_mass * ld3(md.gravity);
        V3 fa = mk3(0, 0, 0);
        for (int f = 0; f < NF; f++)
          if ((in.mask >> f) & 1u)
          {
            const V3 F = ld3(&sc.u[3 * f]);
            fl = fl + F;
            fa = fa + cross(ld3(&sc.footp[f * 3]) - com, F);
          }
        st3(&sc.hd[0], fl);
        st3(&sc.hd[3], fa);
      }
    if (in.prof) prof_tick(in.prof, 20, *in.tprev);
    }
    SMPC_LANES_END_WAVE
    if constexpr (KIN_ONLY)
      return;
    // ---- M1 = Ic0^-1 T(c)^-1 in closed form.  It maps the centroidal momentum [m v_c; Jc w] to the base twist at the
    //      world origin [v_c + c x w; w]:   M1 = [[I/m, [c]x Jc^-1], [0, Jc^-1]],  Jc = J_o - m (|c|^2 I - c c^T) the
    //      composite rotational inertia about the CoM (3x3 SPD, inverted by its adjugate; every lane does it itself:
    //      one phase instead of six Gauss-Jordan sweeps with an LDS round trip and a division each) ----
    SMPC_LANES(NT)
    if (lane < 36)
    {
      const int r = lane / 6, c = lane % 6;
      const SI I0 = ldsi(&sc.Ic[0]);
      const double im = 1.0 / I0.m;
      const V3 cm = im * I0.mc;
      const double cc = dot(cm, cm);
      const double jxx = I0.jxx - I0.m * (cc - cm.x * cm.x), jyy = I0.jyy - I0.m * (cc - cm.y * cm.y), jzz = I0.jzz - I0.m * (cc - cm.z * cm.z);
      const double jxy = I0.jxy + I0.m * cm.x * cm.y, jxz = I0.jxz + I0.m * cm.x * cm.z, jyz = I0.jyz + I0.m * cm.y * cm.z;
      // adjugate of the symmetric 3x3
      const double a00 = jyy * jzz - jyz * jyz, a01 = jxz * jyz - jxy * jzz, a02 = jxy * jyz - jxz * jyy;
      const double a11 = jxx * jzz - jxz * jxz, a12 = jxy * jxz - jxx * jyz, a22 = jxx * jyy - jxy * jxy;
      const double idet = 1.0 / (jxx * a00 + jxy * a01 + jxz * a02);
      const M3 Ji = M3{a00 * idet, a01 * idet, a02 * idet, a01 * idet, a11 * idet, a12 * idet, a02 * idet, a12 * idet, a22 * idet};
      double val;
      if (c < 3)
        val = (r == c) ? im : 0.0;
      else
      {
        const V3 jc = c == 3 ? mk3(Ji.a00, Ji.a10, Ji.a20) : (c == 4 ? mk3(Ji.a01, Ji.a11, Ji.a21) : mk3(Ji.a02, Ji.a12, Ji.a22));
        const V3 top = cross(cm, jc); // column c - 3 of [c]x Jc^-1
        val = r == 0 ? top.x : (r == 1 ? top.y : (r == 2 ? top.z : (r == 3 ? jc.x : (r == 4 ? jc.y : jc.z))));
      }
      sc.gjB_()[lane] = val;
    if (in.prof) prof_tick(in.prof, 22, *in.tprev);
    }
    SMPC_LANES_END_WAVE
    // Agbi = X0^-1 * M1,  X0^-1 = [[R^T, -R^T [p]x],[0, R^T]]
    SMPC_LANES(NT)
    if (lane < 36)
    {
      const int r = lane / 6, c = lane % 6;
      const M3 R = ldm3(&sc.oR[0]);
      const V3 p = ld3(&sc.op_()[0]);
      const V3 ml = mk3(sc.gjB_()[0 * 6 + c], sc.gjB_()[1 * 6 + c], sc.gjB_()[2 * 6 + c]);
      const V3 ma = mk3(sc.gjB_()[3 * 6 + c], sc.gjB_()[4 * 6 + c], sc.gjB_()[5 * 6 + c]);
      V3 out;
      if (r < 3)
        out = tmul(R, ml - cross(p, ma));
      else
        out = tmul(R, ma);
      const int rr = r % 3;
      sc.Agbi[lane] = rr == 0 ? out.x : (rr == 1 ? out.y : out.z);
    if (in.prof) prof_tick(in.prof, 22, *in.tprev);
    }
    SMPC_LANES_END_WAVE
    // ---- base acceleration ----
    SMPC_LANES(NT)
    if (lane < NV)
    {
      double val;
      if (lane < 6)
      {
        val = 0.0;
        for (int m = 0; m < 6; m++)
        {
          double rhs = sc.hd[m] - sc.b0[m];
          for (int k = 6; k < NV; k++)
            rhs -= sc.Ag[m * NV + k] * sc.u[3 * NF + k - 6];
          val += sc.Agbi[lane * 6 + m] * rhs;
        }
      }
      else
        val = sc.u[3 * NF + lane - 6];
      sc.a[lane] = val;
    if (in.prof) prof_tick(in.prof, 23, *in.tprev);
    }
    SMPC_LANES_END_WAVE
    if (in.terminal)
    {
      SMPC_LANES(NT)
      if (lane < NV)
        sc.a[lane] = 0.0;
      SMPC_LANES_END_WAVE
    }
    // ---- x+ = x (+) [dt (v + dt a); dt a] ----
    SMPC_LANES(NT)
    {
      const double dt = md.dt;
      if (lane < 2)
        kino_se3_pair<D, DERIV>(sc, in, vq, dt, lane);
      else if (lane >= 6 && lane < NV)
        sc.xnext[lane + 1] = sc.x[lane + 1] + dt * (vq[lane] + dt * sc.a[lane]);
      if (lane >= 32 && lane < 32 + NV)
      {
        const int i = lane - 32;
        sc.xnext[NQ + i] = vq[i] + dt * sc.a[i];
      }
    if (in.prof) prof_tick(in.prof, 24, *in.tprev);
    }
    SMPC_LANES_END_WAVE

    if constexpr (DERIV)
    {
    // ---- accelerations and composite net forces for the solved a, without another tree sweep:
    //      acc_i(a) = acc_i(0) + sum_{k <= i} a_k S_k
    //      Fc_i(a)  = Fc_i(0) + sum_{k <= i} a_k Ic_i S_k + sum_{k strictly below i} a_k Ic_{j(k)} S_k ----
    //      With aS_k = a_k S_k and dacc_i = sum_{k <= i} aS_k the middle term is Ic_i dacc_i; the last one sums
    //      the per-dof vectors Ic_{j(k)} aS_k.  Both per-dof vectors are formed once (lane = dof), then lane = joint
    //      only adds.
    double * aS = sc.aS();
    double * IaS = sc.IaS();
    SMPC_LANES(NT)
    if (lane < NV)
    {
      const int k = lane, jk = k < 6 ? 0 : k - 5;
      const SV ask = sc.a[k] * ldsv(&sc.S[k * 6]);
      stsv(&aS[k * 6], ask);
      stsv(&IaS[k * 6], ldsi(&sc.Ic[jk * 10]) * ask);
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < NJ)
    {
      const int i = lane;
      const unsigned anci = md.anc[i];
      SV dacc = sv0(), sub = sv0();
      for (int k = 0; k < NV; k++)
      {
        const int jk = k < 6 ? 0 : k - 5;
        if ((anci >> jk) & 1u)
          dacc = dacc + ldsv(&aS[k * 6]);
        else if ((md.anc[jk] >> i) & 1u)
          sub = sub + ldsv(&IaS[k * 6]);
      }
      stsv(&sc.acc[i * 6], ldsv(&sc.acc[i * 6]) + dacc);
      stsv(&sc.Fc[i * 6], ldsv(&sc.Fc[i * 6]) + ldsi(&sc.Ic[i * 10]) * dacc + sub);
    }
    SMPC_LANES_END_WAVE
    if (in.prof) prof_tick(in.prof, 26, *in.tprev);
    kino_deriv_columns<D>(sc, in, Bm);
    } // if constexpr (DERIV)
  }

  // state residual rx = x (-) x_tgt and (optionally) Jlog6 of the base block, by one lane + vector part
  template <class D, bool DERIV>
  SMPC_DEV void kino_state_residual(KinoScratch<D, DERIV> & sc, const double * xt, int lane)
  {
    constexpr int NV = D::NV, NQ = D::NQ;
    // (base block rx[0:6] and its Jlog6: kino_se3_pair, together with the base integration)
    if (lane >= 6 && lane < NV)
      sc.rx[lane] = sc.x[lane + 1] - xt[lane + 1];

    if (lane < NV)
      sc.rx[NV + lane] = sc.x[NQ + lane] - xt[NQ + lane];
  }

  // Residuals, weighted residuals, cost value, constraint values.  Result: sc.red[0] = stage cost.
  template <class D, bool DERIV>
  SMPC_DEV void kino_cost_constraints(KinoScratch<D, DERIV> & sc, const StageIn<D> & in)
  {
    constexpr int NT = 64;
    constexpr int NV = D::NV, NF = D::NF, NU = D::NU, NDX = D::NDX, NA = D::NA, NC = D::NC;
    static_assert(NV <= 20 && NF * 3 <= 12 && NU <= 24 && NF <= 8 && NDX <= 40, "lane map of this kernel assumes a Go2-sized robot");
    const DevModelSmall<D> & md = sc.ml;
    const DevModel<D> & mg = *in.md;
    (void)mg;
    SMPC_LANES(NT)
    {
      kino_state_residual<D, DERIV>(sc, sc.in_x_tgt, lane);
      if (!in.terminal)
      {
        if (lane >= 32 && lane < 32 + NU)
          sc.ru[lane - 32] = sc.u[lane - 32] - sc.in_u_ref[lane - 32];
        if (lane >= 20 && lane < 20 + NF * 3)
        {
          const int i = lane - 20;
          sc.rf[i] = sc.footp[i] - sc.in_foot_ref[i];
        }
        // constraint values
        if (lane < NA)
          sc.cval[lane] = md.kinematics_limits ? sc.x[7 + lane] : 0.0;
        if (lane >= 56 && lane < 56 + NF)
        {
          const int f = lane - 56;
          V3 c = mk3(0, 0, 0);
          if ((in.mask >> f) & 1u)
          {
            const int l = md.foot_joint[f];
            const SV vl = ldsv(&sc.vel[l * 6]);
            c = tmul(ldm3(&sc.oR[l * 9]), vl.l + cross(vl.a, ld3(&sc.footp[f * 3])));
          }
          st3(&sc.cval[NA + 3 * f], c);
        }
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    {
      // weighted residuals
      if (lane < NDX)
      {
        double s = 0.0;
        if (md.w_diag)
          s = md.wxd[lane] * sc.rx[lane];
        else
        {
#pragma unroll
          for (int j = 0; j < NDX; j++)
            s += mg.w_xT[j * NDX + lane] * sc.rx[j];
        }
        sc.Wrx[lane] = s;
      }
      if (lane >= 40 && lane < 46)
      {
        const int i = lane - 40;
        double s = 0.0;
        const double sc10 = in.terminal ? 10.0 : 1.0; // terminal: 10 * w_cent (src/kinodynamics.cpp:361)
        for (int j = 0; j < 6; j++)
          s += sc10 * md.w_cent[i * 6 + j] * sc.hg[j];
        sc.Whg[i] = s;
      }
      if (!in.terminal)
      {
        if (lane >= 46 && lane < 52)
        {
          const int i = lane - 46;
          double s = 0.0;
          for (int j = 0; j < 6; j++)
            s += md.w_centder[i * 6 + j] * sc.hd[j];
          sc.Whd[i] = s;
        }
        if (lane >= 52 && lane < 52 + NF * 3)
        {
          const int i = lane - 52, f = i / 3, r = i % 3;
          double s = 0.0;
          for (int j = 0; j < 3; j++)
            s += sc.wframe_()[r * 3 + j] * sc.rf[f * 3 + j];
          sc.Wrf[i] = s;
        }
      }
    }
    SMPC_LANES_END_WAVE
    if (!in.terminal)
    {
      SMPC_LANES(NT)
      if (lane < NU)
      {
        double s = 0.0;
        if (md.w_diag)
          s = md.wud[lane] * sc.ru[lane];
        else
        {
#pragma unroll
          for (int j = 0; j < NU; j++)
            s += mg.w_uT[j * NU + lane] * sc.ru[j];
        }
        sc.Wru[lane] = s;
      }
      SMPC_LANES_END_WAVE
    }
    // cost = 1/2 sum of r_i (W r)_i over all residual groups: one term pair per lane, then a fixed-order reduction
    SMPC_LANES(NT)
    {
      double c = 0.0;
      if (lane < NDX)
        c = sc.rx[lane] * sc.Wrx[lane];
      else if (lane < NDX + 6)
        c = sc.hg[lane - NDX] * sc.Whg[lane - NDX];
      else if (!in.terminal && lane < NDX + 12)
        c = sc.hd[lane - NDX - 6] * sc.Whd[lane - NDX - 6];
      else if (!in.terminal && lane < NDX + 12 + NF * 3)
        c = sc.rf[lane - NDX - 12] * sc.Wrf[lane - NDX - 12];
      if (!in.terminal && lane < NU)
        c += sc.ru[lane] * sc.Wru[lane];
      sc.part()[lane] = c;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < 8)
    {
      double c = 0.0;
#pragma unroll
      for (int i = 0; i < 8; i++)
        c += sc.part()[lane * 8 + i];
      sc.part8()[lane] = c;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double c = 0.0;
#pragma unroll
      for (int i = 0; i < 8; i++)
        c += sc.part8()[i];
      sc.red[0] = 0.5 * c;
    }
    SMPC_LANES_END_WAVE
    static_assert(NDX + 12 + NF * 3 <= NT, "one residual term per lane");
    (void)NC;
  }

  // AL multipliers for this stage (reference: SolverProxDDP computeMultipliers).
  // Inputs (LDS): sc.e, sc.cval, sc.lam_next (lambda_{t+1}), sc.nu.  Centres from global.
  // Outputs: sc.lamp, sc.vplus, sc.act, sc.red[1] = penalty part of the merit, sc.red[2] = primal infeasibility
  template <class D, bool DERIV>
  SMPC_DEV void kino_multipliers(KinoScratch<D, DERIV> & sc, const StageIn<D> & in, const double * lam_e, const double * nu_e)
  {
    constexpr int NT = 64;
    constexpr int NDX = D::NDX, NC = D::NC, NA = D::NA;
    const DevModelSmall<D> & md = sc.ml;
    const DevModel<D> & mg = *in.md;
    (void)mg;
    const double mu = md.mu;
    SMPC_LANES(NT)
    {
      if (lane < NDX)
        sc.lamp[lane] = lam_e[lane] + sc.e[lane] / mu;
      for (int i = lane; i < NC; i += NT)
      {
        int kind; // 0 absent 1 eq 2 box
        if (i < NA)
          kind = md.kinematics_limits ? 2 : 0;
        else
          kind = ((in.mask >> ((i - NA) / 3)) & 1u) ? 1 : 0;
        double vp = 0.0;
        int act = 0;
        if (kind != 0)
        {
          const double z = sc.cval[i] + mu * nu_e[i];
          double proj = 0.0;
          if (kind == 2)
            proj = fmin(fmax(z, md.qmin[i]), md.qmax[i]);
          vp = (z - proj) / mu;
          act = (z != proj) || kind == 1;
        }
        sc.vplus[i] = vp;
        sc.act[i] = act;
      }
    }
    SMPC_LANES_END_WAVE
    // merit penalty and primal infeasibility: lane i < NDX takes dynamics row i, lane NDX + i constraint row i
    static_assert(NDX + NC <= NT, "one row per lane");
    SMPC_LANES(NT)
    {
      double pen = 0.0, prim = 0.0;
      if (lane < NDX)
      {
        const double lp = sc.lamp[lane], dl = lp - sc.lam_next[lane];
        pen = 0.5 * mu * (lp * lp + dl * dl);
        prim = fabs(sc.e[lane]);
      }
      else if (lane < NDX + NC)
      {
        const int i = lane - NDX;
        const double vp = sc.vplus[i], dv = vp - sc.nu[i];
        pen = 0.5 * mu * (vp * vp + dv * dv);
        if (i < NA)
        {
          if (md.kinematics_limits)
            prim = fmax(fmax(sc.cval[i] - md.qmax[i], md.qmin[i] - sc.cval[i]), 0.0);
        }
        else if ((in.mask >> ((i - NA) / 3)) & 1u)
          prim = fabs(sc.cval[i]);
      }
      sc.part()[lane] = pen;
      sc.part2()[lane] = prim;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane < 8)
    {
      double pen = 0.0, prim = 0.0;
#pragma unroll
      for (int i = 0; i < 8; i++)
      {
        pen += sc.part()[lane * 8 + i];
        prim = fmax(prim, sc.part2()[lane * 8 + i]);
      }
      sc.part8()[lane] = pen;
      sc.part8()[8 + lane] = prim;
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double pen = 0.0, prim = 0.0;
#pragma unroll
      for (int i = 0; i < 8; i++)
      {
        pen += sc.part8()[i];
        prim = fmax(prim, sc.part8()[8 + i]);
      }
      sc.red[1] = pen;
      sc.red[2] = prim;
    }
    SMPC_LANES_END_WAVE
  }

  // Friction-cone rows of the stage (force_cone): per foot in contact  [ -f_z + 1e-4 ; f_x^2 + f_y^2 - mu^2 f_z^2 ] <= 0 on the
  // force part of u (CentroidalFrictionConeResidual in NegativeOrthant, reference src/kinodynamics.cpp:124-129).  After
  // kino_multipliers: projects, adds the penalty / infeasibility to red[1] / red[2], leaves  value | nu+ | active | nu  in the
  // tail of tree block A (dead since the derivative columns), and -- derivative pass -- writes the rows' block of the knot (ek).
  //   nu: multipliers of the rows at the evaluation point, nu_e: AL centres (both 2 NF, global).
  template <class D, bool DERIV>
  SMPC_HD double * kino_cone_scratch(KinoScratch<D, DERIV> & sc)
  {
    static_assert(D::NJ * 9 + D::NV * 6 + D::NJ * 40 >= 612 + 8 * D::NF, "cone scratch behind the assembly tables of tree block A");
    return sc.oR + 612;
  }
  // Jacobian row w (0 / 1) of a foot's cone block w.r.t. its force, at force (fx, fy, fz)
  SMPC_HD void kino_cone_jac(int w, double fx, double fy, double fz, double mu2, double * out3)
  {
    out3[0] = w == 0 ? 0.0 : 2.0 * fx;
    out3[1] = w == 0 ? 0.0 : 2.0 * fy;
    out3[2] = w == 0 ? -1.0 : -2.0 * mu2 * fz;
  }
  template <class D, bool DERIV>
  SMPC_DEV void kino_cone_rows(KinoScratch<D, DERIV> & sc, const StageIn<D> & in, double mu2, const double * nu, const double * dnu,
                               double alpha, const double * nu_e, double * ek)
  {
    constexpr int NT = 64, NF = D::NF, NE = 2 * NF;
    const double mu = sc.ml.mu;
    double * cs = kino_cone_scratch<D, DERIV>(sc);
    SMPC_LANES(NT)
    if (lane < NE)
    {
      const int f = lane / 2, w = lane % 2;
      const bool on = (in.mask >> f) & 1u;
      const double fx = sc.u[3 * f], fy = sc.u[3 * f + 1], fz = sc.u[3 * f + 2];
      const double c = w == 0 ? -fz + 1e-4 : fx * fx + fy * fy - mu2 * fz * fz;
      const double v = nu[lane] + (dnu ? alpha * dnu[lane] : 0.0);
      double vp = 0.0;
      int act = 0;
      if (on)
      {
        const double z = c + mu * nu_e[lane];
        const double proj = fmin(z, 0.0);
        vp = (z - proj) / mu;
        act = z != proj;
      }
      cs[lane] = on ? c : 0.0;
      cs[NE + lane] = vp;
      cs[2 * NE + lane] = act ? 1.0 : 0.0;
      cs[3 * NE + lane] = v;
      if (ek != nullptr)
      {
        double jr[3];
        kino_cone_jac(w, fx, fy, fz, mu2, jr);
        for (int a = 0; a < 3; a++)
          ek[lane * 3 + a] = act ? jr[a] : 0.0;
        ek[3 * NE + lane] = mu * (vp - v);
        ek[4 * NE + lane] = act ? 2.0 * vp - v : 0.0;
        ek[5 * NE + lane] = act ? 1.0 : 0.0;
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double pen = 0.0, prim = sc.red[2];
      for (int i = 0; i < NE; i++)
      {
        const double vp = cs[NE + i], dv = vp - cs[3 * NE + i];
        pen += 0.5 * mu * (vp * vp + dv * dv);
        if ((in.mask >> (i / 2)) & 1u)
          prim = fmax(prim, fmax(cs[i], 0.0));
      }
      sc.red[1] += pen;
      sc.red[2] = prim;
    }
    SMPC_LANES_END_WAVE
  }

  // land_cstr rows of the stage: for every foot that lands here (in contact now, not in the previous stage of the cycle) the
  // equality p_z(foot) = z of its contact pose (reference src/kinodynamics.cpp:134-146).  After kino_multipliers / kino_cone_rows:
  // value | nu+ | nu  go behind the cone scratch; the penalty / infeasibility are added to red[1] / red[2]; the derivative pass
  // writes the rows (d p_z / dq: z row of the foot's world-frame point Jacobian), d and 2 nu+ - nu into the knot block lk.
  template <class D, bool DERIV>
  SMPC_HD double * kino_land_scratch(KinoScratch<D, DERIV> & sc)
  {
    static_assert(D::NJ * 9 + D::NV * 6 + D::NJ * 40 >= 612 + 8 * D::NF + 3 * D::NF, "land scratch behind the cone scratch");
    return sc.oR + 612 + 8 * D::NF;
  }
  template <class D, bool DERIV>
  SMPC_DEV void kino_land_rows(KinoScratch<D, DERIV> & sc, const StageIn<D> & in, unsigned land, const double * land_z, const double * nu,
                               const double * dnu, double alpha, const double * nu_e, double * lk)
  {
    constexpr int NT = 64, NF = D::NF, NV = D::NV;
    const double mu = sc.ml.mu;
    double * ls = kino_land_scratch<D, DERIV>(sc);
    const unsigned rows = land & in.mask;
    SMPC_LANES(NT)
    {
      if (lane < NF)
      {
        const int f = lane;
        const bool on = (rows >> f) & 1u;
        const double c = on ? sc.footp[3 * f + 2] - land_z[f] : 0.0;
        const double v = nu[f] + (dnu ? alpha * dnu[f] : 0.0);
        const double vp = on ? nu_e[f] + c / mu : 0.0; // equality row: always active
        ls[f] = c;
        ls[NF + f] = vp;
        ls[2 * NF + f] = v;
        if (lk != nullptr)
        {
          lk[NF * NV + f] = mu * (vp - v);
          lk[NF * NV + NF + f] = on ? 2.0 * vp - v : 0.0;
        }
      }
      if constexpr (DERIV)
      {
        if (lk != nullptr)
          for (int idx = lane; idx < NF * NV; idx += NT)
          {
            const int f = idx / NV, k = idx % NV;
            lk[idx] = ((rows >> f) & 1u) ? sc.Jfoot[(3 * f + 2) * NV + k] : 0.0; // (world-frame point Jacobian, z row)
          }
      }
    }
    SMPC_LANES_END_WAVE
    SMPC_LANES(NT)
    if (lane == 0)
    {
      double pen = 0.0, prim = sc.red[2];
      for (int f = 0; f < NF; f++)
      {
        const double vp = ls[NF + f], dv = vp - ls[2 * NF + f];
        pen += 0.5 * mu * (vp * vp + dv * dv);
        prim = fmax(prim, fabs(ls[f]));
      }
      sc.red[1] += pen;
      sc.red[2] = prim;
    }
    SMPC_LANES_END_WAVE
  }
} // namespace smpc
