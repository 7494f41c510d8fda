// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_rigid.hpp: rigid-body algorithms on the smpc_robot_model tree, restating what the reference
// obtains from Pinocchio 3.8 (un-vendored; pixi.lock:154) through Aligator's kinodynamics model:
//   forwardKinematics / updateFramePlacements / computeCentroidalMomentum / ccrba / dccrba /
//   computeCentroidalDynamicsDerivatives / jacobianCenterOfMass / getFrameJacobian /
//   getFrameVelocityDerivatives  (call sites: src/robot-handler.cpp:119-139 and, transitively,
//   src/kinodynamics.cpp:56-59,85-87,110-111).
//
// Formulation: everything is expressed in the WORLD frame with spatial vectors [linear; angular]
// taken at the world origin.  For dof k of joint i (parent lam):
//     S_k   = world joint axis column               d_k = v_lam x S_k
//     A_k   = a_lam x S_k + v_lam x d_k
//     dh/dq_k     = S_k x* hc_i + Ic_i d_k
//     dhdot/dq_k  = S_k x* Fc_i + Ic_i A_k + Bc_i d_k
//     dhdot/dv_k  = Bc_i S_k + Ic_i (v_i x S_k + d_k)
//     dhdot/da_k  = Ic_i S_k  (= Ag column)
// with composite (subtree) inertia Ic, momentum hc, force Fc and Bc = sum_l (v_l x* I_l - I_l v_l x
// + (. x* h_l)).  Centroidal quantities are obtained by translating to the CoM.
#pragma once
#include "../include/smpc_robot.h"
#include "orc_se3.hpp"

namespace orc
{
  struct SV // spatial vector [lin; ang]
  {
    V3 l, a;
  };
  inline SV sv_zero() { return SV{v3(0, 0, 0), v3(0, 0, 0)}; }
  inline SV operator+(const SV & x, const SV & y) { return SV{x.l + y.l, x.a + y.a}; }
  inline SV operator-(const SV & x, const SV & y) { return SV{x.l - y.l, x.a - y.a}; }
  inline SV operator*(double s, const SV & x) { return SV{s * x.l, s * x.a}; }
  // motion x motion
  inline SV crm(const SV & v, const SV & m) { return SV{cross(v.a, m.l) + cross(v.l, m.a), cross(v.a, m.a)}; }
  // motion x* force
  inline SV crf(const SV & v, const SV & f) { return SV{cross(v.a, f.l), cross(v.a, f.a) + cross(v.l, f.l)}; }

  // spatial inertia about the world origin: mass, m*c, rotational inertia J about the origin
  struct SI
  {
    double m;
    V3 mc;
    M3 J;
  };
  inline SI si_zero() { return SI{0.0, v3(0, 0, 0), m3_zero()}; }
  inline SI operator+(const SI & a, const SI & b) { return SI{a.m + b.m, a.mc + b.mc, a.J + b.J}; }
  inline SV operator*(const SI & I, const SV & v)
  {
    return SV{I.m * v.l + cross(v.a, I.mc), I.J * v.a + cross(I.mc, v.l)};
  }
  inline Mat si_matrix(const SI & I)
  {
    Mat M(6, 6);
    M3 C = skew(I.mc);
    for (int i = 0; i < 3; i++)
    {
      M(i, i) = I.m;
      for (int j = 0; j < 3; j++)
      {
        M(i, j + 3) = -C(i, j);
        M(i + 3, j) = C(i, j);
        M(i + 3, j + 3) = I.J(i, j);
      }
    }
    return M;
  }
  inline Mat crm_matrix(const SV & v)
  {
    Mat M(6, 6);
    M3 W = skew(v.a), V = skew(v.l);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
      {
        M(i, j) = W(i, j);
        M(i, j + 3) = V(i, j);
        M(i + 3, j + 3) = W(i, j);
      }
    return M;
  }
  inline Mat crf_matrix(const SV & v)
  {
    Mat M(6, 6);
    M3 W = skew(v.a), V = skew(v.l);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
      {
        M(i, j) = W(i, j);
        M(i + 3, j) = V(i, j);
        M(i + 3, j + 3) = W(i, j);
      }
    return M;
  }
  // d -> d x* h as a 6x6 matrix
  inline Mat crf_of_force_matrix(const SV & h)
  {
    Mat M(6, 6);
    M3 F = skew(h.l), N = skew(h.a);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
      {
        M(i, j + 3) = -F(i, j);
        M(i + 3, j) = -F(i, j);
        M(i + 3, j + 3) = -N(i, j);
      }
    return M;
  }
  inline Vec sv_vec(const SV & s) { return Vec{s.l[0], s.l[1], s.l[2], s.a[0], s.a[1], s.a[2]}; }
  inline SV vec_sv(const Vec & v) { return SV{v3(v[0], v[1], v[2]), v3(v[3], v[4], v[5])}; }

  struct Rigid
  {
    const smpc_robot_model * M;
    int nj, nv, nq;
    std::vector<int> jv;      // first v index of joint
    std::vector<int> dof2j;   // dof -> joint
    std::vector<SE3> oMi;
    std::vector<SV> S;        // world columns (nv)
    std::vector<SV> vel, acc; // body spatial velocity / acceleration (world)
    std::vector<SI> I, Ic;
    std::vector<SV> h, hc, F, Fc;
    std::vector<Mat> Bc;
    std::vector<V3> foot_p;   // world foot positions
    V3 com;
    double mass;

    explicit Rigid(const smpc_robot_model * m) : M(m)
    {
      nj = m->njoints;
      nv = m->nv;
      nq = m->nq;
      jv.resize(nj);
      dof2j.resize(nv);
      for (int j = 0; j < nj; j++)
        jv[j] = j == 0 ? 0 : 5 + j;
      for (int k = 0; k < nv; k++)
        dof2j[k] = k < 6 ? 0 : k - 5;
      oMi.resize(nj);
      S.resize(nv);
      vel.resize(nj);
      acc.resize(nj);
      I.resize(nj);
      Ic.resize(nj);
      h.resize(nj);
      hc.resize(nj);
      F.resize(nj);
      Fc.resize(nj);
      Bc.resize(nj);
      foot_p.resize(m->nfeet);
      mass = m->total_mass;
    }

    static M3 axis_rot(int jt, double a)
    {
      V3 w = v3(jt == 1 ? a : 0, jt == 2 ? a : 0, jt == 3 ? a : 0);
      return exp3(w);
    }

    // positions: oMi, S columns, world inertias, composite inertia, CoM, foot positions
    void fk(const double * q)
    {
      oMi[0].R = quat_to_R(q + 3);
      oMi[0].p = v3(q[0], q[1], q[2]);
      for (int j = 1; j < nj; j++)
      {
        SE3 Jp;
        for (int i = 0; i < 9; i++)
          Jp.R.m[i] = M->jp_R[j][i];
        Jp.p = v3(M->jp_p[j][0], M->jp_p[j][1], M->jp_p[j][2]);
        SE3 Jq{axis_rot(M->jtype[j], q[6 + j]), v3(0, 0, 0)};
        oMi[j] = oMi[M->parent[j]] * (Jp * Jq);
      }
      // columns
      for (int k = 0; k < 3; k++)
      {
        V3 e = v3(k == 0, k == 1, k == 2);
        V3 Re = oMi[0].R * e;
        S[k] = SV{Re, v3(0, 0, 0)};
        S[3 + k] = SV{cross(oMi[0].p, Re), Re};
      }
      for (int j = 1; j < nj; j++)
      {
        int jt = M->jtype[j];
        V3 ax = oMi[j].R * v3(jt == 1, jt == 2, jt == 3);
        S[jv[j]] = SV{cross(oMi[j].p, ax), ax};
      }
      // inertias
      for (int j = 0; j < nj; j++)
      {
        V3 cl = v3(M->com[j][0], M->com[j][1], M->com[j][2]);
        V3 c = oMi[j].R * cl + oMi[j].p;
        M3 Il;
        Il(0, 0) = M->inertia[j][0];
        Il(0, 1) = Il(1, 0) = M->inertia[j][1];
        Il(1, 1) = M->inertia[j][2];
        Il(0, 2) = Il(2, 0) = M->inertia[j][3];
        Il(1, 2) = Il(2, 1) = M->inertia[j][4];
        Il(2, 2) = M->inertia[j][5];
        M3 Iw = oMi[j].R * Il * tr(oMi[j].R);
        M3 C = skew(c);
        I[j].m = M->mass[j];
        I[j].mc = M->mass[j] * c;
        I[j].J = Iw + (-M->mass[j]) * (C * C);
      }
      for (int j = 0; j < nj; j++)
        Ic[j] = I[j];
      for (int j = nj - 1; j > 0; j--)
        Ic[M->parent[j]] = Ic[M->parent[j]] + Ic[j];
      com = (1.0 / Ic[0].m) * Ic[0].mc;
      for (int f = 0; f < M->nfeet; f++)
      {
        int j = M->foot_joint[f];
        foot_p[f] = oMi[j].R * v3(M->foot_p[f][0], M->foot_p[f][1], M->foot_p[f][2]) + oMi[j].p;
      }
    }

    // velocities + momenta (needs fk)
    void velocities(const double * v)
    {
      for (int j = 0; j < nj; j++)
      {
        SV vj = sv_zero();
        int nd = j == 0 ? 6 : 1;
        for (int k = 0; k < nd; k++)
          vj = vj + v[jv[j] + k] * S[jv[j] + k];
        vel[j] = j == 0 ? vj : vel[M->parent[j]] + vj;
        h[j] = I[j] * vel[j];
      }
      for (int j = 0; j < nj; j++)
        hc[j] = h[j];
      for (int j = nj - 1; j > 0; j--)
        hc[M->parent[j]] = hc[M->parent[j]] + hc[j];
    }

    // accelerations (a may be null = zero joint accelerations) and net forces F = I a + v x* I v
    void forces(const double * v, const double * a)
    {
      for (int j = 0; j < nj; j++)
      {
        SV aj = sv_zero();
        int nd = j == 0 ? 6 : 1;
        for (int k = 0; k < nd; k++)
        {
          if (a)
            aj = aj + a[jv[j] + k] * S[jv[j] + k];
        }
        if (j > 0)
        {
          // Sdot_k qd_k = (v_parent x S_k) qd_k
          aj = aj + v[jv[j]] * crm(vel[M->parent[j]], S[jv[j]]);
          acc[j] = acc[M->parent[j]] + aj;
        }
        else
          acc[j] = aj; // free-flyer: sum_k (v_0 x S_k) qd_k = v_0 x v_0 = 0
        F[j] = I[j] * acc[j] + crf(vel[j], h[j]);
      }
      for (int j = 0; j < nj; j++)
        Fc[j] = F[j];
      for (int j = nj - 1; j > 0; j--)
        Fc[M->parent[j]] = Fc[M->parent[j]] + Fc[j];
    }

    void compute_Bc()
    {
      for (int j = 0; j < nj; j++)
      {
        Mat Im = si_matrix(I[j]);
        Mat B = mul(crf_matrix(vel[j]), Im);
        add_inplace(B, mul(Im, crm_matrix(vel[j])), -1.0);
        add_inplace(B, crf_of_force_matrix(h[j]));
        Bc[j] = B;
      }
      for (int j = nj - 1; j > 0; j--)
        add_inplace(Bc[M->parent[j]], Bc[j]);
    }

    // translate a force-type spatial vector from world origin to the CoM
    SV to_com(const SV & f) const { return SV{f.l, f.a - cross(com, f.l)}; }

    // centroidal momentum matrix (6 x nv, centroidal frame) -- Pinocchio data.Ag
    Mat Ag() const
    {
      Mat A(6, nv);
      for (int k = 0; k < nv; k++)
      {
        SV c = to_com(Ic[dof2j[k]] * S[k]);
        for (int i = 0; i < 3; i++)
        {
          A(i, k) = c.l[i];
          A(i + 3, k) = c.a[i];
        }
      }
      return A;
    }
    // centroidal momentum hg (needs velocities)
    SV hg() const { return to_com(hc[0]); }
    // dAg * v  (needs forces(v, nullptr))
    SV dAg_v() const { return to_com(Fc[0]); }

    // CoM Jacobian column
    V3 Jcom_col(int k) const { return (1.0 / mass) * (Ic[dof2j[k]] * S[k]).l; }

    bool is_ancestor_dof(int k, int joint) const
    {
      int jk = dof2j[k];
      for (int j = joint; j >= 0; j = M->parent[j])
        if (j == jk)
          return true;
      return false;
    }
    // world linear velocity of foot point induced by dof k (LOCAL_WORLD_ALIGNED linear Jacobian col)
    V3 Jfoot_col(int f, int k) const
    {
      if (!is_ancestor_dof(k, M->foot_joint[f]))
        return v3(0, 0, 0);
      return S[k].l + cross(S[k].a, foot_p[f]);
    }

    // Centroidal dynamics derivatives (needs fk, velocities, forces(v,a), compute_Bc):
    // dh_dq, dhdot_dq, dhdot_dv (6 x nv, centroidal frame). dhdot_da = Ag().
    void centroidal_derivatives(Mat & dh_dq, Mat & dhdot_dq, Mat & dhdot_dv) const
    {
      dh_dq.resize(6, nv);
      dhdot_dq.resize(6, nv);
      dhdot_dv.resize(6, nv);
      const SV h0 = hc[0], F0 = Fc[0];
      for (int k = 0; k < nv; k++)
      {
        const int i = dof2j[k];
        const int lam = M->parent[i];
        SV d = sv_zero(), A = sv_zero();
        if (lam >= 0)
        {
          d = crm(vel[lam], S[k]);
          A = crm(acc[lam], S[k]) + crm(vel[lam], d);
        }
        SV dh = crf(S[k], hc[i]) + Ic[i] * d;
        SV Bd = vec_sv(mul(Bc[i], sv_vec(d)));
        SV dF = crf(S[k], Fc[i]) + Ic[i] * A + Bd;
        SV BS = vec_sv(mul(Bc[i], sv_vec(S[k])));
        SV dFv = BS + Ic[i] * (crm(vel[i], S[k]) + d);
        V3 jc = Jcom_col(k);
        SV dhg = to_com(dh);
        dhg.a = dhg.a - cross(jc, h0.l);
        SV dFg = to_com(dF);
        dFg.a = dFg.a - cross(jc, F0.l);
        SV dFvg = to_com(dFv);
        for (int r = 0; r < 3; r++)
        {
          dh_dq(r, k) = dhg.l[r];
          dh_dq(r + 3, k) = dhg.a[r];
          dhdot_dq(r, k) = dFg.l[r];
          dhdot_dq(r + 3, k) = dFg.a[r];
          dhdot_dv(r, k) = dFvg.l[r];
          dhdot_dv(r + 3, k) = dFvg.a[r];
        }
      }
    }

    // LOCAL linear velocity of foot frame f (3) and its derivatives wrt q and v (3 x nv each)
    void foot_local_velocity(int f, V3 & c, Mat & dc_dq, Mat & dc_dv) const
    {
      const int l = M->foot_joint[f];
      M3 Rt = tr(oMi[l].R); // foot frame rotation = joint rotation (identity placement rotation)
      V3 vw = vel[l].l + cross(vel[l].a, foot_p[f]);
      c = Rt * vw;
      dc_dq.resize(3, nv);
      dc_dv.resize(3, nv);
      for (int k = 0; k < nv; k++)
      {
        if (!is_ancestor_dof(k, l))
          continue;
        const int lam = M->parent[dof2j[k]];
        V3 cv = Rt * (S[k].l + cross(S[k].a, foot_p[f]));
        V3 cq = v3(0, 0, 0);
        if (lam >= 0)
        {
          SV d = crm(vel[lam], S[k]);
          cq = Rt * (d.l + cross(d.a, foot_p[f]));
        }
        for (int r = 0; r < 3; r++)
        {
          dc_dv(r, k) = cv[r];
          dc_dq(r, k) = cq[r];
        }
      }
    }
    // the same for a 6-D foot: the whole LOCAL frame velocity [R^T v_point ; R^T omega] (FrameVelocityResidual(..., LOCAL) unsliced,
    // reference src/kinodynamics.cpp:110-123) and its derivatives (Pinocchio getFrameVelocityDerivatives, LOCAL):
    //   d(R^T u)/dq_k = R^T (du/dq_k - S_k.a x u)  -- the rigid turn of the subtree by S_k cancels, what is left is d_k = v_parent x S_k
    void foot_local_velocity6(int f, Vec & c, Mat & dc_dq, Mat & dc_dv) const
    {
      const int l = M->foot_joint[f];
      const M3 Rt = tr(oMi[l].R);
      const V3 vl = Rt * (vel[l].l + cross(vel[l].a, foot_p[f])), va = Rt * vel[l].a;
      c = Vec{vl[0], vl[1], vl[2], va[0], va[1], va[2]};
      dc_dq.resize(6, nv);
      dc_dv.resize(6, nv);
      for (int k = 0; k < nv; k++)
      {
        if (!is_ancestor_dof(k, l))
          continue;
        const int lam = M->parent[dof2j[k]];
        const V3 cvl = Rt * (S[k].l + cross(S[k].a, foot_p[f])), cva = Rt * S[k].a;
        V3 cql = v3(0, 0, 0), cqa = v3(0, 0, 0);
        if (lam >= 0)
        {
          const SV d = crm(vel[lam], S[k]);
          cql = Rt * (d.l + cross(d.a, foot_p[f]));
          cqa = Rt * d.a;
        }
        for (int r = 0; r < 3; r++)
        {
          dc_dv(r, k) = cvl[r];
          dc_dv(3 + r, k) = cva[r];
          dc_dq(r, k) = cql[r];
          dc_dq(3 + r, k) = cqa[r];
        }
      }
    }
  };

  // ---- configuration-space (manifold) operations: SE3 x R^(nq-7) for q, vector space for v ----
  inline void q_integrate(int nv, const double * q, const double * dq, double * out)
  {
    SE3 M0{quat_to_R(q + 3), v3(q[0], q[1], q[2])};
    SE3 E = exp6(dq);
    V3 p = M0.p + M0.R * E.p;
    double dqt[4], qn[4];
    exp3_quat(v3(dq[3], dq[4], dq[5]), dqt);
    quat_mul(q + 3, dqt, qn);
    // renormalise (first-order, as Pinocchio's firstOrderNormalize is only valid near 1; use exact)
    double n = std::sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    out[0] = p[0];
    out[1] = p[1];
    out[2] = p[2];
    for (int i = 0; i < 4; i++)
      out[3 + i] = qn[i] / n;
    for (int i = 6; i < nv; i++)
      out[i + 1] = q[i + 1] + dq[i];
  }
  // x = (q, v): out = x (+) dx
  inline void x_integrate(int nq, int nv, const double * x, const double * dx, double * out)
  {
    q_integrate(nv, x, dx, out);
    for (int i = 0; i < nv; i++)
      out[nq + i] = x[nq + i] + dx[nv + i];
  }
  // out = x1 (-) x0  (tangent at x0)
  inline void x_difference(int nq, int nv, const double * x0, const double * x1, double * out)
  {
    SE3 M0{quat_to_R(x0 + 3), v3(x0[0], x0[1], x0[2])};
    SE3 M1{quat_to_R(x1 + 3), v3(x1[0], x1[1], x1[2])};
    log6(inv(M0) * M1, out);
    for (int i = 6; i < nv; i++)
      out[i] = x1[i + 1] - x0[i + 1];
    for (int i = 0; i < nv; i++)
      out[nv + i] = x1[nq + i] - x0[nq + i];
  }
} // namespace orc
