// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_linalg.hpp header).  PARITY UNPINNED.
//
// orc_se3.hpp: SO(3)/SE(3) helpers restating the Pinocchio 3.8 conventions used by the reference
// through MultibodyPhaseSpace (reference: src/kinodynamics.cpp:17,46; src/robot-handler.cpp:81-96):
//   motion vectors are [linear; angular]; free-flyer q = [p; quat(x,y,z,w)]; integrate is
//   M * exp6(nu) (right/local perturbation); difference is log6(M0^-1 M1).
// [UPSTREAM-RECALL] pinocchio/spatial/explog.hpp, multibody/liegroup/special-euclidean.hpp.
#pragma once
#include "orc_linalg.hpp"

namespace orc
{
  struct V3
  {
    double x[3];
    double & operator[](int i) { return x[i]; }
    double operator[](int i) const { return x[i]; }
  };
  inline V3 v3(double a, double b, double c) { return V3{{a, b, c}}; }
  inline V3 operator+(const V3 & a, const V3 & b) { ORC_FLOP(3); return v3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
  inline V3 operator-(const V3 & a, const V3 & b) { ORC_FLOP(3); return v3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
  inline V3 operator*(double s, const V3 & a) { ORC_FLOP(3); return v3(s * a[0], s * a[1], s * a[2]); }
  inline V3 cross(const V3 & a, const V3 & b)
  {
    ORC_FLOP(9);
    return v3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
  }
  inline double dot(const V3 & a, const V3 & b) { ORC_FLOP(5); return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

  struct M3
  {
    double m[9];
    double & operator()(int i, int j) { return m[3 * i + j]; }
    double operator()(int i, int j) const { return m[3 * i + j]; }
  };
  inline M3 m3_zero()
  {
    M3 r;
    for (int i = 0; i < 9; i++)
      r.m[i] = 0;
    return r;
  }
  inline M3 m3_id()
  {
    M3 r = m3_zero();
    r(0, 0) = r(1, 1) = r(2, 2) = 1;
    return r;
  }
  inline M3 operator*(const M3 & a, const M3 & b)
  {
    ORC_FLOP(45);
    M3 r = m3_zero();
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++)
          r(i, j) += a(i, k) * b(k, j);
    return r;
  }
  inline M3 operator+(const M3 & a, const M3 & b)
  {
    ORC_FLOP(9);
    M3 r;
    for (int i = 0; i < 9; i++)
      r.m[i] = a.m[i] + b.m[i];
    return r;
  }
  inline M3 operator*(double s, const M3 & a)
  {
    ORC_FLOP(9);
    M3 r;
    for (int i = 0; i < 9; i++)
      r.m[i] = s * a.m[i];
    return r;
  }
  inline V3 operator*(const M3 & a, const V3 & v)
  {
    ORC_FLOP(15);
    return v3(
      a(0, 0) * v[0] + a(0, 1) * v[1] + a(0, 2) * v[2], a(1, 0) * v[0] + a(1, 1) * v[1] + a(1, 2) * v[2],
      a(2, 0) * v[0] + a(2, 1) * v[1] + a(2, 2) * v[2]);
  }
  inline M3 tr(const M3 & a)
  {
    M3 r;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
        r(i, j) = a(j, i);
    return r;
  }
  inline M3 skew(const V3 & v)
  {
    M3 r = m3_zero();
    r(0, 1) = -v[2];
    r(0, 2) = v[1];
    r(1, 0) = v[2];
    r(1, 2) = -v[0];
    r(2, 0) = -v[1];
    r(2, 1) = v[0];
    return r;
  }

  struct SE3
  {
    M3 R;
    V3 p;
  };
  inline SE3 se3_id() { return SE3{m3_id(), v3(0, 0, 0)}; }
  inline SE3 operator*(const SE3 & a, const SE3 & b) { return SE3{a.R * b.R, a.p + a.R * b.p}; }
  inline SE3 inv(const SE3 & a)
  {
    M3 Rt = tr(a.R);
    return SE3{Rt, -1.0 * (Rt * a.p)};
  }

  inline M3 quat_to_R(const double * q) // (x,y,z,w)
  {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    M3 R;
    R(0, 0) = 1 - 2 * (y * y + z * z);
    R(0, 1) = 2 * (x * y - z * w);
    R(0, 2) = 2 * (x * z + y * w);
    R(1, 0) = 2 * (x * y + z * w);
    R(1, 1) = 1 - 2 * (x * x + z * z);
    R(1, 2) = 2 * (y * z - x * w);
    R(2, 0) = 2 * (x * z - y * w);
    R(2, 1) = 2 * (y * z + x * w);
    R(2, 2) = 1 - 2 * (x * x + y * y);
    return R;
  }
  // quaternion product a*b, (x,y,z,w)
  inline void quat_mul(const double * a, const double * b, double * o)
  {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by - ax * bz + ay * bw + az * bx;
    o[2] = aw * bz + ax * by - ay * bx + az * bw;
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
  }
  // ---- scalar coefficient functions (series below 0.05 rad to avoid cancellation) ----
  inline double cfA(double t) { return t < 1e-4 ? 1.0 - t * t / 6.0 : std::sin(t) / t; } // sin t / t
  inline double cfB(double t) // (1 - cos t)/t^2
  {
    const double h = 0.5 * t;
    const double s = h < 1e-4 ? 1.0 - h * h / 6.0 : std::sin(h) / h;
    return 0.5 * s * s;
  }
  inline double cfC(double t) // (t - sin t)/t^3
  {
    const double t2 = t * t;
    if (t < 0.05)
      return 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0 - t2 * t2 * t2 / 362880.0;
    return (t - std::sin(t)) / (t2 * t);
  }
  inline double cfD(double t) // 1/t^2 - (1 + cos t)/(2 t sin t)
  {
    const double t2 = t * t;
    if (t < 0.05)
      return 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0 + t2 * t2 * t2 / 1209600.0;
    return 1.0 / t2 - (1.0 + std::cos(t)) / (2.0 * t * std::sin(t));
  }
  inline double cfQ2(double t) // (t^2 + 2 cos t - 2)/(2 t^4)
  {
    const double t2 = t * t;
    if (t < 0.05)
      return 1.0 / 24.0 - t2 / 720.0 + t2 * t2 / 40320.0;
    return (t2 + 2.0 * std::cos(t) - 2.0) / (2.0 * t2 * t2);
  }
  inline double cfQ3(double t) // (2t - 3 sin t + t cos t)/(2 t^5)
  {
    const double t2 = t * t;
    if (t < 0.05)
      return 1.0 / 120.0 - t2 / 2520.0 + t2 * t2 / 120960.0;
    return (2.0 * t - 3.0 * std::sin(t) + t * std::cos(t)) / (2.0 * t2 * t2 * t);
  }

  // rotation vector -> unit quaternion
  inline void exp3_quat(const V3 & w, double * q)
  {
    const double t = std::sqrt(dot(w, w));
    const double s = 0.5 * cfA(0.5 * t);
    q[0] = s * w[0];
    q[1] = s * w[1];
    q[2] = s * w[2];
    q[3] = std::cos(0.5 * t);
  }
  inline M3 exp3(const V3 & w)
  {
    const double t = std::sqrt(dot(w, w));
    M3 W = skew(w);
    return m3_id() + cfA(t) * W + cfB(t) * (W * W);
  }
  inline V3 log3(const M3 & R)
  {
    // via the quaternion-free formula; valid away from pi (MPC states stay far from it)
    V3 ax = v3(R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1)); // 2 sin(t) * axis
    const double s = 0.5 * std::sqrt(dot(ax, ax));                       // sin t
    const double c = 0.5 * (R(0, 0) + R(1, 1) + R(2, 2) - 1.0);          // cos t
    const double t = std::atan2(s, c);
    // w = t/(2 sin t) * ax
    const double k = t < 1e-4 ? 0.5 * (1.0 + t * t / 6.0) : 0.5 * t / s;
    return k * ax;
  }
  // right Jacobian of SO(3): exp3(w + d) ~ exp3(w) exp3(Jr d)
  inline M3 Jexp3(const V3 & w)
  {
    const double t = std::sqrt(dot(w, w));
    M3 W = skew(w);
    return m3_id() + (-cfB(t)) * W + cfC(t) * (W * W);
  }
  // inverse of the right Jacobian of SO(3)
  inline M3 Jlog3(const V3 & w)
  {
    const double t = std::sqrt(dot(w, w));
    M3 W = skew(w);
    return m3_id() + 0.5 * W + cfD(t) * (W * W);
  }

  // exp6: nu = [v; w] -> SE3
  inline SE3 exp6(const double * nu)
  {
    V3 v = v3(nu[0], nu[1], nu[2]), w = v3(nu[3], nu[4], nu[5]);
    const double t = std::sqrt(dot(w, w));
    M3 W = skew(w);
    M3 V = m3_id() + cfB(t) * W + cfC(t) * (W * W);
    return SE3{exp3(w), V * v};
  }
  inline void log6(const SE3 & M, double * nu)
  {
    V3 w = log3(M.R);
    const double t = std::sqrt(dot(w, w));
    M3 W = skew(w);
    M3 Vinv = m3_id() + (-0.5) * W + cfD(t) * (W * W);
    V3 v = Vinv * M.p;
    nu[0] = v[0];
    nu[1] = v[1];
    nu[2] = v[2];
    nu[3] = w[0];
    nu[4] = w[1];
    nu[5] = w[2];
  }

  // Q block of the SE(3) LEFT Jacobian for xi = [rho; phi] (Barfoot, "State Estimation for
  // Robotics", eq. 7.86b).
  inline M3 se3_Q(const V3 & rho, const V3 & phi)
  {
    const double t = std::sqrt(dot(phi, phi));
    M3 P = skew(rho), F = skew(phi);
    M3 FP = F * P, PF = P * F, FPF = FP * F, FFP = F * FP, PFF = PF * F;
    M3 FPFF = FPF * F, FFPF = F * FPF;
    return 0.5 * P + cfC(t) * (FP + PF + FPF) + cfQ2(t) * (FFP + PFF + (-3.0) * FPF) + cfQ3(t) * (FPFF + FFPF);
  }

  // right Jacobian of SE(3): exp6(nu + d) ~ exp6(nu) exp6(J d), 6x6 row-major.  Jr(nu) = Jl(-nu).
  inline Mat Jexp6(const double * nu)
  {
    V3 v = v3(-nu[0], -nu[1], -nu[2]), w = v3(-nu[3], -nu[4], -nu[5]);
    M3 J3 = Jexp3(v3(nu[3], nu[4], nu[5]));
    M3 Q = se3_Q(v, w);
    Mat J(6, 6);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
      {
        J(i, j) = J3(i, j);
        J(i + 3, j + 3) = J3(i, j);
        J(i, j + 3) = Q(i, j);
      }
    return J;
  }
  // Jlog6(M) = Jexp6(log6 M)^-1  (block-triangular inverse)
  inline Mat Jlog6(const SE3 & M)
  {
    double nu[6];
    log6(M, nu);
    V3 v = v3(-nu[0], -nu[1], -nu[2]), w = v3(-nu[3], -nu[4], -nu[5]);
    M3 Ji = Jlog3(v3(nu[3], nu[4], nu[5]));
    M3 Q = se3_Q(v, w);
    M3 X = (-1.0) * (Ji * Q * Ji);
    Mat J(6, 6);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
      {
        J(i, j) = Ji(i, j);
        J(i + 3, j + 3) = Ji(i, j);
        J(i, j + 3) = X(i, j);
      }
    return J;
  }
  // action matrix of M on motions [v;w]:  [[R, [p]x R],[0, R]]
  inline Mat action_matrix(const SE3 & M)
  {
    Mat A(6, 6);
    M3 pR = skew(M.p) * M.R;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++)
      {
        A(i, j) = M.R(i, j);
        A(i + 3, j + 3) = M.R(i, j);
        A(i, j + 3) = pR(i, j);
      }
    return A;
  }
} // namespace orc
