// smpc_math.h -- small fixed-size FP64 math used inside the kernels: 3-vectors, 3x3 matrices,
// SO(3)/SE(3) exp/log and their Jacobians, spatial (Plucker) vectors and inertias.
//
// Conventions follow Pinocchio (the library the reference's states are defined by; reference
// src/robot-handler.cpp:81-96, src/kinodynamics.cpp:46): motions/forces are [linear; angular],
// free-flyer q = [p; quat(x,y,z,w)], integrate = M*exp6(nu), difference = log6(M0^-1 M1).
// Everything is register-resident scalar code (no local arrays with dynamic indexing).
#pragma once
#include <smpc_backend.h>
#include <math.h>

namespace smpc
{
  struct V3
  {
    double x, y, z;
  };
  SMPC_HD V3 mk3(double x, double y, double z) { return V3{x, y, z}; }
  SMPC_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
  SMPC_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
  SMPC_HD V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
  SMPC_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
  SMPC_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
  SMPC_HD V3 ld3(const double * p) { return V3{p[0], p[1], p[2]}; }
  SMPC_HD void st3(double * p, V3 v)
  {
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
  }

  struct M3 // row-major
  {
    double a00, a01, a02, a10, a11, a12, a20, a21, a22;
  };
  SMPC_HD M3 m3_id() { return M3{1, 0, 0, 0, 1, 0, 0, 0, 1}; }
  SMPC_HD M3 ldm3(const double * p) { return M3{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}; }
  SMPC_HD void stm3(double * p, const M3 & m)
  {
    p[0] = m.a00;
    p[1] = m.a01;
    p[2] = m.a02;
    p[3] = m.a10;
    p[4] = m.a11;
    p[5] = m.a12;
    p[6] = m.a20;
    p[7] = m.a21;
    p[8] = m.a22;
  }
  SMPC_HD V3 operator*(const M3 & m, V3 v)
  {
    return V3{m.a00 * v.x + m.a01 * v.y + m.a02 * v.z, m.a10 * v.x + m.a11 * v.y + m.a12 * v.z,
              m.a20 * v.x + m.a21 * v.y + m.a22 * v.z};
  }
  // m^T v
  SMPC_HD V3 tmul(const M3 & m, V3 v)
  {
    return V3{m.a00 * v.x + m.a10 * v.y + m.a20 * v.z, m.a01 * v.x + m.a11 * v.y + m.a21 * v.z,
              m.a02 * v.x + m.a12 * v.y + m.a22 * v.z};
  }
  SMPC_HD M3 operator*(const M3 & a, const M3 & b)
  {
    return M3{a.a00 * b.a00 + a.a01 * b.a10 + a.a02 * b.a20, a.a00 * b.a01 + a.a01 * b.a11 + a.a02 * b.a21,
              a.a00 * b.a02 + a.a01 * b.a12 + a.a02 * b.a22, a.a10 * b.a00 + a.a11 * b.a10 + a.a12 * b.a20,
              a.a10 * b.a01 + a.a11 * b.a11 + a.a12 * b.a21, a.a10 * b.a02 + a.a11 * b.a12 + a.a12 * b.a22,
              a.a20 * b.a00 + a.a21 * b.a10 + a.a22 * b.a20, a.a20 * b.a01 + a.a21 * b.a11 + a.a22 * b.a21,
              a.a20 * b.a02 + a.a21 * b.a12 + a.a22 * b.a22};
  }
  SMPC_HD M3 operator+(const M3 & a, const M3 & b)
  {
    return M3{a.a00 + b.a00, a.a01 + b.a01, a.a02 + b.a02, a.a10 + b.a10, a.a11 + b.a11,
              a.a12 + b.a12, a.a20 + b.a20, a.a21 + b.a21, a.a22 + b.a22};
  }
  SMPC_HD M3 operator*(double s, const M3 & a)
  {
    return M3{s * a.a00, s * a.a01, s * a.a02, s * a.a10, s * a.a11, s * a.a12, s * a.a20, s * a.a21, s * a.a22};
  }
  SMPC_HD M3 transpose(const M3 & a) { return M3{a.a00, a.a10, a.a20, a.a01, a.a11, a.a21, a.a02, a.a12, a.a22}; }
  SMPC_HD M3 skew(V3 v) { return M3{0, -v.z, v.y, v.z, 0, -v.x, -v.y, v.x, 0}; }

  // ---- scalar coefficient functions; series below 0.05 rad (cancellation-free) ----
  SMPC_HD double cf_sinc(double t) { return t < 1e-4 ? 1.0 - t * t / 6.0 : sin(t) / t; }
  SMPC_HD double cf_B(double t) // (1 - cos t)/t^2
  {
    const double h = 0.5 * t;
    const double s = cf_sinc(h);
    return 0.5 * s * s;
  }
  SMPC_HD double cf_C(double t) // (t - sin t)/t^3
  {
    const double t2 = t * t;
    if (t < 0.05)
      return 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0 - t2 * t2 * t2 / 362880.0;
    return (t - sin(t)) / (t2 * t);
  }
  SMPC_HD double cf_D(double t) // 1/t^2 - (1 + cos t)/(2 t sin t)
  {
    const double t2 = t * t;
    if (t < 0.05)
      return 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0 + t2 * t2 * t2 / 1209600.0;
    return 1.0 / t2 - (1.0 + cos(t)) / (2.0 * t * sin(t));
  }
  SMPC_HD double cf_Q2(double t) // (t^2 + 2 cos t - 2)/(2 t^4)
  {
    const double t2 = t * t;
    if (t < 0.05)
      return 1.0 / 24.0 - t2 / 720.0 + t2 * t2 / 40320.0;
    return (t2 + 2.0 * cos(t) - 2.0) / (2.0 * t2 * t2);
  }
  SMPC_HD double cf_Q3(double t) // (2t - 3 sin t + t cos t)/(2 t^5)
  {
    const double t2 = t * t;
    if (t < 0.05)
      return 1.0 / 120.0 - t2 / 2520.0 + t2 * t2 / 120960.0;
    return (2.0 * t - 3.0 * sin(t) + t * cos(t)) / (2.0 * t2 * t2 * t);
  }

  struct Quat
  {
    double x, y, z, w;
  };
  SMPC_HD M3 quat_to_R(Quat q)
  {
    const double x = q.x, y = q.y, z = q.z, w = q.w;
    return M3{1 - 2 * (y * y + z * z), 2 * (x * y - z * w),     2 * (x * z + y * w),
              2 * (x * y + z * w),     1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
              2 * (x * z - y * w),     2 * (y * z + x * w),     1 - 2 * (x * x + y * y)};
  }
  SMPC_HD Quat quat_mul(Quat a, Quat b)
  {
    return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
  }
  SMPC_HD Quat quat_exp(V3 w)
  {
    const double t = sqrt(dot(w, w));
    const double s = 0.5 * cf_sinc(0.5 * t);
    return Quat{s * w.x, s * w.y, s * w.z, cos(0.5 * t)};
  }
  SMPC_HD M3 exp3(V3 w)
  {
    const double t = sqrt(dot(w, w));
    const M3 W = skew(w);
    return m3_id() + cf_sinc(t) * W + cf_B(t) * (W * W);
  }
  SMPC_HD V3 log3(const M3 & R)
  {
    const V3 ax = mk3(R.a21 - R.a12, R.a02 - R.a20, R.a10 - R.a01);
    const double s = 0.5 * sqrt(dot(ax, ax));
    const double c = 0.5 * (R.a00 + R.a11 + R.a22 - 1.0);
    const double t = atan2(s, c);
    const double k = t < 1e-4 ? 0.5 * (1.0 + t * t / 6.0) : 0.5 * t / s;
    return k * ax;
  }
  // right Jacobian of SO(3) and its inverse
  SMPC_HD M3 Jexp3(V3 w)
  {
    const double t = sqrt(dot(w, w));
    const M3 W = skew(w);
    return m3_id() + (-cf_B(t)) * W + cf_C(t) * (W * W);
  }
  SMPC_HD M3 Jlog3(V3 w)
  {
    const double t = sqrt(dot(w, w));
    const M3 W = skew(w);
    return m3_id() + 0.5 * W + cf_D(t) * (W * W);
  }

  struct SE3
  {
    M3 R;
    V3 p;
  };
  SMPC_HD SE3 se3_mul(const SE3 & a, const SE3 & b) { return SE3{a.R * b.R, a.p + a.R * b.p}; }
  SMPC_HD SE3 se3_inv(const SE3 & a)
  {
    const M3 Rt = transpose(a.R);
    return SE3{Rt, -1.0 * (Rt * a.p)};
  }
  SMPC_HD SE3 exp6(V3 v, V3 w)
  {
    const double t = sqrt(dot(w, w));
    const M3 W = skew(w);
    const M3 V = m3_id() + cf_B(t) * W + cf_C(t) * (W * W);
    return SE3{exp3(w), V * v};
  }
  SMPC_HD void log6(const SE3 & M, V3 & v, V3 & w)
  {
    w = log3(M.R);
    const double t = sqrt(dot(w, w));
    const M3 W = skew(w);
    const M3 Vinv = m3_id() + (-0.5) * W + cf_D(t) * (W * W);
    v = Vinv * M.p;
  }
  // Q block of the SE(3) left Jacobian for xi = [rho; phi] (Barfoot 2017, eq. 7.86b)
  SMPC_HD M3 se3_Q(V3 rho, V3 phi)
  {
    const double t = sqrt(dot(phi, phi));
    const M3 P = skew(rho), F = skew(phi);
    const M3 FP = F * P, PF = P * F;
    const M3 FPF = FP * F;
    const M3 FFP = F * FP, PFF = PF * F;
    const M3 FPFF = FPF * F, FFPF = F * FPF;
    return 0.5 * P + cf_C(t) * (FP + PF + FPF) + cf_Q2(t) * (FFP + PFF + (-3.0) * FPF) + cf_Q3(t) * (FPFF + FFPF);
  }
  // all coefficient functions of one angle from a single sincos(t/2) (sin t = 2 s c, cos t = 1 - 2 s^2)
  struct SE3Coef
  {
    double sinc, sinch, ch, B, C, D, Q2, Q3; // sin t / t, sin(t/2)/(t/2), cos(t/2), cf_B .. cf_Q3
  };
  SMPC_HD SE3Coef se3_coef(double t)
  {
    SE3Coef k;
    const double h = 0.5 * t, t2 = t * t;
    double sh, ch;
    sincos(h, &sh, &ch);
    const double s1 = 2.0 * sh * ch, c1 = 1.0 - 2.0 * sh * sh;
    k.ch = ch;
    k.sinch = h < 1e-4 ? 1.0 - h * h / 6.0 : sh / h;
    k.sinc = t < 1e-4 ? 1.0 - t2 / 6.0 : s1 / t;
    k.B = 0.5 * k.sinch * k.sinch;
    if (t < 0.05)
    {
      k.C = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0 - t2 * t2 * t2 / 362880.0;
      k.D = 1.0 / 12.0 + t2 / 720.0 + t2 * t2 / 30240.0 + t2 * t2 * t2 / 1209600.0;
      k.Q2 = 1.0 / 24.0 - t2 / 720.0 + t2 * t2 / 40320.0;
      k.Q3 = 1.0 / 120.0 - t2 / 2520.0 + t2 * t2 / 120960.0;
    }
    else
    {
      k.C = (t - s1) / (t2 * t);
      k.D = 1.0 / t2 - (1.0 + c1) / (2.0 * t * s1);
      k.Q2 = (t2 + 2.0 * c1 - 2.0) / (2.0 * t2 * t2);
      k.Q3 = (2.0 * t - 3.0 * s1 + t * c1) / (2.0 * t2 * t2 * t);
    }
    return k;
  }
  // se3_Q with precomputed coefficients
  SMPC_HD M3 se3_Q(V3 rho, V3 phi, const SE3Coef & k)
  {
    const M3 P = skew(rho), F = skew(phi);
    const M3 FP = F * P, PF = P * F;
    const M3 FPF = FP * F;
    const M3 FFP = F * FP, PFF = PF * F;
    const M3 FPFF = FPF * F, FFPF = F * FPF;
    return 0.5 * P + k.C * (FP + PF + FPF) + k.Q2 * (FFP + PFF + (-3.0) * FPF) + k.Q3 * (FPFF + FFPF);
  }
  // right Jacobian of SE(3) at nu=[v;w]: J = [[J3, Q],[0, J3]]
  SMPC_HD void Jexp6(V3 v, V3 w, M3 & J3, M3 & Q)
  {
    J3 = Jexp3(w);
    Q = se3_Q(-1.0 * v, -1.0 * w);
  }
  // Jlog6 at nu = log6(M): [[Ji, X],[0, Ji]]
  SMPC_HD void Jlog6(V3 v, V3 w, M3 & Ji, M3 & X)
  {
    Ji = Jlog3(w);
    const M3 Q = se3_Q(-1.0 * v, -1.0 * w);
    X = (-1.0) * (Ji * Q * Ji);
  }

  // ---- spatial algebra ----
  struct SV
  {
    V3 l, a;
  };
  SMPC_HD double v3c(V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
  SMPC_HD SV sv0() { return SV{mk3(0, 0, 0), mk3(0, 0, 0)}; }
  SMPC_HD SV operator+(const SV & x, const SV & y) { return SV{x.l + y.l, x.a + y.a}; }
  SMPC_HD SV operator-(const SV & x, const SV & y) { return SV{x.l - y.l, x.a - y.a}; }
  SMPC_HD SV operator*(double s, const SV & x) { return SV{s * x.l, s * x.a}; }
  SMPC_HD SV crm(const SV & v, const SV & m) { return SV{cross(v.a, m.l) + cross(v.l, m.a), cross(v.a, m.a)}; }
  SMPC_HD SV crf(const SV & v, const SV & f) { return SV{cross(v.a, f.l), cross(v.a, f.a) + cross(v.l, f.l)}; }
  SMPC_HD SV ldsv(const double * p) { return SV{ld3(p), ld3(p + 3)}; }
  SMPC_HD void stsv(double * p, const SV & s)
  {
    st3(p, s.l);
    st3(p + 3, s.a);
  }
  // spatial inertia about the world origin stored as 10 doubles: m, mc(3), J(xx,xy,xz,yy,yz,zz)
  struct SI
  {
    double m;
    V3 mc;
    double jxx, jxy, jxz, jyy, jyz, jzz;
  };
  SMPC_HD SI ldsi(const double * p) { return SI{p[0], mk3(p[1], p[2], p[3]), p[4], p[5], p[6], p[7], p[8], p[9]}; }
  SMPC_HD void stsi(double * p, const SI & I)
  {
    p[0] = I.m;
    p[1] = I.mc.x;
    p[2] = I.mc.y;
    p[3] = I.mc.z;
    p[4] = I.jxx;
    p[5] = I.jxy;
    p[6] = I.jxz;
    p[7] = I.jyy;
    p[8] = I.jyz;
    p[9] = I.jzz;
  }
  SMPC_HD V3 si_J(const SI & I, V3 w)
  {
    return V3{I.jxx * w.x + I.jxy * w.y + I.jxz * w.z, I.jxy * w.x + I.jyy * w.y + I.jyz * w.z,
              I.jxz * w.x + I.jyz * w.y + I.jzz * w.z};
  }
  SMPC_HD SV operator*(const SI & I, const SV & v)
  {
    return SV{I.m * v.l + cross(v.a, I.mc), si_J(I, v.a) + cross(I.mc, v.l)};
  }
  SMPC_HD SI operator+(const SI & a, const SI & b)
  {
    return SI{a.m + b.m,     a.mc + b.mc,   a.jxx + b.jxx, a.jxy + b.jxy,
              a.jxz + b.jxz, a.jyy + b.jyy, a.jyz + b.jyz, a.jzz + b.jzz};
  }
} // namespace smpc
