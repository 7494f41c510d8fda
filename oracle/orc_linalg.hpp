// ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call
// anything under oracle/.  PARITY UNPINNED: the reference's arithmetic for this path lives in
// Aligator 0.16.0 / Pinocchio 3.8.0 / Eigen 3.4.0 (pixi.lock:11,154,38), none of which is
// present here; this is a restatement of their published algorithms, validated by
// self-consistency tests (finite differences, dense KKT solves), not against upstream outputs.
//
// orc_linalg.hpp: minimal run-time-sized dense matrix helpers (stand-in for Eigen::MatrixXd).
#pragma once
#include <cassert>
#include <cmath>
#include <cstring>
#include <vector>

// FLOP instrumentation ("derivative FLOPs counted by instrumentation in the oracle"): compiled in only with
// -DORC_COUNT_FLOPS (tools/count_flops.py builds that variant as oracle/_flops/liborc_flops.so); an FMA counts 2.
#ifdef ORC_COUNT_FLOPS
namespace orc
{
  inline thread_local double flop_counter = 0.0;
}
#define ORC_FLOP(n) (::orc::flop_counter += (double)(n))
#else
#define ORC_FLOP(n) ((void)0)
#endif

namespace orc
{
  typedef std::vector<double> Vec;

  // Storage of the matrices: sizes are run-time (the same code serves every OCP), but the buffers come from per-thread free lists by
  // size class instead of malloc / free -- a stage evaluation builds and drops hundreds of small matrices (the CPU baseline of bench.py
  // is this code: it should not be timing the heap).  Measured on 8 cores, 16 instances, k = 3: 162 -> 165 control-steps/s, i.e. the
  // allocator was never what bounds this code; what does is that sizes are run-time, so nothing is unrolled or vectorised per model)
  template <class T>
  struct PoolAlloc
  {
    typedef T value_type;
    PoolAlloc() noexcept {}
    template <class U>
    PoolAlloc(const PoolAlloc<U> &) noexcept {}
    static constexpr int NCLASS = 20; // 2^3 .. 2^22 elements
    struct Lists
    {
      std::vector<void *> free_[NCLASS];
      ~Lists()
      {
        for (auto & l : free_)
          for (void * p : l)
            ::operator delete(p);
      }
    };
    static Lists & lists()
    {
      static thread_local Lists l;
      return l;
    }
    static int size_class(std::size_t n)
    {
      int c = 0;
      while (((std::size_t)8 << c) < n)
        c++;
      return c;
    }
    T * allocate(std::size_t n)
    {
      const int c = size_class(n);
      if (c >= NCLASS)
        return static_cast<T *>(::operator new(n * sizeof(T)));
      auto & l = lists().free_[c];
      if (!l.empty())
      {
        void * p = l.back();
        l.pop_back();
        return static_cast<T *>(p);
      }
      return static_cast<T *>(::operator new(((std::size_t)8 << c) * sizeof(T)));
    }
    void deallocate(T * p, std::size_t n) noexcept
    {
      const int c = size_class(n);
      if (c >= NCLASS)
        ::operator delete(p);
      else
        lists().free_[c].push_back(p);
    }
    template <class U>
    bool operator==(const PoolAlloc<U> &) const noexcept { return true; }
    template <class U>
    bool operator!=(const PoolAlloc<U> &) const noexcept { return false; }
  };

  struct Mat
  {
    int r = 0, c = 0;
    std::vector<double, PoolAlloc<double>> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    void resize(int r_, int c_)
    {
      r = r_;
      c = c_;
      a.assign((size_t)r_ * c_, 0.0);
    }
    void zero() { std::fill(a.begin(), a.end(), 0.0); }
    double & operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
    static Mat identity(int n)
    {
      Mat m(n, n);
      for (int i = 0; i < n; i++)
        m(i, i) = 1.0;
      return m;
    }
  };

  inline Mat transpose(const Mat & A)
  {
    Mat T(A.c, A.r);
    for (int i = 0; i < A.r; i++)
      for (int j = 0; j < A.c; j++)
        T(j, i) = A(i, j);
    return T;
  }

  // C = A * B
  inline Mat mul(const Mat & A, const Mat & B)
  {
    assert(A.c == B.r);
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; i++)
      for (int k = 0; k < A.c; k++)
      {
        const double aik = A(i, k);
        if (aik == 0.0)
          continue;
        ORC_FLOP(2 * B.c);
        for (int j = 0; j < B.c; j++)
          C(i, j) += aik * B(k, j);
      }
    return C;
  }
  // C = A^T * B
  inline Mat mulTN(const Mat & A, const Mat & B)
  {
    assert(A.r == B.r);
    Mat C(A.c, B.c);
    for (int k = 0; k < A.r; k++)
      for (int i = 0; i < A.c; i++)
      {
        const double aki = A(k, i);
        if (aki == 0.0)
          continue;
        ORC_FLOP(2 * B.c);
        for (int j = 0; j < B.c; j++)
          C(i, j) += aki * B(k, j);
      }
    return C;
  }
  inline Vec mul(const Mat & A, const Vec & x)
  {
    assert(A.c == (int)x.size());
    ORC_FLOP(2 * A.r * A.c);
    Vec y(A.r, 0.0);
    for (int i = 0; i < A.r; i++)
    {
      double s = 0;
      for (int j = 0; j < A.c; j++)
        s += A(i, j) * x[j];
      y[i] = s;
    }
    return y;
  }
  inline Vec mulT(const Mat & A, const Vec & x)
  {
    assert(A.r == (int)x.size());
    ORC_FLOP(2 * A.r * A.c);
    Vec y(A.c, 0.0);
    for (int i = 0; i < A.r; i++)
      for (int j = 0; j < A.c; j++)
        y[j] += A(i, j) * x[i];
    return y;
  }
  inline void add_inplace(Mat & A, const Mat & B, double s = 1.0)
  {
    assert(A.r == B.r && A.c == B.c);
    ORC_FLOP(2 * A.a.size());
    for (size_t i = 0; i < A.a.size(); i++)
      A.a[i] += s * B.a[i];
  }
  inline void axpy(Vec & y, const Vec & x, double s = 1.0)
  {
    assert(x.size() == y.size());
    ORC_FLOP(2 * y.size());
    for (size_t i = 0; i < y.size(); i++)
      y[i] += s * x[i];
  }
  inline double dot(const Vec & a, const Vec & b)
  {
    assert(a.size() == b.size());
    ORC_FLOP(2 * a.size());
    double s = 0;
    for (size_t i = 0; i < a.size(); i++)
      s += a[i] * b[i];
    return s;
  }
  inline double norm_inf(const Vec & a)
  {
    double s = 0;
    for (double v : a)
      s = std::fmax(s, std::fabs(v));
    return s;
  }
  inline double sqnorm(const Vec & a) { return dot(a, a); }

  // In-place lower Cholesky A = L L^T (upper part left untouched). Returns false if not SPD.
  inline bool cholesky(Mat & A)
  {
    const int n = A.r;
    ORC_FLOP((double)n * n * n / 3.0);
    for (int j = 0; j < n; j++)
    {
      double d = A(j, j);
      for (int k = 0; k < j; k++)
        d -= A(j, k) * A(j, k);
      if (!(d > 0.0))
        return false;
      d = std::sqrt(d);
      A(j, j) = d;
      for (int i = j + 1; i < n; i++)
      {
        double s = A(i, j);
        for (int k = 0; k < j; k++)
          s -= A(i, k) * A(j, k);
        A(i, j) = s / d;
      }
    }
    return true;
  }
  // Solve L y = b (in place), L lower from cholesky()
  inline void solve_L(const Mat & L, double * b, int stride = 1)
  {
    const int n = L.r;
    ORC_FLOP(n * n);
    for (int i = 0; i < n; i++)
    {
      double s = b[i * stride];
      for (int k = 0; k < i; k++)
        s -= L(i, k) * b[k * stride];
      b[i * stride] = s / L(i, i);
    }
  }
  // Solve L^T y = b (in place)
  inline void solve_LT(const Mat & L, double * b, int stride = 1)
  {
    const int n = L.r;
    ORC_FLOP(n * n);
    for (int i = n - 1; i >= 0; i--)
    {
      double s = b[i * stride];
      for (int k = i + 1; k < n; k++)
        s -= L(k, i) * b[k * stride];
      b[i * stride] = s / L(i, i);
    }
  }
  // X = (L L^T)^-1 B, column by column, in place on B
  inline void chol_solve_inplace(const Mat & L, Mat & B)
  {
    for (int j = 0; j < B.c; j++)
    {
      solve_L(L, &B.a[j], B.c);
      solve_LT(L, &B.a[j], B.c);
    }
  }
  inline void chol_solve_inplace(const Mat & L, Vec & b)
  {
    solve_L(L, b.data());
    solve_LT(L, b.data());
  }

  // General small inverse by Gauss-Jordan with partial pivoting (used for 6x6 blocks).
  inline Mat inverse(const Mat & A)
  {
    const int n = A.r;
    ORC_FLOP(2.0 * n * n * n);
    Mat M = A, I = Mat::identity(n);
    for (int col = 0; col < n; col++)
    {
      int piv = col;
      for (int i = col + 1; i < n; i++)
        if (std::fabs(M(i, col)) > std::fabs(M(piv, col)))
          piv = i;
      if (piv != col)
        for (int j = 0; j < n; j++)
        {
          std::swap(M(piv, j), M(col, j));
          std::swap(I(piv, j), I(col, j));
        }
      const double d = 1.0 / M(col, col);
      for (int j = 0; j < n; j++)
      {
        M(col, j) *= d;
        I(col, j) *= d;
      }
      for (int i = 0; i < n; i++)
        if (i != col)
        {
          const double f = M(i, col);
          if (f == 0.0)
            continue;
          for (int j = 0; j < n; j++)
          {
            M(i, j) -= f * M(col, j);
            I(i, j) -= f * I(col, j);
          }
        }
    }
    return I;
  }
} // namespace orc
