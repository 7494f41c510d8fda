// accuracy of v_rcp_f64 and of 0 / 1 / 2 Newton refinements against the IEEE quotient (tools/micro: measurements behind
// design decisions).  hipcc --offload-arch=gfx950 -O3 rcp_accuracy.hip -o rcp_accuracy && ./rcp_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const double * x, double * r0, double * r1, double * r2, int n)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  const double a = x[i];
  double r = __builtin_amdgcn_rcp(a);
  r0[i] = r;
  r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
  r1[i] = r;
  r = __builtin_fma(__builtin_fma(-a, r, 1.0), r, r);
  r2[i] = r;
}
int main()
{
  const int n = 1 << 20;
  std::vector<double> x(n), r0(n), r1(n), r2(n);
  srand(1);
  for (int i = 0; i < n; i++)
    x[i] = std::ldexp(1.0 + (double)rand() / RAND_MAX, rand() % 80 - 40) * ((rand() & 1) ? 1 : -1);
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8);
  hipMalloc(&d0, n * 8);
  hipMalloc(&d1, n * 8);
  hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++)
  {
    const double t = 1.0 / x[i];
    e0 = std::fmax(e0, std::fabs(r0[i] - t) / std::fabs(t));
    e1 = std::fmax(e1, std::fabs(r1[i] - t) / std::fabs(t));
    e2 = std::fmax(e2, std::fabs(r2[i] - t) / std::fabs(t));
  }
  std::printf("max relative error: v_rcp_f64 %.3e | +1 Newton %.3e | +2 Newton %.3e  (eps = %.3e)\n", e0, e1, e2, 2.22e-16);
  return 0;
}
