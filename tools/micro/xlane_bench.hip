// Micro-benchmark: cost per element of the inner step "b[i] -= L * x" when L comes from (a) v_readlane pairs,
// (b) a broadcast LDS read, (c) registers only; plus f64 MFMA 16x16x4 issue rate.  One wave per block,
// `waves` blocks per CU worth of grid.  Build: hipcc --offload-arch=gfx950 -O3 xlane_bench.hip -o xlane_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ double rl(double v, int src)
{
  union { double d; int i[2]; } u; u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], src);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], src);
  return u.d;
}

constexpr int N = 32, REP = 200;

__global__ void k_readlane(double * out, long long * cyc)
{
  double b[N], l = threadIdx.x * 1e-3 + 1.0, x = 1e-3 * blockIdx.x;
  for (int i = 0; i < N; i++) b[i] = i + threadIdx.x;
  long long t0 = clock64();
  for (int r = 0; r < REP; r++)
  {
#pragma unroll
    for (int i = 0; i < N; i++)
      b[i] -= rl(l, i) * x;
    l += b[0] * 1e-30;
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < N; i++) s += b[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_lds(double * out, long long * cyc)
{
  __shared__ double L[64];
  L[threadIdx.x] = threadIdx.x * 1e-3 + 1.0;
  __syncthreads();
  double b[N], x = 1e-3 * blockIdx.x;
  for (int i = 0; i < N; i++) b[i] = i + threadIdx.x;
  long long t0 = clock64();
  for (int r = 0; r < REP; r++)
  {
#pragma unroll
    for (int i = 0; i < N; i++)
      b[i] -= ((volatile double *)L)[i] * x;
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < N; i++) s += b[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_lds128(double * out, long long * cyc)
{
  typedef double dv2 __attribute__((ext_vector_type(2)));
  __shared__ dv2 L[32];
  if (threadIdx.x < 32) L[threadIdx.x] = dv2{threadIdx.x * 1e-3 + 1.0, 0.5};
  __syncthreads();
  double b[N], x = 1e-3 * blockIdx.x;
  for (int i = 0; i < N; i++) b[i] = i + threadIdx.x;
  long long t0 = clock64();
  for (int r = 0; r < REP; r++)
  {
#pragma unroll
    for (int i = 0; i < N; i += 2)
    {
      dv2 v = L[i / 2]; asm volatile("" : "+v"(v));
      b[i] -= v.x * x;
      b[i + 1] -= v.y * x;
    }
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < N; i++) s += b[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_reg(double * out, long long * cyc)
{
  double b[N], l = threadIdx.x * 1e-3 + 1.0, x = 1e-3 * blockIdx.x;
  for (int i = 0; i < N; i++) b[i] = i + threadIdx.x;
  long long t0 = clock64();
  for (int r = 0; r < REP; r++)
  {
#pragma unroll
    for (int i = 0; i < N; i++)
      b[i] -= l * x;
    l += b[0] * 1e-30;
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < N; i++) s += b[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma(double * out, long long * cyc)
{
  d4 acc[4];
  for (int i = 0; i < 4; i++) acc[i] = d4{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  long long t0 = clock64();
  for (int r = 0; r < REP; r++)
  {
#pragma unroll
    for (int i = 0; i < N; i++)
      acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i & 3], 0, 0, 0);
  }
  long long t1 = clock64();
  double s = 0; for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_mfma4(double * out, long long * cyc)
{
  double acc[4] = {0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  long long t0 = clock64();
  for (int r = 0; r < REP; r++)
  {
#pragma unroll
    for (int i = 0; i < N; i++)
      acc[i & 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i & 3], 0, 0, 0);
  }
  long long t1 = clock64();
  out[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
void run(const char * name, K kern, int blocks)
{
  double * out; long long * cyc;
  hipMalloc(&out, sizeof(double) * 64 * blocks);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  printf("%-10s blocks %5d: %.1f clock64-ticks/elem (wave), kernel %.3f ms -> %.2f ns/elem/wave\n", name, blocks,
         avg / (double)(N * REP), ms, ms * 1e6 / (double)(N * REP));
  hipFree(out); hipFree(cyc);
}
int main()
{
  for (int blocks : {256, 1024, 2048})
  {
    run("readlane", k_readlane, blocks);
    run("lds_b64", k_lds, blocks);
    run("lds_b128", k_lds128, blocks);
    run("reg", k_reg, blocks);
    run("mfma16x4", k_mfma, blocks);
    run("mfma4x4x4", k_mfma4, blocks);
  }
  return 0;
}
