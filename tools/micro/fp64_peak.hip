// Micro-benchmark: measured FP64 peak of the device this runs on ("verify on the box with a micro-benchmark and
// quote the measured peak").  Two dependency-free loops, all CUs, >= 1 s each:
//   (a) v_fma_f64            8 independent accumulator chains per lane, 8 waves per SIMD-quad (32 per CU)
//   (b) v_mfma_f64_16x16x4   4 independent accumulator tiles per wave
// Output: one JSON line {"fma_tflops": .., "mfma_tflops": .., "sclk_mhz": .., "cus": ..}.
// Build: hipcc --offload-arch=gfx950 -O3 fp64_peak.hip -o fp64_peak.bin      (bench.py runs the binary when present)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int INNER = 4096;

__global__ __launch_bounds__(256) void k_fma(double * out, int outer)
{
  double a0 = threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double x = 1.0 + 1e-12 * blockIdx.x, y = 1e-13;
  for (int o = 0; o < outer; o++)
  {
#pragma unroll 16
    for (int i = 0; i < INNER; i++)
    {
      a0 = __builtin_fma(a0, x, y);
      a1 = __builtin_fma(a1, x, y);
      a2 = __builtin_fma(a2, x, y);
      a3 = __builtin_fma(a3, x, y);
      a4 = __builtin_fma(a4, x, y);
      a5 = __builtin_fma(a5, x, y);
      a6 = __builtin_fma(a6, x, y);
      a7 = __builtin_fma(a7, x, y);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ __launch_bounds__(256) void k_mfma(double * out, int outer)
{
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-9 * blockIdx.x;
  for (int o = 0; o < outer; o++)
  {
#pragma unroll 8
    for (int i = 0; i < INNER / 4; i++)
    {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
  }
  const d4 s = c0 + c1 + c2 + c3;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

#define CK(x)                                                                                                          \
  do                                                                                                                   \
  {                                                                                                                    \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess)                                                                                              \
    {                                                                                                                  \
      std::fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);                              \
      return 1;                                                                                                        \
    }                                                                                                                  \
  } while (0)

template <class K>
static int run(K kernel, int blocks, double flop_per_thread_per_outer, double * out, double target_s, double * tflops)
{
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  int outer = 8;
  float ms = 0;
  for (int pass = 0; pass < 6; pass++) // grow the launch until it runs for target_s
  {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, outer);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms >= target_s * 1e3)
      break;
    const double scale = target_s * 1.1e3 / (ms > 1e-3 ? ms : 1e-3);
    outer = (int)(outer * (scale > 64 ? 64 : scale)) + 1;
  }
  *tflops = flop_per_thread_per_outer * outer * (double)blocks * 256.0 / (ms * 1e-3) / 1e12;
  return 0;
}

int main(int argc, char ** argv)
{
  const double target_s = argc > 1 ? std::atof(argv[1]) : 1.0;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int blocks = cus * 8; // 8 blocks x 4 waves per CU = 8 waves per SIMD
  double * out;
  CK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(double)));
  double fma = 0, mfma = 0;
  if (run(k_fma, blocks, 8.0 * INNER * 2.0, out, target_s, &fma))
    return 1;
  // one MFMA 16x16x4 = 1024 FMA = 2048 flop per wave = 32 flop per thread
  if (run(k_mfma, blocks, (double)INNER * 32.0, out, target_s, &mfma))
    return 1;
  std::printf("{\"fma_tflops\": %.2f, \"mfma_tflops\": %.2f, \"sclk_mhz\": %d, \"cus\": %d, \"seconds_each\": %.1f}\n", fma, mfma,
              p.clockRate / 1000, cus, target_s);
  return 0;
}
