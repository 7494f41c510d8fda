// Micro-benchmark: do instructions of DIFFERENT waves of one SIMD overlap on MI355X?  One workgroup of 8 waves per CU (2 waves per SIMD).
// Each wave runs a dependency-free loop of one kind -- v_mfma_f64_16x16x4 (4 accumulator tiles), v_fma_f64 (8 chains), v_add_u32 / v_xor_b32
// (8 chains), ds_read_b64 (+ the adds that consume the loads) -- and the kernel is timed with the two waves of every SIMD running (a) the same kind, (b) two different kinds.
// If the pair (X, Y) takes max(t_X, t_Y) the two pipes overlap; if it takes t_X + t_Y they share an issue port / a datapath.
// Output: one JSON line of milliseconds.  Build: hipcc --offload-arch=gfx950 -O3 issue_overlap.hip -o issue_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int N = 1 << 16;

__device__ __forceinline__ double run_mfma(int n, double a, double b)
{
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
#pragma unroll 4
  for (int i = 0; i < n / 4; i++)
  {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  const d4 s = c0 + c1 + c2 + c3;
  return s[0] + s[1] + s[2] + s[3];
}
__device__ __forceinline__ double run_fma(int n, double x, double y)
{
  double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
#pragma unroll 4
  for (int i = 0; i < n; i++)
  {
    a0 = __builtin_fma(a0, x, y); a1 = __builtin_fma(a1, x, y); a2 = __builtin_fma(a2, x, y); a3 = __builtin_fma(a3, x, y);
    a4 = __builtin_fma(a4, x, y); a5 = __builtin_fma(a5, x, y); a6 = __builtin_fma(a6, x, y); a7 = __builtin_fma(a7, x, y);
  }
  return a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__device__ __forceinline__ double run_int(int n, unsigned x)
{
  unsigned a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
#pragma unroll 4
  for (int i = 0; i < n; i++)
  {
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a0) : "v"(x)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a1) : "v"(x));
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a2) : "v"(x)); asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a3));
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a4) : "v"(x)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a5) : "v"(x));
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(a6) : "v"(x)); asm volatile("v_and_b32 %0, %0, %1" : "+v"(a7) : "v"(x));
  }
  return (double)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);
}
__device__ __forceinline__ double run_lds(int n, const double * l, int lane)
{
  double s = 0;
  int idx = lane;
#pragma unroll 4
  for (int i = 0; i < n / 2; i++)
  {
    const double v0 = l[idx], v1 = l[idx + 64], v2 = l[idx + 128], v3 = l[idx + 192];
    s += v0 + v1 + v2 + v3;
    idx = (idx + 1) & 63;
  }
  return s;
}
// kind of the first / second wave of every SIMD: 0 idle, 1 mfma, 2 fma64, 3 int valu, 4 lds
__global__ __launch_bounds__(512) void k(double * out, int k0, int k1)
{
  __shared__ double l[512];
  l[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int kind = (wave < 4) ? k0 : k1; // waves 0..3 -> SIMD 0..3 first slot, 4..7 second slot
  double r = 0;
  if (kind == 1) r = run_mfma(N, 1.0 + 1e-9 * lane, 1e-9 * blockIdx.x);
  else if (kind == 2) r = run_fma(N * 2, 1.0 + 1e-12 * blockIdx.x, 1e-13);
  else if (kind == 3) r = run_int(N * 2, threadIdx.x | 1u);
  else if (kind == 4) r = run_lds(N * 2, l, lane);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double * out;
  hipMalloc(&out, (size_t)cus * 512 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char * nm[5] = {"idle", "mfma", "fma64", "int", "lds"};
  printf("{");
  bool first = true;
  for (int k0 = 1; k0 < 5; k0++)
    for (int k1 = 0; k1 < 5; k1++)
    {
      if (k1 != 0 && k1 < k0)
        continue;
      hipLaunchKernelGGL(k, dim3(cus), dim3(512), 0, 0, out, k0, k1);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 3; r++)
        hipLaunchKernelGGL(k, dim3(cus), dim3(512), 0, 0, out, k0, k1);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s\"%s+%s\": %.3f", first ? "" : ", ", nm[k0], nm[k1], ms / 3);
      first = false;
    }
  printf("}\n");
  return 0;
}
