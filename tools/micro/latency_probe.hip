// Micro-benchmark: dependent-issue latencies a single wavefront sees on gfx950 (what bounds the lane-phase kernels: DESIGN 3.1/3.6).
// One wave, dependent chains of N operations, cycles per operation from s_memtime.
// Build: hipcc --offload-arch=gfx950 -O3 latency_probe.hip -o latency_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

constexpr int N = 256;
__device__ __forceinline__ long long clk(double & dep)
{
  long long t;
  asm volatile("" : "+v"(dep)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  asm volatile("" : "+v"(dep)::"memory");
  return t;
}
#define T0 long long t0 = clk(acc);
#define T1(slot)                                                                                                       \
  long long t1 = clk(acc);                                                                                             \
  if (threadIdx.x == 0)                                                                                                \
    cyc[slot] = (double)(t1 - t0) / N;

__global__ void k_probe(double * out, double * cyc, double seed)
{
  __shared__ double lds[1024];
  const int lane = threadIdx.x;
  lds[lane] = seed + lane;
  lds[lane + 64] = seed * 2 + lane;
  __syncthreads();
  double acc = seed + lane * 1e-3;
  { // 0: dependent v_fma_f64 chain
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
      acc = __builtin_fma(acc, 1.0000001, 1e-9);
    T1(0);
  }
  { // 1: two independent fma chains (ILP 2): cycles per pair
    double b = acc + 1.0;
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
    {
      acc = __builtin_fma(acc, 1.0000001, 1e-9);
      b = __builtin_fma(b, 1.0000002, 1e-9);
    }
    acc += b;
    T1(1);
  }
  { // 2: dependent v_add_f64
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
      acc = acc + 1e-9;
    T1(2);
  }
  { // 3: LDS round trip: pointer chase  idx = lds[idx]
    int idx = lane;
    __shared__ int nxt[64];
    nxt[lane] = (lane * 7 + 3) & 63;
    __syncthreads();
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
      idx = ((volatile int *)nxt)[idx];
    acc += idx;
    T1(3);
  }
  { // 4: LDS write -> fence -> read of another lane's value -> fma (one "lane phase" boundary)
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
    {
      lds[128 + lane] = acc;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      acc = __builtin_fma(lds[128 + ((lane + 1) & 63)], 1.0000001, 1e-9);
    }
    T1(4);
  }
  { // 5: full IEEE division chain
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
      acc = 1.0 / (acc + 2.0);
    T1(5);
  }
  { // 6: rcp + 2 Newton steps chain
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
    {
      const double x = acc + 2.0;
      double r = __builtin_amdgcn_rcp(x);
      r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
      r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
      acc = r;
    }
    T1(6);
  }
  { // 7: sqrt chain
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
      acc = sqrt(acc + 2.0);
    T1(7);
  }
  { // 8: sincos chain (ocml)
    T0;
#pragma unroll 4
    for (int i = 0; i < N; i++)
    {
      double s, c;
      sincos(acc, &s, &c);
      acc = s + c;
    }
    T1(8);
  }
  { // 9: v_readlane pair -> SGPR -> fma
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
    {
      union { double d; int w[2]; } u;
      u.d = acc;
      u.w[0] = __builtin_amdgcn_readlane(u.w[0], 5);
      u.w[1] = __builtin_amdgcn_readlane(u.w[1], 5);
      acc = __builtin_fma(u.d, 1.0000001, acc * 1e-9);
    }
    T1(9);
  }
  { // 10: DPP row_shr / ds_swizzle-free cross-lane: __shfl via ds_bpermute
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
      acc = __builtin_fma(__shfl(acc, (lane + 1) & 63), 1.0000001, 1e-9);
    T1(10);
  }
  { // 11: dependent v_mfma_f64_16x16x4 chain (same accumulator)
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 c = {acc, acc, acc, acc};
    T0;
#pragma unroll
    for (int i = 0; i < N; i++)
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, c, 0, 0, 0);
    acc += c[0] + c[1] + c[2] + c[3];
    T1(11);
  }
  { // 12: global load round trip (pointer chase through an L2-resident table)
    T0;
    int idx = lane;
    const int * tab = (const int *)out; // zeros written by the host: idx stays a valid index
#pragma unroll 8
    for (int i = 0; i < N; i++)
      idx = tab[idx] + lane;
    acc += idx;
    T1(12);
  }
  out[64 + lane] = acc;
}

int main()
{
  double *out, *cyc;
  hipMalloc(&out, 4096);
  hipMemset(out, 0, 4096);
  hipMalloc(&cyc, 16 * sizeof(double));
  for (int rep = 0; rep < 2; rep++)
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, out, cyc, 0.5);
  double h[16];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const char * names[13] = {"fma64_dep", "fma64_ilp2_pair", "add64_dep", "lds_chase", "lds_phase_roundtrip", "div64", "rcp_nr", "sqrt64", "sincos64", "readlane_fma",
                            "bpermute_fma", "mfma_f64_dep", "global_chase"};
  std::printf("{");
  for (int i = 0; i < 13; i++)
    std::printf("\"%s\": %.1f%s", names[i], h[i], i < 12 ? ", " : "}\n");
  return 0;
}
