// Micro-benchmark: cost of the hand-over store pattern of lane_tree_body.  W wavefronts (one per block), each writes
// F rows of 512 bytes (one double per lane) into its own tile of F rows, with a little arithmetic between the rows.
//   variant 0: rows in order, plain stores            (what lane_tree_body does)
//   variant 1: the same with non-temporal stores
//   variant 2: plain stores, every wavefront into one of 64 tiles only (L2-resident target: the 0.25 ms reference point)
//   variant 3: 16 bytes per lane (rows of 1 KB, half as many store instructions)
//   variant 4: AoS target: lane l writes its own block of F doubles (8-byte scattered stores: the first lane_tree version)
// Build: hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(64) void k(double * out, int F, size_t tile_doubles, int spin)
{
  const int lane = threadIdx.x;
  size_t tile = blockIdx.x;
  if (V == 2)
    tile &= 63;
  double * base = out + tile * tile_doubles;
  double acc = lane * 1e-3 + blockIdx.x;
  for (int f = 0; f < F; f++)
  {
    for (int s = 0; s < spin; s++)
      acc = __builtin_fma(acc, 1.0000001, 1e-9);
    if (V == 1)
      __builtin_nontemporal_store(acc, &base[(size_t)f * 64 + lane]);
    else if (V == 3)
    {
      if ((f & 1) == 0)
        *reinterpret_cast<d2 *>(&base[(size_t)f * 64 + 2 * lane]) = d2{acc, acc + 1.0};
    }
    else if (V == 4)
      base[(size_t)lane * F + f] = acc;
    else
      base[(size_t)f * 64 + lane] = acc;
  }
}

int main(int argc, char ** argv)
{
  const int W = argc > 1 ? atoi(argv[1]) : 3264, F = argc > 2 ? atoi(argv[2]) : 619, spin = argc > 3 ? atoi(argv[3]) : 10;
  const size_t tile = (size_t)F * 64 + 144;
  double * out;
  hipMalloc(&out, (size_t)W * tile * sizeof(double));
  hipMemset(out, 0, (size_t)W * tile * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  std::printf("{\"waves\": %d, \"rows\": %d, \"GB\": %.3f", W, F, (double)W * F * 512 / 1e9);
  for (int v = 0; v < 5; v++)
  {
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++)
    {
      hipEventRecord(e0);
      switch (v)
      {
      case 0: hipLaunchKernelGGL(k<0>, dim3(W), dim3(64), 0, 0, out, F, tile, spin); break;
      case 1: hipLaunchKernelGGL(k<1>, dim3(W), dim3(64), 0, 0, out, F, tile, spin); break;
      case 2: hipLaunchKernelGGL(k<2>, dim3(W), dim3(64), 0, 0, out, F, tile, spin); break;
      case 3: hipLaunchKernelGGL(k<3>, dim3(W), dim3(64), 0, 0, out, F, tile, spin); break;
      default: hipLaunchKernelGGL(k<4>, dim3(W), dim3(64), 0, 0, out, F, tile, spin); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best)
        best = ms;
    }
    std::printf(", \"v%d_ms\": %.3f", v, best);
  }
  std::printf("}\n");
  return 0;
}
