// Checks the operand / result lane maps assumed for v_mfma_f64_16x16x4_f64 (one wave):
//   A[i][k]: lane = i + 16 k ; B[k][j]: lane = j + 16 k ; D[(lane>>4) + 4 v][lane & 15] in result register v.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double * A, const double * B, double * D)
{
  const int l = threadIdx.x;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], c, 0, 0, 0);
  for (int v = 0; v < 4; v++)
    D[((l >> 4) + 4 * v) * 16 + (l & 15)] = c[v];
}
int main()
{
  double hA[64], hB[64], hD[256], *A, *B, *D;
  for (int i = 0; i < 64; i++) { hA[i] = sin(1.0 + 3.7 * i); hB[i] = cos(0.3 + 1.9 * i); }
  (void)hipMalloc(&A, sizeof hA); (void)hipMalloc(&B, sizeof hB); (void)hipMalloc(&D, sizeof hD);
  (void)hipMemcpy(A, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(B, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D);
  (void)hipMemcpy(hD, D, sizeof hD, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++)
    {
      double s = 0;
      for (int kk = 0; kk < 4; kk++) s = fma(hA[i * 4 + kk], hB[kk * 16 + j], s);
      worst = fmax(worst, fabs(s - hD[i * 16 + j]));
    }
  printf("mfma_f64_16x16x4 layout check: max |diff| = %.3e (%s)\n", worst, worst < 1e-14 ? "OK" : "MISMATCH");
  return worst < 1e-14 ? 0 : 1;
}
