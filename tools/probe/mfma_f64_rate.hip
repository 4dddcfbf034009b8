// What one MI355X sustains on v_mfma_f64_16x16x4_f64 (the guide's matrix-core table
// has no f64 row): back-to-back MFMAs on NACC independent accumulators, WPS waves
// per SIMD, every CU.  Prints TFLOP/s and cycles per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_f64_rate.hip -o /tmp/mfma_f64_rate && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wgs_per_cu) {
  const int cus = 256, iters = 20000;
  double *out;
  hipMalloc(&out, (size_t)cus * wgs_per_cu * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, 100, 1.0, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.0, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = (double)cus * wgs_per_cu * 4 * iters * NACC;
  const double flops = mfmas * 16 * 16 * 4 * 2;
  printf("NACC %d, %d waves/SIMD: %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", NACC, wgs_per_cu,
         flops / ms / 1e9, ms * 1e6 / (mfmas / (cus * 4)));
  hipFree(out);
}
int main() {
  run<4>(1); run<4>(2); run<4>(3); run<4>(4); run<8>(1); run<8>(2); run<16>(1); run<2>(4); run<2>(8);
  return 0;
}
