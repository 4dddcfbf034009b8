// How long does the GPU take to START the 1024 workgroups of one of the engine's
// per-chain kernels (2 wavefronts each, a full register file: 2 waves per SIMD), and
// how long after the last workgroup ends does the NEXT kernel on the stream begin?
// Every workgroup records the 100 MHz wall clock (s_memrealtime) when it starts and
// when it ends; the body spins for a fixed number of shader cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/dispatch_ramp.hip -o /tmp/dispatch_ramp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int VG>
__global__ __launch_bounds__(128, 2) void ramp_kernel(unsigned long long *t0, unsigned long long *t1,
                                                      long long spin, int lds_bytes) {
  extern __shared__ unsigned char smem[];
  const unsigned long long a = wall_clock64();
  // hold many registers live so that the kernel really occupies half a SIMD's file
  double x[VG];
#pragma unroll
  for (int i = 0; i < VG; ++i) x[i] = (double)(threadIdx.x + i);
  const long long c0 = (long long)__builtin_readcyclecounter();
  while ((long long)__builtin_readcyclecounter() - c0 < spin) {
#pragma unroll
    for (int i = 0; i < VG; ++i) x[i] = x[i] * 1.0000001 + 0.5;
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < VG; ++i) s += x[i];
  if (lds_bytes > 0) smem[threadIdx.x] = (unsigned char)s;
  const unsigned long long b = wall_clock64();
  if (threadIdx.x == 0) { t0[blockIdx.x] = a; t1[blockIdx.x] = (s == 12345.678) ? 0 : b; }
}

int main() {
  const int G = 1024, K = 6;
  unsigned long long *d0, *d1;
  hipMalloc(&d0, K * G * 8); hipMalloc(&d1, K * G * 8);
  for (long long spin : {0LL, 20000LL, 80000LL}) {
    for (int lds : {0, 20480, 36864}) {
      hipFuncSetAttribute((const void *)ramp_kernel<100>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      for (int rep = 0; rep < 2; ++rep) {
        for (int k = 0; k < K; ++k)
          hipLaunchKernelGGL(ramp_kernel<100>, dim3(G), dim3(128), lds, 0, d0 + k * G, d1 + k * G, spin, lds);
        hipDeviceSynchronize();
      }
      std::vector<unsigned long long> h0(K * G), h1(K * G);
      hipMemcpy(h0.data(), d0, K * G * 8, hipMemcpyDeviceToHost);
      hipMemcpy(h1.data(), d1, K * G * 8, hipMemcpyDeviceToHost);
      // kernel k = 2..: first start, last start, last end, relative to the previous kernel's last end
      double ramp = 0, dur = 0, gap = 0;
      for (int k = 2; k < K; ++k) {
        auto b0 = h0.begin() + k * G, b1 = h1.begin() + k * G;
        const unsigned long long fs = *std::min_element(b0, b0 + G), ls = *std::max_element(b0, b0 + G);
        const unsigned long long le = *std::max_element(b1, b1 + G);
        const unsigned long long ple = *std::max_element(h1.begin() + (k - 1) * G, h1.begin() + k * G);
        ramp += (ls - fs) / 100.0; dur += (le - fs) / 100.0; gap += ((long long)fs - (long long)ple) / 100.0;
      }
      printf("spin %6lld cycles, LDS %5d B: first->last workgroup start %.2f us, first start->last end %.2f us, "
             "previous kernel's last end -> first start %.2f us\n", spin, lds, ramp / (K - 2), dur / (K - 2), gap / (K - 2));
    }
  }
  return 0;
}
