// Where do a workgroup's wavefronts land?  1024 workgroups of two waves (four per CU, as the
// sweep kernels run): every wave records HW_ID (wave slot, SIMD, CU, SE, XCC) and the
// workgroup holds its slot for a while so that all are resident together.  Prints, per CU of
// XCC 0, the (SIMD, slot) of wave 0 / wave 1 of each resident workgroup, and a histogram of
// SIMD pairs.  Diagnostic (placement-dependent kernel times).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned *out, long long ticks) {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID, all 32 bits
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);     // XCC_ID
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 2 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int main() {
  const int n = 1024;
  unsigned *d;
  hipMalloc(&d, n * 4 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(n), dim3(128), 40 * 1024, 0, d, 200000ll);   // 40 KB of LDS: four workgroups per CU
  hipDeviceSynchronize();
  std::vector<unsigned> h(n * 4);
  hipMemcpy(h.data(), d, n * 4 * sizeof(unsigned), hipMemcpyDeviceToHost);
  int pair_hist[4][4] = {};
  for (int b = 0; b < n; ++b) {
    const unsigned a0 = h[b * 4], a1 = h[b * 4 + 2];
    // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    const int s0 = (a0 >> 4) & 3, s1 = (a1 >> 4) & 3;
    pair_hist[s0][s1]++;
    if (b < 24 || (b % 8 == 0 && b < 200))
      printf("wg %4d: wave0 slot %2u simd %d cu %2u se %u xcc %u | wave1 slot %2u simd %d cu %2u se %u\n", b, a0 & 15, s0,
             (a0 >> 8) & 15, (a0 >> 13) & 7, h[b * 4 + 1] & 15, a1 & 15, s1, (a1 >> 8) & 15, (a1 >> 13) & 7);
  }
  // raw HW_ID of the workgroups that share CU (xcc 0, se 0, cu 1) -- which bits tell them apart
  for (int b = 0; b < n; ++b) {
    const unsigned a0 = h[b * 4];
    if ((h[b * 4 + 1] & 15) == 0 && ((a0 >> 13) & 7) == 0 && ((a0 >> 8) & 15) == 1)
      printf("same CU: wg %4d wave0 %08x wave1 %08x\n", b, a0, h[b * 4 + 2]);
  }
  printf("SIMD of (wave 0, wave 1), workgroups:\n");
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) printf(" %5d", pair_hist[i][j]); printf("\n"); }
  return 0;
}
