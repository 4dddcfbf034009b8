// Calibration probe (diagnostic, not product): clock of s_memtime, latency of
// dependent global loads for several footprints, scalar-load latency, LDS
// read latency, readlane / FMA dependent-issue cost -- one wave per CU like
// the sweep kernel's serial parts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void clock_probe(long long *out, int spin) {
  long long t0 = __builtin_readcyclecounter();
  long long w0 = wall_clock64();
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0000001 + 1e-9;
  long long t1 = __builtin_readcyclecounter();
  long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = (long long)x; }
}

// pointer chase: idx = buf[idx]; all lanes same chain (uniform) or per-lane chains
__global__ void chase(const int *buf, int steps, int stride_lanes, long long *out) {
  int idx = (threadIdx.x * stride_lanes) ;
  // warm
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < steps; ++i) idx = buf[idx];
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = idx; }
}

// 16 independent loads per lane then a dependent use, repeated
__global__ void burst16(const double *buf, int nloads, int reps, int coalesced, long long *out) {
  unsigned base = blockIdx.x * 7919u;
  double acc = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; ++r) {
    double v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const unsigned row = (base + m * 37u + r * 101u) & 511u;
      const unsigned a = coalesced ? (row * 512u + threadIdx.x) : (((threadIdx.x * 8u + (base & 7u)) & 511u) * 512u + row);
      v[m] = (m < nloads) ? buf[a] : 0.0;
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) acc += v[m];
    base = base + (unsigned)(acc != 12345.0);
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (long long)acc; }
}

__global__ void lds_chase(int steps, long long *out) {
  __shared__ int tab[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) tab[i] = (i * 37 + 11) & 1023;
  __syncthreads();
  int idx = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < steps; ++i) idx = tab[idx];
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = idx; }
}

__global__ void fma_chain(int steps, long long *out) {
  double x = threadIdx.x * 1e-3, y = 1.0000001;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < steps; ++i) x = __builtin_fma(x, y, 1e-9);
  long long t1 = __builtin_readcyclecounter();
  int s = 0;
  long long t2 = __builtin_readcyclecounter();
  double z = x;
  for (int i = 0; i < steps; ++i) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(z), i & 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(z), i & 63);
    z = __hiloint2double(hi, lo) * 1.0000001 + threadIdx.x;
  }
  long long t3 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = t3 - t2; out[blockIdx.x * 4 + 2] = (long long)(x + z) + s; }
}

__global__ void barrier_probe(int steps, long long *out) {
  __shared__ double slot[8];
  long long t0 = __builtin_readcyclecounter();
  double x = threadIdx.x;
  for (int i = 0; i < steps; ++i) {
    if (threadIdx.x == 0) slot[0] = x;
    __syncthreads();
    x += slot[0];
    __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (long long)x; }
}

int main() {
  long long *dout; CK(hipMalloc(&dout, 1 << 20));
  std::vector<long long> h(4096);
  // clock
  {
    auto w0 = std::chrono::high_resolution_clock::now();
    clock_probe<<<1, 64>>>(dout, 20000000);
    CK(hipDeviceSynchronize());
    auto w1 = std::chrono::high_resolution_clock::now();
    CK(hipMemcpy(h.data(), dout, 24, hipMemcpyDeviceToHost));
    double us = std::chrono::duration<double, std::micro>(w1 - w0).count();
    printf("clock: readcyclecounter %lld ticks, wall_clock64 %lld ticks in ~%.0f us host => cyclecounter %.1f MHz, wall_clock %.1f MHz\n",
           h[0], h[1], us, h[0] / us, h[1] / us);
  }
  const int blocks = 1024;
  for (size_t mb : {1, 4, 16, 64, 1024}) {
    size_t n = mb * 1024 * 1024 / 4;
    std::vector<int> hb(n);
    // random cyclic permutation with large jumps
    std::vector<int> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = (int)i;
    unsigned long long s = 88172645463325252ull;
    for (size_t i = n - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; size_t j = s % i; std::swap(perm[i], perm[j]); }
    for (size_t i = 0; i < n; ++i) hb[perm[i]] = perm[(i + 1) % n];
    int *db; CK(hipMalloc(&db, n * 4));
    CK(hipMemcpy(db, hb.data(), n * 4, hipMemcpyHostToDevice));
    for (int lanes_stride : {0, 4099}) {
      chase<<<blocks, 64>>>(db, 2000, lanes_stride, dout);
      chase<<<blocks, 64>>>(db, 2000, lanes_stride, dout);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), dout, blocks * 16, hipMemcpyDeviceToHost));
      double sum = 0; for (int b = 0; b < blocks; ++b) sum += h[b * 2];
      printf("chase %4zu MB, %s: %.0f ticks per dependent load (1024 waves in flight)\n", mb, lanes_stride ? "divergent lanes" : "uniform lanes  ", sum / blocks / 2000);
    }
    CK(hipFree(db));
  }
  {
    size_t n = 512 * 512;
    double *db; CK(hipMalloc(&db, n * 8)); CK(hipMemset(db, 0, n * 8));
    for (int nb : {1024, 2048}) for (int nl : {1, 4, 16}) for (int co : {1, 0}) {
      burst16<<<nb, 64>>>(db, nl, 200, co, dout);
      burst16<<<nb, 64>>>(db, nl, 200, co, dout);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), dout, 1024 * 16, hipMemcpyDeviceToHost));
      double sum = 0; for (int b = 0; b < 1024; ++b) sum += h[b * 2];
      printf("burst of %2d loads/lane from a 2 MB matrix, %d waves, %s: %.0f clocks per burst (incl. 16 dependent adds)\n", nl, nb, co ? "coalesced rows" : "scattered     ", sum / 1024 / 200);
    }
  }
  for (int nt : {64, 128, 256}) {
    barrier_probe<<<1024, nt>>>(1000, dout); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), dout, 1024 * 16, hipMemcpyDeviceToHost));
    double sum = 0; for (int b = 0; b < 1024; ++b) sum += h[b * 2];
    printf("write + barrier + read + barrier, %d threads per workgroup: %.0f clocks per pair of barriers\n", nt, sum / 1024 / 1000);
  }
  lds_chase<<<blocks, 64>>>(2000, dout); CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), dout, 16, hipMemcpyDeviceToHost));
  printf("lds dependent read: %.1f ticks\n", h[0] / 2000.0);
  fma_chain<<<blocks, 64>>>(2000, dout); CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), dout, 32, hipMemcpyDeviceToHost));
  printf("dependent f64 fma: %.1f ticks; readlane x2 + fma: %.1f ticks\n", h[0] / 2000.0, h[1] / 2000.0);
  return 0;
}
