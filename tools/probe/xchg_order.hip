// Probe: are same-address LDS exchanges of ONE wavefront instruction resolved
// in ascending lane order on this device?  (The shuffle's previous-step search
// can then be a single ds_wrxchg_rtn per 64 steps.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void probe(const int *keys, int *old_out, int *final_out, int ncase, int nkeys) {
  __shared__ unsigned X[64];
  for (int c = blockIdx.x; c < ncase; c += gridDim.x) {
    if (threadIdx.x < 64) X[threadIdx.x] = 0xFFFFu;
    __syncthreads();
    const int key = keys[c * 64 + threadIdx.x];
    const unsigned old = atomicExch(&X[key], (unsigned)(threadIdx.x + 100));
    old_out[c * 64 + threadIdx.x] = (int)old;
    __syncthreads();
    if ((int)threadIdx.x < nkeys) final_out[c * 64 + threadIdx.x] = (int)X[threadIdx.x];
    __syncthreads();
  }
}

int main() {
  const int ncase = 20000;
  std::vector<int> keys(ncase * 64);
  unsigned long long s = 12345;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  long bad = 0, badfinal = 0;
  for (int nkeys : {1, 2, 5, 17, 64}) {
    for (auto &k : keys) k = (int)(rnd() % nkeys);
    int *dk, *dold, *dfin;
    CK(hipMalloc(&dk, keys.size() * 4)); CK(hipMalloc(&dold, keys.size() * 4)); CK(hipMalloc(&dfin, keys.size() * 4));
    CK(hipMemcpy(dk, keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
    probe<<<1024, 64>>>(dk, dold, dfin, ncase, nkeys);
    CK(hipDeviceSynchronize());
    std::vector<int> old(keys.size()), fin(keys.size());
    CK(hipMemcpy(old.data(), dold, keys.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(fin.data(), dfin, keys.size() * 4, hipMemcpyDeviceToHost));
    for (int c = 0; c < ncase; ++c) {
      int lastw[64];
      for (int i = 0; i < 64; ++i) lastw[i] = 0xFFFF;
      for (int l = 0; l < 64; ++l) {
        const int k = keys[c * 64 + l];
        if (old[c * 64 + l] != lastw[k]) ++bad;
        lastw[k] = l + 100;
      }
      for (int k = 0; k < nkeys; ++k) if (fin[c * 64 + k] != lastw[k]) ++badfinal;
    }
    printf("nkeys %2d: %ld lanes out of ascending-lane order, %ld wrong finals (of %d cases)\n", nkeys, bad, badfinal, ncase);
    CK(hipFree(dk)); CK(hipFree(dold)); CK(hipFree(dfin));
  }
  printf(bad == 0 && badfinal == 0 ? "ASCENDING LANE ORDER HOLDS\n" : "ORDER NOT GUARANTEED\n");
  return 0;
}
