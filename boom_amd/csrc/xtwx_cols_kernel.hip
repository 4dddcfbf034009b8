// Columns of the logit sampler's complete-data precision, as the sweep needs them.
//
// BinomialLogitSpikeSlabSampler rebuilds xtx = sum_i w_i x_i x_i' -- all p^2 of it,
// n p^2 flops -- after every imputation (BinomialLogitCltDataImputer /
// BinomialLogitSpikeSlabSampler.cpp:50-54, the suf_.update loop).  The sweep that
// follows reads far less of it: log_model_prob of "gamma with j flipped" needs
// V[j, gamma] and V[j, j] (SpikeSlabSampler.cpp:171-203), so with k variables
// included only the k vectors V[., g], g in gamma, and the diagonal are ever
// touched -- p k + p of the p^2 elements -- plus one more vector each time a
// variable enters the model.  Those vectors are what is built here, for all chains
// at once:
//
//   request r = (chain c_r, variable g_r):   V_c[., g_r] = Omega^{-1}[., g_r] + X'(w_c o x_{g_r})
//
// as ONE GEMM over the request list: C[r, j] = sum_i (w_{c_r}[i] X[i, g_r]) X[i, j],
// R x p x n, the weighted left operand formed while it is staged.  2 n p R flops
// with R = sum_c k_c instead of 2 n p^2 C / 2: at n = 5e4, p = 1024, k = 10 that is
// 50 times less work for the same draws.
//
// The K (row) range is cut into chunks of a FIXED length, one workgroup per
// (request tile, variable tile, chunk), partial tiles to planes that a second
// kernel adds in chunk order: an element's value depends on n alone -- not on how
// many requests share the launch, so not on how many chains the engine holds.
//
// f64 matrix cores (v_mfma_f64_16x16x4_f64): 256 threads own a 64-request x
// 128-variable tile, each wavefront a 32 x 64 part as 2 x 4 MFMA tiles; 16-row
// panels double-buffered in LDS with a stride of 18 doubles (== 2 mod 32), so
// that the 16 x 4 fragment reads are conflict-free.
#include <hip/hip_runtime.h>
#include "ktimer.h"
#include <stdint.h>

namespace boom_amd {

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

// One 8-byte LDS read as its own instruction, and the explicit wait that goes with it: see
// suf_kernel.hip (left to the compiler the fragment reads pair up into ds_read2_b64, which is
// serviced 16 lanes at a time against 32 banks -- 2-way conflicted on this layout and four
// times the LDS cycles of the ds_read_b64 pair; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.40
// in round 5's counters).
template <int OFF>
__device__ __forceinline__ double lds_read_b64(uint32_t byte_addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(double &a, double &b, double &c, double &d, double &e, double &f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N));
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

constexpr int RT = 64;        // requests per tile
constexpr int JT = 128;       // variables per tile
constexpr int KS = 16;        // rows per staging step
constexpr int LDS_LD = KS + 2;
constexpr int KCHUNK = 2048;  // rows per plane (fixed: see above)

// GATHER: row r of the left operand is w_{c_r} o x_{g_r} (the requests); otherwise it
// is row r of w as it stands (R rows of n: the plain product w X, e.g. X'Wz of every
// chain with w = the chains' weighted latent sums)
// KCH: rows per plane.  ldw (GATHER = false): doubles between the rows of the left operand.
template <bool GATHER, int KCH>
__global__ __launch_bounds__(256, 2) void xtwx_cols_kernel(const double *__restrict__ X, int64_t n, int p,
                                                          const double *__restrict__ w, int64_t ldw,
                                                          const int2 *__restrict__ req, int R,
                                                          double *__restrict__ planes) {
  __shared__ double sA[2][RT * LDS_LD];
  __shared__ double sB[2][JT * LDS_LD];
  const int J0 = blockIdx.x * JT, R0 = blockIdx.y * RT;
  const int64_t kbeg = (int64_t)blockIdx.z * KCH;
  const int64_t kend = (kbeg + KCH < n) ? kbeg + KCH : n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wj = wave & 1;   // the wave's 32 x 64 part of the tile
  const int fr = lane >> 4, fc = lane & 15;

  double4_t acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

  // this thread's share of a panel: row prow of the step, columns pcol0 + 16 it
  const int prow = tid & 15, pcol0 = tid >> 4;
  // (rows behind R and columns behind p read row / column 0 and are zeroed after the load, steps
  // behind the plane's end read its last row: every load of a step is unconditional, so all
  // twelve to sixteen go out together.  As conditional loads -- "in range ? X[...] : 0" -- the
  // compiler gave each its own branch and its own wait: four memory round trips in a row in
  // every 16-row step of the main loop; tools/isa_serial_loads.py)
  const double *xa[4], *wa[4];
  bool rowok[4], colok[8];
  const double *xb[8];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int r = R0 + pcol0 + 16 * it;
    rowok[it] = r < R;
    const int2 q = GATHER ? req[rowok[it] ? r : 0] : make_int2(rowok[it] ? r : 0, 0);
    xa[it] = X + (int64_t)q.y * n;
    wa[it] = w + (int64_t)q.x * (GATHER ? n : ldw);
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int j = J0 + pcol0 + 16 * it;
    colok[it] = j < p;
    xb[it] = X + (int64_t)(colok[it] ? j : 0) * n;
  }
  // fetch: the loads of a step go out (before the step's matrix instructions); finish: what they
  // brought is masked and multiplied (after them), then stashed
  double ra[4], rb[8], rx[4], rw[4];
  bool in = false;
  auto fetch = [&](int64_t k0) {
    const int64_t kr = k0 + prow;
    in = kr < kend;
    const int64_t k = in ? kr : kend - 1;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      rw[it] = wa[it][k];
      rx[it] = GATHER ? xa[it][k] : 1.0;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) rb[it] = xb[it][k];
  };
  auto finish = [&]() {
    asm volatile("" : "+v"(rw[0]), "+v"(rw[1]), "+v"(rw[2]), "+v"(rw[3]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]),
                      "+v"(rb[4]), "+v"(rb[5]), "+v"(rb[6]), "+v"(rb[7]));
    if (GATHER) asm volatile("" : "+v"(rx[0]), "+v"(rx[1]), "+v"(rx[2]), "+v"(rx[3]));
#pragma unroll
    for (int it = 0; it < 4; ++it) ra[it] = (in && rowok[it]) ? (GATHER ? rx[it] * rw[it] : rw[it]) : 0.0;
#pragma unroll
    for (int it = 0; it < 8; ++it) rb[it] = (in && colok[it]) ? rb[it] : 0.0;
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 4; ++it) sA[buf][(pcol0 + 16 * it) * LDS_LD + prow] = ra[it];
#pragma unroll
    for (int it = 0; it < 8; ++it) sB[buf][(pcol0 + 16 * it) * LDS_LD + prow] = rb[it];
  };
  fetch(kbeg);
  finish();
  stash(0);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = kbeg; k0 < kend; k0 += KS) {
    const bool more = k0 + KS < kend;
    if (more) fetch(k0 + KS);
    {
      // fragment reads one matrix step ahead of their use, each an instruction of its own
      const uint32_t pa = lds_addr(&sA[cur][(wr * 32 + fc) * LDS_LD + fr]);
      const uint32_t pb = lds_addr(&sB[cur][(wj * 64 + fc) * LDS_LD + fr]);
      constexpr int C16 = 16 * LDS_LD * 8;   // sixteen columns on
      double a[2][2], b[2][4];
#define BA_RD(S, K)                                                                                   \
      a[S][0] = lds_read_b64<(K) * 32>(pa); a[S][1] = lds_read_b64<(K) * 32 + C16>(pa);                 \
      b[S][0] = lds_read_b64<(K) * 32>(pb); b[S][1] = lds_read_b64<(K) * 32 + C16>(pb);                 \
      b[S][2] = lds_read_b64<(K) * 32 + 2 * C16>(pb); b[S][3] = lds_read_b64<(K) * 32 + 3 * C16>(pb);
#define BA_MM(S)                                                                                      \
      _Pragma("unroll") for (int ta = 0; ta < 2; ++ta)                                                \
      _Pragma("unroll") for (int tb = 0; tb < 4; ++tb)                                                \
        acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[S][ta], b[S][tb], acc[ta][tb], 0, 0, 0);
      static_assert(KS == 16, "four matrix steps per staging step");
      BA_RD(0, 0)
      BA_RD(1, 1)
      lds_wait<6>(a[0][0], a[0][1], b[0][0], b[0][1], b[0][2], b[0][3]);
      BA_MM(0)
      BA_RD(0, 2)
      lds_wait<6>(a[1][0], a[1][1], b[1][0], b[1][1], b[1][2], b[1][3]);
      BA_MM(1)
      BA_RD(1, 3)
      lds_wait<6>(a[0][0], a[0][1], b[0][0], b[0][1], b[0][2], b[0][3]);
      BA_MM(0)
      lds_wait<0>(a[1][0], a[1][1], b[1][0], b[1][1], b[1][2], b[1][3]);
      BA_MM(1)
#undef BA_RD
#undef BA_MM
    }
    if (more) { finish(); stash(cur ^ 1); }   // (the other buffer was last read before the previous barrier)
    __syncthreads();
    cur ^= 1;
  }
  double *o = planes + (size_t)blockIdx.z * (size_t)R * (size_t)p;
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int tb = 0; tb < 4; ++tb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // register q of lane l holds row (l >> 4) + 4 q, column l & 15
        const int r = R0 + wr * 32 + ta * 16 + fr + 4 * q;
        const int j = J0 + wj * 64 + tb * 16 + fc;
        if (r < R && j < p) o[(size_t)r * p + j] = acc[ta][tb][q];
      }
}

// V_c[., g_r] = base[., g_r] + the planes in chunk order; the vector is marked valid
__global__ __launch_bounds__(256) void xtwx_cols_reduce_kernel(const double *__restrict__ planes, int nplanes,
                                                              const int2 *__restrict__ req, int R, int p,
                                                              const double *__restrict__ base,
                                                              double *__restrict__ V, uint32_t *__restrict__ valid,
                                                              int words) {
  const int r = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int2 q = req[r];
  if (j < p) {
    const size_t plane = (size_t)R * (size_t)p;
    const double *s = planes + (size_t)r * p + j;
    double a = s[0];
    for (int z0 = 1; z0 < nplanes; z0 += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (z0 + u < nplanes) ? s[(size_t)(z0 + u) * plane] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (z0 + u < nplanes) a += v[u];
    }
    V[((size_t)q.x * p + (size_t)q.y) * (size_t)p + j] = a + base[(size_t)q.y * p + j];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(valid + (size_t)q.x * words + (q.y >> 5), 1u << (q.y & 31));
}

// out[r, j] = the planes in chunk order (+ diag_base[j, j])
__global__ __launch_bounds__(256) void plain_reduce_kernel(const double *__restrict__ planes, int nplanes,
                                                          int R, int p, const double *__restrict__ diag_base,
                                                          double *__restrict__ out) {
  const size_t count = (size_t)R * (size_t)p;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  // (eight planes' loads in flight at a time; the sum stays in plane order)
  double a = planes[i];
  for (int z0 = 1; z0 < nplanes; z0 += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (z0 + u < nplanes) ? planes[(size_t)(z0 + u) * count + i] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (z0 + u < nplanes) a += v[u];
  }
  if (diag_base) {
    const size_t j = i % (size_t)p;
    a += diag_base[j * p + j];
  }
  out[i] = a;
}

// the requests of a sweep's start: every chain's included variables.  One workgroup
// per chain; the chain's entries are contiguous and sorted (where the block lands in
// the list depends on the order the workgroups arrive: immaterial, see the header).
__global__ __launch_bounds__(256) void xtwx_cols_start_kernel(const uint8_t *__restrict__ gamma, int p,
                                                             int2 *__restrict__ req, int *__restrict__ count,
                                                             uint32_t *__restrict__ valid, int words) {
  __shared__ int s_cnt[256];
  __shared__ int s_base;
  const int c = blockIdx.x, tid = threadIdx.x;
  const uint8_t *g = gamma + (size_t)c * p;
  const int per = (p + 255) / 256;
  const int j0 = tid * per, j1 = (j0 + per < p) ? j0 + per : p;
  int mine = 0;
  for (int j = j0; j < j1; ++j) mine += g[j] ? 1 : 0;
  s_cnt[tid] = mine;
  for (int i = tid; i < words; i += 256) valid[(size_t)c * words + i] = 0u;
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int t = 0; t < 256; ++t) {
      const int v = s_cnt[t];
      s_cnt[t] = tot;
      tot += v;
    }
    s_base = tot ? atomicAdd(count, tot) : 0;
  }
  __syncthreads();
  int at = s_base + s_cnt[tid];
  for (int j = j0; j < j1; ++j)
    if (g[j]) req[at++] = make_int2(c, j);
}

__global__ __launch_bounds__(256) void square_kernel(const double *__restrict__ x, size_t count,
                                                    double *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < count) out[i] = x[i] * x[i];
}

}  // namespace

int xtwx_cols_planes(int64_t n) { return (int)((n + KCHUNK - 1) / KCHUNK); }

// the vectors of V named by req[0, R): planes = workspace of xtwx_cols_planes(n) R p doubles
hipError_t launch_xtwx_cols(hipStream_t stream, const double *X, int64_t n, int p, const double *w,
                            const int32_t *req, int R, const double *base, double *V,
                            uint32_t *valid, int words, double *planes) {
  if (R <= 0) return hipSuccess;
  const int np = xtwx_cols_planes(n);
  KtScope kt(stream, KT_COLS_GEMM);
  hipLaunchKernelGGL((xtwx_cols_kernel<true, KCHUNK>), dim3((p + JT - 1) / JT, (R + RT - 1) / RT, np), dim3(256), 0,
                     stream, X, n, p, w, n, (const int2 *)req, R, planes);
  hipLaunchKernelGGL(xtwx_cols_reduce_kernel, dim3((p + 255) / 256, R), dim3(256), 0, stream,
                     planes, np, (const int2 *)req, R, p, base, V, valid, words);
  return hipGetLastError();
}

// out (R x p, row-major) = U B with U: R rows of n, B: p K-contiguous columns of n
// (+ diag_base[j, j] on every row when given); planes as above
hipError_t launch_rows_times_columns(hipStream_t stream, const double *U, int R, const double *B, int64_t n,
                                     int p, const double *diag_base, double *out, double *planes) {
  if (R <= 0) return hipSuccess;
  const int np = xtwx_cols_planes(n);
  KtScope kt(stream, KT_ROWS_GEMM);
  hipLaunchKernelGGL((xtwx_cols_kernel<false, KCHUNK>), dim3((p + JT - 1) / JT, (R + RT - 1) / RT, np), dim3(256), 0,
                     stream, B, n, p, U, n, (const int2 *)nullptr, R, planes);
  const size_t cnt = (size_t)R * p;
  hipLaunchKernelGGL(plain_reduce_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, planes, np, R,
                     p, diag_base, out);
  return hipGetLastError();
}

// The same product for SHORT rows (the bsts path's X'e: chains x p x T with T a few
// thousand): rows cut into planes of XTE_KCHUNK so that the launch has enough workgroups
// -- 16 row tiles x 16 planes at 1024 chains, T = 2000 -- each reading a 64-chain x
// 128-step slab of the residuals and a 128-variable x 128-step slab of X ONCE, coalesced,
// through LDS (the 16 x 16 tiles of atb_mfma_kernel re-read both operands per tile: 220 MB
// of L2 traffic per round against 42 MB).  ldu: doubles between the rows of U.
constexpr int XTE_KCHUNK = 128;
int xte_planes(int64_t n) { return (int)((n + XTE_KCHUNK - 1) / XTE_KCHUNK); }
hipError_t launch_xte_tiled(hipStream_t stream, const double *U, int64_t ldu, int R, const double *B, int64_t n,
                            int p, double *out, double *planes) {
  if (R <= 0) return hipSuccess;
  const int np = xte_planes(n);
  KtScope kt(stream, KT_XTE_GEMM);
  hipLaunchKernelGGL((xtwx_cols_kernel<false, XTE_KCHUNK>), dim3((p + JT - 1) / JT, (R + RT - 1) / RT, np),
                     dim3(256), 0, stream, B, n, p, U, ldu, (const int2 *)nullptr, R, planes);
  if (!out) return hipGetLastError();   // (planes only: the caller's next kernel adds them)
  const size_t cnt = (size_t)R * p;
  hipLaunchKernelGGL(plain_reduce_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, planes, np, R,
                     p, (const double *)nullptr, out);
  return hipGetLastError();
}

hipError_t launch_xtwx_cols_start(hipStream_t stream, const uint8_t *gamma, int chains, int p,
                                  int32_t *req, int32_t *count, uint32_t *valid, int words) {
  hipError_t err = hipMemsetAsync(count, 0, 4, stream);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(xtwx_cols_start_kernel, dim3(chains), dim3(256), 0, stream, gamma, p, (int2 *)req, count,
                     valid, words);
  return hipGetLastError();
}

hipError_t launch_square(hipStream_t stream, const double *x, size_t count, double *out) {
  hipLaunchKernelGGL(square_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, x, count, out);
  return hipGetLastError();
}

}  // namespace boom_amd
