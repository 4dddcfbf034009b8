// Sufficient statistics of a regression, built once per data set:
//   XtX = X'X, Xty = X'y, yty = y'y, sum(y), column sums of X
// (NeRegSuf(X, y), Models/Glm/RegressionModel.cpp:309-328; Matrix::inner,
// LinAlg/Matrix.cpp:770-774).
//
// XtX is the one GEMM-shaped piece of the hot path (2 n p^2 flops, arithmetic
// intensity p/8 flop/byte), so it goes to the f64 matrix cores:
// v_mfma_f64_16x16x4_f64.  A 256-thread workgroup owns a 64 x 64 tile of XtX;
// each of its 4 wavefronts accumulates a 32 x 32 quadrant as 2 x 2 MFMA tiles.
// X is column-major n x p, so a column's rows are contiguous: the 32-row
// panels of the two 64-column strips are staged through LDS with coalesced
// 256-byte reads and a row stride of 34 doubles (== 2 mod 32 banks) so that
// the A/B fragment reads (16 columns x 4 rows per instruction) are
// conflict-free.
#include <hip/hip_runtime.h>
#include "ktimer.h"
#include <stdint.h>

namespace boom_amd {

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

// One 8-byte LDS read as its own instruction.  Left to the compiler, the two fragment reads of
// a matrix step (columns c and c + 16: a constant 4 KB apart) become ONE ds_read2st64_b64 --
// and the paired forms are serviced 16 lanes at a time against 32 banks
// (MI355X_MICROARCH.md, LDS table: ds_read2_b64 "two accesses, each 4 x 16 contiguous", 8
// cycles; ds_read_b64: 2 x 32 lanes against 64 banks, 2 cycles).  A fragment's 16 lanes of one
// k read 16 doubles of equal row parity, i.e. 8 of the 16 bank pairs: every paired read ran
// 2-way conflicted (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.49 in round 5's counters, on
// layouts that are conflict-free for ds_read_b64) at 16 LDS cycles for what two ds_read_b64
// deliver in 4 -- with eight wavefronts per CU the LDS was as busy as the matrix cores.  The
// compiler does not see these reads, so the waits are explicit too (lds_wait: the values are
// threaded through it, so no use can be scheduled above it).  Reads return in order.
template <int OFF>
__device__ __forceinline__ double lds_read_b64(uint32_t byte_addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
  return v;
}
// wait until at most N of the reads issued so far are outstanding
template <int N>
__device__ __forceinline__ void lds_wait(double &a, double &b, double &c, double &d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

constexpr int TILE = 64;    // XtX tile edge per workgroup
constexpr int KC = 32;      // rows of X per staging step
constexpr int LDP = KC + 2; // padded panel stride (doubles)

__global__ __launch_bounds__(256) void xtx_mfma_kernel(const double *__restrict__ X,
                                                       int64_t n, int p,
                                                       double *__restrict__ xtx, int edge) {
  // two panels in LDS: the next one is fetched (into registers, then LDS) while the
  // matrix cores work on the current one
  __shared__ double sA[2][TILE * LDP];
  __shared__ double sB[2][TILE * LDP];
  // Which tile: blocks b and b + 8 share an XCD (and its L2; dealt round-robin -- observed,
  // MI355X_MICROARCH.md "Workgroup dispatch, XCD placement": a speed assumption, never one
  // of correctness), and the 64 workgroups an XCD holds at a time should be tiles that
  // share panels of X.  So the j-th block of "XCD" x = b % 8 takes tile j % edge^2 of the
  // (j / edge^2) * 8 + x -th edge x edge SUPERTILE of the lower block triangle (edge 8 at
  // p = 4096: an XCD's resident set is one supertile -- 8 + 8 panels for 64 tiles), where
  // the plain (tj, ti) order gave an XCD every eighth tile of a tile row, all eight L2s
  // streaming every panel (52-103 GB of HBM traffic for 3.3 GB of X at n = 1e5, p = 4096:
  // profiles/r04_c4_pmc_traffic.json).  The host picks the edge so that every XCD has work.
  const int tiles = (p + TILE - 1) / TILE, S = (tiles + edge - 1) / edge;
  int ti, tj;
  {
    const unsigned b = blockIdx.x, x = b % 8u, j = b / 8u, e2 = (unsigned)(edge * edge);
    const unsigned q = (j / e2) * 8u + x, in = j % e2;
    if (q >= (unsigned)(S * (S + 1) / 2)) return;
    // supertile q of the lower triangle, row by row: (si, sj), sj <= si
    int si = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
    while ((unsigned)((si + 1) * (si + 2) / 2) <= q) ++si;
    while ((unsigned)(si * (si + 1) / 2) > q) --si;
    const int sj = (int)q - si * (si + 1) / 2;
    ti = si * edge + (int)(in / (unsigned)edge);
    tj = sj * edge + (int)(in % (unsigned)edge);
  }
  if (ti >= tiles || tj > ti) return;  // lower block-triangle only; mirrored on store
  const int I0 = ti * TILE, J0 = tj * TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;  // quadrant of the 64 x 64 tile

  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int fr = lane >> 4;   // k index inside an MFMA step (0..3)
  const int fc = lane & 15;   // row (A) / column (B) inside a 16-wide tile

  // split-K: slice z of gridDim.z takes rows [z, z + 1) * rows_per_slice (whole
  // staging steps) and writes its partial tile to plane z of the output
  const int64_t steps = (n + KC - 1) / KC;
  const int64_t per = (steps + gridDim.z - 1) / gridDim.z;
  const int64_t rbeg = (int64_t)blockIdx.z * per * KC;
  const int64_t rend = (rbeg + per * KC < n) ? rbeg + per * KC : n;
  xtx += (size_t)blockIdx.z * (size_t)p * (size_t)p;
  // a panel = 64 columns x 32 rows of each strip: 2048 doubles, 8 per thread
  const int prow = tid & 31, pcol0 = tid >> 5;   // columns pcol0 + 8 it
  double ra[8], rb[8];
  auto fetch = [&](int64_t r0) {
    const int64_t r = r0 + prow;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int col = pcol0 + 8 * it;
      const int ci = I0 + col, cj = J0 + col;
      ra[it] = (r < rend && ci < p) ? X[(int64_t)ci * n + r] : 0.0;
      rb[it] = (r < rend && cj < p) ? X[(int64_t)cj * n + r] : 0.0;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int col = pcol0 + 8 * it;
      sA[buf][col * LDP + prow] = ra[it];
      sB[buf][col * LDP + prow] = rb[it];
    }
  };
  if (rbeg < rend) {
    fetch(rbeg);
    stash(0);
  }
  __syncthreads();
  int cur = 0;
  for (int64_t r0 = rbeg; r0 < rend; r0 += KC) {
    const bool more = r0 + KC < rend;
    if (more) fetch(r0 + KC);
    {
      // fragment reads one matrix step ahead of their use, each an instruction of its own
      // (lds_read_b64); addresses: this lane's column of the wave's quadrant + k, in bytes
      const uint32_t pa = lds_addr(&sA[cur][(wi * 32 + fc) * LDP + fr]);
      const uint32_t pb = lds_addr(&sB[cur][(wj * 32 + fc) * LDP + fr]);
      constexpr int C16 = 16 * LDP * 8;   // sixteen columns on
      double a[2][2], b[2][2];
      a[0][0] = lds_read_b64<0>(pa); a[0][1] = lds_read_b64<C16>(pa);
      b[0][0] = lds_read_b64<0>(pb); b[0][1] = lds_read_b64<C16>(pb);
#pragma unroll
      for (int kk = 0; kk < KC / 4; ++kk) {
        const int s = kk & 1;
        if (kk + 1 < KC / 4) {
          // (offsets are immediates: kk is a compile-time constant once unrolled)
          switch (kk + 1) {
#define BA_RD(K) case K: a[s ^ 1][0] = lds_read_b64<K * 32>(pa); a[s ^ 1][1] = lds_read_b64<K * 32 + C16>(pa); \
                         b[s ^ 1][0] = lds_read_b64<K * 32>(pb); b[s ^ 1][1] = lds_read_b64<K * 32 + C16>(pb); break;
            BA_RD(1) BA_RD(2) BA_RD(3) BA_RD(4) BA_RD(5) BA_RD(6) BA_RD(7)
#undef BA_RD
          }
          lds_wait<4>(a[s][0], a[s][1], b[s][0], b[s][1]);
        } else {
          lds_wait<0>(a[s][0], a[s][1], b[s][0], b[s][1]);
        }
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)
            acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][ta], b[s][tb], acc[ta][tb], 0, 0, 0);
      }
    }
    if (more) stash(cur ^ 1);   // (the other buffer was last read before the previous barrier)
    __syncthreads();
    cur ^= 1;
  }
  // D layout of v_mfma_f64_16x16x4_f64: register q of lane l holds
  // row (l >> 4) + 4 q, column l & 15.
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = I0 + wi * 32 + ta * 16 + (lane >> 4) + 4 * q;
        const int j = J0 + wj * 32 + tb * 16 + (lane & 15);
        if (i < p && j < p) {
          const double v = acc[ta][tb][q];
          xtx[(int64_t)j * p + i] = v;
          xtx[(int64_t)i * p + j] = v;
        }
      }
}

// The same product with the panels brought in by LDS-DMA (global_load_lds_dwordx4: global ->
// LDS with no vector register in between, 1 KB per wave instruction) instead of 16 loads + 16
// ds_write_b64 per thread and panel ("try it and report").  A DMA writes its
// 64 lanes' 16 bytes one after the other, so the LDS image cannot be padded per column; the
// bank-conflict-free fragment reads come from an XOR swizzle instead, put on the SOURCE side
// (cdna_hip_programming.md 5, rule 21): granule q (two rows) of column c sits at
// c * 16 + (q ^ (c & 15)), i.e. the lane that writes that place loads rows 2 (q ^ (c & 15)) + {0, 1}
// of the column.  Same products in the same order as xtx_mfma_kernel: bitwise the same X'X.
// Needs n even and X 16-byte aligned (a granule is 16 bytes of one column); whole 32-row
// steps by DMA, a last partial step through registers (zero fill).
template <int KX>
__device__ __forceinline__ int panel_at(int c, int k) { return c * KX + ((((k >> 1) ^ (c & (KX / 2 - 1))) << 1) | (k & 1)); }

template <int KX>
__global__ __launch_bounds__(256) void xtx_mfma_glds_kernel(const double *__restrict__ X,
                                                            int64_t n, int p,
                                                            double *__restrict__ xtx, int edge) {
  extern __shared__ __attribute__((aligned(16))) double s_dyn[];
  double (*sA)[TILE * KX] = reinterpret_cast<double (*)[TILE * KX]>(s_dyn);
  double (*sB)[TILE * KX] = reinterpret_cast<double (*)[TILE * KX]>(s_dyn + 2 * TILE * KX);
  const int tiles = (p + TILE - 1) / TILE, S = (tiles + edge - 1) / edge;
  int ti, tj;
  {
    const unsigned b = blockIdx.x, x = b % 8u, j = b / 8u, e2 = (unsigned)(edge * edge);
    const unsigned q = (j / e2) * 8u + x, in = j % e2;
    if (q >= (unsigned)(S * (S + 1) / 2)) return;
    // supertile q of the lower triangle, row by row: (si, sj), sj <= si
    int si = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
    while ((unsigned)((si + 1) * (si + 2) / 2) <= q) ++si;
    while ((unsigned)(si * (si + 1) / 2) > q) --si;
    const int sj = (int)q - si * (si + 1) / 2;
    ti = si * edge + (int)(in / (unsigned)edge);
    tj = sj * edge + (int)(in % (unsigned)edge);
  }
  if (ti >= tiles || tj > ti) return;
  const int I0 = ti * TILE, J0 = tj * TILE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  double4_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int fr = lane >> 4, fc = lane & 15;
  // (the row slices are the register-staged kernel's: whole 32-row steps)
  const int64_t steps = (n + KC - 1) / KC;
  const int64_t per = (steps + gridDim.z - 1) / gridDim.z;
  const int64_t rbeg = (int64_t)blockIdx.z * per * KC;
  const int64_t rend = (rbeg + per * KC < n) ? rbeg + per * KC : n;
  xtx += (size_t)blockIdx.z * (size_t)p * (size_t)p;
  // (columns past p are never written: they stay zero)
  for (int i = tid; i < 4 * TILE * KX; i += 256) s_dyn[i] = 0.0;
  __syncthreads();
  typedef __attribute__((address_space(1))) const void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  // DMA instruction I of a strip (16 of them, four per wave) fills granules [64 I, 64 I + 64):
  // columns 4 I .. 4 I + 3; lane l: column 4 I + (l >> 4), place l & 15
  auto stage = [&](int buf, int64_t r0) {
    if (r0 + KX <= rend) {
      constexpr int GPC = KX / 2;              // granules per column
      constexpr int CPI = 64 / GPC;            // columns per DMA instruction
#pragma unroll
      for (int u = 0; u < GPC / 4; ++u) {
        const int I = 4 * u + wave, c = CPI * I + lane / GPC, g = (lane & (GPC - 1)) ^ (c & (GPC - 1));
        if (I0 + c < p)
          __builtin_amdgcn_global_load_lds((gptr_t)(X + (int64_t)(I0 + c) * n + r0 + 2 * g), (lptr_t)(&sA[buf][128 * I]), 16, 0, 0);
        if (J0 + c < p)
          __builtin_amdgcn_global_load_lds((gptr_t)(X + (int64_t)(J0 + c) * n + r0 + 2 * g), (lptr_t)(&sB[buf][128 * I]), 16, 0, 0);
      }
    } else {
      for (int e = tid; e < TILE * KX; e += 256) {
        const int col = e / KX, prow = e % KX;
        const int64_t r = r0 + prow;
        sA[buf][panel_at<KX>(col, prow)] = (r < rend && I0 + col < p) ? X[(int64_t)(I0 + col) * n + r] : 0.0;
        sB[buf][panel_at<KX>(col, prow)] = (r < rend && J0 + col < p) ? X[(int64_t)(J0 + col) * n + r] : 0.0;
      }
    }
  };
  if (rbeg < rend) stage(0, rbeg);
  __syncthreads();
  int cur = 0;
  for (int64_t r0 = rbeg; r0 < rend; r0 += KX) {
    if (r0 + KX < rend) stage(cur ^ 1, r0 + KX);   // (the other buffer was last read before the previous barrier)
    {
      // fragment reads one matrix step ahead of their use, each an instruction of its own
      // (lds_read_b64).  panel_at(c, 4 kk + fr) = c KX + (((2 kk + (fr >> 1)) ^ (c & 15)) << 1 | (fr & 1)):
      // the columns c and c + 16 of a fragment pair share c & 15 = fc, so step kk's place
      // inside a column, sw(kk), is the same for all four reads
      const uint32_t pa = lds_addr(&sA[cur][(wi * 32 + fc) * KX + (fr & 1)]);
      const uint32_t pb = lds_addr(&sB[cur][(wj * 32 + fc) * KX + (fr & 1)]);
      constexpr int C16 = 16 * KX * 8;   // sixteen columns on
      const uint32_t v = (uint32_t)((fr >> 1) ^ (fc & (KX / 2 - 1)));
      auto sw = [&](int kk) { return (((uint32_t)(2 * kk) ^ v) << 4); };
      double a[2][2], b[2][2];
      {
        const uint32_t o = sw(0);
        a[0][0] = lds_read_b64<0>(pa + o); a[0][1] = lds_read_b64<C16>(pa + o);
        b[0][0] = lds_read_b64<0>(pb + o); b[0][1] = lds_read_b64<C16>(pb + o);
      }
#pragma unroll
      for (int kk = 0; kk < KX / 4; ++kk) {
        const int s = kk & 1;
        if (kk + 1 < KX / 4) {
          const uint32_t o = sw(kk + 1);
          a[s ^ 1][0] = lds_read_b64<0>(pa + o); a[s ^ 1][1] = lds_read_b64<C16>(pa + o);
          b[s ^ 1][0] = lds_read_b64<0>(pb + o); b[s ^ 1][1] = lds_read_b64<C16>(pb + o);
          lds_wait<4>(a[s][0], a[s][1], b[s][0], b[s][1]);
        } else {
          lds_wait<0>(a[s][0], a[s][1], b[s][0], b[s][1]);
        }
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int tb = 0; tb < 2; ++tb)
            acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s][ta], b[s][tb], acc[ta][tb], 0, 0, 0);
      }
    }
    __syncthreads();   // (waits for the DMAs in flight too: the panel is there)
    cur ^= 1;
  }
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = I0 + wi * 32 + ta * 16 + (lane >> 4) + 4 * q;
        const int j = J0 + wj * 32 + tb * 16 + (lane & 15);
        if (i < p && j < p) {
          const double v = acc[ta][tb][q];
          xtx[(int64_t)j * p + i] = v;
          xtx[(int64_t)i * p + j] = v;
        }
      }
}

// sum of the split-K planes in plane order (bitwise reproducible)
__global__ __launch_bounds__(256) void plane_sum_kernel(const double *__restrict__ planes,
                                                        int nplanes, size_t count,
                                                        double *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  double a = planes[i];
  for (int z = 1; z < nplanes; ++z) a += planes[(size_t)z * count + i];
  out[i] = a;
}

// block j < p: sum_r X[r,j], sum_r X[r,j] y[r];  block p: sum y, sum y^2.
// Fixed-shape tree reduction: bitwise reproducible run to run.
__global__ __launch_bounds__(256) void col_reduce_kernel(const double *__restrict__ X,
                                                         const double *__restrict__ y,
                                                         int64_t n, int p,
                                                         double *__restrict__ xty,
                                                         double *__restrict__ xsum,
                                                         double *__restrict__ scalars) {
  __shared__ double s0[256], s1[256];
  const int j = blockIdx.x, tid = threadIdx.x;
  double a = 0.0, b = 0.0;
  if (j < p) {
    const double *col = X + (int64_t)j * n;
    for (int64_t r = tid; r < n; r += 256) {
      const double x = col[r];
      a += x;
      b += x * y[r];
    }
  } else {
    for (int64_t r = tid; r < n; r += 256) {
      const double v = y[r];
      a += v;
      b += v * v;
    }
  }
  s0[tid] = a;
  s1[tid] = b;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (tid < w) {
      s0[tid] += s0[tid + w];
      s1[tid] += s1[tid + w];
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (j < p) {
      xsum[j] = s0[0];
      xty[j] = s1[0];
    } else {
      scalars[0] = s1[0];  // yty
      scalars[1] = s0[0];  // sum y
    }
  }
}

// C = A'B for two matrices stored as K-contiguous columns (A: K x M, column c at
// A + c lda; B: K x N, column j at B + j ldb), C row-major M x N.  This is the
// regression half of observe_data_given_state for all chains at once
// (StateSpaceRegressionModel.cpp:188-200): A = the chains' residual series
// (one column per chain), B = the design matrix, C[chain, j] = x_j'e_chain --
// the design matrix is read once per 16 chains instead of once per chain.
// Four wavefronts per 16 x 16 tile of C; MFMA step s of an 8-row slab uses rows
// k0 + 2 (l >> 4) + s, so that a lane's two operands per matrix are adjacent in
// memory.  The order of the k summation is fixed: bitwise reproducible.
__global__ __launch_bounds__(256) void atb_mfma_kernel(const double *__restrict__ A,
                                                       int64_t lda, int M,
                                                       const double *__restrict__ B,
                                                       int64_t ldb, int N, int K,
                                                       double *__restrict__ C, int ldc) {
  // four wavefronts per tile, each a quarter of the k range (whole 8-row
  // slabs); the partial tiles are added in wave order
  __shared__ double part[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  const int fc = lane & 15, fr = lane >> 4;
  const bool ia = (i0 + fc) < M, jb = (j0 + fc) < N;
  const double *pa = A + (int64_t)(ia ? i0 + fc : 0) * lda + 2 * fr;
  const double *pb = B + (int64_t)(jb ? j0 + fc : 0) * ldb + 2 * fr;
  const int slabs = (K + 7) / 8, per = (slabs + 3) / 4;
  const int kbeg = wave * per * 8;
  const int kend = (wave + 1) * per * 8 < K ? (wave + 1) * per * 8 : K;
  double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
  constexpr int UN = 4;  // slabs in flight
  int k0 = kbeg;
  for (; k0 + 8 * UN <= kend; k0 += 8 * UN) {
    double a0[UN], a1[UN], b0[UN], b1[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      a0[u] = pa[k0 + 8 * u];
      a1[u] = pa[k0 + 8 * u + 1];
      b0[u] = pb[k0 + 8 * u];
      b1[u] = pb[k0 + 8 * u + 1];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ia ? a0[u] : 0.0, jb ? b0[u] : 0.0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ia ? a1[u] : 0.0, jb ? b1[u] : 0.0, acc, 0, 0, 0);
    }
  }
  for (; k0 < kend; k0 += 8) {
    const int ka = k0 + 2 * fr;
    const double a0 = (ia && ka < kend) ? pa[k0] : 0.0, a1 = (ia && ka + 1 < kend) ? pa[k0 + 1] : 0.0;
    const double b0 = (jb && ka < kend) ? pb[k0] : 0.0, b1 = (jb && ka + 1 < kend) ? pb[k0 + 1] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) part[wave - 1][q][lane] = acc[q];
  }
  __syncthreads();
  if (wave == 0) {
    // register q of lane l holds row (l >> 4) + 4 q, column l & 15
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double v = ((acc[q] + part[0][q][lane]) + part[1][q][lane]) + part[2][q][lane];
      const int i = i0 + fr + 4 * q, j = j0 + fc;
      if (i < M && j < N) C[(int64_t)i * ldc + j] = v;
    }
  }
}

}  // namespace

hipError_t launch_atb_mfma(hipStream_t stream, const double *A, int64_t lda, int M,
                           const double *B, int64_t ldb, int N, int K, double *C, int ldc) {
  KtScope kt(stream, KT_XTE_GEMM);
  hipLaunchKernelGGL(atb_mfma_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, stream,
                     A, lda, M, B, ldb, N, K, C, ldc);
  return hipGetLastError();
}

// how many row slices the XtX build wants for this shape: few tiles (p = 512:
// 36 of them for 256 CUs) -> split the rows as well, the partial products
// summed in a fixed order by a second kernel.  The caller provides the
// workspace of suf_row_slices(n, p) * p * p doubles (none for 1 slice).
int suf_row_slices(int64_t n, int p) {
  const int tiles = (p + TILE - 1) / TILE;
  const int lower = tiles * (tiles + 1) / 2;
  int ksplit = (512 + lower - 1) / lower;
  const int64_t steps = (n + KC - 1) / KC;
  if (ksplit > 16) ksplit = 16;
  if (ksplit > steps) ksplit = (int)steps;
  while (ksplit > 1 && (size_t)ksplit * p * p * 8 > ((size_t)256 << 20)) --ksplit;
  return ksplit < 1 ? 1 : ksplit;
}

// supertile edge (see xtx_mfma_kernel): the largest of 8, 4, 2, 1 that leaves every XCD at
// least four supertiles of the lower triangle
static int suf_edge(int tiles) {
  for (int e = 8; e > 1; e >>= 1) {
    const int S = (tiles + e - 1) / e;
    if (S * (S + 1) / 2 >= 32) return e;
  }
  return 1;
}
// 8 x edge^2 blocks for every eight supertiles of the lower triangle
static unsigned suf_grid_blocks(int tiles, int edge) {
  const int S = (tiles + edge - 1) / edge, lower = S * (S + 1) / 2;
  return (unsigned)(((lower + 7) / 8) * 8 * edge * edge);
}

// (A/B builds: -DBA_SUF_NO_DMA keeps the register-staged kernel everywhere)
#ifdef BA_SUF_NO_DMA
static const bool g_suf_dma = false;
#else
static const bool g_suf_dma = true;
#endif

int launch_suf_from_xy(hipStream_t stream, int64_t n, int p, const double *X,
                       const double *y, double *xtx, double *xty,
                       double *scalars, double *xsum, double *planes) {
  const int tiles = (p + TILE - 1) / TILE;
  const int ksplit = planes ? suf_row_slices(n, p) : 1;
  const int edge = suf_edge(tiles);
  KtScope kt(stream, KT_SUF);
  // (panels by LDS-DMA where a 16-byte granule of a column is addressable: see xtx_mfma_glds_kernel)
  const bool dma = g_suf_dma && (n % 2 == 0) && (reinterpret_cast<uintptr_t>(X) % 16 == 0);
  // (32-row panels, two workgroups to a CU: 43.5 ms at n = 1e5, p = 4096 against 44.9 through
  // registers; 64-row panels -- half the barriers, 128 KB of LDS, one workgroup per CU: 48.8)
  constexpr int KX = 32;
  auto kern = dma ? xtx_mfma_glds_kernel<KX> : xtx_mfma_kernel;
  const size_t lds = dma ? (size_t)4 * TILE * KX * 8 : 0;
  if (dma && lds > 65536) {
    if (hipFuncSetAttribute((const void *)xtx_mfma_glds_kernel<KX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
  }
  if (ksplit <= 1) {
    hipLaunchKernelGGL(kern, dim3(suf_grid_blocks(tiles, edge), 1, 1), dim3(256), lds, stream, X, n, p, xtx, edge);
  } else {
    const size_t count = (size_t)p * p;
    hipLaunchKernelGGL(kern, dim3(suf_grid_blocks(tiles, edge), 1, ksplit), dim3(256), lds, stream, X, n, p, planes, edge);
    hipLaunchKernelGGL(plane_sum_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream,
                       planes, ksplit, count, xtx);
  }
  hipLaunchKernelGGL(col_reduce_kernel, dim3(p + 1), dim3(256), 0, stream, X, y,
                     n, p, xty, xsum, scalars);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

}  // namespace boom_amd
