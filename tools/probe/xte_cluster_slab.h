// EXPERIMENT, not part of the product (round 3): X'e of the bsts path by clusters of 4
// chains on the vector ALUs instead of the 16 x 16 f64-MFMA tiles of atb_mfma_kernel.
// Measured as a stand-alone kernel at T = 2000, p = 100, 1024 chains: 42.2 us against
// the MFMA kernel's 34.0 us -- 1024 workgroups each stream 25 columns of X plus four
// residual series = 475 MB of L2 -> CU traffic per round (11 TB/s), four times what the
// 16-chain tiles move.  Kept because it is the building block a round kernel without
// grid-wide dependencies would need.
// X'e of a CLUSTER of chains: the regression half of observe_data_given_state
// (StateSpaceRegressionModel.cpp:188-200; NeRegSuf::add_mixture_data,
// RegressionModel.cpp:356-370) after a state draw -- xty[c, j] = sum_t X[t, j] e_c[t],
// e_c the chain's residual series y - Z alpha (zero where unobserved).
//
// One workgroup of 128 threads computes a SLAB: up to XTE_CLUSTER chains x a range
// of columns.  A thread owns the time steps t = tid + 128 i; for T <= 2048 the
// cluster's residuals at its steps stay in registers (XTE_CLUSTER x 16 doubles) and
// every column of X is read ONCE per cluster, coalesced, the next column's loads in
// flight while the current one is multiplied.  The design matrix (T p doubles, L2
// resident) is therefore read once per XTE_CLUSTER chains.  No matrix cores: the
// product is chains x p x T = 0.4 GFLOP per round at T = 2000, p = 100, 1024 chains --
// what it costs is the traffic and the latency of the column reads, and a 16 x 16 MFMA
// tile would tie 16 chains to each other (see bsts_round_kernel: chains that depend on
// each other wait for each other's slowest sweep).
//
// The order of every sum is fixed by (tid, i) alone -- per thread over its steps in
// increasing t, a DPP tree over the wave, wave 0 + wave 1 -- so a value does not depend
// on which other chains share the cluster, nor on how the columns are cut into slabs:
// the stand-alone kernel and the fused round kernel produce the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace boom_amd {

enum : int { XTE_CLUSTER = 4, XTE_THREADS = 128, XTE_STEPS = 16, XTE_BATCH = 32 };
// LDS the slab routine needs (doubles): the two waves' partial sums of a batch of columns
enum : int { XTE_LDS_DOUBLES = 2 * XTE_BATCH * XTE_CLUSTER };

namespace xte_detail {
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// total of the wave in lane 63
__device__ __forceinline__ double wave_total(double x) {
  x += dpp<0x118, 0xf>(x);
  x += dpp<0x114, 0xf>(x);
  x += dpp<0x112, 0xf>(x);
  x += dpp<0x111, 0xf>(x);
  x += dpp<0x142, 0xa>(x);
  x += dpp<0x143, 0xc>(x);
  return x;
}
}  // namespace xte_detail

// E: residual series, chain c at E + c lde (T doubles); X: column j at X + j T;
// out[c ldc + j] for c in [c0, c0 + nc), j in [j0, j1).  Called by all XTE_THREADS
// threads of the workgroup (it contains barriers); s_part: XTE_LDS_DOUBLES doubles.
__device__ __forceinline__ void xte_slab(const double *__restrict__ X, int T,
                                         const double *__restrict__ E, int64_t lde, int c0, int nc,
                                         int j0, int j1, double *__restrict__ out, int ldc,
                                         double *s_part) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool one_chunk = T <= XTE_THREADS * XTE_STEPS;
  double er[XTE_CLUSTER][XTE_STEPS];
  if (one_chunk) {
#pragma unroll
    for (int c = 0; c < XTE_CLUSTER; ++c)
#pragma unroll
      for (int i = 0; i < XTE_STEPS; ++i) {
        const int t = tid + XTE_THREADS * i;
        er[c][i] = (c < nc && t < T) ? E[(int64_t)(c0 + c) * lde + t] : 0.0;
      }
  }
  for (int jb = j0; jb < j1; jb += XTE_BATCH) {
    const int je = (jb + XTE_BATCH < j1) ? jb + XTE_BATCH : j1;
    if (one_chunk) {
      double xn[XTE_STEPS];
      {
        const double *col = X + (int64_t)jb * T;
#pragma unroll
        for (int i = 0; i < XTE_STEPS; ++i) {
          const int t = tid + XTE_THREADS * i;
          xn[i] = col[t < T ? t : T - 1];   // (steps past T: any value, their residual is 0)
        }
      }
      for (int j = jb; j < je; ++j) {
        double xv[XTE_STEPS];
#pragma unroll
        for (int i = 0; i < XTE_STEPS; ++i) xv[i] = xn[i];
        if (j + 1 < je) {
          const double *col = X + (int64_t)(j + 1) * T;
#pragma unroll
          for (int i = 0; i < XTE_STEPS; ++i) {
            const int t = tid + XTE_THREADS * i;
            xn[i] = col[t < T ? t : T - 1];
          }
        }
        double acc[XTE_CLUSTER];
#pragma unroll
        for (int c = 0; c < XTE_CLUSTER; ++c) {
          acc[c] = 0.0;
#pragma unroll
          for (int i = 0; i < XTE_STEPS; ++i) acc[c] += xv[i] * er[c][i];
          acc[c] = xte_detail::wave_total(acc[c]);
        }
        if (lane == 63) {
#pragma unroll
          for (int c = 0; c < XTE_CLUSTER; ++c) s_part[(wave * XTE_BATCH + (j - jb)) * XTE_CLUSTER + c] = acc[c];
        }
      }
    } else {
      // long series: the same sums, the residuals read where they are used
      for (int j = jb; j < je; ++j) {
        const double *col = X + (int64_t)j * T;
        double acc[XTE_CLUSTER];
#pragma unroll
        for (int c = 0; c < XTE_CLUSTER; ++c) acc[c] = 0.0;
        for (int t = tid; t < T; t += XTE_THREADS) {
          const double x = col[t];
#pragma unroll
          for (int c = 0; c < XTE_CLUSTER; ++c)
            if (c < nc) acc[c] += x * E[(int64_t)(c0 + c) * lde + t];
        }
#pragma unroll
        for (int c = 0; c < XTE_CLUSTER; ++c) acc[c] = xte_detail::wave_total(acc[c]);
        if (lane == 63) {
#pragma unroll
          for (int c = 0; c < XTE_CLUSTER; ++c) s_part[(wave * XTE_BATCH + (j - jb)) * XTE_CLUSTER + c] = acc[c];
        }
      }
    }
    __syncthreads();
    {
      const int nb = je - jb;
      if (tid < nb * XTE_CLUSTER) {
        const int jj = tid / XTE_CLUSTER, c = tid % XTE_CLUSTER;
        if (c < nc)
          out[(int64_t)(c0 + c) * ldc + jb + jj] =
              s_part[(0 * XTE_BATCH + jj) * XTE_CLUSTER + c] + s_part[(1 * XTE_BATCH + jj) * XTE_CLUSTER + c];
      }
    }
    __syncthreads();
  }
}

}  // namespace boom_amd
