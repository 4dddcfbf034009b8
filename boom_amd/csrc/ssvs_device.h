// Device-side building blocks shared by the SSVS kernels (ssvs_kernel.hip: the
// LDS-resident sweep for models of up to 64 variables; ssvs_big_kernel.hip: the
// HBM-resident sweep for larger models): cross-lane helpers, the Cholesky /
// proposal primitives of the LDS kernel, the parallel Fisher-Yates shuffle, the
// table walk, the stream-ordered normal draws and the swap proposal.
#pragma once
#include <hip/hip_runtime.h>

#include "device_rng.h"
#include "diag.h"
#include "ssvs_params.h"

namespace boom_amd {

namespace {

constexpr int WAVE = 64;
#define BA_INF (__builtin_inf())

// (the diagnostic builds' cycle stamps -- STAMP, SUBSTAMP, TSTAMP, HSTAMP -- are diag.h's)

// ---- address spaces ---------------------------------------------------------
// LDS pointers are typed as such so that every access is a ds_* instruction no
// matter how the compiler inlines (a generic pointer would become flat_load).
#define AS_LDS __attribute__((address_space(3)))
typedef AS_LDS double lds_f64;
typedef AS_LDS uint16_t lds_u16;
typedef AS_LDS uint32_t lds_u32;
typedef AS_LDS uint8_t lds_u8;
template <class T>
__device__ __forceinline__ AS_LDS T *to_lds(unsigned char *generic) {
  return (AS_LDS T *)(uintptr_t)generic;
}
// Wave-uniform model data is read back through the scalar cache: a pointer in
// the constant address space makes every (uniform-address) load an s_load, so
// factor elements arrive in SGPRs and feed v_fma_f64 directly.  The data ARE
// rewritten by this wavefront (publish_model); the pointer is re-derived
// through an opaque asm after each rewrite so that no load can move above it.
#define AS_CONST __attribute__((address_space(4)))
typedef AS_CONST const double c_f64;
typedef AS_CONST const int c_i32;

// LDS hand-off between the lanes of ONE wavefront (the wave-cooperative
// routines below are run by a single wave of the workgroup): DS operations of a
// wave complete in order, so only the compiler has to be told.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- cross-lane helpers (DPP / readlane: no LDS round trip) -----------------
// lanes whose DPP source is outside their row (or whose row is masked off)
// receive `fill`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x, double fill) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned long long f = __builtin_bit_cast(unsigned long long, fill);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)f, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(f >> 32), (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// value of lane `src` (wave-uniform index)
__device__ __forceinline__ double bcast_u(double x, int src) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int bcast_u(int x, int src) {
  return __builtin_amdgcn_readlane(x, src);
}
// A wave-uniform value the compiler cannot see to be uniform (it came out of
// vector arithmetic) is moved to scalar registers: it then costs no vector
// register while it waits for its next use, and if it has to be spilled it goes
// to a lane of a vector register, not to scratch memory.
__device__ __forceinline__ double uni(double x) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint64_t uni(uint64_t u) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return ((uint64_t)hi << 32) | lo;
}
// Reductions over the wave, result in every lane.  row_shr 8/4/2/1 leaves each
// row's total in its lane 15; row_bcast:15 / row_bcast:31 carry it to lane 63.
__device__ __forceinline__ double wave_sum(double x) {
  x += dpp_f64<0x118, 0xf>(x, 0.0);
  x += dpp_f64<0x114, 0xf>(x, 0.0);
  x += dpp_f64<0x112, 0xf>(x, 0.0);
  x += dpp_f64<0x111, 0xf>(x, 0.0);
  x += dpp_f64<0x142, 0xa>(x, 0.0);
  x += dpp_f64<0x143, 0xc>(x, 0.0);
  return bcast_u(x, 63);
}
__device__ __forceinline__ double wave_min(double x) {
  x = fmin(x, dpp_f64<0x118, 0xf>(x, x));
  x = fmin(x, dpp_f64<0x114, 0xf>(x, x));
  x = fmin(x, dpp_f64<0x112, 0xf>(x, x));
  x = fmin(x, dpp_f64<0x111, 0xf>(x, x));
  x = fmin(x, dpp_f64<0x142, 0xa>(x, x));
  x = fmin(x, dpp_f64<0x143, 0xc>(x, x));
  return bcast_u(x, 63);
}

// offset (in doubles) of element (m, n), n <= m, in the block-packed factor
__device__ __forceinline__ int bidx(int m, int n) {
  const int I = m >> 3, J = n >> 3;
  return ((I * (I + 1)) / 2 + J) * 64 + (m & 7) * 8 + (n & 7);
}

}  // namespace
}  // namespace boom_amd
#include "ssvs_fill_mfma.h"
namespace boom_amd {
namespace {

// wave-uniform description of the current model
struct Model {
  double logp;  // log_model_prob(gamma)
  double lp;    // log prior of gamma
  double ldv;   // log|V_g|
  double lda;   // log|A_g|  (ldoi)
  double Q;     // |w|^2
  double c;     // b_g' A_g b_g
  double SS;
  bool pd;      // V_g positive definite
  int bad;      // ChainStatus raised while evaluating
};

struct Chain {
  int lane, p, k;
  // LDS
  lds_f64 *Lv, *La, *rdv, *rda, *w, *bg;
  lds_u16 *g, *perm, *perm_alt, *oth, *pred;
  lds_u32 *last;
  lds_u8 *gam, *gam0, *nbr;
  // HBM copy of the model read through the scalar cache (see publish_model)
  double *tab_lp;     // table of exp(log_model_prob(gamma ^ {j}) - log_model_prob(gamma)), j = 0..p-1 (HBM)
  uint8_t *tab_kind;  // 0 / STOP_SLOW / STOP_BAD per j
  double *sc_store;   // global pointer used for the stores
  c_f64 *sc;          // the same memory, constant address space
  // this chain's sufficient statistics
  const double *xty;
  double DF;    // n + prior_df
  double ss0q;  // prior_ss + yty
  // SSVS_MODE_SIGMA_GIVEN (SpikeSlabSampler): scale factors that turn the
  // shared matrices into this chain's, V_c = sv V, A_c = sa A, xty_c = sx xty
  // (all 1 for BregVsSampler, where sigma^2 is integrated out)
  int mode;
  double sv, sa, sx;
};

__device__ __forceinline__ void bind_lds(Chain &ch, unsigned char *smem,
                                         const SsvsLds &lay) {
  ch.Lv = to_lds<double>(smem + lay.Lv);
  ch.La = to_lds<double>(smem + lay.La);
  ch.rdv = to_lds<double>(smem + lay.rdv);
  ch.rda = to_lds<double>(smem + lay.rda);
  ch.w = to_lds<double>(smem + lay.w);
  ch.bg = to_lds<double>(smem + lay.bg);
  ch.g = to_lds<uint16_t>(smem + lay.g);
  ch.perm = to_lds<uint16_t>(smem + lay.perm0);
  ch.perm_alt = to_lds<uint16_t>(smem + lay.perm1);
  ch.oth = to_lds<uint16_t>(smem + lay.oth);
  ch.last = to_lds<uint32_t>(smem + lay.last);
  ch.pred = to_lds<uint16_t>(smem + lay.pred);
  ch.gam = to_lds<uint8_t>(smem + lay.gam);
  ch.gam0 = to_lds<uint8_t>(smem + lay.gam0);
  ch.nbr = to_lds<uint8_t>(smem + lay.nbr);
}

// In-place Cholesky of block-packed lower triangles, lane i owns row i
// (k <= 64).  Left-looking by column: the subtraction order for every entry is
// that of Eigen's unblocked LLT (Eigen/src/Cholesky/LLT.h:313-335) which the
// reference uses (LinAlg/Cholesky.cpp:33-58); a factorisation stops at its first
// non-positive pivot.  logdet = 2 * sum log L_jj.
// The two factorisations of a rebuild (A_g and V_g) side by side: the same
// column-by-column arithmetic as chol_blocks for each, but the two dependent
// chains (dot product, sqrt, divide) interleave, which is what a single
// wavefront per SIMD needs.  A failed factorisation stops advancing (its
// remaining columns are never used); the other one carries on.
__device__ __forceinline__ void chol_blocks2(const Chain &ch, lds_f64 *LA, lds_f64 *rdA,
                                             lds_f64 *LV, lds_f64 *rdV, bool *okA,
                                             bool *okV, double *ldA, double *ldV) {
  const int k = ch.k, i = ch.lane;
  bool oa = true, ov = true;
  for (int j = 0; j < k && (oa || ov); ++j) {
    const bool mine = (i >= j) && (i < k);
    const int ii = mine ? i : j;  // lanes without a row read row j (discarded)
    const int jb = j >> 3;
    const int offi = ((ii >> 3) * ((ii >> 3) + 1) / 2) * 64 + (ii & 7) * 8;
    const int offj = (jb * (jb + 1) / 2) * 64 + (j & 7) * 8;
    double sa = LA[bidx(ii, j)], sv = LV[bidx(ii, j)];
    for (int nb = 0; nb < jb; ++nb) {
      // (a block's row is 64 bytes, 16-byte aligned -- the factors start on 16-byte
      // boundaries of the LDS layout --: sixteen 16-byte reads where 32 8-byte ones were two
      // thirds of the loop's instructions; the same products subtracted in the same order)
      typedef double d2_t __attribute__((ext_vector_type(2)));
      const AS_LDS d2_t *pa = (const AS_LDS d2_t *)(LA + offi + nb * 64), *pb = (const AS_LDS d2_t *)(LA + offj + nb * 64);
      const AS_LDS d2_t *pc = (const AS_LDS d2_t *)(LV + offi + nb * 64), *pd = (const AS_LDS d2_t *)(LV + offj + nb * 64);
      d2_t a[4], b[4], c[4], d[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { a[t] = pa[t]; b[t] = pb[t]; c[t] = pc[t]; d[t] = pd[t]; }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        sa -= a[t].x * b[t].x;
        sv -= c[t].x * d[t].x;
        sa -= a[t].y * b[t].y;
        sv -= c[t].y * d[t].y;
      }
    }
    {
      const int rem = j & 7;
      double a[8], b[8], c[8], d[8];
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        a[t] = (t < rem) ? LA[offi + jb * 64 + t] : 0.0;
        b[t] = (t < rem) ? LA[offj + jb * 64 + t] : 0.0;
        c[t] = (t < rem) ? LV[offi + jb * 64 + t] : 0.0;
        d[t] = (t < rem) ? LV[offj + jb * 64 + t] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < 7; ++t)
        if (t < rem) {
          sa -= a[t] * b[t];
          sv -= c[t] * d[t];
        }
    }
    const double da = bcast_u(sa, j), dv = bcast_u(sv, j);
    if (oa && !(da > 0.0)) oa = false;
    if (ov && !(dv > 0.0)) ov = false;
    const double sda = sqrt(da), sdv = sqrt(dv);
    if (oa) {
      if (i == j) {
        LA[bidx(j, j)] = sda;
        rdA[j] = 1.0 / sda;
      } else if (mine) {
        LA[bidx(i, j)] = sa / sda;
      }
    }
    if (ov) {
      if (i == j) {
        LV[bidx(j, j)] = sdv;
        rdV[j] = 1.0 / sdv;
      } else if (mine) {
        LV[bidx(i, j)] = sv / sdv;
      }
    }
    wave_sync();
  }
  // sum_j log L_jj in column order, the logarithms taken side by side
  double la = 0.0, lv = 0.0;
  const double lga = (oa && i < k) ? log(LA[bidx(i, i)]) : 0.0;
  const double lgv = (ov && i < k) ? log(LV[bidx(i, i)]) : 0.0;
  for (int j = 0; j < k; ++j) {
    la += bcast_u(lga, j);
    lv += bcast_u(lgv, j);
  }
  *okA = oa; *okV = ov;
  *ldA = 2.0 * la; *ldV = 2.0 * lv;
}

// Rebuild everything about the current model gamma (sorted index list g in
// LDS) from scratch: BregVsSampler::set_reg_post_params + log_model_prob.
// Inlined at its (single) call site in each kernel.
// REUSE: the factors (and log prior, log determinants) of this very model are
// already in LDS / M -- restored from the chain's block at the start of a
// launch -- and only what depends on the sufficient statistics X'y, y'y is
// recomputed (state-space path: they move every sweep).
template <bool REUSE>
__device__ __forceinline__ void refactor(const SsvsParams &P, Chain &ch, Model &M, StampCtx &sx) {
  const int lane = ch.lane, p = ch.p, k = ch.k;
  M.bad = 0;
  M.pd = true;
  double lp;
  const double ldv_in = M.ldv, lda_in = M.lda;
  // Everything this function reads from global memory depends on gamma and the list g
  // alone, so all of it is asked for before anything is used: the prior's terms, V_g and
  // A_g (lower triangles, rows padded with zeros to a multiple of 8; element e <-> (m, n),
  // n <= m), b_g and X'y_g.  As written before -- prior sum, then the gathers, then b, then
  // X'y -- a rebuild was four round trips in a row, and a chain that accepts a flip in a
  // one-sweep launch is the chain the whole launch waits for.
  wave_sync();
  const int kpad = (k + 7) & ~7;
  const int nelem = REUSE ? 0 : kpad * (kpad + 1) / 2;
  const int gm = (lane < k) ? ch.g[lane] : 0;
  const double bm = (lane < k) ? P.b[gm] : 0.0;
  const double xg = (lane < k) ? ch.xty[gm] : 0.0;
  // (the first 64 elements of the triangles -- all of them up to k = 8 -- and the first 128
  // prior terms go out here; what is left of either takes the loops below)
  int m0 = 0, n0 = 0;
  double v0 = 0.0, a0 = 0.0;
  if (!REUSE && lane < nelem) {
    int m = (int)((sqrtf(8.0f * (float)lane + 1.0f) - 1.0f) * 0.5f);
    while ((m + 1) * (m + 2) / 2 <= lane) ++m;
    while (m * (m + 1) / 2 > lane) --m;
    m0 = m;
    n0 = lane - m * (m + 1) / 2;
    if (m < k) {
      const size_t o = (size_t)ch.g[m] * p + ch.g[n0];
      v0 = P.V[o] * ch.sv;
      a0 = P.A[o] * ch.sa;
    }
  }
  if (REUSE) {
    lp = M.lp;
  } else {
    // VariableSelectionPrior::logp (VariableSelectionPrior.cpp:271-285)
    double t0 = 0.0, t1 = 0.0;
    if (lane < p) t0 = ch.gam[lane] ? P.l1[lane] : P.l0[lane];
    if (lane + WAVE < p) t1 = ch.gam[lane + WAVE] ? P.l1[lane + WAVE] : P.l0[lane + WAVE];
    double part = 0.0;
    if (lane < p) part += t0;
    if (lane + WAVE < p) part += t1;
    for (int j = lane + 2 * WAVE; j < p; j += WAVE) part += ch.gam[j] ? P.l1[j] : P.l0[j];
    lp = wave_sum(part);
    if (P.max_model_size >= 0 && k > P.max_model_size) lp = -BA_INF;
    if (!(lp > -BA_INF)) lp = -BA_INF;  // also catches NaN from inf - inf
  }
  M.lp = lp;
  M.ldv = M.lda = M.Q = M.c = 0.0;
  M.SS = ch.ss0q;
  if (k == 0) {
    // empty model: BregVsSampler.cpp:217-227 / SpikeSlabSampler.cpp:173-183
    M.logp = ch.mode ? lp : lp - (0.5 * ch.DF - 1.0) * log(ch.ss0q);
    return;
  }
  if (lp == -BA_INF) {
    M.logp = -BA_INF;
    M.pd = false;
    return;
  }
  if (!REUSE && lane < nelem) {
    ch.Lv[bidx(m0, n0)] = v0;
    ch.La[bidx(m0, n0)] = a0;
  }
  for (int e = lane + WAVE; e < nelem; e += WAVE) {
    int m = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    while ((m + 1) * (m + 2) / 2 <= e) ++m;
    while (m * (m + 1) / 2 > e) --m;
    const int n = e - m * (m + 1) / 2;
    double v = 0.0, a = 0.0;
    if (m < k) {
      const size_t o = (size_t)ch.g[m] * p + ch.g[n];
      v = P.V[o] * ch.sv;
      a = P.A[o] * ch.sa;
    }
    ch.Lv[bidx(m, n)] = v;
    ch.La[bidx(m, n)] = a;
  }
  if (!REUSE && lane < kpad) {
    ch.bg[lane] = bm;
    if (lane >= k) {
      ch.rdv[lane] = 0.0;
      ch.rda[lane] = 0.0;
      ch.w[lane] = 0.0;
    }
  }
  // r = A_g b_g + xty_g ; c = b_g' A_g b_g   (only non-zero prior means cost)
  double ab = 0.0;
  unsigned long long nz = __ballot(bm != 0.0);
  while (nz) {
    const int n = __ffsll((long long)nz) - 1;
    nz &= nz - 1;
    const double bn = bcast_u(bm, n);
    const int gn = bcast_u(gm, n);
    if (lane < k) ab += (P.A[(size_t)gm * p + gn] * ch.sa) * bn;
  }
  const double r = (lane < k) ? ab + xg * ch.sx : 0.0;
  M.c = wave_sum(lane < k ? bm * ab : 0.0);
  wave_sync();
  bool okv = true, oka = true;
  if (REUSE) {
    M.lda = lda_in;
    M.ldv = ldv_in;
  } else {
    chol_blocks2(ch, ch.La, ch.rda, ch.Lv, ch.rdv, &oka, &okv, &M.lda, &M.ldv);
  }
  if (!okv) {
    M.pd = false;
    M.logp = -BA_INF;
    return;
  }
  // w = L_V^{-1} r, lane m ends up holding w_m
  double x = r;
  const double rdm = (lane < k) ? ch.rdv[lane] : 0.0;
  // (eight columns of the lane's row at a time -- 64 contiguous bytes of a block --, then
  // eight steps from registers: read inside the dependent loop, every step waited for LDS)
#pragma nounroll
  for (int jb = 0; jb * 8 < k; ++jb) {
    double lr[8];
    const int rowm = (lane >= jb * 8 && lane < k) ? lane : jb * 8;
    const int base = bidx(rowm, jb * 8);
#pragma unroll
    for (int t = 0; t < 8; ++t) lr[t] = ch.Lv[base + t];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = jb * 8 + t;
      if (j < k) {
        const double wj = bcast_u(x, j) * bcast_u(rdm, j);
        if (lane == j) x = wj;
        else if (lane > j && lane < k) x -= lr[t] * wj;
      }
    }
  }
  if (lane < k) ch.w[lane] = x;
  M.Q = wave_sum(lane < k ? x * x : 0.0);
  M.SS = ch.ss0q + M.c - M.Q;
  wave_sync();
  if (ch.mode) {
    // SpikeSlabSampler::log_model_prob, SpikeSlabSampler.cpp:171-203:
    // log pi(g) + .5 log|P_g| - .5 mu'P mu - [.5 log|V_g| - .5 |L^{-1} r|^2]
    if (!oka) {
      M.lda = -BA_INF;
      M.logp = -BA_INF;
      return;
    }
    M.logp = lp + 0.5 * (M.lda - M.ldv) - 0.5 * (M.c - M.Q);
    return;
  }
  if (!(M.SS >= 0.0) || isinf(M.SS)) {
    M.bad = CHAIN_NEGATIVE_SS;
    M.logp = -BA_INF;
    return;
  }
  if (!oka) {
    M.lda = -BA_INF;
    M.logp = -BA_INF;
    return;
  }
  M.logp = lp + 0.5 * (M.lda - M.ldv) - (0.5 * ch.DF - 1.0) * log(M.SS);
}

// flip variable j in the LDS copy of gamma and in the sorted list g
__device__ __forceinline__ void apply_flip(Chain &ch, int j) {
  const int lane = ch.lane, k = ch.k;
  const int gm = (lane < k) ? ch.g[lane] : 0x7fffffff;
  const int below = __popcll(__ballot(lane < k && gm < j));
  const bool add = !ch.gam[j];
  wave_sync();
  if (add) {
    const int up = __shfl_up(gm, 1, WAVE);
    if (lane == below) ch.g[lane] = (uint16_t)j;
    else if (lane > below && lane <= k) ch.g[lane] = (uint16_t)up;
    if (lane == 0) ch.gam[j] = 1;
    ch.k = k + 1;
  } else {
    const int dn = __shfl_down(gm, 1, WAVE);
    if (lane >= below && lane < k - 1) ch.g[lane] = (uint16_t)dn;
    if (lane == 0) ch.gam[j] = 0;
    ch.k = k - 1;
  }
  wave_sync();
}

// point the chain at one of its two (table, model block) slots
__device__ __forceinline__ void bind_slot(Chain &ch, const SsvsParams &P, int chain, int slot) {
  const size_t c = (size_t)slot * P.chains + chain;
  ch.tab_lp = P.table_lp + c * ch.p;
  ch.tab_kind = P.table_kind + c * ch.p;
  ch.sc_store = P.model_scratch + c * P.model_scratch_stride;
}

// Copy the current model's wave-uniform data from LDS to this chain's HBM
// block, make it visible to the scalar cache and re-derive the read pointer.
template <int NB>
__device__ __forceinline__ void publish_model(Chain &ch, const Model &M) {
  const int lane = ch.lane, k = ch.k;
  constexpr int KCAP = NB * 8;
  const SsvsScalarLayout S = ssvs_scalar_layout(KCAP);
  const int kpad = (k + 7) & ~7;
  const int nblk = (kpad / 8) * (kpad / 8 + 1) / 2;
  double *dst = ch.sc_store;
  for (int e = lane; e < nblk * 64; e += WAVE) {
    dst[S.Lv + e] = ch.Lv[e];
    dst[S.La + e] = ch.La[e];
  }
  if (lane < kpad) {
    dst[S.rdv + lane] = ch.rdv[lane];
    dst[S.rda + lane] = ch.rda[lane];
    dst[S.w + lane] = ch.w[lane];
    dst[S.bg + lane] = ch.bg[lane];
    ((int *)(dst + S.g))[lane] = (lane < k) ? (int)ch.g[lane] : 0;
  }
  if (lane == 0) {
    double *sc = dst + S.scal;
    sc[0] = M.logp; sc[1] = M.lp; sc[2] = M.ldv; sc[3] = M.lda;
    sc[4] = M.Q; sc[5] = M.c; sc[6] = M.SS; sc[7] = M.pd ? 1.0 : 0.0;
  }
  if constexpr (NB >= MF_MIN_NB) {
    // what the table fills on the matrix cores multiply by (ssvs_fill_mfma.h); the fence
    // also drops the CU's cached lines of the block's previous contents, for both waves
    diag_inverses(ch.Lv, ch.rdv, dst + S.iv, k, mf_block_rows(k), lane);
    diag_inverses(ch.La, ch.rda, dst + S.ia, k, mf_block_rows(k), lane);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  }
  unsigned long long u = (unsigned long long)dst;
  asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(u) : : "memory");
  ch.sc = (c_f64 *)u;
}

// The inverse copy: after a rejected exact evaluation the factors of the
// current model come back from the chain's HBM block (bitwise what a second
// factorisation would produce, at the cost of one coalesced read).
template <int NB>
__device__ __forceinline__ void restore_model(Chain &ch) {
  const int lane = ch.lane, k = ch.k;
  constexpr int KCAP = NB * 8;
  const SsvsScalarLayout S = ssvs_scalar_layout(KCAP);
  const int kpad = (k + 7) & ~7;
  const int nblk = (kpad / 8) * (kpad / 8 + 1) / 2;
  const double *src = ch.sc_store;
  // (every load goes out before the first LDS write: written load / store / load / store, one
  // call site's copy came out as six L2 round trips in a row)
  const bool row = lane < kpad;
  double r0 = 0.0, r1 = 0.0, r2 = 0.0, r3 = 0.0;
  if (row) {
    r0 = src[S.rdv + lane];
    r1 = src[S.rda + lane];
    r2 = src[S.w + lane];
    r3 = src[S.bg + lane];
  }
  for (int e = lane; e < nblk * 64; e += WAVE) {
    double lv = src[S.Lv + e], la = src[S.La + e];
    asm volatile("" : "+v"(lv), "+v"(la), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
    ch.Lv[e] = lv;
    ch.La[e] = la;
  }
  if (row) {
    ch.rdv[lane] = r0;
    ch.rda[lane] = r1;
    ch.w[lane] = r2;
    ch.bg[lane] = r3;
  }
  wave_sync();
}

// Per-lane forward substitution L x = rhs, x in registers (in: rhs, out:
// solution).  The factor's blocks and reciprocal diagonal come through the
// scalar cache (SGPR operands).  Rows >= k of the last block are zero with
// rd = 0, so their x stays 0.
template <int NB>
__device__ __forceinline__ void solve_blocks(c_f64 *__restrict__ LB,
                                             c_f64 *__restrict__ rd, int k,
                                             double (&x)[NB * 8]) {
#pragma unroll
  for (int I = 0; I < NB; ++I) {
    if (I * 8 < k) {
      double acc[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] = x[I * 8 + r];
#pragma unroll
      for (int J = 0; J < I; ++J) {
        c_f64 *blk = LB + ((I * (I + 1)) / 2 + J) * 64;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
          for (int r = 0; r < 8; ++r) acc[r] -= blk[r * 8 + c] * x[J * 8 + c];
      }
      c_f64 *blk = LB + ((I * (I + 1)) / 2 + I) * 64;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        x[I * 8 + r] = acc[r] * rd[I * 8 + r];
#pragma unroll
        for (int r2 = r + 1; r2 < 8; ++r2) acc[r2] -= blk[r2 * 8 + r] * x[I * 8 + r];
      }
    }
  }
}

// the logit sampler's V, computed column by column: is vector j of this chain's V
// (needed by any model that includes j) still to be computed?  (P.col_valid is the
// chain's own words here.)
__device__ __forceinline__ bool column_missing(const SsvsParams &P, int j) {
  return P.col_valid != nullptr && !((P.col_valid[j >> 5] >> (j & 31)) & 1u);
}

struct Proposal {
  double logp;   // log_model_prob of the flipped model (-inf: impossible)
  bool slow;     // needs the exact path (non-zero prior mean on j)
  bool bad_ss;   // SS' < 0: the reference would throw here
};

// Evaluate this lane's proposal "flip j" against the current model.
// NAT: the wave's lanes hold CONSECUTIVE variables j (table fill), so element
// (g_m, j) of the symmetric matrix is read down column j of row g_m and the 64
// lanes share four cache lines; otherwise (arbitrary j per lane) it is read
// as element (j, g_m), the lane's k elements sharing a few lines of its row.
template <int NB, bool NAT>
__device__ __forceinline__ Proposal eval_proposal(const SsvsParams &P, Chain &ch,
                                                  const Model &M, int j,
                                                  bool valid, StampCtx &sx, int jbase = 0) {
  const int p = ch.p, k = ch.k;
  Proposal out;
  out.logp = -BA_INF;
  out.slow = false;
  out.bad_ss = false;
  const bool add = valid && !ch.gam[j];
  const bool drop = valid && !add;
  const int kn = add ? k + 1 : k - 1;
  // Everything the classification and the epilogue read about variable j is fetched up
  // front, all loads independent of each other (j is a valid index whatever `valid` says):
  // as conditional loads they formed a chain of three dependent round trips to L2 --
  // log prior -> prior mean -> diagonals -- in every proposal round.
  const double l1 = P.l1[j], l0 = P.l0[j];
  const double bj_raw = P.b[j];
  const double vjj_raw = P.v_diag ? P.v_diag[j] : P.V[(size_t)j * p + j];
  const double ajj_raw = P.A[(size_t)j * p + j];
  const double xtyj_raw = ch.xty[j];
  double lpn = -BA_INF;
  if (valid) {
    // log prior of the flipped model; -inf terms must not meet +inf
    if (add) lpn = (l1 == -BA_INF) ? -BA_INF : ((l0 == -BA_INF) ? -BA_INF : M.lp + (l1 - l0));
    else     lpn = (l0 == -BA_INF) ? -BA_INF : ((l1 == -BA_INF) ? -BA_INF : M.lp + (l0 - l1));
    if (P.max_model_size >= 0 && kn > P.max_model_size) lpn = -BA_INF;
  }
  const bool live = valid && (lpn > -BA_INF);
  const double bj = live ? bj_raw : 0.0;
  const bool empty_after = live && drop && (kn == 0);
  const bool slow = live && !empty_after && (bj != 0.0);
  const bool fast = live && !empty_after && !slow;
  out.slow = slow;
  if (empty_after) {
    out.logp = ch.mode ? lpn : lpn - (0.5 * ch.DF - 1.0) * log(ch.ss0q);
  }
  const double vjj = (fast && add) ? vjj_raw * ch.sv : 0.0;
  const double ajj = (fast && add) ? ajj_raw * ch.sa : 0.0;
  const double xtyj = (fast && add) ? xtyj_raw * ch.sx : 0.0;

  constexpr int KCAP = NB * 8;
  const SsvsScalarLayout S = ssvs_scalar_layout(KCAP);
  c_f64 *sc = ch.sc;
  // (a V computed column by column holds the vectors of included variables only)
  const bool nat = NAT || (P.col_valid != nullptr);
  const size_t stride_g = nat ? (size_t)p : 1, lane_off = nat ? (size_t)j : (size_t)j * p;

  double nv = 0.0, dv = 0.0, na = 0.0, ab = 0.0;
  SUBSTAMP(sx, 1);
  if constexpr (NAT && NB >= MF_MIN_NB) {
    // a fill round at capacity 48 / 64: the wavefront's 64 proposals jbase + lane together,
    // on the matrix cores (ssvs_fill_mfma.h)
    const MfSums z = mf_proposal_sums<NB / 2>(P.V, P.A, p, ch.sv, ch.sa, ch.sc_store, S, S.iv, S.ia, ch.g, k, jbase,
                                              (fast ? 1 : 0) | (add ? 2 : 0), ch.lane);
    nv = z.nv; dv = z.dv; na = z.na; ab = z.ab;
    SUBSTAMP(sx, 5);
  } else if constexpr (NB <= 2) {
    // Small capacities (the bsts path: a handful of variables, a fresh table every sweep
    // because X'y moves): BOTH right-hand sides are gathered before either solve, so that
    // a proposal round is one trip to L2 instead of two in a row (32 more registers for
    // the second vector, which only these instances can afford).
    double xv[NB * 8], xa[NB * 8];
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      if (I * 8 < k) {
        // (all sixteen loads of a block go out before any is used, and the compiler is told
        // so: left to itself it moved each A load under the select that consumes it -- a
        // conditional block with its own wait -- and a block's gather became eight L2 round
        // trips in a row, 7.5 k cycles of a fill pass)
        int gms[8];
        double rv[8], ra[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int m = I * 8 + r;
          gms[r] = (m < k) ? (int)ch.g[m] : 0;  // LDS broadcast read
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const size_t o = (size_t)gms[r] * stride_g + lane_off;
          rv[r] = P.V[o];
          ra[r] = P.A[o];
        }
        asm volatile("" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]), "+v"(rv[4]), "+v"(rv[5]), "+v"(rv[6]), "+v"(rv[7]),
                          "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]));
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int m = I * 8 + r;
          const double v = rv[r] * ch.sv, a = ra[r] * ch.sa;
          const double e = (gms[r] == j) ? 1.0 : 0.0;
          const bool on = fast && m < k;
          xv[m] = on ? (add ? v : e) : 0.0;
          xa[m] = on ? (add ? a : e) : 0.0;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) { xv[I * 8 + r] = 0.0; xa[I * 8 + r] = 0.0; }
      }
    }
    // A[j, g] . b_g for the new element of r
#pragma unroll
    for (int I = 0; I < NB; ++I)
      if (I * 8 < k) {
#pragma unroll
        for (int r = 0; r < 8; ++r) ab += xa[I * 8 + r] * sc[S.bg + I * 8 + r];
      }
    SUBSTAMP(sx, 2);
    solve_blocks<NB>(sc + S.Lv, sc + S.rdv, k, xv);
    SUBSTAMP(sx, 3);
    solve_blocks<NB>(sc + S.La, sc + S.rda, k, xa);
#pragma unroll
    for (int I = 0; I < NB; ++I)
      if (I * 8 < k) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          nv += xv[I * 8 + r] * xv[I * 8 + r];
          dv += xv[I * 8 + r] * sc[S.w + I * 8 + r];
          na += xa[I * 8 + r] * xa[I * 8 + r];
        }
      }
    SUBSTAMP(sx, 5);
  } else {
  double x[NB * 8];
#pragma nounroll
  for (int s = 0; s < 2; ++s) {
    const double *Mat = s ? P.A : P.V;
    const double msc = s ? ch.sa : ch.sv;
    c_f64 *LB = sc + (s ? S.La : S.Lv);
    c_f64 *rd = sc + (s ? S.rda : S.rdv);
    // rhs = Mat[g, j] (add) or e_i (drop); branch-free inside a block so that
    // the block's 8 gathers are in flight together
#pragma unroll
    for (int I = 0; I < NB; ++I) {
      if (I * 8 < k) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int m = I * 8 + r;
          const int gm = (m < k) ? (int)ch.g[m] : 0;  // LDS broadcast read
          const double v = Mat[(size_t)gm * stride_g + lane_off] * msc;
          const double e = (gm == j) ? 1.0 : 0.0;
          x[m] = (fast && m < k) ? (add ? v : e) : 0.0;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) x[I * 8 + r] = 0.0;
      }
    }
    if (s == 1) {
      // A[j, g] . b_g for the new element of r
#pragma unroll
      for (int I = 0; I < NB; ++I)
        if (I * 8 < k) {
#pragma unroll
          for (int r = 0; r < 8; ++r) ab += x[I * 8 + r] * sc[S.bg + I * 8 + r];
        }
    }
#if defined(BA_STAMPS2)
    asm volatile("" :: "v"(x[0]), "v"(x[1]) : "memory");
#endif
    SUBSTAMP(sx, s ? 4 : 2);
    solve_blocks<NB>(LB, rd, k, x);
    double n2 = 0.0, dw = 0.0;
#pragma unroll
    for (int I = 0; I < NB; ++I)
      if (I * 8 < k) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          n2 += x[I * 8 + r] * x[I * 8 + r];
          dw += x[I * 8 + r] * sc[S.w + I * 8 + r];
        }
      }
    if (s == 0) {
      nv = n2;
      dv = dw;
    } else {
      na = n2;
    }
#if defined(BA_STAMPS2)
    asm volatile("" :: "v"(n2), "v"(dw) : "memory");
#endif
    SUBSTAMP(sx, s ? 5 : 3);
  }
  }
  if (fast) {
    double ldv, lda, Q;
    bool ok = true;
    if (add) {
      const double d2 = vjj - nv;
      const double da2 = ajj - na;
      if (!(d2 > 0.0) || !(da2 > 0.0)) ok = false;
      const double rj = xtyj + ab;  // b_j == 0 on this path
      const double wn = (rj - dv) / sqrt(d2);
      Q = M.Q + wn * wn;
      ldv = M.ldv + log(d2);
      lda = M.lda + log(da2);
    } else {
      // nv = (V^{-1})_ii, dv = (V^{-1} r)_i = beta~_i
      Q = M.Q - dv * dv / nv;
      ldv = M.ldv + log(nv);
      lda = M.lda + log(na);
    }
    if (ok && ch.mode) {
      out.logp = lpn + 0.5 * (lda - ldv) - 0.5 * (M.c - Q);
    } else if (ok) {
      const double SS = ch.ss0q + M.c - Q;
      if (!(SS >= 0.0) || isinf(SS)) {
        out.bad_ss = true;
      } else {
        out.logp = lpn + 0.5 * (lda - ldv) - (0.5 * ch.DF - 1.0) * log(SS);
      }
    }
  }
  return out;
}

// shuffle(indx) of cpputil/shuffle.hpp:36-46 -- for i = p-1..1: swap(a[i],
// a[oth[i]]) -- without replaying the swaps serially.  Position i is final
// after step i and receives what position x = oth[i] held just before step i.
// That content was deposited by the most recent earlier step t' > i with
// oth[t'] == x (pred[i]); it was position t's content before step t', which
// in turn was deposited by the smallest t'' > t' with oth[t''] == t' (nxt),
// and so on until a slot nobody wrote, which still holds its original value.
// oth[] must be filled for i = 1..p-1.  Result goes to ch.perm (buffers swap).
// MASKED: blocks of NW x 64 links that did not move in a pass leave the passes (every link of
// such a block points at a chain's end, and ends never move): at p = 4096, eight blocks, the
// walks are a third of the helper wavefront's sweep and the mask takes a quarter off them.  A
// template flag, not a run-time choice: the bookkeeping costs the single block of p = 512 one
// per cent and the headline's instance <4, 2, 2> is compiled without it (docs/TRIED.md).
template <bool MASKED = false>
__device__ __forceinline__ void parallel_shuffle(Chain &ch, StampCtx &sx) {
  const int p = ch.p, lane = ch.lane;
  constexpr int NONE = 0xFFFF;
  for (int j = lane; j < p; j += WAVE) ch.last[j] = (uint32_t)NONE;
  wave_sync();
  HSTAMP(sx, 0);
  // ---- previous step with the same target: rounds of 64 steps over decreasing
  // t, lane l <-> step T - l.  One LDS exchange per round does the search: the
  // LDS resolves same-address exchanges of a wavefront instruction in ascending
  // lane order (checked when the engine is created), so a lane gets back the
  // step of the nearest lower lane with its target -- or what earlier rounds
  // left there -- and the array ends up holding each target's smallest step.
  constexpr int RC = 8;
  for (int T0 = p - 1; T0 >= 1; T0 -= RC * WAVE) {
    int key[RC];
#pragma unroll
    for (int r = 0; r < RC; ++r) {
      const int t = T0 - r * WAVE - lane;
      key[r] = (t >= 1) ? (int)ch.oth[t] : 0;
    }
#pragma unroll
    for (int r = 0; r < RC; ++r) {
      const int t = T0 - r * WAVE - lane;
      if (t >= 1) {
        const uint32_t old = __hip_atomic_exchange(&ch.last[key[r]], (uint32_t)t, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WAVEFRONT);
        ch.pred[t] = (uint16_t)old;
      }
    }
  }
  wave_sync();
  HSTAMP(sx, 1);
  // last[x] = smallest t >= 1 with oth[t] == x.  nxt(t) = smallest t' > t with
  // oth[t'] == t: last[t] unless that is the self swap t, then pred[t].
  const int pred0 = ch.last[0];
  wave_sync();
  for (int t = lane; t < p; t += WAVE) {
    if (t >= 1) {
      const int l = ch.last[t];
      ch.last[t] = (uint32_t)((l == t) ? (int)ch.pred[t] : l);
    }
  }
  wave_sync();
  HSTAMP(sx, 2);
  const lds_u16 *src_perm = ch.perm;
  lds_u16 *dst = ch.perm_alt;
  // Each position's source is the END of a chain of nxt links.  Chains are not
  // walked: pointer jumping R[c] <- R[R[c]] (R[c] = nxt(c), or c itself at a
  // chain's end) halves every distance per round, all steps at once, in place
  // (any mix of old and new values still points down the chain).  A lane holds
  // NW steps, their LDS round trips side by side.
  constexpr int NW = 8;
  for (int t = lane; t < p; t += WAVE) {
    if (t >= 1) {
      const uint32_t l = ch.last[t];
      if (l == (uint32_t)NONE) ch.last[t] = (uint32_t)t;
    }
  }
  wave_sync();
  if constexpr (!MASKED) {
  for (bool moved = true; moved;) {
    moved = false;
    for (int tb = 0; tb < p; tb += NW * WAVE) {
      int r1[NW], r2[NW];
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const int t = tb + u * WAVE + lane;
        r1[u] = (t >= 1 && t < p) ? (int)ch.last[t] : 1;
      }
#pragma unroll
      for (int u = 0; u < NW; ++u) r2[u] = (int)ch.last[r1[u]];
      bool ch_any = false;
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const int t = tb + u * WAVE + lane;
        if (t >= 1 && t < p && r2[u] != r1[u]) {
          ch.last[t] = (uint32_t)r2[u];
          ch_any = true;
        }
      }
      moved |= (__any(ch_any) != 0);
      wave_sync();
    }
  }
  } else {
  unsigned long long active = ~0ull;   // (bit b: block b still moving; more than 64 blocks share bit 63)
  while (active) {
    unsigned long long still = 0ull;
    for (int tb = 0, b = 0; tb < p; tb += NW * WAVE, ++b) {
      const unsigned long long bit = 1ull << (b < 63 ? b : 63);
      if (!(active & bit)) continue;
      int r1[NW], r2[NW];
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const int t = tb + u * WAVE + lane;
        r1[u] = (t >= 1 && t < p) ? (int)ch.last[t] : 1;
      }
#pragma unroll
      for (int u = 0; u < NW; ++u) r2[u] = (int)ch.last[r1[u]];
      bool ch_any = false;
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const int t = tb + u * WAVE + lane;
        if (t >= 1 && t < p && r2[u] != r1[u]) {
          ch.last[t] = (uint32_t)r2[u];
          ch_any = true;
        }
      }
      if (__any(ch_any) != 0) still |= bit;
      wave_sync();
    }
    active = still;
  }
  }
  for (int ib = 0; ib < p; ib += NW * WAVE) {
    int c0[NW], src[NW];
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int i = ib + u * WAVE + lane;
      c0[u] = (i >= p) ? NONE : ((i == 0) ? pred0 : (int)ch.pred[i]);
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int i = ib + u * WAVE + lane;
      // nobody deposited anything: the partner's original content (slot 0 keeps its own)
      const int direct = (i >= p || i == 0) ? 0 : (int)ch.oth[i];
      const int via = (int)ch.last[c0[u] == NONE ? 0 : c0[u]];
      src[u] = (c0[u] == NONE) ? direct : via;
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
      const int i = ib + u * WAVE + lane;
      if (i < p) dst[i] = src_perm[src[u]];
    }
  }
  wave_sync();
  lds_u16 *tmp = ch.perm;
  ch.perm = ch.perm_alt;
  ch.perm_alt = tmp;
  HSTAMP(sx, 3);
}

// k standard normals in stream order (distributions/mvn.cpp:114-122), lane m
// receives z_m.  Seven times out of eight Kinderman-Ramage takes its first
// branch (two uniforms, one line of arithmetic); that branch is evaluated for
// every window position at once and a scalar walk picks the draws a
// sequential reader would have made, falling back to the generic transform
// where another branch is due.
__device__ __forceinline__ double draw_normals(WinRng &rng, int k) {
  const double A = 2.216035867166471;
  const int lane = rng.lane;
  double z = 0.0;
  int m = 0;
  while (m < k) {
    if (!rng.have || rng.off > 125) rng.fill(rng.get_pos());
    const uint64_t wb = rng.wbase;
    // start at even offset 2l: (w0, w1) of lane l; at odd offset 2l+1: w1 of
    // lane l and w0 of lane l+1
    const double w0n = __shfl_down(rng.w0, 1);
    const double zE = A * (1.131131635444180 * rng.w0 + rng.w1 - 1);
    const double zO = A * (1.131131635444180 * rng.w1 + w0n - 1);
    const unsigned long long mE = __ballot(rng.w0 < 0.884070402298758);
    const unsigned long long mO = __ballot(rng.w1 < 0.884070402298758);
    int o = rng.off;
    while (m < k && o <= 125) {
      const int l = o >> 1;
      const bool odd = o & 1;
      double zm;
      const unsigned long long fastrun = (odd ? mO : mE) >> l;
      if (fastrun & 1ull) {
        // a RUN of first-branch draws: each takes two numbers, so they start at o, o + 2, ...
        // as long as the bits say so -- handed to lanes m, m + 1, ... in one move instead of
        // one scalar step per draw
        int run = (~fastrun == 0ull) ? 64 : __ffsll((long long)~fastrun) - 1;
        const int room = (125 - o) / 2 + 1;
        run = run < room ? run : room;
        run = run < k - m ? run : k - m;
        const int src = l + (lane - m);
        const double zz = __shfl(odd ? zO : zE, (src >= 0 && src < WAVE) ? src : 0);
        if (lane >= m && lane < m + run) z = zz;
        m += run;
        o += 2 * run;
        continue;
      } else {
        rng.off = o;
        zm = d_norm_rand(rng);
        o = rng.off;
        if (rng.wbase != wb) {  // the window moved: recompute the candidates
          if (lane == m) z = zm;
          ++m;
          break;
        }
      }
      if (lane == m) z = zm;
      ++m;
    }
    rng.off = o;
  }
  return z;
}

// A request to (re)build the model after changing gamma, served at the single
// place in the sweep loop where refactor() is instantiated.
enum : int {
  EV_NONE = 0,
  EV_INIT,     // rebuild, no decision (launch start; after make_valid)
  EV_FORCE,    // flip f1 was accepted on the fast path: rebuild
  EV_TRY_GE,   // exact evaluation of a flip: reject iff log u >  delta
  EV_TRY_LT    // swap move:                  accept iff log u <  delta
};
struct Pending {
  int kind;
  int f1, f2;         // variables to flip (-1: none)
  double lu;          // log u of the decision
  double lfw, lrev;   // log forward / reverse proposal weights (0 for flips)
  bool check_legal;   // EV_INIT after make_valid: -inf => ILLEGAL_START
};

// BregVsSampler::attempt_swap (BregVsSampler.cpp:277-310), proposal half:
// CorrelationMap::propose_swap / proposal_weight (CorrelationMap.cpp:61-115).
// Wave-uniform control flow; every lane walks the same CSR lists.  Fills `pe`
// when a swap is proposed; the evaluation happens at the refactor site.
template <class R>
__device__ __forceinline__ void propose_swap(const SsvsParams &P, Chain &ch,
                                             R &rng, Pending &pe,
                                             int *status) {
  if (P.cm_start == nullptr) return;
  const int k = ch.k, p = ch.p;
  if (k == 0 || k == p) return;
  // Selector::random_included_position, LinAlg/Selector.cpp:297-304
  const int pos = d_random_int(rng, 0, k - 1);
  const int index = ch.g[pos];
  if (!ch.nbr[index]) return;  // no partner above the threshold (the usual case)
  const int lo = P.cm_start[index], hi = P.cm_start[index + 1];
  if (lo == hi) return;
  double total = 0.0;
  for (int i = lo; i < hi; ++i)
    if (!ch.gam[P.cm_idx[i]]) total += P.cm_cor[i];
  if (total == 0.0) return;
  // rmulti_mt on weights / total (distributions/rmulti.cpp:41-78)
  double probsum = 0.0;
  for (int i = lo; i < hi; ++i)
    if (!ch.gam[P.cm_idx[i]]) probsum += P.cm_cor[i] / total;
  const double tmp = d_runif(rng, 0.0, probsum);
  double psum = 0.0, forward_w = 0.0;
  int candidate = -1;
  for (int i = lo; i < hi; ++i) {
    if (ch.gam[P.cm_idx[i]]) continue;
    const double wgt = P.cm_cor[i] / total;
    psum += wgt;
    if (tmp <= psum) {
      candidate = P.cm_idx[i];
      forward_w = wgt;
      break;
    }
  }
  if (candidate < 0) {
    *status = CHAIN_RNG_BRANCH;
    return;
  }
  // reverse weight = proposal_weight(included', candidate, index) where
  // included' = gamma - index + candidate
  double rev;
  {
    const int l2 = P.cm_start[candidate], h2 = P.cm_start[candidate + 1];
    double ans = -BA_INF, tot = 0.0;
    for (int i = l2; i < h2; ++i) {
      const int v = P.cm_idx[i];
      const bool inc = (v == candidate) ? true : ((v == index) ? false : (bool)ch.gam[v]);
      if (!inc) {
        if (v == index) ans = P.cm_cor[i];
        tot += P.cm_cor[i];
      }
    }
    rev = (tot == 0.0) ? 0.0 : ans / tot;
  }
  pe.kind = EV_TRY_LT;
  pe.f1 = index;
  pe.f2 = candidate;
  pe.lfw = log(forward_w);
  pe.lrev = log(rev);
  pe.lu = log(d_runif(rng, 0.0, 1.0));
  pe.check_legal = false;
}

}  // namespace

// ============================================================================
// grid = chains, block = 64 * W; NB = kcap / 8.
//
// A workgroup of W wavefronts serves one chain.  Wave 0 (the master) runs the
// sweep; the other waves exist for the proposal batches: a batch is 64 * W
// proposals, wave w evaluating positions i0 + 64 w + lane against the current
// model, and for the lane-parallel uniforms of the shuffle.  The waves meet at
// two workgroup barriers per command; the current model reaches the helpers
// through the LDS control block (scalars) and the chain's HBM model block
// (factors, read through the scalar cache).
//
// Two equivalent ways of walking a sweep's proposals (same decisions, same
// chain):
//   batch mode  evaluate the next 64 W positions of the permutation; what was
//               evaluated behind the first stop is thrown away.  Best while
//               flips are accepted often (burn-in, ridge-like posteriors).
//   table mode  log_model_prob(gamma ^ {j}) depends on the current model only,
//               and the model changes only when a flip is accepted.  So it is
//               evaluated ONCE for every j after each change (p / (64 W) fill
//               rounds, natural variable order) into a per-chain table of
//               acceptance thresholds E_j = exp(logp_j' - logp), and a sweep's
//               decisions are look-ups against fresh uniforms: u_i <= E[perm[i]].
//               At stationarity (well under one accepted flip per sweep) most
//               sweeps need no evaluation at all.  The table outlives the launch.
// The master picks the mode per sweep from the previous sweep's stop count.
//
// Quiet sweeps fork (table mode, W > 1): shuffle and flips consume p - 1 + nflips
// stream numbers whatever happens, so the tail's stream position is known when
// the sweep starts.  Wave 1 shuffles and walks the table while the master runs
// the tail (swap proposal, sigma, beta) as if no flip were going to be accepted;
// at the join a stop in the walk rolls the tail back.
//
// The master's state machine: PH_BEGIN (shuffle or fork) -> PH_FLIPS (rounds of
// proposals / table walks up to the next stop; a stop queues an event) ->
// PH_SWAP -> PH_TAIL -> [PH_JOIN] -> PH_COMMIT, with model rebuilds served as
// events at the top of the loop.

enum : int { CMD_EXIT = 0, CMD_EVAL = 1, CMD_UNIF = 2, CMD_DECIDE = 3, CMD_SHUFFLE_DECIDE = 4 };
// control block (doubles): 0 cmd, 1 k, 2 i0, 3..8 the current model's scalars
// (their home), 9 nflips; u64 view at 10: flip_pos / uniform base position;
// evaluator slots from 16; 44.. the forked tail's roll-back copy; 48.. the
// launch's scalar accumulators (ACC_* order)
enum : int { CT_CMD = 0, CT_K = 1, CT_I0 = 2, CT_LOGP = 3, CT_LP = 4, CT_LDV = 5,
             CT_LDA = 6, CT_Q = 7, CT_C = 8, CT_NFLIPS = 9, CT_POS = 10, CT_PERMSEL = 12,
             CT_EVMODE = 13, CT_CUR = 14,
             CT_SLOT0 = 16, CT_SLOT_STRIDE = 6, CT_ROLL = 44, CT_ACC = 48 };
// slot: SL_F = permutation position of the wave's earliest stop (-1: none)
enum : int { SL_F = 0, SL_J = 1, SL_KIND = 2, SL_LOGU = 3, SL_MARGIN = 4, SL_DELTA = 5 };
enum : int { STOP_ACCEPT = 1, STOP_SLOW = 2, STOP_BAD = 3 };

__device__ __forceinline__ void shuffle_targets(const PhiloxKey &key, uint64_t pos,
                                                int p, int tid, int nthreads,
                                                lds_u16 *oth, bool whole_range = false) {
  // uniform number pos + t (t = 0..p-2) picks the partner of i = p-1-t:
  // random_int_mt(rng, 0, i) (cpputil/shuffle.hpp:36-46); one Philox block
  // serves two consecutive steps.  whole_range (BinomialLogitSpikeSlabSampler.cpp:
  // 181-187): uniform t (t = 0..p-1) picks the partner of i = t from 0..p-1.
  const int nt = whole_range ? p : p - 1;
  const uint64_t b0 = pos >> 1, b1 = (pos + (uint64_t)(nt - 1)) >> 1;
  for (uint64_t b = b0 + (uint64_t)tid; b <= b1; b += (uint64_t)nthreads) {
    double u[2];
    philox_pair(key, b, &u[0], &u[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long t = (long long)(2 * b + h) - (long long)pos;
      if (t >= 0 && t < nt) {
        if (whole_range) {
          oth[t] = (uint16_t)(int)floor(0.0 + ((double)p - 0.0) * u[h]);
        } else {
          const int i = p - 1 - (int)t;
          oth[i] = (uint16_t)(int)floor(0.0 + ((double)(i + 1) - 0.0) * u[h]);
        }
      }
    }
  }
}

enum : int { EVM_BATCH = 0, EVM_FILL = 1 };

// One wave's share of a round.
//   EVM_BATCH   lane = position i0 + 64 wave + lane: evaluate perm[position],
//               decide, leave the wave's earliest stop (accepted / needs the
//               exact path / negative SS) in its slot;
//   EVM_FILL    lane = variable i0 + 64 wave + lane: evaluate it against the
//               current model and store the result in the chain's table;
// (Decisions by table look-up are decide_walk below.)
template <int NB>
__device__ __forceinline__ void eval_share(const SsvsParams &P, Chain &ch,
                                           const Model &M, const PhiloxKey &key,
                                           uint64_t flip_pos, int nflips, int i0,
                                           int evmode, int wave, lds_f64 *ctl,
                                           StampCtx &sx) {
  const int lane = ch.lane;
  SUBSTAMP(sx, 7);
  const int idx = i0 + WAVE * wave + lane;
  if (evmode == EVM_FILL) {
    const bool valid = idx < ch.p;
    const Proposal pr = eval_proposal<NB, true>(P, ch, M, valid ? idx : 0, valid, sx, idx - lane);
    if (valid) {
      // acceptance threshold in the uniform's own scale: log u <= logp' - logp
      // <=> u <= exp(logp' - logp)   (0 for an impossible model, inf / NaN --
      // never exceeded -- when the current model itself is impossible)
      ch.tab_lp[idx] = exp(pr.logp - M.logp);
      ch.tab_kind[idx] = (uint8_t)(pr.bad_ss ? STOP_BAD : (pr.slow ? STOP_SLOW : 0));
    }
    SUBSTAMP(sx, 6);
    return;
  }
  const bool valid = idx < nflips;
  const int j = valid ? (int)ch.perm[idx] : 0;
  const double u = philox_uniform(key, flip_pos + (uint64_t)idx);
  const double logu = log(u);
#if defined(BA_STAMPS2)
  asm volatile("" :: "v"(logu) : "memory");
#endif
  SUBSTAMP(sx, 7);
  const Proposal pr = eval_proposal<NB, false>(P, ch, M, j, valid, sx);
  const double lpj = pr.logp;
  const bool slow = valid && pr.slow;
  const bool bad = valid && pr.bad_ss;
  const double delta = lpj - M.logp;
  const bool accept = valid && !slow && !bad && !(logu > delta);
  const unsigned long long m_acc = __ballot(accept);
  const unsigned long long m_slow = __ballot(slow);
  const unsigned long long m_bad = __ballot(bad);
  const unsigned long long m_stop = m_acc | m_slow | m_bad;
  const int f = m_stop ? (__ffsll((long long)m_stop) - 1) : WAVE;
  // lanes before f are settled rejections; f itself counts if accepted
  const bool counted = valid && (lane < f || (lane == f && ((m_acc >> f) & 1ull)));
  const double mg = counted && (lpj > -BA_INF) ? fabs(logu - delta) : BA_INF;
  const double mmin = wave_min(mg);
  if (lane == (f < WAVE ? f : 0)) {
    lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * wave;
    int kind = 0;
    if (f < WAVE) kind = ((m_bad >> f) & 1ull) ? STOP_BAD : (((m_acc >> f) & 1ull) ? STOP_ACCEPT : STOP_SLOW);
    sl[SL_F] = (f < WAVE) ? (double)(idx) : -1.0;
    sl[SL_J] = (double)j;
    sl[SL_KIND] = (double)kind;
    sl[SL_LOGU] = logu;
    sl[SL_MARGIN] = mmin;
    sl[SL_DELTA] = delta;
  }
  SUBSTAMP(sx, 6);
}

// A sweep's decisions by table look-up, from position i0 to the first stop:
// one wavefront, two positions per lane and round (both uniforms of the lane's
// Philox block).  spos = -1: reached nflips without a stop.
struct DecideResult {
  int spos, j, kind;
  double logu, margin;
};
// one position of a walk round (a lane holds 2 R of them).  Named scalars, not arrays:
// any array here ends up indexed by the (run-time) sub-round or half of the stop, which
// keeps the whole array in scratch memory -- 128 bytes per lane written in every round
// of every walk with the first version of this routine.
struct WalkPos {
  double u, e;
  int j, kd;
  bool val, acc;
};
__device__ __forceinline__ void walk_pos_load(const Chain &ch, int q, int i0, int nflips, WalkPos &w) {
  w.val = q >= i0 && q < nflips;
  w.j = w.val ? (int)ch.perm[q] : 0;
}
__device__ __forceinline__ void walk_pos_table(const Chain &ch, WalkPos &w) {
  w.e = ch.tab_lp[w.j];
  w.kd = ch.tab_kind[w.j];
}
__device__ __forceinline__ unsigned long long walk_pos_stop(WalkPos &w) {
  const bool special = w.kd != 0;
  w.acc = w.val && !special && !(w.u > w.e);
  return __ballot(w.val && (special || w.acc));
}
__device__ __forceinline__ void decide_walk(const Chain &ch, const PhiloxKey &key,
                                            uint64_t flip_pos, int i0, int nflips,
                                            DecideResult &out) {
  // R Philox blocks (2 R positions) per lane and round, all loads of a round
  // in flight together: the table look-ups are latency, not bandwidth.  The
  // table holds E_j = exp(logp_j' - logp), so a decision is u <= E_j and no
  // logarithm is taken except for the one uniform of a stop.
  constexpr int R = 4;
  const int lane = ch.lane;
  double lane_margin = BA_INF;
  out.spos = -1; out.j = 0; out.kind = 0; out.logu = 0.0;
  while (i0 < nflips) {
    const uint64_t blk0 = (flip_pos + (uint64_t)i0) >> 1;
    const int qb = (int)((long long)(2 * blk0) - (long long)flip_pos);  // position of block blk0, half 0
    WalkPos a0, a1, b0, b1, c0, c1, d0, d1;   // sub-rounds 0..3, halves 0 / 1
#define WALK_ALL(OP) OP(0, a0, a1) OP(1, b0, b1) OP(2, c0, c1) OP(3, d0, d1)
#define WALK_LOAD(r, x0, x1)                                            \
    walk_pos_load(ch, qb + 2 * ((r) * WAVE + lane), i0, nflips, x0);     \
    walk_pos_load(ch, qb + 2 * ((r) * WAVE + lane) + 1, i0, nflips, x1);
    WALK_ALL(WALK_LOAD)
#define WALK_TABLE(r, x0, x1) walk_pos_table(ch, x0); walk_pos_table(ch, x1);
    WALK_ALL(WALK_TABLE)
#define WALK_UNIF(r, x0, x1) philox_pair(key, blk0 + (uint64_t)((r) * WAVE + lane), &x0.u, &x1.u);
    WALK_ALL(WALK_UNIF)
    int f = 1 << 20, fr = 0;  // first stop: 2 lane + half within sub-round fr
    // the stopping position's variable, kind and uniform, taken where the stop is found
    // (every lane keeps ITS candidates of that sub-round; the stop's lane is read below)
    int sj = 0, sk = 0;
    double su = 0.0;
#define WALK_STOP(r, x0, x1)                                                              \
    {                                                                                     \
      const unsigned long long m0 = walk_pos_stop(x0), m1 = walk_pos_stop(x1);            \
      if (f == (1 << 20)) {                                                               \
        const int f0 = m0 ? 2 * (__ffsll((long long)m0) - 1) : 1 << 20;                   \
        const int f1 = m1 ? 2 * (__ffsll((long long)m1) - 1) + 1 : 1 << 20;               \
        const int fm = f0 < f1 ? f0 : f1;                                                 \
        if (fm < (1 << 20)) {                                                             \
          f = fm;                                                                         \
          fr = (r);                                                                       \
          const bool h1 = (fm & 1) != 0;                                                  \
          sj = h1 ? x1.j : x0.j;                                                          \
          sk = h1 ? x1.kd : x0.kd;                                                        \
          su = h1 ? x1.u : x0.u;                                                          \
        }                                                                                 \
      }                                                                                   \
    }
    WALK_ALL(WALK_STOP)
    const int fkey = (f == (1 << 20)) ? (1 << 30) : fr * 2 * WAVE + f;  // order within the round
    // |u / E - 1|, to first order the distance |log u - (logp' - logp)|
#define WALK_MARGIN1(r, h, x)                                                                   \
    {                                                                                           \
      const int me = (r) * 2 * WAVE + 2 * lane + (h);                                           \
      const bool counted = x.val && (me < fkey || (me == fkey && x.acc)) && x.e > 0.0;          \
      const double mg = fabs(x.u - x.e) * __builtin_amdgcn_rcp(x.e);                            \
      if (counted) lane_margin = fmin(lane_margin, mg);                                         \
    }
#define WALK_MARGIN(r, x0, x1) WALK_MARGIN1(r, 0, x0) WALK_MARGIN1(r, 1, x1)
    WALK_ALL(WALK_MARGIN)
#undef WALK_MARGIN
#undef WALK_MARGIN1
#undef WALK_STOP
#undef WALK_UNIF
#undef WALK_TABLE
#undef WALK_LOAD
#undef WALK_ALL
    if (f == (1 << 20)) {
      i0 = qb + 2 * R * WAVE;  // first position of the next round
      continue;
    }
    const int fl = f >> 1;
    out.spos = qb + fr * 2 * WAVE + f;
    out.j = __builtin_amdgcn_readlane(sj, fl);
    out.kind = __builtin_amdgcn_readlane(sk, fl);
    out.logu = log(bcast_u(su, fl));
    break;
  }
  out.margin = wave_min(lane_margin);
}

}  // namespace boom_amd
