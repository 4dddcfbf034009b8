// SSVS Gibbs sweep for many independent chains on gfx950: one chain per
// wavefront (64-thread workgroup), its working set in LDS.
//
// What one sweep computes is BregVsSampler::draw()
// (Models/Glm/PosteriorSamplers/BregVsSampler.cpp:252-261):
//   draw_model_indicators (:353-378)  shuffle indx, p Metropolised flips,
//                                     correlation swap move (:277-310)
//   set_reg_post_params   (:395-484)  V = A_g + S_g, beta~ = V^{-1} r, DF, SS
//   draw_sigma            (:313-324)  sigma^2 = 1/Gamma(DF/2, SS/2)
//   draw_beta             (:326-351)  beta = beta~ + chol(V/sigma^2)^{-T} z
//
// How it is computed here is NOT how the reference does it.  The reference
// evaluates log_model_prob(gamma') from scratch for every proposal (three
// k x k Cholesky factorisations, k = model size).  Here the chain keeps
// L_V = chol(V_g), L_A = chol(A_g), w = L_V^{-1} r and the scalars
// log|V_g|, log|A_g|, ||w||^2, b_g'A_g b_g for the CURRENT model, and the 64
// lanes evaluate the next 64 proposals of the sweep speculatively, each
// against the current model:
//   add j :  l = L_V^{-1} V[g,j],  d2 = V_jj - |l|^2   -> log|V'| = log|V| + log d2
//            la = L_A^{-1} A[g,j], da2 = A_jj - |la|^2 -> log|A'| = log|A| + log da2
//            w_new = (r_j - l.w)/sqrt(d2)              -> |w'|^2 = |w|^2 + w_new^2
//   drop i:  c = L_V^{-1} e_i  -> (V^{-1})_ii = |c|^2, log|V'| = log|V| + log|c|^2,
//            |w'|^2 = |w|^2 - (c.w)^2/|c|^2 ; same with L_A for log|A'|
//   SS' = ss0 + yty + b'Ab - |w'|^2
// The flip uniforms sit at fixed positions of the chain's Philox stream, so
// lane i tests "log u_i <= logp'_i - logp" directly; the first accepting lane
// (in sweep order) wins, everything before it was a correct rejection, and the
// next batch starts right after it.  The resulting Markov chain is the
// reference's chain; only the arithmetic route differs (O(k^2) per proposal
// and 64 proposals at a time instead of O(k^3) one at a time).
//
// Each lane's triangular solve keeps its solution vector in registers
// (static indexing, fully unrolled over 8 x 8 blocks) and streams the factor's
// blocks from LDS with wave-uniform reads: the FMAs of a block row are
// independent, so the solve runs at FMA issue rate instead of LDS latency.
//
// Proposals for a variable with a non-zero prior mean b_j change r for the
// whole model; they take the (rare) exact path: refactor the candidate model.
//
// The Fisher-Yates shuffle of the persistent permutation is done in parallel:
// every final position follows a short chain of "who was swapped into this
// slot last" links instead of replaying the p-1 swaps one after the other.
#include <hip/hip_runtime.h>

#include "device_rng.h"
#include "ssvs_params.h"

namespace boom_amd {

namespace {

constexpr int WAVE = 64;
#define BA_INF (__builtin_inf())

// Diagnostic build only (-DBA_STAMPS): cycles per phase, never in the product.
#ifdef BA_STAMPS
#define STAMP_DECL long long st_last = (long long)__builtin_readcyclecounter(); double st_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) do { const long long t_ = (long long)__builtin_readcyclecounter(); st_ph[i] += (double)(t_ - st_last); st_last = t_; } while (0)
#else
#define STAMP_DECL do { } while (0)
#define STAMP(i) do { } while (0)
#endif
// -DBA_STAMPS -DBA_STAMPS2: the 8 slots time the inside of a proposal batch
// instead (0 uniform+log, 1 classify, 2 V gather, 3 V solve, 4 A gather,
// 5 A solve, 6 epilogue, 7 everything outside the batch)
struct StampCtx { long long last; double ph[8]; };
#if defined(BA_STAMPS) && defined(BA_STAMPS2)
#define SUBSTAMP(c, i) do { const long long t_ = (long long)__builtin_readcyclecounter(); (c).ph[i] += (double)(t_ - (c).last); (c).last = t_; } while (0)
#else
#define SUBSTAMP(c, i) do { } while (0)
#endif
// -DBA_STAMPS -DBA_STAMPS3: the 8 slots time the master's pieces of a forked
// sweep (0 commit, 1 sweep-start copy, 2 fork, 3 swap proposal, 4 sigma,
// 5 normals, 6 back substitution, 7 everything else)
// -DBA_STAMPS -DBA_STAMPS4: the 8 slots time helper wave 1 (0 shuffle uniforms,
// 1 matching rounds, 2 links, 3 walks, 4 table walk, 5 waiting for commands,
// 6 its share of proposal rounds, 7 other)
#if defined(BA_STAMPS) && defined(BA_STAMPS4)
#define HSTAMP(c, i) do { const long long t_ = (long long)__builtin_readcyclecounter(); (c).ph[i] += (double)(t_ - (c).last); (c).last = t_; } while (0)
#else
#define HSTAMP(c, i) do { } while (0)
#endif
#if defined(BA_STAMPS) && defined(BA_STAMPS3)
#define TSTAMP(c, i) do { const long long t_ = (long long)__builtin_readcyclecounter(); (c).ph[i] += (double)(t_ - (c).last); (c).last = t_; } while (0)
#else
#define TSTAMP(c, i) do { } while (0)
#endif

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
      double a[8], b[8], c[8], d[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        a[t] = LA[offi + nb * 64 + t];
        b[t] = LA[offj + nb * 64 + t];
        c[t] = LV[offi + nb * 64 + t];
        d[t] = LV[offj + nb * 64 + t];
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        sa -= a[t] * b[t];
        sv -= c[t] * d[t];
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
  if (REUSE) {
    lp = M.lp;
  } else {
    // VariableSelectionPrior::logp (VariableSelectionPrior.cpp:271-285)
    double part = 0.0;
    for (int j = lane; j < p; j += WAVE) part += ch.gam[j] ? P.l1[j] : P.l0[j];
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
  wave_sync();
  // gather V_g, A_g (lower triangles, rows padded with zeros to a multiple of
  // 8) with all loads independent: element e <-> (m, n), n <= m
  const int kpad = (k + 7) & ~7;
  const int nelem = REUSE ? 0 : kpad * (kpad + 1) / 2;
  for (int e = lane; e < nelem; e += WAVE) {
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
  const int gm = (lane < k) ? ch.g[lane] : 0;
  const double bm = (lane < k) ? P.b[gm] : 0.0;
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
  const double r = (lane < k) ? ab + ch.xty[gm] * ch.sx : 0.0;
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
  for (int j = 0; j < k; ++j) {
    const double wj = bcast_u(x, j) * bcast_u(rdm, j);
    if (lane == j) x = wj;
    else if (lane > j && lane < k) x -= ch.Lv[bidx(lane, j)] * wj;
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
  for (int e = lane; e < nblk * 64; e += WAVE) {
    ch.Lv[e] = src[S.Lv + e];
    ch.La[e] = src[S.La + e];
  }
  if (lane < kpad) {
    ch.rdv[lane] = src[S.rdv + lane];
    ch.rda[lane] = src[S.rda + lane];
    ch.w[lane] = src[S.w + lane];
    ch.bg[lane] = src[S.bg + lane];
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
                                                  bool valid, StampCtx &sx) {
  const int p = ch.p, k = ch.k;
  Proposal out;
  out.logp = -BA_INF;
  out.slow = false;
  out.bad_ss = false;
  const bool add = valid && !ch.gam[j];
  const bool drop = valid && !add;
  const int kn = add ? k + 1 : k - 1;
  double lpn = -BA_INF;
  if (valid) {
    const double l1 = P.l1[j], l0 = P.l0[j];
    // log prior of the flipped model; -inf terms must not meet +inf
    if (add) lpn = (l1 == -BA_INF) ? -BA_INF : ((l0 == -BA_INF) ? -BA_INF : M.lp + (l1 - l0));
    else     lpn = (l0 == -BA_INF) ? -BA_INF : ((l1 == -BA_INF) ? -BA_INF : M.lp + (l0 - l1));
    if (P.max_model_size >= 0 && kn > P.max_model_size) lpn = -BA_INF;
  }
  const bool live = valid && (lpn > -BA_INF);
  const double bj = live ? P.b[j] : 0.0;
  const bool empty_after = live && drop && (kn == 0);
  const bool slow = live && !empty_after && (bj != 0.0);
  const bool fast = live && !empty_after && !slow;
  out.slow = slow;
  if (empty_after) {
    out.logp = ch.mode ? lpn : lpn - (0.5 * ch.DF - 1.0) * log(ch.ss0q);
  }
  const double vjj = (fast && add) ? P.V[(size_t)j * p + j] * ch.sv : 0.0;
  const double ajj = (fast && add) ? P.A[(size_t)j * p + j] * ch.sa : 0.0;
  const double xtyj = (fast && add) ? ch.xty[j] * ch.sx : 0.0;

  constexpr int KCAP = NB * 8;
  const SsvsScalarLayout S = ssvs_scalar_layout(KCAP);
  c_f64 *sc = ch.sc;

  double x[NB * 8];
  double nv = 0.0, dv = 0.0, na = 0.0, ab = 0.0;
  SUBSTAMP(sx, 1);
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
          const double v = (NAT ? Mat[(size_t)gm * p + j] : Mat[(size_t)j * p + gm]) * msc;
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
      if (((odd ? mO : mE) >> l) & 1ull) {
        zm = bcast_u(odd ? zO : zE, l);
        o += 2;
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
                                                lds_u16 *oth) {
  // uniform number pos + t (t = 0..p-2) picks the partner of i = p-1-t:
  // random_int_mt(rng, 0, i) (cpputil/shuffle.hpp:36-46); one Philox block
  // serves two consecutive steps
  const uint64_t b0 = pos >> 1, b1 = (pos + (uint64_t)(p - 2)) >> 1;
  for (uint64_t b = b0 + (uint64_t)tid; b <= b1; b += (uint64_t)nthreads) {
    double u[2];
    philox_pair(key, b, &u[0], &u[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long t = (long long)(2 * b + h) - (long long)pos;
      if (t >= 0 && t < p - 1) {
        const int i = p - 1 - (int)t;
        oth[i] = (uint16_t)(int)floor(0.0 + ((double)(i + 1) - 0.0) * u[h]);
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
    const Proposal pr = eval_proposal<NB, true>(P, ch, M, valid ? idx : 0, valid, sx);
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
    double u[R][2], ej[R][2];
    int jj[R][2], kd[R][2];
    bool val[R][2], acc[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int q = qb + 2 * (r * WAVE + lane) + h;
        val[r][h] = q >= i0 && q < nflips;
        jj[r][h] = val[r][h] ? (int)ch.perm[q] : 0;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        ej[r][h] = ch.tab_lp[jj[r][h]];
        kd[r][h] = ch.tab_kind[jj[r][h]];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
      philox_pair(key, blk0 + (uint64_t)(r * WAVE + lane), &u[r][0], &u[r][1]);
    int f = 1 << 20, fr = 0;  // first stop: 2 lane + half within sub-round fr
#pragma unroll
    for (int r = 0; r < R; ++r) {
      unsigned long long mstop[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool special = kd[r][h] != 0;
        acc[r][h] = val[r][h] && !special && !(u[r][h] > ej[r][h]);
        mstop[h] = __ballot(val[r][h] && (special || acc[r][h]));
      }
      if (f == (1 << 20)) {
        const int f0 = mstop[0] ? 2 * (__ffsll((long long)mstop[0]) - 1) : 1 << 20;
        const int f1 = mstop[1] ? 2 * (__ffsll((long long)mstop[1]) - 1) + 1 : 1 << 20;
        const int fm = f0 < f1 ? f0 : f1;
        if (fm < (1 << 20)) { f = fm; fr = r; }
      }
    }
    const int fkey = (f == (1 << 20)) ? (1 << 30) : fr * 2 * WAVE + f;  // order within the round
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // |u / E - 1|, to first order the distance |log u - (logp' - logp)|
        const int me = r * 2 * WAVE + 2 * lane + h;
        const bool counted = val[r][h] && (me < fkey || (me == fkey && acc[r][h])) && ej[r][h] > 0.0;
        const double mg = fabs(u[r][h] - ej[r][h]) * __builtin_amdgcn_rcp(ej[r][h]);
        if (counted) lane_margin = fmin(lane_margin, mg);
      }
    }
    if (f == (1 << 20)) {
      i0 = qb + 2 * R * WAVE;  // first position of the next round
      continue;
    }
    const int fl = f >> 1, fh = f & 1;
    int sj = 0, sk = 0;
    double su = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r == fr) {
        sj = fh ? jj[r][1] : jj[r][0];
        sk = fh ? kd[r][1] : kd[r][0];
        su = fh ? u[r][1] : u[r][0];
      }
    }
    out.spos = qb + fr * 2 * WAVE + f;
    out.j = __builtin_amdgcn_readlane(sj, fl);
    out.kind = __builtin_amdgcn_readlane(sk, fl);
    out.logu = log(bcast_u(su, fl));
    break;
  }
  out.margin = wave_min(lane_margin);
}

template <int NB, int W, int WPE>
__global__ __launch_bounds__(64 * W, WPE) void ssvs_sweep_kernel(SsvsParams P,
                                                            int nsweeps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int chain = (int)blockIdx.x + P.chain_first;
  const int lane = threadIdx.x & (WAVE - 1);
  const int wave = threadIdx.x >> 6;
  const int p = P.p;
  if ((int)blockIdx.x >= P.chain_count) return;
  if (P.status[chain] != CHAIN_OK) {
    // a chain waiting for a larger-capacity kernel (or in error) just books
    // the sweeps it is owed
    if (threadIdx.x == 0) {
      P.todo[chain] += nsweeps;
      if (P.ran) P.ran[chain] = 0;
    }
    return;
  }
  // sweeps to run now: this launch's plus what earlier launches owe, capped by
  // run_limit (the rest stays owed)
  int owed_after = 0;
  {
    int total = nsweeps + P.todo[chain];
    if (P.run_limit > 0 && total > P.run_limit) {
      owed_after = total - P.run_limit;
      total = P.run_limit;
    }
    if (total == 0) {
      if (threadIdx.x == 0 && P.ran) P.ran[chain] = 0;
      return;  // (every wave of the workgroup takes this exit)
    }
    nsweeps = total;
  }

  constexpr int KCAP = NB * 8;
  const SsvsLds lay = ssvs_lds_layout(p, KCAP);
  Chain ch;
  ch.lane = lane;
  ch.p = p;
  ch.k = 0;
  bind_lds(ch, smem, lay);
  lds_f64 *ctl = to_lds<double>(smem + lay.ctrl);
  ch.xty = P.xty + (size_t)chain * P.xty_stride;
  bind_slot(ch, P, chain, 0);
  ch.sc = (c_f64 *)(unsigned long long)ch.sc_store;
  const double yty = P.yty[(size_t)chain * P.suf_stride];
  const double nobs = P.nobs[(size_t)chain * P.suf_stride];
  ch.DF = nobs + P.prior_df;
  ch.ss0q = P.prior_ss + yty;
  ch.mode = P.mode;
  ch.sv = ch.sa = ch.sx = 1.0;
  if (P.mode) {
    // SpikeSlabSampler works given this chain's sigma^2
    const double inv = 1.0 / P.sigsq[chain];
    ch.sx = inv;
    if (P.slab_scales) { ch.sv = inv; ch.sa = inv; }
  }
  const PhiloxKey key{P.seed_lo, P.seed_hi,
                      (uint32_t)(P.chain_offset + chain), P.stream};

  StampCtx sx_unused;
  sx_unused.last = 0;
  if (W > 1 && wave != 0) {
    // ---- helper waves: serve the master's commands ---------------------------
    sx_unused.last = (long long)__builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) sx_unused.ph[i] = 0.0;
    for (;;) {
      HSTAMP(sx_unused, 7);
      __syncthreads();
      HSTAMP(sx_unused, 5);
      const int cmd = (int)ctl[CT_CMD];
      if (cmd == CMD_EXIT) break;
      const uint64_t upos = ((AS_LDS const uint64_t *)(ctl + CT_POS))[0];
      if (cmd == CMD_EVAL) {
        Model M;
        M.logp = ctl[CT_LOGP]; M.lp = ctl[CT_LP]; M.ldv = ctl[CT_LDV];
        M.lda = ctl[CT_LDA]; M.Q = ctl[CT_Q]; M.c = ctl[CT_C];
        M.SS = 0; M.pd = true; M.bad = 0;
        ch.k = (int)ctl[CT_K];
        ch.perm = to_lds<uint16_t>(smem + (((int)ctl[CT_PERMSEL]) ? lay.perm1 : lay.perm0));
        bind_slot(ch, P, chain, (int)ctl[CT_CUR]);
        unsigned long long a = (unsigned long long)ch.sc_store;
        asm volatile("" : "+s"(a) : : "memory");
        ch.sc = (c_f64 *)a;
        eval_share<NB>(P, ch, M, key, upos, (int)ctl[CT_NFLIPS], (int)ctl[CT_I0],
                       (int)ctl[CT_EVMODE], wave, ctl, sx_unused);
        HSTAMP(sx_unused, 6);
      } else if (cmd == CMD_DECIDE || cmd == CMD_SHUFFLE_DECIDE) {
        if (wave == 1) {
          const int sel = (int)ctl[CT_PERMSEL];
          bind_slot(ch, P, chain, (int)ctl[CT_CUR]);
          ch.perm = to_lds<uint16_t>(smem + (sel ? lay.perm1 : lay.perm0));
          ch.perm_alt = to_lds<uint16_t>(smem + (sel ? lay.perm0 : lay.perm1));
          uint64_t fpos = upos;
          if (cmd == CMD_SHUFFLE_DECIDE) {
            // the whole permutation side of a quiet sweep: shuffle(indx) from
            // stream position upos, then the walk over the new order
            shuffle_targets(key, upos, p, lane, WAVE, ch.oth);
            wave_sync();
            parallel_shuffle(ch, sx_unused);
            fpos = upos + (uint64_t)(p - 1);
          }
          DecideResult dr;
          decide_walk(ch, key, fpos, (int)ctl[CT_I0], (int)ctl[CT_NFLIPS], dr);
          HSTAMP(sx_unused, 4);
          if (lane == 0) {
            lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * 1;
            sl[SL_F] = (double)dr.spos;
            sl[SL_J] = (double)dr.j;
            sl[SL_KIND] = (double)dr.kind;
            sl[SL_LOGU] = dr.logu;
            sl[SL_MARGIN] = dr.margin;
          }
        }
      } else {  // CMD_UNIF: this wave's share of the shuffle uniforms
        if (p > 1) shuffle_targets(key, upos, p, threadIdx.x, WAVE * W, ch.oth);
      }
      HSTAMP(sx_unused, 7);
      __syncthreads();
      HSTAMP(sx_unused, 5);
    }
#if defined(BA_STAMPS) && defined(BA_STAMPS4)
    if (wave == 1 && lane == 0) {
      double *a = P.acc + (size_t)chain * ACC_COUNT;
      for (int i = 0; i < 8; ++i) a[ACC_PHASE0 + i] += sx_unused.ph[i];
    }
#endif
    return;
  }

  // ---- master wave -------------------------------------------------------------
  uint8_t *g_gamma = P.gamma + (size_t)chain * p;
  uint16_t *g_perm = P.perm + (size_t)chain * p;
  int k = 0;
  int status = CHAIN_OK;
  for (int base = 0; base < p; base += WAVE) {
    const int j = base + lane;
    const int inc = (j < p) ? g_gamma[j] : 0;
    if (j < p) {
      ch.gam[j] = (uint8_t)inc;
      ch.perm[j] = g_perm[j];
      ch.nbr[j] = (uint8_t)((P.cm_start != nullptr) && (P.cm_start[j + 1] > P.cm_start[j]));
    }
    const unsigned long long mask = __ballot(inc != 0);
    const int slot = k + __popcll(mask & ((1ull << lane) - 1ull));
    if (inc && slot < KCAP) ch.g[slot] = (uint16_t)j;
    k += __popcll(mask);
  }
  if (k > KCAP) status = CHAIN_MODEL_TOO_LARGE;
  ch.k = k;
  wave_sync();
  bool aborted = false;      // stopped inside a sweep: restore its start
  int kmax = k;
  int trace_at = P.trace_idx ? P.trace_idx[chain] : 0;

  uint64_t pos = uni((uint64_t)P.rng_pos[chain]);
  int failures = uni((int)P.failures[chain]);
  double sigsq = uni((double)P.sigsq[chain]);
  double beta_m = 0.0;  // lane m: coefficient of variable gprev after the last draw
  int gprev = 0, kprev = 0;
  bool beta_valid = false;

  // The launch's accumulators live in LDS, not in registers (the master's
  // registers are needed for what it touches every few instructions): scalars
  // in the control block (slots CT_ACC + ACC_*), the per-variable summaries of
  // the standing model -- lane m <-> variable sum_g[m] -- in four lane arrays.
  lds_f64 *sum_b = to_lds<double>(smem + lay.park + 512);    // sum of beta
  lds_f64 *sum_b2 = to_lds<double>(smem + lay.park + 1024);  // sum of beta^2
  AS_LDS int *sum_n = to_lds<int>(smem + lay.park + 1536);   // sweeps counted
  AS_LDS int *sum_g = to_lds<int>(smem + lay.park + 1792);   // variable (-1: none)
  sum_b[lane] = 0.0; sum_b2[lane] = 0.0; sum_n[lane] = 0; sum_g[lane] = -1;
  if (lane < 8) ctl[CT_ACC + lane] = (lane == ACC_MIN_MARGIN) ? BA_INF : 0.0;
  wave_sync();
#define ACC_ADD(slot, x) do { if (lane == 0) ctl[CT_ACC + (slot)] += (double)(x); } while (0)
#define ACC_MIN(x) do { if (lane == 0) ctl[CT_ACC + ACC_MIN_MARGIN] = fmin(ctl[CT_ACC + ACC_MIN_MARGIN], (x)); } while (0)
  int done = 0;

  const int nflips = P.max_flips;  // already min(max_nflips_, p)
  STAMP_DECL;
  StampCtx sx;
  sx.last = (long long)__builtin_readcyclecounter();
  for (int i = 0; i < 8; ++i) sx.ph[i] = 0.0;

  // The factors / scalars of the current model are built once per launch and
  // after every change of gamma; they stay valid across sweeps (they depend on
  // gamma only, not on sigma or beta).
  Model M;
  M.bad = 0; M.pd = true; M.logp = 0; M.lp = 0; M.ldv = 0; M.lda = 0; M.Q = 0; M.c = 0; M.SS = 0;
  Pending pe;
  pe.kind = EV_INIT; pe.f1 = pe.f2 = -1; pe.lu = 0; pe.lfw = pe.lrev = 0; pe.check_legal = false;
  enum { PH_BEGIN, PH_FLIPS, PH_SWAP, PH_TAIL, PH_JOIN, PH_COMMIT };
  int phase = PH_BEGIN, sweep = 0, i0 = 0;
  bool model_checked = false;  // legality of the start is checked in sweep 0
  int perm_sel = 0;            // which LDS buffer holds the current permutation
  // walking mode of the current sweep (see the kernel's header comment)
  bool use_table = false;     // this sweep decides by table look-up
  // the table of the chain's last launch is still good when nothing but
  // sweeps happened since (the host clears table_keep otherwise)
  // which of the chain's two (table, model block) slots is in use, and what the
  // other one holds: the model one flip (of variable other_var) away, or nothing
  int cur = 0, other_var = -1;
  bool other_ok = false, table_valid_other = false;
  const int model_tag_in = P.model_keep ? P.model_tag[chain] : 0;
  const bool model_kept = (model_tag_in & 0xff) == KCAP && status == CHAIN_OK;
  if (model_kept) cur = (model_tag_in >> 8) & 1;
  bind_slot(ch, P, chain, cur);
  bool table_valid = P.table_keep && model_kept && P.table_tag[chain] == model_tag_in;
  int fill_j = 0;             // next variable of a fill in progress
  // Quiet sweeps fork: wave 1 shuffles and walks the table while the master
  // runs the sweep's tail (swap proposal, sigma, beta) on the assumption that
  // no flip will be accepted -- the tail's stream position is known up front
  // (shuffle and flips consume p - 1 + nflips numbers whatever happens).  At
  // the join a stop in the walk rolls the tail back.
  bool spec = false;          // a forked walk is outstanding
  bool have_dr = false;       // wave 1's slot holds a walk result not yet handled
  int after_join = PH_COMMIT, spec_status = CHAIN_OK;
  // Between a join and the next fork wave 1 has nothing to do, so the master
  // keeps that stretch short: a joined sweep's summaries are committed AFTER the
  // next sweep has been forked, and the sweep-start copy of gamma is skipped
  // while gamma has not moved.
  bool commit_pending = false, gam0_fresh = false;
  // (what a roll-back restores is parked in LDS, not in registers: park[lane] =
  // beta_m, control-block slots CT_ROLL.. = sigma^2, failures, beta_valid)
  lds_f64 *park = to_lds<double>(smem + lay.park);
  int stops_prev = table_valid ? 0 : (1 << 20), stops_now = 0;
  uint64_t flip_pos = 0, pos0 = pos;
  WinRng rng;
  rng.init(key, lane, pos);

  // The chain's model block of the last launch is still this model (nothing but
  // sweeps happened since): take the factors from there instead of factoring.
  if (model_kept) {
    restore_model<NB>(ch);
    {
      const SsvsScalarLayout S = ssvs_scalar_layout(KCAP);
      const double *sc = ch.sc_store + S.scal;
      M.logp = sc[0]; M.lp = sc[1]; M.ldv = sc[2]; M.lda = sc[3];
      M.Q = sc[4]; M.c = sc[5]; M.SS = sc[6]; M.pd = sc[7] != 0.0;
      unsigned long long u = (unsigned long long)ch.sc_store;
      asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(u) : : "memory");
      ch.sc = (c_f64 *)u;
    }
    if (P.suf_changed) {
      refactor<true>(P, ch, M, sx);
      if (M.bad) status = M.bad;
      else publish_model<NB>(ch, M);
    }
    if (lane == 0) {
      ctl[CT_LOGP] = M.logp; ctl[CT_LP] = M.lp; ctl[CT_LDV] = M.ldv;
      ctl[CT_LDA] = M.lda; ctl[CT_Q] = M.Q; ctl[CT_C] = M.c;
    }
    wave_sync();
    pe.kind = EV_NONE;
  }

  while (status == CHAIN_OK) {
    if (pe.kind != EV_NONE && !spec) {
      // ---- the one place where a model is (re)built (a swap proposed by a
      // tail running ahead waits for the join) --------------------------
      // (outside this block only logp, SS and pd of the model live in registers;
      // the scalars the evaluations need have their home in the control block)
      if (pe.kind == EV_FORCE && pe.f2 < 0 && other_ok && pe.f1 == other_var) {
        gam0_fresh = false;
        // The accepted flip leads to the model the other slot still holds (a
        // variable leaving again, or coming back): its factors, scalars and
        // table are there -- bitwise what a rebuild would compute.
        apply_flip(ch, pe.f1);
        cur ^= 1;
        bind_slot(ch, P, chain, cur);
        restore_model<NB>(ch);
        {
          const SsvsScalarLayout S = ssvs_scalar_layout(KCAP);
          const double *sc = ch.sc_store + S.scal;
          M.logp = sc[0]; M.lp = sc[1]; M.ldv = sc[2]; M.lda = sc[3];
          M.Q = sc[4]; M.c = sc[5]; M.SS = sc[6]; M.pd = sc[7] != 0.0;
          unsigned long long u = (unsigned long long)ch.sc_store;
          asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(u) : : "memory");
          ch.sc = (c_f64 *)u;
        }
        if (lane == 0) {
          ctl[CT_LOGP] = M.logp; ctl[CT_LP] = M.lp; ctl[CT_LDV] = M.ldv;
          ctl[CT_LDA] = M.lda; ctl[CT_Q] = M.Q; ctl[CT_C] = M.c;
        }
        wave_sync();
        {  // the slot left behind keeps the model just left: one flip of the same variable away
          const bool t = table_valid;
          table_valid = table_valid_other;
          table_valid_other = t;
        }
        ACC_ADD(ACC_ACCEPTS, 1);
#ifndef BA_STAMPS
        ACC_ADD(ACC_SLOT_HITS, 1);
#endif
        if (!M.pd) status = CHAIN_NOT_PD;
        pe.kind = EV_NONE;
        pe.f1 = pe.f2 = -1;
        pe.lfw = pe.lrev = 0.0;
        pe.check_legal = false;
        STAMP(2);
        continue;
      }
      gam0_fresh = false;
      Model keep = M;
      keep.lp = ctl[CT_LP]; keep.ldv = ctl[CT_LDV]; keep.lda = ctl[CT_LDA];
      keep.Q = ctl[CT_Q]; keep.c = ctl[CT_C];
      if (pe.f1 >= 0) apply_flip(ch, pe.f1);
      if (pe.f2 >= 0) apply_flip(ch, pe.f2);
      bool rejected = false;
      {
        Model Mn;
        refactor<false>(P, ch, Mn, sx);
        if (Mn.bad) {
          status = Mn.bad;
        } else {
          bool acc = true;
          if (pe.kind == EV_TRY_GE || pe.kind == EV_TRY_LT) {
            const double d = (Mn.logp - pe.lfw) - (keep.logp - pe.lrev);
            if (Mn.logp > -BA_INF) ACC_MIN(fabs(pe.lu - d));
            acc = (pe.kind == EV_TRY_GE) ? !(pe.lu > d) : (pe.lu < d);
          }
          if (acc) {
            M = Mn;
            M.logp = uni(M.logp);
            M.SS = uni(M.SS);
            // (the launch's first build is the old model unless make_valid
            // changed gamma)
            if (pe.kind != EV_INIT || pe.check_legal) {
              // a new model: it goes to the other slot, the one in use keeps the
              // model being left (reachable again by one flip if one flip led here)
              table_valid_other = table_valid;
              table_valid = false;
              other_ok = (pe.kind != EV_INIT) && pe.f1 >= 0 && pe.f2 < 0;
              other_var = pe.f1;
              cur ^= 1;
              bind_slot(ch, P, chain, cur);
            }
            if (pe.kind != EV_INIT) ACC_ADD(ACC_ACCEPTS, 1);
          } else {
            // rejected: gamma back, and the old factors from the chain's block
            if (pe.f2 >= 0) apply_flip(ch, pe.f2);
            if (pe.f1 >= 0) apply_flip(ch, pe.f1);
            restore_model<NB>(ch);
            M = keep;
            rejected = true;
          }
        }
      }
      if (status == CHAIN_OK) {
        if (!rejected) {
          publish_model<NB>(ch, M);
          if (lane == 0) {
            ctl[CT_LOGP] = M.logp; ctl[CT_LP] = M.lp; ctl[CT_LDV] = M.ldv;
            ctl[CT_LDA] = M.lda; ctl[CT_Q] = M.Q; ctl[CT_C] = M.c;
          }
          wave_sync();
        }
        if (pe.kind == EV_FORCE && !M.pd) status = CHAIN_NOT_PD;
        if (pe.kind == EV_INIT && pe.check_legal &&
            !(M.logp > -BA_INF && M.logp < BA_INF))
          status = CHAIN_ILLEGAL_START;
      }
      pe.kind = EV_NONE;
      pe.f1 = pe.f2 = -1;
      pe.lfw = pe.lrev = 0.0;
      pe.check_legal = false;
      STAMP(2);
      continue;
    }

    if (phase == PH_BEGIN) {
      if (sweep + (commit_pending ? 1 : 0) >= nsweeps) {
        if (!commit_pending) break;
        phase = PH_COMMIT;  // the last sweep's summaries, then out
        continue;
      }
      {
        const bool ut = (P.walk_policy != 0) && (stops_prev <= 1 || P.walk_policy == 2);
        const bool mc = model_checked || (M.logp > -BA_INF && M.logp < BA_INF);
        const bool will_fork = nflips > 0 && W > 1 && ut && table_valid && mc && p > 1 && P.walk_policy != 3;
        if (commit_pending && !will_fork) {
          phase = PH_COMMIT;  // nothing to overlap with: commit first
          continue;
        }
      }
      STAMP(7);
      TSTAMP(sx, 7);
      if (nflips > 0) {
        // remember the sweep's starting point (restored if the chain has to
        // stop inside this sweep for lack of model capacity)
        if (!gam0_fresh || P.mode) {
          for (int j = lane; j < p; j += WAVE) {
            ch.gam0[j] = ch.gam[j];
            // SpikeSlabSampler shuffles a fresh identity permutation every call
            // (SpikeSlabSampler.cpp:48-57); BregVsSampler's indx persists
            if (P.mode) ch.perm[j] = (uint16_t)j;
          }
          gam0_fresh = true;
        }
        TSTAMP(sx, 1);
        pos0 = pos;
        if (!model_checked && M.logp > -BA_INF && M.logp < BA_INF) model_checked = true;
        use_table = (P.walk_policy != 0) && (stops_prev <= 1 || P.walk_policy == 2);
        stops_prev = stops_now;
        stops_now = 0;
        if (W > 1 && use_table && table_valid && model_checked && p > 1 && P.walk_policy != 3) {
          // ---- fork: wave 1 takes the permutation side of the sweep
          wave_sync();
          if (lane == 0) {
            ctl[CT_CMD] = (double)CMD_SHUFFLE_DECIDE;
            ctl[CT_I0] = 0.0;
            ctl[CT_NFLIPS] = (double)nflips;
            ctl[CT_PERMSEL] = (double)perm_sel;
            ctl[CT_CUR] = (double)cur;
            ((AS_LDS uint64_t *)(ctl + CT_POS))[0] = pos;
          }
          __syncthreads();
          {  // the shuffled order will be in the other buffer
            lds_u16 *tmp = ch.perm;
            ch.perm = ch.perm_alt;
            ch.perm_alt = tmp;
            perm_sel ^= 1;
          }
          flip_pos = pos + (uint64_t)(p - 1);
          pos = flip_pos + (uint64_t)nflips;
          TSTAMP(sx, 2);
          spec = true;
          spec_status = CHAIN_OK;
          park[lane] = beta_m;
          if (lane == 0) {
            ctl[CT_ROLL + 0] = sigsq;
            ctl[CT_ROLL + 1] = (double)failures;
            ctl[CT_ROLL + 2] = beta_valid ? 1.0 : 0.0;
          }
          i0 = 0;
          phase = commit_pending ? PH_COMMIT : PH_SWAP;  // (the joined sweep's summaries ride along)
          STAMP(1);
          continue;
        }
        // ---- shuffle(indx): cpputil/shuffle.hpp:36-46, in place on the
        // persistent permutation.  Uniform t (t = 0..p-2) belongs to i = p-1-t.
        if (W > 1) {
          if (lane == 0) {
            ctl[CT_CMD] = (double)CMD_UNIF;
            ((AS_LDS uint64_t *)(ctl + CT_POS))[0] = pos;
          }
          __syncthreads();
        }
        if (p > 1) shuffle_targets(key, pos, p, threadIdx.x, WAVE * W, ch.oth);
        if (W > 1) __syncthreads(); else wave_sync();
        STAMP(0);
        if (p > 1) { parallel_shuffle(ch, sx); perm_sel ^= 1; }
        flip_pos = pos + (uint64_t)(p > 0 ? p - 1 : 0);
        pos = flip_pos + (uint64_t)nflips;
        STAMP(1);
        if (!model_checked) {
          model_checked = true;
          if (!(M.logp > -BA_INF && M.logp < BA_INF)) {
            // VariableSelectionPrior::make_valid, VariableSelectionPrior.cpp:287-300
            for (int j = 0; j < p; ++j) {
              const double pj = P.pi[j];
              const bool inc = ch.gam[j];
              if ((pj <= 0.0 && inc) || (pj >= 1.0 && !inc)) {
                if (!inc && ch.k >= KCAP) { status = CHAIN_MODEL_TOO_LARGE; aborted = true; break; }
                apply_flip(ch, j);
              }
            }
            pe.kind = EV_INIT;
            pe.check_legal = true;
          }
        }
      }
      i0 = 0;
      phase = PH_FLIPS;
      continue;
    }

    if (phase == PH_FLIPS) {
      if (nflips == 0 || i0 >= nflips) {
        phase = PH_SWAP;
        continue;
      }
      // ---- Metropolised flips, 64 * W proposals per round
      if (use_table && table_valid) {
        // ---- decisions by table look-up, up to the first stop.  With helper
        // waves the walk is wave 1's job (the master's registers are full of
        // chain state; wave 1 has none), otherwise the master's own.
        DecideResult dr;
        if (W > 1) {
          if (!have_dr) {
          if (lane == 0) {
            ctl[CT_CMD] = (double)CMD_DECIDE;
            ctl[CT_I0] = (double)i0;
            ctl[CT_NFLIPS] = (double)nflips;
            ctl[CT_PERMSEL] = (double)perm_sel;
            ctl[CT_CUR] = (double)cur;
            ((AS_LDS uint64_t *)(ctl + CT_POS))[0] = flip_pos;
          }
          __syncthreads();
          __syncthreads();
          }
          have_dr = false;
          const lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * 1;
          dr.spos = uni((int)sl[SL_F]);
          dr.j = uni((int)sl[SL_J]);
          dr.kind = uni((int)sl[SL_KIND]);
          dr.logu = uni((double)sl[SL_LOGU]);
          dr.margin = uni((double)sl[SL_MARGIN]);
        } else {
          decide_walk(ch, key, flip_pos, i0, nflips, dr);
        }
        ACC_MIN(dr.margin);
        STAMP(3);
        if (dr.spos < 0) {  // walked to the end without a stop
          ACC_ADD(ACC_PROPOSALS, nflips - i0);
          i0 = nflips;
          continue;
        }
        ACC_ADD(ACC_PROPOSALS, dr.spos + 1 - i0);
        ++stops_now;
        i0 = dr.spos + 1;
        if (dr.kind == STOP_BAD) {
          status = CHAIN_NEGATIVE_SS;
          break;
        }
        if (!ch.gam[dr.j] && ch.k >= KCAP) {
          status = CHAIN_MODEL_TOO_LARGE;
          aborted = true;
          break;
        }
        pe.f1 = dr.j;
        if (dr.kind == 0) {
          pe.kind = EV_FORCE;
        } else {
          pe.kind = EV_TRY_GE;
          pe.lu = dr.logu;
        }
        continue;
      }
      int evmode = EVM_BATCH, base = i0;
      if (use_table) {
        evmode = EVM_FILL;   // (re)build the table for the current model
        base = fill_j;
      }
      if (W > 1) {
        if (lane == 0) {
          ctl[CT_CMD] = (double)CMD_EVAL;
          ctl[CT_K] = (double)ch.k;
          ctl[CT_I0] = (double)base;
          ctl[CT_NFLIPS] = (double)nflips;
          ctl[CT_PERMSEL] = (double)perm_sel;
            ctl[CT_CUR] = (double)cur;
          ctl[CT_EVMODE] = (double)evmode;
          ((AS_LDS uint64_t *)(ctl + CT_POS))[0] = flip_pos;
        }
        __syncthreads();
      }
      {
        Model Me;
        Me.logp = M.logp; Me.lp = ctl[CT_LP]; Me.ldv = ctl[CT_LDV]; Me.lda = ctl[CT_LDA];
        Me.Q = ctl[CT_Q]; Me.c = ctl[CT_C]; Me.SS = 0; Me.pd = true; Me.bad = 0;
        eval_share<NB>(P, ch, Me, key, flip_pos, nflips, base, evmode, 0, ctl, sx);
      }
      if (W > 1) __syncthreads(); else wave_sync();
      if (evmode == EVM_FILL) {
        fill_j += WAVE * W;
        if (fill_j >= p) {
          fill_j = 0;
          table_valid = true;
        }
        STAMP(3);
        continue;
      }
      // first stop over the whole round, in sweep order
      int wstop = -1, spos = 0;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * w;
        if (wstop < 0) {
          ACC_MIN(sl[SL_MARGIN]);
          const int fw = uni((int)sl[SL_F]);
          if (fw >= 0) { wstop = w; spos = fw; }
        }
      }
      STAMP(3);
      if (wstop < 0) {
        const int n = (nflips - i0 < WAVE * W) ? (nflips - i0) : WAVE * W;
        ACC_ADD(ACC_PROPOSALS, n);
        i0 += WAVE * W;
        continue;
      }
      const lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * wstop;
      const int jf = uni((int)sl[SL_J]);
      const int kind = uni((int)sl[SL_KIND]);
      const int nprop = spos + 1 - i0;
      ACC_ADD(ACC_PROPOSALS, nprop);
      ++stops_now;
      if (kind == STOP_BAD) {
        status = CHAIN_NEGATIVE_SS;
        break;
      }
      if (!ch.gam[jf] && ch.k >= KCAP) {
        // the candidate cannot be held in LDS; if it is a sure rejection that
        // is fine, but we cannot tell without evaluating it
        status = CHAIN_MODEL_TOO_LARGE;
        aborted = true;
        break;
      }
      pe.f1 = jf;
      if (kind == STOP_ACCEPT) {
        pe.kind = EV_FORCE;  // accepted on the fast path: move to the new model
      } else {
        pe.kind = EV_TRY_GE;  // exact path: evaluate the flipped model
        pe.lu = uni((double)sl[SL_LOGU]);
      }
      i0 += nprop;
      continue;
    }

    if (phase == PH_SWAP) {
      TSTAMP(sx, 7);
      rng.set_pos(pos);
      if (nflips > 0) propose_swap(P, ch, rng, pe, &status);
      pos = rng.get_pos();
      phase = PH_TAIL;
      if (spec && (pe.kind != EV_NONE || status != CHAIN_OK)) {
        // a swap was proposed (or the proposal failed): its evaluation has to
        // wait for the walk
        spec_status = status;
        status = CHAIN_OK;
        after_join = PH_TAIL;
        phase = PH_JOIN;
      }
      STAMP(4);
      TSTAMP(sx, 3);
      continue;
    }

    if (phase == PH_JOIN) {
      __syncthreads();
      spec = false;
      const lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * 1;
      if ((int)sl[SL_F] < 0) {
        // no stop: the sweep's flips are all rejected and what ran ahead stands
        ACC_MIN(sl[SL_MARGIN]);
        ACC_ADD(ACC_PROPOSALS, nflips);
        status = spec_status;
        phase = after_join;
        if (after_join == PH_COMMIT && status == CHAIN_OK) {
          commit_pending = true;  // committed after the next fork
          phase = PH_BEGIN;
        }
      } else {
        // roll the tail back and handle the stop
        pos = flip_pos + (uint64_t)nflips;
        // (gprev / kprev are only set at a commit)
        beta_m = park[lane];
        sigsq = ctl[CT_ROLL + 0];
        failures = (int)ctl[CT_ROLL + 1];
        beta_valid = ctl[CT_ROLL + 2] != 0.0;
        pe.kind = EV_NONE; pe.f1 = pe.f2 = -1; pe.lfw = pe.lrev = 0.0; pe.check_legal = false;
        have_dr = true;
        phase = PH_FLIPS;
      }
      STAMP(3);
      continue;
    }

    if (phase == PH_COMMIT) {
      // ---- summaries
      TSTAMP(sx, 7);
      k = ch.k;
      gprev = (lane < k) ? (int)ch.g[lane] : 0;
      kprev = k;
      kmax = k > kmax ? k : kmax;
      // inclusion counts and coefficient moments pile up on chip while the
      // model stands still (lane m <-> variable sum_g[m]) and go to HBM when it
      // moves
      {
        const int gnow = (lane < k) ? gprev : -1;
        const int gold = sum_g[lane];
        if (__any(gnow != gold)) {
          const int n_old = sum_n[lane];
          if (gold >= 0 && n_old) {
            const size_t o = (size_t)chain * p + gold;
            P.inc_count[o] += (unsigned)n_old;
            P.beta_sum[o] += sum_b[lane];
            P.beta_sumsq[o] += sum_b2[lane];
          }
          sum_g[lane] = gnow; sum_n[lane] = 0; sum_b[lane] = 0.0; sum_b2[lane] = 0.0;
        }
        if (lane < k) {
          sum_n[lane] += 1;
          if (beta_valid) { sum_b[lane] += beta_m; sum_b2[lane] += beta_m * beta_m; }
        }
      }
      ACC_ADD(ACC_SIGSQ, sigsq);
      ACC_ADD(ACC_SIGSQ2, sigsq * sigsq);
      ACC_ADD(ACC_K, k);
      if (P.trace_sigsq && trace_at + sweep < P.trace_stride) {
        const size_t o = (size_t)chain * P.trace_stride + trace_at + sweep;
        if (lane == 0) {
          P.trace_sigsq[o] = sigsq;
          P.trace_logp[o] = M.logp;
          P.trace_k[o] = (double)k;
        }
        if (P.rec_idx && lane < k) {  // the sweep's draw itself (recording step)
          P.rec_idx[o * P.rec_cap + lane] = (uint16_t)gprev;
          P.rec_beta[o * P.rec_cap + lane] = beta_valid ? beta_m : 0.0;
        }
      }
      ++done;
      ++sweep;
      commit_pending = false;
      phase = spec ? PH_SWAP : PH_BEGIN;  // (spec: this was the previous sweep's commit, riding on a fork)
      TSTAMP(sx, 0);
      continue;
    }

    // ---- PH_TAIL: sigma, beta, summaries
    TSTAMP(sx, 7);
    k = ch.k;
    rng.set_pos(pos);
    // draw_sigma (BregVsSampler.cpp:313-324)
    if (P.draw_sigma) {
      int bad = 0;
      const double DF = (k == 0) ? ch.DF : ((ch.DF - P.prior_df) + P.prior_df);
      const double SS = (k == 0) ? ch.ss0q : ((M.SS - P.prior_ss) + P.prior_ss);
      sigsq = uni(d_draw_variance(rng, DF, SS, P.sigma_max, &bad));
      if (bad) {
        if (!spec) { status = CHAIN_RNG_BRANCH; break; }
        spec_status = CHAIN_RNG_BRANCH; after_join = PH_COMMIT; phase = PH_JOIN;
        continue;
      }
    }
    pos = uni(rng.get_pos());
    STAMP(5);
    TSTAMP(sx, 4);
    // draw_beta (BregVsSampler.cpp:326-351)
    if (P.draw_beta && k > 0) {
      if (!M.pd) {
        ++failures;
        if (!spec) { status = CHAIN_NOT_PD; break; }
        spec_status = CHAIN_NOT_PD; after_join = PH_COMMIT; phase = PH_JOIN;
        continue;
      }
      failures = 0;
      // k standard normals in stream order (distributions/mvn.cpp:114-122),
      // lane m keeps z_m
      const double z = draw_normals(rng, k);
      pos = uni(rng.get_pos());
      TSTAMP(sx, 5);
      // beta = L^{-T}(w + sigma z): chol(V / sigma^2) = L / sigma
      // (SpikeSlabSampler: rmvn_ivar_mt with the sigma-scaled precision itself)
      const double sigma = P.mode ? 1.0 : sqrt(sigsq);
      double y = (lane < k) ? ch.w[lane] + sigma * z : 0.0;
      const double rdm = (lane < k) ? ch.rdv[lane] : 0.0;
      // column sweep of the back substitution; row i of L is fetched one step
      // ahead of its use
      double lrow = (k > 0 && lane < k - 1) ? ch.Lv[bidx(k - 1, lane)] : 0.0;
      for (int i = k - 1; i >= 0; --i) {
        const double lcur = lrow;
        if (i > 0) lrow = (lane < i - 1) ? ch.Lv[bidx(i - 1, lane)] : 0.0;
        const double xi = bcast_u(y * rdm, i);
        if (lane == i) y = xi;
        else if (lane < i) y -= lcur * xi;
      }
      beta_m = y;
      beta_valid = true;
      TSTAMP(sx, 6);
    } else if (P.draw_beta) {
      beta_valid = true;  // empty model: all coefficients zero
    }
    STAMP(6);
    after_join = PH_COMMIT;
    phase = spec ? PH_JOIN : PH_COMMIT;
  }

  // release the helper waves
  if (W > 1) {
    if (lane == 0) ctl[CT_CMD] = (double)CMD_EXIT;
    __syncthreads();
  }

  wave_sync();
  if (sum_g[lane] >= 0 && sum_n[lane]) {
    const size_t o = (size_t)chain * p + sum_g[lane];
    P.inc_count[o] += (unsigned)sum_n[lane];
    P.beta_sum[o] += sum_b[lane];
    P.beta_sumsq[o] += sum_b2[lane];
  }
  // ---- write the chain back (an aborted sweep leaves no trace: gamma, the
  // permutation and the stream position go back to the sweep's start; sigma,
  // beta are those of the last complete sweep anyway)
  wave_sync();
  {
    const lds_u8 *gsrc = aborted ? ch.gam0 : ch.gam;
    const lds_u16 *psrc = (aborted && p > 1) ? ch.perm_alt : ch.perm;
    for (int j = lane; j < p; j += WAVE) {
      g_gamma[j] = gsrc[j];
      g_perm[j] = psrc[j];
    }
    if (aborted) pos = pos0;
  }
  if (beta_valid) {
    double *g_beta = P.beta + (size_t)chain * p;
    for (int j = lane; j < p; j += WAVE) g_beta[j] = 0.0;
    wave_sync();
    if (lane < kprev) g_beta[gprev] = beta_m;
  } else if (nflips > 0 && done > 0) {
    // coef().set_inc(g) zeroes the coefficients of excluded variables
    // (Models/Glm/GlmCoefs.cpp:89-94) even when the beta draw is suppressed
    const lds_u8 *gsrc = aborted ? ch.gam0 : ch.gam;
    double *g_beta = P.beta + (size_t)chain * p;
    for (int j = lane; j < p; j += WAVE)
      if (!gsrc[j]) g_beta[j] = 0.0;
  }
  if (lane == 0) {
    P.sigsq[chain] = sigsq;
    P.rng_pos[chain] = pos;
    P.failures[chain] = failures;
    P.status[chain] = status;
    P.todo[chain] = nsweeps - done + owed_after;
    if (P.ran) P.ran[chain] = done;
    const int tag = KCAP | (cur << 8);
    P.table_tag[chain] = (table_valid && !aborted && status == CHAIN_OK) ? tag : 0;
    P.model_tag[chain] = (!aborted && status == CHAIN_OK) ? tag : 0;
    if (P.trace_idx) P.trace_idx[chain] = trace_at + done;
    if (P.maxk) atomicMax(P.maxk, kmax);
    double *a = P.acc + (size_t)chain * ACC_COUNT;
    a[ACC_SWEEPS] += done;
    a[ACC_SIGSQ] += ctl[CT_ACC + ACC_SIGSQ];
    a[ACC_SIGSQ2] += ctl[CT_ACC + ACC_SIGSQ2];
    a[ACC_K] += ctl[CT_ACC + ACC_K];
    a[ACC_ACCEPTS] += ctl[CT_ACC + ACC_ACCEPTS];
    a[ACC_PROPOSALS] += ctl[CT_ACC + ACC_PROPOSALS];
#ifndef BA_STAMPS
    a[ACC_SLOT_HITS] += ctl[CT_ACC + ACC_SLOT_HITS];
#endif
    a[ACC_MIN_MARGIN] = fmin(a[ACC_MIN_MARGIN], ctl[CT_ACC + ACC_MIN_MARGIN]);
#if defined(BA_STAMPS) && defined(BA_STAMPS4)
    // (phases are wave 1's)
#elif defined(BA_STAMPS) && (defined(BA_STAMPS2) || defined(BA_STAMPS3))
    SUBSTAMP(sx, 7);
    for (int i = 0; i < 8; ++i) a[ACC_PHASE0 + i] += sx.ph[i];
#elif defined(BA_STAMPS)
    for (int i = 0; i < 8; ++i) { a[ACC_PHASE0 + i] += st_ph[i]; a[ACC_SLOT_HITS] += st_ph[i]; }
#endif
  }
}

// log_model_prob of arbitrary inclusion vectors: one wavefront per vector.
// (BregVsSampler::log_model_prob, BregVsSampler.cpp:216-239)
__global__ __launch_bounds__(64) void ssvs_logp_kernel(SsvsParams P,
                                                       const uint8_t *gammas,
                                                       int ngamma, double *out,
                                                       int *status_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int which = blockIdx.x, lane = threadIdx.x, p = P.p;
  if (which >= ngamma) return;
  const SsvsLds lay = ssvs_lds_layout(p, P.kcap);
  Chain ch;
  ch.lane = lane;
  ch.p = p;
  bind_lds(ch, smem, lay);
  ch.xty = P.xty;
  ch.DF = P.nobs[0] + P.prior_df;
  ch.ss0q = P.prior_ss + P.yty[0];
  ch.mode = 0;
  ch.sv = ch.sa = ch.sx = 1.0;
  const uint8_t *gg = gammas + (size_t)which * p;
  int k = 0;
  for (int base = 0; base < p; base += WAVE) {
    const int j = base + lane;
    const int inc = (j < p) ? gg[j] : 0;
    if (j < p) ch.gam[j] = (uint8_t)inc;
    const unsigned long long mask = __ballot(inc != 0);
    const int slot = k + __popcll(mask & ((1ull << lane) - 1ull));
    if (inc && slot < P.kcap) ch.g[slot] = (uint16_t)j;
    k += __popcll(mask);
  }
  if (k > P.kcap) {
    if (lane == 0) { out[which] = __builtin_nan(""); status_out[which] = CHAIN_MODEL_TOO_LARGE; }
    return;
  }
  ch.k = k;
  __syncthreads();
  Model M;
  StampCtx sx;
  sx.last = 0;
  refactor<false>(P, ch, M, sx);
  if (lane == 0) {
    out[which] = M.logp;
    status_out[which] = M.bad;
  }
}

// Reduce the per-chain summaries over chains into one block of
// (3p + SUMMARY_SCALARS) doubles:
// [inclusion counts | beta sums | beta sums of squares | scalars].
// One workgroup per output column, chains strided over its threads, fixed
// tree: bitwise reproducible.
__global__ __launch_bounds__(256) void ssvs_reduce_summaries_kernel(SsvsParams P,
                                                                    double *out) {
  __shared__ double s0[256], s1[256], s2[256];
  const int p = P.p, j = blockIdx.x, tid = threadIdx.x;
  double a = 0, b = 0, c = 0;
  const bool is_min = (j >= p) && (j - p == ACC_MIN_MARGIN);
#ifdef BA_STAMPS
  const bool is_max = (j >= p) && (j - p == ACC_SLOT_HITS);  // diagnostic builds: the slot carries the slowest chain's cycles
#else
  const bool is_max = false;
#endif
  if (is_min) a = BA_INF;
  for (int chn = tid; chn < P.chains; chn += 256) {
    if (j < p) {
      const size_t o = (size_t)chn * p + j;
      a += (double)P.inc_count[o];
      b += P.beta_sum[o];
      c += P.beta_sumsq[o];
    } else {
      const double x = P.acc[(size_t)chn * ACC_COUNT + (j - p)];
      a = is_min ? fmin(a, x) : (is_max ? fmax(a, x) : a + x);
    }
  }
  s0[tid] = a; s1[tid] = b; s2[tid] = c;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (tid < w) {
      s0[tid] = is_min ? fmin(s0[tid], s0[tid + w]) : (is_max ? fmax(s0[tid], s0[tid + w]) : s0[tid] + s0[tid + w]);
      s1[tid] += s1[tid + w];
      s2[tid] += s2[tid + w];
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (j < p) {
      out[j] = s0[0];
      out[p + j] = s1[0];
      out[2 * p + j] = s2[0];
    } else {
      out[3 * p + (j - p)] = s0[0];
    }
  }
}

// ---- host-side launchers (kept in the kernels' translation unit) -----------
// (NB, W, WPE): model capacity 8 NB; W wavefronts per chain; WPE = waves per
// SIMD the register budget is sized for (W = 4 needs 4 resident waves per SIMD
// for 4 chains per CU, i.e. <= 128 VGPRs, which only the small capacities reach)
template <int NB, int W, int WPE>
static hipError_t launch_sweep_t(hipStream_t stream, const SsvsParams &P,
                                 int nsweeps) {
  const SsvsLds lay = ssvs_lds_layout(P.p, NB * 8);
  hipError_t e = hipFuncSetAttribute((const void *)ssvs_sweep_kernel<NB, W, WPE>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lay.total);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((ssvs_sweep_kernel<NB, W, WPE>), dim3(P.chain_count), dim3(WAVE * W),
                     lay.total, stream, P, nsweeps);
  return hipGetLastError();
}

hipError_t launch_ssvs_sweep(hipStream_t stream, const SsvsParams &P,
                             int nsweeps) {
  const int key = P.kcap * 10 + P.waves;
  switch (key) {
    case 161: return launch_sweep_t<2, 1, 1>(stream, P, nsweeps);
    case 321: return launch_sweep_t<4, 1, 1>(stream, P, nsweeps);
    case 481: return launch_sweep_t<6, 1, 1>(stream, P, nsweeps);
    case 641: return launch_sweep_t<8, 1, 1>(stream, P, nsweeps);
    case 162: return launch_sweep_t<2, 2, 2>(stream, P, nsweeps);
    case 322: return launch_sweep_t<4, 2, 2>(stream, P, nsweeps);
    case 482: return launch_sweep_t<6, 2, 2>(stream, P, nsweeps);
    case 642: return launch_sweep_t<8, 2, 2>(stream, P, nsweeps);
    case 164: return launch_sweep_t<2, 4, 4>(stream, P, nsweeps);
    case 324: return launch_sweep_t<4, 4, 4>(stream, P, nsweeps);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_ssvs_logp(hipStream_t stream, const SsvsParams &P,
                            const uint8_t *gammas, int ngamma, double *out,
                            int *status_out) {
  const SsvsLds lay = ssvs_lds_layout(P.p, P.kcap);
  hipError_t e = hipFuncSetAttribute((const void *)ssvs_logp_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lay.total);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ssvs_logp_kernel, dim3(ngamma), dim3(WAVE), lay.total,
                     stream, P, gammas, ngamma, out, status_out);
  return hipGetLastError();
}

// Device property the shuffle relies on (see parallel_shuffle): same-address
// LDS exchanges of one wavefront instruction are resolved in ascending lane
// order.  One wavefront tries 64 target patterns; *bad counts the lanes whose
// returned value is not that of the nearest lower lane with the same target.
__global__ __launch_bounds__(64) void lds_exchange_order_kernel(int *bad) {
  __shared__ uint32_t X[64];
  const int lane = threadIdx.x;
  int nbad = 0;
  for (int c = 0; c < 64; ++c) {
    X[lane] = 0xFFFFu;
    __syncthreads();
    const int key = (int)((((unsigned)lane * 2654435761u) >> 7) + (unsigned)c * 40503u) % (c + 1);
    const uint32_t old = __hip_atomic_exchange(&X[key], (uint32_t)(lane + 100), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WAVEFRONT);
    unsigned long long mask = ~0ull;
    for (int b = 0; b < 6; ++b) {
      const bool bit = (key >> b) & 1;
      const unsigned long long bal = __ballot(bit);
      mask &= bit ? bal : ~bal;
    }
    const unsigned long long lower = mask & ((1ull << lane) - 1ull);
    const uint32_t want = lower ? (uint32_t)(63 - __clzll((long long)lower) + 100) : 0xFFFFu;
    if (old != want) ++nbad;
    __syncthreads();
  }
  if (nbad) atomicAdd(bad, nbad);
}

hipError_t launch_lds_exchange_order(hipStream_t stream, int *bad_device) {
  hipLaunchKernelGGL(lds_exchange_order_kernel, dim3(1), dim3(64), 0, stream, bad_device);
  return hipGetLastError();
}

hipError_t launch_ssvs_reduce_summaries(hipStream_t stream, const SsvsParams &P,
                                        double *out) {
  hipLaunchKernelGGL(ssvs_reduce_summaries_kernel, dim3(P.p + SUMMARY_SCALARS),
                     dim3(256), 0, stream, P, out);
  return hipGetLastError();
}

}  // namespace boom_amd
