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
// Proposals for a variable with a non-zero prior mean b_j change r for the
// whole model; they take the (rare) exact path: refactor the candidate model.
#include <hip/hip_runtime.h>

#include "device_rng.h"
#include "ssvs_params.h"

namespace boom_amd {

namespace {

constexpr int WAVE = 64;
#define BA_INF (__builtin_inf())

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, WAVE);
  return x;
}
__device__ __forceinline__ double wave_min(double x) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x = fmin(x, __shfl_xor(x, off, WAVE));
  return x;
}
__device__ __forceinline__ double bcast(double x, int src) {
  return __shfl(x, src, WAVE);
}
__device__ __forceinline__ int tri(int m, int n) { return (m * (m + 1)) / 2 + n; }

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
  const SsvsParams *P;
  int lane, p, k;
  // LDS
  double *Lv, *La, *rdv, *rda, *w, *bg, *buf;
  uint16_t *g, *perm, *oth;
  uint8_t *gam;
  // this chain's sufficient statistics
  const double *xty;
  double DF;    // n + prior_df
  double ss0q;  // prior_ss + yty
};

// In-place Cholesky of a packed lower triangle, lane i owns row i (k <= 64).
// Left-looking by column: the subtraction order for every entry is that of
// Eigen's unblocked LLT (Eigen/src/Cholesky/LLT.h:313-335) which the
// reference uses (LinAlg/Cholesky.cpp:33-58).  Returns false at the first
// non-positive pivot.  logdet = 2 * sum log L_jj.
__device__ bool chol_packed(const Chain &ch, double *Lp, double *rd,
                            double *logdet) {
  const int k = ch.k, i = ch.lane;
  double ld = 0.0;
  bool ok = true;
  for (int j = 0; j < k; ++j) {
    const bool mine = (i >= j) && (i < k);
    double s = mine ? Lp[tri(i, j)] : 0.0;
    if (mine) {
      const double *ri = Lp + tri(i, 0);
      const double *rj = Lp + tri(j, 0);
      for (int n = 0; n < j; ++n) s -= ri[n] * rj[n];
    }
    const double d = bcast(s, j);
    if (!(d > 0.0)) {
      ok = false;
      break;
    }
    const double sd = sqrt(d);
    ld += log(sd);
    if (i == j) {
      Lp[tri(j, j)] = sd;
      rd[j] = 1.0 / sd;
    } else if (mine) {
      Lp[tri(i, j)] = s / sd;
    }
    __syncthreads();
  }
  *logdet = 2.0 * ld;
  return ok;
}

// Rebuild everything about the current model gamma (sorted index list g in
// LDS) from scratch: BregVsSampler::set_reg_post_params + log_model_prob.
__device__ void refactor(Chain &ch, Model &M) {
  const SsvsParams &P = *ch.P;
  const int lane = ch.lane, p = ch.p, k = ch.k;
  M.bad = 0;
  M.pd = true;
  // VariableSelectionPrior::logp (VariableSelectionPrior.cpp:271-285)
  double part = 0.0;
  for (int j = lane; j < p; j += WAVE) part += ch.gam[j] ? P.l1[j] : P.l0[j];
  double lp = wave_sum(part);
  if (P.max_model_size >= 0 && k > P.max_model_size) lp = -BA_INF;
  if (!(lp > -BA_INF)) lp = -BA_INF;  // also catches NaN from inf - inf
  M.lp = lp;
  M.ldv = M.lda = M.Q = M.c = 0.0;
  M.SS = ch.ss0q;
  if (k == 0) {
    M.logp = lp - (0.5 * ch.DF - 1.0) * log(ch.ss0q);
    return;
  }
  if (lp == -BA_INF) {
    M.logp = -BA_INF;
    M.pd = false;
    return;
  }
  __syncthreads();
  // gather V_g, A_g (lower triangles) and b_g
  for (int m = 0; m < k; ++m) {
    const size_t row = (size_t)ch.g[m] * p;
    if (lane <= m) {
      const int gn = ch.g[lane];
      ch.Lv[tri(m, lane)] = P.V[row + gn];
      ch.La[tri(m, lane)] = P.A[row + gn];
    }
  }
  const int gm = (lane < k) ? ch.g[lane] : 0;
  if (lane < k) ch.bg[lane] = P.b[gm];
  __syncthreads();
  // r = A_g b_g + xty_g ; c = b_g' A_g b_g
  double r = 0.0, ab = 0.0;
  if (lane < k) {
    for (int n = 0; n < k; ++n) {
      const double bn = ch.bg[n];
      if (bn != 0.0) ab += P.A[(size_t)gm * p + ch.g[n]] * bn;
    }
    r = ab + ch.xty[gm];
  }
  M.c = wave_sum(lane < k ? ch.bg[lane] * ab : 0.0);
  const bool oka = chol_packed(ch, ch.La, ch.rda, &M.lda);
  const bool okv = chol_packed(ch, ch.Lv, ch.rdv, &M.ldv);
  if (!okv) {
    M.pd = false;
    M.logp = -BA_INF;
    return;
  }
  // w = L_V^{-1} r, lane m ends up holding w_m
  double x = r;
  for (int j = 0; j < k; ++j) {
    const double wj = bcast(x, j) * ch.rdv[j];
    if (lane == j) x = wj;
    else if (lane > j && lane < k) x -= ch.Lv[tri(lane, j)] * wj;
  }
  if (lane < k) ch.w[lane] = x;
  M.Q = wave_sum(lane < k ? x * x : 0.0);
  M.SS = ch.ss0q + M.c - M.Q;
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
  __syncthreads();
}

// flip variable j in the LDS copy of gamma and in the sorted list g
__device__ void apply_flip(Chain &ch, int j) {
  const int lane = ch.lane, k = ch.k;
  const int gm = (lane < k) ? ch.g[lane] : 0x7fffffff;
  const int below = __popcll(__ballot(lane < k && gm < j));
  const bool add = !ch.gam[j];
  __syncthreads();
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
  __syncthreads();
}

// per-lane forward substitution L x = rhs with the rhs (and result) in this
// lane's column of buf; returns |x|^2 and x.w
__device__ __forceinline__ void lane_solve(const Chain &ch, const double *Lp,
                                           const double *rd, double *nrm,
                                           double *dotw) {
  const int k = ch.k;
  double *col = ch.buf + ch.lane;
  double n2 = 0.0, dw = 0.0;
  for (int m = 0; m < k; ++m) {
    double acc = col[m * WAVE];
    const double *row = Lp + tri(m, 0);
    for (int n = 0; n < m; ++n) acc -= row[n] * col[n * WAVE];
    const double x = acc * rd[m];
    col[m * WAVE] = x;
    n2 += x * x;
    dw += x * ch.w[m];
  }
  *nrm = n2;
  *dotw = dw;
}

struct Proposal {
  double logp;   // log_model_prob of the flipped model (-inf: impossible)
  bool slow;     // needs the exact path (non-zero prior mean on j)
  bool bad_ss;   // SS' < 0: the reference would throw here
};

// Evaluate this lane's proposal "flip j" against the current model.
__device__ Proposal eval_proposal(Chain &ch, const Model &M, int j, bool valid) {
  const SsvsParams &P = *ch.P;
  const int p = ch.p, k = ch.k, lane = ch.lane;
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
    out.logp = lpn - (0.5 * ch.DF - 1.0) * log(ch.ss0q);
  }
  // ---- V part: rhs = V[g, j] (add) or e_i (drop)
  double ab = 0.0;     // A[j, g] . b_g   (adds)
  double ajj = 0.0, vjj = 0.0;
  {
    double *col = ch.buf + lane;
    for (int m = 0; m < k; ++m) {
      const int gm = ch.g[m];
      double v = 0.0;
      if (fast) v = add ? P.V[(size_t)gm * p + j] : (gm == j ? 1.0 : 0.0);
      col[m * WAVE] = v;
    }
  }
  double nv, dv;
  lane_solve(ch, ch.Lv, ch.rdv, &nv, &dv);
  // ---- A part
  {
    double *col = ch.buf + lane;
    for (int m = 0; m < k; ++m) {
      const int gm = ch.g[m];
      double a = 0.0;
      if (fast) {
        if (add) {
          a = P.A[(size_t)gm * p + j];
          ab += a * ch.bg[m];
        } else {
          a = (gm == j ? 1.0 : 0.0);
        }
      }
      col[m * WAVE] = a;
    }
  }
  double na, da_unused;
  lane_solve(ch, ch.La, ch.rda, &na, &da_unused);
  if (fast) {
    double ldv, lda, Q;
    bool ok = true;
    if (add) {
      vjj = P.V[(size_t)j * p + j];
      ajj = P.A[(size_t)j * p + j];
      const double d2 = vjj - nv;
      const double da2 = ajj - na;
      if (!(d2 > 0.0) || !(da2 > 0.0)) ok = false;
      const double rj = ch.xty[j] + ab;  // b_j == 0 on this path
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
    if (ok) {
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

// BregVsSampler::attempt_swap (BregVsSampler.cpp:277-310) with
// CorrelationMap::propose_swap / proposal_weight (CorrelationMap.cpp:61-115).
// Wave-uniform control flow; every lane walks the same CSR lists.
__device__ void attempt_swap(Chain &ch, Model &M, SeqRng &rng, int *status) {
  const SsvsParams &P = *ch.P;
  if (P.cm_start == nullptr) return;
  const int k = ch.k, p = ch.p;
  if (k == 0 || k == p) return;
  // Selector::random_included_position, LinAlg/Selector.cpp:297-304
  const int pos = d_random_int(rng, 0, k - 1);
  const int index = ch.g[pos];
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
  const double original_logp = M.logp;
  const Model saved = M;
  apply_flip(ch, index);
  apply_flip(ch, candidate);
  Model Mn;
  refactor(ch, Mn);
  if (Mn.bad) { *status = Mn.bad; return; }
  // reverse weight = proposal_weight(included', candidate, index)
  double rev;
  {
    const int l2 = P.cm_start[candidate], h2 = P.cm_start[candidate + 1];
    double ans = -BA_INF, tot = 0.0;
    for (int i = l2; i < h2; ++i) {
      if (!ch.gam[P.cm_idx[i]]) {
        if (P.cm_idx[i] == index) ans = P.cm_cor[i];
        tot += P.cm_cor[i];
      }
    }
    rev = (tot == 0.0) ? 0.0 : ans / tot;
  }
  const double log_num = Mn.logp - log(forward_w);
  const double log_den = original_logp - log(rev);
  const double logu = log(d_runif(rng, 0.0, 1.0));
  if (logu < log_num - log_den) {
    M = Mn;
  } else {
    apply_flip(ch, candidate);
    apply_flip(ch, index);
    refactor(ch, M);
    (void)saved;
  }
}

}  // namespace

// ============================================================================
// grid = chains, block = 64
__global__ __launch_bounds__(64) void ssvs_sweep_kernel(SsvsParams P,
                                                        int nsweeps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int chain = blockIdx.x;
  const int lane = threadIdx.x;
  const int p = P.p;
  if (chain >= P.chains) return;
  if (P.status[chain] != CHAIN_OK) return;

  const SsvsLds lay = ssvs_lds_layout(p, P.kcap);
  Chain ch;
  ch.P = &P;
  ch.lane = lane;
  ch.p = p;
  ch.Lv = (double *)(smem + lay.Lv);
  ch.La = (double *)(smem + lay.La);
  ch.rdv = (double *)(smem + lay.rdv);
  ch.rda = (double *)(smem + lay.rda);
  ch.w = (double *)(smem + lay.w);
  ch.bg = (double *)(smem + lay.bg);
  ch.buf = (double *)(smem + lay.buf);
  ch.g = (uint16_t *)(smem + lay.g);
  ch.perm = (uint16_t *)(smem + lay.perm);
  ch.oth = (uint16_t *)(smem + lay.oth);
  ch.gam = (uint8_t *)(smem + lay.gam);
  ch.xty = P.xty + (size_t)chain * P.xty_stride;
  const double yty = P.yty[(size_t)chain * P.suf_stride];
  const double nobs = P.nobs[(size_t)chain * P.suf_stride];
  ch.DF = nobs + P.prior_df;
  ch.ss0q = P.prior_ss + yty;

  // ---- load chain state
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
    }
    const unsigned long long mask = __ballot(inc != 0);
    const int slot = k + __popcll(mask & ((1ull << lane) - 1ull));
    if (inc && slot < P.kcap) ch.g[slot] = (uint16_t)j;
    k += __popcll(mask);
  }
  if (k > P.kcap) {
    if (lane == 0) P.status[chain] = CHAIN_MODEL_TOO_LARGE;
    return;
  }
  ch.k = k;
  __syncthreads();

  PhiloxKey key{P.seed_lo, P.seed_hi,
                (uint32_t)(P.chain_offset + chain), P.stream};
  uint64_t pos = P.rng_pos[chain];
  int failures = P.failures[chain];
  double sigsq = P.sigsq[chain];
  double beta_m = 0.0;  // lane m: coefficient of g[m] after the last draw
  bool beta_valid = false;

  double acc_sig = 0, acc_sig2 = 0, acc_k = 0, acc_acc = 0, acc_prop = 0;
  double min_margin = BA_INF;
  int done = 0;

  const int nflips = P.max_flips;  // already min(max_nflips_, p)

  for (int sweep = 0; sweep < nsweeps && status == CHAIN_OK; ++sweep) {
    Model M;
    if (nflips > 0) {
      // ---- shuffle(indx): cpputil/shuffle.hpp:36-46, in place on the
      // persistent permutation.  Uniform t (t = 0..p-2) belongs to i = p-1-t.
      for (int t = lane; t < p - 1; t += WAVE) {
        const int i = p - 1 - t;
        const double u = philox_uniform(key, pos + (uint64_t)t);
        ch.oth[i] = (uint16_t)(int)floor(0.0 + ((double)(i + 1) - 0.0) * u);
      }
      __syncthreads();
      if (lane == 0) {
        for (int i = p - 1; i > 0; --i) {
          const int o = ch.oth[i];
          const uint16_t a = ch.perm[i];
          ch.perm[i] = ch.perm[o];
          ch.perm[o] = a;
        }
      }
      __syncthreads();
      const uint64_t flip_pos = pos + (uint64_t)(p > 0 ? p - 1 : 0);
      pos = flip_pos + (uint64_t)nflips;

      refactor(ch, M);
      if (!M.bad && !(M.logp > -BA_INF && M.logp < BA_INF)) {
        // VariableSelectionPrior::make_valid, VariableSelectionPrior.cpp:287-300
        for (int j = 0; j < p; ++j) {
          const double pj = P.pi[j];
          const bool inc = ch.gam[j];
          if ((pj <= 0.0 && inc) || (pj >= 1.0 && !inc)) {
            if (!inc && ch.k >= P.kcap) { status = CHAIN_MODEL_TOO_LARGE; break; }
            apply_flip(ch, j);
          }
        }
        if (status == CHAIN_OK) refactor(ch, M);
        if (status == CHAIN_OK && !M.bad &&
            !(M.logp > -BA_INF && M.logp < BA_INF))
          status = CHAIN_ILLEGAL_START;
      }
      if (M.bad) status = M.bad;

      // ---- p Metropolised flips, 64 proposals at a time
      int i0 = 0;
      while (i0 < nflips && status == CHAIN_OK) {
        const int idx = i0 + lane;
        const bool valid = idx < nflips;
        const int j = valid ? ch.perm[idx] : 0;
        const double u = philox_uniform(key, flip_pos + (uint64_t)idx);
        const double logu = log(u);
        Proposal pr = eval_proposal(ch, M, j, valid);
        const double delta = pr.logp - M.logp;
        const bool accept = valid && !pr.slow && !pr.bad_ss && !(logu > delta);
        const unsigned long long m_acc = __ballot(accept);
        const unsigned long long m_slow = __ballot(valid && pr.slow);
        const unsigned long long m_bad = __ballot(valid && pr.bad_ss);
        const unsigned long long m_stop = m_acc | m_slow | m_bad;
        const int f = m_stop ? (__ffsll((long long)m_stop) - 1) : WAVE;
        // lanes before f are settled rejections
        {
          const bool counted = valid && lane < f;
          const double mg = counted && (pr.logp > -BA_INF) ? fabs(logu - delta) : BA_INF;
          min_margin = fmin(min_margin, wave_min(mg));
        }
        if (f == WAVE) {
          const int n = (nflips - i0 < WAVE) ? (nflips - i0) : WAVE;
          acc_prop += n;
          i0 += WAVE;
          continue;
        }
        acc_prop += f + 1;
        const int jf = __shfl(j, f, WAVE);
        if ((m_bad >> f) & 1ull) {
          status = CHAIN_NEGATIVE_SS;
          break;
        }
        const bool adding = !ch.gam[jf];
        if (adding && ch.k >= P.kcap) {
          // the candidate cannot be held in LDS; if it is a sure rejection
          // that is fine, but we cannot tell without evaluating it
          status = CHAIN_MODEL_TOO_LARGE;
          break;
        }
        if ((m_acc >> f) & 1ull) {
          // accepted on the fast path: move to the new model
          const double mg = fabs(bcast(logu, f) - bcast(delta, f));
          min_margin = fmin(min_margin, mg);
          apply_flip(ch, jf);
          refactor(ch, M);
          if (M.bad) { status = M.bad; break; }
          if (!M.pd) { status = CHAIN_NOT_PD; break; }
          acc_acc += 1;
        } else {
          // exact path: evaluate the flipped model from scratch
          const double lu = bcast(logu, f);
          const Model keep = M;
          apply_flip(ch, jf);
          Model Mn;
          refactor(ch, Mn);
          if (Mn.bad) { status = Mn.bad; break; }
          const double dl = Mn.logp - keep.logp;
          if (Mn.logp > -BA_INF) min_margin = fmin(min_margin, fabs(lu - dl));
          if (lu > dl) {
            apply_flip(ch, jf);
            refactor(ch, M);
          } else {
            M = Mn;
            acc_acc += 1;
          }
        }
        i0 += f + 1;
      }
      if (status != CHAIN_OK) break;
    }

    SeqRng rng{key, pos};
    if (nflips > 0) attempt_swap(ch, M, rng, &status);
    if (status != CHAIN_OK) break;
    if (nflips == 0) {
      refactor(ch, M);  // set_reg_post_params(inc, false)
      if (M.bad) { status = M.bad; break; }
    }
    k = ch.k;

    // ---- draw_sigma (BregVsSampler.cpp:313-324)
    if (P.draw_sigma) {
      int bad = 0;
      const double DF = (k == 0) ? ch.DF : ((ch.DF - P.prior_df) + P.prior_df);
      const double SS = (k == 0) ? ch.ss0q : ((M.SS - P.prior_ss) + P.prior_ss);
      sigsq = d_draw_variance(rng, DF, SS, P.sigma_max, &bad);
      if (bad) { status = CHAIN_RNG_BRANCH; break; }
    }
    // ---- draw_beta (BregVsSampler.cpp:326-351)
    if (P.draw_beta && k > 0) {
      if (!M.pd) {
        ++failures;
        status = CHAIN_NOT_PD;
        break;
      }
      failures = 0;
      // z_i ~ N(0,1) in coefficient order (distributions/mvn.cpp:114-122)
      double z = 0.0;
      for (int m = 0; m < k; ++m) {
        const double zm = d_norm_rand(rng);
        if (lane == m) z = zm;
      }
      // beta = L^{-T}(w + sigma z): chol(V / sigma^2) = L / sigma
      const double sigma = sqrt(sigsq);
      double y = (lane < k) ? ch.w[lane] + sigma * z : 0.0;
      for (int i = k - 1; i >= 0; --i) {
        const double xi = bcast(y, i) * ch.rdv[i];
        if (lane == i) y = xi;
        else if (lane < i) y -= ch.Lv[tri(i, lane)] * xi;
      }
      beta_m = y;
      beta_valid = true;
    } else if (P.draw_beta) {
      beta_valid = true;  // empty model: all coefficients zero
    }
    pos = rng.pos;

    // ---- summaries
    if (lane < k) {
      const size_t o = (size_t)chain * p + ch.g[lane];
      P.inc_count[o] += 1u;
      if (beta_valid) {
        P.beta_sum[o] += beta_m;
        P.beta_sumsq[o] += beta_m * beta_m;
      }
    }
    acc_sig += sigsq;
    acc_sig2 += sigsq * sigsq;
    acc_k += k;
    if (P.trace_sigsq && sweep < P.trace_stride && lane == 0) {
      const size_t o = (size_t)chain * P.trace_stride + sweep;
      P.trace_sigsq[o] = sigsq;
      P.trace_logp[o] = M.logp;
      P.trace_k[o] = (double)k;
    }
    ++done;
  }

  // ---- write the chain back
  k = ch.k;
  __syncthreads();
  for (int j = lane; j < p; j += WAVE) {
    g_gamma[j] = ch.gam[j];
    g_perm[j] = ch.perm[j];
  }
  if (beta_valid) {
    double *g_beta = P.beta + (size_t)chain * p;
    for (int j = lane; j < p; j += WAVE) g_beta[j] = 0.0;
    __syncthreads();
    if (lane < k) g_beta[ch.g[lane]] = beta_m;
  } else if (nflips > 0) {
    // coef().set_inc(g) zeroes the coefficients of excluded variables
    // (Models/Glm/GlmCoefs.cpp:89-94) even when the beta draw is suppressed
    double *g_beta = P.beta + (size_t)chain * p;
    for (int j = lane; j < p; j += WAVE)
      if (!ch.gam[j]) g_beta[j] = 0.0;
  }
  if (lane == 0) {
    P.sigsq[chain] = sigsq;
    P.rng_pos[chain] = pos;
    P.failures[chain] = failures;
    P.status[chain] = status;
    double *a = P.acc + (size_t)chain * ACC_COUNT;
    a[ACC_SWEEPS] += done;
    a[ACC_SIGSQ] += acc_sig;
    a[ACC_SIGSQ2] += acc_sig2;
    a[ACC_K] += acc_k;
    a[ACC_ACCEPTS] += acc_acc;
    a[ACC_PROPOSALS] += acc_prop;
    a[ACC_MIN_MARGIN] = fmin(a[ACC_MIN_MARGIN], min_margin);
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
  ch.P = &P;
  ch.lane = lane;
  ch.p = p;
  ch.Lv = (double *)(smem + lay.Lv);
  ch.La = (double *)(smem + lay.La);
  ch.rdv = (double *)(smem + lay.rdv);
  ch.rda = (double *)(smem + lay.rda);
  ch.w = (double *)(smem + lay.w);
  ch.bg = (double *)(smem + lay.bg);
  ch.buf = (double *)(smem + lay.buf);
  ch.g = (uint16_t *)(smem + lay.g);
  ch.perm = (uint16_t *)(smem + lay.perm);
  ch.oth = (uint16_t *)(smem + lay.oth);
  ch.gam = (uint8_t *)(smem + lay.gam);
  ch.xty = P.xty;
  ch.DF = P.nobs[0] + P.prior_df;
  ch.ss0q = P.prior_ss + P.yty[0];
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
  refactor(ch, M);
  if (lane == 0) {
    out[which] = M.logp;
    status_out[which] = M.bad;
  }
}

// Reduce the per-chain summaries over chains into one block of (3p + 8)
// doubles: [inclusion counts | beta sums | beta sums of squares | scalars].
__global__ void ssvs_reduce_summaries_kernel(SsvsParams P, double *out) {
  const int p = P.p;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < p) {
    double c = 0, s = 0, s2 = 0;
    for (int ch = 0; ch < P.chains; ++ch) {
      const size_t o = (size_t)ch * p + j;
      c += (double)P.inc_count[o];
      s += P.beta_sum[o];
      s2 += P.beta_sumsq[o];
    }
    out[j] = c;
    out[p + j] = s;
    out[2 * p + j] = s2;
  } else if (j < p + SUMMARY_SCALARS) {
    const int a = j - p;
    double v = (a == ACC_MIN_MARGIN) ? BA_INF : 0.0;
    for (int ch = 0; ch < P.chains; ++ch) {
      const double x = P.acc[(size_t)ch * ACC_COUNT + a];
      v = (a == ACC_MIN_MARGIN) ? fmin(v, x) : v + x;
    }
    out[3 * p + a] = v;
  }
}

// ---- host-side launchers (kept in the kernels' translation unit) -----------
hipError_t launch_ssvs_sweep(hipStream_t stream, const SsvsParams &P,
                             int nsweeps) {
  const SsvsLds lay = ssvs_lds_layout(P.p, P.kcap);
  hipError_t e = hipFuncSetAttribute((const void *)ssvs_sweep_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lay.total);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ssvs_sweep_kernel, dim3(P.chains), dim3(WAVE), lay.total,
                     stream, P, nsweeps);
  return hipGetLastError();
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

hipError_t launch_ssvs_reduce_summaries(hipStream_t stream, const SsvsParams &P,
                                        double *out) {
  const int total = P.p + SUMMARY_SCALARS;
  hipLaunchKernelGGL(ssvs_reduce_summaries_kernel, dim3((total + 255) / 256),
                     dim3(256), 0, stream, P, out);
  return hipGetLastError();
}

}  // namespace boom_amd
