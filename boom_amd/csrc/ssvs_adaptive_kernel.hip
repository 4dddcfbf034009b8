// AdaptiveSpikeSlabRegressionSampler::draw() for many chains, one chain per
// wavefront (Models/Glm/PosteriorSamplers/AdaptiveSpikeSlabRegressionSampler.cpp:
// 62-225): what lm.spike runs for more than 100 predictors.
//
//   flips = min(max_flips_, p) Metropolis-Hastings moves, each a birth (u < .5:
//   an excluded variable drawn with probability proportional to its birth rate)
//   or a death (an included one by its death rate), accepted when
//     log u' < [logp(cand) - log(w_j / sum w)] - [logp(cur) - log(back_j / sum back)];
//   an accepted move nudges the variable's rate (adjust_birth_rate /
//   adjust_death_rate); then set_posterior_moments, sigma^2, beta as in
//   BregVsSampler.
//
// The moves of a sweep use three stream numbers each at fixed positions as long
// as no move is accepted, and what a move proposes depends on the model and the
// rates only -- so, like the batch mode of the BregVsSampler kernel, the next 64
// moves are evaluated speculatively, one per lane (weighted draw by binary
// search in the cumulative rates, then the O(k^2) evaluation of the flipped
// model against the current factors through the scalar cache); the first lane
// that accepts wins, the lanes before it were correct rejections, and the model
// is rebuilt once per ACCEPTED move instead of once per proposal.
#include "ktimer.h"
#include "ssvs_device.h"

namespace boom_amd {

namespace {

// inclusive prefix sum over the wave, lane order
__device__ __forceinline__ double ada_prefix(double x) {
  x += dpp_f64<0x111, 0xf>(x, 0.0);
  x += dpp_f64<0x112, 0xf>(x, 0.0);
  x += dpp_f64<0x114, 0xf>(x, 0.0);
  x += dpp_f64<0x118, 0xf>(x, 0.0);
  x += dpp_f64<0x142, 0xa>(x, 0.0);
  x += dpp_f64<0x143, 0xc>(x, 0.0);
  return x;
}

}  // namespace

// grid = chains, block = 64; NB = kcap / 8
template <int NB>
__global__ __launch_bounds__(64, 1) void ssvs_adaptive_kernel(SsvsParams P, int nsweeps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KCAP = NB * 8;
  const int chain = (int)blockIdx.x + P.chain_first;
  const int lane = threadIdx.x;
  const int p = P.p;
  if ((int)blockIdx.x >= P.chain_count) return;
  if (P.status[chain] != CHAIN_OK) {
    if (lane == 0) P.todo[chain] += nsweeps;
    return;
  }
  nsweeps += P.todo[chain];
  if (nsweeps == 0) return;

  const SsvsAdaLds lay = ssvs_ada_lds_layout(p, KCAP);
  Chain ch;
  ch.lane = lane;
  ch.p = p;
  ch.k = 0;
  ch.Lv = to_lds<double>(smem + lay.Lv);
  ch.La = to_lds<double>(smem + lay.La);
  ch.rdv = to_lds<double>(smem + lay.rdv);
  ch.rda = to_lds<double>(smem + lay.rda);
  ch.w = to_lds<double>(smem + lay.w);
  ch.bg = to_lds<double>(smem + lay.bg);
  ch.g = to_lds<uint16_t>(smem + lay.g);
  ch.gam = to_lds<uint8_t>(smem + lay.gam);
  ch.gam0 = to_lds<uint8_t>(smem + lay.gam0);
  ch.perm = ch.perm_alt = ch.oth = ch.pred = nullptr;
  ch.last = nullptr;
  ch.nbr = nullptr;
  lds_f64 *ctl = to_lds<double>(smem + lay.ctrl);
  lds_f64 *birth = to_lds<double>(smem + lay.birth);
  lds_f64 *death = to_lds<double>(smem + lay.death);
  lds_f64 *cumb = to_lds<double>(smem + lay.cumb);
  lds_f64 *cumd = to_lds<double>(smem + lay.cumd);
  lds_f64 *undo_v = to_lds<double>(smem + lay.undo_v);
  AS_LDS int *undo_j = to_lds<int>(smem + lay.undo_j);
  ch.xty = P.xty + (size_t)chain * P.xty_stride;
  const double yty = P.yty[(size_t)chain * P.suf_stride];
  const double nobs = P.nobs[(size_t)chain * P.suf_stride];
  ch.DF = nobs + P.prior_df;
  ch.ss0q = P.prior_ss + yty;
  ch.mode = 0;
  ch.sv = ch.sa = ch.sx = 1.0;
  ch.tab_lp = nullptr;
  ch.tab_kind = nullptr;
  ch.sc_store = P.model_scratch + (size_t)chain * P.model_scratch_stride;
  ch.sc = (c_f64 *)(unsigned long long)ch.sc_store;
  const PhiloxKey key{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), P.stream};

  uint8_t *g_gamma = P.gamma + (size_t)chain * p;
  double *g_birth = P.ada_birth + (size_t)chain * p;
  double *g_death = P.ada_death + (size_t)chain * p;
  int k = 0;
  int status = CHAIN_OK;
  for (int base = 0; base < p; base += WAVE) {
    const int j = base + lane;
    const int inc = (j < p) ? g_gamma[j] : 0;
    if (j < p) {
      ch.gam[j] = (uint8_t)inc;
      birth[j] = g_birth[j];
      death[j] = g_death[j];
    }
    const unsigned long long mask = __ballot(inc != 0);
    const int slot = k + __popcll(mask & ((1ull << lane) - 1ull));
    if (inc && slot < KCAP) ch.g[slot] = (uint16_t)j;
    k += __popcll(mask);
  }
  if (k > KCAP) status = CHAIN_MODEL_TOO_LARGE;
  ch.k = k;
  wave_sync();
  int kmax = k;
  int trace_at = P.trace_idx ? P.trace_idx[chain] : 0;
  uint64_t pos = uni((uint64_t)P.rng_pos[chain]);
  uint64_t iteration = uni((uint64_t)P.ada_iter[chain]);
  int failures = uni((int)P.failures[chain]);
  double sigsq = uni((double)P.sigsq[chain]);
  double beta_m = 0.0;
  int gprev = 0, kprev = 0;
  bool beta_valid = false, aborted = false;
  if (lane < 16) ctl[CT_ACC + lane] = (lane == ACC_MIN_MARGIN || lane == ACC_PHASE0) ? BA_INF : 0.0;
  wave_sync();
#define AACC_ADD(slot, x) do { if (lane == 0) ctl[CT_ACC + (slot)] += (double)(x); } while (0)
#define AACC_MIN(slot, x) do { if (lane == 0) ctl[CT_ACC + (slot)] = fmin(ctl[CT_ACC + (slot)], (x)); } while (0)
  int done = 0;
  StampCtx sx;
  sx.last = 0;
  Model M;
  M.bad = 0; M.pd = true; M.logp = 0; M.lp = 0; M.ldv = 0; M.lda = 0; M.Q = 0; M.c = 0; M.SS = 0;
  WinRng rng;
  rng.init(key, lane, pos);
  const int flips_per_sweep = (P.ada_max_flips < p) ? P.ada_max_flips : p;

  // (re)build the model of the current gamma and publish it for the evaluations
  auto rebuild = [&]() {
    refactor<false>(P, ch, M, sx);
    if (M.bad) { status = M.bad; return; }
    M.logp = uni(M.logp);
    M.SS = uni(M.SS);
    publish_model<NB>(ch, M);
  };
  if (status == CHAIN_OK) rebuild();

  for (int sweep = 0; sweep < nsweeps && status == CHAIN_OK; ++sweep) {
    const uint64_t pos0 = pos;
    int nundo = 0;
    if (flips_per_sweep > 0) {
      for (int j = lane; j < p; j += WAVE) ch.gam0[j] = ch.gam[j];
      wave_sync();
      bool cum_dirty = true;
      double Bsum = 0.0, Dsum = 0.0;
      int i = 0;
      while (i < flips_per_sweep && status == CHAIN_OK) {
        if (cum_dirty) {
          // cumulative rates of the candidates of either move, in variable order
          double cb = 0.0, cd = 0.0;
          for (int base = 0; base < p; base += WAVE) {
            const int j = base + lane;
            const bool in = j < p;
            const bool inc = in && ch.gam[j];
            const double wb = (in && !inc) ? birth[j] : 0.0;
            const double wd = inc ? death[j] : 0.0;
            const double pb = ada_prefix(wb), pd = ada_prefix(wd);
            if (in) { cumb[j] = cb + pb; cumd[j] = cd + pd; }
            cb += bcast_u(pb, 63);
            cd += bcast_u(pd, 63);
          }
          Bsum = cb;
          Dsum = cd;
          cum_dirty = false;
          wave_sync();
        }
        const int kk = ch.k;
        const bool edge = (kk == 0 || kk == p);   // one of the moves is impossible: one move at a time
        const int nb = edge ? 1 : ((flips_per_sweep - i < WAVE) ? flips_per_sweep - i : WAVE);
        const bool valid = lane < nb;
        const uint64_t mypos = pos + 3ull * (uint64_t)lane;
        const double u0 = philox_uniform(key, mypos);
        const bool isbirth = u0 < .5;
        const bool possible = isbirth ? (kk < p) : (kk > 0);
        const double u1 = philox_uniform(key, mypos + 1);
        const double u2 = philox_uniform(key, mypos + 2);
        const double tot = isbirth ? Bsum : Dsum;
        const lds_f64 *cum = isbirth ? cumb : cumd;
        // rmulti_mt: tmp = runif(0, sum); first candidate with tmp <= running sum
        const double tmp = 0.0 + (tot - 0.0) * u1;
        int lo = 0, hi = p - 1;
        if (valid && possible) {
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] >= tmp) hi = mid; else lo = mid + 1;
          }
        }
        int j = lo;
        // (tmp == 0 lands on leading non-candidates: move on to the first candidate)
        if (valid && possible) {
          while (j < p - 1 && ((ch.gam[j] != 0) == isbirth)) ++j;
        }
        const bool cand_ok = valid && possible && ((ch.gam[j] != 0) != isbirth);
        const bool broken_multi = valid && possible && !cand_ok;
        double mmargin = BA_INF;
        if (cand_ok) {
          const double below = (j > 0) ? cum[j - 1] : 0.0;
          mmargin = fmin(fabs(tmp - cum[j]), (tmp > below || j == 0) ? fabs(tmp - below) : BA_INF) / tot;
        }
        const Proposal pr = eval_proposal<NB, false>(P, ch, M, cand_ok ? j : 0, cand_ok, sx);
        const double wj = cand_ok ? (isbirth ? birth[j] : death[j]) : 1.0;
        const double bj = cand_ok ? (isbirth ? death[j] : birth[j]) : 1.0;
        const double fwd = log(wj / tot);
        const double rev = log(bj / ((isbirth ? Dsum : Bsum) + bj));
        const double logu = log(u2);
        const double ratio = (pr.logp - fwd) - (M.logp - rev);
        const bool slow = cand_ok && pr.slow;
        const bool bad = cand_ok && pr.bad_ss;
        const bool accept = cand_ok && !slow && !bad && (logu < ratio);
        const unsigned long long m_acc = __ballot(accept), m_slow = __ballot(slow),
                                 m_bad = __ballot(bad), m_brk = __ballot(broken_multi);
        const unsigned long long m_stop = m_acc | m_slow | m_bad | m_brk;
        const int f = m_stop ? (__ffsll((long long)m_stop) - 1) : WAVE;
        const bool counted = cand_ok && (lane < f || (lane == f && ((m_acc >> f) & 1ull)));
        const double mg = (counted && pr.logp > -BA_INF && M.logp > -BA_INF) ? fabs(logu - ratio) : BA_INF;
        {  // (reductions by the whole wave, then lane 0 files them)
          const double wm = wave_min(mg), wmm = wave_min(counted ? mmargin : BA_INF);
          AACC_MIN(ACC_MIN_MARGIN, wm);
          AACC_MIN(ACC_PHASE0, wmm);
        }
        if (f == WAVE) {       // nb settled rejections (or an impossible move)
          AACC_ADD(ACC_PROPOSALS, nb);
          i += nb;
          pos += edge ? (uni((int)possible) ? 3ull : 1ull) : 3ull * (uint64_t)nb;
          continue;
        }
        // lane f stops the batch: the moves before it are settled
        AACC_ADD(ACC_PROPOSALS, f + 1);
        i += f + 1;
        pos += 3ull * (uint64_t)(f + 1);
        if ((m_brk >> f) & 1ull) { status = CHAIN_RNG_BRANCH; break; }
        if ((m_bad >> f) & 1ull) { status = CHAIN_NEGATIVE_SS; break; }
        const int jf = bcast_u(j, f);
        const bool fbirth = bcast_u((int)isbirth, f) != 0;
        const double ffwd = bcast_u(fwd, f), frev = bcast_u(rev, f), flogu = bcast_u(logu, f);
        if (fbirth && ch.k >= KCAP) { status = CHAIN_MODEL_TOO_LARGE; aborted = true; break; }
        // move to the candidate model; the exact path (non-zero prior mean on the
        // variable) decides only now
        const double cur_logp = M.logp;
        const Model keep = M;
        apply_flip(ch, jf);
        rebuild();
        if (status != CHAIN_OK) break;
        const double ratio2 = (M.logp - ffwd) - (cur_logp - frev);
        bool acc = true;
        if ((m_slow >> f) & 1ull) {
          if (M.logp > -BA_INF && cur_logp > -BA_INF) AACC_MIN(ACC_MIN_MARGIN, fabs(flogu - ratio2));
          acc = flogu < ratio2;
        }
        if (acc) {
          // adjust_birth_rate / adjust_death_rate (.cpp:194-200, :228-234)
          double alpha = exp(ratio2);
          if (alpha > 1.0) alpha = 1.0;
          double adjustment = P.ada_step / ((1.0 + (double)iteration) / (double)p);
          adjustment *= (alpha - P.ada_target);
          lds_f64 *rate = fbirth ? birth : death;
          if (nundo >= ADA_UNDO_CAP) { status = CHAIN_RNG_BRANCH; break; }
          if (lane == 0) {
            undo_v[nundo] = rate[jf];
            undo_j[nundo] = fbirth ? jf : -1 - jf;
            rate[jf] = rate[jf] * exp(adjustment);
          }
          ++nundo;
          AACC_ADD(ACC_ACCEPTS, 1);
          if (!M.pd) { status = CHAIN_NOT_PD; break; }
          cum_dirty = true;
          wave_sync();
        } else {
          // rejected on the exact path (rare: a variable with a non-zero prior
          // mean): back to the standing model, factored again
          apply_flip(ch, jf);
          rebuild();
          (void)keep;
        }
      }
      if (status != CHAIN_OK) {
        if (aborted) {
          // the sweep leaves no trace: gamma, the rates it changed, the stream position
          for (int j = lane; j < p; j += WAVE) ch.gam[j] = ch.gam0[j];
          wave_sync();
          if (lane == 0) {
            for (int t = nundo - 1; t >= 0; --t) {
              const int code = undo_j[t];
              if (code >= 0) birth[code] = undo_v[t]; else death[-1 - code] = undo_v[t];
            }
          }
          pos = pos0;
          wave_sync();
        }
        break;
      }
    }
    // ---- set_posterior_moments is the standing model; draw_residual_variance
    k = ch.k;
    rng.set_pos(pos);
    if (P.draw_sigma) {
      int bad = 0;
      const double DF = (k == 0) ? ch.DF : ((ch.DF - P.prior_df) + P.prior_df);
      const double SS = (k == 0) ? ch.ss0q : ((M.SS - P.prior_ss) + P.prior_ss);
      sigsq = uni(d_draw_variance(rng, DF, SS, P.sigma_max, &bad));
      if (bad) { status = CHAIN_RNG_BRANCH; break; }
    }
    pos = uni(rng.get_pos());
    // ---- draw_coefficients: rmvn_ivar_mt(mean, V / sigma^2)
    if (P.draw_beta && k > 0) {
      if (!M.pd) { ++failures; status = CHAIN_NOT_PD; break; }
      failures = 0;
      const double z = draw_normals(rng, k);
      pos = uni(rng.get_pos());
      const double sigma = sqrt(sigsq);
      double y = (lane < k) ? ch.w[lane] + sigma * z : 0.0;
      const double rdm = (lane < k) ? ch.rdv[lane] : 0.0;
      // (eight rows of L at a time, as in ssvs_sweep_body.h)
#pragma nounroll
      for (int ib = (k - 1) >> 3; ib >= 0; --ib) {
        double lr[8];
        const int base = bidx(ib * 8, lane < ib * 8 + 8 ? lane : 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) lr[t] = ch.Lv[base + t * 8];
#pragma unroll
        for (int t = 7; t >= 0; --t) {
          const int r = ib * 8 + t;
          if (r < k) {
            const double xi = bcast_u(y * rdm, r);
            if (lane == r) y = xi;
            else if (lane < r) y -= lr[t] * xi;
          }
        }
      }
      beta_m = y;
      beta_valid = true;
    } else if (P.draw_beta) {
      beta_valid = true;
    }
    // ---- summaries, traces, the draw record
    gprev = (lane < k) ? (int)ch.g[lane] : 0;
    kprev = k;
    kmax = k > kmax ? k : kmax;
    if (lane < k) {
      const size_t o = (size_t)chain * p + gprev;
      const unsigned c0 = P.inc_count[o];   // (loads first, then the stores: one round trip, not three)
      const double b0 = P.beta_sum[o], q0 = P.beta_sumsq[o];
      P.inc_count[o] = c0 + 1u;
      if (beta_valid) {
        P.beta_sum[o] = b0 + beta_m;
        P.beta_sumsq[o] = q0 + beta_m * beta_m;
      }
    }
    AACC_ADD(ACC_SIGSQ, sigsq);
    AACC_ADD(ACC_SIGSQ2, sigsq * sigsq);
    AACC_ADD(ACC_K, k);
    if (P.trace_sigsq && trace_at + sweep < P.trace_stride) {
      const size_t o = (size_t)chain * P.trace_stride + trace_at + sweep;
      if (lane == 0) {
        P.trace_sigsq[o] = sigsq;
        P.trace_logp[o] = M.logp;
        P.trace_k[o] = (double)k;
      }
      if (P.rec_idx && lane < k) {
        P.rec_idx[o * P.rec_cap + lane] = (uint16_t)gprev;
        P.rec_beta[o * P.rec_cap + lane] = beta_valid ? beta_m : 0.0;
      }
    }
    ++iteration;
    ++done;
  }

  // ---- write the chain back
  wave_sync();
  for (int j = lane; j < p; j += WAVE) {
    g_gamma[j] = ch.gam[j];
    g_birth[j] = birth[j];
    g_death[j] = death[j];
  }
  if (beta_valid && done > 0) {
    double *g_beta = P.beta + (size_t)chain * p;
    for (int j = lane; j < p; j += WAVE) g_beta[j] = 0.0;
    wave_sync();
    if (lane < kprev) g_beta[gprev] = beta_m;
  } else if (flips_per_sweep > 0 && done > 0) {
    double *g_beta = P.beta + (size_t)chain * p;
    for (int j = lane; j < p; j += WAVE)
      if (!ch.gam[j]) g_beta[j] = 0.0;
  }
  if (lane == 0) {
    P.sigsq[chain] = sigsq;
    P.rng_pos[chain] = pos;
    P.ada_iter[chain] = iteration;
    P.failures[chain] = failures;
    P.status[chain] = status;
    P.todo[chain] = nsweeps - done;
    P.table_tag[chain] = 0;
    P.model_tag[chain] = 0;
    if (P.trace_idx) P.trace_idx[chain] = trace_at + done;
    if (P.maxk) atomicMax(P.maxk, kmax);
    double *a = P.acc + (size_t)chain * ACC_COUNT;
    a[ACC_SWEEPS] += done;
    a[ACC_SIGSQ] += ctl[CT_ACC + ACC_SIGSQ];
    a[ACC_SIGSQ2] += ctl[CT_ACC + ACC_SIGSQ2];
    a[ACC_K] += ctl[CT_ACC + ACC_K];
    a[ACC_ACCEPTS] += ctl[CT_ACC + ACC_ACCEPTS];
    a[ACC_PROPOSALS] += ctl[CT_ACC + ACC_PROPOSALS];
    a[ACC_MIN_MARGIN] = fmin(a[ACC_MIN_MARGIN], ctl[CT_ACC + ACC_MIN_MARGIN]);
    // (adaptive mode: scalar 8 carries the smallest relative distance of an
    // rmulti uniform from a boundary of the cumulative rates)
    a[ACC_PHASE0] = (a[ACC_PHASE0] == 0.0) ? ctl[CT_ACC + ACC_PHASE0]
                                           : fmin(a[ACC_PHASE0], ctl[CT_ACC + ACC_PHASE0]);
  }
}

template <int NB>
static hipError_t launch_adaptive_t(hipStream_t stream, const SsvsParams &P, int nsweeps) {
  const SsvsAdaLds lay = ssvs_ada_lds_layout(P.p, NB * 8);
  hipError_t e = hipFuncSetAttribute((const void *)ssvs_adaptive_kernel<NB>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total);
  if (e != hipSuccess) return e;
  KtScope kt(stream, KT_SSVS_ADAPTIVE);
  hipLaunchKernelGGL((ssvs_adaptive_kernel<NB>), dim3(P.chain_count), dim3(WAVE), lay.total, stream,
                     P, nsweeps);
  return hipGetLastError();
}

hipError_t launch_ssvs_adaptive(hipStream_t stream, const SsvsParams &P, int nsweeps) {
  switch (P.kcap) {
    case 16: return launch_adaptive_t<2>(stream, P, nsweeps);
    case 32: return launch_adaptive_t<4>(stream, P, nsweeps);
    case 48: return launch_adaptive_t<6>(stream, P, nsweeps);
    case 64: return launch_adaptive_t<8>(stream, P, nsweeps);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace boom_amd
