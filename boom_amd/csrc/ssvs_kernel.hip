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
#include "ktimer.h"
#include "ssvs_sweep_body.h"

namespace boom_amd {

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

// The kernel: which chain, the sweeps, and -- between launches that overlap (engine.hip,
// pipelined sweeps) -- the hand-over.  A launch lasts as long as its slowest chain; when the
// next launch is already queued on another stream its workgroups move into the slots the
// early finishers leave, each taking over a chain that IS done (in the order the chains
// finish), so nothing idles between two launches and no chain is ever in two places.
enum : int { Q_PUSH = 0, Q_POP = 1, Q_READY = 2 };
template <int NB, int W, int WPE>
__global__ __launch_bounds__(64 * W, WPE) void ssvs_sweep_kernel(SsvsParams P, int nsweeps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_chain;
  if ((int)blockIdx.x >= P.chain_count) return;
  // A one-wave chain's sweep is one long chain of dependent instructions: whenever this
  // wavefront has one ready it goes first (the state-space path runs its normals
  // generator, two multiply-bound wavefronts to a SIMD, beside the start of this launch).
  // (Not for the multi-wave instances: with every wavefront of the launch raised, config
  // 2 measured 43.7 instead of 44.6 M sweeps/s.)
  if (W == 1) __builtin_amdgcn_s_setprio(3);
  int chain = (int)blockIdx.x + P.chain_first;
  if (P.q_in) {
    if (threadIdx.x == 0) {
      // every workgroup of this launch started in a slot that a workgroup of the previous
      // launch left AFTER appending its chain, or beside previous-launch workgroups that are
      // still running and will append: ticket t is served after a bounded wait
      const int t = atomicAdd(P.q_in + Q_POP, 1);
      int c = -1;
      for (int spin = 0; spin < (1 << 22); ++spin) {
        c = __hip_atomic_load(P.q_in + Q_READY + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if (c >= 0) break;
        __builtin_amdgcn_s_sleep(32);
      }
      s_chain = c;
    }
    __syncthreads();
    chain = s_chain;
    if (chain < 0) {   // (seconds of waiting: reported, not hung)
      if (threadIdx.x == 0) atomicExch(P.q_error, 1);
      return;
    }
    __threadfence();   // the chain's state as the workgroup that appended it left it
  }
  if (P.snap_gamma) {
    // the chain as this launch finds it (la_copy's list, engine.hip)
    const size_t o = (size_t)chain * (size_t)P.p;
    for (int j = threadIdx.x; j < P.p; j += WAVE * W) {
      P.snap_gamma[o + j] = P.gamma[o + j];
      P.snap_beta[o + j] = P.beta[o + j];
      P.snap_perm[o + j] = P.perm[o + j];
      P.snap_inc[o + j] = P.inc_count[o + j];
      P.snap_bsum[o + j] = P.beta_sum[o + j];
      P.snap_bsumsq[o + j] = P.beta_sumsq[o + j];
    }
    if (threadIdx.x < ACC_COUNT)
      P.snap_acc[(size_t)chain * ACC_COUNT + threadIdx.x] = P.acc[(size_t)chain * ACC_COUNT + threadIdx.x];
    if (threadIdx.x == 0) {
      P.snap_sigsq[chain] = P.sigsq[chain];
      P.snap_pos[chain] = P.rng_pos[chain];
      P.snap_fail[chain] = P.failures[chain];
    }
  }
  ssvs_sweep_body<NB, W, WPE>(P, nsweeps, chain, smem);
  if (P.q_out) {
    __syncthreads();
    __threadfence();
    if (threadIdx.x == 0) {
      const int t = atomicAdd(P.q_out + Q_PUSH, 1);
      __hip_atomic_store(P.q_out + Q_READY + t, chain, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
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
  KtScope kt(stream, KT_SSVS);
  hipLaunchKernelGGL((ssvs_sweep_kernel<NB, W, WPE>), dim3(P.chain_count), dim3(WAVE * W),
                     lay.total, stream, P, nsweeps);
  return hipGetLastError();
}

hipError_t launch_ssvs_sweep(hipStream_t stream, const SsvsParams &P,
                             int nsweeps) {
  const int key = P.kcap * 10 + P.waves;
  switch (key) {
#ifdef BA_ONLY_322   // (compile-time experiments on the hot instance only)
    case 322: return launch_sweep_t<4, 2, 2>(stream, P, nsweeps);
    default: return hipErrorInvalidValue;
#else
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
#endif
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
