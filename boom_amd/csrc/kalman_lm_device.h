// The lane-major local-level state draw (kalman_lm_kernel) and the step that runs ahead of
// it (kalman_prepare_kernel) as device functions: kalman_kernel.hip wraps them in their own
// kernels, ss_round_kernel.hip runs them inside the chains' persistent round loop.
#pragma once
#include <hip/hip_runtime.h>

#include "device_rng.h"
#include "diag.h"
#include "kalman_params.h"
#include "stream_normals.h"

namespace boom_amd {

namespace {

// (diagnostic build -DBA_KSTAMPS: chain 0 prints its cycles per phase -- KSTAMP, diag.h)

#ifndef BA_HAVE_WAVE_HELPERS   // (ssvs_device.h, when it comes first, has the same four)
constexpr int WAVE = 64;

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x, double fill) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned long long f = __builtin_bit_cast(unsigned long long, fill);
  const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)f, (int)(unsigned)u, CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(f >> 32), (int)(unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double bcast_u(double x, int src) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int bcast_u(int x, int src) {
  return __builtin_amdgcn_readlane(x, src);
}
__device__ __forceinline__ double wave_sum(double x) {
  x += dpp_f64<0x118, 0xf>(x, 0.0);
  x += dpp_f64<0x114, 0xf>(x, 0.0);
  x += dpp_f64<0x112, 0xf>(x, 0.0);
  x += dpp_f64<0x111, 0xf>(x, 0.0);
  x += dpp_f64<0x142, 0xa>(x, 0.0);
  x += dpp_f64<0x143, 0xc>(x, 0.0);
  return bcast_u(x, 63);
}
#endif

typedef __attribute__((address_space(3))) double AS_LDS_F64;

// x -> A x + B
struct Aff { double A, B; };
__device__ __forceinline__ Aff aff_after(const Aff &later, const Aff &earlier) {
  Aff r;
  r.A = later.A * earlier.A;
  r.B = later.A * earlier.B + later.B;
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Aff aff_dpp(const Aff &f) {
  Aff r;
  r.A = dpp_f64<CTRL, ROW_MASK>(f.A, 1.0);  // lanes without a source get the identity
  r.B = dpp_f64<CTRL, ROW_MASK>(f.B, 0.0);
  return r;
}
// inclusive scan over the wave: lane i ends with f_i o f_{i-1} o ... o f_0
__device__ __forceinline__ Aff wave_scan(Aff f) {
  f = aff_after(f, aff_dpp<0x111, 0xf>(f));  // row_shr:1
  f = aff_after(f, aff_dpp<0x112, 0xf>(f));  // row_shr:2
  f = aff_after(f, aff_dpp<0x114, 0xf>(f));  // row_shr:4
  f = aff_after(f, aff_dpp<0x118, 0xf>(f));  // row_shr:8
  f = aff_after(f, aff_dpp<0x142, 0xa>(f));  // row_bcast:15 into rows 1, 3
  f = aff_after(f, aff_dpp<0x143, 0xc>(f));  // row_bcast:31 into rows 2, 3
  return f;
}
// x -> (a x + b) / (c x + d)
struct Mob { double a, b, c, d; };
__device__ __forceinline__ Mob mob_after(const Mob &l, const Mob &e) {
  Mob r;
  r.a = l.a * e.a + l.b * e.c;
  r.b = l.a * e.b + l.b * e.d;
  r.c = l.c * e.a + l.d * e.c;
  r.d = l.c * e.b + l.d * e.d;
  return r;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Mob mob_dpp(const Mob &f) {
  Mob r;
  r.a = dpp_f64<CTRL, ROW_MASK>(f.a, 1.0);
  r.b = dpp_f64<CTRL, ROW_MASK>(f.b, 0.0);
  r.c = dpp_f64<CTRL, ROW_MASK>(f.c, 0.0);
  r.d = dpp_f64<CTRL, ROW_MASK>(f.d, 1.0);
  return r;
}
__device__ __forceinline__ Mob wave_scan(Mob f) {
  f = mob_after(f, mob_dpp<0x111, 0xf>(f));
  f = mob_after(f, mob_dpp<0x112, 0xf>(f));
  f = mob_after(f, mob_dpp<0x114, 0xf>(f));
  f = mob_after(f, mob_dpp<0x118, 0xf>(f));
  f = mob_after(f, mob_dpp<0x142, 0xa>(f));
  f = mob_after(f, mob_dpp<0x143, 0xc>(f));
  return f;
}
// inclusive prefix sum, lane order
__device__ __forceinline__ double wave_prefix_sum(double x) {
  x += dpp_f64<0x111, 0xf>(x, 0.0);
  x += dpp_f64<0x112, 0xf>(x, 0.0);
  x += dpp_f64<0x114, 0xf>(x, 0.0);
  x += dpp_f64<0x118, 0xf>(x, 0.0);
  x += dpp_f64<0x142, 0xa>(x, 0.0);
  x += dpp_f64<0x143, 0xc>(x, 0.0);
  return x;
}
// the value of lane - 1 (lane 0 gets `first`): wave_shr:1
__device__ __forceinline__ double lane_before(double x, double first) {
  return dpp_f64<0x138, 0xf>(x, first);
}

// ---------------------------------------------------------------------------------
// The same state draw for a series of at most LM_TP = 2048 steps, in the LANE-MAJOR
// layout (kalman_params.h): thread i of the chain's 128 owns steps 16 i .. 16 i + 15
// through ALL passes, so what the passes hand to each other -- K_t, (v_t - v+_t) / F_t,
// alpha+_t, d_{t-1} -- never leaves its registers, and what comes from or goes to memory
// (X columns, y, the normals, the residuals, the state) is element 128 j + i for its
// j-th step: coalesced as it stands.  No work arrays, no transposes, one chunk.  The
// backward pass keeps the ownership and scans the lanes' composites in reverse order
// (wave 1 before wave 0, lane 63 before lane 0).
// Measured where the kernel above spent its time at T = 2000 (chain 0's cycle stamps):
// y* 33 k cycles of 86 k (one memory round trip per included variable and chunk, one
// after the other), the block transposes 11 k, the backward pass 12 k, the correction
// pass 22 k.  Here: the variables' columns two at a time, everything else as above.
struct LmSlots {   // slot s of the normals array: row s / 128 = 2 j + kind, thread s % 128
  int T, nfirst, nper, dI, dL, dH;
  __device__ __forceinline__ int count() const { return 2 * LM_TP; }
  // a time step's two normals (state error, observation: rows 2 j and 2 j + 1 of thread q % 128)
  // are one slot pair -- consecutive draws, i.e. the two halves of one Philox block's Box-Muller
  // pair whenever the step's first draw has an even global number (stream_normals.h)
  __device__ __forceinline__ int npairs(int) const { return LM_TP; }
  __device__ __forceinline__ void pair(int q, int, int *sa, int *sb) const {
    const int th = q & (LM_THREADS - 1), j = q >> 7;
    *sa = (2 * j) * LM_THREADS + th;
    *sb = (2 * j + 1) * LM_THREADS + th;
  }
  __device__ __forceinline__ int draw(int s) const {
    const int row = s >> 7, kind = row & 1;
    const int t = LM_BS * (s & (LM_THREADS - 1)) + (row >> 1);
    if (t >= T) return -1;
    if (t == 0) return kind == 0 ? (dI ? 0 : -1) : (dH ? dI : -1);
    if (kind == 0) return dL ? nfirst + (t - 1) * nper : -1;
    return dH ? nfirst + (t - 1) * nper + dL : -1;
  }
};

// The prepare step shared by the chain's two wavefronts (the round kernel,
// ss_round_kernel.hip): the wave that leads draws the level variance and posts the job; both
// take sub-chunks of SN_SUB slots from a counter in LDS until none is left (the regression's
// wave joins when its sweep is done); the leader waits for the last one and finishes the step.
enum : int { SN_SUB = 128 };   // slot PAIRS per sub-chunk (two per lane: one round of normals_pairs)
struct NormalsShare {
  int32_t seq;              // the job posted (> 0: its number), or -(number): none this time
  // the leader takes sub-chunks 0, 1, ... and writes `lo` = the next one it will take; the
  // helper takes nsub - 1, nsub - 2, ... and writes `hi` = the next one IT will take, `hfin` =
  // the lowest one it has finished.  Plain words, no read-modify-write: where the two meet
  // both may make the same sub-chunk -- the same numbers into the same places.
  int32_t lo, hi, hfin, bad;
  uint32_t pos_lo, pos_hi;  // the state stream's position the draws start from
  int32_t N, nfirst, nper, dI, dL, dH;
};
struct KalmanLmLds {
  NormalsLds norm;              // (only a chain whose normals were not prepared uses it)
  double x[2][8];               // the two waves' scan totals
  uint32_t mask[LM_THREADS];    // (H == 0 only: the threads' observed masks)
  NormalsShare share;           // (the round kernel only)
};
// Every thread of the chain's workgroup (128) calls it; returns true when the chain's state
// was drawn (false: the chain sat the round out or stopped -- the same in every thread).
// COHERENT: the residuals are read by OTHER workgroups of the same launch (the round
// kernel's X'e tiles): stored write-through at agent scope.
template <bool COHERENT>
__device__ __forceinline__ bool kalman_lm_body(const SsParams &P, const int draw_level, const int chain,
                                               KalmanLmLds &lds) {
  constexpr int BS = LM_BS, NT = LM_THREADS;
  constexpr int LM_VB = 5;                 // variables' columns in flight together in the y* pass
  NormalsLds &s_norm = lds.norm;
  double (&s_x)[2][8] = lds.x;
  uint32_t (&s_mask)[LM_THREADS] = lds.mask;
  int tid_ = threadIdx.x;
  BA_OPAQUE_V(tid_);
  const int tid = tid_, lane = tid & 63, wave = tid >> 6;
  // (the chain's scalars first, all in flight together, then the decisions)
  int32_t *prep_n_slot = P.prep_n + (size_t)P.zbuf * P.chains + chain;
  const int prep_n = *prep_n_slot, status_in = P.status[chain];
  const int ran = P.only_ran ? P.only_ran[chain] : 1;
  double level_sigsq = P.level_sigsq[chain];
  const double sigsq_obs = P.sigsq[chain];
  const bool prepared = P.prepared != 0 && prep_n > 0;
  // Every thread has read the chain's words before thread 0 rewrites any of them: a wave that
  // came to its loads late -- woken at a barrier after the other, its instructions not yet
  // fetched -- found the prepare step's count already taken back and went to make the normals
  // again, alone (one chain in a million rounds of the round kernel, ss_round_kernel.hip;
  // found by the chain's forecast variances coming out of an LDS exchange the waves no
  // longer shared).
  __syncthreads();
  if (P.prepared != 0 && prep_n < 0 && status_in == CHAIN_OK) {   // the prepare step failed
    if (tid == 0) { P.status[chain] = -prep_n; *prep_n_slot = 0; }
    return false;
  }
  if (status_in != CHAIN_OK || ran == 0) {
    if (prepared && tid == 0) {
      P.pos_state[chain] = P.prep_pos_state[(size_t)P.zbuf * P.chains + chain];
      P.pos_level[chain] = P.prep_pos_level[(size_t)P.zbuf * P.chains + chain];
      P.level_sigsq[chain] = P.prep_level_sigsq[(size_t)P.zbuf * P.chains + chain];
      *prep_n_slot = 0;
    }
    return false;
  }
  const int T = P.T, p = P.p;
  int status = CHAIN_OK;
#ifdef BA_KSTAMPS
  long long kph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, klast = (long long)__builtin_readcyclecounter();
#endif
  const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
  if (draw_level && !prepared) {
    SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, 1u}, P.pos_level[chain]};
    int bad = 0;
    const double DF = P.level_n[chain] + P.level_prior_df;
    const double SS = P.level_sumsq[chain] + P.level_prior_ss;
    level_sigsq = d_draw_variance(rng, DF, SS, P.level_sigma_max, &bad);
    if (bad) status = CHAIN_RNG_BRANCH;
    __syncthreads();   // (every thread has read the position it draws from)
    if (tid == 0) {
      P.pos_level[chain] = rng.pos;
      P.level_sigsq[chain] = level_sigsq;
    }
  }
  if (status != CHAIN_OK) {
    if (tid == 0) P.status[chain] = status;
    return false;
  }
  if (P.level_used && tid == 0) P.level_used[chain] = level_sigsq;
  const double *beta = P.beta + (size_t)chain * p;
  double *base = P.scratch + (size_t)chain * P.scratch_stride;
  double *sF = base + (size_t)P.TP;                         // residuals (input of the X'e GEMM)
  double *sst = base + (size_t)SS_STATE_ARRAY * P.TP;       // the state draw
  double *szz = base + (size_t)(5 + 2 * P.zbuf) * P.TP;     // the sweep's normals, LmSlots layout (2 TP; two buffers)
  const double q = level_sigsq, level_sigma = sqrt(level_sigsq);
  const double H = sigsq_obs, sqrtH = sqrt(H), sd0 = sqrt(P.P0);
  const int dI = (sd0 != 0.0), dL = (level_sigma != 0.0), dH = (sqrtH != 0.0);
  const int nfirst = dI + dH, nper = dL + dH;
  const int N = nfirst + (T - 1) * nper;
  KSTAMP(0);
  if (!prepared || prep_n != N) {   // (another count: see kalman_simsmooth_kernel)
    const uint64_t bpos0 = prepared ? P.prep_pos_state[(size_t)P.zbuf * P.chains + chain] : P.pos_state[chain];
    status = stream_normals(s_norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, bpos0, N, szz,
                            &P.pos_state[chain], LmSlots{T, nfirst, nper, dI, dL, dH}, ss_slot_serve(P));
    if (status != CHAIN_OK) {
      if (tid == 0) P.status[chain] = status;
      return false;
    }
    __syncthreads();
  }
  if (prepared && tid == 0) *prep_n_slot = 0;   // consumed
  KSTAMP(1);

  // the thread's steps: tl + j, j < 16; bit j of inm: inside the series, of obm: observed
  const int tl = BS * tid;
  const uint32_t inm = (tl + BS <= T) ? 0xffffu : (tl < T ? (1u << (T - tl)) - 1u : 0u);
  const uint32_t obm = P.obs_mask[tid] & inm;
#define LM_IN(j) (((inm >> (j)) & 1u) != 0u)
#define LM_OB(j) (((obm >> (j)) & 1u) != 0u)

  // ---- 1. y*_t = y_t - x_t'beta: the included variables' columns LM_VB at a time (their
  // loads in flight together); products accumulate in variable order, as GlmCoefs::predict
  double ys[BS];
  {
    // (y itself is asked for AFTER the variables' loop: asked for up front -- sixteen more
    // registers through a loop that holds LM_VB x 16 loads in flight -- the round kernel's
    // instance waited for each of its loads in turn to park it in scratch memory, ten memory
    // round trips in a row)
    double pred[BS], yv[BS];
#pragma unroll
    for (int j = 0; j < BS; ++j) pred[j] = 0.0;
    for (int vb = 0; vb < p; vb += WAVE) {
      const int jv = vb + lane;
      const double bj = (jv < p) ? beta[jv] : 0.0;
      unsigned long long mk = __ballot(bj != 0.0);
      while (mk) {
        // up to LM_VB variables of the batch (absent ones repeat the first with a zero
        // coefficient that is never added)
        int l[LM_VB];
        int cnt = 0;
#pragma unroll
        for (int v = 0; v < LM_VB; ++v) {
          const bool have = mk != 0;
          l[v] = have ? __ffsll((long long)mk) - 1 : l[0];
          if (have) { mk &= mk - 1; ++cnt; }
        }
        double b[LM_VB];
        const double *c[LM_VB];
#pragma unroll
        for (int v = 0; v < LM_VB; ++v) {
          b[v] = bcast_u(bj, l[v]);
          c[v] = P.Xt + (size_t)(vb + l[v]) * LM_TP + tid;
        }
        double x[LM_VB][BS];
#pragma unroll
        for (int v = 0; v < LM_VB; ++v)
#pragma unroll
          for (int j = 0; j < BS; ++j) x[v][j] = c[v][j * NT];
#pragma unroll
        for (int v = 0; v < LM_VB; ++v) {
          if (v < cnt) {
#pragma unroll
            for (int j = 0; j < BS; ++j) pred[j] += x[v][j] * b[v];
          }
        }
      }
    }
    int yoff = tid;
    asm volatile("" : "+v"(yoff) : "v"(pred[0]));   // (not before the loop: the compiler would put them back there)
#pragma unroll
    for (int j = 0; j < BS; ++j) yv[j] = P.yt[j * NT + yoff];
    // (and the differences taken here and now, in sixteen registers: left to itself the compiler
    // keeps y and the predictions apart until each step's first use, thirty-two registers more
    // through the loads of the normals)
#pragma unroll
    for (int j = 0; j < BS; ++j) ys[j] = yv[j] - pred[j];
    asm volatile("" : "+v"(ys[0]), "+v"(ys[1]), "+v"(ys[2]), "+v"(ys[3]), "+v"(ys[4]), "+v"(ys[5]), "+v"(ys[6]), "+v"(ys[7]),
                      "+v"(ys[8]), "+v"(ys[9]), "+v"(ys[10]), "+v"(ys[11]), "+v"(ys[12]), "+v"(ys[13]), "+v"(ys[14]), "+v"(ys[15]));
#pragma unroll
    for (int j = 0; j < BS; ++j) ys[j] = LM_IN(j) ? ys[j] : 0.0;
  }
  KSTAMP(2);
  __builtin_amdgcn_sched_barrier(0);   // (registers: y* is done before the normals come in)
  // ---- 2. the sweep's normals: state error (initial state at t = 0) and observation error
  double zL[BS], zH[BS];
#pragma unroll
  for (int j = 0; j < BS; ++j) {
    zL[j] = szz[(2 * j) * NT + tid];
    zH[j] = szz[(2 * j + 1) * NT + tid];
  }
#pragma unroll
  for (int j = 0; j < BS; ++j) {
    const bool first = (tl + j == 0);
    zL[j] = (LM_IN(j) && (first ? dI : dL)) ? zL[j] : 0.0;
    zH[j] = (LM_IN(j) && dH) ? zH[j] : 0.0;
  }
  KSTAMP(3);

  // ---- 3 + 4. forward pass (see kalman_simsmooth_kernel): variances as a scan of Moebius
  // maps, alpha+ as a prefix sum, the filter on w = y* - y+ as a scan of affine maps
  double K[BS], al[BS], ef[BS];
  {
    const bool moebius = H > 0.0;
    const double r = moebius ? q / H : 0.0, s1 = 1.0 / (1.0 + r);
    const double u_in = moebius ? P.P0 / H : P.P0;
    Mob M;
    M.a = 1.0; M.b = 0.0; M.c = 0.0; M.d = 1.0;
    double asum = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      Mob mt;
      mt.a = 1.0;
      mt.b = LM_OB(j) ? r * s1 : (LM_IN(j) ? r : 0.0);
      mt.c = LM_OB(j) ? s1 : 0.0;
      mt.d = LM_OB(j) ? s1 : 1.0;
      M = mob_after(mt, M);
      asum += !LM_IN(j) ? 0.0 : ((tl + j == 0) ? P.a0 + sd0 * zL[j] : level_sigma * zL[j]);
      al[j] = asum;
    }
    Mob G;
    G.a = 1.0; G.b = 0.0; G.c = 0.0; G.d = 1.0;
    if (moebius) G = wave_scan(M);
    const double aincl = wave_prefix_sum(asum);
    if (lane == WAVE - 1) {
      s_x[wave][0] = G.a; s_x[wave][1] = G.b; s_x[wave][2] = G.c; s_x[wave][3] = G.d;
      s_x[wave][4] = aincl;
    }
    if (!moebius) s_mask[tid] = obm | (inm << 16);
    __syncthreads();
    Mob G0;
    G0.a = s_x[0][0]; G0.b = s_x[0][1]; G0.c = s_x[0][2]; G0.d = s_x[0][3];
    const double a0tot = s_x[0][4];
    double w[BS];
    {
      const double carry = (wave == 1 ? a0tot : 0.0) + (aincl - asum);
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        al[j] += carry;
        w[j] = LM_IN(j) ? ys[j] - (al[j] + sqrtH * zH[j]) : 0.0;
      }
    }
    double Fv[BS];
    if (moebius) {
      Mob E;
      E.a = lane_before(G.a, 1.0);
      E.b = lane_before(G.b, 0.0);
      E.c = lane_before(G.c, 0.0);
      E.d = lane_before(G.d, 1.0);
      if (wave == 1) E = mob_after(E, G0);
      double u = (E.a * u_in + E.b) / (E.c * u_in + E.d);
#pragma unroll
      for (int j = 0; j < BS; ++j) {
        const double kk = u / (u + 1.0);
        Fv[j] = H * (u + 1.0);
        K[j] = LM_OB(j) ? kk : 0.0;
        u = LM_OB(j) ? kk + r : (LM_IN(j) ? u + r : u);
      }
    } else {
      // H == 0: the plain recursion, thread after thread (every thread runs all of it and
      // keeps the values of its own steps)
      double Pv = u_in;
      for (int th = 0; th < NT; ++th) {
        const uint32_t mm = s_mask[th];
        double Pl = Pv;
#pragma unroll
        for (int j = 0; j < BS; ++j) {
          const bool inx = ((mm >> (16 + j)) & 1u) != 0u, obx = ((mm >> j) & 1u) != 0u;
          const double PZ = Pl, Fi = PZ + H;
          const double Ki = obx ? PZ / Fi : 0.0;
          if (th == tid) { Fv[j] = Fi; K[j] = Ki; }
          if (obx) Pl = Pl + (-1.0) * PZ * Ki;
          if (inx) Pl = Pl + q;
        }
        Pv = Pl;
      }
    }
    bool badF = false;
#pragma unroll
    for (int j = 0; j < BS; ++j) badF = badF || (LM_IN(j) && !(Fv[j] > 0.0));
    const bool anybad = __any(badF) != 0;
    Aff C;
    C.A = 1.0; C.B = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      Aff f;
      f.A = 1.0 - K[j];
      f.B = K[j] * w[j];
      C = aff_after(f, C);
    }
    const Aff Gd = wave_scan(C);
    __syncthreads();   // (everyone has read the first exchange)
    if (lane == WAVE - 1) {
      s_x[wave][0] = Gd.A; s_x[wave][1] = Gd.B; s_x[wave][2] = anybad ? 1.0 : 0.0;
    }
    __syncthreads();
    Aff D0;
    D0.A = s_x[0][0]; D0.B = s_x[0][1];
    const bool bad_any = (s_x[0][2] != 0.0) || (s_x[1][2] != 0.0);
    __syncthreads();
    if (bad_any) {
      if (tid == 0) P.status[chain] = CHAIN_FORECAST_VARIANCE;
      return false;
    }
    Aff Ed;
    Ed.A = lane_before(Gd.A, 1.0);
    Ed.B = lane_before(Gd.B, 0.0);
    if (wave == 1) Ed = aff_after(Ed, D0);
    double delta = Ed.B;   // (delta_0 = 0 goes in)
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      ef[j] = LM_OB(j) ? (w[j] - delta) / Fv[j] : 0.0;   // (v_t - v+_t) / F_t
      delta = (1.0 - K[j]) * delta + K[j] * w[j];
    }
  }
  KSTAMP(4);

  // ---- 5. backward: d_{t-1} = e_t / F_t + (1 - K_t) d_t, d_{T-1} = 0, over the same
  // ownership: the thread's own steps latest first, d_{lo-1} = (f_lo o ... o f_hi)(d_hi),
  // then the composites in scan order = time downwards
  double dm[BS];   // d_{t-1} of the thread's own steps
  {
    Aff C;
    C.A = 1.0; C.B = 0.0;
#pragma unroll
    for (int j = BS - 1; j >= 0; --j) {
      Aff f;
      f.A = 1.0 - K[j];   // (1 at steps past T: K is 0 there, and e / F too)
      f.B = ef[j];
      C = aff_after(f, C);
    }
    Aff Cr;
    Cr.A = __shfl(C.A, WAVE - 1 - lane);
    Cr.B = __shfl(C.B, WAVE - 1 - lane);
    const Aff G = wave_scan(Cr);
    if (lane == WAVE - 1) { s_x[wave][0] = G.A; s_x[wave][1] = G.B; }
    __syncthreads();
    Aff D1;
    D1.A = s_x[1][0]; D1.B = s_x[1][1];
    Aff E;
    E.A = lane_before(G.A, 1.0);
    E.B = lane_before(G.B, 0.0);
    if (wave == 0) E = aff_after(E, D1);
    double d = __shfl(E.B, WAVE - 1 - lane);   // d at the thread's latest step (0 goes in at T - 1)
#pragma unroll
    for (int j = BS - 1; j >= 0; --j) {
      d = (1.0 - K[j]) * d + ef[j];
      dm[j] = d;
    }
    __syncthreads();
  }
  KSTAMP(5);

  // ---- 6. forward: mean correction m_t = P0 d_{-1} + q sum_{s<t} d_s, the state draw
  // alpha+_t + m_t, the level model's sufficient statistics, and the regression's given
  // the state: residual e_t = y_t - alpha_t at observed t (0 elsewhere), e'e, #observed
  // (nothing is stored before the last exchange: a barrier waits for the stores in flight)
  {
    // (loaded here, not under the forward pass: a barrier there would wait for it)
    double yv[BS];
    {
      // (four steps per address register, made HERE: the sixteen addresses computed at the
      // top of the kernel sat in scratch memory until now -- ten serial reloads)
      const double *yb = P.yt + tid;
#pragma unroll
      for (int g = 0; g < BS / 4; ++g) {
        const double *q = yb + (size_t)4 * g * NT;
        asm volatile("" : "+v"(q));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) yv[4 * g + jj] = q[jj * NT];
      }
    }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      acc += !LM_IN(j) ? 0.0 : ((tl + j == 0) ? P.P0 * dm[j] : q * dm[j]);
      dm[j] = acc;   // m_t less the carry
    }
    const double incl = wave_prefix_sum(acc);
    if (lane == WAVE - 1) s_x[wave][0] = incl;
    __syncthreads();
    const double carry = (wave == 1 ? s_x[0][0] : 0.0) + (incl - acc);
    double last = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      al[j] = LM_IN(j) ? al[j] + (dm[j] + carry) : 0.0;   // the state draw
      if (LM_IN(j)) last = al[j];
    }
    if (lane == WAVE - 1) s_x[wave][1] = last;
    __syncthreads();
    const double prev0 = lane_before(last, wave == 1 ? s_x[0][1] : 0.0);   // the state before the thread's first step
    double lev_ss_part = 0.0, part_q = 0.0, part_n = 0.0;
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      if (LM_IN(j) && tl + j > 0) {
        const double diff = al[j] - (j == 0 ? prev0 : al[j - 1]);
        lev_ss_part += diff * diff;
      }
      yv[j] = LM_OB(j) ? yv[j] - al[j] : 0.0;   // the residual
      if (LM_OB(j)) { part_q += yv[j] * yv[j]; part_n += 1.0; }
    }
    const double a = wave_sum(lev_ss_part), b2 = wave_sum(part_q), c = wave_sum(part_n);
    if (lane == 0) { s_x[wave][2] = a; s_x[wave][3] = b2; s_x[wave][4] = c; }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BS; ++j) {
      sst[j * NT + tid] = al[j];
      if (COHERENT)
        __hip_atomic_store(&sF[j * NT + tid], yv[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        sF[j * NT + tid] = yv[j];
    }
  }
#undef LM_IN
#undef LM_OB
  KSTAMP(6);
  if (tid == 0) {
    P.yty[chain] = s_x[0][3] + s_x[1][3];
    P.nobs[chain] = s_x[0][4] + s_x[1][4];
    P.level_n[chain] = (double)(T - 1);
    P.level_sumsq[chain] = s_x[0][2] + s_x[1][2];
#ifdef BA_KSTAMPS
    KSTAMP(7);
    if (chain == 0 && draw_level)
      printf("kalman lane-major phases (cycles): start %lld normals %lld ystar %lld z %lld forward %lld backward %lld correction %lld suf %lld\n",
             kph[0], kph[1], kph[2], kph[3], kph[4], kph[5], kph[6], kph[7]);
#endif
    P.status[chain] = status;
  }
  return true;
}

// The two pieces of a state draw that do not depend on the same round's regression
// sweep, done ahead of it on the engine's second stream (SsParams::prepared):
// ZeroMeanGaussianConjSampler::draw for the level variance (its own stream, the level
// model's sufficient statistics of the previous state draw) and the standard normals of
// simulate_forward (stream positions only), into the normals buffer P.zbuf.  The step
// for round r + 1 goes out behind round r's state draw and runs beside round r's X'e
// GEMM, plane sum and the start of round r + 1's SSVS launch (whose wavefronts raise
// their issue priority: the generator is bound by 32-bit multiplies, two such wavefronts
// to a SIMD took 9 us from a launch of one-wave chains).  Beside the state draw itself
// it cannot run: that kernel's wavefronts fill the register files (measured: 64 us
// instead of 30 when the prepare step's workgroups got there first).
// Which draws exist (a zero variance draws nothing, Bmath/rnorm.cpp:63-64) is taken from
// the level variance just drawn and a positive observation variance; the main kernel
// checks the count.  A failure is handed over as a negative count.
// ONE_WAVE: the calling wavefront alone (stream_normals.h).
// status_in: the chain's status word as the round found it (a chain that is parked or
// stopped prepares nothing).
template <bool ONE_WAVE>
__device__ __forceinline__ void kalman_prepare_body(const SsParams &P, const int draw_level, const int chain,
                                                    const int status_in, NormalsLds &s_norm) {
  typedef NormalsTeam<ONE_WAVE> Team;
  if (status_in != CHAIN_OK) return;
  const int T = P.T;
  const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
  const uint64_t pos_level0 = P.pos_level[chain], pos_state0 = P.pos_state[chain];
  const double level0 = P.level_sigsq[chain];
  double level_sigsq = level0;
  int status = CHAIN_OK;
  if (draw_level) {
    SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, 1u}, pos_level0};
    int bad = 0;
    const double DF = P.level_n[chain] + P.level_prior_df;
    const double SS = P.level_sumsq[chain] + P.level_prior_ss;
    level_sigsq = d_draw_variance(rng, DF, SS, P.level_sigma_max, &bad);
    if (bad) status = CHAIN_RNG_BRANCH;
    Team::sync();   // (every thread has read the statistics and the position it draws from)
    if (Team::tid() == 0) {
      P.pos_level[chain] = rng.pos;
      P.level_sigsq[chain] = level_sigsq;
    }
  }
  const int dI = (sqrt(P.P0) != 0.0), dL = (sqrt(level_sigsq) != 0.0), dH = 1;
  const int N = (dI + dH) + (T - 1) * (dL + dH);
  double *szz = P.scratch + (size_t)chain * P.scratch_stride + (size_t)(5 + 2 * P.zbuf) * P.TP;
  if (status == CHAIN_OK) {   // (uniform: every thread made the same draw)
    if (P.lane_major)
      status = stream_normals<ONE_WAVE>(s_norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, pos_state0, N, szz,
                              &P.pos_state[chain], LmSlots{T, dI + dH, dL + dH, dI, dL, dH}, ss_slot_serve(P));
    else
      status = stream_normals<ONE_WAVE>(s_norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, pos_state0, N, szz,
                                        &P.pos_state[chain], NormalsInOrder{N}, ss_slot_serve(P));
  }
  if (Team::tid() == 0) {
    const size_t slot = (size_t)P.zbuf * P.chains + chain;
    P.prep_n[slot] = (status == CHAIN_OK) ? N : -status;
    P.prep_pos_state[slot] = pos_state0;
    P.prep_pos_level[slot] = pos_level0;
    P.prep_level_sigsq[slot] = level0;
  }
}

// ---- the same step with the sub-chunks of the normals shared by the two wavefronts
__device__ __forceinline__ int lds_ld(const int32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(int32_t *p, int32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The calling wavefront's part of a shared job: sub-chunks of SN_SUB slot pairs (a pair = one
// Philox block, two normals: stream_normals.h; round 6 -- no lists, no second phase).
struct NormalsShareCtx {
  LmSlots slots;
  uint64_t bslot0;
  PhiloxKey key;
  double *szz;
};
__device__ __forceinline__ NormalsShareCtx normals_share_ctx(const SsParams &P, const int chain, KalmanLmLds &lds) {
  NormalsShare &J = lds.share;
  return NormalsShareCtx{LmSlots{P.T, J.nfirst, J.nper, J.dI, J.dL, J.dH},
                         (((uint64_t)J.pos_hi << 32) | J.pos_lo) / STATE_SLOT_STRIDE,
                         PhiloxKey{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), 2u},
                         P.scratch + (size_t)chain * P.scratch_stride + (size_t)(5 + 2 * P.zbuf) * P.TP};
}
__device__ __forceinline__ void normals_share_chunk(const NormalsShareCtx &X, const int c) {
  const int S = X.slots.npairs(0), c0 = c * SN_SUB, nc = (S - c0 < SN_SUB) ? S - c0 : SN_SUB;
  normals_pairs<NormalsTeam<true>>(X.key, X.bslot0, X.szz, X.slots, c0, nc);
}
// the leading wavefront: what kalman_prepare_body does, the normals through the shared job;
// seq: the job's number (the other wavefront asks for it by that number)
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
// hook: called between the sub-chunks (the round kernel: has the chain's X'e tile formed? then
// this wavefront's share of the tile product goes first)
template <class Hook = NoHook>
__device__ __forceinline__ void kalman_prepare_lead(const SsParams &P, const int chain, const int status_in, const int seq,
                                                    KalmanLmLds &lds, Hook hook = Hook()) {
  typedef NormalsTeam<true> Team;
  NormalsShare &J = lds.share;
  const int lane = Team::tid();
  if (status_in != CHAIN_OK) {   // (a chain that is parked or stopped prepares nothing)
    if (lane == 0) lds_st(&J.seq, -seq);
    return;
  }
  const int T = P.T;
  const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
  const uint64_t pos_level0 = P.pos_level[chain], pos_state0 = P.pos_state[chain];
  const double level0 = P.level_sigsq[chain];
  double level_sigsq = level0;
  int status = CHAIN_OK;
  {
    SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, 1u}, pos_level0};
    int bad = 0;
    const double DF = P.level_n[chain] + P.level_prior_df;
    const double SS = P.level_sumsq[chain] + P.level_prior_ss;
    level_sigsq = d_draw_variance(rng, DF, SS, P.level_sigma_max, &bad);
    if (bad) status = CHAIN_RNG_BRANCH;
    Team::sync();   // (every lane has read the statistics and the position it draws from)
    if (lane == 0) {
      P.pos_level[chain] = rng.pos;
      P.level_sigsq[chain] = level_sigsq;
    }
  }
  const int dI = (sqrt(P.P0) != 0.0), dL = (sqrt(level_sigsq) != 0.0), dH = 1;
  const int N = (dI + dH) + (T - 1) * (dL + dH);
  if (status == CHAIN_OK) {
    const int nsub = (LM_TP + SN_SUB - 1) / SN_SUB;
    if (lane == 0) {
      J.lo = 0; J.hi = nsub - 1; J.hfin = nsub; J.bad = 0;
      J.pos_lo = (uint32_t)pos_state0; J.pos_hi = (uint32_t)(pos_state0 >> 32);
      J.N = N; J.nfirst = dI + dH; J.nper = dL + dH; J.dI = dI; J.dL = dL; J.dH = dH;
      lds_st(&J.seq, seq);
    }
    Team::sync();
    const NormalsShareCtx X = normals_share_ctx(P, chain, lds);
    int c = 0;
    for (; c < nsub; ++c) {
      if (c > __builtin_amdgcn_readfirstlane(lds_ld(&J.hi))) break;   // (the helper has the rest)
      if (lane == 0) lds_st(&J.lo, c + 1);
      normals_share_chunk(X, c);
      hook();
    }
    // (what the helper took is finished: sub-chunks c .. nsub - 1)
    while (__builtin_amdgcn_readfirstlane(lds_ld(&J.hfin)) > c) __builtin_amdgcn_s_sleep(1);
    if (lds_ld(&J.bad)) status = CHAIN_RNG_BRANCH;
    if (lane == 0) P.pos_state[chain] = pos_state0 + (uint64_t)N * STATE_SLOT_STRIDE;
  } else if (lane == 0) {
    lds_st(&J.seq, -seq);
  }
  if (lane == 0) {
    const size_t slot = (size_t)P.zbuf * P.chains + chain;
    P.prep_n[slot] = (status == CHAIN_OK) ? N : -status;
    P.prep_pos_state[slot] = pos_state0;
    P.prep_pos_level[slot] = pos_level0;
    P.prep_level_sigsq[slot] = level0;
  }
}
// the other wavefront, when it has nothing else to do: job `seq`'s sub-chunks, if there are any left
__device__ __forceinline__ void kalman_prepare_help(const SsParams &P, const int chain, const int seq, KalmanLmLds &lds) {
  NormalsShare &J = lds.share;
  int s;
  while ((s = lds_ld(&J.seq)) != seq && s != -seq) __builtin_amdgcn_s_sleep(1);
  if (s > 0) {
    const int lane = NormalsTeam<true>::tid();
    int lowest = -1;
    NormalsShareCtx X{};
    for (;;) {
      const int h = __builtin_amdgcn_readfirstlane(lds_ld(&J.hi));
      if (h < __builtin_amdgcn_readfirstlane(lds_ld(&J.lo))) break;   // (the leader has it, or had)
      if (lane == 0) lds_st(&J.hi, h - 1);
      if (lowest < 0) X = normals_share_ctx(P, chain, lds);
      normals_share_chunk(X, h);
      lowest = h;
    }
    if (lowest >= 0) {
      // (this wave's stores are out before the leader is told)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) lds_st(&J.hfin, lowest);
    }
  }
}

}  // namespace

}  // namespace boom_amd
