// Device-side random number streams for the many-chain engine (gfx950).
//
// One Philox4x32-10 counter stream per (sampler seed, global chain id, stream
// id) stands in for the private BOOM::RNG every PosteriorSampler owns
// (Models/PosteriorSamplers/PosteriorSampler.cpp:34-40, distributions/rng.hpp:27-54).
// Uniform number i of a stream is 64-bit half (i & 1) of Philox block (i >> 1),
// mapped to [0,1) as (x >> 11) * 2^-53, so any lane can fetch any position:
// the p-1 shuffle uniforms and the p flip uniforms of a sweep are generated
// lane-parallel, the data-dependent tail (swap move, sigma, beta) walks the
// stream sequentially exactly like the reference's rng() calls.
//
// PROVENANCE.  The transforms below have to reproduce, draw for draw, what the
// reference's Bmath routines make of a stream of uniforms -- same constants,
// same operation order, same number of uniforms consumed on every branch --
// because the parity bar is "the reference's chain on the same stream".  Bmath is
// BOOM's in-tree fork of R's nmath (GPL-2+ / LGPL, (C) R Core Team and the
// authors below); the algorithms are published ones:
//   normal       A. J. Kinderman and J. G. Ramage (1976), "Computer generation of
//                normal random variables", JASA 71, 893-896, with J. Leydold's
//                correction of the tail branch (R's KINDERMAN_RAMAGE)
//   exponential  J. H. Ahrens and U. Dieter (1972), "Computer methods for sampling
//                from the exponential and normal distributions", CACM 15, 873-882
//   gamma        J. H. Ahrens and U. Dieter (1982), "Generating gamma variates by
//                a modified rejection technique" (GD, a >= 1), CACM 25, 47-54;
//                (1974) "Computer methods for sampling from gamma, beta, Poisson
//                and binomial distributions" (GS, a < 1), Computing 12, 223-246
// The arithmetic therefore cannot differ from nmath's; what is ours is the
// form: templates over the stream type (sequential Philox view / register
// window), loops instead of recursion, no R_FINITE / ML_ERROR handling (the
// callers guarantee the argument ranges), wave-uniform execution.
//
// The transforms follow the reference's Bmath routines so that a chain
// consumes the stream in the reference's order:
//   runif_mt      Bmath/runif.cpp:45-52
//   random_int_mt distributions/random_int.cpp:26-29
//   norm_rand     Bmath/snorm.cpp:287-340 (Kinderman-Ramage, Leydold fix)
//   exp_rand      Bmath/sexp.cpp:58-104
//   rgamma_mt     Bmath/rgamma.cpp:80-259 (GD for a >= 1, GS for .3 <= a < 1)
//   rtrun_gamma   distributions/trun_gamma.cpp:73-148 (rejection, adaptive
//                 rejection (BoundedAdaptiveRejectionSampler.cpp), slice)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The round kernel (ss_round_kernel.hip) runs the sweep and the state draw inside a loop
// over rounds; the compiler then hoists every round-invariant address computation of both
// to the top of the kernel and keeps it alive -- in scratch memory -- across everything.
// There (BA_ROUND_KERNEL) the thread and chain indices are made opaque once per round, so
// that what is derived from them is computed where it is used.  Elsewhere: nothing.
#ifdef BA_ROUND_KERNEL
#define BA_OPAQUE_V(x) asm volatile("" : "+v"(x))
#define BA_OPAQUE_S(x) asm volatile("" : "+s"(x))
#else
#define BA_OPAQUE_V(x) do { } while (0)
#define BA_OPAQUE_S(x) do { } while (0)
#endif

namespace boom_amd {

struct PhiloxKey {
  uint32_t k0, k1;   // sampler seed
  uint32_t chain;    // global chain id
  uint32_t stream;   // sampler id within the chain
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1,
                                              uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // full 32 x 32 -> 64 products (one v_mad_u64_u32 each instead of a
    // v_mul_hi_u32 / v_mul_lo_u32 pair)
    const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// uniform number `idx` of the stream, in [0, 1)
__device__ __forceinline__ double philox_uniform(const PhiloxKey &key,
                                                 uint64_t idx) {
  const uint64_t block = idx >> 1;
  uint32_t o[4];
  philox4x32_10((uint32_t)block, (uint32_t)(block >> 32), key.chain,
                key.stream, key.k0, key.k1, o);
  const uint64_t x = (idx & 1) ? ((uint64_t)o[2] | ((uint64_t)o[3] << 32))
                               : ((uint64_t)o[0] | ((uint64_t)o[1] << 32));
  return (double)(x >> 11) * 0x1.0p-53;
}

// both uniforms of one Philox block: numbers 2 * block and 2 * block + 1
__device__ __forceinline__ void philox_pair(const PhiloxKey &key, uint64_t block,
                                            double *u0, double *u1) {
  uint32_t o[4];
  philox4x32_10((uint32_t)block, (uint32_t)(block >> 32), key.chain,
                key.stream, key.k0, key.k1, o);
  *u0 = (double)(((uint64_t)o[0] | ((uint64_t)o[1] << 32)) >> 11) * 0x1.0p-53;
  *u1 = (double)(((uint64_t)o[2] | ((uint64_t)o[3] << 32)) >> 11) * 0x1.0p-53;
}

// A SLOT of a substream (stream_normals.h, the imputers of probit_kernel.hip): draw `index`
// of a stream owns the positions [index * stride, (index + 1) * stride).  A draw that needs
// more uniforms than its slot serves goes on in the slot's SPILL stream -- the same chain,
// the stream id with its top bit set, position index << SPILL_SHIFT -- instead of reading
// the next draw's numbers (rounds 1-3 stopped the chain there).  The oracle's Philox mode
// does the same (bo_rng_slot).  A million uniforms further the draw is given up.
enum : uint32_t { SPILL_STREAM_BIT = 0x80000000u };
enum : int { SPILL_SHIFT = 20 };

// Sequential view of a stream (the reference's `rng()`).
struct SeqRng {
  PhiloxKey key;
  uint64_t pos;
  uint64_t limit = ~0ull;   // a slot: the first position that is not its own ...
  uint64_t spill = 0;       // ... and where the draw goes on in the spill stream
  __device__ __forceinline__ double operator()() {
    if (pos >= limit) { key.stream |= SPILL_STREAM_BIT; pos = spill; limit = ~0ull; }
    return philox_uniform(key, pos++);
  }
  // slot `index` of a stream of `stride` positions per draw, `serve` <= stride of them handed out
  static __device__ __forceinline__ SeqRng slot(const PhiloxKey &k, uint64_t index, uint32_t stride, uint32_t serve) {
    SeqRng r{k, index * stride};
    r.limit = r.pos + serve;
    r.spill = index << SPILL_SHIFT;
    return r;
  }
  __device__ __forceinline__ bool overran() const {
    return (key.stream & SPILL_STREAM_BIT) && pos - spill > (1ull << SPILL_SHIFT);
  }
};

// Sequential view that keeps the Philox block it last computed: the two numbers of a
// block cost one evaluation (a draw that starts at an even position and consumes two
// uniforms -- the common case of norm_rand -- is one Philox call).
struct PairRng {
  PhiloxKey key;
  uint64_t pos, block;
  double w0, w1;
  bool have;
  uint64_t limit, spill;   // (see SeqRng)
  __device__ __forceinline__ void init(const PhiloxKey &k, uint64_t p) {
    key = k; pos = p; block = 0; w0 = w1 = 0.0; have = false; limit = ~0ull; spill = 0;
  }
  __device__ __forceinline__ void init_slot(const PhiloxKey &k, uint64_t index, uint32_t stride, uint32_t serve) {
    init(k, index * stride);
    limit = pos + serve;
    spill = index << SPILL_SHIFT;
  }
  __device__ __forceinline__ double operator()() {
    if (pos >= limit) { key.stream |= SPILL_STREAM_BIT; pos = spill; limit = ~0ull; have = false; }
    const uint64_t b = pos >> 1;
    if (!have || b != block) {
      philox_pair(key, b, &w0, &w1);
      block = b;
      have = true;
    }
    const double u = (pos & 1) ? w1 : w0;
    ++pos;
    return u;
  }
  __device__ __forceinline__ bool overran() const {
    return (key.stream & SPILL_STREAM_BIT) && pos - spill > (1ull << SPILL_SHIFT);
  }
};

// The same sequential view for code that a whole wavefront executes in
// lockstep (all 64 lanes active, wave-uniform control flow): the wave
// generates a window of 128 consecutive uniforms at once -- lane l holds both
// numbers of block (window base >> 1) + l -- and every rng() is a v_readlane
// instead of ten Philox rounds.  The cursor is a wave-uniform 32-bit offset so
// that the bookkeeping stays on the scalar unit.
struct WinRng {
  PhiloxKey key;
  int lane;
  uint64_t wbase;   // even stream position of the window's first number
  int off;          // next number = wbase + off
  bool have;        // a window is loaded (then off <= 128)
  double w0, w1;
  __device__ __forceinline__ void init(const PhiloxKey &k, int l, uint64_t p) {
    key = k; lane = l; w0 = w1 = 0.0;
    wbase = p; off = 0; have = false;
  }
  __device__ __forceinline__ uint64_t get_pos() const { return wbase + (uint64_t)off; }
  __device__ __forceinline__ void fill(uint64_t p) {
    wbase = p & ~1ull;
    off = __builtin_amdgcn_readfirstlane((int)(p & 1ull));
    have = true;
    philox_pair(key, (wbase >> 1) + (uint64_t)lane, &w0, &w1);
  }
  // move the cursor (forward or back inside the window keeps the window)
  __device__ __forceinline__ void set_pos(uint64_t p) {
    if (have && p >= wbase && p - wbase <= 128) {
      off = __builtin_amdgcn_readfirstlane((int)(p - wbase));
    } else {
      wbase = p;
      off = 0;
      have = false;
    }
  }
  __device__ __forceinline__ double operator()() {
    if (!have || off > 127) fill(get_pos());
    const int o = off;
    off = o + 1;
    const double w = (o & 1) ? w1 : w0;
    const int src = o >> 1;
    const int lo = __builtin_amdgcn_readlane(__double2loint(w), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(w), src);
    return __hiloint2double(hi, lo);
  }
};

// key / position of either view of a stream
__device__ __forceinline__ PhiloxKey stream_key(const SeqRng &r) { return r.key; }
__device__ __forceinline__ PhiloxKey stream_key(const WinRng &r) { return r.key; }
__device__ __forceinline__ uint64_t stream_pos(const SeqRng &r) { return r.pos; }
__device__ __forceinline__ uint64_t stream_pos(const WinRng &r) { return r.get_pos(); }
__device__ __forceinline__ void stream_seek(SeqRng &r, uint64_t p) { r.pos = p; }
__device__ __forceinline__ void stream_seek(WinRng &r, uint64_t p) {
  // (wave-uniform by construction)
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32));
  r.set_pos(((uint64_t)hi << 32) | lo);
}

template <class R>
__device__ __forceinline__ double d_runif(R &r, double a, double b) {
  if (a == b) return a;
  return a + (b - a) * r();
}

template <class R>
__device__ __forceinline__ int d_random_int(R &r, int lo, int hi) {
  return (int)floor(d_runif(r, (double)lo, (double)(hi + 1)));
}

template <class R>
__device__ __forceinline__ double d_norm_rand(R &r) {
  const double A = 2.216035867166471;
  const double C1 = 0.398942280401433, C2 = 0.180025191068563;
#define BA_KR_G(x) (C1 * exp(-(x) * (x) / 2.0) - C2 * (A - (x)))
  double u1 = r(), u2, u3, tt;
  if (u1 < 0.884070402298758) {
    u2 = r();
    return A * (1.131131635444180 * u1 + u2 - 1);
  }
  if (u1 >= 0.973310954173898) {
    for (;;) {
      u2 = r();
      u3 = r();
      tt = (A * A - 2 * log(u3));
      if (u2 * u2 < (A * A) / tt)
        return (u1 < 0.986655477086949) ? sqrt(tt) : -sqrt(tt);
    }
  }
  if (u1 >= 0.958720824790463) {
    for (;;) {
      u2 = r();
      u3 = r();
      tt = A - 0.630834801921960 * fmin(u2, u3);
      if (fmax(u2, u3) <= 0.755591531667601) return (u2 < u3) ? tt : -tt;
      if (0.034240503750111 * fabs(u2 - u3) <= BA_KR_G(tt))
        return (u2 < u3) ? tt : -tt;
    }
  }
  if (u1 >= 0.911312780288703) {
    for (;;) {
      u2 = r();
      u3 = r();
      tt = 0.479727404222441 + 1.105473661022070 * fmin(u2, u3);
      if (fmax(u2, u3) <= 0.872834976671790) return (u2 < u3) ? tt : -tt;
      if (0.049264496373128 * fabs(u2 - u3) <= BA_KR_G(tt))
        return (u2 < u3) ? tt : -tt;
    }
  }
  for (;;) {
    u2 = r();
    u3 = r();
    tt = 0.479727404222441 - 0.595507138015940 * fmin(u2, u3);
    if (tt < 0.) continue;
    if (fmax(u2, u3) <= 0.805577924423817) return (u2 < u3) ? tt : -tt;
    if (0.053377549506886 * fabs(u2 - u3) <= BA_KR_G(tt))
      return (u2 < u3) ? tt : -tt;
  }
#undef BA_KR_G
}

// rnorm_mt, Bmath/rnorm.cpp:55-67: no draw when sigma == 0
template <class R>
__device__ __forceinline__ double d_rnorm(R &r, double mu, double sigma) {
  if (sigma == 0.) return mu;
  return mu + sigma * d_norm_rand(r);
}

template <class R>
__device__ __forceinline__ double d_exp_rand(R &r) {
  // q[k-1] = sum_{i=1..k} log(2)^i / i!
  const double q[16] = {
      0.6931471805599453, 0.9333736875190459, 0.9888777961838675,
      0.9984959252914960, 0.9998292811061389, 0.9999833164100727,
      0.9999985691438767, 0.9999998906925558, 0.9999999924734159,
      0.9999999995283275, 0.9999999999728814, 0.9999999999985598,
      0.9999999999999289, 0.9999999999999968, 0.9999999999999999,
      1.0000000000000000};
  double a = 0., u = r();
  while (u <= 0.0 || u >= 1.0) u = r();
  for (;;) {
    u += u;
    if (u > 1.0) break;
    a += q[0];
  }
  u -= 1.;
  if (u <= q[0]) return a + u;
  int i = 0;
  double ustar = r(), umin = ustar;
  do {
    ustar = r();
    if (ustar < umin) umin = ustar;
    i++;
  } while (u > q[i]);
  return a + umin * q[0];
}

// rloggamma_small_alpha (Bmath/rloggamma_small_alpha.cpp:43-79): log of a
// Gamma(alpha, 1) draw for alpha < 0.3, the rejection sampler of Liu, Martin and
// Syring.  A rare branch (shape = DF / 2 < 0.3 needs fewer than 0.6 degrees of
// freedom); kept out of line.  *bad = 1 after 1000 rejections, as the reference.
#ifndef BA_RARE
#define BA_RARE __forceinline__
#endif
__device__ BA_RARE double d_rloggamma_small_alpha(SeqRng &rng, double alpha, int *bad) {
  const double e = 2.718281828459045;   // exp(1)
  const double w = alpha / (e * (1 - alpha));
  const double r = 1.0 / (1 + w);
  const double lambda = (1.0 / alpha) - 1.0;
  const double log_w = log(w), log_lambda = log(lambda);
  for (int i = 0; i < 1000; ++i) {
    const double u = rng();
    const double z = (u <= r) ? -log(u / r) : log(rng()) / lambda;
    const double log_h = -z - exp(-z / alpha);
    const double log_eta = (z >= 0) ? -z : log_w + log_lambda + lambda * z;
    if (log_h >= log(rng()) + log_eta) return -z / alpha;
  }
  *bad = 1;
  return 0.0;
}

// Rmath::rgamma_mt(rng, a, scale): rloggamma_small_alpha for a < 0.3, GS for
// a < 1, GD otherwise (Bmath/rgamma.cpp:80-259).
template <class R>
__device__ __forceinline__ double d_rgamma_scale(R &rng, double a, double scale,
                                        int *bad) {
  const double kSqrt32 = 5.656854, kInvE = 0.36787944117144232159;
  const double q1 = 0.04166669, q2 = 0.02083148, q3 = 0.00801191,
               q4 = 0.00144121, q5 = -7.388e-5, q6 = 2.4511e-4, q7 = 2.424e-4;
  const double a1 = 0.3333333, a2 = -0.250003, a3 = 0.2000062,
               a4 = -0.1662921, a5 = 0.1423657, a6 = -0.1367177,
               a7 = 0.1233795;
  if (a < .3) {
    // (the out-of-line routine reads the stream through the plain sequential
    // view, whatever view the caller uses: same numbers, same positions)
    SeqRng sr{stream_key(rng), stream_pos(rng)};
    int b2 = 0;
    const double lg = d_rloggamma_small_alpha(sr, a, &b2);
    stream_seek(rng, sr.pos);
    *bad |= __builtin_amdgcn_readfirstlane(b2);
    const double x = exp(lg + log(scale));
    const unsigned long long xb = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xb);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xb >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  }
  if (a < 1.) {  // GS
    const double e = 1.0 + kInvE * a;
    double x;
    for (;;) {
      for (;;) {
        const double p = e * rng();
        if (p >= 1.0) {
          x = -log((e - p) / a);
          if (d_exp_rand(rng) >= (1.0 - a) * log(x)) break;
        } else {
          x = exp(log(p) / a);
          if (d_exp_rand(rng) >= x) break;
        }
      }
      if (x > 0) return scale * x;
    }
  }
  const double s2 = a - 0.5, s = sqrt(s2), d = kSqrt32 - s * 12.0;
  double t = d_norm_rand(rng);
  double x = s + 0.5 * t;
  const double x_squared = x * x;
  if (t >= 0.0) return scale * x_squared;
  double u = rng();
  if (d * u <= t * t * t) return scale * x_squared;
  const double rr = 1.0 / a;
  const double q_zero =
      ((((((q7 * rr + q6) * rr + q5) * rr + q4) * rr + q3) * rr + q2) * rr + q1) * rr;
  double b, slope, c;
  if (a <= 3.686) {
    b = 0.463 + s + 0.178 * s2;
    slope = 1.235;
    c = 0.195 / s - 0.079 + 0.16 * s;
  } else if (a <= 13.022) {
    b = 1.654 + 0.0076 * s2;
    slope = 1.68 / s + 0.275;
    c = 0.062 / s + 0.024;
  } else {
    b = 1.77;
    slope = 0.75;
    c = 0.1515 / s;
  }
  double q, v;
  if (x > 0.0) {
    v = t / (s + s);
    if (fabs(v) <= 0.25)
      q = q_zero + 0.5 * t * t *
                   ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
    else
      q = q_zero - s * t + 0.25 * t * t + (s2 + s2) * log1p(v);
    if (log(1.0 - u) <= q) return scale * x_squared;
  }
  for (;;) {
    const double e = d_exp_rand(rng);
    u = rng();
    u = u + u - 1.0;
    t = (u < 0.0) ? b - slope * e : b + slope * e;
    if (t >= -0.71874483771719) {
      v = t / (s + s);
      if (fabs(v) <= 0.25)
        q = q_zero + 0.5 * t * t *
                     ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v + a1) * v;
      else
        q = q_zero - s * t + 0.25 * t * t + (s2 + s2) * log(1.0 + v);
      if (q > 0.0) {
        const double w = expm1(q);
        if (c * fabs(u) <= w * exp(e - 0.5 * t * t)) break;
      }
    }
  }
  x = s + 0.5 * t;
  return scale * x * x;
}

// ---- rtrun_gamma_mt beyond the plain rejection branch --------------------------
// (distributions/trun_gamma.cpp:82-104: the truncation point is at or above the
// mode).  Rare -- a sigma upper limit tighter than the posterior wants -- and
// scalar by nature, so it is kept out of line (one copy per kernel, nothing of
// it in the hot path's registers): wave-uniform callers execute it in lockstep.

#ifndef BA_RARE
#define BA_RARE __forceinline__
#endif
// dtrun_gamma(x, a, b, cut, log = true, normalize = false), trun_gamma.cpp:34-48
__device__ __forceinline__ double d_dtrun_gamma_log(double x, double a, double b, double cut) {
  if (a < 0 || b < 0 || cut < 0 || x < cut) return -__builtin_inf();
  return (a - 1) * log(x) - b * x;
}
// rexp_mt(rng, lam) (Rmath_dist.cpp:221-223)
template <class R>
__device__ __forceinline__ double d_rexp(R &r, double lam) { return (1.0 / lam) * d_exp_rand(r); }
// rtrun_exp_mt(rng, lam, lo, hi) = rpiecewise_log_linear_mt(rng, -lam, lo, hi)
// (distributions/trun_exp.cpp:38-70; finite lo < hi here)
template <class R>
__device__ __forceinline__ double d_rtrun_exp(R &r, double lam, double lo, double hi) {
  const double slope = -lam;
  if (fabs(hi - lo) < 1e-7) return lo;
  double u = 0.0;
  const double eps = 2.2250738585072014e-308;
  while (u < eps || u >= 1.0 - eps) u = r();
  double x = log(u) + slope * hi;
  double y = log(1 - u) + slope * lo;
  if (x < y) { const double t = x; x = y; y = t; }
  return (x + log1p(exp(y - x))) / slope;
}

// BoundedAdaptiveRejectionSampler (distributions/BoundedAdaptiveRejectionSampler.cpp)
// on logf(x) = (a - 1) log x - b x over [cut, inf), cut to the right of the mode.
// The hull lives across the wave: lane i holds point i (abscissa, log density,
// slope, knot, cdf), up to ARS_CAP = 64 points (the oracle's cap; the hull of a
// gamma tail is tight after a handful), so there is no per-lane array and no
// scratch memory.  Every lane of the calling wave must be active; all lanes
// consume the same numbers and return the same draw.  Searches follow
// std::lower_bound's probe sequence, which is what the reference runs over its
// knots_ / cdf_ vectors -- also where rounding has left them out of order.
enum { ARS_CAP = 64 };
__device__ __forceinline__ double ars_lane(double v, int k) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, k);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), k);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int ars_lower_bound(double v, int n, double value) {
  int first = 0, count = n;
  while (count > 0) {
    const int step = count / 2;
    if (ars_lane(v, first + step) < value) {
      first += step + 1;
      count -= step + 1;
    } else {
      count = step;
    }
  }
  return first;
}
__device__ BA_RARE double d_ars_gamma_tail(SeqRng &rng, double a, double b, double cut, int *bad) {
  const int lane = (int)(threadIdx.x & 63);
  int n = 1;
  // (lanes past n hold copies of older points: never read)
  double xs = cut, ys = d_dtrun_gamma_log(cut, a, b, cut), ds = (a - 1) / cut - b, kn = cut, cdf = 0.0;
  if (ds >= 0) { *bad = 2; return 1.0; }
  for (int level = 0; level <= 1001; ++level) {
    // update_cdf (.cpp:109-138): cdf[k] = (cdf[k - 1] + inc1[k]) - inc2[k], in order
    {
      const double y = ys - ars_lane(ys, 0), dinv = 1.0 / ds;
      const double knext = __shfl_down(kn, 1);
      const double inc1 = (lane == n - 1) ? 0 : dinv * exp(y - ds * xs + ds * knext);
      const double inc2 = dinv * exp(y - ds * xs + ds * kn);
      double last = 0;
      for (int k = 0; k < n; ++k) {
        last = (last + ars_lane(inc1, k)) - ars_lane(inc2, k);
        if (lane == k) cdf = last;
      }
    }
    // draw_safely (.cpp:150-183)
    const double u = d_runif(rng, 0.0, ars_lane(cdf, n - 1));
    int k = ars_lower_bound(cdf, n, u);
    double cand;
    if (k + 1 >= n) {   // (k == n cannot happen: u <= cdf[n - 1])
      k = n - 1;
      cand = ars_lane(kn, k) + d_rexp(rng, -1 * ars_lane(ds, k));
    } else {
      cand = d_rtrun_exp(rng, -1 * ars_lane(ds, k), ars_lane(kn, k), ars_lane(kn, k + 1));
    }
    const double target = d_dtrun_gamma_log(cand, a, b, cut);
    const double hull = ars_lane(ys, k) + ars_lane(ds, k) * (cand - ars_lane(xs, k));
    const double logu = hull - d_rexp(rng, 1.0);
    if (logu <= target) return cand;
    // add_point (.cpp:61-83): insert before the first knot >= cand
    if (n >= ARS_CAP) { *bad = 2; return 1.0; }
    const int pos = ars_lower_bound(kn, n, cand);
    {
      const double xu = __shfl_up(xs, 1), yu = __shfl_up(ys, 1), du = __shfl_up(ds, 1);
      if (lane > pos) { xs = xu; ys = yu; ds = du; }
      if (lane == pos) { xs = cand; ys = target; ds = (a - 1) / cand - b; }
    }
    ++n;
    // refresh_knots / compute_knot (:85-107)
    {
      const double x1 = __shfl_up(xs, 1), y1 = __shfl_up(ys, 1), d1 = __shfl_up(ds, 1);
      if (lane == 0 || ds == d1) {
        kn = (lane == 0) ? xs : x1;
      } else {
        double ans = (y1 - d1 * x1) - (ys - ds * xs);
        ans /= (ds - d1);
        kn = ans;
      }
    }
  }
  *bad = 2;
  return 1.0;
}

// rtg_init / rtg_slice x 5 (trun_gamma.cpp:100-104, :110-148): shape <= 1
__device__ BA_RARE double d_slice_gamma_tail(SeqRng &rng, double a, double b, double cut) {
  double x = cut;
  for (int it = 0; it < 5; ++it) {
    const double logpstar = d_dtrun_gamma_log(x, a, b, cut) - d_rexp(rng, 1.0);
    const double lo = cut;
    double hi = x;
    {  // rtg_init
      double f = d_dtrun_gamma_log(hi, a, b, cut) - logpstar;
      double fprime = ((a - 1) / hi) - b;
      int attempts = 0;
      while (f > sqrt(2.220446049250313e-16)) {
        hi -= f / fprime;
        f = d_dtrun_gamma_log(hi, a, b, cut) - logpstar;
        fprime = ((a - 1) / cut) - b;
        if (++attempts > 1000) break;
      }
    }
    x = d_runif(rng, lo, hi);
    int trials = 0;
    bool gave_up = false;
    while (d_dtrun_gamma_log(x, a, b, cut) < logpstar) {
      hi = x;
      x = d_runif(rng, lo, hi);
      if (++trials > 1000) { gave_up = true; break; }
    }
    if (gave_up) x = cut;
  }
  return x;
}

// GenericGaussianVarianceSampler::draw,
// Models/PosteriorSamplers/GenericGaussianVarianceSampler.cpp:44-63:
// sigma^2 = 1 / Gamma(shape = DF/2, rate = SS/2), truncated to
// sigma <= sigma_max when that is finite (rtrun_gamma_mt,
// distributions/trun_gamma.cpp:73-106), all three regimes: rejection from the
// plain gamma (truncation point below the mode -- the common case, inline),
// adaptive rejection (beyond the mode, shape > 1) and the slice sampler
// (shape <= 1), both out of line.  Every lane of the calling wave must be active.
// *bad: 1 = shape < .3 in the plain gamma draw, 2 = the adaptive-rejection
// sampler gave up (truncation point exactly at the mode, or 64 hull points).
template <class R>
__device__ __forceinline__ double d_draw_variance(R &rng, double DF, double SS,
                                         double sigma_max, int *bad) {
  if (sigma_max == 0.0) return 0.0;
  const double a = DF / 2, b = SS / 2;
  if (isinf(sigma_max)) return 1.0 / d_rgamma_scale(rng, a, 1.0 / b, bad);
  const double cut = 1.0 / (sigma_max * sigma_max);
  const double mode = (a - 1) / b;
  if (!(cut < mode)) {
    // (the out-of-line routines read the stream through the plain sequential
    // view, whatever view the caller uses: same numbers, same positions)
    SeqRng sr{stream_key(rng), stream_pos(rng)};
    int b2 = 0;
    const double x = (a > 1) ? d_ars_gamma_tail(sr, a, b, cut, &b2) : d_slice_gamma_tail(sr, a, b, cut);
    stream_seek(rng, sr.pos);
    // (every lane computed the same thing; tell the compiler, whose callers
    // steer wave-uniform control flow by these values)
    *bad |= __builtin_amdgcn_readfirstlane(b2);
    const unsigned long long xb = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xb);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xb >> 32));
    return 1.0 / __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  }
  double x;
  do {
    x = d_rgamma_scale(rng, a, 1.0 / b, bad);
  } while (x < cut && !*bad);
  return 1.0 / x;
}

}  // namespace boom_amd
