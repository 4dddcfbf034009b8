/* TEST INFRASTRUCTURE -- CPU oracle, not part of the shipped product.
 * See boom_oracle.h for scope, usage rules and parity status ("PINNED").
 * All file:line citations are relative to the BOOM reference tree. */
#define _GNU_SOURCE
#include "boom_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define BO_NEG_INF (-INFINITY)

/* ====================================================================== */
/*                                   RNG                                  */
/* ====================================================================== */

/* std::mt19937_64 (Matsumoto & Nishimura 2000; ISO C++ [rand.predef]). */
void bo_rng_seed_mt(bo_rng *r, uint64_t seed) {
  memset(r, 0, sizeof(*r));
  r->kind = BO_RNG_MT;
  r->mt[0] = seed;
  for (int i = 1; i < 312; ++i) {
    r->mt[i] =
        6364136223846793005ULL * (r->mt[i - 1] ^ (r->mt[i - 1] >> 62)) + (uint64_t)i;
  }
  r->mti = 312;
}

static uint64_t mt_next(bo_rng *r) {
  const uint64_t UM = 0xFFFFFFFF80000000ULL, LM = 0x7FFFFFFFULL;
  const uint64_t MATRIX_A = 0xB5026F5AA96619E9ULL;
  if (r->mti >= 312) {
    uint64_t *mt = r->mt;
    int i;
    for (i = 0; i < 312 - 156; ++i) {
      uint64_t x = (mt[i] & UM) | (mt[i + 1] & LM);
      mt[i] = mt[i + 156] ^ (x >> 1) ^ ((x & 1ULL) ? MATRIX_A : 0ULL);
    }
    for (; i < 311; ++i) {
      uint64_t x = (mt[i] & UM) | (mt[i + 1] & LM);
      mt[i] = mt[i + (156 - 312)] ^ (x >> 1) ^ ((x & 1ULL) ? MATRIX_A : 0ULL);
    }
    uint64_t x = (mt[311] & UM) | (mt[0] & LM);
    mt[311] = mt[155] ^ (x >> 1) ^ ((x & 1ULL) ? MATRIX_A : 0ULL);
    r->mti = 0;
  }
  uint64_t x = r->mt[r->mti++];
  x ^= (x >> 29) & 0x5555555555555555ULL;
  x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
  x ^= (x << 37) & 0xFFF7EEE000000000ULL;
  x ^= (x >> 43);
  return x;
}

/* Philox4x32-10, Salmon, Moraes, Dror & Shaw, "Parallel random numbers: as
 * easy as 1, 2, 3", SC'11. */
void bo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                      uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void bo_rng_seed_philox(bo_rng *r, uint64_t seed, uint32_t chain,
                        uint32_t stream, uint64_t pos) {
  memset(r, 0, sizeof(*r));
  r->kind = BO_RNG_PHILOX;
  r->seed = seed;
  r->chain = chain;
  r->stream = stream;
  r->pos = pos;
  if (stream == 2) {
    r->slot_stride = BO_STATE_SLOT_STRIDE;
    r->slot = pos / BO_STATE_SLOT_STRIDE;
  }
}

static int g_slot_limit = 0;
void bo_set_slot_limit(int uniforms) { g_slot_limit = uniforms; }
void bo_rng_slot(bo_rng *r, uint64_t index, uint64_t stride) {
  uint64_t serve = stride;
  if (g_slot_limit > 0 && (uint64_t)g_slot_limit < stride) {
    serve = (uint64_t)g_slot_limit;
    /* (the state stream: a whole Philox block, the two uniforms of norm_rand's first branch) */
    if (r->stream == 2 || r->stream == (2 | BO_SPILL_STREAM_BIT)) serve = serve >= 2 ? (serve & ~(uint64_t)1) : stride;
  }
  r->stream &= ~BO_SPILL_STREAM_BIT;
  r->pos = index * stride;
  r->limit = r->pos + serve;
  r->spill = index << BO_SPILL_SHIFT;
}

/* RNG::operator(), distributions/rng.hpp:45.  For the MT engine this is
 * libstdc++'s uniform_real_distribution<double>(0,1) = generate_canonical<
 * double,53>(mt19937_64): one 64-bit draw, converted to double (round to
 * nearest), divided by 2^64, and clamped below 1. */
double bo_unif(bo_rng *r) {
  if (r->kind == BO_RNG_MT) {
    uint64_t x = mt_next(r);
    double u = (double)x / 18446744073709551616.0;
    if (u >= 1.0) u = nextafter(1.0, 0.0);
    return u;
  } else {
    if (r->limit && r->pos >= r->limit) {   /* the slot is used up: on in its spill stream */
      r->stream |= BO_SPILL_STREAM_BIT;
      r->pos = r->spill;
      r->limit = 0;
    }
    uint64_t block = r->pos >> 1;
    uint32_t ctr[4] = {(uint32_t)block, (uint32_t)(block >> 32), r->chain,
                       r->stream};
    uint32_t key[2] = {(uint32_t)r->seed, (uint32_t)(r->seed >> 32)};
    uint32_t o[4];
    bo_philox4x32_10(ctr, key, o);
    int h = (int)(r->pos & 1);
    uint64_t x = (uint64_t)o[2 * h] | ((uint64_t)o[2 * h + 1] << 32);
    r->pos++;
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
  }
}

/* seed_rng, distributions/rng.cpp:39-47: llround(u * 2^64) (the reference
 * multiplies by double(UINT64_MAX) == 2^64), retried while <= 2.  llround
 * overflows for u >= 0.5; on x86-64 that yields INT64_MIN, i.e. 2^63 once
 * converted to the unsigned seed type -- restated explicitly so the oracle
 * does not depend on undefined behaviour. */
uint64_t bo_seed_rng(bo_rng *r) {
  uint64_t ans = 0;
  while (ans <= 2) {
    double u = bo_runif(r, 0, 1) * 18446744073709551616.0;
    if (u >= 9223372036854775808.0) {
      ans = 0x8000000000000000ULL;
    } else {
      ans = (uint64_t)llround(u);
    }
  }
  return ans;
}

/* Rmath::runif_mt, Bmath/runif.cpp:45-52 (no draw when a == b). */
double bo_runif(bo_rng *r, double a, double b) {
  if (a == b) return a;
  return a + (b - a) * bo_unif(r);
}

/* random_int_mt, distributions/random_int.cpp:26-29 */
int bo_random_int(bo_rng *r, int lo, int hi) {
  double tmp = bo_runif(r, (double)lo, (double)(hi + 1));
  return (int)floor(tmp);
}

/* shuffle, cpputil/shuffle.hpp:36-46 (in place; persistent vector) */
void bo_shuffle(bo_rng *r, int *v, int n) {
  if (n <= 0) return;
  for (int i = n - 1; i > 0; --i) {
    int other = bo_random_int(r, 0, i);
    int tmp = v[i];
    v[i] = v[other];
    v[other] = tmp;
  }
}

/* norm_rand, KINDERMAN_RAMAGE branch, Bmath/snorm.cpp:137-141, 287-340 */
#define KR_A 2.216035867166471
#define KR_C1 0.398942280401433
#define KR_C2 0.180025191068563
static double kr_g(double x) {
  return KR_C1 * exp(-x * x / 2.0) - KR_C2 * (KR_A - x);
}
double bo_norm_rand(bo_rng *r) {
  double u1, u2, u3, tt;
  u1 = bo_unif(r);
  if (u1 < 0.884070402298758) {
    u2 = bo_unif(r);
    return KR_A * (1.131131635444180 * u1 + u2 - 1);
  }
  if (u1 >= 0.973310954173898) { /* tail */
    for (;;) {
      u2 = bo_unif(r);
      u3 = bo_unif(r);
      tt = (KR_A * KR_A - 2 * log(u3));
      if (u2 * u2 < (KR_A * KR_A) / tt)
        return (u1 < 0.986655477086949) ? sqrt(tt) : -sqrt(tt);
    }
  }
  if (u1 >= 0.958720824790463) { /* region 3 */
    for (;;) {
      u2 = bo_unif(r);
      u3 = bo_unif(r);
      tt = KR_A - 0.630834801921960 * fmin(u2, u3);
      if (fmax(u2, u3) <= 0.755591531667601) return (u2 < u3) ? tt : -tt;
      if (0.034240503750111 * fabs(u2 - u3) <= kr_g(tt))
        return (u2 < u3) ? tt : -tt;
    }
  }
  if (u1 >= 0.911312780288703) { /* region 2 */
    for (;;) {
      u2 = bo_unif(r);
      u3 = bo_unif(r);
      tt = 0.479727404222441 + 1.105473661022070 * fmin(u2, u3);
      if (fmax(u2, u3) <= 0.872834976671790) return (u2 < u3) ? tt : -tt;
      if (0.049264496373128 * fabs(u2 - u3) <= kr_g(tt))
        return (u2 < u3) ? tt : -tt;
    }
  }
  for (;;) { /* region 1 */
    u2 = bo_unif(r);
    u3 = bo_unif(r);
    tt = 0.479727404222441 - 0.595507138015940 * fmin(u2, u3);
    if (tt < 0.) continue;
    if (fmax(u2, u3) <= 0.805577924423817) return (u2 < u3) ? tt : -tt;
    if (0.053377549506886 * fabs(u2 - u3) <= kr_g(tt))
      return (u2 < u3) ? tt : -tt;
  }
}

/* rnorm_mt, Bmath/rnorm.cpp:55-67 (no draw when sigma == 0) */
double bo_rnorm(bo_rng *r, double mu, double sigma) {
  if (sigma == 0.) return mu;
  if (r->kind == BO_RNG_PHILOX && r->slot_stride) {
    /* The state stream (id 2) in Philox mode: the normals of simulate_forward have a slot each
     * (draw number s owns positions [s stride, (s + 1) stride)), and -- round 6 -- TWO draws
     * share one Philox block: draws 2 j and 2 j + 1 are the Box-Muller pair of the two uniforms
     * at the start of slot 2 j,
     *     R = sqrt(-2 log(1 - u1)),  z_{2j} = R cos(2 pi u2),  z_{2j+1} = R sin(2 pi u2)
     * (G. E. P. Box and M. E. Muller (1958), Ann. Math. Statist. 29, 610-611; 1 - u1 is in
     * (0, 1]).  Exact standard normals, half the generator's work of a Kinderman-Ramage draw per
     * block and no rejection branch: the device makes 2 T of them per chain and round
     * (stream_normals.h).  What ties this stream to the reference is distributional either way
     * (tests/test_substream_bridge.py): the MT mode -- the one the compiled reference pins --
     * reads one sequential stream through norm_rand as the reference does. */
    const uint64_t s = r->slot;
    r->slot += 1;
    r->stream &= ~BO_SPILL_STREAM_BIT;
    r->limit = 0;
    r->pos = (s & ~(uint64_t)1) * r->slot_stride;
    const double u1 = bo_unif(r), u2 = bo_unif(r);
    const double R = sqrt(-2.0 * log(1.0 - u1)), th = 6.283185307179586 * u2;
    const double z = (s & 1) ? R * sin(th) : R * cos(th);
    r->pos = r->slot * r->slot_stride;
    return mu + sigma * z;
  }
  return mu + sigma * bo_norm_rand(r);
}

/* exp_rand, Bmath/sexp.cpp:58-104 */
double bo_exp_rand(bo_rng *r) {
  static const double q[] = {
      0.6931471805599453, 0.9333736875190459, 0.9888777961838675,
      0.9984959252914960, 0.9998292811061389, 0.9999833164100727,
      0.9999985691438767, 0.9999998906925558, 0.9999999924734159,
      0.9999999995283275, 0.9999999999728814, 0.9999999999985598,
      0.9999999999999289, 0.9999999999999968, 0.9999999999999999,
      1.0000000000000000};
  double a = 0., u, ustar, umin;
  int i;
  u = bo_unif(r);
  while (u <= 0.0 || u >= 1.0) u = bo_unif(r);
  for (;;) {
    u += u;
    if (u > 1.0) break;
    a += q[0];
  }
  u -= 1.;
  if (u <= q[0]) return a + u;
  i = 0;
  ustar = bo_unif(r);
  umin = ustar;
  do {
    ustar = bo_unif(r);
    if (ustar < umin) umin = ustar;
    i++;
  } while (u > q[i]);
  return a + umin * q[0];
}

/* Rmath::rgamma_mt(rng, a, scale), Bmath/rgamma.cpp:80-259, reached through
 * BOOM::rgamma_mt(rng, a, b) = Rmath::rgamma_mt(rng, a, 1/b)
 * (distributions/Rmath_dist.cpp:72-74). */
static double rgamma_scale(bo_rng *rng, double a, double scale, int *status) {
  const double sqrt32 = 5.656854;
  const double exp_m1 = 0.36787944117144232159;
  const double q1 = 0.04166669, q2 = 0.02083148, q3 = 0.00801191,
               q4 = 0.00144121, q5 = -7.388e-5, q6 = 2.4511e-4, q7 = 2.424e-4;
  const double a1 = 0.3333333, a2 = -0.250003, a3 = 0.2000062,
               a4 = -0.1662921, a5 = 0.1423657, a6 = -0.1367177,
               a7 = 0.1233795;
  double s, s2, d, q0, b, si, c;
  double e, p, q, r, t, u, v, w, x, ret_val;

  if (a < .3) {
    /* rloggamma_small_alpha, Bmath/rloggamma_small_alpha.cpp:43-79 (Liu, Martin
     * and Syring's rejection sampler for the log of a small-shape gamma) */
    const double ee = 2.718281828459045; /* exp(1) */
    const double w0 = a / (ee * (1 - a));
    const double r0 = 1.0 / (1 + w0);
    const double lambda = (1.0 / a) - 1.0;
    const double log_w = log(w0), log_lambda = log(lambda);
    for (int i = 0; i < 1000; ++i) {
      double u0 = bo_unif(rng);
      double z = u0 <= r0 ? -log(u0 / r0) : log(bo_unif(rng)) / lambda;
      double log_h = -z - exp(-z / a);
      double log_eta = (z >= 0) ? -z : log_w + log_lambda + lambda * z;
      if (log_h >= log(bo_unif(rng)) + log_eta) return exp(-z / a + log(scale));
    }
    *status = BO_ERR_UNSUPPORTED_RNG_BRANCH; /* "Max number of attempts exceeded." */
    return NAN;
  } else if (a < 1.) { /* GS */
    e = 1.0 + exp_m1 * a;
    for (;;) {
      p = e * bo_unif(rng);
      if (p >= 1.0) {
        x = -log((e - p) / a);
        if (bo_exp_rand(rng) >= (1.0 - a) * log(x)) break;
      } else {
        x = exp(log(p) / a);
        if (bo_exp_rand(rng) >= x) break;
      }
    }
    if (x > 0) return scale * x;
    return rgamma_scale(rng, a, scale, status);
  }
  /* GD, a >= 1 */
  s2 = a - 0.5;
  s = sqrt(s2);
  d = sqrt32 - s * 12.0;
  t = bo_norm_rand(rng);
  x = s + 0.5 * t;
  ret_val = x * x;
  if (t >= 0.0) return scale * ret_val;
  u = bo_unif(rng);
  if (d * u <= t * t * t) return scale * ret_val;
  r = 1.0 / a;
  q0 = ((((((q7 * r + q6) * r + q5) * r + q4) * r + q3) * r + q2) * r + q1) * r;
  if (a <= 3.686) {
    b = 0.463 + s + 0.178 * s2;
    si = 1.235;
    c = 0.195 / s - 0.079 + 0.16 * s;
  } else if (a <= 13.022) {
    b = 1.654 + 0.0076 * s2;
    si = 1.68 / s + 0.275;
    c = 0.062 / s + 0.024;
  } else {
    b = 1.77;
    si = 0.75;
    c = 0.1515 / s;
  }
  if (x > 0.0) {
    v = t / (s + s);
    if (fabs(v) <= 0.25)
      q = q0 + 0.5 * t * t *
                   ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v +
                    a1) * v;
    else
      q = q0 - s * t + 0.25 * t * t + (s2 + s2) * log1p(v);
    if (log(1.0 - u) <= q) return scale * ret_val;
  }
  for (;;) {
    e = bo_exp_rand(rng);
    u = bo_unif(rng);
    u = u + u - 1.0;
    if (u < 0.0)
      t = b - si * e;
    else
      t = b + si * e;
    if (t >= -0.71874483771719) {
      v = t / (s + s);
      if (fabs(v) <= 0.25)
        q = q0 + 0.5 * t * t *
                     ((((((a7 * v + a6) * v + a5) * v + a4) * v + a3) * v + a2) * v +
                      a1) * v;
      else
        q = q0 - s * t + 0.25 * t * t + (s2 + s2) * log(1.0 + v);
      if (q > 0.0) {
        w = expm1(q);
        if (c * fabs(u) <= w * exp(e - 0.5 * t * t)) break;
      }
    }
  }
  x = s + 0.5 * t;
  return scale * x * x;
}

double bo_rgamma(bo_rng *r, double a, double b, int *status) {
  return rgamma_scale(r, a, 1.0 / b, status);
}

/* dtrun_gamma(x, a, b, cut, logscale = true, normalize = false),
 * distributions/trun_gamma.cpp:34-48 */
static double dtrun_gamma_log(double x, double a, double b, double cut) {
  if (a < 0 || b < 0 || cut < 0 || x < cut) return BO_NEG_INF;
  return (a - 1) * log(x) - b * x;
}

/* rexp_mt(rng, lam) = exp_rand / lam (Rmath_dist.cpp:221-223, Bmath/rexp.cpp:53-59) */
static double bo_rexp(bo_rng *r, double lam) { return (1.0 / lam) * bo_exp_rand(r); }

/* rtrun_exp_mt(rng, lam, lo, hi) = rpiecewise_log_linear_mt(rng, -lam, lo, hi),
 * distributions/trun_exp.cpp:38-70 (finite limits, lo < hi on this path) */
static double bo_rtrun_exp(bo_rng *r, double lam, double lo, double hi) {
  const double slope = -lam;
  if (fabs(hi - lo) < 1e-7) return lo;
  double u = 0.0;
  const double eps = 2.2250738585072014e-308; /* numeric_limits<double>::min() */
  while (u < eps || u >= 1.0 - eps) u = bo_runif(r, 0, 1);
  double x = log(u) + slope * hi;
  double y = log(1 - u) + slope * lo;
  if (x < y) { double t = x; x = y; y = t; }
  return (x + log1p(exp(y - x))) / slope;   /* lse2, cpputil/lse.hpp:31-39 */
}

/* BoundedAdaptiveRejectionSampler for the log-concave tail of a gamma density
 * to the right of its mode (distributions/BoundedAdaptiveRejectionSampler.cpp):
 * target logf(x) = (a - 1) log x - b x, dlogf(x) = (a - 1) / x - b, support
 * [cut, inf).  Points are kept sorted; knots are where neighbouring tangents
 * cross; cdf_[k] integrates the outer hull over [knots_[k], knots_[k + 1]). */
#define BO_ARS_CAP 64
typedef struct {
  int n;
  double x[BO_ARS_CAP], y[BO_ARS_CAP], d[BO_ARS_CAP], knots[BO_ARS_CAP], cdf[BO_ARS_CAP];
  double a, b, cut;
} bo_ars;

static void ars_refresh(bo_ars *s) {
  /* refresh_knots, :85-91 + compute_knot, :93-107 */
  s->knots[0] = s->x[0];
  for (int k = 1; k < s->n; ++k) {
    double y2 = s->y[k], y1 = s->y[k - 1], d2 = s->d[k], d1 = s->d[k - 1];
    double x2 = s->x[k], x1 = s->x[k - 1];
    if (d2 == d1) {
      s->knots[k] = x1;
    } else {
      double ans = (y1 - d1 * x1) - (y2 - d2 * x2);
      ans /= (d2 - d1);
      s->knots[k] = ans;
    }
  }
}
static void ars_update_cdf(bo_ars *s) {
  /* update_cdf, :109-138 */
  const int n = s->n;
  const double y0 = s->y[0];
  double last = 0;
  for (int k = 0; k < n; ++k) {
    double d = s->d[k];
    double y = s->y[k] - y0;
    double z = s->x[k];
    double dinv = 1.0 / d;
    double inc1 = (k == n - 1) ? 0 : dinv * exp(y - d * z + d * s->knots[k + 1]);
    double inc2 = dinv * exp(y - d * z + d * s->knots[k]);
    s->cdf[k] = last + inc1 - inc2;
    last = s->cdf[k];
  }
}
/* std::lower_bound's probe sequence: the reference searches knots_ and cdf_ with
 * it, and follows it even where rounding has left those arrays out of order */
static int ars_lower_bound(const double *v, int n, double value) {
  int first = 0, count = n;
  while (count > 0) {
    int step = count / 2;
    if (v[first + step] < value) {
      first += step + 1;
      count -= step + 1;
    } else {
      count = step;
    }
  }
  return first;
}
static double ars_draw(bo_rng *r, double a, double b, double cut, int *status) {
  bo_ars s;
  s.n = 1; s.a = a; s.b = b; s.cut = cut;
  s.x[0] = cut;
  s.y[0] = dtrun_gamma_log(cut, a, b, cut);
  s.d[0] = (a - 1) / cut - b;
  s.knots[0] = cut;
  if (s.d[0] >= 0) { *status = BO_ERR_UNSUPPORTED_RNG_BRANCH; return NAN; }
  ars_update_cdf(&s);
  for (int level = 0; level <= 1001; ++level) {
    /* draw_safely, :150-183 */
    double u = bo_runif(r, 0, s.cdf[s.n - 1]);
    int k = ars_lower_bound(s.cdf, s.n, u);
    double cand;
    if (k + 1 == s.n) {
      cand = s.knots[s.n - 1] + bo_rexp(r, -1 * s.d[s.n - 1]);
    } else {
      cand = bo_rtrun_exp(r, -1 * s.d[k], s.knots[k], s.knots[k + 1]);
    }
    double target = dtrun_gamma_log(cand, a, b, cut);
    double hull = s.y[k] + s.d[k] * (cand - s.x[k]);
    double logu = hull - bo_rexp(r, 1);
    if (logu <= target) return cand;
    /* add_point, :61-83 (knots_ is what lower_bound searches) */
    if (s.n >= BO_ARS_CAP) { *status = BO_ERR_UNSUPPORTED_RNG_BRANCH; return NAN; }
    int pos = ars_lower_bound(s.knots, s.n, cand);
    for (int i = s.n; i > pos; --i) { s.x[i] = s.x[i - 1]; s.y[i] = s.y[i - 1]; s.d[i] = s.d[i - 1]; }
    s.x[pos] = cand;
    s.y[pos] = dtrun_gamma_log(cand, a, b, cut);
    s.d[pos] = (a - 1) / cand - b;
    ++s.n;
    ars_refresh(&s);
    ars_update_cdf(&s);
  }
  *status = BO_ERR_UNSUPPORTED_RNG_BRANCH;
  return NAN;
}

/* rtg_init / rtg_slice, distributions/trun_gamma.cpp:110-148 */
static double rtg_init(double x, double a, double b, double cut, double logpstar) {
  double f = dtrun_gamma_log(x, a, b, cut) - logpstar;
  double fprime = ((a - 1) / x) - b;
  int attempts = 0;
  while (f > sqrt(2.220446049250313e-16)) {
    x -= f / fprime;
    f = dtrun_gamma_log(x, a, b, cut) - logpstar;
    fprime = ((a - 1) / cut) - b;
    if (++attempts > 1000) break;
  }
  return x;
}
static double rtg_slice(bo_rng *r, double x, double a, double b, double cut) {
  double logpstar = dtrun_gamma_log(x, a, b, cut) - bo_rexp(r, 1.0);
  double lo = cut;
  double hi = rtg_init(x, a, b, cut, logpstar);
  x = bo_runif(r, lo, hi);
  int trials = 0;
  while (dtrun_gamma_log(x, a, b, cut) < logpstar) {
    hi = x;
    x = bo_runif(r, lo, hi);
    if (++trials > 1000) return cut;
  }
  return x;
}

/* rtrun_gamma_mt(rng, a, b, cut, nslice = 5), distributions/trun_gamma.cpp:73-106 */
double bo_rtrun_gamma(bo_rng *r, double a, double b, double cut, int *status) {
  double mode = (a - 1) / b;
  double x = cut;
  if (cut < mode) {
    do {
      x = bo_rgamma(r, a, b, status);
      if (*status) return NAN;
    } while (x < cut);
    return x;
  }
  if (a > 1) return ars_draw(r, a, b, cut, status);
  for (int i = 0; i < 5; ++i) x = rtg_slice(r, x, a, b, cut);
  return x;
}

/* rmulti_mt_impl, distributions/rmulti.cpp:41-78 (probsum by plain sum; the
 * reference switches to an abs-norm for n > 35, equal for non-negative prob) */
int bo_rmulti(bo_rng *r, const double *prob, int n, int *status) {
  double probsum = 0;
  for (int i = 0; i < n; ++i) probsum += prob[i];
  if (!isfinite(probsum) || probsum <= 0) {
    *status = BO_ERR_UNSUPPORTED_RNG_BRANCH;
    return 0;
  }
  double tmp = bo_runif(r, 0, probsum);
  double psum = 0;
  for (int i = 0; i < n; ++i) {
    psum += prob[i];
    if (tmp <= psum) return i;
  }
  *status = BO_ERR_UNSUPPORTED_RNG_BRANCH;
  return 0;
}

/* ====================================================================== */
/*                                 LinAlg                                 */
/* ====================================================================== */
#define IDX(i, j, n) ((size_t)(j) * (size_t)(n) + (size_t)(i))

/* Cholesky::decompose -> Eigen::LLT (LinAlg/Cholesky.cpp:33-58; the unblocked
 * left-looking kernel, Eigen/src/Cholesky/LLT.h:313-335).  Reads the lower
 * triangle of A; L is full storage with the strict upper triangle zeroed.
 * A non-positive pivot means "not positive definite" (the LDLT fallback in the
 * reference leaves pos_def_ false, so callers see the same thing). */
int bo_chol(int n, const double *A, double *L) {
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) L[IDX(i, j, n)] = (i >= j) ? A[IDX(i, j, n)] : 0.0;
  for (int k = 0; k < n; ++k) {
    double x = L[IDX(k, k, n)];
    for (int j = 0; j < k; ++j) x -= L[IDX(k, j, n)] * L[IDX(k, j, n)];
    if (!(x > 0.0)) return 0;
    x = sqrt(x);
    L[IDX(k, k, n)] = x;
    for (int i = k + 1; i < n; ++i) {
      double s = L[IDX(i, k, n)];
      for (int j = 0; j < k; ++j) s -= L[IDX(i, j, n)] * L[IDX(k, j, n)];
      L[IDX(i, k, n)] = s / x;
    }
  }
  return 1;
}

/* SpdMatrix::logdet(bool&), LinAlg/SpdMatrix.cpp:261-299 */
double bo_spd_logdet(int n, const double *A, int *ok) {
  *ok = 1;
  if (n == 0) return BO_NEG_INF;
  if (n == 1) {
    if (A[0] <= 0) {
      *ok = 0;
      return BO_NEG_INF;
    }
    return log(A[0]);
  }
  if (n == 2) {
    double det = A[0] * A[3] - A[2] * A[2];
    if (det <= 0) {
      *ok = 0;
      return BO_NEG_INF;
    }
    return log(det);
  }
  double *L = (double *)malloc(sizeof(double) * (size_t)n * n);
  if (!bo_chol(n, A, L)) {
    free(L);
    *ok = 0;
    return BO_NEG_INF;
  }
  double ans = 0.0;
  for (int i = 0; i < n; ++i) ans += log(L[IDX(i, i, n)]);
  ans *= 2;
  free(L);
  return ans;
}

static void lsolve_inplace(int n, const double *L, double *x) {
  for (int i = 0; i < n; ++i) {
    double s = x[i];
    for (int j = 0; j < i; ++j) s -= L[IDX(i, j, n)] * x[j];
    x[i] = s / L[IDX(i, i, n)];
  }
}
static void ltsolve_inplace(int n, const double *L, double *x) {
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int j = i + 1; j < n; ++j) s -= L[IDX(j, i, n)] * x[j];
    x[i] = s / L[IDX(i, i, n)];
  }
}

/* SpdMatrix::solve(Vector, bool&), LinAlg/SpdMatrix.cpp:330-341 */
int bo_spd_solve(int n, const double *A, const double *rhs, double *x) {
  double *L = (double *)malloc(sizeof(double) * (size_t)n * n);
  if (!bo_chol(n, A, L)) {
    free(L);
    for (int i = 0; i < n; ++i) x[i] = BO_NEG_INF;
    return 0;
  }
  memcpy(x, rhs, sizeof(double) * n);
  lsolve_inplace(n, L, x);
  ltsolve_inplace(n, L, x);
  free(L);
  return 1;
}

/* SpdMatrix::Mdist(x), LinAlg/SpdMatrix.cpp:363-378 */
double bo_spd_mdist(int n, const double *A, const double *x) {
  double ans = 0;
  for (int j = 0; j < n; ++j) {
    ans += x[j] * x[j] * A[IDX(j, j, n)];
    for (int i = j + 1; i < n; ++i) ans += 2 * x[j] * x[i] * A[IDX(i, j, n)];
  }
  return ans;
}

/* NeRegSuf(X, y), Models/Glm/RegressionModel.cpp:309-328 */
void bo_neregsuf(int n, int p, const double *X, const double *y, double *xtx,
                 double *xty, double *yty, double *sumy, double *xsum) {
  double q = 0, sy = 0;
  for (int i = 0; i < n; ++i) {
    q += y[i] * y[i];
    sy += y[i];
  }
  *yty = q;
  *sumy = sy;
  for (int j = 0; j < p; ++j) {
    const double *xj = X + (size_t)j * n;
    double s = 0, t = 0;
    for (int i = 0; i < n; ++i) {
      s += xj[i];
      t += xj[i] * y[i];
    }
    xsum[j] = s;
    xty[j] = t;
    for (int k = 0; k <= j; ++k) {
      const double *xk = X + (size_t)k * n;
      double d = 0;
      for (int i = 0; i < n; ++i) d += xj[i] * xk[i];
      xtx[IDX(j, k, p)] = d;
      xtx[IDX(k, j, p)] = d;
    }
  }
}

/* ====================================================================== */
/*                                  SSVS                                  */
/* ====================================================================== */
struct bo_ssvs {
  int p;
  /* sufficient statistics (NeRegSuf, RegressionModel.hpp:156-231) */
  double *xtx, *xty, yty, n, sumy, *xsum;
  /* priors */
  double *b, *ominv;     /* slab: MvnGivenScalarSigma (b, Omega^{-1}) */
  double prior_df, prior_ss; /* 2*alpha, 2*beta of the ChisqModel */
  double *pi, *logpi, *logcpi;
  int64_t max_model_size;
  double sigma_max;
  double swap_threshold;
  int max_nflips, draw_beta, draw_sigma;
  /* state */
  uint8_t *gamma;
  double *beta, sigsq;
  int *indx; /* persistent permutation, BregVsSampler.hpp:219 */
  bo_rng rng;
  /* mutable workspace members (BregVsSampler.hpp:238-240) */
  int k;          /* size of posterior_mean_ / V */
  double *pm;     /* posterior_mean_ */
  double *V;      /* unscaled_posterior_precision_ (k x k) */
  double DF, SS;
  int failure_count;
  /* CorrelationMap (CorrelationMap.cpp:41-59) */
  int cm_filled;
  int *cm_start; /* p+1 */
  int *cm_idx;
  double *cm_cor;
  /* scratch */
  int *g;
  double *w1, *w2, *w3, *M1, *M2;
  double min_margin;
  /* 1: the read-only inputs (sufficient statistics, priors, correlation map)
   * belong to another bo_ssvs (bo_ssvs_clone_shared, many-chain baseline) */
  int shared;
};

static void *xcalloc(size_t n, size_t sz) {
  void *p = calloc(n ? n : 1, sz);
  if (!p) abort();
  return p;
}

bo_ssvs *bo_ssvs_create(int p, const double *xtx, const double *xty,
                        double yty, double n, double sumy, const double *xsum,
                        const double *prior_mean, const double *ominv,
                        double prior_df, double sigma_guess, const double *pi) {
  bo_ssvs *s = (bo_ssvs *)xcalloc(1, sizeof(bo_ssvs));
  size_t pp = (size_t)p * p;
  s->p = p;
  s->xtx = (double *)xcalloc(pp, sizeof(double));
  s->xty = (double *)xcalloc(p, sizeof(double));
  s->xsum = (double *)xcalloc(p, sizeof(double));
  s->b = (double *)xcalloc(p, sizeof(double));
  s->ominv = (double *)xcalloc(pp, sizeof(double));
  s->pi = (double *)xcalloc(p, sizeof(double));
  s->logpi = (double *)xcalloc(p, sizeof(double));
  s->logcpi = (double *)xcalloc(p, sizeof(double));
  memcpy(s->xtx, xtx, pp * sizeof(double));
  memcpy(s->xty, xty, p * sizeof(double));
  memcpy(s->xsum, xsum, p * sizeof(double));
  memcpy(s->b, prior_mean, p * sizeof(double));
  memcpy(s->ominv, ominv, pp * sizeof(double));
  memcpy(s->pi, pi, p * sizeof(double));
  s->yty = yty;
  s->n = n;
  s->sumy = sumy;
  /* ChisqModel(df, sigma): alpha = df/2, beta = df*sigma^2/2
   * (ChisqModel.cpp:56-57); prior_df = 2 alpha, prior_ss = 2 beta
   * (BregVsSampler.cpp:211-214). */
  double alpha = prior_df / 2.0;
  double beta = prior_df * sigma_guess * sigma_guess / 2.0;
  s->prior_df = 2 * alpha;
  s->prior_ss = 2 * beta;
  /* VariableSelectionPrior::ensure_log_probabilities,
   * VariableSelectionPrior.cpp:310-317 */
  for (int j = 0; j < p; ++j) {
    s->logpi[j] = log(pi[j]);
    s->logcpi[j] = log(1 - pi[j]);
  }
  s->max_model_size = -1;
  s->sigma_max = INFINITY;
  s->swap_threshold = 0.8; /* CorrelationMap.hpp:37 */
  s->max_nflips = p;
  s->draw_beta = 1;
  s->draw_sigma = 1;
  s->gamma = (uint8_t *)xcalloc(p, 1);
  s->beta = (double *)xcalloc(p, sizeof(double));
  s->sigsq = 1.0;
  s->indx = (int *)xcalloc(p, sizeof(int));
  for (int j = 0; j < p; ++j) s->indx[j] = j;
  s->pm = (double *)xcalloc(p, sizeof(double));
  s->V = (double *)xcalloc(pp, sizeof(double));
  s->DF = BO_NEG_INF;
  s->SS = BO_NEG_INF;
  s->g = (int *)xcalloc(p, sizeof(int));
  s->w1 = (double *)xcalloc(p, sizeof(double));
  s->w2 = (double *)xcalloc(p, sizeof(double));
  s->w3 = (double *)xcalloc(p, sizeof(double));
  s->M1 = (double *)xcalloc(pp, sizeof(double));
  s->M2 = (double *)xcalloc(pp, sizeof(double));
  s->min_margin = INFINITY;
  bo_rng_seed_philox(&s->rng, 0, 0, 0, 0);
  return s;
}

/* The prior objects a BregVsSampler was constructed with can be changed under it (ctor #5,
 * BregVsSampler.hpp:98-106: "external copies of the pointers ... can be modified"): the
 * next draw reads the new values.  Any argument may be NULL / NaN: unchanged. */
void bo_ssvs_set_priors(bo_ssvs *s, const double *prior_mean, const double *ominv,
                        double prior_df, double sigma_guess, const double *pi) {
  const int p = s->p;
  if (prior_mean) memcpy(s->b, prior_mean, (size_t)p * sizeof(double));
  if (ominv) memcpy(s->ominv, ominv, (size_t)p * p * sizeof(double));
  if (prior_df == prior_df && sigma_guess == sigma_guess) {
    s->prior_df = 2 * (prior_df / 2.0);
    s->prior_ss = 2 * (prior_df * sigma_guess * sigma_guess / 2.0);
  }
  if (pi) {
    memcpy(s->pi, pi, (size_t)p * sizeof(double));
    for (int j = 0; j < p; ++j) {
      s->logpi[j] = log(pi[j]);
      s->logcpi[j] = log(1 - pi[j]);
    }
  }
}

void bo_ssvs_destroy(bo_ssvs *s) {
  if (!s) return;
  if (!s->shared) {
    free(s->xtx); free(s->xty); free(s->xsum); free(s->b); free(s->ominv);
    free(s->pi); free(s->logpi); free(s->logcpi);
    free(s->cm_start); free(s->cm_idx); free(s->cm_cor);
  }
  free(s->gamma); free(s->beta);
  free(s->indx); free(s->pm); free(s->V);
  free(s->g); free(s->w1); free(s->w2); free(s->w3);
  free(s->M1); free(s->M2);
  free(s);
}

void bo_ssvs_set_options(bo_ssvs *s, int64_t max_model_size,
                         double sigma_upper_limit, double swap_threshold,
                         int max_flips, int draw_beta, int draw_sigma) {
  s->max_model_size = max_model_size;
  s->sigma_max = sigma_upper_limit;
  if (swap_threshold != s->swap_threshold) s->cm_filled = 0;
  s->swap_threshold = swap_threshold;
  s->max_nflips = max_flips < 0 ? s->p : max_flips;
  s->draw_beta = draw_beta;
  s->draw_sigma = draw_sigma;
}

void bo_ssvs_set_state(bo_ssvs *s, const uint8_t *gamma, const double *beta,
                       double sigsq) {
  memcpy(s->gamma, gamma, s->p);
  if (beta) memcpy(s->beta, beta, sizeof(double) * s->p);
  s->sigsq = sigsq;
}
void bo_ssvs_get_state(const bo_ssvs *s, uint8_t *gamma, double *beta,
                       double *sigsq) {
  if (gamma) memcpy(gamma, s->gamma, s->p);
  if (beta) memcpy(beta, s->beta, sizeof(double) * s->p);
  if (sigsq) *sigsq = s->sigsq;
}
void bo_ssvs_get_perm(const bo_ssvs *s, int *perm) {
  memcpy(perm, s->indx, sizeof(int) * s->p);
}
bo_rng *bo_ssvs_rng(bo_ssvs *s) { return &s->rng; }
double bo_ssvs_min_margin(const bo_ssvs *s) { return s->min_margin; }

void bo_ssvs_set_suf(bo_ssvs *s, const double *xty, double yty, double n,
                     double sumy, const double *xsum) {
  memcpy(s->xty, xty, sizeof(double) * s->p);
  memcpy(s->xsum, xsum, sizeof(double) * s->p);
  s->yty = yty;
  s->n = n;
  s->sumy = sumy;
}

/* VariableSelectionPrior::logp, VariableSelectionPrior.cpp:271-285 */
static double spike_logp(const bo_ssvs *s, const uint8_t *g, int nvars) {
  if (s->max_model_size >= 0 && nvars > s->max_model_size) return BO_NEG_INF;
  double ans = 0;
  for (int i = 0; i < s->p; ++i) {
    ans += g[i] ? s->logpi[i] : s->logcpi[i];
    if (!isfinite(ans)) return BO_NEG_INF;
  }
  return ans;
}

static int gather_index(const bo_ssvs *s, const uint8_t *g, int *idx) {
  int k = 0;
  for (int j = 0; j < s->p; ++j)
    if (g[j]) idx[k++] = j;
  return k;
}

/* Selector::select(SpdMatrix), LinAlg/Selector.cpp:411-426 */
static void select_spd(const double *A, int p, const int *idx, int k,
                       double *out) {
  for (int c = 0; c < k; ++c)
    for (int r = 0; r < k; ++r) out[IDX(r, c, k)] = A[IDX(idx[r], idx[c], p)];
}

/* BregVsSampler::set_reg_post_params, BregVsSampler.cpp:395-484.
 * Returns ldoi (or -inf when V is not positive definite); *status is set on
 * the error exits that throw in the reference. */
static double set_reg_post_params(bo_ssvs *s, const uint8_t *g, int do_ldoi,
                                  int *status) {
  int *idx = s->g;
  int k = gather_index(s, g, idx);
  if (k == 0) return 0;
  double *prior_mean = s->w1;
  double *xty = s->w2;
  double *A = s->M1; /* unscaled prior precision, selected */
  double *S = s->M2; /* xtx, selected */
  for (int i = 0; i < k; ++i) prior_mean[i] = s->b[idx[i]];
  select_spd(s->ominv, s->p, idx, k, A);
  int ok = 1;
  double ldoi = do_ldoi ? bo_spd_logdet(k, A, &ok) : 0.0;
  select_spd(s->xtx, s->p, idx, k, S);
  for (int i = 0; i < k; ++i) xty[i] = s->xty[idx[i]];
  s->k = k;
  for (size_t i = 0; i < (size_t)k * k; ++i) s->V[i] = A[i] + S[i];
  /* posterior_mean_ = A * prior_mean + xty */
  double *rhs = s->w3;
  for (int i = 0; i < k; ++i) {
    double acc = 0;
    for (int j = 0; j < k; ++j) acc += A[IDX(i, j, k)] * prior_mean[j];
    rhs[i] = acc + xty[i];
  }
  int pd = bo_spd_solve(k, s->V, rhs, s->pm);
  if (!pd) {
    for (int i = 0; i < k; ++i) s->pm[i] = 0;
    return BO_NEG_INF;
  }
  s->DF = s->n + s->prior_df;
  s->SS = s->prior_ss;
  if (!isfinite(s->SS)) {
    *status = BO_ERR_NEGATIVE_SS;
    return ldoi;
  }
  double dot = 0;
  for (int i = 0; i < k; ++i) dot += s->pm[i] * xty[i];
  double likelihood_ss = s->yty - 2 * dot + bo_spd_mdist(k, S, s->pm);
  s->SS += likelihood_ss;
  if (!isfinite(s->SS)) {
    *status = BO_ERR_NEGATIVE_SS;
    return ldoi;
  }
  double *diff = s->w3;
  for (int i = 0; i < k; ++i) diff[i] = s->pm[i] - prior_mean[i];
  double prior_mismatch_ss = bo_spd_mdist(k, A, diff);
  s->SS += prior_mismatch_ss;
  if (s->SS < 0 || !isfinite(s->SS)) {
    *status = BO_ERR_NEGATIVE_SS;
  }
  return ldoi;
}

/* BregVsSampler::log_model_prob, BregVsSampler.cpp:216-239 */
static double log_model_prob(bo_ssvs *s, const uint8_t *g, int *status) {
  int nvars = 0;
  for (int j = 0; j < s->p; ++j) nvars += g[j] ? 1 : 0;
  if (nvars == 0) {
    double ss = s->yty + s->prior_ss;
    double df = s->n + s->prior_df;
    return spike_logp(s, g, 0) - (.5 * df - 1) * log(ss);
  }
  double ans = spike_logp(s, g, nvars);
  if (ans == BO_NEG_INF) return ans;
  double ldoi = set_reg_post_params(s, g, 1, status);
  if (*status) return BO_NEG_INF;
  if (ldoi <= BO_NEG_INF) return BO_NEG_INF;
  int ok = 1;
  ans += .5 * (ldoi - bo_spd_logdet(s->k, s->V, &ok));
  ans -= (.5 * s->DF - 1) * log(s->SS);
  return ans;
}

double bo_ssvs_log_model_prob(bo_ssvs *s, const uint8_t *gamma, int *status) {
  *status = 0;
  return log_model_prob(s, gamma, status);
}

/* BregVsSampler::mcmc_one_flip, BregVsSampler.cpp:241-250 */
static double mcmc_one_flip(bo_ssvs *s, uint8_t *g, int which, double logp_old,
                            int *status) {
  g[which] = !g[which];
  double logp_new = log_model_prob(s, g, status);
  if (*status) return logp_old;
  double u = bo_runif(&s->rng, 0, 1);
  double lu = log(u), delta = logp_new - logp_old;
  double margin = fabs(lu - delta);
  if (margin < s->min_margin) s->min_margin = margin;
  if (lu > delta) {
    g[which] = !g[which];
    return logp_old;
  }
  return logp_new;
}

/* CorrelationMap::fill, CorrelationMap.cpp:41-59, on RegSuf::centered_xtx
 * (RegressionModel.cpp:53-57). */
static void correlation_map_fill(bo_ssvs *s) {
  int p = s->p;
  free(s->cm_start); free(s->cm_idx); free(s->cm_cor);
  s->cm_start = (int *)xcalloc(p + 1, sizeof(int));
  double *sd = (double *)xcalloc(p, sizeof(double));
  double n = s->n;
  #define COV(i, j) ((s->xtx[IDX(i, j, p)] + (-n) * (s->xsum[i] / n) * (s->xsum[j] / n)) / (n - 1))
  for (int i = 0; i < p; ++i) {
    double v = COV(i, i);
    sd[i] = sqrt(v);
    if (!(sd[i] > 0.0)) sd[i] = 1.0;
  }
  int count = 0;
  for (int pass = 0; pass < 2; ++pass) {
    count = 0;
    for (int i = 0; i < p; ++i) {
      if (pass == 1) s->cm_start[i] = count;
      for (int j = 0; j < p; ++j) {
        if (j == i) continue;
        double c = fabs(COV(i, j) / (sd[i] * sd[j]));
        if (c >= s->swap_threshold) {
          if (pass == 1) {
            s->cm_idx[count] = j;
            s->cm_cor[count] = c;
          }
          ++count;
        }
      }
    }
    if (pass == 0) {
      s->cm_idx = (int *)xcalloc(count, sizeof(int));
      s->cm_cor = (double *)xcalloc(count, sizeof(double));
    }
  }
  s->cm_start[p] = count;
  #undef COV
  free(sd);
  s->cm_filled = 1;
}

/* GlmCoefs::set_inc, Models/Glm/GlmCoefs.cpp:89-94: installing a new
 * inclusion pattern zeroes the coefficients of the excluded variables. */
static void set_inc(bo_ssvs *s, const uint8_t *g) {
  memcpy(s->gamma, g, s->p);
  for (int j = 0; j < s->p; ++j)
    if (!g[j]) s->beta[j] = 0.0;
}

/* BregVsSampler::attempt_swap, BregVsSampler.cpp:277-310, with
 * CorrelationMap::propose_swap / proposal_weight (CorrelationMap.cpp:61-115)
 * and Selector::random_included_position (LinAlg/Selector.cpp:297-304). */
static void attempt_swap(bo_ssvs *s, int *status) {
  if (s->swap_threshold >= 1.0) return;
  if (!s->cm_filled) correlation_map_fill(s);
  int p = s->p;
  uint8_t *included = (uint8_t *)malloc(p);
  memcpy(included, s->gamma, p);
  int nvars = 0;
  for (int j = 0; j < p; ++j) nvars += included[j];
  if (nvars == 0 || nvars == p) {
    free(included);
    return;
  }
  int pos = bo_random_int(&s->rng, 0, nvars - 1);
  int index = -1;
  for (int j = 0, c = 0; j < p; ++j) {
    if (included[j]) {
      if (c == pos) { index = j; break; }
      ++c;
    }
  }
  /* propose_swap */
  int lo = s->cm_start[index], hi = s->cm_start[index + 1];
  if (lo == hi) { free(included); return; }
  int ncand = 0;
  int *swaps = (int *)malloc(sizeof(int) * (hi - lo));
  double *weights = (double *)malloc(sizeof(double) * (hi - lo));
  double total = 0;
  for (int i = lo; i < hi; ++i) {
    if (!included[s->cm_idx[i]]) {
      swaps[ncand] = s->cm_idx[i];
      weights[ncand] = s->cm_cor[i];
      total += weights[ncand];
      ++ncand;
    }
  }
  if (total == 0) { free(swaps); free(weights); free(included); return; }
  for (int i = 0; i < ncand; ++i) weights[i] /= total;
  int which = bo_rmulti(&s->rng, weights, ncand, status);
  double forward_w = weights[which];
  int candidate = swaps[which];
  free(swaps); free(weights);
  if (*status) { free(included); return; }

  double original_logp = log_model_prob(s, included, status);
  included[index] = 0;
  included[candidate] = 1;
  /* reverse weight: proposal_weight(included, candidate, index), i.e. the
   * table of `candidate`, looking for `index` among its excluded correlates */
  double reverse_w;
  {
    int l2 = s->cm_start[candidate], h2 = s->cm_start[candidate + 1];
    double ans = BO_NEG_INF, tot = 0;
    for (int i = l2; i < h2; ++i) {
      if (!included[s->cm_idx[i]]) {
        if (s->cm_idx[i] == index) ans = s->cm_cor[i];
        tot += s->cm_cor[i];
      }
    }
    reverse_w = (tot == 0) ? 0 : ans / tot;
  }
  double log_num = log_model_prob(s, included, status) - log(forward_w);
  double log_den = original_logp - log(reverse_w);
  double logu = log(bo_runif(&s->rng, 0, 1));
  if (logu < log_num - log_den) set_inc(s, included);
  free(included);
}

/* BregVsSampler::draw_model_indicators, BregVsSampler.cpp:353-378 */
static void draw_model_indicators(bo_ssvs *s, int *status) {
  int p = s->p;
  uint8_t *g = (uint8_t *)malloc(p);
  memcpy(g, s->gamma, p);
  bo_shuffle(&s->rng, s->indx, p);
  double logp = log_model_prob(s, g, status);
  if (!*status && !isfinite(logp)) {
    /* VariableSelectionPrior::make_valid, VariableSelectionPrior.cpp:287-300 */
    for (int i = 0; i < p; ++i) {
      if (s->pi[i] <= 0.0 && g[i]) g[i] = 0;
      if (s->pi[i] >= 1.0 && !g[i]) g[i] = 1;
    }
    logp = log_model_prob(s, g, status);
  }
  if (!*status && !isfinite(logp)) *status = BO_ERR_ILLEGAL_START;
  if (*status) { free(g); return; }
  int n = s->max_nflips < p ? s->max_nflips : p;
  for (int i = 0; i < n && !*status; ++i) {
    logp = mcmc_one_flip(s, g, s->indx[i], logp, status);
  }
  set_inc(s, g);
  free(g);
  if (*status) return;
  attempt_swap(s, status);
}

/* GenericGaussianVarianceSampler::draw,
 * Models/PosteriorSamplers/GenericGaussianVarianceSampler.cpp:44-63 */
static double variance_draw(bo_rng *rng, double prior_df, double prior_ss,
                            double sigma_max, double data_df, double data_ss,
                            int *status) {
  double DF = data_df + prior_df;
  double SS = data_ss + prior_ss;
  if (sigma_max == 0.0) return 0.0;
  if (sigma_max == INFINITY) return 1.0 / bo_rgamma(rng, DF / 2, SS / 2, status);
  return 1.0 / bo_rtrun_gamma(rng, DF / 2, SS / 2,
                              1.0 / (sigma_max * sigma_max), status);
}

static int ssvs_draw(bo_ssvs *s);

/* BregVsSampler::draw_sigma, BregVsSampler.cpp:313-324 */
static void draw_sigma(bo_ssvs *s, int nvars, int *status) {
  double df, ss;
  if (nvars == 0) {
    ss = s->yty;
    df = s->n;
  } else {
    df = s->DF - s->prior_df;
    ss = s->SS - s->prior_ss;
  }
  s->sigsq = variance_draw(&s->rng, s->prior_df, s->prior_ss, s->sigma_max, df,
                           ss, status);
}

/* BregVsSampler::draw_beta, BregVsSampler.cpp:326-351, with
 * rmvn_precision_upper_cholesky_mt (distributions/mvn.cpp:114-122) */
static void draw_beta(bo_ssvs *s, int nvars, int *status) {
  if (nvars == 0) return;
  int k = s->k;
  double *P = s->M1, *L = s->M2;
  for (size_t i = 0; i < (size_t)k * k; ++i) P[i] = s->V[i] / s->sigsq;
  if (bo_chol(k, P, L)) {
    double *z = s->w1;
    for (int i = 0; i < k; ++i) z[i] = bo_rnorm(&s->rng, 0, 1);
    ltsolve_inplace(k, L, z); /* Usolve_inplace(L^T, z) */
    for (int i = 0; i < k; ++i) s->pm[i] = z[i] + s->pm[i];
    /* set_included_coefficients */
    memset(s->beta, 0, sizeof(double) * s->p);
    for (int j = 0, c = 0; j < s->p; ++j)
      if (s->gamma[j]) s->beta[j] = s->pm[c++];
    s->failure_count = 0;
  } else {
    if (++s->failure_count > 10) {
      *status = BO_ERR_NOT_PD;
      return;
    }
    *status = ssvs_draw(s);
  }
}

/* BregVsSampler::draw, BregVsSampler.cpp:252-261 */
static int ssvs_draw(bo_ssvs *s) {
  int status = 0;
  if (s->max_nflips > 0) draw_model_indicators(s, &status);
  if (status) return status;
  int nvars = 0;
  for (int j = 0; j < s->p; ++j) nvars += s->gamma[j];
  if (s->draw_beta || s->draw_sigma) {
    set_reg_post_params(s, s->gamma, 0, &status);
    if (status) return status;
  }
  if (s->draw_sigma) draw_sigma(s, nvars, &status);
  if (status) return status;
  if (s->draw_beta) draw_beta(s, nvars, &status);
  return status;
}

int bo_ssvs_draw(bo_ssvs *s) { return ssvs_draw(s); }

/* BregVsSampler::logpri, BregVsSampler.cpp:380-393, at the sampler's current
 * state: log p(gamma) + log p(sigma^2) + log N(beta_g | b_g, sigma^2 Omega_g).
 *   p(sigma^2): GenericGaussianVarianceSampler::log_prior
 *     (GenericGaussianVarianceSampler.cpp:81-91) = Gamma(df/2, ss/2) log
 *     density of 1/sigma^2 plus the Jacobian -2 log sigma^2;
 *   dmvn(y, mu, siginv, log) (distributions/mvn.cpp): -k/2 log 2pi +
 *     1/2 log|siginv| - 1/2 Mdist(y - mu; siginv), siginv = Omega^{-1}_g / sigma^2
 *     (MvnGivenScalarSigma.cpp:74-77). */
double bo_ssvs_logpri(bo_ssvs *s) {
  const int p = s->p;
  const int k = gather_index(s, s->gamma, s->g);
  double ans = spike_logp(s, s->gamma, k);
  if (!(ans > BO_NEG_INF)) return ans;
  const double a = 0.5 * s->prior_df, b = 0.5 * s->prior_ss, x = 1.0 / s->sigsq;
  ans += a * log(b) - lgamma(a) + (a - 1.0) * log(x) - b * x - 2.0 * log(s->sigsq);
  if (k > 0) {
    double *P = (double *)xcalloc((size_t)k * k, sizeof(double));
    double *d = (double *)xcalloc((size_t)k, sizeof(double));
    select_spd(s->ominv, p, s->g, k, P);
    for (int i = 0; i < k * k; ++i) P[i] /= s->sigsq;
    for (int i = 0; i < k; ++i) d[i] = s->beta[s->g[i]] - s->b[s->g[i]];
    int ok = 1;
    const double ld = bo_spd_logdet(k, P, &ok);
    ans += -0.5 * k * log(2.0 * M_PI) + 0.5 * ld - 0.5 * bo_spd_mdist(k, P, d);
    free(P);
    free(d);
  }
  return ans;
}

/* ---- convenience-ctor prior assembly ---------------------------------- */
/* BregVsSampler ctor #1, BregVsSampler.cpp:37-44, 48-85 */
void bo_breg_prior_ctor1(int p, const double *xtx, double yty, double n,
                         double sumy, double prior_nobs, double expected_rsq,
                         double expected_model_size,
                         int first_term_is_intercept, double *b, double *ominv,
                         double *pi, double *prior_df, double *sigma_guess) {
  double ybar = sumy / n;
  /* RegSuf::SST = yty - n * ybar^2 (RegressionModel.cpp) */
  double sst = yty - n * ybar * ybar;
  double sample_variance = sst / (n - 1);
  *sigma_guess = sqrt(sample_variance * (1 - expected_rsq));
  *prior_df = prior_nobs;
  for (int j = 0; j < p; ++j) b[j] = 0.0;
  if (first_term_is_intercept) b[0] = ybar;
  for (size_t i = 0; i < (size_t)p * p; ++i) ominv[i] = xtx[i] * (prior_nobs / n);
  double prob = expected_model_size / p;
  if (prob > 1) prob = 1.0;
  for (int j = 0; j < p; ++j) pi[j] = prob;
  if (first_term_is_intercept) pi[0] = 1.0;
}

/* BregVsSampler ctor #2, BregVsSampler.cpp:87-142 */
void bo_breg_prior_ctor2(int p, const double *xtx, double n, double sumy,
                         double prior_sigma_nobs, double prior_sigma_guess,
                         double prior_beta_nobs, double diagonal_shrinkage,
                         double prior_inclusion_probability,
                         int force_intercept, double *b, double *ominv,
                         double *pi, double *prior_df, double *sigma_guess) {
  *prior_df = prior_sigma_nobs;
  *sigma_guess = prior_sigma_guess;
  for (int j = 0; j < p; ++j) b[j] = 0.0;
  b[0] = sumy / n;
  for (size_t i = 0; i < (size_t)p * p; ++i)
    ominv[i] = xtx[i] * (prior_beta_nobs / n);
  double alpha = diagonal_shrinkage;
  if (alpha < 1.0) {
    for (int j = 0; j < p; ++j) {
      double d = ominv[IDX(j, j, p)];
      ominv[IDX(j, j, p)] = d + d * (alpha / (1 - alpha));
    }
    for (size_t i = 0; i < (size_t)p * p; ++i) ominv[i] *= (1 - alpha);
  } else {
    for (int j = 0; j < p; ++j)
      for (int i = 0; i < p; ++i)
        if (i != j) ominv[IDX(i, j, p)] = 0.0;
  }
  for (int j = 0; j < p; ++j) pi[j] = prior_inclusion_probability;
  if (force_intercept) pi[0] = 1.0;
}

/* ====================================================================== */
/*     AdaptiveSpikeSlabRegressionSampler (birth / death moves, rates)     */
/* ====================================================================== */
/* Models/Glm/PosteriorSamplers/AdaptiveSpikeSlabRegressionSampler.cpp:62-225:
 * what lm.spike uses for more than 100 predictors.  Works on a bo_ssvs (same
 * sufficient statistics, priors, state and stream), plus the adaptive birth /
 * death rates. */
struct bo_adaptive {
  bo_ssvs *s;
  double *birth, *death;     /* birth_rates_, death_rates_ (all 1 at the start) */
  int max_flips;             /* max_flips_ = 100 */
  int allow_model_selection;
  uint64_t iteration;        /* iteration_count_ */
  double step, target;       /* step_size_ = .001, target_acceptance_rate_ = .345 */
  double cur_logp;           /* current_log_model_prob_ */
  /* set_posterior_moments' members */
  double ldoi, DF, SS;
  int k;
  /* smallest |log u - log MH ratio| and smallest relative distance of an
   * rmulti uniform from a boundary of its cumulative sums (decision margins) */
  double min_margin, min_multi_margin;
};

bo_adaptive *bo_adaptive_create(bo_ssvs *s) {
  bo_adaptive *a = (bo_adaptive *)xcalloc(1, sizeof(bo_adaptive));
  a->s = s;
  a->birth = (double *)xcalloc(s->p, sizeof(double));
  a->death = (double *)xcalloc(s->p, sizeof(double));
  for (int j = 0; j < s->p; ++j) a->birth[j] = a->death[j] = 1.0;
  a->max_flips = 100;
  a->allow_model_selection = 1;
  a->step = .001;
  a->target = .345;
  a->cur_logp = BO_NEG_INF;
  a->min_margin = INFINITY;
  a->min_multi_margin = INFINITY;
  return a;
}
void bo_adaptive_destroy(bo_adaptive *a) {
  if (!a) return;
  free(a->birth);
  free(a->death);
  free(a);
}
void bo_adaptive_set_options(bo_adaptive *a, int max_flips, double step,
                             double target) {
  if (max_flips >= 0) a->max_flips = max_flips;
  if (step > 0) a->step = step;
  if (target > 0) a->target = target;
}
void bo_adaptive_get_rates(const bo_adaptive *a, double *birth, double *death) {
  memcpy(birth, a->birth, sizeof(double) * a->s->p);
  memcpy(death, a->death, sizeof(double) * a->s->p);
}
double bo_adaptive_min_margin(const bo_adaptive *a) { return a->min_margin; }
double bo_adaptive_min_multi_margin(const bo_adaptive *a) { return a->min_multi_margin; }

/* set_posterior_moments, .cpp:129-151 (posterior mean into s->pm, unscaled
 * posterior precision into s->V) */
static void adaptive_set_posterior_moments(bo_adaptive *a, const uint8_t *g) {
  bo_ssvs *s = a->s;
  int *idx = s->g;
  int k = gather_index(s, g, idx);
  a->k = s->k = k;
  double *Om = s->M1, *V = s->V, *mu = s->w1, *rhs = s->w2;
  select_spd(s->ominv, s->p, idx, k, Om);
  int ok;
  a->ldoi = bo_spd_logdet(k, Om, &ok);
  for (int i = 0; i < k; ++i) mu[i] = s->b[idx[i]];
  select_spd(s->xtx, s->p, idx, k, V);
  for (size_t i = 0; i < (size_t)k * k; ++i) V[i] = Om[i] + V[i];
  /* xty_g + Omega^{-1}_g mu_g */
  for (int i = 0; i < k; ++i) {
    double t = 0;
    for (int j = 0; j < k; ++j) t += Om[IDX(i, j, k)] * mu[j];
    rhs[i] = s->xty[idx[i]] + t;
  }
  bo_spd_solve(k, V, rhs, s->pm);
  a->DF = s->prior_df + s->n;
  /* relative_sse(GlmCoefs(posterior_mean, inc)): yty + b'XtX_g b - 2 b'xty_g
   * (RegressionModel.cpp:59-70) */
  double sse = s->yty;
  if (k > 0) {
    double quad = 0, lin = 0;
    for (int i = 0; i < k; ++i) {
      double t = 0;
      for (int j = 0; j < k; ++j) t += s->xtx[IDX(idx[i], idx[j], s->p)] * s->pm[j];
      quad += s->pm[i] * t;
      lin += s->pm[i] * s->xty[idx[i]];
    }
    sse += quad - 2 * lin;
  }
  double *d = s->w3;
  for (int i = 0; i < k; ++i) d[i] = s->pm[i] - mu[i];
  a->SS = s->prior_ss + sse + bo_spd_mdist(k, Om, d);
}

/* log_model_prob, .cpp:87-126 (the memo map returns what a recomputation
 * returns: the value is a function of gamma alone) */
static double adaptive_log_model_prob(bo_adaptive *a, const uint8_t *g) {
  bo_ssvs *s = a->s;
  int nvars = 0;
  for (int j = 0; j < s->p; ++j) nvars += g[j];
  if (nvars == 0) {
    double ss = s->yty + s->prior_ss;
    double df = s->n + s->prior_df;
    return spike_logp(s, g, nvars) - (.5 * df - 1) * log(ss);
  }
  double ans = spike_logp(s, g, nvars);
  if (ans == BO_NEG_INF) return ans;
  adaptive_set_posterior_moments(a, g);
  if (a->ldoi <= BO_NEG_INF) return BO_NEG_INF;
  int ok;
  double ldv = bo_spd_logdet(a->k, s->V, &ok);
  ans += .5 * (a->ldoi - ldv) - (.5 * a->DF - 1) * log(a->SS);
  return ans;
}

/* rmulti_mt on a weight vector (distributions/rmulti.cpp:41-78), recording
 * how close the uniform came to a boundary */
static int adaptive_rmulti(bo_adaptive *a, const double *w, int n, int *status) {
  double probsum = 0;
  for (int i = 0; i < n; ++i) probsum += w[i];
  if (!isfinite(probsum) || probsum <= 0) {
    *status = BO_ERR_UNSUPPORTED_RNG_BRANCH;
    return 0;
  }
  double tmp = bo_runif(&a->s->rng, 0, probsum);
  double psum = 0;
  int ans = -1;
  for (int i = 0; i < n; ++i) {
    psum += w[i];
    double m = fabs(tmp - psum) / probsum;
    if (m < a->min_multi_margin) a->min_multi_margin = m;
    if (ans < 0 && tmp <= psum) ans = i;
  }
  if (ans < 0) {
    *status = BO_ERR_UNSUPPORTED_RNG_BRANCH;
    return 0;
  }
  return ans;
}

static void adaptive_adjust(bo_adaptive *a, double *rate, double mh_alpha) {
  if (mh_alpha > 1.0) mh_alpha = 1.0;
  double adjustment = a->step / ((1.0 + (double)a->iteration) / a->s->p);
  adjustment *= (mh_alpha - a->target);
  *rate *= exp(adjustment);
}

/* birth_move (.cpp:165-192) when birth != 0, death_move (.cpp:202-226) otherwise */
static void adaptive_move(bo_adaptive *a, uint8_t *g, int birth, int *status) {
  bo_ssvs *s = a->s;
  const int p = s->p;
  const double *from = birth ? a->birth : a->death;   /* rates of the candidates */
  const double *back = birth ? a->death : a->birth;   /* rates of the reverse move */
  int *cand = s->g;   /* candidate variables, ascending */
  double *w = s->w1;
  int m = 0;
  for (int j = 0; j < p; ++j)
    if ((g[j] != 0) != (birth != 0)) { cand[m] = j; w[m] = from[j]; ++m; }
  if (m == 0) return;
  int which = adaptive_rmulti(a, w, m, status);
  if (*status) return;
  const int j = cand[which];
  double wsum = 0;
  for (int i = 0; i < m; ++i) wsum += w[i];
  const double wj = w[which];
  g[j] = (uint8_t)(birth ? 1 : 0);
  const double cand_logp = adaptive_log_model_prob(a, g);
  const double num = cand_logp - log(wj / wsum);
  /* the reverse move's candidates: the new model's included (birth) or excluded
   * (death) variables, summed in index order (Selector::sparse_sum) */
  double bsum = 0;
  for (int i = 0; i < p; ++i)
    if ((g[i] != 0) == (birth != 0)) bsum += back[i];
  const double den = a->cur_logp - log(back[j] / bsum);
  const double ratio = num - den;
  const double logu = log(bo_runif(&s->rng, 0, 1));
  const double margin = fabs(logu - ratio);
  if (margin < a->min_margin) a->min_margin = margin;
  if (logu < ratio) {
    a->cur_logp = cand_logp;
    adaptive_adjust(a, birth ? &a->birth[j] : &a->death[j], exp(ratio));
  } else {
    g[j] = (uint8_t)(birth ? 0 : 1);
  }
}

/* draw(), .cpp:62-85 */
int bo_adaptive_draw(bo_adaptive *a) {
  bo_ssvs *s = a->s;
  int status = 0;
  uint8_t *g = (uint8_t *)malloc(s->p);
  memcpy(g, s->gamma, s->p);
  if (a->allow_model_selection) {
    int flips = a->max_flips < s->p ? a->max_flips : s->p;
    a->cur_logp = adaptive_log_model_prob(a, g);
    for (int i = 0; i < flips && !status; ++i) {
      double u = bo_runif(&s->rng, 0, 1);
      adaptive_move(a, g, u < .5, &status);
    }
    /* coef().set_inc: excluded coefficients are zeroed (GlmCoefs.cpp:89-94) */
    memcpy(s->gamma, g, s->p);
    for (int j = 0; j < s->p; ++j)
      if (!g[j]) s->beta[j] = 0.0;
  }
  if (!status) {
    adaptive_set_posterior_moments(a, g);
    s->DF = a->DF;
    s->SS = a->SS;
    /* draw_residual_variance, .cpp:158-163 */
    s->sigsq = variance_draw(&s->rng, s->prior_df, s->prior_ss, s->sigma_max,
                             a->DF - s->prior_df, a->SS - s->prior_ss, &status);
  }
  if (!status && a->k > 0) {
    /* draw_coefficients: rmvn_ivar_mt(mean, V / sigsq) (mvn.cpp:103-122) */
    int k = a->k;
    double *P = s->M1, *L = s->M2;
    for (size_t i = 0; i < (size_t)k * k; ++i) P[i] = s->V[i] / s->sigsq;
    if (!bo_chol(k, P, L)) {
      status = BO_ERR_NOT_PD;
    } else {
      double *z = s->w1;
      for (int i = 0; i < k; ++i) z[i] = bo_rnorm(&s->rng, 0, 1);
      ltsolve_inplace(k, L, z);
      memset(s->beta, 0, sizeof(double) * s->p);
      for (int j = 0, c = 0; j < s->p; ++j)
        if (g[j]) { s->beta[j] = z[c] + s->pm[c]; ++c; }
    }
  } else if (!status) {
    memset(s->beta, 0, sizeof(double) * s->p);
  }
  free(g);
  ++a->iteration;
  return status;
}

/* StateSpaceRegressionModel::simulate_forecast with a local level
 * (StateSpaceRegressionModel.cpp:214-219, :256-278; advance_to_timestamp
 * StateSpaceModelBase.cpp:455-467; simulate_next_state :439-452): per step the
 * state error, then the observation noise, plus the regression prediction
 * x_i'beta (GlmCoefs::predict: a dense dot with Beta()). */
void bo_ss_simulate_forecast(bo_rng *rng, int horizon, int p, const double *newX,
                             const double *beta, double sigsq_obs,
                             double sigsq_level, double final_state, double *out) {
  double state = final_state;
  const double sd_level = sqrt(sigsq_level), sd_obs = sqrt(sigsq_obs);
  for (int i = 0; i < horizon; ++i) {
    state = state + bo_rnorm(rng, 0, sd_level);
    double ans = bo_rnorm(rng, state, sd_obs);
    double pred = 0;
    for (int j = 0; j < p; ++j) pred += newX[IDX(i, j, horizon)] * beta[j];
    out[i] = ans + pred;
  }
}

/* ---- many chains (cpu_baseline leg) ------------------------------------ */
typedef struct {
  int p;
  const double *xtx, *xty, *xsum, *prior_mean, *ominv, *pi;
  double yty, n, sumy, prior_df, sigma_guess;
  int64_t max_model_size;
  double sigma_upper_limit, swap_threshold;
  int max_flips;
  uint64_t seed;
  int chain_lo, chain_hi, nsweeps;
  uint8_t *gamma;
  double *beta, *sigsq;
  int status;
  const bo_ssvs *shared_template;
} chain_job;

/* A second sampler on the SAME read-only inputs as `t` (pointers shared: X'X,
 * Omega^{-1}, the prior vectors and the already filled correlation map), with
 * its own state, stream and workspace -- what running many chains of one
 * model on a multi-core host looks like.  `t` must outlive the clone and its
 * correlation map must be filled and left alone. */
static bo_ssvs *bo_ssvs_clone_shared(const bo_ssvs *t) {
  bo_ssvs *s = (bo_ssvs *)xcalloc(1, sizeof(bo_ssvs));
  const int p = t->p;
  const size_t pp = (size_t)p * p;
  *s = *t;
  s->shared = 1;
  s->gamma = (uint8_t *)xcalloc(p, 1);
  s->beta = (double *)xcalloc(p, sizeof(double));
  s->sigsq = 1.0;
  s->indx = (int *)xcalloc(p, sizeof(int));
  for (int j = 0; j < p; ++j) s->indx[j] = j;
  s->pm = (double *)xcalloc(p, sizeof(double));
  s->V = (double *)xcalloc(pp, sizeof(double));
  s->DF = BO_NEG_INF;
  s->SS = BO_NEG_INF;
  s->k = 0;
  s->failure_count = 0;
  s->g = (int *)xcalloc(p, sizeof(int));
  s->w1 = (double *)xcalloc(p, sizeof(double));
  s->w2 = (double *)xcalloc(p, sizeof(double));
  s->w3 = (double *)xcalloc(p, sizeof(double));
  s->M1 = (double *)xcalloc(pp, sizeof(double));
  s->M2 = (double *)xcalloc(pp, sizeof(double));
  s->min_margin = INFINITY;
  return s;
}

static void *chain_worker(void *arg) {
  chain_job *j = (chain_job *)arg;
  /* one sampler object per thread, re-pointed at chain after chain: the
   * workspace stays hot and nothing read-only is copied */
  bo_ssvs *s = bo_ssvs_clone_shared(j->shared_template);
  for (int c = j->chain_lo; c < j->chain_hi; ++c) {
    bo_ssvs_set_state(s, j->gamma + (size_t)c * j->p, j->beta + (size_t)c * j->p,
                      j->sigsq[c]);
    for (int i = 0; i < j->p; ++i) s->indx[i] = i;
    s->DF = BO_NEG_INF;
    s->SS = BO_NEG_INF;
    s->failure_count = 0;
    bo_rng_seed_philox(&s->rng, j->seed, (uint32_t)c, 0, 0);
    for (int i = 0; i < j->nsweeps; ++i) {
      int st = ssvs_draw(s);
      if (st) { j->status = st; break; }
    }
    bo_ssvs_get_state(s, j->gamma + (size_t)c * j->p, j->beta + (size_t)c * j->p,
                      &j->sigsq[c]);
  }
  bo_ssvs_destroy(s);
  return NULL;
}

int bo_ssvs_run_chains(int p, const double *xtx, const double *xty, double yty,
                       double n, double sumy, const double *xsum,
                       const double *prior_mean, const double *ominv,
                       double prior_df, double sigma_guess, const double *pi,
                       int64_t max_model_size, double sigma_upper_limit,
                       double swap_threshold, int max_flips, uint64_t seed,
                       int chains, int nsweeps, int nthreads, uint8_t *gamma,
                       double *beta, double *sigsq) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > chains) nthreads = chains;
  /* the read-only inputs and the correlation map exist once */
  bo_ssvs *tmpl = bo_ssvs_create(p, xtx, xty, yty, n, sumy, xsum, prior_mean, ominv,
                                 prior_df, sigma_guess, pi);
  bo_ssvs_set_options(tmpl, max_model_size, sigma_upper_limit, swap_threshold,
                      max_flips, 1, 1);
  if (swap_threshold < 1.0) correlation_map_fill(tmpl);
  pthread_t *th = (pthread_t *)xcalloc(nthreads, sizeof(pthread_t));
  chain_job *jobs = (chain_job *)xcalloc(nthreads, sizeof(chain_job));
  for (int t = 0; t < nthreads; ++t) {
    chain_job *j = &jobs[t];
    j->p = p; j->xtx = xtx; j->xty = xty; j->xsum = xsum;
    j->prior_mean = prior_mean; j->ominv = ominv; j->pi = pi;
    j->yty = yty; j->n = n; j->sumy = sumy; j->prior_df = prior_df;
    j->sigma_guess = sigma_guess; j->max_model_size = max_model_size;
    j->sigma_upper_limit = sigma_upper_limit; j->swap_threshold = swap_threshold;
    j->max_flips = max_flips; j->seed = seed; j->nsweeps = nsweeps;
    j->chain_lo = (int)((int64_t)chains * t / nthreads);
    j->chain_hi = (int)((int64_t)chains * (t + 1) / nthreads);
    j->gamma = gamma; j->beta = beta; j->sigsq = sigsq;
    j->shared_template = tmpl;
    pthread_create(&th[t], NULL, chain_worker, j);
  }
  int status = 0;
  for (int t = 0; t < nthreads; ++t) {
    pthread_join(th[t], NULL);
    if (jobs[t].status) status = jobs[t].status;
  }
  free(th);
  free(jobs);
  bo_ssvs_destroy(tmpl);
  return status;
}

/* ====================================================================== */
/*                 SpikeSlabSampler (sigma^2-conditional SSVS)            */
/* ====================================================================== */
struct bo_sss {
  int p, slab_kind;
  double *xtx, *xty, *mu, *prec, *pi, *logpi, *logcpi;
  int64_t max_model_size;
  int max_flips;
  int shuffle_kind; /* 0: SpikeSlabSampler's Fisher-Yates; 1: BinomialLogitSpikeSlabSampler's */
  uint8_t *gamma;
  double *beta;
  bo_rng rng;
  int *g, *indx;
  double *w1, *w2, *w3, *M1, *M2, *M3;
};

bo_sss *bo_sss_create(int p, const double *xtx, const double *xty, int slab_kind,
                      const double *mu, const double *prec, const double *pi) {
  bo_sss *s = (bo_sss *)xcalloc(1, sizeof(bo_sss));
  size_t pp = (size_t)p * p;
  s->p = p;
  s->slab_kind = slab_kind;
  s->xtx = (double *)xcalloc(pp, sizeof(double));
  s->xty = (double *)xcalloc(p, sizeof(double));
  s->mu = (double *)xcalloc(p, sizeof(double));
  s->prec = (double *)xcalloc(pp, sizeof(double));
  s->pi = (double *)xcalloc(p, sizeof(double));
  s->logpi = (double *)xcalloc(p, sizeof(double));
  s->logcpi = (double *)xcalloc(p, sizeof(double));
  memcpy(s->xtx, xtx, pp * sizeof(double));
  memcpy(s->xty, xty, p * sizeof(double));
  memcpy(s->mu, mu, p * sizeof(double));
  memcpy(s->prec, prec, pp * sizeof(double));
  memcpy(s->pi, pi, p * sizeof(double));
  for (int j = 0; j < p; ++j) {
    s->logpi[j] = log(pi[j]);
    s->logcpi[j] = log(1 - pi[j]);
  }
  s->max_model_size = -1;
  s->max_flips = -1;
  s->gamma = (uint8_t *)xcalloc(p, 1);
  s->beta = (double *)xcalloc(p, sizeof(double));
  s->g = (int *)xcalloc(p, sizeof(int));
  s->indx = (int *)xcalloc(p, sizeof(int));
  s->w1 = (double *)xcalloc(p, sizeof(double));
  s->w2 = (double *)xcalloc(p, sizeof(double));
  s->w3 = (double *)xcalloc(p, sizeof(double));
  s->M1 = (double *)xcalloc(pp, sizeof(double));
  s->M2 = (double *)xcalloc(pp, sizeof(double));
  s->M3 = (double *)xcalloc(pp, sizeof(double));
  bo_rng_seed_philox(&s->rng, 0, 0, 3, 0);
  return s;
}
void bo_sss_destroy(bo_sss *s) {
  if (!s) return;
  free(s->xtx); free(s->xty); free(s->mu); free(s->prec); free(s->pi);
  free(s->logpi); free(s->logcpi); free(s->gamma); free(s->beta); free(s->g);
  free(s->indx); free(s->w1); free(s->w2); free(s->w3); free(s->M1); free(s->M2);
  free(s->M3);
  free(s);
}
void bo_sss_set_options(bo_sss *s, int64_t max_model_size, int max_flips) {
  s->max_model_size = max_model_size;
  s->max_flips = max_flips;
}
void bo_sss_set_state(bo_sss *s, const uint8_t *gamma, const double *beta) {
  memcpy(s->gamma, gamma, s->p);
  if (beta) memcpy(s->beta, beta, sizeof(double) * s->p);
}
void bo_sss_get_state(const bo_sss *s, uint8_t *gamma, double *beta) {
  if (gamma) memcpy(gamma, s->gamma, s->p);
  if (beta) memcpy(beta, s->beta, sizeof(double) * s->p);
}
bo_rng *bo_sss_rng(bo_sss *s) { return &s->rng; }
void bo_sss_set_shuffle_kind(bo_sss *s, int kind) { s->shuffle_kind = kind; }

static double sss_spike_logp(const bo_sss *s, const uint8_t *g, int nvars) {
  if (s->max_model_size >= 0 && nvars > s->max_model_size) return BO_NEG_INF;
  double ans = 0;
  for (int i = 0; i < s->p; ++i) {
    ans += g[i] ? s->logpi[i] : s->logcpi[i];
    if (!isfinite(ans)) return BO_NEG_INF;
  }
  return ans;
}

/* slab_prior_->siginv() restricted to the model: prec or prec / sigsq */
static void sss_select_precision(const bo_sss *s, const int *idx, int k,
                                 double sigsq, double *out) {
  for (int c = 0; c < k; ++c)
    for (int r = 0; r < k; ++r) {
      double v = s->prec[IDX(idx[r], idx[c], s->p)];
      out[IDX(r, c, k)] = (s->slab_kind == 1) ? v / sigsq : v;
    }
}

/* SpikeSlabSampler::log_model_prob, SpikeSlabSampler.cpp:171-203 */
double bo_sss_log_model_prob(bo_sss *s, const uint8_t *g, double sigsq) {
  int *idx = s->g;
  int k = 0;
  for (int j = 0; j < s->p; ++j)
    if (g[j]) idx[k++] = j;
  double numerator = sss_spike_logp(s, g, k);
  if (numerator == BO_NEG_INF || k == 0) return numerator;
  double *prec = s->M1;
  sss_select_precision(s, idx, k, sigsq, prec);
  int ok = 1;
  numerator += .5 * bo_spd_logdet(k, prec, &ok);
  if (numerator == BO_NEG_INF) return numerator;
  double *mu = s->w1, *pmu = s->w2;
  for (int i = 0; i < k; ++i) mu[i] = s->mu[idx[i]];
  double dot = 0;
  for (int i = 0; i < k; ++i) {
    double acc = 0;
    for (int j = 0; j < k; ++j) acc += prec[IDX(i, j, k)] * mu[j];
    pmu[i] = acc;
  }
  for (int i = 0; i < k; ++i) dot += mu[i] * pmu[i];
  numerator -= .5 * dot;
  for (int c = 0; c < k; ++c)
    for (int r = 0; r < k; ++r)
      prec[IDX(r, c, k)] += s->xtx[IDX(idx[r], idx[c], s->p)] / sigsq;
  double *L = s->M2;
  if (!bo_chol(k, prec, L)) return BO_NEG_INF;
  double denominator = 0;
  for (int i = 0; i < k; ++i) denominator += log(L[IDX(i, i, k)]);
  double *S = s->w3;
  for (int i = 0; i < k; ++i) S[i] = s->xty[idx[i]] / sigsq + pmu[i];
  lsolve_inplace(k, L, S);
  double nsq = 0;
  for (int i = 0; i < k; ++i) nsq += S[i] * S[i];
  denominator -= .5 * nsq;
  return numerator - denominator;
}

/* draw_inclusion_indicators + draw_model_indicators,
 * SpikeSlabSampler.cpp:40-95 (the permutation restarts from the identity on
 * every call; max_flips_ limits only when > 0) */
int bo_sss_draw_model_indicators(bo_sss *s, double sigsq) {
  int p = s->p;
  uint8_t *g = (uint8_t *)malloc(p);
  memcpy(g, s->gamma, p);
  for (int j = 0; j < p; ++j) s->indx[j] = j;
  if (s->shuffle_kind == 1) {
    /* BinomialLogitSpikeSlabSampler::draw_model_indicators,
     * BinomialLogitSpikeSlabSampler.cpp:181-187: every position swaps with a
     * position drawn from the whole range */
    for (int i = 0; i < p; ++i) {
      int j = bo_random_int(&s->rng, 0, p - 1);
      int t = s->indx[i];
      s->indx[i] = s->indx[j];
      s->indx[j] = t;
    }
  } else {
    for (int i = p - 1; i > 0; --i) {
      int j = bo_random_int(&s->rng, 0, i);
      if (j != i) {
        int t = s->indx[i];
        s->indx[i] = s->indx[j];
        s->indx[j] = t;
      }
    }
  }
  double logp = bo_sss_log_model_prob(s, g, sigsq);
  if (!isfinite(logp)) {
    for (int i = 0; i < p; ++i) {
      if (s->pi[i] <= 0.0 && g[i]) g[i] = 0;
      if (s->pi[i] >= 1.0 && !g[i]) g[i] = 1;
    }
    logp = bo_sss_log_model_prob(s, g, sigsq);
  }
  if (!isfinite(logp)) {
    free(g);
    return BO_ERR_ILLEGAL_START;
  }
  int n = p;
  if (s->max_flips > 0 && s->max_flips < n) n = s->max_flips;
  for (int i = 0; i < n; ++i) {
    int which = s->indx[i];
    g[which] = !g[which];
    double logp_new = bo_sss_log_model_prob(s, g, sigsq);
    double u = bo_runif(&s->rng, 0, 1);
    if (log(u) > logp_new - logp) {
      g[which] = !g[which];
    } else {
      logp = logp_new;
    }
  }
  /* coef().set_inc: excluded coefficients go to zero */
  memcpy(s->gamma, g, p);
  for (int j = 0; j < p; ++j)
    if (!g[j]) s->beta[j] = 0.0;
  free(g);
  return BO_OK;
}

/* draw_beta / draw_coefficients_given_inclusion, SpikeSlabSampler.cpp:97-138,
 * rmvn_ivar_mt, distributions/mvn.cpp:104-122 */
int bo_sss_draw_beta(bo_sss *s, double sigsq) {
  int p = s->p;
  int *idx = s->g;
  int k = 0;
  for (int j = 0; j < p; ++j)
    if (s->gamma[j]) idx[k++] = j;
  if (k == 0) {
    memset(s->beta, 0, sizeof(double) * p);  /* model_->drop_all() */
    return BO_OK;
  }
  double *prec = s->M1, *pmu = s->w2, *mean = s->w3, *L = s->M2;
  sss_select_precision(s, idx, k, sigsq, prec);
  for (int i = 0; i < k; ++i) {
    double acc = 0;
    for (int j = 0; j < k; ++j) acc += prec[IDX(i, j, k)] * s->mu[idx[j]];
    pmu[i] = acc;
  }
  for (int c = 0; c < k; ++c)
    for (int r = 0; r < k; ++r)
      prec[IDX(r, c, k)] += s->xtx[IDX(idx[r], idx[c], p)] / sigsq;
  for (int i = 0; i < k; ++i) pmu[i] += s->xty[idx[i]] / sigsq;
  if (!bo_spd_solve(k, prec, pmu, mean)) return BO_ERR_NOT_PD;
  if (!bo_chol(k, prec, L)) return BO_ERR_NOT_PD;
  double *z = s->w1;
  for (int i = 0; i < k; ++i) z[i] = bo_rnorm(&s->rng, 0, 1);
  ltsolve_inplace(k, L, z);
  memset(s->beta, 0, sizeof(double) * p);
  for (int i = 0; i < k; ++i) s->beta[idx[i]] = z[i] + mean[i];
  return BO_OK;
}

/* ====================================================================== */
/*                               state space                              */
/* ====================================================================== */
struct bo_ss {
  int T, p;
  double *y, *X; /* X: T x p column-major */
  uint8_t *observed;
  bo_ssvs *reg;
  /* local level */
  double level_sigsq, level_prior_df, level_prior_ss, level_sigma_max;
  double a0, P0;
  double level_n, level_sumsq; /* ZeroMeanGaussianModel suf */
  bo_rng level_rng, state_rng;
  int latent_initialized;
  double *state;
  /* filter nodes: v, F, K, r for the data filter and the simulation filter */
  double *v, *F, *K, *r, *vs, *Fs, *Ks, *rs;
};

bo_ss *bo_ss_create(int T, int p, const double *y, const double *X,
                    const uint8_t *observed, const double *prior_mean,
                    const double *ominv, double prior_df, double sigma_guess,
                    const double *pi, double level_df,
                    double level_sigma_guess, double level_sigma_upper_limit,
                    double initial_state_mean, double initial_state_variance,
                    double initial_level_sigma) {
  bo_ss *m = (bo_ss *)xcalloc(1, sizeof(bo_ss));
  m->T = T;
  m->p = p;
  m->y = (double *)xcalloc(T, sizeof(double));
  m->X = (double *)xcalloc((size_t)T * p, sizeof(double));
  m->observed = (uint8_t *)xcalloc(T, 1);
  memcpy(m->y, y, sizeof(double) * T);
  memcpy(m->X, X, sizeof(double) * (size_t)T * p);
  for (int t = 0; t < T; ++t) m->observed[t] = observed ? observed[t] : 1;
  /* The ctor adds every RegressionData to the regression model, whose
   * NeRegSuf::Update accumulates xtx, xty, ... and then fixes xtx
   * (StateSpaceRegressionModel.cpp:100-125, :160-165;
   * RegressionModel.cpp:381-404).  Missing points do not update the
   * sufficient statistics (Models/Policies/SufstatDataPolicy.hpp:166-167),
   * so the fixed xtx is over observed rows only. */
  double *xtx = (double *)xcalloc((size_t)p * p, sizeof(double));
  double *xty = (double *)xcalloc(p, sizeof(double));
  double *xsum = (double *)xcalloc(p, sizeof(double));
  double yty = 0, sumy = 0, nobs0 = 0;
  for (int t = 0; t < T; ++t) {
    if (!m->observed[t]) continue;
    nobs0 += 1;
    for (int j = 0; j < p; ++j) {
      double xj = X[IDX(t, j, T)];
      xty[j] += xj * y[t];
      xsum[j] += xj;
      for (int i = 0; i < p; ++i) xtx[IDX(i, j, p)] += X[IDX(t, i, T)] * xj;
    }
    yty += y[t] * y[t];
    sumy += y[t];
  }
  m->reg = bo_ssvs_create(p, xtx, xty, yty, nobs0, sumy, xsum, prior_mean,
                          ominv, prior_df, sigma_guess, pi);
  free(xtx); free(xty); free(xsum);
  m->level_sigsq = initial_level_sigma * initial_level_sigma;
  m->level_prior_df = 2 * (level_df / 2.0);
  m->level_prior_ss = 2 * (level_df * level_sigma_guess * level_sigma_guess / 2.0);
  m->level_sigma_max = level_sigma_upper_limit;
  m->a0 = initial_state_mean;
  m->P0 = initial_state_variance;
  m->state = (double *)xcalloc(T, sizeof(double));
  m->v = (double *)xcalloc(T, sizeof(double));
  m->F = (double *)xcalloc(T, sizeof(double));
  m->K = (double *)xcalloc(T, sizeof(double));
  m->r = (double *)xcalloc(T, sizeof(double));
  m->vs = (double *)xcalloc(T, sizeof(double));
  m->Fs = (double *)xcalloc(T, sizeof(double));
  m->Ks = (double *)xcalloc(T, sizeof(double));
  m->rs = (double *)xcalloc(T, sizeof(double));
  bo_rng_seed_philox(&m->level_rng, 0, 0, 1, 0);
  bo_rng_seed_philox(&m->state_rng, 0, 0, 2, 0);
  return m;
}

void bo_ss_destroy(bo_ss *m) {
  if (!m) return;
  bo_ssvs_destroy(m->reg);
  free(m->y); free(m->X); free(m->observed); free(m->state);
  free(m->v); free(m->F); free(m->K); free(m->r);
  free(m->vs); free(m->Fs); free(m->Ks); free(m->rs);
  free(m);
}

bo_ssvs *bo_ss_regression(bo_ss *m) { return m->reg; }
bo_rng *bo_ss_level_rng(bo_ss *m) { return &m->level_rng; }
bo_rng *bo_ss_state_rng(bo_ss *m) { return &m->state_rng; }
void bo_ss_set_level_sigsq(bo_ss *m, double sigsq) { m->level_sigsq = sigsq; }
double bo_ss_level_sigsq(const bo_ss *m) { return m->level_sigsq; }
const double *bo_ss_state(const bo_ss *m) { return m->state; }
void bo_ss_level_suf(const bo_ss *m, double *n, double *sumsq) {
  *n = m->level_n;
  *sumsq = m->level_sumsq;
}

/* ScalarMarginalDistribution::update, ScalarKalmanFilter.cpp:41-83,
 * specialised to state dimension 1 with Z = 1, T = 1 (IdentityMatrix),
 * RQR = sigma^2_level (LocalLevelStateModel.cpp:32-91).  (a, P) enter as the
 * one-step-ahead moments and leave as the next ones. */
static int marginal_update(double y, int missing, double H, double rqr,
                           double *a, double *P, double *v, double *F,
                           double *K) {
  double PZ = *P;
  *F = PZ + H;
  if (*F <= 0) return BO_ERR_FORECAST_VARIANCE;
  double TPZ = PZ;
  if (!missing) {
    *K = TPZ / *F;
    double mu = *a;
    *v = y - mu;
  } else {
    *K = 0.0;
    *v = 0;
  }
  if (!missing) {
    *a = *a + *K * *v;
  }
  if (!missing) {
    *P = *P + (-1.0) * TPZ * *K;
  }
  *P = *P + rqr;
  return 0;
}

/* observation_variance(t), StateSpaceRegressionModel.cpp:167-177 */
static double observation_variance(const bo_ss *m, int t) {
  int n = m->observed[t] ? 1 : 0;
  if (n == 0) ++n;
  return m->reg->sigsq / n;
}

/* fast_disturbance_smooth, ScalarKalmanFilter.cpp:168-196.  Returns the
 * initial scaled state error r_{-1}. */
static double disturbance_smooth(int T, const double *v, const double *F,
                                 const double *K, double *rout) {
  double r = 0.0;
  for (int t = T - 1; t >= 0; --t) {
    double coefficient = (v[t] / F[t]) - K[t] * r;
    double rt_1 = r; /* T^T r with T = 1 */
    rt_1 += coefficient; /* Z = 1 */
    rout[t] = r;
    r = rt_1;
  }
  return r;
}

/* Base::impute_state, StateSpaceModelBase.cpp:278-291 = clear_client_data
 * (:248-254) + simulate_forward (:771-790) + propagate_disturbances (:858-891)
 */
int bo_ss_impute_state(bo_ss *m, bo_rng *rng) {
  int T = m->T, p = m->p;
  bo_ssvs *reg = m->reg;
  /* clear_client_data: NeRegSuf::clear keeps xtx (RegressionModel.cpp:372-379),
   * state-model suf cleared */
  double *xty = (double *)xcalloc(p, sizeof(double));
  double *xsum = (double *)xcalloc(p, sizeof(double));
  double yty = 0, nobs = 0, sumy = 0;
  m->level_n = 0;
  m->level_sumsq = 0;

  /* simulate_forward: filter.update() over the adjusted observations
   * (ScalarKalmanFilter.cpp:132-161; adjusted_observation
   * StateSpaceRegressionModel.cpp:65-77,179-181: y_t - x_t . Beta) */
  double a = m->a0, P = m->P0;
  int status = 0;
  for (int t = 0; t < T && !status; ++t) {
    int missing = !m->observed[t];
    double ystar = BO_NEG_INF;
    if (!missing) {
      double pred = 0;
      int nvars = 0;
      for (int j = 0; j < p; ++j) nvars += reg->gamma[j];
      if (nvars > 0)
        for (int j = 0; j < p; ++j) pred += m->X[IDX(t, j, T)] * reg->beta[j];
      ystar = (m->y[t] - pred) / 1;
    }
    status = marginal_update(ystar, missing, observation_variance(m, t),
                             m->level_sigsq, &a, &P, &m->v[t], &m->F[t],
                             &m->K[t]);
  }
  /* forward simulation + simulation filter (StateSpaceModelBase.cpp:771-790,
   * :430-443, :849-852; LocalLevelStateModel.cpp:62-69) */
  double as = m->a0, Ps = m->P0;
  double level_sigma = sqrt(m->level_sigsq);
  for (int t = 0; t < T && !status; ++t) {
    if (t == 0) {
      m->state[0] = bo_rnorm(rng, m->a0, sqrt(m->P0));
    } else {
      m->state[t] = m->state[t - 1] + bo_rnorm(rng, 0, level_sigma);
    }
    double H = observation_variance(m, t);
    double ysim = bo_rnorm(rng, m->state[t], sqrt(H));
    status = marginal_update(ysim, !m->observed[t], H, m->level_sigsq, &as, &Ps,
                             &m->vs[t], &m->Fs[t], &m->Ks[t]);
  }
  if (status) { free(xty); free(xsum); return status; }

  /* propagate_disturbances, StateSpaceModelBase.cpp:858-891 */
  double r0 = disturbance_smooth(T, m->v, m->F, m->K, m->r);
  double r0s = disturbance_smooth(T, m->vs, m->Fs, m->Ks, m->rs);
  double mean_sim = m->a0 + m->P0 * r0s;
  double mean_obs = m->a0 + m->P0 * r0;
  for (int t = 0; t < T; ++t) {
    if (t > 0) {
      mean_sim = mean_sim + m->level_sigsq * m->rs[t - 1];
      mean_obs = mean_obs + m->level_sigsq * m->r[t - 1];
    }
    m->state[t] += mean_obs - mean_sim;
    /* observe_state (LocalLevelStateModel.cpp:52-58; GaussianSuf::update_raw) */
    if (t > 0) {
      double diff = m->state[t] - m->state[t - 1];
      m->level_n += 1;
      m->level_sumsq += diff * diff;
    }
    /* observe_data_given_state, StateSpaceRegressionModel.cpp:188-200 +
     * NeRegSuf::add_mixture_data, RegressionModel.cpp:356-370 */
    if (m->observed[t]) {
      double resid = m->y[t] - m->state[t];
      for (int j = 0; j < p; ++j) {
        double xj = m->X[IDX(t, j, T)];
        xty[j] += xj * (resid * 1.0);
        xsum[j] += xj * 1.0;
      }
      yty += resid * resid * 1.0;
      nobs += 1.0;
      sumy += resid * 1.0;
    }
  }
  bo_ssvs_set_suf(reg, xty, yty, nobs, sumy, xsum);
  free(xty);
  free(xsum);
  return 0;
}

/* StateSpacePosteriorSampler::draw, StateSpacePosteriorSampler.cpp:42-64 */
int bo_ss_draw(bo_ss *m) {
  int status = 0;
  if (!m->latent_initialized) {
    status = bo_ss_impute_state(m, &m->state_rng);
    if (status) return status;
    m->latent_initialized = 1;
  }
  status = ssvs_draw(m->reg);
  if (status) return status;
  /* ZeroMeanGaussianConjSampler::draw, ZeroMeanGaussianConjSampler.cpp:57-60 */
  m->level_sigsq = variance_draw(&m->level_rng, m->level_prior_df,
                                 m->level_prior_ss, m->level_sigma_max,
                                 m->level_n, m->level_sumsq, &status);
  if (status) return status;
  return bo_ss_impute_state(m, &m->state_rng);
}

/* ====================================================================== *
 * Structural time series: StateSpaceRegressionModel with ANY list of state
 * models added by add_state, in any order (StateSpaceModelBase.hpp:637-638; the
 * transition / variance matrices are block diagonal, Filters/SparseMatrix.hpp:2196):
 *   LocalLevelStateModel                       (1 component, 1 variance)
 *   LocalLinearTrendStateModel                 (2 components, one
 *                                               ZeroMeanMvnIndependenceSampler per
 *                                               variance, as bsts builds it)
 *   SeasonalStateModel(nseasons, duration)     (nseasons - 1 components, 1 variance;
 *                                               T / RQR are the seasonal matrices on
 *                                               steps INTO a new season and identity /
 *                                               zero inside one,
 *                                               SeasonalStateModel.cpp:89-104, :248-258)
 *   ArStateModel(lags)                         (lags components, phi, 1 variance,
 *                                               ArPosteriorSampler)
 *   StaticInterceptStateModel                  (1 component, T = 1, no state error, no
 *                                               parameter and no sampler: the value is its
 *                                               initial draw, moved by the smoother only;
 *                                               StaticInterceptStateModel.hpp:35-131, .cpp:29-54)
 *   TrigStateModel(period, frequencies)        (2 components per frequency, T = the 2 x 2
 *                                               rotations [[c, s], [-s, c]] of 2 pi f / period,
 *                                               Z = 1 at each pair's first component, ONE error
 *                                               variance for all components with its
 *                                               ZeroMeanGaussianConjSampler;
 *                                               TrigStateModel.cpp:130-223)
 *   SemilocalLinearTrendStateModel             (3 components: level, slope, the slope's
 *                                               long-run mean; T = [[1, 1, 0], [0, phi, 1 - phi],
 *                                               [0, 0, 1]]; errors on level and slope; the level's
 *                                               ZeroMeanGaussianConjSampler and the slope's
 *                                               NonzeroMeanAr1Sampler (mu, phi, sigma), as bsts
 *                                               builds it; SemilocalLinearTrend.cpp:29-272,
 *                                               NonzeroMeanAr1Sampler.cpp:51-155,
 *                                               NonzeroMeanAr1Model.cpp:31-120)
 * SURVEY 8f row f2.  State vector = the blocks one after the other, dimension
 * m <= 64.
 *   Z      ones at the first element of each block
 *          (LocalLinearTrend.cpp:37, SeasonalStateModel.cpp:148-152, ArStateModel.cpp)
 *   T      local level [1]; trend [[1, 1], [0, 1]] (LocalLinearTrendMatrix);
 *          seasonal: first row -1, identity below the diagonal
 *          (SeasonalStateSpaceMatrix, Filters/SparseMatrix.cpp:1141-1149);
 *          autoregression: first row phi, identity below the diagonal
 *   RQR    diagonal: level; level, slope; the seasonal / autoregression block's
 *          first element
 * ====================================================================== */
#define BO_SSM_MAX 64
#define BO_SSM_BLOCKS 8
#define BO_AR_MAX 16
typedef struct {
  int kind;            /* BO_BLK_* */
  int first, dim;      /* position in the state vector */
  int nvar;            /* variance parameters: 2 for the local linear trend, else 1 */
  int nseasons, duration, t0;   /* seasonal: t0 = time_of_first_observation */
  int lags;                     /* autoregression */
  double sigsq[2], prior_df[2], prior_ss[2], sigma_max[2];
  double suf_n[2], suf_ss[2];
  /* MvnSuf of the trend's state errors (Welford form, MvnBase.cpp:71-86) */
  double mv_n, mv_ybar[2], mv_sumsq[2];
  bo_rng rng[2];       /* one per variance sampler (autoregression: the ArPosteriorSampler's) */
  /* ArStateModel: coefficients and the NeRegSuf of the block's first element on the
   * block's previous value */
  double phi[BO_AR_MAX], ar_xtx[BO_AR_MAX * BO_AR_MAX], ar_xty[BO_AR_MAX], ar_yty, ar_n;
  /* SemilocalLinearTrendStateModel: the slope's NonzeroMeanAr1Model (mu, phi; sigsq[1]), its
   * Ar1Suf, the sampler's priors (mean: N(m, s^2); phi: N(m, s^2)) and switches */
  double sl_mu, sl_phi, sl_mean_prior[2], sl_phi_prior[2];
  int sl_truncate_phi, sl_force_positive;
  double a1_sumsq, a1_sum, a1_cross, a1_n, a1_first, a1_last;
  /* TrigStateModel: the rotation of pair q (components first + 2 q, first + 2 q + 1) */
  int nfreq;
  double trig_cos[BO_SSM_MAX / 2], trig_sin[BO_SSM_MAX / 2];
} bo_ssm_block;

struct bo_ssm {
  int T, p, m, nblocks;
  bo_ssm_block blk[BO_SSM_BLOCKS];
  double *y, *X;
  uint8_t *observed;
  bo_ssvs *reg;
  double a0[BO_SSM_MAX], P0[BO_SSM_MAX]; /* initial mean, initial variance (diagonal) */
  int nz, zpos[BO_SSM_MAX];               /* the components Z selects (observation_matrix: ones there) */
  bo_rng state_rng;
  int latent_initialized;
  double *state; /* m x T, column t = state at t */
  double *v, *F, *K, *r, *vs, *Fs, *Ks, *rs;
  /* draw_phi's proposals come from rmvn_ivar (no _mt): the reference draws them from
   * GlobalRng::rng, not from the sampler's generator.  MT mode points this at the
   * restated global generator (after it seeded the samplers); NULL: the block's rng. */
  bo_rng *ar_global_rng;
};

static void ssm_alloc_series(bo_ssm *m) {
  const size_t mT = (size_t)(m->m > 0 ? m->m : 1) * m->T;
  free(m->state); free(m->K); free(m->r); free(m->Ks); free(m->rs);
  m->state = (double *)xcalloc(mT, sizeof(double));
  m->K = (double *)xcalloc(mT, sizeof(double));
  m->r = (double *)xcalloc(mT, sizeof(double));
  m->Ks = (double *)xcalloc(mT, sizeof(double));
  m->rs = (double *)xcalloc(mT, sizeof(double));
}

/* the Philox sampler id of variance parameter v of block b: level 1, slope 6,
 * seasonal 7, autoregression 12 for the first block of its family (local level and
 * local linear trend are one family), + 16 for every earlier block of the family */
int bo_ssm_block_stream_id(const bo_ssm *m, int b, int v) {
  const int kind = m->blk[b].kind;
  /* (the semilocal trend's two samplers: its level variance is of the level family -- id 1 --,
   * the NonzeroMeanAr1Sampler a family of its own, id 14) */
  if (kind == BO_BLK_SEMILOCAL && v == 1) {
    int occ = 0;
    for (int i = 0; i < b; ++i) occ += m->blk[i].kind == BO_BLK_SEMILOCAL;
    return 14 + 16 * occ;
  }
  const int fam = (kind == BO_BLK_LOCAL_LINEAR_TREND || kind == BO_BLK_SEMILOCAL) ? BO_BLK_LOCAL_LEVEL : kind;
  int occ = 0;
  for (int i = 0; i < b; ++i) {
    const int k = m->blk[i].kind;
    if (((k == BO_BLK_LOCAL_LINEAR_TREND || k == BO_BLK_SEMILOCAL) ? BO_BLK_LOCAL_LEVEL : k) == fam) ++occ;
  }
  const int base = (kind == BO_BLK_SEASONAL) ? 7 : (kind == BO_BLK_AR) ? 12 : (kind == BO_BLK_TRIG) ? 13
                   : (v == 0 ? 1 : 6);
  return base + 16 * occ;
}

/* StateSpaceRegressionModel(y, X, observed) + BregVsSampler, no state yet */
bo_ssm *bo_ssm_create_empty(int T, int p, const double *y, const double *X,
                            const uint8_t *observed, const double *prior_mean,
                            const double *ominv, double prior_df, double sigma_guess,
                            const double *pi) {
  bo_ssm *m = (bo_ssm *)xcalloc(1, sizeof(bo_ssm));
  m->T = T;
  m->p = p;
  m->y = (double *)xcalloc(T, sizeof(double));
  m->X = (double *)xcalloc((size_t)T * p, sizeof(double));
  m->observed = (uint8_t *)xcalloc(T, 1);
  memcpy(m->y, y, sizeof(double) * T);
  memcpy(m->X, X, sizeof(double) * (size_t)T * p);
  for (int t = 0; t < T; ++t) m->observed[t] = observed ? observed[t] : 1;
  /* (the regression part is bo_ss_create's: fixed xtx over the observed rows) */
  double *xtx = (double *)xcalloc((size_t)p * p, sizeof(double));
  double *xty = (double *)xcalloc(p, sizeof(double));
  double *xsum = (double *)xcalloc(p, sizeof(double));
  double yty = 0, sumy = 0, nobs0 = 0;
  for (int t = 0; t < T; ++t) {
    if (!m->observed[t]) continue;
    nobs0 += 1;
    for (int j = 0; j < p; ++j) {
      double xj = X[IDX(t, j, T)];
      xty[j] += xj * y[t];
      xsum[j] += xj;
      for (int i = 0; i < p; ++i) xtx[IDX(i, j, p)] += X[IDX(t, i, T)] * xj;
    }
    yty += y[t] * y[t];
    sumy += y[t];
  }
  m->reg = bo_ssvs_create(p, xtx, xty, yty, nobs0, sumy, xsum, prior_mean,
                          ominv, prior_df, sigma_guess, pi);
  free(xtx); free(xty); free(xsum);
  m->v = (double *)xcalloc(T, sizeof(double));
  m->F = (double *)xcalloc(T, sizeof(double));
  m->vs = (double *)xcalloc(T, sizeof(double));
  m->Fs = (double *)xcalloc(T, sizeof(double));
  ssm_alloc_series(m);
  bo_rng_seed_philox(&m->state_rng, 0, 0, 2, 0);
  return m;
}

/* model->add_state(...): kind BO_BLK_*; iparams = {nseasons, season_duration,
 * time_of_first_observation} (seasonal) or {lags} (autoregression), ignored otherwise;
 * the variance arrays have one entry per variance parameter (two for the local linear
 * trend: level, slope): ChisqModel(df, sigma_guess) prior, sigma upper limit (inf:
 * none), initial sigma; initial_phi: lags entries (NULL: zeros); the block's initial
 * state is N(mean, diag(variance)), dim entries each.  Before the first draw. */
int bo_ssm_add_block(bo_ssm *m, int kind, const int *iparams, const double *var_df,
                     const double *var_sigma_guess, const double *var_sigma_upper_limit,
                     const double *var_initial_sigma, const double *initial_phi,
                     const double *initial_state_mean, const double *initial_state_variance) {
  if (m->nblocks >= BO_SSM_BLOCKS) return BO_ERR_INVALID;
  bo_ssm_block *b = &m->blk[m->nblocks];
  memset(b, 0, sizeof(*b));
  b->kind = kind;
  b->nvar = 1;
  b->duration = 1;
  switch (kind) {
    case BO_BLK_LOCAL_LEVEL: b->dim = 1; break;
    case BO_BLK_LOCAL_LINEAR_TREND: b->dim = 2; b->nvar = 2; break;
    case BO_BLK_SEASONAL:
      if (!iparams || iparams[0] < 2 || iparams[1] < 1) return BO_ERR_INVALID;
      b->nseasons = iparams[0];
      b->duration = iparams[1];
      b->t0 = iparams[2];
      b->dim = b->nseasons - 1;
      break;
    case BO_BLK_AR:
      if (!iparams || iparams[0] < 1 || iparams[0] > BO_AR_MAX) return BO_ERR_INVALID;
      b->lags = iparams[0];
      b->dim = b->lags;
      break;
    case BO_BLK_STATIC_INTERCEPT: b->dim = 1; b->nvar = 0; break;   /* (no parameter, no sampler) */
    case BO_BLK_TRIG:
      /* iparams = {number of frequencies}; initial_phi = the rotations' (cos, sin) pairs, as
       * the transition matrix holds them (TrigStateModel.cpp:144-153) */
      if (!iparams || iparams[0] < 1 || 2 * iparams[0] > BO_SSM_MAX || !initial_phi) return BO_ERR_INVALID;
      b->nfreq = iparams[0];
      b->dim = 2 * b->nfreq;
      for (int q = 0; q < b->nfreq; ++q) {
        b->trig_cos[q] = initial_phi[2 * q];
        b->trig_sin[q] = initial_phi[2 * q + 1];
      }
      break;
    case BO_BLK_SEMILOCAL:
      /* iparams = {force_stationary, force_ar1_positive}; initial_phi = {slope mean prior mu,
       * sigma; slope AR(1) coefficient prior mu, sigma; initial mu, initial phi}; the var_* arrays:
       * level, slope; the initial state: level, slope (the third component is mu) */
      if (!iparams || !initial_phi) return BO_ERR_INVALID;
      if (iparams[1] && !iparams[0]) return BO_ERR_INVALID;   /* (one-sided truncation: not restated) */
      b->dim = 3;
      b->nvar = 2;
      b->sl_truncate_phi = iparams[0] != 0;
      b->sl_force_positive = iparams[1] != 0;
      b->sl_mean_prior[0] = initial_phi[0]; b->sl_mean_prior[1] = initial_phi[1];
      b->sl_phi_prior[0] = initial_phi[2]; b->sl_phi_prior[1] = initial_phi[3];
      b->sl_mu = initial_phi[4];
      b->sl_phi = initial_phi[5];
      break;
    default: return BO_ERR_INVALID;
  }
  if (m->m + b->dim > BO_SSM_MAX) return BO_ERR_INVALID;
  b->first = m->m;
  /* observation_matrix: one at the block's first component (every pair's first: trig) */
  for (int i = 0; i < b->dim; i += (kind == BO_BLK_TRIG ? 2 : b->dim)) m->zpos[m->nz++] = b->first + i;
  for (int v = 0; v < b->nvar; ++v) {
    b->sigsq[v] = var_initial_sigma[v] * var_initial_sigma[v];
    b->prior_df[v] = 2 * (var_df[v] / 2.0);
    b->prior_ss[v] = 2 * (var_df[v] * var_sigma_guess[v] * var_sigma_guess[v] / 2.0);
    b->sigma_max[v] = var_sigma_upper_limit[v];
  }
  for (int i = 0; i < b->dim; ++i) {
    m->a0[b->first + i] = initial_state_mean[i];
    m->P0[b->first + i] = initial_state_variance[i];
  }
  if (kind == BO_BLK_SEMILOCAL) {
    /* initial_state_mean()[2] = slope_->mu() (kept current by ssm_refresh_semilocal), variance 0 */
    m->a0[b->first + 2] = b->sl_mu;
    m->P0[b->first + 2] = 0.0;
  }
  if (kind == BO_BLK_AR)
    for (int i = 0; i < b->lags; ++i) b->phi[i] = initial_phi ? initial_phi[i] : 0.0;
  m->m += b->dim;
  m->nblocks += 1;
  for (int v = 0; v < b->nvar; ++v)
    bo_rng_seed_philox(&b->rng[v], 0, 0, (uint32_t)bo_ssm_block_stream_id(m, m->nblocks - 1, v), 0);
  ssm_alloc_series(m);
  return 0;
}

/* the first block of a kind (-1: none) */
static int ssm_find(const bo_ssm *m, int kind) {
  for (int b = 0; b < m->nblocks; ++b)
    if (m->blk[b].kind == kind) return b;
  return -1;
}

/* the template of rounds 2-3: a trend block (trend = 1: local level, 2: local linear
 * trend) + an optional SeasonalStateModel(nseasons, 1); three-element arrays indexed
 * level, slope, seasonal */
bo_ssm *bo_ssm_create(int T, int p, const double *y, const double *X,
                      const uint8_t *observed, const double *prior_mean,
                      const double *ominv, double prior_df, double sigma_guess,
                      const double *pi, int trend, int nseasons,
                      const double *var_df, const double *var_sigma_guess,
                      const double *var_sigma_upper_limit,
                      const double *var_initial_sigma,
                      const double *initial_state_mean,
                      const double *initial_state_variance) {
  bo_ssm *m = bo_ssm_create_empty(T, p, y, X, observed, prior_mean, ominv, prior_df,
                                  sigma_guess, pi);
  bo_ssm_add_block(m, trend == 2 ? BO_BLK_LOCAL_LINEAR_TREND : BO_BLK_LOCAL_LEVEL, NULL, var_df,
                   var_sigma_guess, var_sigma_upper_limit, var_initial_sigma, NULL,
                   initial_state_mean, initial_state_variance);
  if (nseasons > 0) {
    const int ip[3] = {nseasons, 1, 0};
    bo_ssm_add_block(m, BO_BLK_SEASONAL, ip, var_df + 2, var_sigma_guess + 2,
                     var_sigma_upper_limit + 2, var_initial_sigma + 2, NULL,
                     initial_state_mean + trend, initial_state_variance + trend);
  }
  return m;
}

/* model->add_state(new ArStateModel(lags)) with an ArPosteriorSampler(ChisqModel(df,
 * sigma_guess)) [+ set_sigma_upper_limit]; before the first draw.  The block's
 * initial state is N(mean, diag(variance)). */
int bo_ssm_add_ar(bo_ssm *m, int lags, double prior_df, double sigma_guess,
                  double sigma_upper_limit, double initial_sigma, const double *initial_phi,
                  const double *initial_state_mean, const double *initial_state_variance) {
  if (ssm_find(m, BO_BLK_AR) >= 0) return BO_ERR_INVALID;
  const int ip[3] = {lags, 0, 0};
  return bo_ssm_add_block(m, BO_BLK_AR, ip, &prior_df, &sigma_guess, &sigma_upper_limit,
                          &initial_sigma, initial_phi, initial_state_mean, initial_state_variance);
}
int bo_ssm_nblocks(const bo_ssm *m) { return m->nblocks; }
bo_rng *bo_ssm_block_rng(bo_ssm *m, int b, int v) { return &m->blk[b].rng[v]; }
/* block b: variances (nvar), sufficient statistics (n, sum of squares per variance),
 * autoregression coefficients (lags); any pointer may be NULL */
void bo_ssm_block_get(const bo_ssm *m, int b, double *sigsq, double *suf_n, double *suf_ss,
                      double *phi) {
  const bo_ssm_block *B = &m->blk[b];
  for (int v = 0; v < B->nvar; ++v) {
    if (sigsq) sigsq[v] = B->sigsq[v];
    if (suf_n) suf_n[v] = B->suf_n[v];
    if (suf_ss) suf_ss[v] = B->suf_ss[v];
  }
  if (phi) for (int i = 0; i < B->lags; ++i) phi[i] = B->phi[i];
  if (phi && B->kind == BO_BLK_SEMILOCAL) { phi[0] = B->sl_phi; phi[1] = B->sl_mu; }
}
void bo_ssm_block_set_sigsq(bo_ssm *m, int b, const double *sigsq) {
  for (int v = 0; v < m->blk[b].nvar; ++v) m->blk[b].sigsq[v] = sigsq[v];
}
void bo_ssm_block_get_ar_suf(const bo_ssm *m, int b, double *xtx, double *xty, double *yty,
                             double *n) {
  const bo_ssm_block *B = &m->blk[b];
  const int L = B->lags;
  for (int i = 0; i < L * L; ++i) xtx[i] = B->ar_xtx[i];
  for (int i = 0; i < L; ++i) xty[i] = B->ar_xty[i];
  *yty = B->ar_yty;
  *n = B->ar_n;
}
static bo_rng g_unused_rng;   /* (what the template's accessors hand out for a sampler the model does not have) */
bo_rng *bo_ssm_ar_rng(bo_ssm *m) {
  const int b = ssm_find(m, BO_BLK_AR);
  return b < 0 ? &g_unused_rng : &m->blk[b].rng[0];
}
void bo_ssm_set_global_rng(bo_ssm *m, bo_rng *global) { m->ar_global_rng = global; }
void bo_ssm_get_ar(const bo_ssm *m, double *phi, double *sigsq) {
  bo_ssm_block_get(m, ssm_find(m, BO_BLK_AR), sigsq, NULL, NULL, phi);
}
void bo_ssm_get_ar_suf(const bo_ssm *m, double *xtx, double *xty, double *yty, double *n) {
  bo_ssm_block_get_ar_suf(m, ssm_find(m, BO_BLK_AR), xtx, xty, yty, n);
}

void bo_ssm_destroy(bo_ssm *m) {
  if (!m) return;
  bo_ssvs_destroy(m->reg);
  free(m->y); free(m->X); free(m->observed); free(m->state);
  free(m->v); free(m->F); free(m->K); free(m->r);
  free(m->vs); free(m->Fs); free(m->Ks); free(m->rs);
  free(m);
}
bo_ssvs *bo_ssm_regression(bo_ssm *m) { return m->reg; }
/* the template's variance parameters: 0 level, 1 slope (block 0), 2 the first seasonal block */
static bo_ssm_block *ssm_template_var(const bo_ssm *m, int which, int *v) {
  *v = which == 1 ? 1 : 0;
  if (which < 2) return (bo_ssm_block *)&m->blk[0];
  const int b = ssm_find(m, BO_BLK_SEASONAL);
  return b < 0 ? NULL : (bo_ssm_block *)&m->blk[b];
}
bo_rng *bo_ssm_variance_rng(bo_ssm *m, int which) {
  int v;
  bo_ssm_block *B = ssm_template_var(m, which, &v);
  return (B && v < B->nvar) ? &B->rng[v] : &g_unused_rng;
}
bo_rng *bo_ssm_state_rng(bo_ssm *m) { return &m->state_rng; }
int bo_ssm_state_dimension(const bo_ssm *m) { return m->m; }
const double *bo_ssm_state(const bo_ssm *m) { return m->state; }
void bo_ssm_get_variances(const bo_ssm *m, double *sigsq) {
  for (int i = 0; i < 3; ++i) {
    int v;
    const bo_ssm_block *B = ssm_template_var(m, i, &v);
    sigsq[i] = (B && v < B->nvar) ? B->sigsq[v] : 0.0;
  }
}
void bo_ssm_set_variances(bo_ssm *m, const double *sigsq) {
  for (int i = 0; i < 3; ++i) {
    int v;
    bo_ssm_block *B = ssm_template_var(m, i, &v);
    if (B && v < B->nvar) B->sigsq[v] = sigsq[i];
  }
}

/* SeasonalStateModel::new_season, SeasonalStateModel.cpp:248-258 */
static int blk_new_season(const bo_ssm_block *b, int t) {
  t -= b->t0;
  if (t < 0) t -= b->duration * t;
  return (t % b->duration) == 0;
}
/* does block b's transition matrix of time t (the step from t to t + 1) move anything?
 * (state_transition_matrix(t), SeasonalStateModel.cpp:89-92: identity inside a season) */
static int blk_moves(const bo_ssm_block *b, int t) {
  return b->kind != BO_BLK_SEASONAL || blk_new_season(b, t + 1);
}

/* x <- T_t x (multiply_inplace of the block-diagonal transition matrix) */
static void ssm_T(const bo_ssm *m, double *x, int t) {
  for (int bi = 0; bi < m->nblocks; ++bi) {
    const bo_ssm_block *b = &m->blk[bi];
    double *s = x + b->first;
    if (b->kind == BO_BLK_LOCAL_LINEAR_TREND) {
      s[0] = s[0] + s[1];
    } else if (b->kind == BO_BLK_SEASONAL) {
      if (!blk_moves(b, t)) continue;
      const int n = b->dim;
      double tmp[BO_SSM_MAX], first = 0;
      for (int i = 0; i < n; ++i) {
        first -= s[i];
        if (i > 0) tmp[i] = s[i - 1];
      }
      tmp[0] = first;
      for (int i = 0; i < n; ++i) s[i] = tmp[i];
    } else if (b->kind == BO_BLK_AR) {
      /* AutoRegressionTransitionMatrix::multiply_inplace, Filters/SparseMatrix.cpp:1297-1310 */
      double first_entry = 0;
      for (int i = b->lags - 1; i >= 0; --i) {
        first_entry += b->phi[i] * s[i];
        if (i > 0) s[i] = s[i - 1]; else s[i] = first_entry;
      }
    } else if (b->kind == BO_BLK_SEMILOCAL) {
      /* SemilocalLinearTrendMatrix::multiply_inplace, SemilocalLinearTrend.cpp:78-82 */
      s[0] += s[1];
      s[1] = b->sl_phi * s[1] + (1 - b->sl_phi) * s[2];
    } else if (b->kind == BO_BLK_TRIG) {
      /* BlockDiagonalMatrixBlock of DenseMatrix rotations: lhs = rotation * rhs, a row at a time
       * (Matrix::mult: the row's products summed from the left) */
      for (int q = 0; q < b->nfreq; ++q) {
        const double c = b->trig_cos[q], sn = b->trig_sin[q], x0 = s[2 * q], x1 = s[2 * q + 1];
        s[2 * q] = c * x0 + sn * x1;
        s[2 * q + 1] = -sn * x0 + c * x1;
      }
    }
  }
}
/* x <- T_t' x (Tmult) */
static void ssm_Tt(const bo_ssm *m, double *x, int t) {
  for (int bi = 0; bi < m->nblocks; ++bi) {
    const bo_ssm_block *b = &m->blk[bi];
    double *s = x + b->first;
    if (b->kind == BO_BLK_LOCAL_LINEAR_TREND) {
      s[1] = s[0] + s[1];
    } else if (b->kind == BO_BLK_SEASONAL) {
      if (!blk_moves(b, t)) continue;
      const int n = b->dim;
      double tmp[BO_SSM_MAX];
      for (int i = 0; i < n; ++i) tmp[i] = -s[0] + (i + 1 < n ? s[i + 1] : 0.0);
      for (int i = 0; i < n; ++i) s[i] = tmp[i];
    } else if (b->kind == BO_BLK_AR) {
      /* AutoRegressionTransitionMatrix::Tmult, Filters/SparseMatrix.cpp:1286-1295 */
      const int n = b->lags;
      double tmp[BO_AR_MAX];
      for (int i = 0; i < n; ++i) tmp[i] = b->phi[i] * s[0] + (i + 1 < n ? s[i + 1] : 0);
      for (int i = 0; i < n; ++i) s[i] = tmp[i];
    } else if (b->kind == BO_BLK_SEMILOCAL) {
      /* SemilocalLinearTrendMatrix::Tmult, SemilocalLinearTrend.cpp:65-76 */
      const double r0 = s[0], r1 = s[1], r2 = s[2];
      s[0] = r0;
      s[1] = r0 + b->sl_phi * r1;
      s[2] = (1 - b->sl_phi) * r1 + r2;
    } else if (b->kind == BO_BLK_TRIG) {
      /* the rotations' transposes */
      for (int q = 0; q < b->nfreq; ++q) {
        const double c = b->trig_cos[q], sn = b->trig_sin[q], x0 = s[2 * q], x1 = s[2 * q + 1];
        s[2 * q] = c * x0 + -sn * x1;
        s[2 * q + 1] = sn * x0 + c * x1;
      }
    }
  }
}
static double ssm_Zdot(const bo_ssm *m, const double *x) {
  double ans = x[m->zpos[0]];
  for (int b = 1; b < m->nz; ++b) ans += x[m->zpos[b]];
  return ans;
}
/* the diagonal of RQR_t */
static void ssm_rqr(const bo_ssm *m, double *d, int t) {
  for (int i = 0; i < m->m; ++i) d[i] = 0;
  for (int bi = 0; bi < m->nblocks; ++bi) {
    const bo_ssm_block *b = &m->blk[bi];
    if (b->kind == BO_BLK_SEASONAL && !blk_moves(b, t)) continue;   /* RQR1_ = ZeroMatrix */
    if (b->kind == BO_BLK_STATIC_INTERCEPT) continue;                /* state_variance_matrix: ZeroMatrix(1) */
    if (b->kind == BO_BLK_TRIG) {
      /* ConstantMatrixParamView(2 nfreq, sigsq): sigsq on the whole diagonal */
      for (int i = 0; i < b->dim; ++i) d[b->first + i] = b->sigsq[0];
      continue;
    }
    d[b->first] = b->sigsq[0];
    /* (semilocal: UpperLeftDiagonalMatrix(level, slope; 3): the third diagonal element is 0) */
    if (b->kind == BO_BLK_LOCAL_LINEAR_TREND || b->kind == BO_BLK_SEMILOCAL) d[b->first + 1] = b->sigsq[1];
  }
}

/* ScalarMarginalDistribution::update, ScalarKalmanFilter.cpp:41-83; P is m x m
 * column-major, (a, P) enter as the one-step-ahead moments of time t and leave as
 * those of t + 1 */
static int ssm_update(const bo_ssm *M, int t, double y, int missing, double H,
                      double *a, double *P, double *v, double *F, double *K) {
  const int m = M->m;
  double PZ[BO_SSM_MAX], TPZ[BO_SSM_MAX], rqr[BO_SSM_MAX];
  for (int i = 0; i < m; ++i) {
    PZ[i] = P[IDX(i, M->zpos[0], m)];
    for (int b = 1; b < M->nz; ++b) PZ[i] += P[IDX(i, M->zpos[b], m)];
  }
  *F = ssm_Zdot(M, PZ) + H;
  if (*F <= 0) return BO_ERR_FORECAST_VARIANCE;
  for (int i = 0; i < m; ++i) TPZ[i] = PZ[i];
  ssm_T(M, TPZ, t);
  if (!missing) {
    for (int i = 0; i < m; ++i) K[i] = TPZ[i] / *F;
    *v = y - ssm_Zdot(M, a);
  } else {
    for (int i = 0; i < m; ++i) K[i] = 0.0;
    *v = 0;
  }
  ssm_T(M, a, t);
  if (!missing)
    for (int i = 0; i < m; ++i) a[i] += K[i] * *v;
  /* sandwich_inplace (Filters/SparseMatrix.cpp:1748-1763): T times every
   * column, then T times every row */
  double col[BO_SSM_MAX];
  for (int j = 0; j < m; ++j) {
    for (int i = 0; i < m; ++i) col[i] = P[IDX(i, j, m)];
    ssm_T(M, col, t);
    for (int i = 0; i < m; ++i) P[IDX(i, j, m)] = col[i];
  }
  for (int i = 0; i < m; ++i) {
    for (int j = 0; j < m; ++j) col[j] = P[IDX(i, j, m)];
    ssm_T(M, col, t);
    for (int j = 0; j < m; ++j) P[IDX(i, j, m)] = col[j];
  }
  if (!missing)
    for (int j = 0; j < m; ++j)
      for (int i = 0; i < m; ++i) P[IDX(i, j, m)] += -1.0 * TPZ[i] * K[j];
  ssm_rqr(M, rqr, t);
  for (int i = 0; i < m; ++i) P[IDX(i, i, m)] += rqr[i];
  /* fix_near_symmetry, SpdMatrix.cpp:350-357 */
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < i; ++j) {
      double value = .5 * (P[IDX(i, j, m)] + P[IDX(j, i, m)]);
      P[IDX(i, j, m)] = P[IDX(j, i, m)] = value;
    }
  return 0;
}

/* fast_disturbance_smooth, ScalarKalmanFilter.cpp:168-196: rout[:, t] = r_t,
 * r0 = the initial scaled state error */
static void ssm_disturbance_smooth(const bo_ssm *M, const double *v,
                                   const double *F, const double *K,
                                   double *rout, double *r0) {
  const int m = M->m, T = M->T;
  double r[BO_SSM_MAX];
  for (int i = 0; i < m; ++i) r[i] = 0.0;
  for (int t = T - 1; t >= 0; --t) {
    double kr = 0;
    for (int i = 0; i < m; ++i) kr += K[IDX(i, t, m)] * r[i];
    double coefficient = (v[t] / F[t]) - kr;
    double rt_1[BO_SSM_MAX];
    for (int i = 0; i < m; ++i) rt_1[i] = r[i];
    ssm_Tt(M, rt_1, t);
    for (int b = 0; b < M->nz; ++b) rt_1[M->zpos[b]] += coefficient;
    for (int i = 0; i < m; ++i) rout[IDX(i, t, m)] = r[i];
    for (int i = 0; i < m; ++i) r[i] = rt_1[i];
  }
  for (int i = 0; i < m; ++i) r0[i] = r[i];
}

/* Ar1Suf::update_raw, NonzeroMeanAr1Model.cpp:39-49 */
static void ar1_update(bo_ssm_block *b, double y) {
  if (b->a1_n == 0) {
    b->a1_first = y;
  } else {
    b->a1_cross += y * b->a1_last;
  }
  b->a1_n += 1;
  b->a1_sum += y;
  b->a1_sumsq += y * y;
  b->a1_last = y;
}

static double ssm_observation_variance(const bo_ssm *m, int t) {
  (void)t;
  return m->reg->sigsq; /* one observation per time point, missing or not (StateSpaceRegressionModel.cpp:167-177) */
}

/* the state errors of the step from t to t + 1, every state model in turn
 * (simulate_state_error(rng, eta, t)) */
static void ssm_state_error(const bo_ssm *M, bo_rng *rng, double *eta, int t) {
  for (int i = 0; i < M->m; ++i) eta[i] = 0;
  for (int bi = 0; bi < M->nblocks; ++bi) {
    const bo_ssm_block *b = &M->blk[bi];
    if (b->kind == BO_BLK_LOCAL_LEVEL) {
      eta[b->first] = bo_rnorm(rng, 0, sqrt(b->sigsq[0]));          /* LocalLevelStateModel.cpp:62-64 */
    } else if (b->kind == BO_BLK_LOCAL_LINEAR_TREND) {
      /* ZeroMeanMvnModel::sim = rmvn_mt(0, Sigma), Sigma diagonal (MvnBase.cpp:257) */
      double z0 = bo_rnorm(rng, 0, 1), z1 = bo_rnorm(rng, 0, 1);
      eta[b->first] = sqrt(b->sigsq[0]) * z0 + 0.0;
      eta[b->first + 1] = sqrt(b->sigsq[1]) * z1 + 0.0;
    } else if (b->kind == BO_BLK_SEASONAL) {
      /* SeasonalStateModel.cpp:124-146: only when the next time point starts a season */
      if (blk_new_season(b, t + 1)) eta[b->first] = bo_rnorm(rng, 0, sqrt(b->sigsq[0]));
    } else if (b->kind == BO_BLK_SEMILOCAL) {
      /* SemilocalLinearTrend.cpp:189-194 */
      eta[b->first] = bo_rnorm(rng, 0, sqrt(b->sigsq[0]));
      eta[b->first + 1] = bo_rnorm(rng, 0, sqrt(b->sigsq[1]));
      eta[b->first + 2] = 0;
    } else if (b->kind == BO_BLK_STATIC_INTERCEPT) {
      /* StaticInterceptStateModel.hpp:52-54: eta[0] = 0.0, the generator is not touched */
    } else if (b->kind == BO_BLK_TRIG) {
      /* TrigStateModel.cpp:218-223: rnorm_mt(rng, 0, sigma) per component, in order */
      const double sigma = sqrt(b->sigsq[0]);
      for (int i = 0; i < b->dim; ++i) eta[b->first + i] = bo_rnorm(rng, 0, sigma);
    } else {
      /* ArStateModel::simulate_state_error, ArStateModel.cpp:85-90: rnorm_mt(rng) * sigma() */
      eta[b->first] = bo_rnorm(rng, 0, 1) * sqrt(b->sigsq[0]);
    }
  }
}

/* Base::impute_state for the structural model */
int bo_ssm_impute_state(bo_ssm *M, bo_rng *rng) {
  const int T = M->T, p = M->p, m = M->m;
  bo_ssvs *reg = M->reg;
  if (M->nblocks == 0) return BO_ERR_INVALID;   /* "No state has been defined." */
  double *xty = (double *)xcalloc(p, sizeof(double));
  double *xsum = (double *)xcalloc(p, sizeof(double));
  double yty = 0, nobs = 0, sumy = 0;
  for (int bi = 0; bi < M->nblocks; ++bi) {
    bo_ssm_block *b = &M->blk[bi];
    for (int i = 0; i < 2; ++i) { b->suf_n[i] = 0; b->suf_ss[i] = 0; }
    b->mv_n = 0;
    b->mv_ybar[0] = b->mv_ybar[1] = 0;
    b->mv_sumsq[0] = b->mv_sumsq[1] = 0;
    b->a1_sumsq = b->a1_sum = b->a1_cross = b->a1_n = b->a1_first = b->a1_last = 0;   /* Ar1Suf::clear */
    if (b->kind == BO_BLK_AR) {
      /* clear_client_data -> ArModel's NeRegSuf::clear */
      for (int i = 0; i < b->lags * b->lags; ++i) b->ar_xtx[i] = 0;
      for (int i = 0; i < b->lags; ++i) b->ar_xty[i] = 0;
      b->ar_yty = 0;
      b->ar_n = 0;
    }
  }

  /* a semilocal trend's initial_state_mean()[2] is slope_->mu(), whatever it is now */
  for (int bi = 0; bi < M->nblocks; ++bi)
    if (M->blk[bi].kind == BO_BLK_SEMILOCAL) M->a0[M->blk[bi].first + 2] = M->blk[bi].sl_mu;
  double a[BO_SSM_MAX];
  double *P = (double *)xcalloc((size_t)m * m, sizeof(double));
  for (int i = 0; i < m; ++i) { a[i] = M->a0[i]; P[IDX(i, i, m)] = M->P0[i]; }
  int status = 0;
  int nvars = 0;
  for (int j = 0; j < p; ++j) nvars += reg->gamma[j];
  for (int t = 0; t < T && !status; ++t) {
    int missing = !M->observed[t];
    double ystar = BO_NEG_INF;
    if (!missing) {
      double pred = 0;
      if (nvars > 0)
        for (int j = 0; j < p; ++j) pred += M->X[IDX(t, j, T)] * reg->beta[j];
      ystar = M->y[t] - pred;
    }
    status = ssm_update(M, t, ystar, missing, ssm_observation_variance(M, t), a, P,
                        &M->v[t], &M->F[t], &M->K[(size_t)t * m]);
  }
  /* simulate_forward (StateSpaceModelBase.cpp:771-790): initial state by
   * rmvn_mt(mean, variance) per state model (StateModel.cpp:47-56; a diagonal
   * variance: mean_i + sd_i z_i, every z drawn, mvn.cpp:54-60, :80-85), state
   * errors per model in order, then the observation */
  double as[BO_SSM_MAX];
  for (int i = 0; i < m * m; ++i) P[i] = 0;
  for (int i = 0; i < m; ++i) { as[i] = M->a0[i]; P[IDX(i, i, m)] = M->P0[i]; }
  for (int t = 0; t < T && !status; ++t) {
    double *st = M->state + (size_t)t * m;
    if (t == 0) {
      for (int bi = 0; bi < M->nblocks; ++bi) {
        const bo_ssm_block *b = &M->blk[bi];
        const int f = b->first;
        if (b->kind == BO_BLK_SEMILOCAL) {
          /* SemilocalLinearTrend.cpp:262-270 */
          st[f] = bo_rnorm(rng, M->a0[f], sqrt(M->P0[f]));
          st[f + 1] = bo_rnorm(rng, M->a0[f + 1], sqrt(M->P0[f + 1]));
          st[f + 2] = b->sl_mu;
        } else if (b->kind == BO_BLK_LOCAL_LEVEL || b->kind == BO_BLK_STATIC_INTERCEPT) {
          /* LocalLevelStateModel::simulate_initial_state, LocalLevelStateModel.cpp:66-69;
           * StaticInterceptStateModel.cpp:39-43 */
          st[f] = bo_rnorm(rng, M->a0[f], sqrt(M->P0[f]));
        } else {
          /* StateModelBase::simulate_initial_state: rmvn_mt(mean, variance), diagonal here */
          double z[BO_SSM_MAX];
          for (int i = 0; i < b->dim; ++i) z[i] = bo_rnorm(rng, 0, 1);
          for (int i = 0; i < b->dim; ++i) st[f + i] = sqrt(M->P0[f + i]) * z[i] + M->a0[f + i];
        }
      }
    } else {
      double eta[BO_SSM_MAX];
      ssm_state_error(M, rng, eta, t - 1);
      const double *prev = M->state + (size_t)(t - 1) * m;
      for (int i = 0; i < m; ++i) st[i] = prev[i];
      ssm_T(M, st, t - 1);
      for (int i = 0; i < m; ++i) st[i] += eta[i];
    }
    const double H = ssm_observation_variance(M, t);
    const double ysim = bo_rnorm(rng, ssm_Zdot(M, st), sqrt(H));
    status = ssm_update(M, t, ysim, !M->observed[t], H, as, P, &M->vs[t], &M->Fs[t],
                        &M->Ks[(size_t)t * m]);
  }
  free(P);
  if (status) { free(xty); free(xsum); return status; }

  /* propagate_disturbances, StateSpaceModelBase.cpp:858-891 */
  double r0[BO_SSM_MAX], r0s[BO_SSM_MAX], mean_sim[BO_SSM_MAX], mean_obs[BO_SSM_MAX],
      rqr[BO_SSM_MAX];
  ssm_disturbance_smooth(M, M->v, M->F, M->K, M->r, r0);
  ssm_disturbance_smooth(M, M->vs, M->Fs, M->Ks, M->rs, r0s);
  for (int i = 0; i < m; ++i) {
    mean_sim[i] = M->a0[i] + M->P0[i] * r0s[i];
    mean_obs[i] = M->a0[i] + M->P0[i] * r0[i];
  }
  for (int t = 0; t < T; ++t) {
    double *st = M->state + (size_t)t * m;
    if (t > 0) {
      ssm_T(M, mean_sim, t - 1);
      ssm_T(M, mean_obs, t - 1);
      ssm_rqr(M, rqr, t - 1);
      for (int i = 0; i < m; ++i) {
        mean_sim[i] += rqr[i] * M->rs[IDX(i, t - 1, m)];
        mean_obs[i] += rqr[i] * M->r[IDX(i, t - 1, m)];
      }
    }
    for (int i = 0; i < m; ++i) st[i] += mean_obs[i] - mean_sim[i];
    if (t == 0) {
      /* observe_initial_state (StateSpaceModelBase::observe_state(0)): only the semilocal trend
       * does anything, SemilocalLinearTrend.cpp:178-180: slope_->suf()->update_raw(state[1]) */
      for (int bi = 0; bi < M->nblocks; ++bi) {
        bo_ssm_block *b = &M->blk[bi];
        if (b->kind == BO_BLK_SEMILOCAL) ar1_update(b, st[b->first + 1]);
      }
    }
    if (t > 0) {
      const double *then = M->state + (size_t)(t - 1) * m;
      for (int bi = 0; bi < M->nblocks; ++bi) {
        bo_ssm_block *b = &M->blk[bi];
        const int f = b->first;
        if (b->kind == BO_BLK_LOCAL_LEVEL) {
          /* LocalLevelStateModel::observe_state, LocalLevelStateModel.cpp:52-58 */
          double diff = st[f] - then[f];
          b->suf_n[0] += 1;
          b->suf_ss[0] += diff * diff;
        } else if (b->kind == BO_BLK_LOCAL_LINEAR_TREND) {
          /* LocalLinearTrendStateModel::observe_state, LocalLinearTrend.cpp:53-63
           * + MvnSuf::update_raw, MvnBase.cpp:71-86 (diagonal of sumsq only) */
          double err[2] = {st[f] - (then[f] + then[f + 1]), st[f + 1] - then[f + 1]};
          b->mv_n += 1.0;
          for (int i = 0; i < 2; ++i) {
            double w = (err[i] - b->mv_ybar[i]) / b->mv_n;
            b->mv_ybar[i] += w;
            b->mv_sumsq[i] += w * w * (b->mv_n - 1);
            double w2 = err[i] - b->mv_ybar[i];
            b->mv_sumsq[i] += w2 * w2 * 1;
          }
        } else if (b->kind == BO_BLK_SEMILOCAL) {
          /* SemilocalLinearTrend.cpp:168-176: the level's error into its GaussianSuf, the
           * current slope into the Ar1Suf */
          double change_in_level = st[f] - then[f] - then[f + 1];
          b->suf_n[0] += 1;
          b->suf_ss[0] += change_in_level * change_in_level;
          ar1_update(b, st[f + 1]);
        } else if (b->kind == BO_BLK_STATIC_INTERCEPT) {
          /* observe_state: "There is nothing to do here." (StaticInterceptStateModel.hpp:45-47) */
        } else if (b->kind == BO_BLK_TRIG) {
          /* TrigStateModel::observe_state, TrigStateModel.cpp:182-193: every component's
           * now - (rotation * then) into the error distribution's GaussianSuf */
          for (int q = 0; q < b->nfreq; ++q) {
            const double c = b->trig_cos[q], sn = b->trig_sin[q];
            const double r0 = c * then[f + 2 * q] + sn * then[f + 2 * q + 1];
            const double r1 = -sn * then[f + 2 * q] + c * then[f + 2 * q + 1];
            const double e0 = st[f + 2 * q] - r0, e1 = st[f + 2 * q + 1] - r1;
            b->suf_n[0] += 1;
            b->suf_ss[0] += e0 * e0;
            b->suf_n[0] += 1;
            b->suf_ss[0] += e1 * e1;
          }
        } else if (b->kind == BO_BLK_SEASONAL) {
          /* SeasonalStateModelBase::observe_state, SeasonalStateModel.cpp:74-86 */
          if (blk_new_season(b, t)) {
            double sum = 0;
            for (int i = 0; i < b->dim; ++i) sum += then[f + i];
            double mu = -1 * sum;
            double delta = st[f] - mu;
            b->suf_n[0] += 1;
            b->suf_ss[0] += delta * delta;
          }
        } else {
          /* ArStateModel::observe_state (ArStateModel.cpp:64-69): suf()->add_mixture_data(
           * now[0], then, 1.0), NeRegSuf::add_mixture_data RegressionModel.cpp:356-370 */
          const int L = b->lags;
          const double yy = st[f], *x = then + f;
          for (int j = 0; j < L; ++j)
            for (int i = 0; i < L; ++i) b->ar_xtx[IDX(i, j, L)] += x[i] * x[j] * 1.0;
          for (int i = 0; i < L; ++i) b->ar_xty[i] += (yy * 1.0) * x[i];
          b->ar_yty += yy * yy * 1.0;
          b->ar_n += 1.0;
        }
      }
    }
    if (M->observed[t]) {
      /* observe_data_given_state, StateSpaceRegressionModel.cpp:188-200 */
      double resid = M->y[t] - ssm_Zdot(M, st);
      for (int j = 0; j < p; ++j) {
        double xj = M->X[IDX(t, j, T)];
        xty[j] += xj * resid;
        xsum[j] += xj;
      }
      yty += resid * resid;
      nobs += 1.0;
      sumy += resid;
    }
  }
  for (int bi = 0; bi < M->nblocks; ++bi) {
    bo_ssm_block *b = &M->blk[bi];
    if (b->kind != BO_BLK_LOCAL_LINEAR_TREND) continue;
    /* ZeroMeanMvnIndependenceSampler::draw reads df = suf->n() and
     * center_sumsq(mu = 0)(i, i) = sumsq_ii + n ybar_i^2 (MvnBase.cpp:157-161) */
    for (int i = 0; i < 2; ++i) {
      b->suf_n[i] = b->mv_n;
      b->suf_ss[i] = b->mv_sumsq[i] + b->mv_ybar[i] * b->mv_ybar[i] * b->mv_n;
    }
  }
  bo_ssvs_set_suf(reg, xty, yty, nobs, sumy, xsum);
  free(xty);
  free(xsum);
  return 0;
}
void bo_ssm_get_suf(const bo_ssm *m, double *n, double *ss) {
  for (int i = 0; i < 3; ++i) {
    int v;
    const bo_ssm_block *B = ssm_template_var(m, i, &v);
    n[i] = (B && v < B->nvar) ? B->suf_n[v] : 0.0;
    ss[i] = (B && v < B->nvar) ? B->suf_ss[v] : 0.0;
  }
}

/* ArModel::check_stationary (Models/TimeSeries/ArModel.cpp:142-170): true if
 * sum |phi| < 1; otherwise the reference finds the roots of 1 - phi_1 z - ... -
 * phi_p z^p with a Jenkins-Traub solver (cpputil/Polynomial.cpp, TOMS 493) and wants
 * them all outside the unit circle.  Restated with the equivalent step-down
 * (Levinson) recursion: all roots lie outside the unit circle iff every partial
 * autocorrelation has modulus < 1.  The two can only disagree when a root sits
 * within rounding of the circle. */
static int ar_check_stationary(int L, const double *phi) {
  double s = 0, a[BO_SSM_MAX], b[BO_SSM_MAX];
  for (int i = 0; i < L; ++i) s += fabs(phi[i]);
  if (s < 1) return 1;
  for (int i = 0; i < L; ++i) a[i] = phi[i];
  for (int k = L; k >= 1; --k) {
    const double r = a[k - 1];
    if (!(fabs(r) < 1)) return 0;
    const double den = 1 - r * r;
    for (int j = 0; j + 1 < k; ++j) b[j] = (a[j] + r * a[k - 2 - j]) / den;
    for (int j = 0; j + 1 < k; ++j) a[j] = b[j];
  }
  return 1;
}
int bo_test_ar_check_stationary(int L, const double *phi) { return ar_check_stationary(L, phi); }

/* Tn2Sampler (distributions/Tn2Sampler.cpp:25-131): adaptive rejection sampling of a
 * standard normal restricted to [lo, hi], logf = -x^2/2, outer hull of tangents at
 * the points x.  Restated as written, including update_cdf's increment
 * (exp(y - y0) / d) * expm1(d * knots[k + 1] - knots[k]). */
typedef struct {
  int n;
  double x[BO_ARS_CAP], logf[BO_ARS_CAP], dlogf[BO_ARS_CAP], knots[BO_ARS_CAP + 1],
      cdf[BO_ARS_CAP];
} bo_tn2;
static void tn2_refresh(bo_tn2 *s) {
  const int n = s->n;
  s->knots[0] = s->x[0];
  s->knots[n] = s->x[n - 1];
  for (int k = 1; k < n; ++k) {   /* compute_knot(k) */
    double y2 = s->logf[k], y1 = s->logf[k - 1], d2 = s->dlogf[k], d1 = s->dlogf[k - 1];
    double x2 = s->x[k], x1 = s->x[k - 1];
    double ans = (y1 - d1 * x1) - (y2 - d2 * x2);
    ans /= (d2 - d1);
    s->knots[k] = ans;
  }
  const double y0 = s->logf[0];
  for (int k = 0; k < n; ++k) {   /* update_cdf */
    double d = s->dlogf[k];
    double y = s->logf[k] + d * (s->knots[k] - s->x[k]);
    double increment;
    if (fabs(d) < .00000000001) {
      increment = exp(y - y0) * (s->knots[k + 1] - s->knots[k]);
    } else {
      increment = (exp(y - y0) / d) * expm1(d * s->knots[k + 1] - s->knots[k]);
    }
    s->cdf[k] = (k == 0 ? increment : s->cdf[k - 1] + increment);
  }
}
static double tn2_draw(bo_rng *r, double lo, double hi, int *status) {
  bo_tn2 s;
  s.n = 2;
  s.x[0] = lo; s.x[1] = hi;
  s.logf[0] = -.5 * lo * lo; s.logf[1] = -.5 * hi * hi;
  s.dlogf[0] = -lo; s.dlogf[1] = -hi;
  tn2_refresh(&s);
  for (int level = 0; level <= 1001; ++level) {
    double u = bo_runif(r, 0, s.cdf[s.n - 1]);
    int k = ars_lower_bound(s.cdf, s.n, u);
    if (k >= s.n) break;   /* (past the end of cdf in the reference) */
    double klo = s.knots[k], khi = s.knots[k + 1];
    double lam = -1 * s.dlogf[k];
    double cand;
    if (lam == 0 || fabs(khi - klo) < sqrt(DBL_EPSILON)) {
      cand = bo_runif(r, klo, khi);
    } else {
      cand = bo_rtrun_exp(r, lam, klo, khi);
    }
    double target = -.5 * cand * cand;
    double logu = (s.logf[k] + s.dlogf[k] * (cand - s.x[k])) - bo_rexp(r, 1);
    if (logu < target) return cand;
    /* add_point: report_error when the candidate is outside [x[0], x.back()] */
    if (cand > s.x[s.n - 1] || cand < s.x[0] || s.n >= BO_ARS_CAP) break;
    int pos = ars_lower_bound(s.x, s.n, cand);
    for (int i = s.n; i > pos; --i) { s.x[i] = s.x[i - 1]; s.logf[i] = s.logf[i - 1]; s.dlogf[i] = s.dlogf[i - 1]; }
    s.x[pos] = cand;
    s.logf[pos] = -.5 * cand * cand;
    s.dlogf[pos] = -cand;
    ++s.n;
    tn2_refresh(&s);
  }
  *status = BO_ERR_UNSUPPORTED_RNG_BRANCH;
  return NAN;
}

/* rtrun_norm_2_mt (distributions/trun_norm.cpp:273-325) for lo, hi finite */
double bo_rtrun_norm_2(bo_rng *rng, double mu, double sigma, double lo, double hi,
                       int *status) {
  if (lo < mu && hi > mu) {
    if ((hi - lo) / sigma > .5) {
      double y = lo - 1;
      while (y < lo || y > hi) y = bo_rnorm(rng, mu, sigma);
      return y;
    } else {
      const double ln_sqrt_2pi = 0.918938533204672741780329736406;
      const double phi_mu = -(ln_sqrt_2pi + 0.5 * 0.0 * 0.0 + log(sigma));   /* dnorm(mu, mu, sigma, true) */
      double phi = phi_mu, u = phi + 1, y = 0;
      while (u > phi) {
        y = bo_runif(rng, lo, hi);
        const double x = (y - mu) / sigma;
        phi = -(ln_sqrt_2pi + 0.5 * x * x + log(sigma));
        u = phi_mu - bo_rexp(rng, 1);
      }
      return y;
    }
  }
  hi = (hi - mu) / sigma;
  lo = (lo - mu) / sigma;
  if (hi < 0) {
    /* rtrun_norm_2_mt(rng, 0, 1, -hi, -lo): not interior, standardising by (0, 1) changes nothing */
    double y = tn2_draw(rng, -hi, -lo, status);
    return mu - sigma * y;
  }
  double y = tn2_draw(rng, lo, hi, status);
  return y * sigma + mu;
}
#define rtrun_norm_2 bo_rtrun_norm_2

/* ArPosteriorSampler::draw_phi / draw_phi_univariate / draw_sigma,
 * Models/TimeSeries/PosteriorSamplers/ArPosteriorSampler.cpp:91-143, :78-89 */
static void ar_draw(bo_ssm *M, bo_ssm_block *B, int *status) {
  const int L = B->lags;
  double phi_hat[BO_AR_MAX], P[BO_AR_MAX * BO_AR_MAX], Lc[BO_AR_MAX * BO_AR_MAX],
      z[BO_AR_MAX];
  bo_rng *rng = &B->rng[0];
  if (!bo_spd_solve(L, B->ar_xtx, B->ar_xty, phi_hat)) { *status = BO_ERR_NOT_PD; return; }
  int ok = 0, attempts = 0;
  while (!ok && ++attempts <= 3) {
    /* rmvn_ivar(phi_hat, xtx / sigsq), mvn.cpp:99-122: on GlobalRng::rng */
    bo_rng *grng = M->ar_global_rng ? M->ar_global_rng : rng;
    for (int i = 0; i < L * L; ++i) P[i] = B->ar_xtx[i] / B->sigsq[0];
    if (!bo_chol(L, P, Lc)) { *status = BO_ERR_NOT_PD; return; }
    for (int i = 0; i < L; ++i) z[i] = bo_rnorm(grng, 0, 1);
    ltsolve_inplace(L, Lc, z);
    for (int i = 0; i < L; ++i) z[i] = z[i] + phi_hat[i];
    ok = ar_check_stationary(L, z);
    if (ok) for (int i = 0; i < L; ++i) B->phi[i] = z[i];
  }
  if (!ok) {
    double phi[BO_AR_MAX] = {0};
    for (int i = 0; i < L; ++i) phi[i] = B->phi[i];
    if (!ar_check_stationary(L, phi)) { *status = BO_ERR_INVALID; return; }
    for (int i = 0; i < L; ++i) {
      const double initial_phi = phi[i];
      double lo = -1, hi = 1;
      const double ivar = B->ar_xtx[IDX(i, i, L)];
      double dot = 0;
      for (int j = 0; j < L; ++j) dot += phi[j] * B->ar_xtx[IDX(j, i, L)];
      const double mu = (B->ar_xty[i] - (dot - phi[i] * B->ar_xtx[IDX(i, i, L)])) / ivar;
      for (;;) {
        const double candidate = rtrun_norm_2(rng, mu, sqrt(1.0 / ivar), lo, hi, status);
        if (*status) return;
        phi[i] = candidate;
        if (ar_check_stationary(L, phi)) break;
        if (candidate > initial_phi) hi = candidate; else lo = candidate;
      }
    }
    for (int i = 0; i < L; ++i) B->phi[i] = phi[i];
  }
  /* draw_sigma: ss = phi' xtx phi - 2 phi' xty + yty, df = n */
  double quad = 0, lin = 0;
  for (int i = 0; i < L; ++i) {
    double row = 0;
    for (int j = 0; j < L; ++j) row += B->ar_xtx[IDX(i, j, L)] * B->phi[j];
    quad += B->phi[i] * row;
    lin += B->phi[i] * B->ar_xty[i];
  }
  const double ss = quad - 2 * lin + B->ar_yty;
  B->sigsq[0] = variance_draw(rng, B->prior_df[0], B->prior_ss[0], B->sigma_max[0], B->ar_n, ss,
                              status);
}

/* NonzeroMeanAr1Sampler::draw (NonzeroMeanAr1Sampler.cpp:51-155) on the slope's Ar1Suf
 * (NonzeroMeanAr1Model.cpp:103-120): draw_mu, draw_phi, draw_sigma, one generator */
static void semilocal_slope_draw(bo_ssm_block *b, int *status) {
  bo_rng *rng = &b->rng[1];
  const double n = b->a1_n;
  const double lag_sumsq = b->a1_sumsq - pow(b->a1_last, 2);
  const double lag_sum = b->a1_sum - b->a1_last;
  const double sum_excluding_first = b->a1_sum - b->a1_first;
  const double sumsq_excluding_first = b->a1_sumsq - pow(b->a1_first, 2);
  {  /* draw_mu */
    const double phi = b->sl_phi, sigsq = b->sigsq[1];
    const double prior_sigsq = b->sl_mean_prior[1] * b->sl_mean_prior[1];
    double ivar = (1 + (n - 1) * pow(1 - phi, 2)) / sigsq;
    ivar += 1.0 / prior_sigsq;
    double mean = (1 - phi) * (sum_excluding_first - phi * lag_sum) + b->a1_first;
    mean /= sigsq;
    mean += b->sl_mean_prior[0] / prior_sigsq;
    mean /= ivar;
    const double sd = sqrt(1.0 / ivar);
    b->sl_mu = bo_rnorm(rng, mean, sd);
  }
  {  /* draw_phi */
    const double mu = b->sl_mu, sigsq = b->sigsq[1];
    const double prior_sigsq = b->sl_phi_prior[1] * b->sl_phi_prior[1];
    double ivar = lag_sumsq - 2 * lag_sum * mu + (n - 1) * mu * mu;   /* centered_lag_sumsq(mu) */
    ivar /= sigsq;
    ivar += 1.0 / prior_sigsq;
    /* centered_cross(mu), NonzeroMeanAr1Model.cpp */
    double mean = b->a1_cross - mu * (sum_excluding_first + lag_sum) + (n - 1) * mu * mu;
    mean /= sigsq;
    mean += b->sl_phi_prior[0] / prior_sigsq;
    mean /= ivar;
    const double sd = sqrt(1.0 / ivar);
    double phi;
    if (b->sl_truncate_phi) {
      const double lower_limit = b->sl_force_positive ? 0 : -1;
      phi = bo_rtrun_norm_2(rng, mean, sd, lower_limit, 1, status);
      if (*status) return;
    } else {
      phi = bo_rnorm(rng, mean, sd);
    }
    b->sl_phi = phi;
  }
  {  /* draw_sigma: sigsq_sampler_.draw(rng, suf->n(), suf->model_sumsq(mu, phi)) */
    const double mu = b->sl_mu, phi = b->sl_phi;
    double ss = pow(b->a1_first - mu, 2);
    ss += sumsq_excluding_first - 2 * phi * b->a1_cross -
          2 * (1 - phi) * mu * sum_excluding_first + phi * phi * lag_sumsq +
          2 * phi * (1 - phi) * mu * lag_sum +
          (n - 1) * pow(mu * (1 - phi), 2);
    b->sigsq[1] = variance_draw(rng, b->prior_df[1], b->prior_ss[1], b->sigma_max[1], n, ss, status);
  }
}

/* StateSpacePosteriorSampler::draw, StateSpacePosteriorSampler.cpp:42-64: the
 * regression, then each state model's samplers in the order the models were added
 * (local level: its variance; local linear trend: level, slope; seasonal: its
 * variance; autoregression: phi, sigma), then impute_state */
int bo_ssm_draw(bo_ssm *m) {
  int status = 0;
  if (!m->latent_initialized) {
    status = bo_ssm_impute_state(m, &m->state_rng);
    if (status) return status;
    m->latent_initialized = 1;
  }
  status = ssvs_draw(m->reg);
  if (status) return status;
  for (int bi = 0; bi < m->nblocks; ++bi) {
    bo_ssm_block *b = &m->blk[bi];
    if (b->kind == BO_BLK_AR) {
      ar_draw(m, b, &status);
      if (status) return status;
      continue;
    }
    if (b->kind == BO_BLK_SEMILOCAL) {
      /* the trend's two samplers in the order bsts attaches them
       * (create_state_model.cpp:624-672): the level's variance, then the slope model */
      b->sigsq[0] = variance_draw(&b->rng[0], b->prior_df[0], b->prior_ss[0], b->sigma_max[0],
                                  b->suf_n[0], b->suf_ss[0], &status);
      if (status) return status;
      semilocal_slope_draw(b, &status);
      if (status) return status;
      continue;
    }
    for (int i = 0; i < b->nvar; ++i) {
      double draw = variance_draw(&b->rng[i], b->prior_df[i], b->prior_ss[i],
                                  b->sigma_max[i], b->suf_n[i], b->suf_ss[i], &status);
      if (status) return status;
      if (b->kind == BO_BLK_LOCAL_LINEAR_TREND) {
        /* ZeroMeanMvnIndependenceSampler.cpp:63-70: siginv(i, i) = 1 / draw, and
         * the model's Sigma is the inverse of that again */
        double siginv = 1.0 / draw;
        draw = 1.0 / siginv;
      }
      b->sigsq[i] = draw;
    }
  }
  return bo_ssm_impute_state(m, &m->state_rng);
}

/* StateSpaceRegressionModel::simulate_forecast for the structural model
 * (StateSpaceRegressionModel.cpp:216-219, :256-278; simulate_next_state
 * StateSpaceModelBase.cpp:439-443) from a model object that holds the blocks, their
 * variances and coefficients: newX horizon x p column-major, final_state the state at
 * the last time point T - 1 (m values); forecast step i is time T + i */
void bo_ssm_forecast_model(const bo_ssm *M, bo_rng *rng, int horizon, int p, const double *newX,
                           const double *beta, double sigsq_obs, const double *final_state,
                           double *out) {
  double st[BO_SSM_MAX];
  for (int i = 0; i < M->m; ++i) st[i] = final_state[i];
  const double sd_obs = sqrt(sigsq_obs);
  for (int i = 0; i < horizon; ++i) {
    double eta[BO_SSM_MAX];
    /* advance_to_timestamp (StateSpaceModelBase.cpp:455-459) calls simulate_next_state(rng,
     * state, time_dimension() + time++) with time starting at -1: forecast step i is
     * simulated as "time period T - 1 + i", i.e. with the transition matrix and the state
     * errors of index T - 2 + i (it matters for a seasonal model with duration > 1 only) */
    const int tm = M->T - 2 + i;
    ssm_state_error(M, rng, eta, tm);
    ssm_T(M, st, tm);
    for (int k = 0; k < M->m; ++k) st[k] += eta[k];
    double ans = bo_rnorm(rng, ssm_Zdot(M, st), sd_obs);
    double pred = 0;
    for (int j = 0; j < p; ++j) pred += newX[IDX(i, j, horizon)] * beta[j];
    out[i] = ans + pred;
  }
}
/* the template's form: sigsq = (level, slope, seasonal) */
void bo_ssm_simulate_forecast_ar(bo_rng *rng, int horizon, int p, const double *newX,
                                 const double *beta, double sigsq_obs, int trend, int nseasons,
                                 const double *sigsq, int ar_lags, const double *phi,
                                 double ar_sigsq, const double *final_state, double *out) {
  bo_ssm M;
  memset(&M, 0, sizeof(M));
  bo_ssm_block *b = &M.blk[M.nblocks++];
  b->kind = trend == 2 ? BO_BLK_LOCAL_LINEAR_TREND : BO_BLK_LOCAL_LEVEL;
  b->first = 0; b->dim = trend; b->nvar = trend; b->duration = 1;
  b->sigsq[0] = sigsq[0]; b->sigsq[1] = sigsq[1];
  M.m = trend;
  if (nseasons > 0) {
    b = &M.blk[M.nblocks++];
    b->kind = BO_BLK_SEASONAL; b->first = M.m; b->dim = nseasons - 1; b->nvar = 1;
    b->nseasons = nseasons; b->duration = 1; b->sigsq[0] = sigsq[2];
    M.m += b->dim;
  }
  if (ar_lags > 0) {
    b = &M.blk[M.nblocks++];
    b->kind = BO_BLK_AR; b->first = M.m; b->dim = ar_lags; b->lags = ar_lags; b->nvar = 1;
    b->duration = 1; b->sigsq[0] = ar_sigsq;
    for (int i = 0; i < ar_lags; ++i) b->phi[i] = phi[i];
    M.m += ar_lags;
  }
  /* (observation_matrix: one at every block's first component) */
  for (int k = 0; k < M.nblocks; ++k) M.zpos[M.nz++] = M.blk[k].first;
  bo_ssm_forecast_model(&M, rng, horizon, p, newX, beta, sigsq_obs, final_state, out);
}
void bo_ssm_simulate_forecast(bo_rng *rng, int horizon, int p, const double *newX,
                              const double *beta, double sigsq_obs, int trend, int nseasons,
                              const double *sigsq, const double *final_state, double *out) {
  bo_ssm_simulate_forecast_ar(rng, horizon, p, newX, beta, sigsq_obs, trend, nseasons, sigsq, 0,
                              NULL, 0.0, final_state, out);
}

/* ====================================================================== *
 * BinomialProbitSpikeSlabSampler (the probit member):
 * data augmentation with truncated normals, then SpikeSlabSampler (bo_sss) on
 * the complete-data sufficient statistics X'NX (fixed) and X'z.
 * Models/Glm/PosteriorSamplers/BinomialProbitSpikeSlabSampler.cpp:40-85,
 * BinomialProbitDataImputer.cpp:30-73.
 * ====================================================================== */

/* TnSampler (distributions/trun_norm.cpp:108-228): the bounded adaptive
 * rejection sampler for a standard normal restricted to x > a (a > 0), logf =
 * -x^2/2.  Same structure as ars_draw above; the acceptance test is strict. */
static double tn_draw(bo_rng *r, double a, int *status) {
  bo_ars s;
  s.n = 1;
  s.x[0] = a;
  s.y[0] = -.5 * a * a;
  s.d[0] = -a;
  s.knots[0] = a;
  ars_update_cdf(&s);
  for (int level = 0; level <= 1001; ++level) {
    double u = bo_runif(r, 0, s.cdf[s.n - 1]);
    int k = ars_lower_bound(s.cdf, s.n, u);
    double cand;
    if (k + 1 == s.n) {
      cand = s.knots[s.n - 1] + bo_rexp(r, -1 * s.d[s.n - 1]);
    } else {
      cand = bo_rtrun_exp(r, -1 * s.d[k], s.knots[k], s.knots[k + 1]);
    }
    double target = -.5 * cand * cand;
    double hull = s.y[k] + s.d[k] * (cand - s.x[k]);
    double logu = hull - bo_rexp(r, 1);
    if (logu < target) return cand;
    if (s.n >= BO_ARS_CAP) { *status = BO_ERR_UNSUPPORTED_RNG_BRANCH; return NAN; }
    int pos = ars_lower_bound(s.knots, s.n, cand);
    for (int i = s.n; i > pos; --i) { s.x[i] = s.x[i - 1]; s.y[i] = s.y[i - 1]; s.d[i] = s.d[i - 1]; }
    s.x[pos] = cand;
    s.y[pos] = -.5 * cand * cand;
    s.d[pos] = -cand;
    ++s.n;
    ars_refresh(&s);
    ars_update_cdf(&s);
  }
  *status = BO_ERR_UNSUPPORTED_RNG_BRANCH;
  return NAN;
}
/* trun_norm_mt(rng, a): standard normal given x > a (trun_norm.cpp:231-241) */
static double trun_norm_std(bo_rng *r, double a, int *status) {
  if (a <= 0) {
    for (;;) {
      double x = bo_rnorm(r, 0, 1);
      if (x > a) return x;
    }
  }
  return tn_draw(r, a, status);
}
/* rtrun_norm_mt(rng, mu, sigma, a, gt), trun_norm.cpp:36-47 */
double bo_rtrun_norm(bo_rng *r, double mu, double sigma, double a, int gt, int *status) {
  if (gt) return mu + sigma * trun_norm_std(r, (a - mu) / sigma, status);
  return mu - sigma * trun_norm_std(r, (mu - a) / sigma, status);
}
/* log of the standard normal cdf (lower or upper tail) */
static double log_pnorm_std(double x, int lower) {
  double z = lower ? -x : x;
  return log(0.5 * erfc(z / 1.4142135623730951));
}
/* trun_norm_moments(mu, sigma, cutpoint, positive_support, ...), trun_norm.cpp:243-269 */
static void trun_norm_moments(double mu, double sigma, double cut, int positive,
                              double *mean, double *variance) {
  const double sigsq = sigma * sigma;
  const double log_phi_const = -0.918938533204672741780329736406; /* -log sqrt(2 pi) */
  if (positive) {
    double alpha = (cut - mu) / sigma;
    double phi_ratio = exp((log_phi_const - .5 * alpha * alpha) - log_pnorm_std(alpha, 0));
    *mean = mu + sigma * phi_ratio;
    double delta = phi_ratio * (phi_ratio - alpha);
    *variance = sigsq * (1 - delta);
  } else {
    double beta = (cut - mu) / sigma;
    double phi_ratio = exp((log_phi_const - .5 * beta * beta) - log_pnorm_std(beta, 1));
    *mean = mu - sigma * phi_ratio;
    *variance = sigsq * (1 - beta * phi_ratio - phi_ratio * phi_ratio);
  }
  if (*variance < 0) *variance = 0;
}
/* BinomialProbitDataImputer::impute, BinomialProbitDataImputer.cpp:30-73 */
static double probit_impute(bo_rng *r, int clt, double ntrials, double nsuccess,
                            double eta, int *status) {
  long n = lround(ntrials), y = lround(nsuccess);
  double mean, variance, ans = 0;
  if (y > clt) {
    trun_norm_moments(eta, 1, 0, 1, &mean, &variance);
    ans += bo_rnorm(r, y * mean, sqrt(y * variance));
  } else {
    for (long i = 0; i < y; ++i) ans += bo_rtrun_norm(r, eta, 1, 0, 1, status);
  }
  if (n - y > clt) {
    trun_norm_moments(eta, 1, 0, 0, &mean, &variance);
    ans += bo_rnorm(r, (n - y) * mean, sqrt((n - y) * variance));
  } else {
    for (long i = 0; i < n - y; ++i) ans += bo_rtrun_norm(r, eta, 1, 0, 0, status);
  }
  return ans;
}

struct bo_probit {
  int n, p, clt;
  double *X, *y, *nt; /* X n x p column-major */
  bo_sss *sss;        /* holds X'NX, gamma, beta and the sampler's RNG */
  /* substream = 0: the imputation reads the sampler's own RNG in sequence (the
   * reference; MT engine).  substream = 1 (the device's convention, Philox):
   * observation i of sweep s reads stream 8 from position (s n + i) 256 */
  int substream;
  bo_rng imp_rng;
  uint64_t sweep;
};
#define BO_PROBIT_STRIDE 4096

bo_probit *bo_probit_create(int n, int p, const double *X, const double *y,
                            const double *ntrials, const double *mu, const double *prec,
                            const double *pi, int clt_threshold) {
  bo_probit *m = (bo_probit *)xcalloc(1, sizeof(bo_probit));
  m->n = n; m->p = p; m->clt = clt_threshold;
  m->X = (double *)xcalloc((size_t)n * p, sizeof(double));
  m->y = (double *)xcalloc(n, sizeof(double));
  m->nt = (double *)xcalloc(n, sizeof(double));
  memcpy(m->X, X, sizeof(double) * (size_t)n * p);
  memcpy(m->y, y, sizeof(double) * n);
  memcpy(m->nt, ntrials, sizeof(double) * n);
  /* refresh_xtx: sum_i n_i x_i x_i' (BinomialProbitSpikeSlabSampler.cpp:71-77) */
  double *xtx = (double *)xcalloc((size_t)p * p, sizeof(double));
  double *xty = (double *)xcalloc(p, sizeof(double));
  for (int i = 0; i < n; ++i)
    for (int b = 0; b < p; ++b)
      for (int a = 0; a < p; ++a)
        xtx[IDX(a, b, p)] += X[IDX(i, a, n)] * X[IDX(i, b, n)] * ntrials[i];
  m->sss = bo_sss_create(p, xtx, xty, 0, mu, prec, pi);
  free(xtx); free(xty);
  bo_rng_seed_philox(&m->imp_rng, 0, 0, 8, 0);
  return m;
}
void bo_probit_destroy(bo_probit *m) {
  if (!m) return;
  bo_sss_destroy(m->sss);
  free(m->X); free(m->y); free(m->nt);
  free(m);
}
bo_sss *bo_probit_sss(bo_probit *m) { return m->sss; }
bo_rng *bo_probit_imputer_rng(bo_probit *m) { return &m->imp_rng; }
void bo_probit_use_substreams(bo_probit *m, int on) { m->substream = on; }

/* BinomialProbitSpikeSlabSampler::draw */
int bo_probit_draw(bo_probit *m) {
  const int n = m->n, p = m->p;
  bo_sss *s = m->sss;
  int status = 0;
  double *xtz = s->xty;
  for (int j = 0; j < p; ++j) xtz[j] = 0.0;
  for (int i = 0; i < n; ++i) {
    double eta = 0;
    for (int j = 0; j < p; ++j)
      if (s->gamma[j]) eta += m->X[IDX(i, j, n)] * s->beta[j];
    bo_rng *r = &s->rng;
    if (m->substream) {
      r = &m->imp_rng;
      bo_rng_slot(r, m->sweep * (uint64_t)n + (uint64_t)i, BO_PROBIT_STRIDE);
    }
    double sum_of_z = probit_impute(r, m->clt, m->nt[i], m->y[i], eta, &status);
    if (status) return status;
    for (int j = 0; j < p; ++j) xtz[j] += m->X[IDX(i, j, n)] * sum_of_z;
  }
  ++m->sweep;
  status = bo_sss_draw_model_indicators(s, 1.0);
  if (status) return status;
  return bo_sss_draw_beta(s, 1.0);
}


/* ====================================================================== *
 * BinomialLogitSpikeSlabSampler (the logit member; BASELINE
 * config 5 with the reference's own auxiliary-mixture imputer in place of the
 * Polya-Gamma one it does not have):
 *   BinomialLogitAuxmixSampler::impute_latent_data  (BinomialLogitAuxmixSampler.cpp:77-97)
 *   BinomialLogitCltDataImputer::impute_small_sample (BinomialLogitDataImputer.cpp:128-144)
 *   rtrun_logit_mt (distributions/trun_logit.cpp:163-174)
 *   NormalMixtureApproximation::unmix (NormalMixtureApproximation.cpp:280-290)
 *   BinomialLogitSpikeSlabSampler::draw (BinomialLogitSpikeSlabSampler.cpp:50-79,
 *   :87-117, :178-226)
 * Observations with more than clt_threshold trials take the reference's
 * large-sample branch (BinomialLogitCltDataImputer::impute_large_sample,
 * BinomialLogitDataImputer.cpp:155-211): the counts of failures / successes per
 * mixture component by two multinomial draws (Bmath/rmultinom.cpp:82-136 over
 * BOOM::binomial_distribution, distributions/BinomialDistribution.cpp: Kachitvichyanukul
 * and Schmeiser's BTPE for n p >= 30, inversion below), then ONE normal draw for the
 * information-weighted sum with the truncated-normal moments of every cell.
 * ====================================================================== */
static const double LOGIT_MIX_SIGMA[9] = {0.88437229872213, 1.16097607474416, 1.28021991084306,
                                          1.3592552924727,  1.67589879794907, 2.20287232043947,
                                          2.20507148325819, 2.91944313615144, 3.90807611741308};
static const double LOGIT_MIX_WEIGHT[9] = {0.038483985581272, 0.13389889791451,  0.0657842076622429,
                                           0.105680086433879, 0.345939491553619, 0.0442261124345564,
                                           0.193289780660134, 0.068173066865908, 0.00452437089387876};

/* BOOM::binomial_distribution(n, p)(rng) (distributions/BinomialDistribution.cpp:24-158),
 * what Rmath::rbinom_mt calls (Bmath/rbinom.cpp:66-69): BTPE for n p >= 30, sequential
 * inversion below.  Same statements in the same order, so the same uniforms are consumed. */
unsigned bo_rbinom(bo_rng *rng, unsigned n, double pp) {
  double c = 0, fm = 0, npq = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, qn = 0;
  double xl = 0, xll = 0, xlr = 0, xm = 0, xr = 0;
  int m = 0, ix = 0;
  double p = (pp < 1. - pp) ? pp : 1. - pp;
  double q = 1. - p;
  double np = n * p;
  double r = p / q;
  double g = r * (n + 1);
  const double psave = pp;
  if (np < 30) {
    qn = pow(q, (double)n);
    /* draw_np_small */
    for (;;) {
      ix = 0;
      double f = qn;
      double u = bo_unif(rng);
      for (;;) {
        if (u < f) goto finis;
        if (ix > 110) break;
        u -= f;
        ix++;
        f *= (g / ix - r);
      }
    }
  } else {
    double ffm = np + p, al;
    m = (int)ffm;
    fm = m;
    npq = np * q;
    p1 = (int)(2.195 * sqrt(npq) - 4.6 * q) + 0.5;
    xm = fm + 0.5;
    xl = xm - p1;
    xr = xm + p1;
    c = 0.134 + 20.5 / (15.3 + fm);
    al = (ffm - xl) / (ffm - xl * p);
    xll = al * (1.0 + 0.5 * al);
    al = (xr - ffm) / (xr * q);
    xlr = al * (1.0 + 0.5 * al);
    p2 = p1 * (1.0 + c + c);
    p3 = p2 + c / xll;
    p4 = p3 + c / xlr;
  }
  for (;;) {
    double u = bo_unif(rng) * p4;
    double v = bo_unif(rng);
    double x;
    int k;
    if (u <= p1) { /* triangular region */
      ix = (int)(xm - p1 * v + u);
      goto finis;
    }
    if (u <= p2) { /* parallelogram region */
      x = xl + (u - p1) / c;
      v = v * c + 1.0 - fabs(xm - x) / p1;
      if (v > 1.0 || v <= 0.) continue;
      ix = (int)x;
    } else {
      if (u > p3) { /* right tail */
        ix = (int)(xr - log(v) / xlr);
        if ((unsigned)ix > n) continue;
        v = v * (u - p3) * xlr;
      } else { /* left tail */
        ix = (int)(xl + log(v) / xll);
        if (ix < 0) continue;
        v = v * (u - p2) * xll;
      }
    }
    k = abs(ix - m);
    if (k <= 20 || k >= npq / 2 - 1) {
      double f = 1.0;
      if (m < ix) {
        for (int i = m + 1; i <= ix; i++) f *= (g / i - r);
      } else if (m != ix) {
        for (int i = ix + 1; i <= m; i++) f /= (g / i - r);
      }
      if (v <= f) goto finis;
    } else {
      double amaxp = (k / npq) * ((k * (k / 3. + 0.625) + 0.1666666666666) / npq + 0.5);
      double ynorm = -1.0 * k * k / (2.0 * npq);
      double alv = log(v);
      if (alv < ynorm - amaxp) goto finis;
      if (alv <= ynorm + amaxp) {
        double x1 = ix + 1, f1 = fm + 1.0, z = n + 1 - fm, w = n - ix + 1.0;
        double z2 = z * z, x2 = x1 * x1, f2 = f1 * f1, w2 = w * w;
        if (alv <= xm * log(f1 / x1) + (n - m + 0.5) * log(z / w) + (ix - m) * log(w * p / (x1 * q)) +
                       (13860.0 - (462.0 - (132.0 - (99.0 - 140.0 / f2) / f2) / f2) / f2) / f1 / 166320.0 +
                       (13860.0 - (462.0 - (132.0 - (99.0 - 140.0 / z2) / z2) / z2) / z2) / z / 166320.0 +
                       (13860.0 - (462.0 - (132.0 - (99.0 - 140.0 / x2) / x2) / x2) / x2) / x1 / 166320.0 +
                       (13860.0 - (462.0 - (132.0 - (99.0 - 140.0 / w2) / w2) / w2) / w2) / w / 166320.)
          goto finis;
      }
    }
  }
finis:
  if (psave > 0.5) ix = (int)n - ix;
  return (unsigned)ix;
}

/* Rmath::rmultinom_mt(rng, n, prob, rN), Bmath/rmultinom.cpp:82-136 (prob sums to 1) */
static int bo_rmultinom(bo_rng *rng, int n, const double *prob, int K, int *rN) {
  double p_tot = 0.;
  for (int k = 0; k < K; ++k) {
    const double pp = prob[k];
    if (!isfinite(pp) || pp < 0. || pp > 1.) return BO_ERR_UNSUPPORTED_RNG_BRANCH;
    p_tot += pp;
    rN[k] = 0;
  }
  if (fabs(p_tot - 1.) > 1e-7) return BO_ERR_UNSUPPORTED_RNG_BRANCH;
  if (n == 0) return 0;
  for (int k = 0; k < K - 1; ++k) {
    const double pp = prob[k] / p_tot;
    /* Rmath::rbinom_mt -> binomial_distribution: n = 0, p = 0 and p = 1 go through the
     * same code (inversion with qn = 1, resp. 0): one uniform either way */
    rN[k] = (int)bo_rbinom(rng, (unsigned)n, pp);
    n -= rN[k];
    if (n <= 0) return 0;
    p_tot -= prob[k];
  }
  rN[K - 1] = n;
  return 0;
}
unsigned bo_test_rbinom(bo_rng *r, unsigned n, double p) { return bo_rbinom(r, n, p); }

/* BinomialLogitCltDataImputer::impute_large_sample, BinomialLogitDataImputer.cpp:155-211 */
static int logit_impute_large(bo_rng *rng, double ntrials, double nsuccess, double eta,
                              double *sum_out, double *info_out) {
  double information = 0.0;
  double p0[9], p1[9];
  int N0[9], N1[9];
  /* plogis(0, eta, 1, lower / upper), Bmath/plogis.cpp:43-63 */
  const double xz = (0 - eta) / 1.0;
  const double neg_support = 1 / (1 + exp(-xz)), pos_support = 1 / (1 + exp(xz));
  double s0 = 0, s1 = 0;
  for (int m = 0; m < 9; ++m) {
    const double z = (0 - eta) / LOGIT_MIX_SIGMA[m];
    p0[m] = LOGIT_MIX_WEIGHT[m] / neg_support * (0.5 * erfc(-z / 1.4142135623730951));  /* pnorm(0, eta, sigma, true) */
    p1[m] = LOGIT_MIX_WEIGHT[m] / pos_support * (0.5 * erfc(z / 1.4142135623730951));   /* ... upper tail */
  }
  for (int m = 0; m < 9; ++m) { s0 += p0[m]; s1 += p1[m]; }
  for (int m = 0; m < 9; ++m) { p0[m] /= s0; p1[m] /= s1; }
  int rc = bo_rmultinom(rng, (int)(ntrials - nsuccess), p0, 9, N0);
  if (rc) return rc;
  rc = bo_rmultinom(rng, (int)nsuccess, p1, 9, N1);
  if (rc) return rc;
  double simulation_mean = 0, simulation_variance = 0;
  for (int m = 0; m < 9; ++m) {
    const int total_obs = N0[m] + N1[m];
    if (total_obs == 0) continue;
    const double sigsq = LOGIT_MIX_SIGMA[m] * LOGIT_MIX_SIGMA[m], sig4 = sigsq * sigsq;
    information += total_obs / sigsq;
    double tmean, tvar;
    if (N0[m] > 0) {
      trun_norm_moments(eta, LOGIT_MIX_SIGMA[m], 0, 0, &tmean, &tvar);
      simulation_mean += N0[m] * tmean / sigsq;
      simulation_variance += N0[m] * tvar / sig4;
    }
    if (N1[m] > 0) {
      trun_norm_moments(eta, LOGIT_MIX_SIGMA[m], 0, 1, &tmean, &tvar);
      simulation_mean += N1[m] * tmean / sigsq;
      simulation_variance += N1[m] * tvar / sig4;
    }
  }
  *sum_out = bo_rnorm(rng, simulation_mean, sqrt(simulation_variance));
  *info_out = information;
  return 0;
}


/* ---- Polya-Gamma augmentation (BASELINE config 5 as worded) ---------------------------
 * NO REFERENCE: BOOM has no Polya-Gamma sampler.  What is restated here
 * is the published algorithm -- Polson, Scott and Windle (2013), "Bayesian inference for
 * logistic models using Polya-Gamma latent variables", JASA 108, sec. 4 and the
 * supplement's Algorithm 1: PG(1, z) = J*(1, z / 2) / 4 by Devroye's alternating-series
 * method, the proposal a truncated inverse-Gaussian below t = 0.64 and an exponential
 * above it -- so that the device kernel has a CPU twin to be compared with draw for draw.
 * PARITY UNPINNED: the posterior it leads to is checked against the (reference-pinned)
 * auxiliary-mixture sampler's distributionally (tests/test_polya_gamma.py).
 *   omega_i ~ PG(n_i, x_i'beta) = sum of n_i PG(1, .) draws; beyond clt_threshold trials
 *   a normal draw with the exact mean n tanh(z/2) / (2 z) and variance
 *   n (sinh z - z) / (4 z^3 cosh^2(z/2)) (the moments of PG(n, z)) -- the same switch to a
 *   central-limit draw the reference's own imputers make at that threshold.
 * Given omega: y_i - n_i / 2 = kappa_i is the information-weighted response and omega_i
 * the information, exactly the (sum, information) pair the auxiliary-mixture imputer
 * hands to SufficientStatistics::update (BinomialLogitAuxmixSampler.cpp:60-66). */
#define BO_PG_TRUNC 0.64
static double pg_pnorm(double x) { return 0.5 * erfc(-x / 1.4142135623730951); }
/* series coefficient a_n(x) of the Jacobi density */
static double pg_a(int n, double x) {
  const double K = (n + 0.5) * 3.14159265358979323846;
  if (x > BO_PG_TRUNC) return K * exp(-0.5 * K * K * x);
  const double expnt = -1.5 * (log(0.5 * 3.14159265358979323846) + log(x)) + log(K) - 2.0 * (n + 0.5) * (n + 0.5) / x;
  return exp(expnt);
}
/* inverse Gaussian IG(1 / z, 1) truncated to (0, t) */
static double pg_rtigauss(bo_rng *r, double z, int *status) {
  const double t = BO_PG_TRUNC;
  double X = t + 1.0;
  if (z < 1.0 / t) {   /* mu = 1 / z > t (z = 0: the Levy limit, alpha = 1) */
    double alpha = 0.0;
    int it = 0;
    while (bo_unif(r) > alpha) {
      double E1 = bo_exp_rand(r), E2 = bo_exp_rand(r);
      while (E1 * E1 > 2 * E2 / t) { E1 = bo_exp_rand(r); E2 = bo_exp_rand(r); }
      X = 1 + E1 * t;
      X = t / (X * X);
      alpha = exp(-0.5 * z * z * X);
      if (++it > 10000) { *status = BO_ERR_UNSUPPORTED_RNG_BRANCH; return t; }
    }
  } else {
    const double mu = 1.0 / z;
    int it = 0;
    while (X > t) {
      double Y = bo_norm_rand(r);
      Y *= Y;
      const double half_mu = 0.5 * mu, mu_Y = mu * Y;
      X = mu + half_mu * mu_Y - half_mu * sqrt(4 * mu_Y + mu_Y * mu_Y);
      if (bo_unif(r) > mu / (mu + X)) X = mu * mu / X;
      if (++it > 10000) { *status = BO_ERR_UNSUPPORTED_RNG_BRANCH; return t; }
    }
  }
  return X;
}
/* PG(1, z) */
static double pg_draw1(bo_rng *r, double z, int *status) {
  z = fabs(z) * 0.5;
  const double t = BO_PG_TRUNC;
  const double fz = 0.125 * 3.14159265358979323846 * 3.14159265358979323846 + 0.5 * z * z;
  for (int tries = 0; tries < 10000; ++tries) {
    double X;
    {
      /* p = (pi / 2 / fz) exp(-fz t): mass of the exponential tail; q = 2 exp(-z) P(IG(1/z, 1) < t) */
      const double b = sqrt(1.0 / t) * (t * z - 1), a = -1.0 * sqrt(1.0 / t) * (t * z + 1);
      const double x0 = log(fz) + fz * t;
      const double xb = x0 - z + log(pg_pnorm(b)), xa = x0 + z + log(pg_pnorm(a));
      const double qdivp = 4 / 3.14159265358979323846 * (exp(xb) + exp(xa));
      if (bo_unif(r) < 1.0 / (1.0 + qdivp)) X = t + bo_exp_rand(r) / fz;
      else X = pg_rtigauss(r, z, status);
    }
    if (*status) return 0.25 * X;
    double S = pg_a(0, X);
    const double Y = bo_unif(r) * S;
    int n = 0, go = 1;
    while (go) {
      ++n;
      if (n & 1) {
        S -= pg_a(n, X);
        if (Y <= S) return 0.25 * X;
      } else {
        S += pg_a(n, X);
        if (Y > S) go = 0;
      }
      if (n > 1000) { *status = BO_ERR_UNSUPPORTED_RNG_BRANCH; return 0.25 * X; }
    }
  }
  *status = BO_ERR_UNSUPPORTED_RNG_BRANCH;
  return 0.0;
}
/* PG(n, z): the sum of n PG(1, z) draws, or beyond `clt` trials the normal with PG(n, z)'s mean and variance */
static double pg_draw(bo_rng *r, long n, double z, long clt, int *status) {
  if (n <= 0) return 0.0;
  if (n > clt) {
    const double az = fabs(z);
    double mean, var;
    if (az < 1e-4) {   /* the two ratios' series around 0 */
      mean = n * (0.25 - az * az / 48.0);
      var = n * (1.0 / 24.0 - az * az / 120.0);
    } else {
      const double th = tanh(0.5 * az), ch = cosh(0.5 * az);
      mean = n * th / (2 * az);
      var = n * (sinh(az) - az) / (4 * az * az * az * ch * ch);
    }
    double x = bo_rnorm(r, mean, sqrt(var));
    if (!(x > 0)) x = mean;   /* (a draw in the far lower tail of the approximation) */
    return x;
  }
  double sum = 0.0;
  for (long i = 0; i < n; ++i) sum += pg_draw1(r, z, status);
  return sum;
}
double bo_test_rpg(bo_rng *r, long n, double z, long clt, int *status) { return pg_draw(r, n, z, clt, status); }

struct bo_logit {
  int n, p, clt;
  double *X, *y, *nt;
  bo_sss *sss;      /* (X'WX, X'Wz), gamma, beta, the sampler's RNG */
  bo_rng worker_rng; /* the imputation worker's own RNG (Imputer.hpp:136-142) */
  int substream;     /* 1: observation i of sweep s reads from position (s n + i) * 256 */
  int imputer;       /* 0: the reference's auxiliary mixture; 1: Polya-Gamma (stream 10, stride BO_PG_STRIDE) */
  uint64_t sweep;
  double logw[9];
};
#define BO_PG_STRIDE 4096
#define BO_LOGIT_STRIDE 256

bo_logit *bo_logit_create(int n, int p, const double *X, const double *y,
                          const double *ntrials, const double *mu, const double *prec,
                          const double *pi, int clt_threshold) {
  bo_logit *m = (bo_logit *)xcalloc(1, sizeof(bo_logit));
  m->n = n; m->p = p; m->clt = clt_threshold;
  m->X = (double *)xcalloc((size_t)n * p, sizeof(double));
  m->y = (double *)xcalloc(n, sizeof(double));
  m->nt = (double *)xcalloc(n, sizeof(double));
  memcpy(m->X, X, sizeof(double) * (size_t)n * p);
  memcpy(m->y, y, sizeof(double) * n);
  memcpy(m->nt, ntrials, sizeof(double) * n);
  double *xtx = (double *)xcalloc((size_t)p * p, sizeof(double));
  double *xty = (double *)xcalloc(p, sizeof(double));
  m->sss = bo_sss_create(p, xtx, xty, 0, mu, prec, pi);
  m->sss->shuffle_kind = 1;
  free(xtx); free(xty);
  for (int s = 0; s < 9; ++s) m->logw[s] = log(LOGIT_MIX_WEIGHT[s]);
  bo_rng_seed_philox(&m->worker_rng, 0, 0, 9, 0);
  return m;
}
void bo_logit_destroy(bo_logit *m) {
  if (!m) return;
  bo_sss_destroy(m->sss);
  free(m->X); free(m->y); free(m->nt);
  free(m);
}
bo_sss *bo_logit_sss(bo_logit *m) { return m->sss; }
bo_rng *bo_logit_worker_rng(bo_logit *m) { return &m->worker_rng; }
void bo_logit_use_substreams(bo_logit *m, int on) { m->substream = on; }
void bo_logit_set_imputer(bo_logit *m, int kind) { m->imputer = kind; }
void bo_logit_get_suf(const bo_logit *m, double *xtx, double *xty) {
  memcpy(xtx, m->sss->xtx, sizeof(double) * (size_t)m->p * m->p);
  memcpy(xty, m->sss->xty, sizeof(double) * m->p);
}

int bo_logit_draw(bo_logit *m) {
  const int n = m->n, p = m->p;
  bo_sss *s = m->sss;
  int status = 0;
  /* clear_latent_data + the worker's pass over the data */
  memset(s->xtx, 0, sizeof(double) * (size_t)p * p);
  memset(s->xty, 0, sizeof(double) * p);
  for (int i = 0; i < n; ++i) {
    double eta = 0;
    for (int j = 0; j < p; ++j)
      if (s->gamma[j]) eta += m->X[IDX(i, j, n)] * s->beta[j];
    const long nt = lround(m->nt[i]), ys = lround(m->y[i]);
    bo_rng *r = &m->worker_rng;
    if (m->substream) bo_rng_slot(r, m->sweep * (uint64_t)n + (uint64_t)i, m->imputer ? BO_PG_STRIDE : BO_LOGIT_STRIDE);
    double sum = 0, info = 0;
    if (m->imputer == 1) {
      /* omega ~ PG(n_i, eta); (kappa, omega) is the (sum, information) pair */
      info = pg_draw(r, nt, eta, m->clt, &status);
      if (status) return status;
      sum = (double)ys - 0.5 * (double)nt;
    } else if (nt > m->clt) {
      status = logit_impute_large(r, m->nt[i], m->y[i], eta, &sum, &info);
      if (status) return status;
    }
    for (long t = 0; t < nt && nt <= m->clt && m->imputer == 0; ++t) {
      const int success = t < ys;
      /* rtrun_logit_mt(rng, eta, 0, success) */
      const double cutpoint_prob = 1 / (1 + exp(-(0 - eta)));   /* plogis(cutpoint - mean) */
      const double u = success ? bo_runif(r, cutpoint_prob, 1) : bo_runif(r, 0, cutpoint_prob);
      const double latent = (0.0 + 1.0 * log(u / (1. - u))) + eta;   /* qlogis(u) + mean */
      /* unmix(latent - eta) */
      double wsp[9], mx = BO_NEG_INF, nc = 0;
      const double v = latent - eta;
      for (int c = 0; c < 9; ++c) {
        const double xs = (v - 0.0) / LOGIT_MIX_SIGMA[c];
        wsp[c] = m->logw[c] + -(0.918938533204672741780329736406 + 0.5 * xs * xs + log(LOGIT_MIX_SIGMA[c]));
        if (wsp[c] > mx) mx = wsp[c];
      }
      for (int c = 0; c < 9; ++c) { wsp[c] = exp(wsp[c] - mx); nc += wsp[c]; }
      for (int c = 0; c < 9; ++c) wsp[c] /= nc;
      const int ind = bo_rmulti(r, wsp, 9, &status);
      if (status) return status;
      const double w = 1.0 / (LOGIT_MIX_SIGMA[ind] * LOGIT_MIX_SIGMA[ind]);
      info += w;
      sum += latent * w;
    }
    /* SufficientStatistics::update(x, sum, info): xtx += info x x', xty += sum x */
    for (int b = 0; b < p; ++b) {
      const double xb = m->X[IDX(i, b, n)];
      s->xty[b] += xb * sum;
      for (int a = 0; a < p; ++a) s->xtx[IDX(a, b, p)] += m->X[IDX(i, a, n)] * xb * info;
    }
  }
  ++m->sweep;
  status = bo_sss_draw_model_indicators(s, 1.0);
  if (status) return status;
  return bo_sss_draw_beta(s, 1.0);
}

/* ======================================================================
 * PoissonRegressionSpikeSlabSampler (the Poisson member)
 *   PoissonRegressionSpikeSlabSampler::draw   (PoissonRegressionSpikeSlabSampler.cpp:55-59)
 *   PoissonRegressionDataImputer::impute_latent_data_point
 *                                             (PoissonRegressionAuxMixSampler.cpp:59-81)
 *   PoissonDataImputer::impute                (PoissonDataImputer.cpp:36-96)
 *   unmix_poisson_augmented_data              (poisson_mixture_approximation_table.cpp:45-62)
 *   NormalMixtureApproximation::unmix         (NormalMixtureApproximation.cpp:280-290)
 *   Rmath::rbeta_mt, Cheng's algorithm BC     (Bmath/rbeta.cpp:59-128; the shape pair is
 *                                             always (y, 1): min = 1 <= 1)
 *   rexv_mt                                   (distributions/extreme_value.cpp:60-67)
 * The normal-mixture approximations of NegLogGamma(n) are DATA of the reference
 * (create_poisson_mixture_approximation_table, with the interpolation / refit its
 * approximate(n) performs): the caller hands in, for every distinct count in the data
 * and for 1, the mixture the reference's table yields (tests: a fixture generated from
 * the compiled reference; a BOOM-side binding: the table itself).
 * ====================================================================== */
struct bo_poisson {
  int n, p;
  double *X, *y, *exposure;
  bo_sss *sss;          /* (X'WX, X'Wz), gamma, beta, the sampler's RNG */
  bo_rng worker_rng;    /* the imputation worker's own RNG */
  int substream;        /* 1: observation i of sweep s reads from position (s n + i) * BO_POISSON_STRIDE */
  uint64_t sweep;
  /* mixtures: for count counts[m] (ascending), components [off[m], off[m + 1]) */
  int ncounts;
  int64_t *counts;
  int *off;
  double *mu, *sigma, *logw;
  int64_t largest_index;   /* counts at or beyond it: the Gaussian limit */
};
#define BO_POISSON_STRIDE 256

bo_poisson *bo_poisson_create(int n, int p, const double *X, const double *y, const double *exposure,
                              const double *mu, const double *prec, const double *pi, int ncounts,
                              const int64_t *counts, const int *ncomp, const double *mix_mu,
                              const double *mix_sigma, const double *mix_weight, int64_t largest_index) {
  bo_poisson *m = (bo_poisson *)xcalloc(1, sizeof(bo_poisson));
  m->n = n; m->p = p;
  m->X = (double *)xcalloc((size_t)n * p, sizeof(double));
  m->y = (double *)xcalloc(n, sizeof(double));
  m->exposure = (double *)xcalloc(n, sizeof(double));
  memcpy(m->X, X, sizeof(double) * (size_t)n * p);
  memcpy(m->y, y, sizeof(double) * n);
  memcpy(m->exposure, exposure, sizeof(double) * n);
  double *xtx = (double *)xcalloc((size_t)p * p, sizeof(double));
  double *xty = (double *)xcalloc(p, sizeof(double));
  m->sss = bo_sss_create(p, xtx, xty, 0, mu, prec, pi);
  free(xtx); free(xty);
  m->ncounts = ncounts;
  m->counts = (int64_t *)xcalloc(ncounts, sizeof(int64_t));
  m->off = (int *)xcalloc(ncounts + 1, sizeof(int));
  memcpy(m->counts, counts, sizeof(int64_t) * ncounts);
  for (int i = 0; i < ncounts; ++i) m->off[i + 1] = m->off[i] + ncomp[i];
  const int tot = m->off[ncounts];
  m->mu = (double *)xcalloc(tot, sizeof(double));
  m->sigma = (double *)xcalloc(tot, sizeof(double));
  m->logw = (double *)xcalloc(tot, sizeof(double));
  memcpy(m->mu, mix_mu, sizeof(double) * tot);
  memcpy(m->sigma, mix_sigma, sizeof(double) * tot);
  for (int i = 0; i < tot; ++i) m->logw[i] = log(mix_weight[i]);
  m->largest_index = largest_index;
  bo_rng_seed_philox(&m->worker_rng, 0, 0, 11, 0);
  return m;
}
void bo_poisson_destroy(bo_poisson *m) {
  if (!m) return;
  bo_sss_destroy(m->sss);
  free(m->X); free(m->y); free(m->exposure);
  free(m->counts); free(m->off); free(m->mu); free(m->sigma); free(m->logw);
  free(m);
}
bo_sss *bo_poisson_sss(bo_poisson *m) { return m->sss; }
bo_rng *bo_poisson_worker_rng(bo_poisson *m) { return &m->worker_rng; }
void bo_poisson_use_substreams(bo_poisson *m, int on) { m->substream = on; }
void bo_poisson_get_suf(const bo_poisson *m, double *xtx, double *xty) {
  memcpy(xtx, m->sss->xtx, sizeof(double) * (size_t)m->p * m->p);
  memcpy(xty, m->sss->xty, sizeof(double) * m->p);
}

/* Rmath::rbeta_mt(rng, aa, 1) with aa >= 1 (Bmath/rbeta.cpp: a = min = 1, b = aa: algorithm BC) */
static double poisson_rbeta_a_1(bo_rng *rng, double aa) {
  const double expmax = 1024 * 0.693147180559945309417232121458;   /* max_exponent * M_LN2 */
  const double a = (aa < 1.0) ? aa : 1.0, b = (aa < 1.0) ? 1.0 : aa;
  const double alpha = a + b;
  const double beta = 1.0 / a, delta = 1.0 + b - a;
  const double k1 = delta * (0.0138889 + 0.0416667 * a) / (b * beta - 0.777778);
  const double k2 = 0.25 + (0.5 + 0.25 / delta) * a;
  double u1, u2, v, w, y, z;
  for (;;) {
    u1 = bo_unif(rng);
    u2 = bo_unif(rng);
    if (u1 < 0.5) {
      y = u1 * u2;
      z = u1 * y;
      if (0.25 * u2 + z - y >= k1) continue;
    } else {
      z = u1 * u1 * u2;
      if (z <= 0.25) {
        v = beta * log(u1 / (1.0 - u1));
        w = (v <= expmax) ? b * exp(v) : DBL_MAX;
        break;
      }
      if (z >= k2) continue;
    }
    v = beta * log(u1 / (1.0 - u1));
    w = (v <= expmax) ? b * exp(v) : DBL_MAX;
    if (alpha * (log(alpha / (a + w)) + v) - 1.3862944 >= log(z)) break;
  }
  /* aa == a only when aa <= 1 (then a / (a + w)); here aa >= 1 = a: for aa == 1 both are 1 */
  double ans = (aa == a) ? a / (a + w) : w / (a + w);
  if (isnan(ans)) {
    const double zero = DBL_EPSILON, one = 1.0 - zero;
    if (aa == a) return isfinite(a) ? zero : one;
    return isfinite(w) ? zero : one;
  }
  return ans;
}
double bo_test_rbeta_a1(bo_rng *r, double a) { return poisson_rbeta_a_1(r, a); }

/* unmix_poisson_augmented_data (poisson_mixture_approximation_table.cpp:45-62) */
static int poisson_unmix(const bo_poisson *m, bo_rng *rng, double u, int64_t nevents, double *mu,
                         double *sigsq) {
  if (nevents >= m->largest_index) {
    *mu = -log((double)nevents);
    *sigsq = 1.0 / (double)nevents;
    return 0;
  }
  int lo = 0, hi = m->ncounts;
  while (lo < hi) {
    const int mid = (lo + hi) / 2;
    if (m->counts[mid] < nevents) lo = mid + 1; else hi = mid;
  }
  if (lo >= m->ncounts || m->counts[lo] != nevents) return BO_ERR_UNSUPPORTED_RNG_BRANCH;
  const int c0 = m->off[lo], nc = m->off[lo + 1] - c0;
  double wsp[32], mx = BO_NEG_INF, tot = 0;
  if (nc > 32) return BO_ERR_UNSUPPORTED_RNG_BRANCH;
  for (int c = 0; c < nc; ++c) {
    const double xs = (u - m->mu[c0 + c]) / m->sigma[c0 + c];
    wsp[c] = m->logw[c0 + c] + -(0.918938533204672741780329736406 + 0.5 * xs * xs + log(m->sigma[c0 + c]));
    if (wsp[c] > mx) mx = wsp[c];
  }
  for (int c = 0; c < nc; ++c) { wsp[c] = exp(wsp[c] - mx); tot += wsp[c]; }
  for (int c = 0; c < nc; ++c) wsp[c] /= tot;
  int status = 0;
  const int ind = bo_rmulti(rng, wsp, nc, &status);
  if (status) return status;
  *mu = m->mu[c0 + ind];
  *sigsq = m->sigma[c0 + ind] * m->sigma[c0 + ind];
  return 0;
}

int bo_poisson_draw(bo_poisson *m) {
  const int n = m->n, p = m->p;
  bo_sss *s = m->sss;
  int status = 0;
  memset(s->xtx, 0, sizeof(double) * (size_t)p * p);
  memset(s->xty, 0, sizeof(double) * p);
  for (int i = 0; i < n; ++i) {
    double eta = 0;
    for (int j = 0; j < p; ++j)
      if (s->gamma[j]) eta += m->X[IDX(i, j, n)] * s->beta[j];
    const int64_t y = (int64_t)llround(m->y[i]);
    const double exposure = m->exposure[i];
    bo_rng *r = &m->worker_rng;
    if (m->substream) bo_rng_slot(r, m->sweep * (uint64_t)n + (uint64_t)i, BO_POISSON_STRIDE);
    /* PoissonDataImputer::impute: eta here is log_lambda = x'beta (the exposure enters
     * through the event times) */
    const double t_final = y > 0 ? exposure * poisson_rbeta_a_1(r, (double)y) : 0.0;
    const double delta = exposure - t_final;
    double z_ext;
    if (fabs(eta) < 600) {
      z_ext = -log(delta + (1.0 / exp(eta)) * bo_exp_rand(r));   /* rexp_mt(rng, lambda) = exp_rand / lambda */
    } else if (delta > 0) {
      const double err = -log((1.0 / 1.0) * bo_exp_rand(r)) * 1.0 + 0.0;   /* rexv_mt(rng, 0, 1) */
      const double xx = log(delta), yy = -err - eta;
      const double hi2 = xx < yy ? yy : xx, lo2 = xx < yy ? xx : yy;
      z_ext = -(hi2 + log1p(exp(lo2 - hi2)));                    /* -lse2(log(delta), -err - eta) */
    } else {
      z_ext = eta + (-log((1.0 / 1.0) * bo_exp_rand(r)) * 1.0 + 0.0);
    }
    double mu_e, sig_e, mu_i = 0, sig_i = 1;
    status = poisson_unmix(m, r, z_ext - eta, 1, &mu_e, &sig_e);
    if (status) return status;
    double z_int = 0;
    if (y > 0) {
      z_int = -log(t_final);
      status = poisson_unmix(m, r, z_int - eta, y, &mu_i, &sig_i);
      if (status) return status;
    }
    /* WeightedRegSuf::add_data(x, y, w): xtx += w x x', xty += w y x -- the internal point
     * first, then the external one (PoissonRegressionAuxMixSampler.cpp:74-80) */
    for (int pass = (y > 0 ? 0 : 1); pass < 2; ++pass) {
      const double w = pass == 0 ? 1.0 / sig_i : 1.0 / sig_e;
      const double resp = pass == 0 ? z_int - mu_i : z_ext - mu_e;
      for (int b = 0; b < p; ++b) {
        const double xb = m->X[IDX(i, b, n)];
        s->xty[b] += xb * (w * resp);
        for (int a = 0; a < p; ++a) s->xtx[IDX(a, b, p)] += m->X[IDX(i, a, n)] * xb * w;
      }
    }
  }
  ++m->sweep;
  status = bo_sss_draw_model_indicators(s, 1.0);
  if (status) return status;
  return bo_sss_draw_beta(s, 1.0);
}
