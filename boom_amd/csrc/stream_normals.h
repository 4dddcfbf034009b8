// The standard normals a sequential reader of a chain's Philox stream would draw
// with Kinderman-Ramage (Bmath/snorm.cpp:287-340), N of them in stream order,
// produced by the two wavefronts of a workgroup together.  Shared by the Kalman
// kernels (kalman_kernel.hip: local level; ssm_kernel.hip: trend + seasonal).
#pragma once
#include <hip/hip_runtime.h>

#include "device_rng.h"
#include "ssvs_params.h"

namespace boom_amd {

// LDS hand-off between the lanes of one wavefront (DS operations of a wave
// complete in order: only the compiler has to be told)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Kinderman-Ramage (Bmath/snorm.cpp:287-340, the transform of d_norm_rand) on
// uniforms that sit in LDS, starting at offset o; *used = uniforms consumed, 0
// if the draw would read past `limit`.
__device__ __forceinline__ double norm_from_lds(const __attribute__((address_space(3))) double *u, int o,
                                             int limit, int *used) {
  const double A = 2.216035867166471;
  const double C1 = 0.398942280401433, C2 = 0.180025191068563;
  int pos = o;
  *used = 0;
  if (pos + 2 > limit) return 0.0;
  const double u1 = u[pos++];
  double u2, u3, tt, z = 0.0;
  bool done = false;
  if (u1 < 0.884070402298758) {
    u2 = u[pos++];
    z = A * (1.131131635444180 * u1 + u2 - 1);
    done = true;
  } else if (u1 >= 0.973310954173898) {
    while (!done && pos + 2 <= limit) {
      u2 = u[pos++];
      u3 = u[pos++];
      tt = (A * A - 2 * log(u3));
      if (u2 * u2 < (A * A) / tt) {
        z = (u1 < 0.986655477086949) ? sqrt(tt) : -sqrt(tt);
        done = true;
      }
    }
  } else {
    // the three middle regions run ONE loop with per-lane constants (the same
    // operations on the same numbers as the reference's three copies of it, so
    // the same draws; one exp per round instead of three code paths):
    //   tt = t0 + t1 min(u2, u3);  accept when max(u2, u3) <= thr or
    //   coef |u2 - u3| <= C1 exp(-tt^2 / 2) - C2 (A - tt)
    const bool r2 = u1 >= 0.958720824790463, r3 = !r2 && u1 >= 0.911312780288703;
    const double t0 = r2 ? A : 0.479727404222441;
    const double t1 = r2 ? -0.630834801921960 : (r3 ? 1.105473661022070 : -0.595507138015940);
    const double thr = r2 ? 0.755591531667601 : (r3 ? 0.872834976671790 : 0.805577924423817);
    const double coef = r2 ? 0.034240503750111 : (r3 ? 0.049264496373128 : 0.053377549506886);
    while (!done && pos + 2 <= limit) {
      u2 = u[pos++];
      u3 = u[pos++];
      // (written as the reference writes it: A - c min, resp. c0 + c min, c0 - c min)
      tt = r2 ? A - 0.630834801921960 * fmin(u2, u3)
              : (r3 ? 0.479727404222441 + 1.105473661022070 * fmin(u2, u3)
                    : 0.479727404222441 - 0.595507138015940 * fmin(u2, u3));
      if (tt < 0.) continue;   // (only the last region can get there)
      if (fmax(u2, u3) <= thr ||
          coef * fabs(u2 - u3) <= (C1 * exp(-(tt) * (tt) / 2.0) - C2 * (A - (tt)))) {
        z = (u2 < u3) ? tt : -tt;
        done = true;
      }
    }
    (void)t0; (void)t1;
  }
  if (done) *used = pos - o;
  return z;
}

enum : int { NB_START = 512, NB_UNIF = NB_START + 64 };
enum : int { NLEV = 8, NORD = NB_START / 2 / 64, JT = 0xFFFF };  // <= 256 draws per block

// LDS of the generator: per wave one window of the stream
struct NormalsLds {
  double u[2][NB_UNIF];            // the window's uniforms
  double z[2][NB_START];           // the normal that starts at each offset
  uint8_t n1[2][NB_START];         // uniforms it consumes (0: ran out)
  uint16_t j[2][NLEV][NB_START];   // jump tables of the stream walk
  uint16_t slow[2][NB_START];      // offsets whose draw leaves the first branch
  int hand[4];                     // entry offset of the next window, draws so far, done, status
};

// szz[0 .. N) = the next N normals of stream `key` from position bpos0; the
// position after the last one goes to *pos_out (written by one lane).  Both
// waves (threadIdx.x < 128) call this; returns the chain status (CHAIN_OK or
// CHAIN_RNG_BRANCH when a draw is longer than the window margin).
__device__ __forceinline__ int stream_normals(NormalsLds &L, const PhiloxKey &key, uint64_t bpos0,
                                              int N, double *szz, uint64_t *pos_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int WAVE = 64;
  double *Lu = L.u[wave];
  double *Lz = L.z[wave];
  uint8_t *Ln1 = L.n1[wave];
  uint16_t (*Lj)[NB_START] = L.j[wave];
  uint16_t *Lslow = L.slow[wave];
    if (threadIdx.x == 0) { L.hand[0] = 0; L.hand[1] = 0; L.hand[2] = (N == 0); L.hand[3] = CHAIN_OK; }
    __syncthreads();
    for (int round = 0; !L.hand[2]; ++round) {
      const uint64_t wstart = bpos0 + (uint64_t)(2 * round + wave) * NB_START;
      // uniforms wstart .. wstart + NB_UNIF - 1, both numbers of every Philox block
      const uint64_t b0 = wstart >> 1;
      for (int i = 0; i * WAVE < NB_UNIF / 2 + 1; ++i) {
        const uint64_t blk = b0 + (uint64_t)(i * WAVE + lane);
        double u0, u1;
        philox_pair(key, blk, &u0, &u1);
        const long long o0 = (long long)(2 * blk) - (long long)wstart;
        if (o0 >= 0 && o0 < NB_UNIF) Lu[o0] = u0;
        if (o0 + 1 >= 0 && o0 + 1 < NB_UNIF) Lu[o0 + 1] = u1;
      }
      wave_lds_sync();
      // the draw that would start at every offset, lane-parallel and
      // speculative: the first Kinderman-Ramage branch (88 % of the draws, two
      // uniforms, one line) for all of them; the offsets that take another
      // branch are compacted so that the divergent code runs once per window,
      // not once per pass
      int nslow = 0;
      for (int ob = 0; ob < NB_START; ob += WAVE) {
        const int o = ob + lane;
        const double u1 = Lu[o];
        const bool fast = u1 < 0.884070402298758;
        if (fast) {
          Lz[o] = 2.216035867166471 * (1.131131635444180 * u1 + Lu[o + 1] - 1);
          Ln1[o] = 2;
        }
        const unsigned long long sm = __ballot(!fast);
        if (!fast) Lslow[nslow + __popcll(sm & ((1ull << lane) - 1ull))] = (uint16_t)o;
        nslow += __popcll(sm);
      }
      wave_lds_sync();
      for (int sb = 0; sb < nslow; sb += WAVE) {
        if (sb + lane < nslow) {
          const int o = Lslow[sb + lane];
          int used;
          const double z = norm_from_lds((const __attribute__((address_space(3))) double *)Lu, o, NB_UNIF, &used);
          Lz[o] = z;
          Ln1[o] = (uint8_t)used;
        }
      }
      wave_lds_sync();
      // The sequential reader's walk e -> e + used(e) -> ... without walking:
      // jump tables J_k[o] = offset after 2^k draws from o (binary lifting), then
      // draw number r of the window starts where the bits of r lead from e.
      for (int o = lane; o < NB_START; o += WAVE) {
        const int u1 = Ln1[o];
        Lj[0][o] = (uint16_t)(u1 ? o + u1 : JT);
      }
      wave_lds_sync();
      for (int k = 1; k < NLEV; ++k) {
        // (a lane's NB_START / 64 entries side by side: two LDS round trips per level)
        int a1[NB_START / WAVE], a2[NB_START / WAVE];
#pragma unroll
        for (int i = 0; i < NB_START / WAVE; ++i) a1[i] = Lj[k - 1][lane + i * WAVE];
#pragma unroll
        for (int i = 0; i < NB_START / WAVE; ++i) a2[i] = Lj[k - 1][a1[i] < NB_START ? a1[i] : 0];
#pragma unroll
        for (int i = 0; i < NB_START / WAVE; ++i)
          Lj[k][lane + i * WAVE] = (uint16_t)((a1[i] < NB_START) ? a2[i] : JT);
        wave_lds_sync();
      }
      // The hand-off, in window order (wave 0's window, then wave 1's), is only
      // the reader's way THROUGH the window -- how many draws start inside it and
      // where it leaves -- found by one descent over the jump tables (the largest
      // number of draws that stay inside, level by level); the look-ups that find
      // and write the draws themselves follow outside the chain, both waves at once.
      int my_entry = 0, my_n = 0, my_m = 0;
      for (int turn = 0; turn < 2; ++turn) {
        __syncthreads();
        if (turn != wave || L.hand[2]) continue;
        const int entry = __builtin_amdgcn_readfirstlane(L.hand[0]);
        const int n = __builtin_amdgcn_readfirstlane(L.hand[1]);
        const int want = N - n;
        int o = entry, cnt = 0;
#pragma unroll
        for (int k = NLEV - 1; k >= 0; --k) {
          const int nx = __builtin_amdgcn_readfirstlane((int)Lj[k][o]);
          if (nx < NB_START) { o = nx; cnt += 1 << k; }
        }
        int m = cnt + 1;                                   // draws that start inside the window
        int ex = __builtin_amdgcn_readfirstlane((int)Lj[0][o]);   // where the last of them ends
        const bool finished = (m >= want);
        if (finished) {  // the sweep's last draw is in this window: the position after `want` draws
          m = want;
          ex = entry;
#pragma unroll
          for (int k = 0; k < NLEV; ++k)
            if ((want >> k) & 1) ex = __builtin_amdgcn_readfirstlane((int)Lj[k][ex < NB_START ? ex : 0]);
        }
        // a draw longer than the margin would break the fixed windows
        const bool broken = (ex == JT) || (!finished && ex < NB_START);
        if (lane == 0) {
          L.hand[0] = ex - NB_START;
          L.hand[1] = n + m;
          L.hand[2] = finished || broken;
          if (broken) L.hand[3] = CHAIN_RNG_BRANCH;
          if (finished && !broken) *pos_out = wstart + (uint64_t)ex;  // stream position after the sweep's last draw
        }
        my_entry = entry; my_n = n; my_m = broken ? 0 : m;
      }
      if (my_m > 0) {
        // draw number r of the window starts where the bits of r lead from the entry
        int node[NORD];
#pragma unroll
        for (int i = 0; i < NORD; ++i) node[i] = my_entry;
#pragma unroll
        for (int k = 0; k < NLEV; ++k) {
          int nx[NORD];
#pragma unroll
          for (int i = 0; i < NORD; ++i) nx[i] = Lj[k][node[i] < NB_START ? node[i] : 0];
#pragma unroll
          for (int i = 0; i < NORD; ++i) {
            const int r = lane + i * WAVE;
            if ((r >> k) & 1) node[i] = (node[i] < NB_START) ? nx[i] : JT;
          }
        }
#pragma unroll
        for (int i = 0; i < NORD; ++i) {
          const int r = lane + i * WAVE;
          if (r < my_m && node[i] < NB_START) szz[my_n + r] = Lz[node[i]];
        }
      }
      __syncthreads();
    }
  return L.hand[3];
}

}  // namespace boom_amd
