// The standard normals of one simulate_forward (StateSpaceModelBase.cpp:771-790), N of
// them, by all threads of a workgroup.
//
// The reference reads ONE sequential stream: normal i starts where normal i - 1
// stopped, and Kinderman-Ramage (Bmath/snorm.cpp:287-340) consumes a data-dependent
// number of uniforms (two in 88 % of the draws, more in the tail regions), so a
// parallel reader of that stream first has to find where every draw starts.  Rounds 1
// and 2 did exactly that (windows of uniforms in LDS, the draw starting at every
// offset, jump tables, a binary-lifting walk): parity-exact, and 55 % of the Kalman
// kernel.  A counter-based generator does not need it: normal i of the chain's state
// stream reads its uniforms from the FIXED position i * STATE_SLOT_STRIDE (256 i) (the way the
// probit / logit imputers give every observation its own substream, probit_kernel.hip),
// so every draw is independent of every other and costs its own uniforms only.  The
// oracle's Philox mode does the same (bo_rnorm on stream 2); its MT mode -- the one
// pinned on the compiled reference -- reads in sequence, as the reference does.
#pragma once
#include <hip/hip_runtime.h>

#include "device_rng.h"
#include "ssvs_params.h"

namespace boom_amd {

// uniforms reserved per normal of the state stream.  A draw that runs past them goes on in
// its slot's spill stream (device_rng.h; rounds 1-3 reported CHAIN_RNG_BRANCH and stopped
// the chain): the rejection loops accept with
// probability ~0.55 per round of two uniforms, so 64 uniforms (31 rounds) would be
// exceeded about once per 1e11 slow draws -- every few hundred thousand sweep rounds of
// 1024 chains -- and 256 (127 rounds) never (1e-44).  The stride costs nothing: counters,
// not memory.
enum : int { STATE_SLOT_STRIDE = 256 };
// (ba_set_slot_limit, for the tests: at least the two uniforms of the first branch, an even number)
template <class Params>
__host__ __device__ inline int ss_slot_serve(const Params &P) {
  return (P.slot_limit >= 2 && P.slot_limit < STATE_SLOT_STRIDE) ? (P.slot_limit & ~1) : STATE_SLOT_STRIDE;
}

// LDS hand-off between the lanes of one wavefront (DS operations of a wave
// complete in order: only the compiler has to be told)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The draws that leave Kinderman-Ramage's first branch (11.6 %: rejection loops with a
// log or an exp per round) would make every wavefront wait for its unluckiest lane at
// every draw; so a chunk of the stream is done in two phases: all threads take the
// first branch of their draws and put the others on two lists in LDS (tail region /
// the three middle regions), then the lists are worked off densely -- wavefronts full
// of draws that all loop, the three middle regions in ONE loop with per-lane constants
// (the same operations on the same numbers as the reference's three copies of it).
enum : int { SN_CHUNK = 4096 };
struct NormalsLds {
  uint16_t tail[SN_CHUNK], mid[SN_CHUNK];
  int ntail, nmid;
};

// szz[i] = normal i of this sweep, i < N (or the slots' layout); the stream position moves on by N slots.
// Every thread of the workgroup calls it (it contains barriers); returns the chain
// status, the same in every thread.
// Which draw lands in which slot of the output array.  count(): slots; draw(s): the
// index of the draw that belongs in slot s, or -1 (none).  Threads walk the SLOTS, so
// the stores are coalesced whatever the layout.
struct NormalsInOrder {   // szz[i] = draw i
  int N;
  __device__ __forceinline__ int count() const { return N; }
  __device__ __forceinline__ int draw(int s) const { return s; }
};
// ONE_WAVE: the calling wavefront alone makes the draws (the round kernel of
// ss_round_kernel.hip: the chain's other wavefront is in the regression sweep meanwhile);
// the team is then the wave's 64 lanes and the hand-offs are wave-level.
template <bool ONE_WAVE>
struct NormalsTeam {
  static __device__ __forceinline__ int tid() {
    int t = (int)threadIdx.x;
    BA_OPAQUE_V(t);
    return ONE_WAVE ? (t & 63) : t;
  }
  static __device__ __forceinline__ int nth() { return ONE_WAVE ? 64 : (int)blockDim.x; }
  static __device__ __forceinline__ void sync() {
    if (ONE_WAVE) wave_lds_sync(); else __syncthreads();
  }
  static __device__ __forceinline__ int sync_or(int x) {
    if (ONE_WAVE) return __ballot(x != 0) != 0ull ? 1 : 0;
    return __syncthreads_or(x);
  }
};
// The two phases of a stretch of slots, for one team (a workgroup, or ONE wavefront:
// NormalsTeam).  Phase 1, slots [c0, c0 + nc) of the output array: the draws of the first
// branch, the others on the team's lists ltail / lmid (entry = slot - c0 + ioff; the counts
// go on from where they stand).  Phase 2: the lists' draws (slot = origin + entry); returns
// this thread's "a draw overran its slot and its spill stream" flag.
template <class Team, class Slots>
__device__ __forceinline__ void normals_phase1(uint16_t *ltail, uint16_t *lmid, int *lntail, int *lnmid,
                                               const PhiloxKey &key, uint64_t bpos0, double *szz, const Slots slots,
                                               const int c0, const int nc, const int ioff) {
  const double A = 2.216035867166471;
  const int tid = Team::tid(), nth = Team::nth();
    // ---- phase 1: u1 and u2 of every draw (one Philox block: a slot starts at an even
    // position), the first branch where it applies
    // (four slots per thread and round: four independent Philox blocks in flight -- one
    // block is a chain of ten dependent rounds, and the kernel runs two waves to a SIMD)
    for (int i0 = tid; i0 < nc; i0 += 4 * nth) {
      int dr[4];
      double u1[4], u2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * nth;
        dr[u] = (i < nc) ? slots.draw(c0 + i) : -1;
        const uint64_t start = bpos0 + (uint64_t)(dr[u] < 0 ? 0 : dr[u]) * STATE_SLOT_STRIDE;
        philox_pair(key, start >> 1, &u1[u], &u2[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * nth;
        const bool live = dr[u] >= 0;
        const bool fast = live && u1[u] < 0.884070402298758;
        const bool tail = live && u1[u] >= 0.973310954173898;
        const bool mid = live && !fast && !tail;
        if (fast) szz[c0 + i] = A * (1.131131635444180 * u1[u] + u2[u] - 1);
        // one list reservation per wavefront and list (same-address LDS atomics of many
        // lanes run one after the other: they were most of this function's time)
        const unsigned long long mt = __ballot(tail), mm = __ballot(mid);
        const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
        if (mt) {
          int base = 0;
          if ((int)(threadIdx.x & 63) == __ffsll((long long)mt) - 1) base = atomicAdd(lntail, __popcll(mt));
          base = __builtin_amdgcn_readlane(base, __ffsll((long long)mt) - 1);
          if (tail) ltail[base + __popcll(mt & below)] = (uint16_t)(i + ioff);
        }
        if (mm) {
          int base = 0;
          if ((int)(threadIdx.x & 63) == __ffsll((long long)mm) - 1) base = atomicAdd(lnmid, __popcll(mm));
          base = __builtin_amdgcn_readlane(base, __ffsll((long long)mm) - 1);
          if (mid) lmid[base + __popcll(mm & below)] = (uint16_t)(i + ioff);
        }
      }
    }
}
template <class Team, class Slots>
__device__ __forceinline__ int normals_phase2(const uint16_t *ltail, const uint16_t *lmid, const int *lntail,
                                              const int *lnmid, const PhiloxKey &key, uint64_t bpos0, double *szz,
                                              const Slots slots, int serve, const int c0) {
  const double A = 2.216035867166471;
  const double C1 = 0.398942280401433, C2 = 0.180025191068563;
  const int tid = Team::tid(), nth = Team::nth();
  const uint64_t bslot0 = bpos0 / STATE_SLOT_STRIDE;   // (the stream position is a whole number of slots)
  int bad = 0;
    // ---- phase 2a: the tail region
    const int ntail = *lntail, nmid = *lnmid;
    for (int q = tid; q < ntail; q += nth) {
      const int i = ltail[q];
      PairRng r;
      r.init_slot(key, bslot0 + (uint64_t)slots.draw(c0 + i), STATE_SLOT_STRIDE, (uint32_t)serve);
      const double u1 = r();
      double z;
      for (;;) {
        const double u2 = r();
        const double u3 = r();
        const double tt = (A * A - 2 * log(u3));
        if (u2 * u2 < (A * A) / tt) {
          z = (u1 < 0.986655477086949) ? sqrt(tt) : -sqrt(tt);
          break;
        }
        if (r.overran()) { bad = 1; z = 0.0; break; }
      }
      szz[c0 + i] = z;
    }
    // ---- phase 2b: the middle regions, one loop:
    //   tt = t0 + t1 min(u2, u3);  accept when max(u2, u3) <= thr or
    //   coef |u2 - u3| <= C1 exp(-tt^2 / 2) - C2 (A - tt)
    for (int q = tid; q < nmid; q += nth) {
      const int i = lmid[q];
      PairRng r;
      r.init_slot(key, bslot0 + (uint64_t)slots.draw(c0 + i), STATE_SLOT_STRIDE, (uint32_t)serve);
      const double u1 = r();
      const bool r2 = u1 >= 0.958720824790463, r3 = !r2 && u1 >= 0.911312780288703;
      const double thr = r2 ? 0.755591531667601 : (r3 ? 0.872834976671790 : 0.805577924423817);
      const double coef = r2 ? 0.034240503750111 : (r3 ? 0.049264496373128 : 0.053377549506886);
      double z;
      for (;;) {
        const double u2 = r();
        const double u3 = r();
        // (written as the reference writes it: A - c min, resp. c0 + c min, c0 - c min)
        const double tt = r2 ? A - 0.630834801921960 * fmin(u2, u3)
                             : (r3 ? 0.479727404222441 + 1.105473661022070 * fmin(u2, u3)
                                   : 0.479727404222441 - 0.595507138015940 * fmin(u2, u3));
        const bool ok = !(tt < 0.);   // (only the last region can fail this)
        if (ok && (fmax(u2, u3) <= thr ||
                   coef * fabs(u2 - u3) <= (C1 * exp(-(tt) * (tt) / 2.0) - C2 * (A - (tt))))) {
          z = (u2 < u3) ? tt : -tt;
          break;
        }
        if (r.overran()) { bad = 1; z = 0.0; break; }
      }
      szz[c0 + i] = z;
    }
  return bad;
}
// slots [c0, c0 + nc) by one team with lists of its own for the stretch
template <class Team, class Slots>
__device__ __forceinline__ int normals_chunk(uint16_t *ltail, uint16_t *lmid, int *lntail, int *lnmid,
                                             const PhiloxKey &key, uint64_t bpos0, double *szz, const Slots slots,
                                             int serve, const int c0, const int nc) {
  if (Team::tid() == 0) { *lntail = 0; *lnmid = 0; }
  Team::sync();
  normals_phase1<Team>(ltail, lmid, lntail, lnmid, key, bpos0, szz, slots, c0, nc, 0);
  Team::sync();
  const int bad = normals_phase2<Team>(ltail, lmid, lntail, lnmid, key, bpos0, szz, slots, serve, c0);
  Team::sync();
  return bad;
}
template <bool ONE_WAVE = false, class Slots>
__device__ __forceinline__ int stream_normals(NormalsLds &L, const PhiloxKey &key, uint64_t bpos0, int N,
                                              double *szz, uint64_t *pos_out, const Slots slots,
                                              int serve = STATE_SLOT_STRIDE) {
  typedef NormalsTeam<ONE_WAVE> Team;
  const int S = slots.count();
  int bad = 0;
  for (int c0 = 0; c0 < S; c0 += SN_CHUNK) {
    const int nc = (S - c0 < SN_CHUNK) ? S - c0 : SN_CHUNK;
    bad |= normals_chunk<Team>(L.tail, L.mid, &L.ntail, &L.nmid, key, bpos0, szz, slots, serve, c0, nc);
  }
  bad = Team::sync_or(bad);
  if (Team::tid() == 0) *pos_out = bpos0 + (uint64_t)N * STATE_SLOT_STRIDE;
  return bad ? CHAIN_RNG_BRANCH : CHAIN_OK;
}
__device__ __forceinline__ int stream_normals(NormalsLds &L, const PhiloxKey &key, uint64_t bpos0, int N,
                                              double *szz, uint64_t *pos_out, int serve = STATE_SLOT_STRIDE) {
  return stream_normals<false>(L, key, bpos0, N, szz, pos_out, NormalsInOrder{N}, serve);
}

}  // namespace boom_amd
