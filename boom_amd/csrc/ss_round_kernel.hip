// The bsts "local level + regression" sweep round as ONE persistent kernel: a chain's
// workgroup (two wavefronts) loops over the rounds of a call by itself,
//
//   StateSpacePosteriorSampler::draw()     (StateSpacePosteriorSampler.cpp:42-64)
//     observation_model()->sample_posterior()   BregVsSampler::draw       wave 0  (ssvs_sweep_body.h)
//     state_model(s)->sample_posterior()        level variance            wave 1, a round ahead
//     impute_state: the sweep's normals                                    wave 1, a round ahead
//     impute_state: filter, smoother, state, sufficient statistics        both    (kalman_lm_device.h)
//     observe_data_given_state: X'(y - state)                              wave 0, in tiles of chains
//
// instead of three launches per round that every chain leaves and enters together.  The
// separate kernels (ssvs_sweep_kernel, kalman_lm_kernel, the tiled X'e GEMM and its plane
// sum) made a round last as long as its SLOWEST chain's sweep + a state draw + a GEMM +
// three launch gaps: 134 us, of which the mean chain's own work is about half -- the 6 % of
// the chains that accept a flip in a round take twice as long in the sweep, and every
// round waited for them.  Here nothing waits for anything but what it needs (109 us a round):
//
// * sweep -> state draw is program order inside the workgroup; what the state draw needs and
//   the sweep does not give (the level variance from the previous state draw's statistics,
//   2 T standard normals: 78 wave-us of integer multiplies) is made by wave 1 while wave 0 is
//   in the X'e step and the next sweep -- wave 0 takes what is left of it when its sweep is
//   done (kalman_prepare_lead / _help, kalman_lm_device.h: the sub-chunks from both ends, no
//   read-modify-write);
// * the regression's sufficient statistic X'e needs the design matrix (1.6 MB at T = 2000,
//   p = 100): read once per chain it would be 1.6 GB of L2 traffic per round, so chains
//   share it in TILES OF UP TO 16 FORMED IN ARRIVAL ORDER: a chain that has drawn its state
//   takes the round's next ticket (tile = ticket / 16), writes its name into the tile and
//   polls the tile's names; with all SIXTEEN there (tiles are exact: the round's first tile
//   is the short one when the chain count is no multiple of 16; closing a tile early on a
//   timer made stragglers do sixteen rows alone and launches 2-4 x slower) -- or, as a safety
//   for launches that share the machine with another engine's, when the tile's first member
//   has waited SsRoundParams::close_ticks (1 ms) and closed it, a compare-and-swap that moves
//   the ticket counter to the next tile -- member s multiplies rows s, s + n, ... of the
//   lane-major series (128 time steps each) of
//   ALL the tile's residual series with the matching slab of X on the f64 matrix cores
//   (v_mfma_f64_16x16x4_f64: members x variables) and stores the partial products into the
//   members' planes; every member adds its own sixteen planes in row order as soon as none
//   of them holds the "not yet" pattern it left there.  Arrival order costs the time 16
//   chains take to arrive (measured: 6.7 us of a chain's round waiting for its tile's
//   members); the value of X'e does not depend on who shared the tile (a partial product is
//   one series' row times one slab, four consecutive steps per MFMA in the order of the tiled
//   GEMM of the separate launches: bitwise the same sums);
// * no workgroup ever waits for one that is not running: a tile is waited for by workgroups
//   that took its tickets only, so two engines whose launches share the machine cannot lock
//   each other out; a chain alone in its tile does all sixteen rows.
//
// What crosses workgroups (residual series, planes, names, ticket words) is stored
// write-through and loaded past the L1 (sc1 accesses: relaxed agent-scope atomics, sc1
// buffer loads), a name after the s_waitcnt vmcnt(0) of the data it stands for -- no
// buffer_wbl2 / buffer_inv of a whole L2 on the way (MI355X_MICROARCH.md, inter-workgroup
// visibility: the measured forms; DESIGN 6 "tried": those cost the fused GEMM 60 us).  Each
// hop is 1-3 us on a busy chip, so the step is built from as few as it can be: ticket, names,
// series, planes.
#define BA_ROUND_KERNEL
#include "ktimer.h"
#include "ssvs_sweep_body.h"

#define BA_HAVE_WAVE_HELPERS
#include "kalman_lm_device.h"

namespace boom_amd {

namespace {

enum : int { RT = SS_ROUND_TILE };
constexpr int VG = 7;   // variable tiles (of 16) a pass of the tile product covers
// "not yet": what a chain leaves in its planes before it joins a tile (a quiet NaN no sum of
// finite products is)
constexpr unsigned long long PLANE_EMPTY = 0x7ff8badc0ffee0ddull;

__device__ __forceinline__ double ld_coh(const double *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_coh(const int32_t *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coh(double *p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coh(int32_t *p, int32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// this thread's stores have left (write-through ones: are where every workgroup sees them)
__device__ __forceinline__ void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The kernels this one is made of read their chain's scalars (status word, stream positions,
// variances, statistics) with wave-uniform loads, which the compiler may send through the
// SCALAR cache -- and the vector stores that wrote them a round ago do not update that cache
// (in a kernel of their own those loads come first and find it empty).  So after every
// barrier behind which such values may be read: invalidate it.
__device__ __forceinline__ void fresh_scalars() { asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); }

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
// 16 bytes past the L1 (buffer_load_dwordx4 ... sc1); byte offsets below 2^31
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_buffer(const void *base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00027000);
}
__device__ __forceinline__ d2 ld_coh16(__amdgpu_buffer_rsrc_t r, uint32_t byte_offset) {
  return __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_offset, 0, 16));
}

// diagnostic build (-DBA_RSTAMPS): 100 MHz ticks per phase, summed over the launch's rounds,
// per chain and wave, into F.stamps (chains x 2 x 8)
// (RSTAMP: diag.h)

// One member's share of a tile's product: out[member, variable] over the 128 steps of rows slot,
// slot + n, ... of the lane-major series, four steps a matrix instruction in time order (lane l
// feeds member / variable l & 15 at step l >> 4 of the four) -- the tiled GEMM's products in the
// tiled GEMM's order (xtwx_cols_kernel<false, 128>: the separate launches' X'e, bit for bit),
// stored into the members' planes.  mine: lane l < 16 holds the chain of place l.
// Round 6: a row's 32 residual fragments are ALL asked for before the first product.  They come
// from other workgroups' stores, past the L1 (sc1), at 1 - 2 us a round trip on the busy chip;
// fetched a batch ahead of their use (round 5) every one of the eight batches waited most of a
// trip -- the product took 12.6 us whether a wave multiplied seven variable tiles or four (the
// attempt to share it between the chain's two wavefronts) -- so the matrix cores were not what
// it waited for.  X is read-only and comes through the L1: its fragments stay a batch ahead, in
// two passes of four and three variable tiles over the same residual fragments (registers).
template <int VGW>
__device__ __forceinline__ void tile_product_pass(const __amdgpu_buffer_rsrc_t xb, const double (&a)[LM_THREADS / 4],
                                                  const SsParams &S, const SsRoundParams &F, const int mine, const int lo,
                                                  const int n, const int rr, const int p, const int fc, const int fk,
                                                  const int jbase) {
  d4 acc[VGW];
#pragma unroll
  for (int v = 0; v < VGW; ++v) acc[v] = d4{0.0, 0.0, 0.0, 0.0};
  uint32_t xo[VGW];
#pragma unroll
  for (int v = 0; v < VGW; ++v) {
    int j = jbase + 16 * v + fc;
    j = j < p ? j : p - 1;
    xo[v] = (uint32_t)(((size_t)j * LM_TP + (size_t)rr * LM_THREADS) * 8) + 8u * fk;
  }
  constexpr int NBAT = LM_THREADS / 16;
  double b[2][4][VGW];
  auto fetch = [&](int bi, int m0) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < VGW; ++v)
        b[bi][u][v] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(xb, (int)(xo[v] + 32u * (m0 + u)), 0, 0));
  };
  fetch(0, 0);
#pragma unroll
  for (int bt = 0; bt < NBAT; ++bt) {
    if (bt + 1 < NBAT) fetch((bt + 1) & 1, 4 * (bt + 1));
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < VGW; ++v)
        acc[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * bt + u], b[bt & 1][u][v], acc[v], 0, 0, 0);
  }
  // to the members' planes of this row (register q of lane l: member (l >> 4) + 4 q)
#pragma unroll
  for (int v = 0; v < VGW; ++v) {
    const int j = jbase + 16 * v + fc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int mi = fk + 4 * q;
      const int mc = __shfl(mine, (lo + mi) & (RT - 1));
      if (mi < n && j < p) st_coh(F.planes + ((size_t)rr * S.chains + mc) * p + j, acc[v][q]);
    }
  }
}
__device__ __forceinline__ void tile_product(const SsvsParams &P, const SsParams &S, const SsRoundParams &F,
                                             const int mine, const int lo, const int n, const int slot, const int lane) {
  const int p = P.p;
  const int fc = lane & 15, fk = lane >> 4;
  const int mem_a = __shfl(mine, lo + (fc < n ? fc : 0));
  // (the residual series' resource starts at the launch's FIRST chain: a tile's members are
  // chains of this launch, at most the resident count of them, so the 32-bit byte offset
  // stays far below 2^31 whatever the engine's chain count -- from the engine's chain 0 it
  // passed 2^31 at 14 563 chains and read zeros)
  const __amdgpu_buffer_rsrc_t eb = coh_buffer(S.scratch + (size_t)P.chain_first * S.scratch_stride), xb = coh_buffer(S.Xt);
  const uint32_t eo = (uint32_t)(((size_t)(mem_a - P.chain_first) * S.scratch_stride + S.TP) * 8) + 8u * fk;
  constexpr int VGA = 4, VGB = VG - VGA;
  for (int rr = slot; rr < RT; rr += n) {
    const uint32_t er = eo + (uint32_t)rr * LM_THREADS * 8;
    double a[LM_THREADS / 4];
#pragma unroll
    for (int m = 0; m < LM_THREADS / 4; ++m)
      a[m] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(eb, (int)(er + 32u * m), 0, 16));
    for (int j0 = 0; j0 < p; j0 += 16 * VG) {
      tile_product_pass<VGA>(xb, a, S, F, mine, lo, n, rr, p, fc, fk, j0);
      if (j0 + 16 * VGA < p) tile_product_pass<VGB>(xb, a, S, F, mine, lo, n, rr, p, fc, fk, j0 + 16 * VGA);
    }
  }
}

}  // namespace

size_t ss_round_lds(int p, int kcap) {
  return ((ssvs_lds_layout(p, kcap).total + 15u) & ~15u) + sizeof(KalmanLmLds);
}

template <int NB>
__global__ __launch_bounds__(LM_THREADS, 2) void ss_round_kernel(SsvsParams P, SsParams S, SsRoundParams F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int p = P.p, C = P.chain_count;
  // LDS: the sweep's working set | the state draw's (the normals generator's lists first)
  const SsvsLds lay = ssvs_lds_layout(p, NB * 8);
  KalmanLmLds &klds = *reinterpret_cast<KalmanLmLds *>(smem + ((lay.total + 15u) & ~15u));
  S.prepared = 1;
  S.only_ran = nullptr;
  P.ran = nullptr;
  P.run_limit = 0;
  P.xty_planes = nullptr;
#ifdef BA_RSTAMPS
  long long rph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rlast = wall_clock64();
  const long long rstart = rlast;
#endif
  {
    // the first round's level variance and normals (wave 1); the chain's planes empty (wave 0)
    int chain = (int)blockIdx.x + P.chain_first, tid = threadIdx.x;
    BA_OPAQUE_S(chain);
    BA_OPAQUE_V(tid);
    if (tid == 0) klds.share.seq = 0;
    __syncthreads();
    if ((tid >> 6) == 1) {
      kalman_prepare_lead(S, chain, S.status[chain], 1, klds);
    } else {
      for (int z = 0; z < RT; ++z)
        for (int j = tid; j < p; j += WAVE)
          st_coh(F.planes + ((size_t)z * S.chains + chain) * p + j, __builtin_bit_cast(double, PLANE_EMPTY));
    }
  }
  for (int r = 0; r < F.rounds; ++r) {
    // (opaque once per round: device_rng.h, BA_OPAQUE_*)
    int chain = (int)blockIdx.x + P.chain_first, tid = threadIdx.x;
    BA_OPAQUE_S(chain);
    BA_OPAQUE_V(tid);
    // (... and the launch's constants that the state draw turns into vector values -- sqrt(P0),
    // the level prior's terms, array offsets: computed once before the loop they lived in
    // scratch memory for the whole launch, six of them)
    BA_OPAQUE_S(S.P0); BA_OPAQUE_S(S.a0); BA_OPAQUE_S(S.TP); BA_OPAQUE_S(S.T);
    BA_OPAQUE_S(S.level_prior_df); BA_OPAQUE_S(S.level_prior_ss); BA_OPAQUE_S(S.level_sigma_max);
    BA_OPAQUE_S(P.sigma_max); BA_OPAQUE_S(P.prior_df); BA_OPAQUE_S(P.prior_ss);
    const int lane = tid & 63, wave = tid >> 6;
    // ---- 1. the regression's draw (wave 0; wave 1 is still making this round's normals)
    if (wave == 0) {
      fresh_scalars();   // (the plane sum just stored X'e, which the sweep reads as a scalar where it can)
#ifndef RK_PRIO
#define RK_PRIO 3
#endif
      __builtin_amdgcn_s_setprio(RK_PRIO);
      ssvs_sweep_body<NB, 1, 1>(P, 1, chain, smem);
      __builtin_amdgcn_s_setprio(0);
      kalman_prepare_help(S, chain, r + 1, klds);   // (what is left of this round's normals)
    }
    stores_done();
    RSTAMP(0);   // wave 0: sweep; wave 1: the rest of variance + normals
    __syncthreads();
    fresh_scalars();
    RSTAMP(1);   // ... waiting for the other wave
    P.model_keep = 1;   // (from here on the chain's model block is its own last sweep's)
    // ---- 2. the state, the level model's and the regression's sufficient statistics
    // (a chain the sweep parked, or one in error, sits the round out)
    const bool drew = kalman_lm_body<true>(S, 1, chain, klds);
#ifdef BA_RSTAMPS
    // (diagnostic: the round's tag in the residual series' last padding element -- X is zero there)
    if (tid == LM_THREADS - 1 && S.T < LM_TP)
      st_coh(S.scratch + (size_t)chain * S.scratch_stride + S.TP + (LM_TP - 1), (double)(F.debug_seq * 1000 + r));
#endif
    stores_done();
    __syncthreads();
    fresh_scalars();
    if (F.debug && !drew && tid == 0) {
      // (diagnostic: a chain that stopped in the state draw -- what the draw was given)
      const int at = atomicAdd(F.debug + 1, 1);
      if (at < 4) {
        double *o = reinterpret_cast<double *>(F.debug + 16 * 17) + at * 8;   // (behind the sums' fifteen records)
        o[0] = chain; o[1] = r; o[2] = S.status[chain];
        o[3] = __hip_atomic_load(S.sigsq + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o[4] = __hip_atomic_load(S.level_sigsq + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o[5] = S.prep_n[(size_t)S.zbuf * S.chains + chain];
        o[6] = __hip_atomic_load(S.level_sumsq + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        o[7] = __hip_atomic_load(S.level_n + chain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    RSTAMP(2);   // state draw
    if (wave == 1) {
      // ---- 3a. the NEXT round's level variance (from this draw's statistics) and normals
      // (round 6, tried: this wave taking three of the product's seven variable tiles, between two
      // sub-chunks of normals or before them -- 4 to 6 us a round SLOWER: the product was waiting
      // for its loads, not for the matrix cores, and a second wave waits just as long; and the
      // regression sweep by BOTH wavefronts -- ssvs_sweep_body<NB, 2, 2>, this wave leaving the
      // normals when wave 0 is through with the tile and the two finishing them behind the sweep
      // --: 118 us a round against 103: a bsts round's sweep refills its table every time, which
      // the two-wave form does command by command across the workgroup's barrier)
      if (r + 1 < F.rounds) kalman_prepare_lead(S, chain, drew ? (int)CHAIN_OK : (int)CHAIN_RNG_BRANCH, r + 2, klds);
      continue;
    }
    // ---- 3b. X'e (wave 0).  A ticket, the chain's name into its place (sixteen places to a
    // tile; where the chains are no multiple of sixteen the round's FIRST tile is the short
    // one: its places start at `lo`, and the chains that are ahead of the others do the extra
    // rows) ...
    int32_t *const names = F.members + (size_t)r * ((size_t)C * RT + 2 * RT);
    int32_t *const sizes = F.sizes + (size_t)r * C;
    int ticket = 0;
    if (lane == 0) ticket = atomicAdd(F.ticket + r, 1);
    const int off = (RT - (C & (RT - 1))) & (RT - 1);   // (places run `off` ahead of the ticket counter)
    ticket = __builtin_amdgcn_readfirstlane(ticket) + off;
    const int tile = ticket >> 4, lo = tile == 0 ? off : 0;
    if (lane == 0) st_coh(names + ticket, chain);
    // ... and the tile's names: all of them -- or, had the tile's first member to wait so long
    // that the other chains cannot be running (an engine that shares the machine with another
    // one's launch), the ones that got there: it closes the tile with a compare-and-swap that
    // moves the ticket counter to the next tile
    int hi = RT, mine = -1;
    {
      const long long t0 = wall_clock64();
      for (;;) {
        int v = -1;
        if (lane < RT) v = ld_coh(names + tile * RT + lane);
        else if (lane == RT) v = ld_coh(sizes + tile);
        const int closed = __builtin_amdgcn_readlane(v, RT);
        const unsigned long long have = __ballot(lane < RT && v >= 0);
        hi = closed > 0 ? closed : RT;
        const unsigned long long want = ((1ull << hi) - 1ull) & ~((1ull << lo) - 1ull);
        if ((have & want) == want) {
          mine = v;
          break;
        }
        if ((ticket & (RT - 1)) == lo && closed <= 0 && wall_clock64() - t0 > F.close_ticks) {
          if (lane == 0) {
            const int at = ld_coh(F.ticket + r) + off;   // the next place
            if ((at >> 4) == tile && atomicCAS(F.ticket + r, at - off, ((tile + 1) << 4) - off) == at - off)
              st_coh(sizes + tile, at & (RT - 1));
          }
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    const int n = hi - lo, slot = (ticket & (RT - 1)) - lo;   // members, and which of them this chain is
    RSTAMP(3);   // ticket, names
    // ... and multiply this member's rows
    tile_product(P, S, F, mine, lo, n, slot, lane);
    RSTAMP(4);   // the member's share of the product
    // ... and add the chain's own planes, in row order, once every member's share is there
    if (drew) {
      for (int j0 = 0; j0 < p; j0 += WAVE) {
        const int j = j0 + lane;
        const double *pl = F.planes + (size_t)chain * p + (j < p ? j : 0);
        double v[RT];
        for (;;) {
          bool missing = false;
#pragma unroll
          for (int z = 0; z < RT; ++z) {
            v[z] = ld_coh(pl + (size_t)z * S.chains * p);
            missing = missing || __builtin_bit_cast(unsigned long long, v[z]) == PLANE_EMPTY;
          }
          if (!__any(missing && j < p)) break;
          __builtin_amdgcn_s_sleep(2);
        }
        double a = v[0];
#pragma unroll
        for (int z = 1; z < RT; ++z) a += v[z];
        if (F.debug && j < p && !(a == a)) {
          // (diagnostic, ba_ss_set_tuning 6: a sum that is not a number -- who, when, which row)
          const int at = atomicAdd(F.debug, 1);
          if (at < 15) {
            int zbad = 0;
            for (int z = 0; z < RT; ++z) if (!(v[z] == v[z])) zbad = z;
            int32_t *o = F.debug + 16 + at * 16;
            o[0] = chain; o[1] = r; o[2] = j; o[3] = zbad; o[4] = tile; o[5] = slot; o[6] = n;
            o[7] = (int32_t)(__builtin_bit_cast(unsigned long long, v[zbad]) >> 32);
            o[8] = (int32_t)__builtin_bit_cast(unsigned long long, v[zbad]);
            o[9] = F.rounds;
          }
        }
#ifdef BA_RSTAMPS
        if (__any(j < p && !(a == a))) rph[7] += 1000000;
#endif
        if (j < p) {
          S.xty[(size_t)chain * p + j] = a;
#pragma unroll
          for (int z = 0; z < RT; ++z)
            st_coh(const_cast<double *>(pl) + (size_t)z * S.chains * p, __builtin_bit_cast(double, PLANE_EMPTY));
        }
      }
    }
    RSTAMP(5);   // waiting for the other members' shares, plane sum
    // ---- 4. what the callers' loop reads of this round's draw (engine.hip, look-ahead)
    if (F.rgamma) {
      const size_t at = ((size_t)F.rec_slot * S.chains + chain) * F.rec_len + F.rec_first + r;
      for (int j = lane; j < p; j += WAVE) {
        F.rgamma[at * p + j] = P.gamma[(size_t)chain * p + j];
        F.rbeta[at * p + j] = P.beta[(size_t)chain * p + j];
      }
      if (lane == 0) {
        F.rsig[at] = P.sigsq[chain];
        F.rvar[at] = S.level_used[chain];
      }
      const int reg = F.reg_of_chain[chain];
      if (reg >= 0) {
        const double *src = S.scratch + (size_t)chain * S.scratch_stride + (size_t)SS_STATE_ARRAY * S.TP;
        double *dst = F.rstate + (((size_t)F.rec_slot * F.nreg + reg) * F.rec_len + F.rec_first + r) * (size_t)S.TP;
        for (int j = lane; j < S.TP; j += WAVE) dst[j] = src[j];
      }
    }
    stores_done();
    RSTAMP(6);   // record
  }
#ifdef BA_RSTAMPS
  {
    long long bad = rph[7];
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
    rph[7] = bad;
  }
  if (F.stamps && (threadIdx.x & 63) == 0) {
    double *o = F.stamps + ((size_t)(blockIdx.x + P.chain_first) * 2 + (threadIdx.x >> 6)) * 8;
    if (threadIdx.x >> 6) {   // (wave 1 has no phases 5, 6: when the workgroup started and ended, ticks mod 2^40)
      rph[5] = rstart & ((1ll << 40) - 1);
      rph[6] = wall_clock64() & ((1ll << 40) - 1);
    }
    for (int i = 0; i < 8; ++i) o[i] += (double)rph[i];
  }
#endif
}

template <int NB>
static hipError_t launch_round_t(hipStream_t stream, const SsvsParams &P, const SsParams &S, const SsRoundParams &F,
                                 int *max_resident) {
  const size_t lds = ss_round_lds(P.p, NB * 8);
  hipError_t e = hipFuncSetAttribute((const void *)ss_round_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return e;
  if (max_resident) {
    int per_cu = 0, dev = 0, cus = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)ss_round_kernel<NB>, LM_THREADS, lds);
    if (e != hipSuccess) return e;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    *max_resident = per_cu * cus;
    return hipSuccess;
  }
  KtScope kt(stream, KT_SS_ROUND);
  hipLaunchKernelGGL((ss_round_kernel<NB>), dim3(P.chain_count), dim3(LM_THREADS), lds, stream, P, S, F);
  return hipGetLastError();
}

// max_resident != nullptr: no launch -- how many chains' workgroups the device holds at once
hipError_t launch_ss_round(hipStream_t stream, const SsvsParams &P, const SsParams &S, const SsRoundParams &F,
                           int *max_resident) {
  switch (P.kcap) {
    case 16: return launch_round_t<2>(stream, P, S, F, max_resident);
    case 32: return launch_round_t<4>(stream, P, S, F, max_resident);
    case 48: return launch_round_t<6>(stream, P, S, F, max_resident);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace boom_amd
