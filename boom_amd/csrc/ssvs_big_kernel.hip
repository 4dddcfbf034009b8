// SSVS Gibbs sweep for chains whose model has outgrown the LDS-resident kernel
// (ssvs_kernel.hip: at most 64 included variables).  The reference has no size
// limit -- BregVsSampler::set_reg_post_params factors whatever k x k system the
// current model asks for (Models/Glm/PosteriorSamplers/BregVsSampler.cpp:395-426)
// -- so neither has the engine: a chain that stops with
// CHAIN_MODEL_TOO_LARGE in the LDS kernel is picked up here, with its factors
// resident in HBM (the chain's model block, ssvs_scalar_layout(big_kcap)) and
// streamed through the scalar cache.
//
// Same Markov chain, same stream positions, same table / two-slot scheme as the
// LDS kernel (see its header comment); what differs is where things live and
// how the k-sized pieces are cut:
//   * per-lane triangular solves (one proposal, or one row of a factor, per
//     lane) run over PANELS of 64 rows: the panel being solved sits in 64
//     registers, finished panels are parked in a per-wave HBM scratch
//     (coalesced [row][lane] layout) and re-read 8 values at a time while the
//     factor's 8 x 8 blocks arrive as SGPR operands (s_load);
//   * a model is (re)built by the bordering method: row i of chol(M_g) is the
//     solution of L[0:i,0:i] x = M_g[0:i, i], i.e. the SAME per-lane solve, 64
//     rows at a time against the finished panels, followed by the 64 x 64
//     diagonal tile (trailing update with the wave's own parked rows through
//     the scalar cache, then a lane = row Cholesky in LDS);
//   * the back substitution of the beta draw works tile by tile on coalesced
//     reads of the factor.
// One workgroup of two wavefronts per chain: the master (wave 0) runs the
// sweep, wave 1 shares the table fills and the shuffle uniforms.  No forked
// quiet sweeps here: the chains that need this kernel are few and their cost
// is the O(p k^2) table fill after every accepted flip.
#include "ktimer.h"
#include "ssvs_device.h"
#include "ssvs_fill_mfma.h"

namespace boom_amd {

namespace {

enum : int { BCMD_EXIT = 0, BCMD_EVAL = 1, BCMD_UNIF = 2 };

// inclusive prefix sum over the wave, lane order
__device__ __forceinline__ double big_prefix(double x) {
  x += dpp_f64<0x111, 0xf>(x, 0.0);
  x += dpp_f64<0x112, 0xf>(x, 0.0);
  x += dpp_f64<0x114, 0xf>(x, 0.0);
  x += dpp_f64<0x118, 0xf>(x, 0.0);
  x += dpp_f64<0x142, 0xa>(x, 0.0);
  x += dpp_f64<0x143, 0xc>(x, 0.0);
  return x;
}

// offset (doubles) of 8 x 8 block (I, J), J <= I, of a block-packed factor
__device__ __forceinline__ int blk_off(int I, int J) { return ((I * (I + 1)) / 2 + J) * 64; }

// make this wave's global stores visible to its own scalar loads and re-derive
// a constant-address-space pointer that no load can be hoisted above
__device__ __forceinline__ c_f64 *scalar_view(const double *ptr) {
  unsigned long long u = uni((uint64_t)ptr);  // (wave-uniform by construction; tell the compiler)
  asm volatile("s_waitcnt vmcnt(0)\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(u) : : "memory");
  return (c_f64 *)u;
}

// Per-lane forward substitution L x = rhs over panels of 64 rows.
//   LB, rd   block-packed factor and reciprocal diagonal (scalar cache)
//   k        rows of the system (rows >= k of the last 8-block are zero, rd = 0)
//   npan     panels to solve (ceil(k / 64))
//   xs       this wave's parking space, element (row m, lane l) at m * 64 + l
//   rhs(I, a)  fills a[0..63] with the right-hand side of rows 64 I ..
//   fin(I, a)  sees the solved panel
//   park_all   also park the last panel (callers that read xs afterwards)
template <class Rhs, class Fin>
__device__ __forceinline__ void big_solve(c_f64 *__restrict__ LB, c_f64 *__restrict__ rd, int k,
                                          int npan, bool park_all, double *__restrict__ xs,
                                          int lane, Rhs rhs, Fin fin) {
  for (int I = 0; I < npan; ++I) {
    double a[64];
    rhs(I, a);
    for (int J = 0; J < I; ++J) {
#pragma nounroll
      for (int cb = 0; cb < 8; ++cb) {
        double xj[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) xj[c] = xs[(size_t)(J * 64 + cb * 8 + c) * 64 + lane];
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) {
          if (I * 64 + rb * 8 < k) {
            c_f64 *blk = LB + blk_off(I * 8 + rb, J * 8 + cb);
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
              for (int c = 0; c < 8; ++c) a[rb * 8 + r] -= blk[r * 8 + c] * xj[c];
          }
        }
      }
    }
#pragma unroll
    for (int rb = 0; rb < 8; ++rb) {
      if (I * 64 + rb * 8 < k) {
#pragma unroll
        for (int cb = 0; cb < rb; ++cb) {
          c_f64 *blk = LB + blk_off(I * 8 + rb, I * 8 + cb);
#pragma unroll
          for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int c = 0; c < 8; ++c) a[rb * 8 + r] -= blk[r * 8 + c] * a[cb * 8 + c];
        }
        c_f64 *blk = LB + blk_off(I * 8 + rb, I * 8 + rb);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          a[rb * 8 + r] = a[rb * 8 + r] * rd[I * 64 + rb * 8 + r];
#pragma unroll
          for (int r2 = r + 1; r2 < 8; ++r2) a[rb * 8 + r2] -= blk[r2 * 8 + r] * a[rb * 8 + r];
        }
      }
    }
    fin(I, a);
    if (park_all || I + 1 < npan) {
#pragma unroll
      for (int r = 0; r < 64; ++r) xs[(size_t)(I * 64 + r) * 64 + lane] = a[r];
    }
  }
}

// In-place Cholesky of one 64 x 64 tile in LDS (block-packed lower triangle,
// lane i owns row i, kk <= 64 rows): the single-matrix form of chol_blocks2.
__device__ __forceinline__ void chol_tile(lds_f64 *T, lds_f64 *rdt, int kk, int lane,
                                          bool *ok_out, double *logdet_sum) {
  const int i = lane;
  bool ok = true;
  for (int j = 0; j < kk && ok; ++j) {
    const bool mine = (i >= j) && (i < kk);
    const int ii = mine ? i : j;
    const int jb = j >> 3;
    const int offi = ((ii >> 3) * ((ii >> 3) + 1) / 2) * 64 + (ii & 7) * 8;
    const int offj = (jb * (jb + 1) / 2) * 64 + (j & 7) * 8;
    double s = T[bidx(ii, j)];
    for (int nb = 0; nb < jb; ++nb) {
      // (a block's row is 64 bytes, 16-byte aligned: four 16-byte reads each for the lane's
      // row and for row j, where eight 8-byte ones were two thirds of the loop's instructions)
      typedef double d2_t __attribute__((ext_vector_type(2)));
      const AS_LDS d2_t *pa = (const AS_LDS d2_t *)(T + offi + nb * 64);
      const AS_LDS d2_t *pb = (const AS_LDS d2_t *)(T + offj + nb * 64);
      d2_t a[4], b[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { a[t] = pa[t]; b[t] = pb[t]; }
#pragma unroll
      for (int t = 0; t < 4; ++t) { s -= a[t].x * b[t].x; s -= a[t].y * b[t].y; }
    }
    {
      const int rem = j & 7;
#pragma unroll
      for (int t = 0; t < 7; ++t)
        if (t < rem) s -= T[offi + jb * 64 + t] * T[offj + jb * 64 + t];
    }
    const double d = bcast_u(s, j);
    if (!(d > 0.0)) { ok = false; break; }
    const double sd = sqrt(d);
    if (i == j) {
      T[bidx(j, j)] = sd;
      rdt[j] = 1.0 / sd;
    } else if (mine) {
      T[bidx(i, j)] = s / sd;
    }
    wave_sync();
  }
  double l = 0.0;
  const double lg = (ok && i < kk) ? log(T[bidx(i, i)]) : 0.0;
  for (int j = 0; j < kk; ++j) l += bcast_u(lg, j);
  *ok_out = ok;
  *logdet_sum = l;
}

// The same factorisation with the lane's row IN REGISTERS (lane i = row i, acc[c] = its entry
// in column c), right-looking: column j is scaled, handed round through LDS (64 doubles, every
// lane reads all of them: broadcast reads) and subtracted from the columns to its right.  Entry
// (i, c) receives l_i0 l_c0, l_i1 l_c1, ... in that order, each as one fused multiply-subtract
// -- the subtractions of chol_tile's dot products, in their order, so the factor is bitwise
// chol_tile's.  What differs is the critical path: chol_tile's column j is a dot product of
// length j (j dependent fused multiply-adds behind 2 j LDS reads, 3.7 k cycles a column at
// k = 64: 38 % of a model build); here a column is one multiply-add deep.  col: 2 x 64 doubles
// of LDS.  Entries above the diagonal hold numbers nobody reads.
__device__ __forceinline__ void chol_tile_regs(double (&acc)[64], lds_f64 *col, int kk, int lane,
                                               bool *ok_out, double *logdet_sum, double *rd_out) {
  typedef double d2_t __attribute__((ext_vector_type(2)));
  bool ok = true;
  double rdl = 0.0, diag = 1.0;
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    if (j < kk && ok) {
      const double d = bcast_u(acc[j], j);
      if (!(d > 0.0)) {
        ok = false;
      } else {
        const double sd = sqrt(d);
        double cj = (lane == j) ? sd : acc[j] / sd;
        cj = (lane >= j && lane < kk) ? cj : 0.0;
        acc[j] = cj;
        if (lane == j) { rdl = 1.0 / sd; diag = sd; }
        if (j + 1 < 64) {
          lds_f64 *cb = col + (j & 1) * 64;   // (two buffers: one hand-off a column)
          cb[lane] = cj;
          wave_sync();
          const AS_LDS d2_t *cp = (const AS_LDS d2_t *)cb;
#pragma unroll
          for (int c0 = (j + 1) & ~1; c0 < 64; c0 += 2) {
            const d2_t v = cp[c0 >> 1];
            if (c0 > j) acc[c0] -= cj * v.x;
            acc[c0 + 1] -= cj * v.y;
          }
        }
      }
    }
  }
  double l = 0.0;
  const double lg = (ok && lane < kk) ? log(diag) : 0.0;
  for (int j = 0; j < kk; ++j) l += bcast_u(lg, j);
  *ok_out = ok;
  *logdet_sum = l;
  *rd_out = rdl;
}

struct BigCtx {
  int kcap;
  SsvsScalarLayout S;
  double *xs;        // this wave's solve parking space
  lds_f64 *tile, *rdt, *y, *bst;
  lds_u16 *gst;
};

// The heavy routines below are real functions (one copy of their unrolled
// solves each): their arguments arrive in vector registers / through memory, so
// what is wave-uniform has to be declared uniform again for the factor loads to
// be scalar loads and the branches scalar branches.
template <class T>
__device__ __forceinline__ T *uni_ptr(T *q) { return (T *)uni((uint64_t)q); }
__device__ __forceinline__ uint32_t uni_u(uint32_t x) { return (uint32_t)uni((int)x); }
__device__ __forceinline__ void uniform_copy(Chain &c, const Chain &in, int lane) {
  c = in;
  c.lane = lane;
  c.p = uni(in.p);
  c.k = uni(in.k);
  c.tab_lp = uni_ptr(in.tab_lp);
  c.tab_kind = uni_ptr(in.tab_kind);
  c.sc_store = uni_ptr(in.sc_store);
  c.xty = uni_ptr(in.xty);
  c.DF = uni(in.DF);
  c.ss0q = uni(in.ss0q);
  c.mode = uni(in.mode);
  c.sv = uni(in.sv);
  c.sa = uni(in.sa);
  c.sx = uni(in.sx);
}
__device__ __forceinline__ void uniform_copy(BigCtx &b, const BigCtx &in) {
  b = in;
  b.kcap = uni(in.kcap);
  b.xs = uni_ptr(in.xs);
  b.S.Lv = uni_u(in.S.Lv); b.S.La = uni_u(in.S.La); b.S.rdv = uni_u(in.S.rdv); b.S.rda = uni_u(in.S.rda);
  b.S.w = uni_u(in.S.w); b.S.bg = uni_u(in.S.bg); b.S.g = uni_u(in.S.g); b.S.scal = uni_u(in.S.scal);
  b.S.total = uni_u(in.S.total);
  b.S.iv = uni_u(in.S.iv); b.S.ia = uni_u(in.S.ia);
}
__device__ __forceinline__ void uniform_copy(Model &m, const Model &in) {
  m.logp = uni(in.logp); m.lp = uni(in.lp); m.ldv = uni(in.ldv); m.lda = uni(in.lda);
  m.Q = uni(in.Q); m.c = uni(in.c); m.SS = uni(in.SS);
  m.pd = uni((int)in.pd) != 0; m.bad = uni(in.bad);
}
// the launch parameters the heavy routines use
struct BigP {
  const double *V, *A, *b, *l1, *l0;
  const double *vdiag;        // the chain's diagonal of V when V is computed column by column, or nullptr
  int64_t max_model_size;
};
__device__ __forceinline__ void uniform_copy(BigP &q, const SsvsParams &P) {
  q.V = uni_ptr(P.V); q.A = uni_ptr(P.A); q.b = uni_ptr(P.b); q.l1 = uni_ptr(P.l1); q.l0 = uni_ptr(P.l0);
  q.max_model_size = (int64_t)uni((uint64_t)P.max_model_size);
  q.vdiag = nullptr;
}

// the sorted index list of the model in LDS from the LDS copy of gamma
__device__ __forceinline__ void rebuild_g(Chain &ch, int kcap) {
  const int lane = ch.lane, p = ch.p;
  wave_sync();
  int k = 0;
  for (int base = 0; base < p; base += WAVE) {
    const int j = base + lane;
    const int inc = (j < p) ? (int)ch.gam[j] : 0;
    const unsigned long long mask = __ballot(inc != 0);
    const int slot = k + __popcll(mask & ((1ull << lane) - 1ull));
    if (inc && slot < kcap) ch.g[slot] = (uint16_t)j;
    k += __popcll(mask);
  }
  ch.k = k;
  wave_sync();
}

// Build everything about the model in ch.g (k <= kcap variables) into the HBM
// block `dst`: both factors (unless REUSE: they are there already), b_g, g,
// w = L_V^{-1} r, the scalars.  One wavefront.  BregVsSampler::
// set_reg_post_params + log_model_prob (BregVsSampler.cpp:216-239, :395-484).
// A real function (not inlined at its call sites): it is the rare path and its
// unrolled solves are large.
#ifdef BA_BSTAMPS
__device__ unsigned long long g_mstamp[16];
#define MST(i) do { const long long t_ = (long long)__builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_mstamp[i], (unsigned long long)(t_ - mst_)); mst_ = t_; } while (0)
#else
#define MST(i) do { } while (0)
#endif
#ifdef BA_BSTAMPS
__device__ unsigned long long g_bstamp[16];
#define BST(i) do { const long long t_ = (long long)__builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_bstamp[i], (unsigned long long)(t_ - bst_)); bst_ = t_; } while (0)
#else
#define BST(i) do { } while (0)
#endif
// part: -1 = this wavefront does all of it; 0 / 1 = the chain's two wavefronts share the
// factorisations (round 6): wave 1 (part 1) factors A_g and returns, wave 0 (part 0) factors
// V_g and does everything else; they meet at workgroup barriers -- around the one piece of LDS
// both need, the tile copy the diagonal blocks' inverses are taken from, and at the end, where
// wave 1 hands over A_g's verdict and log determinant through xch[BX_OKA], xch[BX_LDA].  Both
// take every early exit together (same k, same log prior).
enum : int { BX_K = 32, BX_SLOT = 33, BX_REUSE = 34, BX_OKA = 35, BX_LDA = 36 };
__device__ __forceinline__ void big_build_body(const BigP &P, Chain &ch, Model &M, double *dst,
                                               const BigCtx &bx, const bool REUSE, const int part = -1,
                                               lds_f64 *xch = nullptr) {
  const bool two = part >= 0;
  const int lane = ch.lane, p = ch.p, k = ch.k, kcap = bx.kcap;
#ifdef BA_BSTAMPS
  long long bst_ = (long long)__builtin_readcyclecounter();
#endif
  const SsvsScalarLayout &S = bx.S;
  M.bad = 0;
  M.pd = true;
  double lp;
  const double ldv_in = M.ldv, lda_in = M.lda;
  if (REUSE) {
    lp = M.lp;
  } else {
    double part = 0.0;
    for (int j = lane; j < p; j += WAVE) part += ch.gam[j] ? P.l1[j] : P.l0[j];
    lp = wave_sum(part);
    if (P.max_model_size >= 0 && k > P.max_model_size) lp = -BA_INF;
    if (!(lp > -BA_INF)) lp = -BA_INF;
  }
  M.lp = lp;
  M.ldv = M.lda = M.Q = M.c = 0.0;
  M.SS = ch.ss0q;
  if (k == 0) {
    M.logp = ch.mode ? lp : lp - (0.5 * ch.DF - 1.0) * log(ch.ss0q);
    return;
  }
  if (lp == -BA_INF) {
    M.logp = -BA_INF;
    M.pd = false;
    return;
  }
  const int npan = (k + 63) >> 6;
  const int kpad8 = (k + 7) & ~7;
  // index list, prior means; zero tails of the k-vectors
  if (part != 1)
  for (int m = lane; m < kcap; m += WAVE) {
    ((int *)(dst + S.g))[m] = (m < k) ? (int)ch.g[m] : 0;
    dst[S.bg + m] = (m < k) ? P.b[ch.g[m]] : 0.0;
    if (m >= k) {
      dst[S.w + m] = 0.0;
      if (!REUSE) { dst[S.rdv + m] = 0.0; dst[S.rda + m] = 0.0; }
      ch.w[m] = 0.0;
      ch.rdv[m] = 0.0;
    }
  }
  bool okv = true, oka = true;
  if (REUSE) {
    M.lda = lda_in;
    M.ldv = ldv_in;
    for (int m = lane; m < k; m += WAVE) ch.rdv[m] = dst[S.rdv + m];
  } else {
#pragma nounroll
    for (int s = 1; s >= 0; --s) {       // A first, then V (order of chol_blocks2's pair is immaterial)
      if (two && s != part) continue;
      const double *Mat = s ? P.A : P.V;
      const double msc = s ? ch.sa : ch.sv;
      double *Lst = dst + (s ? S.La : S.Lv);
      double *rdst = dst + (s ? S.rda : S.rdv);
      bool ok = true;
      double ld = 0.0;
      lds_f64 *colbuf = bx.tile + (part == 1 ? 128 : 0);   // (the columns' way round the lanes, per wave)
      // (no early exit from the loop: a two-wave build meets at the barriers of every panel)
      for (int I = 0; I < npan; ++I) {
        const int row = I * 64 + lane;
        const bool valid = row < k;
        const int gi = valid ? (int)ch.g[row] : 0;
        const double *Mrow = Mat + (size_t)gi * p;
        const int kk = (k - I * 64 < 64) ? (k - I * 64) : 64;
        double acc[64];
        double rdl = 0.0;
        bool did = false;
        if (ok) {
        if (I > 0) {
          c_f64 *LB = scalar_view(Lst);
          c_f64 *rd = (c_f64 *)((unsigned long long)LB + ((unsigned long long)rdst - (unsigned long long)Lst));
          big_solve(LB, rd, I * 64, I, true, bx.xs, lane,
                    [&](int J, double (&a)[64]) {
                      // (sixteen gathers at a time, all out before the first is used, and the
                      // compiler told so -- tools/isa_serial_loads.py: it hangs every load whose
                      // value goes through a per-lane select under a branch of its own, with a
                      // wait of its own, and a panel's gather was 64 round trips in a row)
#pragma unroll
                      for (int r0 = 0; r0 < 64; r0 += 16) {
                        double raw[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) raw[r] = Mrow[(int)ch.g[J * 64 + r0 + r]];  // (J < I: a full panel of real variables)
                        asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]),
                        "+v"(raw[8]), "+v"(raw[9]), "+v"(raw[10]), "+v"(raw[11]), "+v"(raw[12]), "+v"(raw[13]), "+v"(raw[14]), "+v"(raw[15]));
#pragma unroll
                        for (int r = 0; r < 16; ++r) a[r0 + r] = valid ? raw[r] * msc : 0.0;
                      }
                    },
                    [&](int, double (&)[64]) {});
        }
        BST(I > 0 ? 1 : 0);
        // diagonal tile: M[g_i, g_c] - sum_m x_i[m] x_c[m]
        // (columns in blocks of eight, only those the tile has: a chain that hovers just
        // above a multiple of 64 variables builds a last tile of one or two rows, and all
        // 64 columns of it cost 4 096 FMAs a tile row for nothing)
#pragma unroll
        for (int cb = 0; cb < 8; ++cb) {
          if (cb * 8 < kk) {
            double raw[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const int col = I * 64 + cb * 8 + c;
              // (the list's entries behind k are whatever an earlier model left: any valid index
              // will do -- a branch on col < k here made the eight loads eight round trips)
              const int gc = (int)ch.g[col];
              raw[c] = Mrow[gc < p ? gc : 0];
            }
            asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]));
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const int col = I * 64 + cb * 8 + c;
              acc[cb * 8 + c] = (valid && col <= row) ? raw[c] * msc : 0.0;
            }
          } else {
#pragma unroll
            for (int c = cb * 8; c < cb * 8 + 8; ++c) acc[c] = 0.0;
          }
        }
        if (I > 0) {
          c_f64 *XS = scalar_view(bx.xs);
          for (int m0 = 0; m0 < I * 64; m0 += 8) {
            double xi[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) xi[t] = bx.xs[(size_t)(m0 + t) * 64 + lane];
#pragma unroll
            for (int cb = 0; cb < 8; ++cb) {
              if (cb * 8 < kk) {
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                  for (int c = cb * 8; c < cb * 8 + 8; ++c) acc[c] -= xi[t] * XS[(size_t)(m0 + t) * 64 + c];
              }
            }
          }
        }
        BST(2);
        wave_sync();
        double ldt = 0.0;
        // (the tile's LDS holds the columns on their way round the lanes; the factor itself stays
        // in acc)
        chol_tile_regs(acc, colbuf, kk, lane, &ok, &ldt, &rdl);
        ld += ldt;
        wave_sync();
        BST(3);
        did = ok;
        }
        // what the table fills on the matrix cores multiply by (ssvs_fill_mfma.h): the
        // inverses of this tile's 16 x 16 diagonal blocks, from a copy of the tile in LDS
        const int nI_all = mf_block_rows(k), nI_here = (nI_all - 4 * I < 4) ? nI_all - 4 * I : 4;
        const bool need_inv = k <= MF_ROWS * MF_MAX_BLOCK_ROWS && nI_here > 0;
        auto inverses = [&]() {
          if (did && need_inv) {
#pragma unroll
            for (int c = 0; c < 64; ++c)
              if ((c >> 3) <= (lane >> 3)) bx.tile[bidx(lane, c)] = (valid && c <= lane) ? acc[c] : 0.0;
            bx.rdt[lane] = rdl;
            wave_sync();
            diag_inverses(bx.tile, bx.rdt, dst + (s ? S.ia : S.iv) + (size_t)(4 * I) * (MF_ROWS * MF_ROWS), kk, nI_here, lane);
            wave_sync();
          }
        };
        // rows of this tile row go to the factor: parked off-diagonal part, the tile, rd
        auto store = [&]() {
        if (did && row < kpad8) {
          for (int m0 = 0; m0 < I * 64; m0 += 8) {   // (eight at a time: a rolled copy waited for every load)
            double xv[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) xv[t] = bx.xs[(size_t)(m0 + t) * 64 + lane];
#pragma unroll
            for (int t = 0; t < 8; ++t) Lst[bidx(row, m0 + t)] = valid ? xv[t] : 0.0;
          }
#pragma unroll
          for (int c = 0; c < 64; ++c)
            if ((c >> 3) <= (lane >> 3))
              Lst[bidx(row, I * 64 + c)] = (valid && c <= lane && c < kk) ? acc[c] : 0.0;
          rdst[row] = valid ? rdl : 0.0;
        }
        if (did && s == 0 && valid) ch.rdv[row] = rdl;
        };
        if (two && need_inv) {
          // (the tile copy is the one piece of LDS the two waves share: first V's wave takes its
          // inverses while A's stores its rows, then the other way round; the third barrier
          // keeps the next panel's columns out of the tile until both are through with it)
          __syncthreads();
          if (part == 0) inverses(); else store();
          __syncthreads();
          if (part == 0) store(); else inverses();
          __syncthreads();
        } else {
          inverses();
          store();
        }
        BST(4);
      }
      if (s) { oka = ok; M.lda = 2.0 * ld; }
      else   { okv = ok; M.ldv = 2.0 * ld; }
    }
  }
  // the factor just stored is read back below (and by the tail) with vector loads
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  if (two && !REUSE) {
    if (part == 1 && lane == 0) { xch[BX_OKA] = oka ? 1.0 : 0.0; xch[BX_LDA] = M.lda; }
    __syncthreads();
    if (part == 1) return;
    oka = xch[BX_OKA] != 0.0;
    M.lda = xch[BX_LDA];
  }
  if (!okv) {
    M.pd = false;
    M.logp = -BA_INF;
    return;
  }
  BST(5);
  BST(6);
  // r = A_g b_g + xty_g, c = b_g' A_g b_g (only non-zero prior means cost)
  double cpart = 0.0;
  {
    bool anyb = false;
    for (int m = lane; m < k; m += WAVE) anyb = anyb || (P.b[ch.g[m]] != 0.0);
    const bool any = __any(anyb) != 0;
    for (int m = lane; m < kcap; m += WAVE) {
      double r = 0.0;
      if (m < k) {
        const int gm = ch.g[m];
        double ab = 0.0;
        if (any) {
          for (int n = 0; n < k; ++n) {
            const int gn = ch.g[n];
            const double bn = P.b[gn];
            if (bn != 0.0) ab += (P.A[(size_t)gm * p + gn] * ch.sa) * bn;
          }
          cpart += P.b[gm] * ab;
        }
        r = ab + ch.xty[gm] * ch.sx;
      }
      bx.y[m] = r;
    }
  }
  M.c = wave_sum(cpart);
  wave_sync();
  // w = L_V^{-1} r, tile row by tile row: t_i = r_i - sum_{m < 64 I} L[i, m] w_m, then
  // the forward substitution inside the tile (lane = local row)
  const double *Lv = dst + S.Lv;
  __builtin_amdgcn_s_waitcnt(0);
  double qpart = 0.0;
  for (int I = 0; I < npan; ++I) {
    const int row = I * 64 + lane;
    const bool valid = row < k;
    double t = valid ? bx.y[row] : 0.0;
    if (valid)
      for (int m0 = 0; m0 < I * 64; m0 += 8) {
        double lv8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) lv8[u] = Lv[bidx(row, m0 + u)];
#pragma unroll
        for (int u = 0; u < 8; ++u) t -= lv8[u] * ch.w[m0 + u];
      }
    const double rdm = valid ? ch.rdv[row] : 0.0;
    const int kk = (k - I * 64 < 64) ? (k - I * 64) : 64;
    // (the lane's row of the tile sixteen columns at a time, every load unconditional -- a lane
    // reads its own diagonal where the column lies above it -- and out before the first is used:
    // as 64 loads each under the condition "below the diagonal" they were 64 round trips in a
    // row, and the array that held them went to scratch memory)
    const int rowc = valid ? row : I * 64;
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += 16) {
      if (j0 < kk) {
        double raw[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int lr = rowc - I * 64;
          raw[u] = Lv[bidx(rowc, I * 64 + ((j0 + u < lr) ? j0 + u : lr))];
        }
        asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]),
                          "+v"(raw[8]), "+v"(raw[9]), "+v"(raw[10]), "+v"(raw[11]), "+v"(raw[12]), "+v"(raw[13]), "+v"(raw[14]), "+v"(raw[15]));
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int j = j0 + u;
          if (j < kk) {
            const double wj = bcast_u(t, j) * bcast_u(rdm, j);
            if (lane == j) t = wj;
            else if (lane > j && valid) t -= raw[u] * wj;
          }
        }
      }
    }
    if (valid) {
      ch.w[row] = t;
      dst[S.w + row] = t;
      qpart += t * t;
    }
    wave_sync();
  }
  M.Q = wave_sum(qpart);
  BST(7);
  M.SS = ch.ss0q + M.c - M.Q;
  if (ch.mode) {
    if (!oka) { M.lda = -BA_INF; M.logp = -BA_INF; return; }
    M.logp = lp + 0.5 * (M.lda - M.ldv) - 0.5 * (M.c - M.Q);
    return;
  }
  if (!(M.SS >= 0.0) || isinf(M.SS)) {
    M.bad = CHAIN_NEGATIVE_SS;
    M.logp = -BA_INF;
    return;
  }
  if (!oka) { M.lda = -BA_INF; M.logp = -BA_INF; return; }
  M.logp = lp + 0.5 * (M.lda - M.ldv) - (0.5 * ch.DF - 1.0) * log(M.SS);
}

__device__ __forceinline__ void store_scalars(double *dst, const SsvsScalarLayout &S, const Model &M, int lane) {
  if (lane == 0) {
    double *sc = dst + S.scal;
    sc[0] = M.logp; sc[1] = M.lp; sc[2] = M.ldv; sc[3] = M.lda;
    sc[4] = M.Q; sc[5] = M.c; sc[6] = M.SS; sc[7] = M.pd ? 1.0 : 0.0;
  }
}
__device__ __forceinline__ void load_scalars(const double *src, const SsvsScalarLayout &S, Model &M) {
  const double *sc = src + S.scal;
  M.logp = sc[0]; M.lp = sc[1]; M.ldv = sc[2]; M.lda = sc[3];
  M.Q = sc[4]; M.c = sc[5]; M.SS = sc[6]; M.pd = sc[7] != 0.0;
  M.bad = 0;
}

// this lane's proposal "flip j" against the current model (factors through sc)
template <bool NAT>
__device__ __forceinline__ Proposal big_eval(const BigP &P, Chain &ch, const Model &M,
                                             const BigCtx &bx, c_f64 *sc, int j, bool valid, int jbase) {
  const int p = ch.p, k = ch.k, lane = ch.lane;
  const SsvsScalarLayout &S = bx.S;
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
  if (empty_after) out.logp = ch.mode ? lpn : lpn - (0.5 * ch.DF - 1.0) * log(ch.ss0q);
  const double vjj = (fast && add) ? (P.vdiag ? P.vdiag[j] : P.V[(size_t)j * p + j]) * ch.sv : 0.0;
  const double ajj = (fast && add) ? P.A[(size_t)j * p + j] * ch.sa : 0.0;
  const double xtyj = (fast && add) ? ch.xty[j] * ch.sx : 0.0;
  const int npan = (k + 63) >> 6;
  double nv = 0.0, dv = 0.0, na = 0.0, ab = 0.0;
  if (NAT && k <= MF_ROWS * MF_MAX_BLOCK_ROWS) {
    // a fill round of a model of at most 128 variables: the wavefront's 64 proposals
    // jbase + lane together, on the matrix cores
    const MfSums z = mf_proposal_sums<8>(P.V, P.A, p, ch.sv, ch.sa, ch.sc_store, S, S.iv, S.ia, ch.g, k, jbase,
                                      (fast ? 1 : 0) | (add ? 2 : 0), lane);
    nv = z.nv; dv = z.dv; na = z.na; ab = z.ab;
  } else
#pragma nounroll
  for (int s = 0; s < 2; ++s) {
    const double *Mat = s ? P.A : P.V;
    const double msc = s ? ch.sa : ch.sv;
    c_f64 *LB = sc + (s ? S.La : S.Lv);
    c_f64 *rd = sc + (s ? S.rda : S.rdv);
    double n2 = 0.0, dw = 0.0, abl = 0.0;
    big_solve(LB, rd, k, npan, false, bx.xs, lane,
              [&](int I, double (&a)[64]) {
#pragma unroll
                for (int r0 = 0; r0 < 64; r0 += 16) {
                  // (16 gathers in flight at a time: their addresses need registers too; all of
                  // them out before the first is used, see the build's panels)
                  int gms[16];
                  double raw[16];
#pragma unroll
                  for (int r = 0; r < 16; ++r) {
                    const int m = I * 64 + r0 + r;
                    const int gr = (int)ch.g[m];   // (behind k: any valid index, see the build)
                    gms[r] = gr < p ? gr : 0;
                    raw[r] = NAT ? Mat[(size_t)gms[r] * p + j] : Mat[(size_t)j * p + gms[r]];
                  }
                  asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]), "+v"(raw[7]),
                  "+v"(raw[8]), "+v"(raw[9]), "+v"(raw[10]), "+v"(raw[11]), "+v"(raw[12]), "+v"(raw[13]), "+v"(raw[14]), "+v"(raw[15]));
#pragma unroll
                  for (int r = 0; r < 16; ++r) {
                    const int m = I * 64 + r0 + r;
                    const double e = (gms[r] == j) ? 1.0 : 0.0;
                    a[r0 + r] = (fast && m < k) ? (add ? raw[r] * msc : e) : 0.0;
                    abl += a[r0 + r] * sc[S.bg + m];
                  }
                }
              },
              [&](int I, double (&a)[64]) {
#pragma unroll
                for (int r = 0; r < 64; ++r) {
                  n2 += a[r] * a[r];
                  dw += a[r] * sc[S.w + I * 64 + r];
                }
              });
    if (s == 0) { nv = n2; dv = dw; }
    else        { na = n2; ab = abl; }
  }
  if (fast) {
    double ldv, lda, Q;
    bool ok = true;
    if (add) {
      const double d2 = vjj - nv;
      const double da2 = ajj - na;
      if (!(d2 > 0.0) || !(da2 > 0.0)) ok = false;
      const double rj = xtyj + ab;
      const double wn = (rj - dv) / sqrt(d2);
      Q = M.Q + wn * wn;
      ldv = M.ldv + log(d2);
      lda = M.lda + log(da2);
    } else {
      Q = M.Q - dv * dv / nv;
      ldv = M.ldv + log(nv);
      lda = M.lda + log(na);
    }
    if (ok && ch.mode) {
      out.logp = lpn + 0.5 * (lda - ldv) - 0.5 * (M.c - Q);
    } else if (ok) {
      const double SS = ch.ss0q + M.c - Q;
      if (!(SS >= 0.0) || isinf(SS)) out.bad_ss = true;
      else out.logp = lpn + 0.5 * (lda - ldv) - (0.5 * ch.DF - 1.0) * log(SS);
    }
  }
  return out;
}

}  // namespace

// grid = chains, block = 128.  Only chains parked with CHAIN_MODEL_TOO_LARGE
// are this kernel's; it runs the sweeps they are owed (todo) and hands them back
// with status CHAIN_OK (or parks them again if even big_kcap is too small).
//
// Both wavefronts run ONE loop: the master (wave 0) advances its state machine
// until it needs something the workgroup does together -- a table-fill round,
// the shuffle uniforms -- or something big enough to exist only once in the
// code (a model build); it posts that as a command, the waves meet at a barrier,
// do it, and meet again.  So every heavy routine has a single call site.
__global__ __launch_bounds__(128, 2) void ssvs_big_kernel(SsvsParams P, int nsweeps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int W = 2;
  const int chain = (int)blockIdx.x + P.chain_first;
  int lane = threadIdx.x & (WAVE - 1);   // (not const: made opaque once per pass of the phase machine, below)
  const int wave = uni((int)(threadIdx.x >> 6));
  const int p = P.p, kcap = P.big_kcap;
  if ((int)blockIdx.x >= P.chain_count) return;
  const int status_in = P.status[chain];
  __syncthreads();
  if (status_in != CHAIN_MODEL_TOO_LARGE) return;
  int owed_after = 0;
  {
    int total = nsweeps + P.todo[chain];
    if (P.run_limit > 0 && total > P.run_limit) {
      owed_after = total - P.run_limit;
      total = P.run_limit;
    }
    nsweeps = total;
  }
  __syncthreads();

  const SsvsBigLds lay = ssvs_big_lds_layout(p, kcap);
  Chain ch;
  ch.lane = lane;
  ch.p = p;
  ch.k = 0;
  ch.Lv = ch.La = nullptr;
  ch.rda = nullptr;
  ch.bg = nullptr;
  ch.rdv = to_lds<double>(smem + lay.rd);
  ch.w = to_lds<double>(smem + lay.w);
  ch.g = to_lds<uint16_t>(smem + lay.g);
  ch.perm = to_lds<uint16_t>(smem + lay.perm0);
  ch.perm_alt = to_lds<uint16_t>(smem + lay.perm1);
  ch.oth = to_lds<uint16_t>(smem + lay.oth);
  ch.last = to_lds<uint32_t>(smem + lay.last);
  ch.pred = to_lds<uint16_t>(smem + lay.pred);
  ch.gam = to_lds<uint8_t>(smem + lay.gam);
  ch.gam0 = to_lds<uint8_t>(smem + lay.gam0);
  ch.nbr = to_lds<uint8_t>(smem + lay.nbr);
  lds_f64 *ctl = to_lds<double>(smem + lay.ctrl);
  BigCtx bx;
  bx.kcap = kcap;
  bx.S = ssvs_scalar_layout(kcap);
  bx.xs = P.big_xs + ((size_t)chain * W + wave) * (size_t)kcap * 64;
  bx.tile = to_lds<double>(smem + lay.tile);
  bx.rdt = to_lds<double>(smem + lay.rdt);
  bx.y = to_lds<double>(smem + lay.y);
  bx.bst = to_lds<double>(smem + lay.bst);
  bx.gst = to_lds<uint16_t>(smem + lay.gst);
  BigP BP;
  BP.V = P.V + (size_t)chain * (size_t)P.v_chain_stride;   // (the logit sampler: every chain has its own V)
  BP.A = P.A; BP.b = P.b; BP.l1 = P.l1; BP.l0 = P.l0;
  BP.vdiag = nullptr;
  if (P.col_valid) {
    P.col_valid += (size_t)chain * (size_t)P.col_words;
    BP.vdiag = P.v_diag + (size_t)chain * (size_t)p;
  }
  BP.max_model_size = P.max_model_size;
  ch.xty = P.xty + (size_t)chain * P.xty_stride;
  const double yty = P.yty[(size_t)chain * P.suf_stride];
  const double nobs = P.nobs[(size_t)chain * P.suf_stride];
  ch.DF = nobs + P.prior_df;
  ch.ss0q = P.prior_ss + yty;
  ch.mode = P.mode;
  ch.sv = ch.sa = ch.sx = 1.0;
  if (P.mode) {
    const double inv = 1.0 / P.sigsq[chain];
    ch.sx = inv;
    if (P.slab_scales) { ch.sv = inv; ch.sa = inv; }
  }
  const PhiloxKey key{P.seed_lo, P.seed_hi, (uint32_t)(P.chain_offset + chain), P.stream};
  auto slot_block = [&](int slot) {
    return P.big_model + ((size_t)slot * P.chains + chain) * P.big_model_stride;
  };
  auto bind = [&](int slot) {
    const size_t c = (size_t)slot * P.chains + chain;
    ch.tab_lp = P.table_lp + c * p;
    ch.tab_kind = P.table_kind + c * p;
    ch.sc_store = slot_block(slot);
  };

  enum : int { BCMD_NONE = -1, BCMD_BUILD = 3 };
  enum : int { PH_INIT, PH_BEGIN, PH_SHUFFLED, PH_FLIPS, PH_SWAP, PH_TAIL, PH_ADA };
  enum : int { BR_INIT, BR_VALID, BR_TRY };

  // ---- master-only state (wave 1 carries it along unused) ---------------------
  uint8_t *g_gamma = P.gamma + (size_t)chain * p;
  uint16_t *g_perm = P.perm + (size_t)chain * p;
  int status = CHAIN_OK;
  bool aborted = false;
  int need_col = -1;   // CHAIN_NEED_COLUMN_BIG: the variable whose vector of V is missing
  int kmax = 0, trace_at = 0, failures = 0, done = 0, klast = 0;
  uint64_t pos = 0, flip_pos = 0, pos0 = 0;
  double sigsq = 1.0;
  bool beta_valid = false;
  const int nflips = P.max_flips;
  const int big_tag = 0x40000000 | (kcap << 1);
  Model M;
  M.bad = 0; M.pd = true; M.logp = 0; M.lp = 0; M.ldv = 0; M.lda = 0; M.Q = 0; M.c = 0; M.SS = 0;
  int cur = 0, other_var = -1;
  bool other_ok = false, table_valid = false, table_valid_other = false;
  bool model_checked = false;
  int phase = PH_INIT, sweep = 0, i0 = 0, fill_base = 0;
  // a pending build: reason, where, and (BR_TRY) the decision riding on it
  int b_reason = BR_INIT, b_slot = 0, b_after = PH_BEGIN;
  bool b_reuse = false;
  int t_f1 = -1, t_f2 = -1, t_kind = 0;
  double t_lu = 0.0, t_lfw = 0.0, t_lrev = 0.0;
  WinRng rng;
  rng.init(key, lane, 0);
  // ---- the adaptive sampler's moves (AdaptiveSpikeSlabRegressionSampler.cpp:62-225, see
  // ssvs_adaptive_kernel.hip) on a model of more than 64 variables: the rates and
  // their cumulative sums live in HBM, a move's candidate model is read off the table
  const bool adaptive = P.adaptive != 0;
  double *g_birth = adaptive ? P.ada_birth + (size_t)chain * p : nullptr;
  double *g_death = adaptive ? P.ada_death + (size_t)chain * p : nullptr;
  double *cumb = adaptive ? P.ada_ws + (size_t)chain * 4 * (size_t)p : nullptr;
  double *cumd = cumb + p, *birth0 = cumb + 2 * (size_t)p, *death0 = cumb + 3 * (size_t)p;
  const int ada_flips = (P.ada_max_flips < p) ? P.ada_max_flips : p;
  uint64_t iteration = 0;
  int ada_i = 0;
  bool cum_dirty = true, t_birth = false, rates_saved = false;
  double Bsum = 0.0, Dsum = 0.0;

#define BACC_ADD(slot, x) do { if (lane == 0) ctl[CT_ACC + (slot)] += (double)(x); } while (0)
#define BACC_MIN(x) do { if (lane == 0) ctl[CT_ACC + ACC_MIN_MARGIN] = fmin(ctl[CT_ACC + ACC_MIN_MARGIN], (x)); } while (0)
  auto publish_ctl = [&]() {
    if (lane == 0) {
      ctl[CT_LOGP] = M.logp; ctl[CT_LP] = M.lp; ctl[CT_LDV] = M.ldv;
      ctl[CT_LDA] = M.lda; ctl[CT_Q] = M.Q; ctl[CT_C] = M.c;
    }
    wave_sync();
  };
  auto reload_tail_vectors = [&]() {  // w, 1 / diag(L_V) of the model in the bound slot
    __builtin_amdgcn_s_waitcnt(0);
    for (int m = lane; m < kcap; m += WAVE) {
      ch.w[m] = ch.sc_store[bx.S.w + m];
      ch.rdv[m] = ch.sc_store[bx.S.rdv + m];
    }
    wave_sync();
  };
  auto flip_in_lds = [&](int f1, int f2) {
    wave_sync();
    if (lane == 0) {
      ch.gam[f1] = (uint8_t)!ch.gam[f1];
      if (f2 >= 0) ch.gam[f2] = (uint8_t)!ch.gam[f2];
    }
    rebuild_g(ch, kcap);
  };
  auto request_try = [&](int f1, int f2, int kind, double lu, double lfw, double lrev, int after) {
    t_f1 = f1; t_f2 = f2; t_kind = kind; t_lu = lu; t_lfw = lfw; t_lrev = lrev;
    flip_in_lds(f1, f2);
    b_reason = BR_TRY; b_slot = cur ^ 1; b_reuse = false; b_after = after;
  };

  if (wave == 0) {
    for (int j = lane; j < p; j += WAVE) {
      ch.gam[j] = g_gamma[j];
      ch.perm[j] = g_perm[j];
      ch.nbr[j] = (uint8_t)((P.cm_start != nullptr) && (P.cm_start[j + 1] > P.cm_start[j]));
    }
    rebuild_g(ch, kcap);
    if (ch.k > kcap) status = CHAIN_MODEL_TOO_LARGE;  // parked again: the host grows big_kcap
    kmax = ch.k;
    trace_at = P.trace_idx ? P.trace_idx[chain] : 0;
    pos = uni((uint64_t)P.rng_pos[chain]);
    pos0 = pos;
    failures = uni((int)P.failures[chain]);
    sigsq = uni((double)P.sigsq[chain]);
    if (adaptive) iteration = uni((uint64_t)P.ada_iter[chain]);
    rng.init(key, lane, pos);
    if (lane < 16) ctl[CT_ACC + lane] = (lane == ACC_MIN_MARGIN || (adaptive && lane == ACC_PHASE0)) ? BA_INF : 0.0;
    wave_sync();
  }

#ifdef BA_BSTAMPS
  // diagnostic build: where the master's time goes, by command (tools/big_phases.py)
  double bph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long bt0 = (long long)__builtin_readcyclecounter();
#endif
  for (;;) {
    // (what the compiler derives from the lane number once before this loop -- dozens of per-lane
    // addresses -- it keeps in scratch memory for the whole launch; opaque per pass, they are
    // recomputed where they are used: ssvs_sweep_body.h does the same)
    asm volatile("" : "+v"(lane));
    ch.lane = lane;
#ifdef BA_BSTAMPS
    const int bphase_at_entry = phase;
    long long mst_ = (long long)__builtin_readcyclecounter();
#endif
    if (wave == 0) {
      int cmd = BCMD_NONE;
      while (cmd == BCMD_NONE) {
        if (status != CHAIN_OK) { cmd = BCMD_EXIT; break; }
        if (phase == PH_INIT) {
          if (nsweeps <= 0) { cmd = BCMD_EXIT; break; }
          // the current model: kept from the chain's last launch, or built now
          const int tag_in = P.model_keep ? P.model_tag[chain] : 0;
          const bool kept = (tag_in & ~1) == big_tag;
          phase = PH_BEGIN;
          if (kept) {
            cur = tag_in & 1;
            bind(cur);
            load_scalars(ch.sc_store, bx.S, M);
            reload_tail_vectors();
            if (!P.suf_changed) {
              table_valid = P.table_keep && P.table_tag[chain] == tag_in;
              publish_ctl();
              continue;
            }
          }
          bind(cur);
          b_reason = BR_INIT; b_slot = cur; b_reuse = kept; b_after = PH_BEGIN;
          cmd = BCMD_BUILD;
          break;
        }
        if (phase == PH_BEGIN) {
          if (sweep >= nsweeps) { cmd = BCMD_EXIT; break; }
          pos0 = pos;
          if (adaptive) {
            // ---- the sweep's birth / death moves; what an aborted sweep has to undo is saved first
            if (ada_flips <= 0) { phase = PH_TAIL; continue; }
            for (int j = lane; j < p; j += WAVE) {
              ch.gam0[j] = ch.gam[j];
              birth0[j] = g_birth[j];
              death0[j] = g_death[j];
            }
            wave_sync();
            rates_saved = true;
            ada_i = 0;
            cum_dirty = true;
            phase = PH_ADA;
            continue;
          }
          // ---- draw_model_indicators (BregVsSampler.cpp:353-378)
          if (nflips <= 0) { phase = PH_SWAP; continue; }
          for (int j = lane; j < p; j += WAVE) {
            ch.gam0[j] = ch.gam[j];
            if (P.mode) ch.perm[j] = (uint16_t)j;
          }
          wave_sync();
          if (lane == 0) ((AS_LDS uint64_t *)(ctl + CT_POS))[0] = pos;
          phase = PH_SHUFFLED;
          cmd = BCMD_UNIF;
          break;
        }
        if (phase == PH_SHUFFLED) {
          StampCtx sx;
          sx.last = 0;
          if (P.mode == 2) {
            // BinomialLogitSpikeSlabSampler's shuffle (see ssvs_kernel.hip): the steps run in order
            if (lane == 0 && p > 1) {
              for (int i = 0; i < p; ++i) {
                const int j = ch.oth[i];
                const uint16_t a = ch.perm[i];
                ch.perm[i] = ch.perm[j];
                ch.perm[j] = a;
              }
            }
            wave_sync();
          } else if (p > 1) parallel_shuffle<true>(ch, sx);
          MST(0);
          flip_pos = pos + (uint64_t)(P.mode == 2 ? p : (p > 0 ? p - 1 : 0));
          pos = flip_pos + (uint64_t)nflips;
          i0 = 0;
          phase = PH_FLIPS;
          if (!model_checked) {
            model_checked = true;
            if (!(M.logp > -BA_INF && M.logp < BA_INF)) {
              // VariableSelectionPrior::make_valid (VariableSelectionPrior.cpp:287-300)
              for (int j = 0; j < p; ++j) {
                const double pj = P.pi[j];
                const bool inc = ch.gam[j];
                if ((pj <= 0.0 && inc) || (pj >= 1.0 && !inc)) {
                  if (!inc && column_missing(P, j)) { status = CHAIN_NEED_COLUMN_BIG; need_col = j; aborted = true; break; }
                  wave_sync();
                  if (lane == 0) ch.gam[j] = (uint8_t)!inc;
                  wave_sync();
                }
              }
              if (aborted) continue;
              rebuild_g(ch, kcap);
              if (ch.k > kcap) { status = CHAIN_MODEL_TOO_LARGE; aborted = true; continue; }
              bind(cur);
              b_reason = BR_VALID; b_slot = cur; b_reuse = false; b_after = PH_FLIPS;
              cmd = BCMD_BUILD;
              break;
            }
          }
          continue;
        }
        if (phase == PH_FLIPS) {
          if (i0 >= nflips) { phase = PH_SWAP; continue; }
          if (!table_valid) {
            // (re)build the table of acceptance thresholds for the current model:
            // p / 128 rounds of the whole workgroup
            if (fill_base >= p) {
              fill_base = 0;
              table_valid = true;
              continue;
            }
            if (lane == 0) {
              ctl[CT_K] = (double)ch.k;
              ctl[CT_I0] = (double)fill_base;
              ctl[CT_CUR] = (double)cur;
            }
            fill_base += WAVE * W;
            cmd = BCMD_EVAL;
            break;
          }
          DecideResult dr;
          MST(1);
          decide_walk(ch, key, flip_pos, i0, nflips, dr);
          MST(2);
          BACC_MIN(dr.margin);
          if (dr.spos < 0) {
            BACC_ADD(ACC_PROPOSALS, nflips - i0);
            i0 = nflips;
            continue;
          }
          BACC_ADD(ACC_PROPOSALS, dr.spos + 1 - i0);
          i0 = dr.spos + 1;
          if (dr.kind == STOP_BAD) { status = CHAIN_NEGATIVE_SS; continue; }
          if (!ch.gam[dr.j] && ch.k >= kcap) { status = CHAIN_MODEL_TOO_LARGE; aborted = true; continue; }
          if (!ch.gam[dr.j] && column_missing(P, dr.j)) { status = CHAIN_NEED_COLUMN_BIG; need_col = dr.j; aborted = true; continue; }
          if (dr.kind == 0 && other_ok && dr.j == other_var) {
            // the flip leads back to the model the other slot still holds
            flip_in_lds(dr.j, -1);
            cur ^= 1;
            bind(cur);
            __builtin_amdgcn_s_waitcnt(0);
            load_scalars(ch.sc_store, bx.S, M);
            reload_tail_vectors();
            const bool t = table_valid;
            table_valid = table_valid_other;
            table_valid_other = t;
            publish_ctl();
            BACC_ADD(ACC_ACCEPTS, 1);
            BACC_ADD(ACC_SLOT_HITS, 1);
            if (!M.pd) status = CHAIN_NOT_PD;
            continue;
          }
          if (dr.kind == 0) request_try(dr.j, -1, 0, 0.0, 0.0, 0.0, PH_FLIPS);
          else              request_try(dr.j, -1, 1, dr.logu, 0.0, 0.0, PH_FLIPS);
          cmd = BCMD_BUILD;
          break;
        }
        if (phase == PH_ADA) {
          if (ada_i >= ada_flips) { phase = PH_TAIL; continue; }
          if (!table_valid) {
            // log_model_prob of every single-variable change of the current model
            if (fill_base >= p) {
              fill_base = 0;
              table_valid = true;
              continue;
            }
            if (lane == 0) {
              ctl[CT_K] = (double)ch.k;
              ctl[CT_I0] = (double)fill_base;
              ctl[CT_CUR] = (double)cur;
            }
            fill_base += WAVE * W;
            cmd = BCMD_EVAL;
            break;
          }
          if (cum_dirty) {
            // cumulative rates of the candidates of either move, in variable order
            double cb = 0.0, cd = 0.0;
            __builtin_amdgcn_s_waitcnt(0);
            for (int base = 0; base < p; base += WAVE) {
              const int j = base + lane;
              const bool in = j < p;
              const bool inc = in && ch.gam[j];
              const double wb = (in && !inc) ? g_birth[j] : 0.0;
              const double wd = inc ? g_death[j] : 0.0;
              const double pb = big_prefix(wb), pd = big_prefix(wd);
              if (in) { cumb[j] = cb + pb; cumd[j] = cd + pd; }
              cb += bcast_u(pb, 63);
              cd += bcast_u(pd, 63);
            }
            Bsum = cb;
            Dsum = cd;
            cum_dirty = false;
            __builtin_amdgcn_s_waitcnt(0);
            wave_sync();
          }
          // the next (up to) 64 moves, one per lane, as long as none is accepted
          const int kk = ch.k;
          const bool edge = (kk == 0 || kk == p);
          const int nb = edge ? 1 : ((ada_flips - ada_i < WAVE) ? ada_flips - ada_i : WAVE);
          const bool valid = lane < nb;
          const uint64_t mypos = pos + 3ull * (uint64_t)lane;
          const double u0 = philox_uniform(key, mypos);
          const bool isbirth = u0 < .5;
          const bool possible = isbirth ? (kk < p) : (kk > 0);
          const double u1 = philox_uniform(key, mypos + 1);
          const double u2 = philox_uniform(key, mypos + 2);
          const double tot = isbirth ? Bsum : Dsum;
          const double *cum = isbirth ? cumb : cumd;
          const double tmp = 0.0 + (tot - 0.0) * u1;
          int lo = 0, hi = p - 1;
          if (valid && possible) {
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (cum[mid] >= tmp) hi = mid; else lo = mid + 1;
            }
          }
          int j = lo;
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
          const double plogp = cand_ok ? ch.tab_lp[j] : 0.0;
          const int pkind = cand_ok ? (int)ch.tab_kind[j] : 0;
          const double wj = cand_ok ? (isbirth ? g_birth[j] : g_death[j]) : 1.0;
          const double bj = cand_ok ? (isbirth ? g_death[j] : g_birth[j]) : 1.0;
          const double fwd = log(wj / tot);
          const double rev = log(bj / ((isbirth ? Dsum : Bsum) + bj));
          const double logu = log(u2);
          const double ratio = (plogp - fwd) - (M.logp - rev);
          const bool slow = cand_ok && pkind == STOP_SLOW;
          const bool bad = cand_ok && pkind == STOP_BAD;
          const bool accept = cand_ok && !slow && !bad && (logu < ratio);
          const unsigned long long m_acc = __ballot(accept), m_slow = __ballot(slow),
                                   m_bad = __ballot(bad), m_brk = __ballot(broken_multi);
          const unsigned long long m_stop = m_acc | m_slow | m_bad | m_brk;
          const int f = m_stop ? (__ffsll((long long)m_stop) - 1) : WAVE;
          const bool counted = cand_ok && (lane < f || (lane == f && ((m_acc >> f) & 1ull)));
          const double mg = (counted && plogp > -BA_INF && M.logp > -BA_INF) ? fabs(logu - ratio) : BA_INF;
          {
            const double wm = wave_min(mg), wmm = wave_min(counted ? mmargin : BA_INF);
            BACC_MIN(wm);
            if (lane == 0) ctl[CT_ACC + ACC_PHASE0] = fmin(ctl[CT_ACC + ACC_PHASE0], wmm);
          }
          if (f == WAVE) {
            BACC_ADD(ACC_PROPOSALS, nb);
            ada_i += nb;
            pos += edge ? (uni((int)possible) ? 3ull : 1ull) : 3ull * (uint64_t)nb;
            continue;
          }
          BACC_ADD(ACC_PROPOSALS, f + 1);
          ada_i += f + 1;
          pos += 3ull * (uint64_t)(f + 1);
          if ((m_brk >> f) & 1ull) { status = CHAIN_RNG_BRANCH; continue; }
          if ((m_bad >> f) & 1ull) { status = CHAIN_NEGATIVE_SS; continue; }
          const int jf = bcast_u(j, f);
          t_birth = bcast_u((int)isbirth, f) != 0;
          if (t_birth && ch.k >= kcap) { status = CHAIN_MODEL_TOO_LARGE; aborted = true; continue; }
          // the candidate model is built in the other slot; a variable with a non-zero
          // prior mean (the exact path) decides only then
          request_try(jf, -1, ((m_slow >> f) & 1ull) ? 2 : 0, bcast_u(logu, f), bcast_u(fwd, f),
                      bcast_u(rev, f), PH_ADA);
          cmd = BCMD_BUILD;
          break;
        }
        if (phase == PH_SWAP) {
          MST(3);
          // ---- attempt_swap (BregVsSampler.cpp:277-310)
          phase = PH_TAIL;
          rng.set_pos(pos);
          if (nflips > 0) {
            Pending pe;
            pe.kind = EV_NONE; pe.f1 = pe.f2 = -1; pe.lu = 0; pe.lfw = pe.lrev = 0; pe.check_legal = false;
            propose_swap(P, ch, rng, pe, &status);
            pos = rng.get_pos();
            if (status != CHAIN_OK) continue;
            if (pe.kind == EV_TRY_LT) {  // (a swap keeps the model size)
              request_try(pe.f1, pe.f2, 2, pe.lu, pe.lfw, pe.lrev, PH_TAIL);
              cmd = BCMD_BUILD;
              break;
            }
          }
          continue;
        }
        MST(4);
        // ---- PH_TAIL: draw_sigma (BregVsSampler.cpp:313-324)
        const int k = ch.k;
        rng.set_pos(pos);
        if (P.draw_sigma) {
          int bad = 0;
          const double DF = (k == 0) ? ch.DF : ((ch.DF - P.prior_df) + P.prior_df);
          const double SS = (k == 0) ? ch.ss0q : ((M.SS - P.prior_ss) + P.prior_ss);
          sigsq = uni(d_draw_variance(rng, DF, SS, P.sigma_max, &bad));
          if (bad) { status = CHAIN_RNG_BRANCH; continue; }
        }
        pos = uni(rng.get_pos());
        MST(5);
        // ---- draw_beta (BregVsSampler.cpp:326-351): beta = L^{-T}(w + sigma z)
        if (P.draw_beta && k > 0) {
          if (!M.pd) { ++failures; status = CHAIN_NOT_PD; continue; }
          failures = 0;
          const double sigma = P.mode ? 1.0 : sqrt(sigsq);
          const int npan = (k + 63) >> 6;
          for (int q = 0; q < npan; ++q) {
            const int kq = (k - q * 64 < 64) ? (k - q * 64) : 64;
            const double z = draw_normals(rng, kq);
            if (lane < kq) bx.y[q * 64 + lane] = ch.w[q * 64 + lane] + sigma * z;
          }
          pos = uni(rng.get_pos());
          wave_sync();
          const double *Lv = ch.sc_store + bx.S.Lv;
          for (int I = npan - 1; I >= 0; --I) {
            const int col = I * 64 + lane;            // this lane's unknown in the tile
            const int kk = (k - I * 64 < 64) ? (k - I * 64) : 64;
            double yv = (lane < kk) ? bx.y[col] : 0.0;
            const double rdm = (lane < kk) ? ch.rdv[col] : 0.0;
            // column `col` of the tile below the diagonal, 16 rows at a time
#pragma nounroll
            for (int ib = 48; ib >= 0; ib -= 16) {
              if (ib < kk) {
                double lt[16];
#pragma unroll
                for (int t = 0; t < 16; ++t)
                  lt[t] = (ib + t > lane && ib + t < kk) ? Lv[bidx(I * 64 + ib + t, col)] : 0.0;
#pragma unroll
                for (int t = 15; t >= 0; --t) {
                  const int i = ib + t;
                  if (i < kk) {
                    const double xi = bcast_u(yv * rdm, i);
                    if (lane == i) yv = xi;
                    else if (lane < i) yv -= lt[t] * xi;
                  }
                }
              }
            }
            if (lane < kk) bx.y[col] = yv;
            wave_sync();
            // earlier tiles: y_J[c] -= sum_i L[64 I + i][64 J + c] x_I[i]
            for (int J = 0; J < I; ++J) {
              double acc = 0.0;
              for (int ii = 0; ii < kk; ii += 8) {
                double l8[8];
#pragma unroll
                for (int t = 0; t < 8; ++t)
                  l8[t] = (ii + t < kk) ? Lv[bidx(I * 64 + ii + t, J * 64 + lane)] : 0.0;
#pragma unroll
                for (int t = 0; t < 8; ++t) acc += l8[t] * bx.y[I * 64 + ((ii + t < kk) ? ii + t : 0)];
              }
              bx.y[J * 64 + lane] -= acc;
            }
            wave_sync();
          }
          beta_valid = true;
        } else if (P.draw_beta) {
          beta_valid = true;
        }
        MST(6);
        // ---- summaries, traces, the draw record
        kmax = k > kmax ? k : kmax;
        for (int m = lane; m < k; m += WAVE) {
          // (three loads, then three stores: as three read-modify-writes in a row they were
          // three memory round trips a sweep)
          const size_t o = (size_t)chain * p + ch.g[m];
          const unsigned c0 = P.inc_count[o];
          const double b0 = P.beta_sum[o], q0 = P.beta_sumsq[o];
          const double b = beta_valid ? bx.y[m] : 0.0;
          P.inc_count[o] = c0 + 1u;
          if (beta_valid) {
            P.beta_sum[o] = b0 + b;
            P.beta_sumsq[o] = q0 + b * b;
          }
        }
        BACC_ADD(ACC_SIGSQ, sigsq);
        BACC_ADD(ACC_SIGSQ2, sigsq * sigsq);
        BACC_ADD(ACC_K, k);
        if (P.trace_sigsq && trace_at + sweep < P.trace_stride) {
          const size_t o = (size_t)chain * P.trace_stride + trace_at + sweep;
          if (lane == 0) {
            P.trace_sigsq[o] = sigsq;
            P.trace_logp[o] = M.logp;
            P.trace_k[o] = (double)k;
          }
          if (P.rec_idx) {
            for (int m = lane; m < k && m < P.rec_cap; m += WAVE) {
              P.rec_idx[o * P.rec_cap + m] = ch.g[m];
              P.rec_beta[o * P.rec_cap + m] = beta_valid ? bx.y[m] : 0.0;
            }
          }
        }
        if (beta_valid) {  // coefficients of the last complete draw (written back at the end)
          for (int m = lane; m < k; m += WAVE) {
            bx.bst[m] = bx.y[m];
            bx.gst[m] = ch.g[m];
          }
          klast = k;
        }
        wave_sync();
        ++done;
        ++sweep;
        ++iteration;
        rates_saved = false;
        phase = PH_BEGIN;
        MST(7);
      }
      MST(8);
      if (lane == 0) {
        ctl[CT_CMD] = (double)cmd;
        if (cmd == BCMD_BUILD) { ctl[BX_K] = (double)ch.k; ctl[BX_SLOT] = (double)b_slot; ctl[BX_REUSE] = b_reuse ? 1.0 : 0.0; }
      }
    }
#ifdef BA_BSTAMPS
    const long long bt1 = (long long)__builtin_readcyclecounter();
    bph[0] += (double)(bt1 - bt0);
    if (wave == 0 && lane == 0) atomicAdd(&g_bstamp[8 + (bphase_at_entry & 7)], (unsigned long long)(bt1 - bt0));
#endif
    __syncthreads();
    const int cmd = (int)ctl[CT_CMD];
#ifdef BA_BSTAMPS
    bph[4 + (cmd & 3)] += 1.0;
#endif
    if (cmd == BCMD_EXIT) break;
    if (cmd == BCMD_UNIF) {
      const uint64_t upos = ((AS_LDS const uint64_t *)(ctl + CT_POS))[0];
      if (p > 1) shuffle_targets(key, upos, p, threadIdx.x, WAVE * W, ch.oth, P.mode == 2);
    } else if (cmd == BCMD_EVAL) {
      Model Me;
      Me.logp = ctl[CT_LOGP]; Me.lp = ctl[CT_LP]; Me.ldv = ctl[CT_LDV];
      Me.lda = ctl[CT_LDA]; Me.Q = ctl[CT_Q]; Me.c = ctl[CT_C];
      Me.SS = 0; Me.pd = true; Me.bad = 0;
      if (wave != 0) {
        ch.k = (int)ctl[CT_K];
        bind((int)ctl[CT_CUR]);
      }
      c_f64 *sc = scalar_view(ch.sc_store);
      const int idx = (int)ctl[CT_I0] + WAVE * wave + lane;
      const bool valid = idx < p;
      const Proposal pr = big_eval<true>(BP, ch, Me, bx, sc, valid ? idx : 0, valid, idx - lane);
      if (valid) {
        // (the adaptive moves compare log model probabilities themselves)
        ch.tab_lp[idx] = adaptive ? pr.logp : exp(pr.logp - Me.logp);
        ch.tab_kind[idx] = (uint8_t)(pr.bad_ss ? STOP_BAD : (pr.slow ? STOP_SLOW : 0));
      }
    } else if (wave == 0 || (int)ctl[BX_REUSE] == 0) {  // BCMD_BUILD
      // (both wavefronts where factors are to be computed: wave 1 takes A_g's)
      const bool reuse = wave == 0 ? b_reuse : false;
      if (wave != 0) ch.k = (int)ctl[BX_K];
      double *dst = slot_block(wave == 0 ? b_slot : (int)ctl[BX_SLOT]);
      Model Mn = M;
      big_build_body(BP, ch, Mn, dst, bx, reuse, reuse ? -1 : wave, ctl);
      if (wave != 0) {
        // (wave 1's part ends with the factor)
      } else if (Mn.bad) {
        status = Mn.bad;
      } else if (b_reason != BR_TRY) {
        M = Mn;
        store_scalars(dst, bx.S, M, lane);
        table_valid = false;
        other_ok = false;
        table_valid_other = false;
        publish_ctl();
        if (b_reason == BR_VALID && !(M.logp > -BA_INF && M.logp < BA_INF)) status = CHAIN_ILLEGAL_START;
        phase = b_after;
      } else {
        // the decision riding on the build: t_kind 0 forced (accepted on the
        // table), 1 accept unless log u > delta (flip), 2 accept iff log u <
        // delta (swap move)
        bool acc = true;
        if (t_kind != 0) {
          const double d = (Mn.logp - t_lfw) - (M.logp - t_lrev);
          if (Mn.logp > -BA_INF) BACC_MIN(fabs(t_lu - d));
          acc = (t_kind == 1) ? !(t_lu > d) : (t_lu < d);
        }
        if (acc && adaptive) {
          // adjust_birth_rate / adjust_death_rate (.cpp:194-200, :228-234) with the
          // acceptance probability of the move just made
          double alpha = exp((Mn.logp - t_lfw) - (M.logp - t_lrev));
          if (alpha > 1.0) alpha = 1.0;
          double adjustment = P.ada_step / ((1.0 + (double)iteration) / (double)p);
          adjustment *= (alpha - P.ada_target);
          double *rate = t_birth ? g_birth : g_death;
          if (lane == 0) rate[t_f1] = rate[t_f1] * exp(adjustment);
          cum_dirty = true;
          if (!Mn.pd) status = CHAIN_NOT_PD;
        }
        if (acc) {
          M = Mn;
          store_scalars(dst, bx.S, M, lane);
          table_valid_other = table_valid;
          table_valid = false;
          other_ok = (t_f2 < 0);
          other_var = t_f1;
          cur ^= 1;
          bind(cur);
          publish_ctl();
          BACC_ADD(ACC_ACCEPTS, 1);
          if (t_kind == 0 && !M.pd) status = CHAIN_NOT_PD;
        } else {
          // rejected: gamma back; the slot in use was never touched, the other
          // one no longer holds the model left behind
          flip_in_lds(t_f1, t_f2);
          other_ok = false;
          table_valid_other = false;
          reload_tail_vectors();
        }
        phase = b_after;
      }
    }
    __syncthreads();
#ifdef BA_BSTAMPS
    bt0 = (long long)__builtin_readcyclecounter();
    bph[cmd & 3] += (cmd == 0) ? 0.0 : (double)(bt0 - bt1);
#endif
  }
  if (wave != 0) return;

  // ---- write the chain back
  wave_sync();
  {
    const lds_u8 *gsrc = aborted ? ch.gam0 : ch.gam;
    const lds_u16 *psrc = (aborted && p > 1) ? ch.perm_alt : ch.perm;
    for (int j = lane; j < p; j += WAVE) {
      g_gamma[j] = gsrc[j];
      if (!adaptive) g_perm[j] = psrc[j];
      if (adaptive && aborted && rates_saved) {   // the aborted sweep's rate changes
        g_birth[j] = birth0[j];
        g_death[j] = death0[j];
      }
    }
    if (aborted) pos = pos0;
  }
  if (beta_valid && done > 0) {
    double *g_beta = P.beta + (size_t)chain * p;
    for (int j = lane; j < p; j += WAVE) g_beta[j] = 0.0;
    wave_sync();
    for (int m = lane; m < klast; m += WAVE) g_beta[bx.gst[m]] = bx.bst[m];
  } else if (nflips > 0 && done > 0) {
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
    if (adaptive) P.ada_iter[chain] = iteration;
    if (P.col_request) P.col_request[chain] = need_col;
    P.todo[chain] = nsweeps - done + owed_after;
    if (P.ran) P.ran[chain] = done;
    const int tag = big_tag | cur;
    const bool good = !aborted && status == CHAIN_OK && nsweeps > 0;
    P.table_tag[chain] = (table_valid && good) ? tag : 0;
    P.model_tag[chain] = good ? tag : 0;
    if (P.trace_idx) P.trace_idx[chain] = trace_at + done;
    if (P.maxk) atomicMax(P.maxk, kmax);
    double *a = P.acc + (size_t)chain * ACC_COUNT;
    a[ACC_SWEEPS] += done;
    a[ACC_SIGSQ] += ctl[CT_ACC + ACC_SIGSQ];
    a[ACC_SIGSQ2] += ctl[CT_ACC + ACC_SIGSQ2];
    a[ACC_K] += ctl[CT_ACC + ACC_K];
    a[ACC_ACCEPTS] += ctl[CT_ACC + ACC_ACCEPTS];
    a[ACC_PROPOSALS] += ctl[CT_ACC + ACC_PROPOSALS];
    a[ACC_SLOT_HITS] += ctl[CT_ACC + ACC_SLOT_HITS];
    a[ACC_MIN_MARGIN] = fmin(a[ACC_MIN_MARGIN], ctl[CT_ACC + ACC_MIN_MARGIN]);
#ifdef BA_BSTAMPS
    if (blockIdx.x == 0)
      printf("build stamps (cycles, all chains so far): head/none %llu solve %llu update %llu chol_tile %llu store %llu fence %llu inverses %llu rhs+w %llu\n",
             g_bstamp[0], g_bstamp[1], g_bstamp[2], g_bstamp[3], g_bstamp[4], g_bstamp[5], g_bstamp[6], g_bstamp[7]);
    if (blockIdx.x == 0)
      printf("master stamps (cycles, all chains so far): shuffle %llu before-walk %llu walk %llu before-swap %llu swap %llu sigma %llu beta %llu summaries %llu other %llu\n",
             g_mstamp[0], g_mstamp[1], g_mstamp[2], g_mstamp[3], g_mstamp[4], g_mstamp[5], g_mstamp[6], g_mstamp[7], g_mstamp[8]);
    if (blockIdx.x == 0)
      printf("master stamps by phase at entry (cycles, all chains so far): INIT %llu BEGIN %llu SHUFFLED %llu FLIPS %llu SWAP %llu TAIL %llu ADA %llu\n",
             g_bstamp[8], g_bstamp[9], g_bstamp[10], g_bstamp[11], g_bstamp[12], g_bstamp[13], g_bstamp[14]);
    if (!adaptive)
      for (int i = 0; i < 8; ++i) a[ACC_PHASE0 + i] += bph[i];   // master | EVAL | UNIF | BUILD cycles, then their counts (exit, EVAL, UNIF, BUILD)
#endif
    if (adaptive)
      a[ACC_PHASE0] = (a[ACC_PHASE0] == 0.0) ? ctl[CT_ACC + ACC_PHASE0]
                                             : fmin(a[ACC_PHASE0], ctl[CT_ACC + ACC_PHASE0]);
  }
}

// log_model_prob of inclusion vectors of more than 64 variables (BregVsSampler::
// log_model_prob, BregVsSampler.cpp:216-239): one wavefront per vector, the
// large-model kernel's build on a private model block and solve workspace.
// which[w] = the vector's row in gammas / out / status_out.
__global__ __launch_bounds__(64) void ssvs_big_logp_kernel(SsvsParams P, int kcap, const uint8_t *gammas,
                                                           const int *which, int nwhich, double *model_ws,
                                                           double *xs_ws, double *out, int *status_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int w = (int)blockIdx.x, lane = threadIdx.x, p = P.p;
  if (w >= nwhich) return;
  const int row = which[w];
  const SsvsBigLds lay = ssvs_big_lds_layout(p, kcap);
  Chain ch;
  ch.lane = lane;
  ch.p = p;
  ch.k = 0;
  ch.Lv = ch.La = nullptr;
  ch.rda = nullptr;
  ch.bg = nullptr;
  ch.rdv = to_lds<double>(smem + lay.rd);
  ch.w = to_lds<double>(smem + lay.w);
  ch.g = to_lds<uint16_t>(smem + lay.g);
  ch.gam = to_lds<uint8_t>(smem + lay.gam);
  BigCtx bx;
  bx.kcap = kcap;
  bx.S = ssvs_scalar_layout(kcap);
  bx.xs = xs_ws + (size_t)w * (size_t)kcap * 64;
  bx.tile = to_lds<double>(smem + lay.tile);
  bx.rdt = to_lds<double>(smem + lay.rdt);
  bx.y = to_lds<double>(smem + lay.y);
  bx.bst = to_lds<double>(smem + lay.bst);
  bx.gst = to_lds<uint16_t>(smem + lay.gst);
  BigP BP;
  BP.V = P.V; BP.A = P.A; BP.b = P.b; BP.l1 = P.l1; BP.l0 = P.l0;
  BP.vdiag = nullptr;
  BP.max_model_size = P.max_model_size;
  ch.xty = P.xty;
  ch.DF = P.nobs[0] + P.prior_df;
  ch.ss0q = P.prior_ss + P.yty[0];
  ch.mode = 0;
  ch.sv = ch.sa = ch.sx = 1.0;
  const uint8_t *gg = gammas + (size_t)row * p;
  for (int j = lane; j < p; j += WAVE) ch.gam[j] = gg[j];
  rebuild_g(ch, kcap);
  if (ch.k > kcap) {
    if (lane == 0) { out[row] = __builtin_nan(""); status_out[row] = CHAIN_MODEL_TOO_LARGE; }
    return;
  }
  Model M;
  M.logp = M.lp = M.ldv = M.lda = M.Q = M.c = M.SS = 0.0;
  M.pd = true;
  M.bad = 0;
  big_build_body(BP, ch, M, model_ws + (size_t)w * bx.S.total, bx, false);
  if (lane == 0) {
    out[row] = M.logp;
    status_out[row] = M.bad;
  }
}

hipError_t launch_ssvs_big_logp(hipStream_t stream, const SsvsParams &P, int kcap, const uint8_t *gammas,
                                const int *which, int nwhich, double *model_ws, double *xs_ws, double *out,
                                int *status_out) {
  const SsvsBigLds lay = ssvs_big_lds_layout(P.p, kcap);
  hipError_t e = hipFuncSetAttribute((const void *)ssvs_big_logp_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(ssvs_big_logp_kernel, dim3(nwhich), dim3(WAVE), lay.total, stream, P, kcap, gammas, which,
                     nwhich, model_ws, xs_ws, out, status_out);
  return hipGetLastError();
}

hipError_t launch_ssvs_big(hipStream_t stream, const SsvsParams &P, int nsweeps) {
  const SsvsBigLds lay = ssvs_big_lds_layout(P.p, P.big_kcap);
  hipError_t e = hipFuncSetAttribute((const void *)ssvs_big_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lay.total);
  if (e != hipSuccess) return e;
  KtScope kt(stream, KT_SSVS_BIG);
  hipLaunchKernelGGL(ssvs_big_kernel, dim3(P.chain_count), dim3(2 * WAVE), lay.total, stream, P,
                     nsweeps);
  return hipGetLastError();
}

}  // namespace boom_amd
