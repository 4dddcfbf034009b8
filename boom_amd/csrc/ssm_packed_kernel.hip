// The structural state-space kernel of ssm_kernel.hip for state dimension m <= 16, FOUR
// CHAINS PER WAVEFRONT: a state-sized vector holds one component per lane, so at m <= 16
// three quarters of every instruction of the one-chain-per-wavefront kernels did nothing --
// and the passes are bound by how many instructions a step issues (a wave64 instruction
// occupies its SIMD for four cycles whatever its lanes do), not by memory.  Here the four
// rows of 16 lanes of a wavefront hold four chains (row g = chain 4 * workgroup + g): the
// same instruction stream advances four chains -- the block list, the steps into new
// seasons, the missing observations are the SAME for every chain of a job, so control flow
// stays uniform -- and a workgroup is two wavefronts for 4 chains instead of two for one,
// which also leaves every wavefront a SIMD of its own (1024 chains = 512 wavefronts).
//
// What changes against ssm_kernel.hip: reductions and broadcasts stay inside a row
// (row_shr + row_newbcast instead of v_readlane), per-chain scalars of a step (y*, w, F,
// (v - v+) / F) live in LDS series instead of lane-time registers, a chain's failure is a
// per-row flag instead of a branch.  Same arithmetic per chain as the general kernel (the
// reductions add in the same order), same storage, same draws.
#include <hip/hip_runtime.h>

#include "ktimer.h"

#include "ssg_device.h"

namespace boom_amd {

namespace {

constexpr int PG = 4;        // chains per wavefront
constexpr int PM = 16;       // lanes per chain
constexpr int PBL = 32;      // steps per block of the passes

// sum over the lane's row of 16 (lanes that do not take part hold 0), in every lane of the row
__device__ __forceinline__ double gsum(double x) {
  x += sdpp<0x111, 0xf>(x, 0.0);  // row_shr:1
  x += sdpp<0x112, 0xf>(x, 0.0);
  x += sdpp<0x114, 0xf>(x, 0.0);
  x += sdpp<0x118, 0xf>(x, 0.0);
  return sdpp<0x15F, 0xf>(x, 0.0);   // row_newbcast:15
}
__device__ __forceinline__ double gbelow(double x) { return sdpp<0x111, 0xf>(x, 0.0); }   // the row's lane - 1
__device__ __forceinline__ double gabove(double x) { return sdpp<0x101, 0xf>(x, 0.0); }   // ... lane + 1
// the value at the row's lane j (j the same in every row)
__device__ __forceinline__ double gpick(double x, int j, int gl) { return gsum(gl == j ? x : 0.0); }

__device__ __forceinline__ double pzdot(const LaneInfo &L, double x, int gl) { return gsum(L.zsel(gl) ? x : 0.0); }

// (see vecT / vecTt in ssg_device.h; gl: the lane within its row)
__device__ __forceinline__ double pvecT(const Blocks &B, const LaneInfo &L, double x, int gl, unsigned mv) {
  double y = x;
  const double above = gabove(x);
  if (L.kind == SSG_LOCAL_LINEAR_TREND && gl == L.first) y = x + above;
  if (B.armask) {
    const double below = gbelow(x);
    if (L.kind == SSG_AR) y = below;
    unsigned am = B.armask;
    while (am) {
      const int b = __ffs((int)am) - 1;
      am &= am - 1;
      const double tot = gsum(L.blk == b ? L.phi * x : 0.0);
      if (L.blk == b && gl == L.first) y = tot;
    }
  }
  unsigned sm = mv & B.seasmask;
  while (sm) {
    const int b = __ffs((int)sm) - 1;
    sm &= sm - 1;
    const double tot = gsum(L.blk == b ? x : 0.0);
    if (L.blk == b && gl == L.first + sprev(L.cur, L.dim)) y = -tot;
  }
  return y;
}
__device__ __forceinline__ double pvecTt(const Blocks &B, const LaneInfo &L, double x, int gl, unsigned mv) {
  double y = x;
  const double below = gbelow(x);
  if (L.kind == SSG_LOCAL_LINEAR_TREND && gl == L.first + 1) y = below + x;
  if (B.armask) {
    const double above = gabove(x);
    unsigned am = B.armask;
    while (am) {
      const int b = __ffs((int)am) - 1;
      am &= am - 1;
      const double firstv = gpick(x, Blocks::first_of(B.udesc(b)), gl);
      if (L.blk == b) y = L.phi * firstv + ((gl + 1 < L.first + L.dim) ? above : 0.0);
    }
  }
  unsigned sm = mv & B.seasmask;
  while (sm) {
    const int b = __ffs((int)sm) - 1;
    sm &= sm - 1;
    const int c1 = Blocks::first_of(B.udesc(b)) + (int)(B.urc(b) >> 16);
    const double firstv = gpick(x, c1, gl);
    if (L.blk == b) y = (gl == c1) ? -firstv : x - firstv;
  }
  return y;
}
__device__ __forceinline__ unsigned pmoving(const Blocks &B) { return B.moving() & 0xffffu; }
__device__ __forceinline__ void padvance(Blocks &B, LaneInfo &L, unsigned mv, int gl) {
  B.advance(mv, gl);
  if (L.moves(mv)) L.cur = sprev(L.cur, L.dim);
}
__device__ __forceinline__ void pretreat(Blocks &B, LaneInfo &L, unsigned mv, int gl) {
  B.retreat(mv, gl);
  if (L.moves(mv)) L.cur = snext(L.cur, L.dim);
}

// a row's block of n doubles between its chain's HBM array and its LDS buffer
__device__ __forceinline__ void gblk_load(double *lds, const double *gmem, int n, int gl) {
  for (int i = gl; i < n; i += PM) lds[i] = gmem[i];
}
__device__ __forceinline__ void gblk_store(double *gmem, const double *lds, int n, int gl) {
  for (int i = gl; i < n; i += PM) gmem[i] = lds[i];
}

}  // namespace

// LDS per chain of the passes, in doubles: two block buffers | P | normals / disturbances |
// five series of the block's per-step scalars | the autoregression blocks' xtx rows
__host__ __device__ inline int ssg_packed_row_doubles(int m, int ld, int nerr, int nar) {
  return 2 * PBL * m + m * ld + (PBL * (nerr + 1) + PM + 8) + 5 * PBL + nar * AR_MAX * (AR_MAX + 1);
}

// grid = ceil(chains / 4), block = 256 (all four wavefronts share the prologue -- the samplers,
// y*, the normals, chain by chain -- two go on to the passes)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void ssg_packed_kernel(SsParams P, int draw_variances) {
  extern __shared__ __align__(16) unsigned char s_raw[];
  __shared__ int s_flag[PG], s_live[PG];
  __shared__ double s_sig2[PG][SSG_MAX_VAR];
  __shared__ double s_phi[PG][SSG_MAX_AR * AR_MAX];
  __shared__ double s_tv[PG][PM];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, gl = lane & 15;
  const int chain0 = (int)blockIdx.x * PG + P.chain_first, chain_end = P.chain_first + P.chain_count;
  if (chain0 >= chain_end) return;
  const SsmParams &M = P.ssm;
  const SsgSpec &Q = *M.spec;
  const int T = P.T, p = P.p, m = M.m, nb = M.nblocks, ld = M.ld, NE = M.nerr;
  const int rowd = ssg_packed_row_doubles(m, ld, NE, M.nar);
  NormalsLds &s_norm = *reinterpret_cast<NormalsLds *>(s_raw);
  ArLds &s_ar = *reinterpret_cast<ArLds *>(s_raw);

  // ---- prologue, chain by chain: the state models' samplers, y*, the sweep's normals
  for (int c4 = 0; c4 < PG; ++c4) {
    const int chain = chain0 + c4;
    const bool live = chain < chain_end && P.status[chain] == CHAIN_OK && !(P.only_ran && P.only_ran[chain] == 0);
    if (tid == 0) { s_live[c4] = live ? 1 : 0; s_flag[c4] = CHAIN_OK; }
    if (!live) {
      // (a row without a chain runs along on harmless values and stores nothing)
      if (tid < SSG_MAX_VAR) s_sig2[c4][tid] = 1.0;
      if (tid < SSG_MAX_AR * AR_MAX) s_phi[c4][tid] = 0.0;
      continue;
    }
    const uint32_t gchain = (uint32_t)(P.chain_offset + chain);
    int status = CHAIN_OK;
    __syncthreads();
    if (tid < SSG_MAX_VAR) s_sig2[c4][tid] = M.var_sigsq[(size_t)chain * SSG_MAX_VAR + tid];
    __syncthreads();
    if (draw_variances) {
      for (int b = 0; b < nb; ++b) {
        const SsgBlock &K = Q.blk[b];
        if (K.kind == SSG_AR) continue;
        for (int v = 0; v < K.nvar; ++v) {
          const int vi = K.var0 + v;
          const size_t at = (size_t)chain * SSG_MAX_VAR + vi;
          SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, (uint32_t)K.sid[v]}, M.pos_var[at]};
          int bad = 0;
          const double DF = M.var_n[at] + Q.prior_df[vi];
          const double SSQ = M.var_ss[at] + Q.prior_ss[vi];
          double draw = d_draw_variance(rng, DF, SSQ, Q.sigma_max[vi], &bad);
          if (bad) status = CHAIN_RNG_BRANCH;
          if (K.kind == SSG_LOCAL_LINEAR_TREND) draw = 1.0 / (1.0 / draw);
          __syncthreads();   // (everybody has read the old position)
          if (tid == 0) {
            M.pos_var[at] = rng.pos;
            M.var_sigsq[at] = draw;
            s_sig2[c4][vi] = draw;
          }
        }
      }
    }
    for (int b = 0; b < nb && status == CHAIN_OK; ++b) {
      const SsgBlock &K = Q.blk[b];
      if (K.kind != SSG_AR) continue;
      const int L = K.lags, vi = K.var0;
      const size_t at = (size_t)chain * SSG_MAX_VAR + vi;
      double *gphi = M.ar_phi + ((size_t)chain * SSG_MAX_AR + K.ar_index) * AR_MAX;
      __syncthreads();
      if (wave == 0) {
        double ph = (lane < L) ? gphi[lane] : 0.0;
        double sig2a = M.var_sigsq[at];
        if (draw_variances) {
          SeqRng rng{PhiloxKey{P.seed_lo, P.seed_hi, gchain, (uint32_t)K.sid[0]}, M.pos_var[at]};
          const double *suf = M.ar_suf + ((size_t)chain * SSG_MAX_AR + K.ar_index) * AR_SUF_STRIDE;
          const int st = ar_draw(s_ar, suf, L, Q.prior_df[vi], Q.prior_ss[vi], Q.sigma_max[vi], rng, ph, sig2a, lane);
          if (st != CHAIN_OK) {
            if (lane == 0) s_flag[c4] = st;
          } else {
            if (lane < L) gphi[lane] = ph;
            if (lane == 0) {
              M.var_sigsq[at] = sig2a;
              M.pos_var[at] = rng.pos;
            }
          }
        }
        if (lane < AR_MAX) s_phi[c4][K.ar_index * AR_MAX + lane] = (lane < L) ? ph : 0.0;
        if (lane == 0) s_sig2[c4][vi] = sig2a;
      }
      __syncthreads();
    }
    __syncthreads();
    if (status != CHAIN_OK && tid == 0) s_flag[c4] = status;
    __syncthreads();
    if (s_flag[c4] != CHAIN_OK) continue;

    double *w0 = P.scratch + (size_t)chain * P.scratch_stride;
    double *szz = M.work + (size_t)chain * M.work_stride + (size_t)(2 * m + NE) * T;
    const double *beta = P.beta + (size_t)chain * p;
    // y*_t = y_t - x_t'beta (blocks of 64 steps, the four waves in turn)
    for (int tb = wave * WAVE; tb < T; tb += 4 * WAVE) {
      const int t = tb + lane;
      double pred = 0.0;
      for (int base = 0; base < p; base += WAVE) {
        const int j = base + lane;
        const double bj = (j < p) ? beta[j] : 0.0;
        unsigned long long mk = __ballot(bj != 0.0);
        while (mk) {
          const int l = __ffsll((long long)mk) - 1;
          mk &= mk - 1;
          const double bb = rl(bj, l);
          pred += P.X[(size_t)(base + l) * T + (t < T ? t : T - 1)] * bb;
        }
      }
      if (t < T) w0[t] = P.y[t] - pred;
    }
    // the normals of simulate_forward, in stream order (see ssm_kernel.hip)
    int nfirst = 0, nconst = 0, nseas = 0;
    const int dH = (sqrt(P.sigsq[chain]) != 0.0);
    for (int b = 0; b < nb; ++b) {
      const SsgBlock &K = Q.blk[b];
      const bool nz = s_sig2[c4][K.var0] != 0.0;
      if (K.kind == SSG_LOCAL_LEVEL) { nfirst += (Q.P0[K.first] != 0.0) ? 1 : 0; nconst += nz ? 1 : 0; }
      else if (K.kind == SSG_LOCAL_LINEAR_TREND) { nfirst += 2; nconst += 2; }
      else if (K.kind == SSG_AR) { nfirst += K.dim; nconst += 1; }
      else { nfirst += K.dim; if (nz) nseas += seasons_started(T - 1, K.duration, K.phase); }
    }
    const int N = (nfirst + dH) + (T - 1) * (nconst + dH) + nseas;
    const int st = stream_normals(s_norm, PhiloxKey{P.seed_lo, P.seed_hi, gchain, 2u}, P.pos_state[chain], N,
                                  szz, &P.pos_state[chain], ss_slot_serve(P));
    if (st != CHAIN_OK && tid == 0) s_flag[c4] = st;
    __syncthreads();
  }
  __syncthreads();
  if (wave >= 2) return;

  // ---- the passes: row g of the wavefront = chain chain0 + g
  const int chain = chain0 + g;
  const bool act = s_live[g] != 0 && s_flag[g] == CHAIN_OK;
  const int cq = act ? chain : chain0;      // (a row without a chain reads some chain's data and writes nothing)
  double *s_row = reinterpret_cast<double *>(s_raw) + (size_t)g * rowd;
  double *s_blk0 = s_row;
  double *s_blk1 = s_blk0 + PBL * m;
  double *s_P = s_blk1 + PBL * m;
  double *s_z = s_P + m * ld;
  double *s_ys = s_z + (PBL * (NE + 1) + PM + 8);   // y*, then w = y* - y+, then (v - v+) / F
  double *s_F = s_ys + PBL;
  double *s_res = s_F + PBL;
  double *s_tmp = s_res + PBL;                      // (two spare series)
  double *s_axx = s_tmp + 2 * PBL;
  (void)s_tmp;

  Blocks B;
  B.load(Q, nb, gl);
  B.always &= 0xffffu; B.seasmask &= 0xffffu; B.armask &= 0xffffu;
  LaneInfo LI{-1, 0, 0, 0, 0, 0.0};
  int var_l = 0, cbefore_l = 0, ipos_l = 0;
  unsigned sbefore_l = 0;
  bool init_l = false;
  int nconst_l = 0, nfirst_l = 0;       // (the same in a row's lanes; per chain: a variance may be exactly 0)
  unsigned seas_active_l = 0;
  {
    int cb = 0, ip = 0;
    unsigned sb = 0;
    for (int b = 0; b < nb; ++b) {
      const unsigned d = B.udesc(b);
      const int f = Blocks::first_of(d), n = Blocks::dim_of(d), kd = Blocks::kind_of(d), v0 = Blocks::var0_of(d);
      const bool mine = gl >= f && gl < f + n;
      const bool second = kd == SSG_LOCAL_LINEAR_TREND && gl == f + 1;
      if (mine) {
        LI.blk = b; LI.kind = kd; LI.first = f; LI.dim = n;
        var_l = v0 + (second ? 1 : 0);
        cbefore_l = cb + (second ? 1 : 0);
        sbefore_l = sb;
        if (kd == SSG_AR) LI.phi = s_phi[g][Blocks::arx_of(d) * AR_MAX + (gl - f)];
      }
      if (kd == SSG_LOCAL_LEVEL) {
        const bool drawn = Q.P0[f] != 0.0;
        if (mine) { ipos_l = ip; init_l = drawn; }
        ip += drawn ? 1 : 0;
      } else {
        if (mine) { ipos_l = ip + (gl - f); init_l = true; }
        ip += n;
      }
      const bool nz = s_sig2[g][v0] != 0.0;
      if (kd == SSG_LOCAL_LEVEL) cb += nz ? 1 : 0;
      else if (kd == SSG_LOCAL_LINEAR_TREND) cb += 2;
      else if (kd == SSG_AR) cb += 1;
      else {
        if (nz) seas_active_l |= 1u << b;
        sb |= 1u << b;
      }
    }
    nconst_l = cb;
    nfirst_l = ip;
  }
  double a0l = 0.0, P0l = 0.0;
  if (gl < m) { a0l = Q.a0[gl]; P0l = Q.P0[gl]; }
  const double sig_l = (gl < m) ? s_sig2[g][var_l] : 0.0;
  const double sd_l = sqrt(sig_l);
  const double H = P.sigsq[cq], sqrtH = sqrt(H);
  const int dH_l = (sqrtH != 0.0) ? 1 : 0;
  double *w0 = P.scratch + (size_t)cq * P.scratch_stride;
  double *sres = w0 + T;
  double *wk = M.work + (size_t)cq * M.work_stride;
  double *gK = wk;
  double *gst = gK + (size_t)m * T;
  double *gd = gst + (size_t)m * T;
  double *szz = gd + (size_t)NE * T;
  auto seasonal_draws = [&](int t) -> int {
    int o = 0;
    unsigned sm = B.seasmask;
    while (sm) {
      const int b = __ffs((int)sm) - 1;
      sm &= sm - 1;
      const unsigned dpw = (unsigned)__builtin_amdgcn_readlane((int)B.dp, b);
      if ((seas_active_l >> b) & 1u) o += seasons_started(t, (int)(dpw & 0xffffu), (int)(dpw >> 16));
    }
    return o;
  };
  const int N_l = (nfirst_l + dH_l) + (T - 1) * (nconst_l + dH_l) + seasonal_draws(T - 1);
  auto zoffset = [&](int t) -> int { return (nfirst_l + dH_l) + (t - 1) * (nconst_l + dH_l) + seasonal_draws(t - 1); };
  double *blk = wave == 0 ? s_blk0 : s_blk1;
  int bad_l = 0;

  if (wave == 1) {
    // ---- the variance recursion P_t -> F_t, K_t (see ssm_kernel.hip: the filtered form, a
    // column pass and a row pass per step)
    for (int e = gl; e < m * ld; e += PM) s_P[e] = 0.0;
    wave_lds_sync();
    if (gl < m) s_P[gl * ld + gl] = P0l;
    wave_lds_sync();
    seek(B, LI, 0, 0);
    for (int tb = 0; tb < T; tb += PBL) {
      const int nstep = (T - tb < PBL) ? T - tb : PBL;
      const int ob_l = (lane < nstep && P.observed[tb + lane]) ? 1 : 0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
        const unsigned mv = pmoving(B);
        double PZ = 0.0;
#pragma nounroll
        for (int b = 0; b < nb; ++b) {
          const int zl = Blocks::first_of(B.udesc(b)) + (int)(B.urc(b) >> 16);
          if (gl < m) PZ += s_P[zl * ld + gl];
        }
        const double F = pzdot(LI, PZ, gl) + H;
        if (!(F > 0.0)) bad_l = 1;
        if (gl == 0) s_F[s] = F;
        const double Finv = 1.0 / F;
        const double TPZ = pvecT(B, LI, PZ, gl, mv);
        if (gl < m) {
          blk[s * m + gl] = obs ? TPZ * Finv : 0.0;
          s_tv[g][gl] = PZ;
        }
        wave_lds_sync();
#pragma nounroll
        for (int b = 0; b < nb; ++b) {
          const unsigned d = B.udesc(b);
          const int f = Blocks::first_of(d), n = Blocks::dim_of(d), kd = Blocks::kind_of(d);
          if (gl >= m) continue;
          double *col = s_P + f * ld + gl;
          const double *tv = s_tv[g];
          if (kd == SSG_LOCAL_LEVEL) {
            double v = col[0];
            if (obs) v -= (tv[f] * PZ) * Finv;
            if (gl == f) v += s_sig2[g][Blocks::var0_of(d)];
            col[0] = v;
          } else if (kd == SSG_LOCAL_LINEAR_TREND) {
            double v0 = col[0], v1 = col[ld];
            if (obs) {
              v0 -= (tv[f] * PZ) * Finv;
              v1 -= (tv[f + 1] * PZ) * Finv;
              col[ld] = v1;
            }
            col[0] = v0 + v1;
          } else if (kd == SSG_SEASONAL) {
            const bool moves = (mv >> b) & 1u;
            if (obs || moves) {
              double cs = 0.0;
#pragma nounroll
              for (int i = 0; i < n; ++i) {
                double v = col[i * ld];
                if (obs) {
                  v -= (tv[f + i] * PZ) * Finv;
                  col[i * ld] = v;
                }
                cs -= v;
              }
              if (moves) col[sprev((int)(B.urc(b) >> 16), n) * ld] = cs;
            }
          } else {
            const double *ph = s_phi[g] + Blocks::arx_of(d) * AR_MAX;
            double cs = 0.0;
#pragma nounroll
            for (int q = n - 1; q >= 0; --q) {
              double v = col[q * ld];
              if (obs) v -= (tv[f + q] * PZ) * Finv;
              cs += ph[q] * v;
              if (q + 1 < n) col[(q + 1) * ld] = v;
            }
            col[0] = cs;
          }
        }
        wave_lds_sync();
        unsigned tm = mv & ~((unsigned)__ballot(Blocks::kind_of(B.desc) == SSG_LOCAL_LEVEL) & 0xffffu);
        while (tm) {
          const int b = __ffs((int)tm) - 1;
          tm &= tm - 1;
          const unsigned d = B.udesc(b);
          const int f = Blocks::first_of(d), n = Blocks::dim_of(d), kd = Blocks::kind_of(d);
          if (gl >= m) continue;
          double *row = s_P + gl * ld + f;
          const double sg = s_sig2[g][Blocks::var0_of(d)];
          if (kd == SSG_LOCAL_LINEAR_TREND) {
            const double a = row[0], bb = row[1];
            row[0] = (a + bb) + (gl == f ? sg : 0.0);
            if (gl == f + 1) row[1] = bb + s_sig2[g][Blocks::var0_of(d) + 1];
          } else if (kd == SSG_SEASONAL) {
            const int w = sprev((int)(B.urc(b) >> 16), n);
            double cs = 0.0;
#pragma nounroll
            for (int j = 0; j < n; ++j) cs -= row[j];
            row[w] = cs + (gl == f + w ? sg : 0.0);
          } else {
            const double *ph = s_phi[g] + Blocks::arx_of(d) * AR_MAX;
            double cs = 0.0;
#pragma nounroll
            for (int q = n - 1; q >= 0; --q) {
              const double v = row[q];
              cs += ph[q] * v;
              if (q + 1 < n) row[q + 1] = v;
            }
            row[0] = cs + (gl == f ? sg : 0.0);
          }
        }
        wave_lds_sync();
        padvance(B, LI, mv, gl);
      }
      wave_lds_sync();
      if (act) {
        gblk_store(gK + (size_t)tb * m, blk, nstep * m, gl);
        for (int i = gl; i < nstep; i += PM) sres[tb + i] = s_F[i];
      }
      wave_lds_sync();
    }
    if (gsum(gl < m ? (double)bad_l : 0.0) != 0.0 && gl == 0 && act) s_flag[g] = CHAIN_FORECAST_VARIANCE;
  } else {
    // ---- simulate alpha+_t, y+_t and w_t = y*_t - y+_t
    double alpha = 0.0;
    seek(B, LI, 0, -1);
    for (int tb = 0; tb < T; tb += PBL) {
      const int nstep = (T - tb < PBL) ? T - tb : PBL;
      const int zstart = tb == 0 ? 0 : zoffset(tb);
      const int zend = (tb + nstep >= T) ? N_l : zoffset(tb + nstep);
      gblk_load(s_z, szz + zstart, zend - zstart, gl);
      gblk_load(s_ys, w0 + tb, nstep, gl);
      wave_lds_sync();
      int zo = 0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        if (tb + s == 0) {
          const double z = (gl < m && init_l) ? s_z[ipos_l] : 0.0;
          alpha = (gl < m) ? sqrt(P0l) * z + a0l : 0.0;
          zo = nfirst_l;
          padvance(B, LI, 0u, gl);
        } else {
          const unsigned mv = pmoving(B);
          const unsigned actm = mv & seas_active_l;
          alpha = pvecT(B, LI, alpha, gl, mv);
          padvance(B, LI, mv, gl);
          bool err = false;
          if (LI.kind == SSG_LOCAL_LEVEL) err = sig_l != 0.0;
          else if (LI.kind == SSG_LOCAL_LINEAR_TREND) err = true;
          else if (LI.kind == SSG_AR) err = gl == LI.first;
          else if (LI.kind == SSG_SEASONAL) err = ((actm >> LI.blk) & 1u) && gl == LI.first + LI.cur;
          const double z = err ? s_z[zo + cbefore_l + __popc(actm & sbefore_l)] : 0.0;
          alpha += sd_l * z;
          zo += nconst_l + __popc(actm);
        }
        const double zh = dH_l ? s_z[zo] : 0.0;
        zo += dH_l;
        const double yplus = pzdot(LI, alpha, gl) + sqrtH * zh;
        const double w = s_ys[s] - yplus;
        if (gl == 0) s_ys[s] = w;
        if (gl < m) blk[s * m + gl] = alpha;
      }
      wave_lds_sync();
      if (act) {
        gblk_store(gst + (size_t)tb * m, blk, nstep * m, gl);
        gblk_store(w0 + tb, s_ys, nstep, gl);
      }
      wave_lds_sync();
    }
  }
  __threadfence_block();
  __syncthreads();
  if (wave != 0) return;
  const bool act2 = act && s_flag[g] == CHAIN_OK;

  // ---- the filter on w = y* - y+
  {
    double delta = 0.0;
    seek(B, LI, 0, 0);
    for (int tb = 0; tb < T; tb += PBL) {
      const int nstep = (T - tb < PBL) ? T - tb : PBL;
      gblk_load(blk, gK + (size_t)tb * m, nstep * m, gl);
      gblk_load(s_ys, w0 + tb, nstep, gl);
      gblk_load(s_F, sres + tb, nstep, gl);
      wave_lds_sync();
      const int ob_l = (lane < nstep && P.observed[tb + lane]) ? 1 : 0;
#pragma nounroll
      for (int s = 0; s < nstep; ++s) {
        const double K = (gl < m) ? blk[s * m + gl] : 0.0;
        const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
        const unsigned mv = pmoving(B);
        const double e = obs ? s_ys[s] - pzdot(LI, delta, gl) : 0.0;
        const double ef = obs ? e / s_F[s] : 0.0;
        if (gl == 0) s_ys[s] = ef;
        delta = pvecT(B, LI, delta, gl, mv) + K * e;
        padvance(B, LI, mv, gl);
      }
      wave_lds_sync();
      if (act2) gblk_store(w0 + tb, s_ys, nstep, gl);
      wave_lds_sync();
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);

  // ---- backward: r_{t-1} = T_t' r_t + Z ((v_t - v+_t) / F_t - K_t' r_t)
  double r = 0.0;
  seek(B, LI, T, -1);
  for (int tb = ((T - 1) / PBL) * PBL; tb >= 0; tb -= PBL) {
    const int nstep = (T - tb < PBL) ? T - tb : PBL;
    gblk_load(blk, gK + (size_t)tb * m, nstep * m, gl);
    gblk_load(s_ys, w0 + tb, nstep, gl);
    wave_lds_sync();
#pragma nounroll
    for (int s = nstep - 1; s >= 0; --s) {
      const double K = (gl < m) ? blk[s * m + gl] : 0.0;
      const unsigned mv = pmoving(B);
      if (gl < m) {
        bool carrier;
        if (LI.kind == SSG_SEASONAL) carrier = gl == LI.first + LI.cur;
        else if (LI.kind == SSG_LOCAL_LINEAR_TREND) carrier = true;
        else carrier = gl == LI.first;
        if (carrier) s_z[var_l * PBL + s] = r;
      }
      const double kr = gsum(K * r);
      const double coef = s_ys[s] - kr;
      r = pvecTt(B, LI, r, gl, mv);
      pretreat(B, LI, mv, gl);
      if (LI.zsel(gl)) r += coef;
      if (gl >= m) r = 0.0;
    }
    wave_lds_sync();
    if (act2)
      for (int e = 0; e < NE; ++e) gblk_store(gd + (size_t)e * T + tb, s_z + e * PBL, nstep, gl);
    wave_lds_sync();
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);

  // ---- forward: the mean correction, the state draw, the sufficient statistics
  double mc = P0l * r;
  double prev = 0.0, suf_l = 0.0;
  double mv_ybar = 0.0, mv_sumsq = 0.0, mv_n = 0.0;
  double yty = 0.0, nobs = 0.0;
  double axy = 0.0, ayy = 0.0;
  for (int e2 = gl; e2 < M.nar * AR_MAX * (AR_MAX + 1); e2 += PM) s_axx[e2] = 0.0;
  wave_lds_sync();
  double *oblk = s_blk1;
  seek(B, LI, 0, -1);
  for (int tb = 0; tb < T; tb += PBL) {
    const int nstep = (T - tb < PBL) ? T - tb : PBL;
    gblk_load(blk, gst + (size_t)tb * m, nstep * m, gl);
    for (int e = 0; e < NE; ++e)
      for (int i = gl; i < nstep; i += PM) s_z[e * PBL + i] = (tb + i > 0) ? gd[(size_t)e * T + tb + i - 1] : 0.0;
    wave_lds_sync();
    const double y_l = (lane < nstep) ? P.y[tb + lane] : 0.0;
    const int ob_l = (lane < nstep && P.observed[tb + lane]) ? 1 : 0;
#pragma nounroll
    for (int s = 0; s < nstep; ++s) {
      const double ap = (gl < m) ? blk[s * m + gl] : 0.0;
      unsigned mv = 0;
      double tot_then = 0.0;
      if (tb + s > 0) {
        mv = pmoving(B);
        unsigned sm = mv & B.seasmask;
        while (sm) {
          const int b = __ffs((int)sm) - 1;
          sm &= sm - 1;
          const double tb_ = gsum(LI.blk == b ? prev : 0.0);
          if (LI.blk == b) tot_then = tb_;
        }
        mc = pvecT(B, LI, mc, gl, mv);
        padvance(B, LI, mv, gl);
        bool carrier = false;
        if (gl < m) {
          if (LI.kind == SSG_SEASONAL) carrier = LI.moves(mv) && gl == LI.first + LI.cur;
          else if (LI.kind == SSG_LOCAL_LINEAR_TREND) carrier = true;
          else carrier = gl == LI.first;
        }
        if (carrier) mc += sig_l * s_z[var_l * PBL + s];
      } else {
        padvance(B, LI, 0u, gl);
      }
      const double st = (gl < m) ? ap + mc : 0.0;
      const double then1 = gabove(prev);
      if (tb + s > 0) {
        if (LI.kind == SSG_LOCAL_LEVEL) {
          const double diff = st - prev;
          suf_l += diff * diff;
        } else if (LI.kind == SSG_LOCAL_LINEAR_TREND) {
          const double err = st - ((gl == LI.first) ? prev + then1 : prev);
          mv_n += 1.0;
          const double wv = (err - mv_ybar) / mv_n;
          mv_ybar += wv;
          mv_sumsq += wv * wv * (mv_n - 1);
          const double w2 = err - mv_ybar;
          mv_sumsq += w2 * w2;
        } else if (LI.kind == SSG_SEASONAL) {
          if (LI.moves(mv) && gl == LI.first + LI.cur) {
            const double dl = st - (-1.0 * tot_then);
            suf_l += dl * dl;
          }
        }
        unsigned am = B.armask;
        while (am) {
          const int b = __ffs((int)am) - 1;
          am &= am - 1;
          const unsigned d = B.udesc(b);
          const int f = Blocks::first_of(d), n = Blocks::dim_of(d);
          const double yy = gpick(st, f, gl);
          double *rowx = s_axx + ((size_t)Blocks::arx_of(d) * AR_MAX + (LI.blk == b ? gl - f : 0)) * (AR_MAX + 1);
#pragma nounroll
          for (int q = 0; q < n; ++q) {
            const double pq = gpick(prev, f + q, gl);
            if (LI.blk == b) rowx[q] += prev * pq * 1.0;
          }
          if (LI.blk == b) {
            axy += (yy * 1.0) * prev;
            ayy += yy * yy * 1.0;
          }
        }
      }
      prev = st;
      if (gl < m) {
        int idx = gl;
        if (LI.kind == SSG_SEASONAL) {
          const int q = gl - LI.first, c = LI.cur;
          idx = LI.first + (q >= c ? q - c : q - c + LI.dim);
        }
        oblk[s * m + idx] = st;
      }
      const bool obs = __builtin_amdgcn_readlane(ob_l, s) != 0;
      const double resid = obs ? rl(y_l, s) - pzdot(LI, st, gl) : 0.0;
      if (gl == 0) s_res[s] = resid;
      if (obs) { yty += resid * resid; nobs += 1.0; }
    }
    wave_lds_sync();
    if (act2) {
      gblk_store(gst + (size_t)tb * m, oblk, nstep * m, gl);
      gblk_store(sres + tb, s_res, nstep, gl);
    }
    wave_lds_sync();
  }
  // publish the sufficient statistics
  for (int b = 0; b < nb; ++b) {
    const unsigned d = B.udesc(b);
    const int f = Blocks::first_of(d), n = Blocks::dim_of(d), kd = Blocks::kind_of(d);
    const size_t at = (size_t)cq * SSG_MAX_VAR + Blocks::var0_of(d);
    if (kd == SSG_LOCAL_LEVEL) {
      if (gl == f && act2) {
        M.var_n[at] = (double)(T - 1);
        M.var_ss[at] = suf_l;
      }
    } else if (kd == SSG_LOCAL_LINEAR_TREND) {
      const double ssv = mv_sumsq + mv_ybar * mv_ybar * mv_n;
      if ((gl == f || gl == f + 1) && act2) {
        M.var_n[at + (gl - f)] = mv_n;
        M.var_ss[at + (gl - f)] = ssv;
      }
    } else if (kd == SSG_SEASONAL) {
      const double tot = gsum(LI.blk == b ? suf_l : 0.0);
      const unsigned dpw = (unsigned)__builtin_amdgcn_readlane((int)B.dp, b);
      if (gl == f && act2) {
        M.var_n[at] = (double)seasons_started(T - 1, (int)(dpw & 0xffffu), (int)(dpw >> 16));
        M.var_ss[at] = tot;
      }
    } else {
      double *suf = M.ar_suf + ((size_t)cq * SSG_MAX_AR + Blocks::arx_of(d)) * AR_SUF_STRIDE;
      if (LI.blk == b && act2) {
        const int i = gl - f;
        const double *rowx = s_axx + ((size_t)Blocks::arx_of(d) * AR_MAX + i) * (AR_MAX + 1);
        for (int q = 0; q < n; ++q) suf[i * AR_MAX + q] = rowx[q];
        suf[AR_SUF_XTY + i] = axy;
      }
      if (gl == f && act2) {
        suf[AR_SUF_YTY] = ayy;
        suf[AR_SUF_N] = (double)(T - 1);
      }
    }
  }
  if (gl == 0 && s_live[g] != 0) {
    if (act2) {
      P.yty[chain] = yty;
      P.nobs[chain] = nobs;
    }
    P.status[chain] = s_flag[g];
  }
}

size_t ssm_packed_lds(const SsmParams &M) {
  size_t need = (size_t)PG * ssg_packed_row_doubles(M.m, M.ld, M.nerr, M.nar) * sizeof(double);
  if (need < sizeof(NormalsLds)) need = sizeof(NormalsLds);
  if (need < sizeof(ArLds)) need = sizeof(ArLds);
  return (need + 15) & ~(size_t)15;
}

hipError_t launch_ssm_packed(hipStream_t stream, const SsParams &P, int draw_variances) {
  const dim3 grid((P.chain_count + PG - 1) / PG), block(256);
  const size_t lds = ssm_packed_lds(P.ssm);
  if (lds > 65536) {   // (more than 64 KB of dynamic LDS has to be asked for -- per device, so every time)
    hipError_t err = hipFuncSetAttribute((const void *)ssg_packed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return err;
  }
  hipLaunchKernelGGL(ssg_packed_kernel, grid, block, lds, stream, P, draw_variances);
  return hipGetLastError();
}

}  // namespace boom_amd
