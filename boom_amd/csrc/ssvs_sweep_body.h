// The SSVS sweep of one chain as a device function (see ssvs_kernel.hip for what it computes
// and how): ssvs_kernel.hip wraps it in the sweep kernel, ss_round_kernel.hip runs its
// one-wave form inside the bsts chains' persistent round loop.
#pragma once
#include "ssvs_device.h"

namespace boom_amd {

template <int NB, int W, int WPE>
__device__ __forceinline__ void ssvs_sweep_body(SsvsParams P, int nsweeps, const int chain,
                                                unsigned char *smem) {
  // (the shuffle's mask of finished blocks, ssvs_device.h: the multi-wave instances of capacity
  // 48 / 64 -- large models come with many variables; the headline's <4, 2, 2> and the bsts round's
  // one-wave instances keep the plain loop their single block is fastest with)
  constexpr bool SHUFFLE_MASKED = NB >= 6 && W > 1;
  P.V += (size_t)chain * (size_t)P.v_chain_stride;   // (a chain's own V: the logit sampler's X'WX moves with its latent data)
  if (P.col_valid) {
    P.col_valid += (size_t)chain * (size_t)P.col_words;
    P.v_diag += (size_t)chain * (size_t)P.p;
  }
  int tid_ = threadIdx.x;
  BA_OPAQUE_V(tid_);
  int lane = tid_ & (WAVE - 1);
  const int wave = tid_ >> 6;
  const int p = P.p;
#ifdef BA_PSTAMPS
  long long pst[8];
  pst[0] = (long long)__builtin_readcyclecounter();
#endif
  // State-space rounds: the planes of the X'e GEMM that fed this launch are added here
  // (SsvsParams::xty_planes).  Their loads go out FIRST, ahead of everything the prologue
  // waits for, so that they cost no round trip of their own (a chain's one-sweep launch is
  // a chain of such trips; in place this one was 6-7 k cycles of 95 k).
  constexpr int FOLD_Z = 16;
  const bool fold = (W == 1) && P.xty_planes != nullptr;
  const bool fold_early = fold && P.xty_nplanes <= FOLD_Z && p <= 2 * WAVE;
  double pl[2][FOLD_Z];
  if (fold_early) {
    const double *src = P.xty_planes + (size_t)chain * p;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = lane + e * WAVE;
#pragma unroll
      for (int z = 0; z < FOLD_Z; ++z)
        pl[e][z] = (j < p && z < P.xty_nplanes) ? src[(size_t)z * P.xty_plane_stride + j] : 0.0;
    }
  }
  if (P.status[chain] != CHAIN_OK) {
    // a chain waiting for a larger-capacity kernel (or in error) just books
    // the sweeps it is owed
    if (threadIdx.x == 0) {
      P.todo[chain] += nsweeps;
      if (P.ran) P.ran[chain] = 0;
    }
    return;
  }
  // sweeps to run now: this launch's plus what earlier launches owe, capped by
  // run_limit (the rest stays owed)
  int owed_after = 0;
  {
    int total = nsweeps + P.todo[chain];
    if (P.run_limit > 0 && total > P.run_limit) {
      owed_after = total - P.run_limit;
      total = P.run_limit;
    }
    if (total == 0) {
      if (threadIdx.x == 0 && P.ran) P.ran[chain] = 0;
      return;  // (every wave of the workgroup takes this exit)
    }
    nsweeps = total;
  }

  constexpr int KCAP = NB * 8;
  const SsvsLds lay = ssvs_lds_layout(p, KCAP);
  Chain ch;
  ch.lane = lane;
  ch.p = p;
  ch.k = 0;
  bind_lds(ch, smem, lay);
  lds_f64 *ctl = to_lds<double>(smem + lay.ctrl);
  PSTAMP(1);
  ch.xty = P.xty + (size_t)chain * P.xty_stride;
  if (fold_early) {
    // (plain_reduce_kernel's values in its order: one kernel, its launch gap and a 13 MB
    // round trip less per round)
    double *dst = const_cast<double *>(P.xty) + (size_t)chain * P.xty_stride;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = lane + e * WAVE;
      double a = pl[e][0];
#pragma unroll
      for (int z = 1; z < FOLD_Z; ++z)
        if (z < P.xty_nplanes) a += pl[e][z];
      if (j < p) dst[j] = a;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);   // (vmcnt(0): the stores are in before the sweep's first look)
  } else if (fold) {
    double *dst = const_cast<double *>(P.xty) + (size_t)chain * P.xty_stride;
    const double *src = P.xty_planes + (size_t)chain * p;
    for (int j = lane; j < p; j += WAVE) {
      double a = src[j];
      for (int z0 = 1; z0 < P.xty_nplanes; z0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = (z0 + u < P.xty_nplanes) ? src[(size_t)(z0 + u) * P.xty_plane_stride + j] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (z0 + u < P.xty_nplanes) a += v[u];
      }
      dst[j] = a;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0);   // (vmcnt(0): the stores are in before the sweep's first look)
  }
  PSTAMP(2);
  bind_slot(ch, P, chain, 0);
  ch.sc = (c_f64 *)(unsigned long long)ch.sc_store;
  const double yty = P.yty[(size_t)chain * P.suf_stride];
  const double nobs = P.nobs[(size_t)chain * P.suf_stride];
  ch.DF = nobs + P.prior_df;
  ch.ss0q = P.prior_ss + yty;
  ch.mode = P.mode;
  ch.sv = ch.sa = ch.sx = 1.0;
  if (P.mode) {
    // SpikeSlabSampler works given this chain's sigma^2
    const double inv = 1.0 / P.sigsq[chain];
    ch.sx = inv;
    if (P.slab_scales) { ch.sv = inv; ch.sa = inv; }
  }
  const PhiloxKey key{P.seed_lo, P.seed_hi,
                      (uint32_t)(P.chain_offset + chain), P.stream};

  StampCtx sx_unused;
  sx_unused.last = 0;
  if (W > 1 && wave != 0) {
    // ---- helper waves: serve the master's commands ---------------------------
    sx_unused.last = (long long)__builtin_readcyclecounter();
    for (int i = 0; i < 8; ++i) sx_unused.ph[i] = 0.0;
    for (;;) {
      HSTAMP(sx_unused, 7);
      __syncthreads();
      HSTAMP(sx_unused, 5);
      const int cmd = (int)ctl[CT_CMD];
      if (cmd == CMD_EXIT) break;
      const uint64_t upos = ((AS_LDS const uint64_t *)(ctl + CT_POS))[0];
      if (cmd == CMD_EVAL) {
        Model M;
        M.logp = ctl[CT_LOGP]; M.lp = ctl[CT_LP]; M.ldv = ctl[CT_LDV];
        M.lda = ctl[CT_LDA]; M.Q = ctl[CT_Q]; M.c = ctl[CT_C];
        M.SS = 0; M.pd = true; M.bad = 0;
        ch.k = (int)ctl[CT_K];
        ch.perm = to_lds<uint16_t>(smem + (((int)ctl[CT_PERMSEL]) ? lay.perm1 : lay.perm0));
        bind_slot(ch, P, chain, (int)ctl[CT_CUR]);
        unsigned long long a = (unsigned long long)ch.sc_store;
        asm volatile("" : "+s"(a) : : "memory");
        ch.sc = (c_f64 *)a;
        eval_share<NB>(P, ch, M, key, upos, (int)ctl[CT_NFLIPS], (int)ctl[CT_I0],
                       (int)ctl[CT_EVMODE], wave, ctl, sx_unused);
        HSTAMP(sx_unused, 6);
      } else if (cmd == CMD_DECIDE || cmd == CMD_SHUFFLE_DECIDE) {
        if (cmd == CMD_SHUFFLE_DECIDE) {
          // the shuffle's p - 1 uniforms: every wavefront of the workgroup its share, the
          // master too before it starts on the tail (it has the slack: wave 1's side of a
          // quiet sweep is the longer one) -- then all meet once more
          shuffle_targets(key, upos, p, threadIdx.x, WAVE * W, ch.oth);
          __syncthreads();
        }
        if (wave == 1) {
          const int sel = (int)ctl[CT_PERMSEL];
          bind_slot(ch, P, chain, (int)ctl[CT_CUR]);
          ch.perm = to_lds<uint16_t>(smem + (sel ? lay.perm1 : lay.perm0));
          ch.perm_alt = to_lds<uint16_t>(smem + (sel ? lay.perm0 : lay.perm1));
          uint64_t fpos = upos;
          if (cmd == CMD_SHUFFLE_DECIDE) {
            // the permutation side of a quiet sweep: shuffle(indx) from stream position
            // upos (its targets are in ch.oth: above), then the walk over the new order
            parallel_shuffle<SHUFFLE_MASKED>(ch, sx_unused);
            fpos = upos + (uint64_t)(p - 1);
          }
          DecideResult dr;
          decide_walk(ch, key, fpos, (int)ctl[CT_I0], (int)ctl[CT_NFLIPS], dr);
          HSTAMP(sx_unused, 4);
          if (lane == 0) {
            lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * 1;
            sl[SL_F] = (double)dr.spos;
            sl[SL_J] = (double)dr.j;
            sl[SL_KIND] = (double)dr.kind;
            sl[SL_LOGU] = dr.logu;
            sl[SL_MARGIN] = dr.margin;
          }
        }
      } else {  // CMD_UNIF: this wave's share of the shuffle uniforms
        if (p > 1) shuffle_targets(key, upos, p, threadIdx.x, WAVE * W, ch.oth, P.mode == 2);
      }
      HSTAMP(sx_unused, 7);
      __syncthreads();
      HSTAMP(sx_unused, 5);
    }
#if defined(BA_STAMPS) && defined(BA_STAMPS4)
    if (wave == 1 && lane == 0) {
      double *a = P.acc + (size_t)chain * ACC_COUNT;
      for (int i = 0; i < 8; ++i) a[ACC_PHASE0 + i] += sx_unused.ph[i];
    }
#endif
    return;
  }

  // ---- master wave -------------------------------------------------------------
  uint8_t *g_gamma = P.gamma + (size_t)chain * p;
  uint16_t *g_perm = P.perm + (size_t)chain * p;
  int k = 0;
  int status = CHAIN_OK;
  for (int base = 0; base < p; base += WAVE) {
    const int j = base + lane;
    const int inc = (j < p) ? g_gamma[j] : 0;
    if (j < p) {
      ch.gam[j] = (uint8_t)inc;
      ch.perm[j] = g_perm[j];
      ch.nbr[j] = (uint8_t)((P.cm_start != nullptr) && (P.cm_start[j + 1] > P.cm_start[j]));
    }
    const unsigned long long mask = __ballot(inc != 0);
    const int slot = k + __popcll(mask & ((1ull << lane) - 1ull));
    if (inc && slot < KCAP) ch.g[slot] = (uint16_t)j;
    k += __popcll(mask);
  }
  if (k > KCAP) status = CHAIN_MODEL_TOO_LARGE;
  ch.k = k;
  wave_sync();
  bool aborted = false;      // stopped inside a sweep: restore its start
  int need_col = -1;         // CHAIN_NEED_COLUMN: the variable whose vector of V is missing
  int kmax = k;
  int trace_at = (P.trace_row0 >= 0) ? P.trace_row0 : (P.trace_idx ? P.trace_idx[chain] : 0);

  uint64_t pos = uni((uint64_t)P.rng_pos[chain]);
  int failures = uni((int)P.failures[chain]);
  double sigsq = uni((double)P.sigsq[chain]);
  double beta_m = 0.0;  // lane m: coefficient of variable gprev after the last draw
  int gprev = 0, kprev = 0;
  bool beta_valid = false;

  // The launch's accumulators live in LDS, not in registers (the master's
  // registers are needed for what it touches every few instructions): scalars
  // in the control block (slots CT_ACC + ACC_*), the per-variable summaries of
  // the standing model -- lane m <-> variable sum_g[m] -- in four lane arrays.
  lds_f64 *sum_b = to_lds<double>(smem + lay.park + 512);    // sum of beta
  lds_f64 *sum_b2 = to_lds<double>(smem + lay.park + 1024);  // sum of beta^2
  AS_LDS int *sum_n = to_lds<int>(smem + lay.park + 1536);   // sweeps counted
  AS_LDS int *sum_g = to_lds<int>(smem + lay.park + 1792);   // variable (-1: none)
  sum_b[lane] = 0.0; sum_b2[lane] = 0.0; sum_n[lane] = 0; sum_g[lane] = -1;
  if (lane < 8) ctl[CT_ACC + lane] = (lane == ACC_MIN_MARGIN) ? BA_INF : 0.0;
  wave_sync();
#define ACC_ADD(slot, x) do { if (lane == 0) ctl[CT_ACC + (slot)] += (double)(x); } while (0)
#define ACC_MIN(x) do { if (lane == 0) ctl[CT_ACC + ACC_MIN_MARGIN] = fmin(ctl[CT_ACC + ACC_MIN_MARGIN], (x)); } while (0)
  int done = 0;

  const int nflips = P.max_flips;  // already min(max_nflips_, p)
  STAMP_DECL;
  StampCtx sx;
  sx.last = (long long)__builtin_readcyclecounter();
  for (int i = 0; i < 8; ++i) sx.ph[i] = 0.0;

  // The factors / scalars of the current model are built once per launch and
  // after every change of gamma; they stay valid across sweeps (they depend on
  // gamma only, not on sigma or beta).
  Model M;
  M.bad = 0; M.pd = true; M.logp = 0; M.lp = 0; M.ldv = 0; M.lda = 0; M.Q = 0; M.c = 0; M.SS = 0;
  Pending pe;
  pe.kind = EV_INIT; pe.f1 = pe.f2 = -1; pe.lu = 0; pe.lfw = pe.lrev = 0; pe.check_legal = false;
  enum { PH_BEGIN, PH_FLIPS, PH_SWAP, PH_TAIL, PH_JOIN, PH_COMMIT };
  int phase = PH_BEGIN, sweep = 0, i0 = 0;
  bool model_checked = false;  // legality of the start is checked in sweep 0
  int perm_sel = 0;            // which LDS buffer holds the current permutation
  // walking mode of the current sweep (see the kernel's header comment)
  bool use_table = false;     // this sweep decides by table look-up
  // the table of the chain's last launch is still good when nothing but
  // sweeps happened since (the host clears table_keep otherwise)
  // which of the chain's two (table, model block) slots is in use, and what the
  // other one holds: the model one flip (of variable other_var) away, or nothing
  int cur = 0, other_var = -1;
  bool other_ok = false, table_valid_other = false;
  const int model_tag_in = P.model_keep ? P.model_tag[chain] : 0;
  const bool model_kept = (model_tag_in & 0xff) == KCAP && status == CHAIN_OK;
  if (model_kept) cur = (model_tag_in >> 8) & 1;
  bind_slot(ch, P, chain, cur);
  bool table_valid = P.table_keep && model_kept && P.table_tag[chain] == model_tag_in;
  int fill_j = 0;             // next variable of a fill in progress
  // Quiet sweeps fork: wave 1 shuffles and walks the table while the master
  // runs the sweep's tail (swap proposal, sigma, beta) on the assumption that
  // no flip will be accepted -- the tail's stream position is known up front
  // (shuffle and flips consume p - 1 + nflips numbers whatever happens).  At
  // the join a stop in the walk rolls the tail back.
  bool spec = false;          // a forked walk is outstanding
  bool have_dr = false;       // wave 1's slot holds a walk result not yet handled
  int after_join = PH_COMMIT, spec_status = CHAIN_OK;
  // Between a join and the next fork wave 1 has nothing to do, so the master
  // keeps that stretch short: a joined sweep's summaries are committed AFTER the
  // next sweep has been forked, and the sweep-start copy of gamma is skipped
  // while gamma has not moved.
  bool commit_pending = false, gam0_fresh = false;
  // (what a roll-back restores is parked in LDS, not in registers: park[lane] =
  // beta_m, control-block slots CT_ROLL.. = sigma^2, failures, beta_valid)
  lds_f64 *park = to_lds<double>(smem + lay.park);
  int stops_prev = table_valid ? 0 : (1 << 20), stops_now = 0;
  uint64_t flip_pos = 0, pos0 = pos;
  WinRng rng;
  rng.init(key, lane, pos);

  PSTAMP(3);
  // The chain's model block of the last launch is still this model (nothing but
  // sweeps happened since): take the factors from there instead of factoring.
  if (model_kept) {
    {
      // (the model's scalars are asked for BEFORE the factors: one round trip for both)
      const SsvsScalarLayout S = ssvs_scalar_layout(KCAP);
      const double *sc = ch.sc_store + S.scal;
      const double s0 = sc[0], s1 = sc[1], s2 = sc[2], s3 = sc[3], s4 = sc[4], s5 = sc[5], s6 = sc[6], s7 = sc[7];
      restore_model<NB>(ch);
      M.logp = s0; M.lp = s1; M.ldv = s2; M.lda = s3;
      M.Q = s4; M.c = s5; M.SS = s6; M.pd = s7 != 0.0;
      unsigned long long u = (unsigned long long)ch.sc_store;
      asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(u) : : "memory");
      ch.sc = (c_f64 *)u;
    }
    if (P.suf_changed) {
      refactor<true>(P, ch, M, sx);
      if (M.bad) status = M.bad;
      else publish_model<NB>(ch, M);
    }
    if (lane == 0) {
      ctl[CT_LOGP] = M.logp; ctl[CT_LP] = M.lp; ctl[CT_LDV] = M.ldv;
      ctl[CT_LDA] = M.lda; ctl[CT_Q] = M.Q; ctl[CT_C] = M.c;
    }
    wave_sync();
    pe.kind = EV_NONE;
  }

  PSTAMP(4);
  while (status == CHAIN_OK) {
    if constexpr ((W > 1 && !(NB == 4 && W == 2)) || (W == 1 && NB >= 4)) {
      // (every multi-wave instance but the headline's <4, 2, 2>, which needs no scratch as it
      // is: what the compiler derives from the lane number once before this loop
      // -- two dozen per-lane addresses -- it then keeps in scratch memory for the whole launch;
      // made opaque per pass of the state machine, they are recomputed where they are used)
      asm volatile("" : "+v"(lane));
      ch.lane = lane;
    }
    if (pe.kind != EV_NONE && !spec) {
      // ---- the one place where a model is (re)built (a swap proposed by a
      // tail running ahead waits for the join) --------------------------
      // (outside this block only logp, SS and pd of the model live in registers;
      // the scalars the evaluations need have their home in the control block)
      if (pe.kind == EV_FORCE && pe.f2 < 0 && other_ok && pe.f1 == other_var) {
        gam0_fresh = false;
        // The accepted flip leads to the model the other slot still holds (a
        // variable leaving again, or coming back): its factors, scalars and
        // table are there -- bitwise what a rebuild would compute.
        apply_flip(ch, pe.f1);
        cur ^= 1;
        bind_slot(ch, P, chain, cur);
        restore_model<NB>(ch);
        {
          const SsvsScalarLayout S = ssvs_scalar_layout(KCAP);
          const double *sc = ch.sc_store + S.scal;
          M.logp = sc[0]; M.lp = sc[1]; M.ldv = sc[2]; M.lda = sc[3];
          M.Q = sc[4]; M.c = sc[5]; M.SS = sc[6]; M.pd = sc[7] != 0.0;
          unsigned long long u = (unsigned long long)ch.sc_store;
          asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(u) : : "memory");
          ch.sc = (c_f64 *)u;
        }
        if (lane == 0) {
          ctl[CT_LOGP] = M.logp; ctl[CT_LP] = M.lp; ctl[CT_LDV] = M.ldv;
          ctl[CT_LDA] = M.lda; ctl[CT_Q] = M.Q; ctl[CT_C] = M.c;
        }
        wave_sync();
        {  // the slot left behind keeps the model just left: one flip of the same variable away
          const bool t = table_valid;
          table_valid = table_valid_other;
          table_valid_other = t;
        }
        ACC_ADD(ACC_ACCEPTS, 1);
#ifndef BA_STAMPS
        ACC_ADD(ACC_SLOT_HITS, 1);
#endif
        if (!M.pd) status = CHAIN_NOT_PD;
        pe.kind = EV_NONE;
        pe.f1 = pe.f2 = -1;
        pe.lfw = pe.lrev = 0.0;
        pe.check_legal = false;
        STAMP(2);
        continue;
      }
      gam0_fresh = false;
      Model keep;   // (member by member: a struct copy goes through a stack slot)
      keep.logp = M.logp; keep.SS = M.SS; keep.pd = M.pd; keep.bad = M.bad;
      keep.lp = ctl[CT_LP]; keep.ldv = ctl[CT_LDV]; keep.lda = ctl[CT_LDA];
      keep.Q = ctl[CT_Q]; keep.c = ctl[CT_C];
      if (pe.f1 >= 0) apply_flip(ch, pe.f1);
      if (pe.f2 >= 0) apply_flip(ch, pe.f2);
      bool rejected = false;
      {
        Model Mn;
        refactor<false>(P, ch, Mn, sx);
        if (Mn.bad) {
          status = Mn.bad;
        } else {
          bool acc = true;
          if (pe.kind == EV_TRY_GE || pe.kind == EV_TRY_LT) {
            const double d = (Mn.logp - pe.lfw) - (keep.logp - pe.lrev);
            if (Mn.logp > -BA_INF) ACC_MIN(fabs(pe.lu - d));
            acc = (pe.kind == EV_TRY_GE) ? !(pe.lu > d) : (pe.lu < d);
          }
          if (acc) {
            M = Mn;
            M.logp = uni(M.logp);
            M.SS = uni(M.SS);
            // (the launch's first build is the old model unless make_valid
            // changed gamma)
            if (pe.kind != EV_INIT || pe.check_legal) {
              // a new model: it goes to the other slot, the one in use keeps the
              // model being left (reachable again by one flip if one flip led here)
              table_valid_other = table_valid;
              table_valid = false;
              other_ok = (pe.kind != EV_INIT) && pe.f1 >= 0 && pe.f2 < 0;
              other_var = pe.f1;
              cur ^= 1;
              bind_slot(ch, P, chain, cur);
            }
            if (pe.kind != EV_INIT) ACC_ADD(ACC_ACCEPTS, 1);
          } else {
            // rejected: gamma back, and the old factors from the chain's block
            if (pe.f2 >= 0) apply_flip(ch, pe.f2);
            if (pe.f1 >= 0) apply_flip(ch, pe.f1);
            restore_model<NB>(ch);
            M.logp = keep.logp; M.lp = keep.lp; M.ldv = keep.ldv; M.lda = keep.lda;
            M.Q = keep.Q; M.c = keep.c; M.SS = keep.SS; M.pd = keep.pd; M.bad = keep.bad;
            rejected = true;
          }
        }
      }
      if (status == CHAIN_OK) {
        if (!rejected) {
          publish_model<NB>(ch, M);
          if (lane == 0) {
            ctl[CT_LOGP] = M.logp; ctl[CT_LP] = M.lp; ctl[CT_LDV] = M.ldv;
            ctl[CT_LDA] = M.lda; ctl[CT_Q] = M.Q; ctl[CT_C] = M.c;
          }
          wave_sync();
        }
        if (pe.kind == EV_FORCE && !M.pd) status = CHAIN_NOT_PD;
        if (pe.kind == EV_INIT && pe.check_legal &&
            !(M.logp > -BA_INF && M.logp < BA_INF))
          status = CHAIN_ILLEGAL_START;
      }
      pe.kind = EV_NONE;
      pe.f1 = pe.f2 = -1;
      pe.lfw = pe.lrev = 0.0;
      pe.check_legal = false;
      STAMP(2);
      continue;
    }

    if (phase == PH_BEGIN) {
      if (sweep + (commit_pending ? 1 : 0) >= nsweeps) {
        if (!commit_pending) break;
        phase = PH_COMMIT;  // the last sweep's summaries, then out
        continue;
      }
      {
        const bool ut = (P.walk_policy != 0) && (stops_prev <= 1 || P.walk_policy == 2);
        const bool mc = model_checked || (M.logp > -BA_INF && M.logp < BA_INF);
        const bool will_fork = nflips > 0 && W > 1 && ut && table_valid && mc && p > 1 && P.walk_policy != 3 && P.mode != 2;
        if (commit_pending && !will_fork) {
          phase = PH_COMMIT;  // nothing to overlap with: commit first
          continue;
        }
      }
      STAMP(7);
      TSTAMP(sx, 7);
      if (nflips > 0) {
        // remember the sweep's starting point (restored if the chain has to
        // stop inside this sweep for lack of model capacity)
        if (!gam0_fresh || P.mode) {
          for (int j = lane; j < p; j += WAVE) {
            ch.gam0[j] = ch.gam[j];
            // SpikeSlabSampler shuffles a fresh identity permutation every call
            // (SpikeSlabSampler.cpp:48-57); BregVsSampler's indx persists
            if (P.mode) ch.perm[j] = (uint16_t)j;
          }
          gam0_fresh = true;
        }
        TSTAMP(sx, 1);
        pos0 = pos;
        if (!model_checked && M.logp > -BA_INF && M.logp < BA_INF) model_checked = true;
        use_table = (P.walk_policy != 0) && (stops_prev <= 1 || P.walk_policy == 2);
        stops_prev = stops_now;
        stops_now = 0;
        if (W > 1 && use_table && table_valid && model_checked && p > 1 && P.walk_policy != 3 && P.mode != 2) {
          // ---- fork: wave 1 takes the permutation side of the sweep
          wave_sync();
          if (lane == 0) {
            ctl[CT_CMD] = (double)CMD_SHUFFLE_DECIDE;
            ctl[CT_I0] = 0.0;
            ctl[CT_NFLIPS] = (double)nflips;
            ctl[CT_PERMSEL] = (double)perm_sel;
            ctl[CT_CUR] = (double)cur;
            ((AS_LDS uint64_t *)(ctl + CT_POS))[0] = pos;
          }
          __syncthreads();
          shuffle_targets(key, pos, p, threadIdx.x, WAVE * W, ch.oth);   // (the master's share: see the helpers' loop)
          __syncthreads();
          {  // the shuffled order will be in the other buffer
            lds_u16 *tmp = ch.perm;
            ch.perm = ch.perm_alt;
            ch.perm_alt = tmp;
            perm_sel ^= 1;
          }
          flip_pos = pos + (uint64_t)(p - 1);
          pos = flip_pos + (uint64_t)nflips;
          TSTAMP(sx, 2);
          spec = true;
          spec_status = CHAIN_OK;
          park[lane] = beta_m;
          if (lane == 0) {
            ctl[CT_ROLL + 0] = sigsq;
            ctl[CT_ROLL + 1] = (double)failures;
            ctl[CT_ROLL + 2] = beta_valid ? 1.0 : 0.0;
          }
          i0 = 0;
          phase = commit_pending ? PH_COMMIT : PH_SWAP;  // (the joined sweep's summaries ride along)
          STAMP(1);
          continue;
        }
        // ---- shuffle(indx): cpputil/shuffle.hpp:36-46, in place on the
        // persistent permutation.  Uniform t (t = 0..p-2) belongs to i = p-1-t.
        if (W > 1) {
          if (lane == 0) {
            ctl[CT_CMD] = (double)CMD_UNIF;
            ((AS_LDS uint64_t *)(ctl + CT_POS))[0] = pos;
          }
          __syncthreads();
        }
        if (p > 1) shuffle_targets(key, pos, p, threadIdx.x, WAVE * W, ch.oth, P.mode == 2);
        if (W > 1) __syncthreads(); else wave_sync();
        STAMP(0);
        if (P.mode == 2) {
          // BinomialLogitSpikeSlabSampler's shuffle: for i = 0..p-1 swap(indx[i],
          // indx[j_i]), j_i anywhere in the range -- a later step can move what an
          // earlier one placed, so the steps run in order (a sweep of this
          // sampler is dominated by its n p^2 sufficient statistics anyway)
          if (lane == 0 && p > 1) {
            for (int i = 0; i < p; ++i) {
              const int j = ch.oth[i];
              const uint16_t a = ch.perm[i];
              ch.perm[i] = ch.perm[j];
              ch.perm[j] = a;
            }
          }
          wave_sync();
        } else if (p > 1) { parallel_shuffle<SHUFFLE_MASKED>(ch, sx); perm_sel ^= 1; }
        flip_pos = pos + (uint64_t)(P.mode == 2 ? p : (p > 0 ? p - 1 : 0));
        pos = flip_pos + (uint64_t)nflips;
        STAMP(1);
        if (!model_checked) {
          model_checked = true;
          if (!(M.logp > -BA_INF && M.logp < BA_INF)) {
            // VariableSelectionPrior::make_valid, VariableSelectionPrior.cpp:287-300
            for (int j = 0; j < p; ++j) {
              const double pj = P.pi[j];
              const bool inc = ch.gam[j];
              if ((pj <= 0.0 && inc) || (pj >= 1.0 && !inc)) {
                if (!inc && ch.k >= KCAP) { status = CHAIN_MODEL_TOO_LARGE; aborted = true; break; }
                if (!inc && column_missing(P, j)) { status = CHAIN_NEED_COLUMN; need_col = j; aborted = true; break; }
                apply_flip(ch, j);
              }
            }
            pe.kind = EV_INIT;
            pe.check_legal = true;
          }
        }
      }
      i0 = 0;
      phase = PH_FLIPS;
      continue;
    }

    if (phase == PH_FLIPS) {
      if (nflips == 0 || i0 >= nflips) {
        phase = PH_SWAP;
        continue;
      }
      // ---- Metropolised flips, 64 * W proposals per round
      if (use_table && table_valid) {
        // ---- decisions by table look-up, up to the first stop.  With helper
        // waves the walk is wave 1's job (the master's registers are full of
        // chain state; wave 1 has none), otherwise the master's own.
        DecideResult dr;
        if (W > 1) {
          if (!have_dr) {
          if (lane == 0) {
            ctl[CT_CMD] = (double)CMD_DECIDE;
            ctl[CT_I0] = (double)i0;
            ctl[CT_NFLIPS] = (double)nflips;
            ctl[CT_PERMSEL] = (double)perm_sel;
            ctl[CT_CUR] = (double)cur;
            ((AS_LDS uint64_t *)(ctl + CT_POS))[0] = flip_pos;
          }
          __syncthreads();
          __syncthreads();
          }
          have_dr = false;
          const lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * 1;
          dr.spos = uni((int)sl[SL_F]);
          dr.j = uni((int)sl[SL_J]);
          dr.kind = uni((int)sl[SL_KIND]);
          dr.logu = uni((double)sl[SL_LOGU]);
          dr.margin = uni((double)sl[SL_MARGIN]);
        } else {
          decide_walk(ch, key, flip_pos, i0, nflips, dr);
        }
        ACC_MIN(dr.margin);
        STAMP(3);
        if (dr.spos < 0) {  // walked to the end without a stop
          ACC_ADD(ACC_PROPOSALS, nflips - i0);
          i0 = nflips;
          continue;
        }
        ACC_ADD(ACC_PROPOSALS, dr.spos + 1 - i0);
        ++stops_now;
        i0 = dr.spos + 1;
        if (dr.kind == STOP_BAD) {
          status = CHAIN_NEGATIVE_SS;
          break;
        }
        if (!ch.gam[dr.j] && ch.k >= KCAP) {
          status = CHAIN_MODEL_TOO_LARGE;
          aborted = true;
          break;
        }
        if (!ch.gam[dr.j] && column_missing(P, dr.j)) {
          status = CHAIN_NEED_COLUMN;
          need_col = dr.j;
          aborted = true;
          break;
        }
        pe.f1 = dr.j;
        if (dr.kind == 0) {
          pe.kind = EV_FORCE;
        } else {
          pe.kind = EV_TRY_GE;
          pe.lu = dr.logu;
        }
        continue;
      }
      int evmode = EVM_BATCH, base = i0;
      if (use_table) {
        evmode = EVM_FILL;   // (re)build the table for the current model
        base = fill_j;
      }
      if (W > 1) {
        if (lane == 0) {
          ctl[CT_CMD] = (double)CMD_EVAL;
          ctl[CT_K] = (double)ch.k;
          ctl[CT_I0] = (double)base;
          ctl[CT_NFLIPS] = (double)nflips;
          ctl[CT_PERMSEL] = (double)perm_sel;
            ctl[CT_CUR] = (double)cur;
          ctl[CT_EVMODE] = (double)evmode;
          ((AS_LDS uint64_t *)(ctl + CT_POS))[0] = flip_pos;
        }
        __syncthreads();
      }
      {
        Model Me;
        Me.logp = M.logp; Me.lp = ctl[CT_LP]; Me.ldv = ctl[CT_LDV]; Me.lda = ctl[CT_LDA];
        Me.Q = ctl[CT_Q]; Me.c = ctl[CT_C]; Me.SS = 0; Me.pd = true; Me.bad = 0;
        eval_share<NB>(P, ch, Me, key, flip_pos, nflips, base, evmode, 0, ctl, sx);
      }
      if (W > 1) __syncthreads(); else wave_sync();
      if (evmode == EVM_FILL) {
        fill_j += WAVE * W;
        if (fill_j >= p) {
          fill_j = 0;
          table_valid = true;
        }
        STAMP(3);
        continue;
      }
      // first stop over the whole round, in sweep order
      int wstop = -1, spos = 0;
#pragma unroll
      for (int w = 0; w < W; ++w) {
        const lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * w;
        if (wstop < 0) {
          ACC_MIN(sl[SL_MARGIN]);
          const int fw = uni((int)sl[SL_F]);
          if (fw >= 0) { wstop = w; spos = fw; }
        }
      }
      STAMP(3);
      if (wstop < 0) {
        const int n = (nflips - i0 < WAVE * W) ? (nflips - i0) : WAVE * W;
        ACC_ADD(ACC_PROPOSALS, n);
        i0 += WAVE * W;
        continue;
      }
      const lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * wstop;
      const int jf = uni((int)sl[SL_J]);
      const int kind = uni((int)sl[SL_KIND]);
      const int nprop = spos + 1 - i0;
      ACC_ADD(ACC_PROPOSALS, nprop);
      ++stops_now;
      if (kind == STOP_BAD) {
        status = CHAIN_NEGATIVE_SS;
        break;
      }
      if (!ch.gam[jf] && ch.k >= KCAP) {
        // the candidate cannot be held in LDS; if it is a sure rejection that
        // is fine, but we cannot tell without evaluating it
        status = CHAIN_MODEL_TOO_LARGE;
        aborted = true;
        break;
      }
      if (!ch.gam[jf] && column_missing(P, jf)) {
        // the model with jf needs vector jf of this chain's V: the host computes it
        status = CHAIN_NEED_COLUMN;
        need_col = jf;
        aborted = true;
        break;
      }
      pe.f1 = jf;
      if (kind == STOP_ACCEPT) {
        pe.kind = EV_FORCE;  // accepted on the fast path: move to the new model
      } else {
        pe.kind = EV_TRY_GE;  // exact path: evaluate the flipped model
        pe.lu = uni((double)sl[SL_LOGU]);
      }
      i0 += nprop;
      continue;
    }

    if (phase == PH_SWAP) {
      TSTAMP(sx, 7);
      rng.set_pos(pos);
      if (nflips > 0) propose_swap(P, ch, rng, pe, &status);
      pos = rng.get_pos();
      phase = PH_TAIL;
      if (spec && (pe.kind != EV_NONE || status != CHAIN_OK)) {
        // a swap was proposed (or the proposal failed): its evaluation has to
        // wait for the walk
        spec_status = status;
        status = CHAIN_OK;
        after_join = PH_TAIL;
        phase = PH_JOIN;
      }
      STAMP(4);
      TSTAMP(sx, 3);
      continue;
    }

    if (phase == PH_JOIN) {
      __syncthreads();
      spec = false;
      const lds_f64 *sl = ctl + CT_SLOT0 + CT_SLOT_STRIDE * 1;
      if ((int)sl[SL_F] < 0) {
        // no stop: the sweep's flips are all rejected and what ran ahead stands
        ACC_MIN(sl[SL_MARGIN]);
        ACC_ADD(ACC_PROPOSALS, nflips);
        status = spec_status;
        phase = after_join;
        if (after_join == PH_COMMIT && status == CHAIN_OK) {
          commit_pending = true;  // committed after the next fork
          phase = PH_BEGIN;
        }
      } else {
        // roll the tail back and handle the stop
        pos = flip_pos + (uint64_t)nflips;
        // (gprev / kprev are only set at a commit)
        beta_m = park[lane];
        sigsq = ctl[CT_ROLL + 0];
        failures = (int)ctl[CT_ROLL + 1];
        beta_valid = ctl[CT_ROLL + 2] != 0.0;
        pe.kind = EV_NONE; pe.f1 = pe.f2 = -1; pe.lfw = pe.lrev = 0.0; pe.check_legal = false;
        have_dr = true;
        phase = PH_FLIPS;
      }
      STAMP(3);
      continue;
    }

    if (phase == PH_COMMIT) {
      // ---- summaries
      TSTAMP(sx, 7);
      k = ch.k;
      gprev = (lane < k) ? (int)ch.g[lane] : 0;
      kprev = k;
      kmax = k > kmax ? k : kmax;
      // inclusion counts and coefficient moments pile up on chip while the
      // model stands still (lane m <-> variable sum_g[m]) and go to HBM when it
      // moves
      {
        const int gnow = (lane < k) ? gprev : -1;
        const int gold = sum_g[lane];
        if (__any(gnow != gold)) {
          const int n_old = sum_n[lane];
          if (gold >= 0 && n_old) {
            // (the three read-modify-writes as three loads, then three stores: written one
            // after the other they were three memory round trips in a row -- the compiler
            // cannot know that the arrays do not overlap)
            const size_t o = (size_t)chain * p + gold;
            const unsigned c0 = P.inc_count[o];
            const double b0 = P.beta_sum[o], q0 = P.beta_sumsq[o];
            P.inc_count[o] = c0 + (unsigned)n_old;
            P.beta_sum[o] = b0 + sum_b[lane];
            P.beta_sumsq[o] = q0 + sum_b2[lane];
          }
          sum_g[lane] = gnow; sum_n[lane] = 0; sum_b[lane] = 0.0; sum_b2[lane] = 0.0;
        }
        if (lane < k) {
          sum_n[lane] += 1;
          if (beta_valid) { sum_b[lane] += beta_m; sum_b2[lane] += beta_m * beta_m; }
        }
      }
      ACC_ADD(ACC_SIGSQ, sigsq);
      ACC_ADD(ACC_SIGSQ2, sigsq * sigsq);
      ACC_ADD(ACC_K, k);
      if (P.trace_sigsq && trace_at + sweep < P.trace_stride) {
        const size_t o = (size_t)chain * P.trace_stride + trace_at + sweep;
        if (lane == 0) {
          P.trace_sigsq[o] = sigsq;
          P.trace_logp[o] = M.logp;
          P.trace_k[o] = (double)k;
        }
        if (P.rec_idx && lane < k) {  // the sweep's draw itself (recording step)
          P.rec_idx[o * P.rec_cap + lane] = (uint16_t)gprev;
          P.rec_beta[o * P.rec_cap + lane] = beta_valid ? beta_m : 0.0;
        }
      }
      ++done;
      ++sweep;
      commit_pending = false;
      phase = spec ? PH_SWAP : PH_BEGIN;  // (spec: this was the previous sweep's commit, riding on a fork)
      TSTAMP(sx, 0);
      continue;
    }

    // ---- PH_TAIL: sigma, beta, summaries
    TSTAMP(sx, 7);
    k = ch.k;
    rng.set_pos(pos);
    // draw_sigma (BregVsSampler.cpp:313-324)
    if (P.draw_sigma) {
      int bad = 0;
      const double DF = (k == 0) ? ch.DF : ((ch.DF - P.prior_df) + P.prior_df);
      const double SS = (k == 0) ? ch.ss0q : ((M.SS - P.prior_ss) + P.prior_ss);
      sigsq = uni(d_draw_variance(rng, DF, SS, P.sigma_max, &bad));
      if (bad) {
        if (!spec) { status = CHAIN_RNG_BRANCH; break; }
        spec_status = CHAIN_RNG_BRANCH; after_join = PH_COMMIT; phase = PH_JOIN;
        continue;
      }
    }
    pos = uni(rng.get_pos());
    STAMP(5);
    TSTAMP(sx, 4);
    // draw_beta (BregVsSampler.cpp:326-351)
    if (P.draw_beta && k > 0) {
      if (!M.pd) {
        ++failures;
        if (!spec) { status = CHAIN_NOT_PD; break; }
        spec_status = CHAIN_NOT_PD; after_join = PH_COMMIT; phase = PH_JOIN;
        continue;
      }
      failures = 0;
      // k standard normals in stream order (distributions/mvn.cpp:114-122),
      // lane m keeps z_m
      const double z = draw_normals(rng, k);
      pos = uni(rng.get_pos());
      TSTAMP(sx, 5);
      // beta = L^{-T}(w + sigma z): chol(V / sigma^2) = L / sigma
      // (SpikeSlabSampler: rmvn_ivar_mt with the sigma-scaled precision itself)
      const double sigma = P.mode ? 1.0 : sqrt(sigsq);
      double y = (lane < k) ? ch.w[lane] + sigma * z : 0.0;
      const double rdm = (lane < k) ? ch.rdv[lane] : 0.0;
      // column sweep of the back substitution, eight rows of L at a time: a block row's
      // eight reads go out together (rows of one 8 x 8 block row are 64 bytes apart), then its
      // eight steps run from registers -- fetched one row ahead of its use, every step had
      // waited for most of an LDS round trip
#pragma nounroll
      for (int ib = (k - 1) >> 3; ib >= 0; --ib) {
        double lr[8];
        const int base = bidx(ib * 8, lane < ib * 8 + 8 ? lane : 0);   // (row 8 ib, this lane's column)
#pragma unroll
        for (int t = 0; t < 8; ++t) lr[t] = ch.Lv[base + t * 8];
#pragma unroll
        for (int t = 7; t >= 0; --t) {
          const int i = ib * 8 + t;
          if (i < k) {
            const double xi = bcast_u(y * rdm, i);
            if (lane == i) y = xi;
            else if (lane < i) y -= lr[t] * xi;
          }
        }
      }
      beta_m = y;
      beta_valid = true;
      TSTAMP(sx, 6);
    } else if (P.draw_beta) {
      beta_valid = true;  // empty model: all coefficients zero
    }
    STAMP(6);
    after_join = PH_COMMIT;
    phase = spec ? PH_JOIN : PH_COMMIT;
  }

  PSTAMP(5);
  // release the helper waves
  if (W > 1) {
    if (lane == 0) ctl[CT_CMD] = (double)CMD_EXIT;
    __syncthreads();
  }

  wave_sync();
#ifndef BA_STAMPS
  // (the chain's scalar accumulators, lane i <-> slot i: read here, with the summaries'
  // loads, written at the end -- not a round trip of lane 0's own after everything else)
  double *acc_row = P.acc + (size_t)chain * ACC_COUNT;
  const double acc_old = (lane < 8) ? acc_row[lane] : 0.0;
#endif
  if (sum_g[lane] >= 0 && sum_n[lane]) {
    const size_t o = (size_t)chain * p + sum_g[lane];
    P.inc_count[o] += (unsigned)sum_n[lane];
    P.beta_sum[o] += sum_b[lane];
    P.beta_sumsq[o] += sum_b2[lane];
  }
  // ---- write the chain back (an aborted sweep leaves no trace: gamma, the
  // permutation and the stream position go back to the sweep's start; sigma,
  // beta are those of the last complete sweep anyway)
  wave_sync();
  {
    const lds_u8 *gsrc = aborted ? ch.gam0 : ch.gam;
    const lds_u16 *psrc = (aborted && p > 1) ? ch.perm_alt : ch.perm;
    for (int j = lane; j < p; j += WAVE) {
      g_gamma[j] = gsrc[j];
      g_perm[j] = psrc[j];
    }
    if (aborted) pos = pos0;
  }
  if (beta_valid) {
    double *g_beta = P.beta + (size_t)chain * p;
    for (int j = lane; j < p; j += WAVE) g_beta[j] = 0.0;
    wave_sync();
    if (lane < kprev) g_beta[gprev] = beta_m;
  } else if (nflips > 0 && done > 0) {
    // coef().set_inc(g) zeroes the coefficients of excluded variables
    // (Models/Glm/GlmCoefs.cpp:89-94) even when the beta draw is suppressed
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
    if (P.col_request) P.col_request[chain] = need_col;
    P.todo[chain] = nsweeps - done + owed_after;
    if (P.ran) P.ran[chain] = done;
    const int tag = KCAP | (cur << 8);
    P.table_tag[chain] = (table_valid && !aborted && status == CHAIN_OK) ? tag : 0;
    P.model_tag[chain] = (!aborted && status == CHAIN_OK) ? tag : 0;
    if (P.trace_idx) P.trace_idx[chain] = trace_at + done;
    if (P.maxk) atomicMax(P.maxk, kmax);
#ifdef BA_STAMPS
    double *a = P.acc + (size_t)chain * ACC_COUNT;
    a[ACC_SWEEPS] += done;
    a[ACC_SIGSQ] += ctl[CT_ACC + ACC_SIGSQ];
    a[ACC_SIGSQ2] += ctl[CT_ACC + ACC_SIGSQ2];
    a[ACC_K] += ctl[CT_ACC + ACC_K];
    a[ACC_ACCEPTS] += ctl[CT_ACC + ACC_ACCEPTS];
    a[ACC_PROPOSALS] += ctl[CT_ACC + ACC_PROPOSALS];
    a[ACC_MIN_MARGIN] = fmin(a[ACC_MIN_MARGIN], ctl[CT_ACC + ACC_MIN_MARGIN]);
#endif
#ifdef BA_PSTAMPS
    PSTAMP(6);
    if (chain == 0 || chain == 517)
      printf("ssvs chain %d: status/todo %lld, plane sum %lld, gamma+scalars %lld, restore+refactor %lld, sweeps %lld, epilogue %lld\n",
             chain, pst[1] - pst[0], pst[2] - pst[1], pst[3] - pst[2], pst[4] - pst[3], pst[5] - pst[4], pst[6] - pst[5]);
#endif
#if defined(BA_STAMPS) && defined(BA_STAMPS4)
    // (phases are wave 1's)
#elif defined(BA_STAMPS) && (defined(BA_STAMPS2) || defined(BA_STAMPS3))
    SUBSTAMP(sx, 7);
    for (int i = 0; i < 8; ++i) a[ACC_PHASE0 + i] += sx.ph[i];
#elif defined(BA_STAMPS)
    for (int i = 0; i < 8; ++i) { a[ACC_PHASE0 + i] += st_ph[i]; a[ACC_SLOT_HITS] += st_ph[i]; }
#endif
  }
#ifndef BA_STAMPS
  if (lane < 8) {
    const double inc = (lane == ACC_SWEEPS) ? (double)done : ctl[CT_ACC + lane];
    acc_row[lane] = (lane == ACC_MIN_MARGIN) ? fmin(acc_old, inc) : acc_old + inc;
  }
#endif
}

}  // namespace boom_amd
