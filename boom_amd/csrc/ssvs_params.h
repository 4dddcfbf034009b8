// Launch parameters and memory layout shared by the host side (engine.hip) and
// the SSVS sweep kernel (ssvs_kernel.hip).
#pragma once
#include <stdint.h>

namespace boom_amd {

// per-chain status words written by the kernels (0 = ok)
enum ChainStatus : int32_t {
  CHAIN_OK = 0,
  CHAIN_NOT_PD = 1,
  CHAIN_NEGATIVE_SS = 2,
  CHAIN_ILLEGAL_START = 3,
  CHAIN_RNG_BRANCH = 4,
  CHAIN_FORECAST_VARIANCE = 5,
  CHAIN_MODEL_TOO_LARGE = 6,
  // (never seen by a caller) the sweep stopped at a proposal to ADD a variable whose
  // column of the chain's own V has not been computed for this sweep's latent data:
  // the host computes it (col_request names it) and the sweep is replayed
  CHAIN_NEED_COLUMN = 7,
  CHAIN_NEED_COLUMN_BIG = 8   // the same, from the large-model kernel
};

// number of doubles in the reduced summary block: 3p + SUMMARY_SCALARS
enum { SUMMARY_SCALARS = 16 };
// per-chain scalar accumulators (doubles)
enum {
  ACC_SWEEPS = 0,
  ACC_SIGSQ = 1,
  ACC_SIGSQ2 = 2,
  ACC_K = 3,
  ACC_ACCEPTS = 4,
  ACC_PROPOSALS = 5,
  ACC_MIN_MARGIN = 6,
  ACC_SLOT_HITS = 7,  // accepted flips served from the chain's other slot (diagnostic stamp builds: slowest chain's cycles)
  // 8..15: per-phase cycle counters, filled only by the -DBA_STAMPS diagnostic
  // build (shuffle uniforms, shuffle serial, refactor, proposal batches, swap,
  // sigma, beta, rest); zero in the production library
  ACC_PHASE0 = 8,
  ACC_COUNT = 16
};

struct SsvsParams {
  int32_t p;
  int32_t chains;        // chains of the engine (array extents)
  int32_t chain_first;   // this launch works on chains [chain_first, chain_first + chain_count)
  int32_t chain_count;
  int64_t chain_offset;
  int32_t kcap;  // largest model the LDS working set can hold: 16/32/48/64
  int32_t waves; // wavefronts per chain (1, 2 or 4)
  // 0: BregVsSampler (sigma^2 integrated out); 1: SpikeSlabSampler (given sigma^2);
  // 2: BinomialLogitSpikeSlabSampler -- as 1, with its own way of shuffling the visiting
  // order (every position swaps with one drawn from the whole range: p uniforms) and V per chain
  int32_t mode;
  int64_t v_chain_stride;  // doubles between the chains' V matrices (0: one shared V)
  // A chain's own V computed column by column, as the sweep needs it (the logit
  // sampler: V_c = Omega^{-1} + X'W_cX moves with the latent data of every sweep, and a
  // sweep reads only the columns of variables that are or become included, p k of its
  // p^2 elements).  col_valid: bit j of chain c's words = vector j of V_c (elements
  // (., j), at V_c + j p) holds this sweep's values; v_diag: the diagonal, chains x p.
  // A proposal to add j with the bit clear parks the chain (CHAIN_NEED_COLUMN,
  // col_request[chain] = j).  nullptr: V is complete.
  const uint32_t *col_valid;
  int32_t col_words;       // words per chain
  int32_t *col_request;    // chains
  const double *v_diag;    // chains x p
  // how a sweep's proposals are walked (ssvs_kernel.hip): 0 batch mode only, 1
  // adaptive (table look-ups after quiet sweeps; the default), 2 always the
  // table, 3 adaptive without the forked quiet sweep (diagnostic A/B)
  int32_t walk_policy;
  int32_t slab_scales;  // mode 1: slab precision is Omega^{-1} / sigma^2 (MvnGivenScalarSigma)

  // shared, read-only (HBM; L2 / Infinity-Cache resident in practice)
  const double *V;    // XtX + Omega^{-1}, p x p (symmetric, full storage)
  const double *A;    // Omega^{-1}, p x p
  const double *b;    // prior mean, p
  const double *l1;   // log pi_j
  const double *l0;   // log(1 - pi_j)
  const double *pi;   // pi_j (make_valid)
  // sufficient statistics: shared (stride 0) or per chain (state space)
  const double *xty;  // [chain * xty_stride + j]
  int64_t xty_stride;
  const double *yty;  // [chain * suf_stride]
  const double *nobs; // [chain * suf_stride]
  int32_t suf_stride;

  double prior_df, prior_ss, sigma_max, swap_threshold;
  int64_t max_model_size;  // < 0: none
  int32_t max_flips;       // min(max_nflips_, p) already applied; 0 = no selection
  int32_t draw_beta, draw_sigma;

  // CorrelationMap as CSR (CorrelationMap.cpp:41-59); cm_start == nullptr when
  // the swap move is disabled (threshold >= 1)
  const int32_t *cm_start;
  const int32_t *cm_idx;
  const double *cm_cor;

  // per-chain state (HBM)
  uint8_t *gamma;     // chains x p
  double *beta;       // chains x p
  double *sigsq;      // chains
  uint16_t *perm;     // chains x p   (BregVsSampler::indx, persistent)
  uint64_t *rng_pos;  // chains       (position in the sampler's stream)
  int32_t *status;    // chains
  int32_t *failures;  // chains       (failure_count_)
  // Sweeps still owed to each chain.  A launch adds its nsweeps and runs the
  // total; a chain that outgrows the launch's model capacity stops at a sweep
  // boundary (state restored to the end of its last complete sweep), keeps its
  // remaining count here with status CHAIN_MODEL_TOO_LARGE, and is resumed by
  // the host with a larger-capacity kernel.
  int32_t *todo;      // chains
  // Catch-up launches of the state-space path (sweeps alternate with the Kalman
  // kernel, so owed sweeps are repaid one per launch): at most run_limit sweeps
  // per chain and launch (0 = no limit), ran[chain] = sweeps done by this launch.
  int32_t run_limit;
  int32_t *ran;       // chains, or nullptr
  int32_t *maxk;      // 1: largest model size seen (capacity adaptation)
  int32_t *trace_idx; // chains: next trace slot

  // RNG key
  uint32_t seed_lo, seed_hi, stream;

  // Per-chain copy of the current model's wave-uniform data (both Cholesky
  // factors, their reciprocal diagonals, w, b_g, g) in HBM, laid out by
  // ssvs_scalar_layout(); the proposal evaluation reads it through the scalar
  // cache (s_load) so that factor elements arrive as SGPR operands of the FMAs.
  // per-chain table of acceptance thresholds exp(logp(gamma ^ {j}) - logp(gamma))
  // for the current model
  double *table_lp;             // chains x p
  uint8_t *table_kind;          // chains x p
  int32_t *table_tag;           // chains: capacity the table was built with, 0 = stale
  // The model block itself survives a launch the same way: model_tag[chain] =
  // capacity it was published with (0: rebuild); model_keep as table_keep;
  // suf_changed: xty / yty moved since (state-space path) -- the factors are
  // reused, the right-hand side and everything after it recomputed.
  int32_t *model_tag;
  int32_t model_keep, suf_changed;
  int32_t table_keep;           // 0: ignore the tags (something other than sweeps happened)
  double *model_scratch;        // 2 slots x chains x model_scratch_stride doubles
  int64_t model_scratch_stride;
  // Tables and model blocks exist twice per chain (slot s of chain c at
  // [s * chains + c]): the slot a chain is not using keeps the model it just
  // left, one flip away.  Noise variables enter a model and leave it again, so
  // the flip that undoes the last accepted one -- or redoes it -- finds factors
  // and table ready.  The tags carry the slot in bits 8+.

  // summaries (per chain; reduced over chains by a second kernel)
  uint32_t *inc_count;  // chains x p
  double *beta_sum;     // chains x p
  double *beta_sumsq;   // chains x p
  double *acc;          // chains x ACC_COUNT
  // optional traces (nullptr = off): chains x trace_stride
  double *trace_sigsq, *trace_logp, *trace_k;
  int32_t trace_stride;
  // optional record of every sweep's draw (nullptr = off), same slots as the
  // traces: the included variables and their coefficients, rec_cap per slot
  uint16_t *rec_idx;   // chains x trace_stride x rec_cap
  double *rec_beta;    // chains x trace_stride x rec_cap
  int32_t rec_cap;     // >= the launch's model capacity

  // ---- AdaptiveSpikeSlabRegressionSampler (ssvs_adaptive_kernel.hip; mode 2):
  // per-chain birth / death rates and iteration counts, the sampler's options
  double *ada_birth, *ada_death;   // chains x p
  uint64_t *ada_iter;              // chains
  double ada_step, ada_target;     // step_size_, target_acceptance_rate_
  int32_t ada_max_flips;           // max_flips_ (100); 0 = no model selection
  // Launches that hand chains over to one another (engine.hip: pipelined sweeps).  A
  // queue is int32 [push count | pop count | ready[chains]]: a workgroup of the launch that
  // fills q_out appends its chain when it is done with it; a workgroup of the next launch
  // (q_in = that queue) takes the next ready chain instead of chain blockIdx.x.  nullptr:
  // chain blockIdx.x, nothing appended.
  int32_t *q_in, *q_out;
  int32_t *q_error;                // set to 1 by a workgroup that waited for a chain in vain
  // Look-ahead batches that overlap (engine.hip): the batch's rows of the draw record start
  // at trace_row0 (>= 0; -1: at trace_idx[chain]), and the chain's state at the batch's
  // start is saved by the workgroup that takes the chain over (snap_gamma != nullptr) --
  // what a rewind restores.
  int32_t trace_row0;
  uint8_t *snap_gamma;
  double *snap_beta, *snap_sigsq, *snap_bsum, *snap_bsumsq, *snap_acc;
  uint16_t *snap_perm;
  uint64_t *snap_pos;
  int32_t *snap_fail;
  uint32_t *snap_inc;
  int32_t adaptive;                // 1: the launch serves the adaptive sampler (ssvs_big_kernel's mode switch)
  double *ada_ws;                  // chains x 4 p (large-model kernel): cumulative birth / death rates, the rates at the sweep's start

  // ---- HBM-resident path (ssvs_big_kernel.hip): models of more than 64
  // variables.  Capacity big_kcap (a multiple of 64); per-chain model blocks laid
  // out by ssvs_scalar_layout(big_kcap), two slots like model_scratch; per-wave
  // parking space for the solution panels of the per-lane triangular solves.
  int32_t big_kcap;
  double *big_model;          // 2 slots x chains x big_model_stride doubles
  int64_t big_model_stride;
  double *big_xs;             // chains x 2 waves x big_kcap x 64 doubles
  // state-space rounds: the X'e GEMM's split-K planes are still to be added (the plane
  // sum folded into this launch -- one-wave launches over all chains only): xty of a
  // chain is written from xty_planes[z][chain * p + j], z < xty_nplanes, in plane order,
  // before anything reads it.  nullptr: xty is there.
  const double *xty_planes;
  int32_t xty_nplanes;
  int64_t xty_plane_stride;
};

// ---- LDS layout of one chain (one wavefront) --------------------------------
// The two Cholesky factors are stored "block packed": 8 x 8 blocks, lower
// block-triangle only, block (I, J) at index I(I+1)/2 + J, each block 64
// doubles row-major (512 B, 16-byte aligned rows) so that a proposal's
// per-lane triangular solve streams whole blocks with ds_read_b128 and keeps
// its solution vector in registers.  kcap is a multiple of 8.
// doubles first, then 16-bit, then bytes; all offsets in bytes.
struct SsvsLds {
  uint32_t Lv, La, rdv, rda, w, bg, ctrl, park, g, perm0, perm1, oth, last, pred, gam, gam0, nbr, total;
};

static inline __host__ __device__ SsvsLds ssvs_lds_layout(int p, int kcap) {
  SsvsLds L;
  const uint32_t nb = (uint32_t)kcap / 8;
  const uint32_t fac = nb * (nb + 1) / 2 * 64 * 8;
  const uint32_t pv = (((uint32_t)p * 2) + 15u) & ~15u;
  uint32_t o = 0;
  L.Lv = o;    o += fac;
  L.La = o;    o += fac;
  L.rdv = o;   o += (uint32_t)kcap * 8;
  L.rda = o;   o += (uint32_t)kcap * 8;
  L.w = o;     o += (uint32_t)kcap * 8;
  L.bg = o;    o += (uint32_t)kcap * 8;
  L.ctrl = o;  o += 512;  // control block shared by a chain's wavefronts
  // per-lane state of the master parked in LDS rather than in registers:
  // beta across a forked sweep, coefficient sums, sums of squares, counts | variable
  L.park = o;  o += 4 * 512;
  L.g = o;     o += (((uint32_t)kcap * 2) + 15u) & ~15u;
  L.perm0 = o; o += pv;
  L.perm1 = o; o += pv;
  L.oth = o;   o += pv;
  L.last = o;  o += 2 * pv;  // 32-bit entries (LDS exchange)
  L.pred = o;  o += pv;
  L.gam = o;   o += ((uint32_t)p + 15u) & ~15u;
  L.gam0 = o;  o += ((uint32_t)p + 15u) & ~15u;  // gamma at the start of the sweep
  L.nbr = o;   o += ((uint32_t)p + 15u) & ~15u;  // 1: the variable has partners in the correlation map
  L.total = o;
  return L;
}

// ---- LDS layout of one chain in the HBM-resident kernel -------------------------
// No factors here (they live in the chain's HBM block): one 64 x 64 diagonal
// tile while it is being factored, the k-vectors the tail needs, the
// permutation work arrays and gamma.
struct SsvsBigLds {
  uint32_t tile, rdt, w, y, rd, bst, ctrl, g, gst, perm0, perm1, oth, last, pred, gam, gam0, nbr, total;
};
static inline __host__ __device__ SsvsBigLds ssvs_big_lds_layout(int p, int kcap) {
  SsvsBigLds L;
  const uint32_t pv = (((uint32_t)p * 2) + 15u) & ~15u;
  uint32_t o = 0;
  L.tile = o;  o += 36 * 64 * 8;            // lower block-triangle of 8 x 8 blocks
  L.rdt = o;   o += 64 * 8;
  L.w = o;     o += (uint32_t)kcap * 8;
  L.y = o;     o += (uint32_t)kcap * 8;
  L.rd = o;    o += (uint32_t)kcap * 8;
  L.bst = o;   o += (uint32_t)kcap * 8;     // coefficients of the last complete draw
  L.ctrl = o;  o += 512;
  L.g = o;     o += (((uint32_t)kcap * 2) + 15u) & ~15u;
  L.gst = o;   o += (((uint32_t)kcap * 2) + 15u) & ~15u;  // ... and their variables
  L.perm0 = o; o += pv;
  L.perm1 = o; o += pv;
  L.oth = o;   o += pv;
  L.last = o;  o += 2 * pv;
  L.pred = o;  o += pv;
  L.gam = o;   o += ((uint32_t)p + 15u) & ~15u;
  L.gam0 = o;  o += ((uint32_t)p + 15u) & ~15u;
  L.nbr = o;   o += ((uint32_t)p + 15u) & ~15u;
  L.total = o;
  return L;
}

// ---- LDS layout of one chain in the adaptive (birth / death) kernel -------------
struct SsvsAdaLds {
  uint32_t Lv, La, rdv, rda, w, bg, ctrl, birth, death, cumb, cumd, undo_v, undo_j, g, gam, gam0, total;
};
enum { ADA_UNDO_CAP = 128 };
static inline __host__ __device__ SsvsAdaLds ssvs_ada_lds_layout(int p, int kcap) {
  SsvsAdaLds L;
  const uint32_t nb = (uint32_t)kcap / 8;
  const uint32_t fac = nb * (nb + 1) / 2 * 64 * 8;
  const uint32_t pd = (uint32_t)p * 8;
  uint32_t o = 0;
  L.Lv = o;     o += fac;
  L.La = o;     o += fac;
  L.rdv = o;    o += (uint32_t)kcap * 8;
  L.rda = o;    o += (uint32_t)kcap * 8;
  L.w = o;      o += (uint32_t)kcap * 8;
  L.bg = o;     o += (uint32_t)kcap * 8;
  L.ctrl = o;   o += 512;
  L.birth = o;  o += pd;
  L.death = o;  o += pd;
  L.cumb = o;   o += pd;      // inclusive prefix sums of the excluded variables' birth rates
  L.cumd = o;   o += pd;      // ... of the included variables' death rates
  L.undo_v = o; o += ADA_UNDO_CAP * 8;   // rates changed by the sweep in progress (old values)
  L.undo_j = o; o += ADA_UNDO_CAP * 4;
  L.g = o;      o += (((uint32_t)kcap * 2) + 15u) & ~15u;
  L.gam = o;    o += ((uint32_t)p + 15u) & ~15u;
  L.gam0 = o;   o += ((uint32_t)p + 15u) & ~15u;
  L.total = o;
  return L;
}

// offsets (in doubles) inside one chain's model_scratch block
struct SsvsScalarLayout {
  uint32_t Lv, La, rdv, rda, w, bg, g, scal, iv, ia, total;
};
static inline __host__ __device__ SsvsScalarLayout ssvs_scalar_layout(int kcap) {
  SsvsScalarLayout S;
  const uint32_t nb = (uint32_t)kcap / 8;
  const uint32_t fac = nb * (nb + 1) / 2 * 64;
  uint32_t o = 0;
  S.Lv = o;  o += fac;
  S.La = o;  o += fac;
  S.rdv = o; o += (uint32_t)kcap;
  S.rda = o; o += (uint32_t)kcap;
  S.w = o;   o += (uint32_t)kcap;
  S.bg = o;  o += (uint32_t)kcap;
  S.g = o;   o += (uint32_t)kcap / 2;  // int32 indices, two per double
  S.scal = o; o += 8;                  // logp, lp, ldv, lda, Q, c, SS, pd (the model's scalars)
  // the inverses of the factors' 16 x 16 diagonal blocks (ssvs_fill_mfma.h), row-major
  S.iv = o;  o += (uint32_t)kcap * 16;
  S.ia = o;  o += (uint32_t)kcap * 16;
  S.total = (o + 7u) & ~7u;            // whole 64-byte lines per chain
  return S;
}

}  // namespace boom_amd
