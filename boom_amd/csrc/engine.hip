// Host side of the C-ABI (include/boom_amd.h): owns the device buffers of one
// engine, assembles priors the way BregVsSampler's constructors do, launches
// the kernels and turns per-chain status words back into the reference's
// error messages.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/boom_amd.h"
#include "kalman_params.h"
#include "ktimer.h"
#include "probit_params.h"
#include "ssvs_params.h"

namespace boom_amd {
// ssvs_kernel.hip
hipError_t launch_ssvs_sweep(hipStream_t stream, const SsvsParams &P, int nsweeps);
// ssvs_big_kernel.hip
hipError_t launch_ssvs_big(hipStream_t stream, const SsvsParams &P, int nsweeps);
// ssvs_adaptive_kernel.hip
hipError_t launch_ssvs_adaptive(hipStream_t stream, const SsvsParams &P, int nsweeps);
hipError_t launch_ssvs_logp(hipStream_t stream, const SsvsParams &P,
                            const uint8_t *gammas, int ngamma, double *out,
                            int *status_out);
hipError_t launch_lds_exchange_order(hipStream_t stream, int *bad_device);
hipError_t launch_ssvs_reduce_summaries(hipStream_t stream, const SsvsParams &P,
                                        double *out);
// suf_kernel.hip
int suf_row_slices(int64_t n, int p);
int launch_suf_from_xy(hipStream_t stream, int64_t n, int p, const double *X,
                       const double *y, double *xtx, double *xty,
                       double *scalars /* yty, sumy */, double *xsum,
                       double *planes /* suf_row_slices(n, p) * p * p doubles, or null */);
// kalman_kernel.hip
hipError_t launch_ssm_simsmooth(hipStream_t stream, const SsParams &P, int draw_variances);
hipError_t launch_ssm_forecast(hipStream_t stream, const SsParams &P, int horizon, const double *newX,
                               uint64_t *pos_forecast, double *out);
hipError_t launch_probit_impute(hipStream_t stream, const ProbitParams &P, double *planes);
hipError_t launch_logit_impute(hipStream_t stream, const ProbitParams &P, const double *Xsq,
                               const double *slab_precision, double *v_diag, double *planes,
                               int polya_gamma);
// xtwx_cols_kernel.hip
hipError_t launch_ssvs_big_logp(hipStream_t stream, const SsvsParams &P, int kcap, const uint8_t *gammas,
                                const int *which, int nwhich, double *model_ws, double *xs_ws, double *out,
                                int *status_out);
hipError_t launch_predict(hipStream_t stream, const double *trace_k, const uint16_t *rec_idx,
                          const double *rec_beta, int stride, int cap, int first_draw, int ndraws,
                          int chains, int p, const double *newX, int nnew, double *out);
int xtwx_cols_planes(int64_t n);
int xte_planes(int64_t n);
hipError_t launch_xtwx_cols(hipStream_t stream, const double *X, int64_t n, int p, const double *w,
                            const int32_t *req, int R, const double *base, double *V,
                            uint32_t *valid, int words, double *planes);
hipError_t launch_xtwx_cols_start(hipStream_t stream, const uint8_t *gamma, int chains, int p,
                                  int32_t *req, int32_t *count, uint32_t *valid, int words);
hipError_t launch_square(hipStream_t stream, const double *x, size_t count, double *out);
hipError_t launch_kalman_simsmooth(hipStream_t stream, const SsParams &P,
                                   int draw_level);
hipError_t launch_kalman_main(hipStream_t stream, const SsParams &P, int draw_level);
hipError_t launch_kalman_xte(hipStream_t stream, const SsParams &P, bool planes_only);
hipError_t launch_kalman_prepare(hipStream_t stream, const SsParams &P, int draw_level);
hipError_t launch_ss_round(hipStream_t stream, const SsvsParams &P, const SsParams &S, const SsRoundParams &F,
                           int *max_resident);
size_t ss_round_lds(int p, int kcap);
hipError_t launch_ss_forecast(hipStream_t stream, const SsParams &P, int horizon, const double *newX,
                              uint64_t *pos_forecast, double *out);
}  // namespace boom_amd

using namespace boom_amd;

namespace {

thread_local std::string g_error = "";

int fail(int code, const std::string &msg) {
  g_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                       \
  do {                                                                      \
    hipError_t err__ = (expr);                                              \
    if (err__ != hipSuccess) {                                              \
      return fail(BA_E_HIP, std::string(#expr) + ": " +                     \
                                hipGetErrorString(err__));                  \
    }                                                                       \
  } while (0)

template <class T>
struct DevBuf {
  T *ptr = nullptr;
  size_t count = 0;
  ~DevBuf() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
  hipError_t resize(size_t n) {
    if (n == count && ptr) return hipSuccess;
    release();
    if (n == 0) return hipSuccess;
    hipError_t e = hipMalloc((void **)&ptr, n * sizeof(T));
    if (e == hipSuccess) count = n;
    return e;
  }
};

const char *status_message(int st) {
  switch (st) {
    case CHAIN_NOT_PD:
      return "The posterior information matrix is not positive definite.  "
             "Check your data or consider adjusting your prior.";
    case CHAIN_NEGATIVE_SS:
      return "Illegal data caused negative sum of squares in "
             "Breg::set_reg_post_params.";
    case CHAIN_ILLEGAL_START:
      return "BregVsSampler did not start with a legal configuration.";
    case CHAIN_RNG_BRANCH:
      return "Variance draw failed: lower bound must be to the right of the mode "
             "of logf in BoundedAdaptiveRejectionSampler, or a rejection sampler "
             "exceeded its number of attempts.";
    case CHAIN_FORECAST_VARIANCE:
      return "Found a zero (or negative) forecast variance!";
    case CHAIN_MODEL_TOO_LARGE:
      return "A chain's model size exceeded the engine's working capacity "
             "(a pinned max_model_size_hint, or more than 1024 variables in "
             "the model).";
    default:
      return "unknown chain status";
  }
}

int status_code(int st) {
  switch (st) {
    case CHAIN_NOT_PD: return BA_E_NOT_PD;
    case CHAIN_NEGATIVE_SS: return BA_E_NEGATIVE_SS;
    case CHAIN_ILLEGAL_START: return BA_E_ILLEGAL_START;
    case CHAIN_RNG_BRANCH: return BA_E_RNG_BRANCH;
    case CHAIN_FORECAST_VARIANCE: return BA_E_FORECAST_VARIANCE;
    case CHAIN_MODEL_TOO_LARGE: return BA_E_MODEL_TOO_LARGE;
    default: return BA_E_INVALID;
  }
}

}  // namespace

// ba_set_kernel_timing: the event pairs of the launches since the last read
struct KtSpan { int cls; hipEvent_t a, b; };
struct KTimer {
  std::vector<KtSpan> spans;
  std::vector<hipEvent_t> open;   // begin events by class (launches do not nest within a class)
  std::vector<hipEvent_t> pool;
  double ms[KT_CLASSES] = {};
  int64_t launches[KT_CLASSES] = {};
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  // fold the finished spans into the totals (the caller has synchronised the stream)
  void collect() {
    for (const KtSpan &sp : spans) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) { ms[sp.cls] += t; ++launches[sp.cls]; }
      pool.push_back(sp.a);
      pool.push_back(sp.b);
    }
    spans.clear();
  }
  ~KTimer() {
    for (const KtSpan &sp : spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (hipEvent_t e : open) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : pool) (void)hipEventDestroy(e);
  }
};
namespace {
thread_local KTimer *g_kt = nullptr;   // the timer of the engine whose entry point this thread is in
}
namespace boom_amd {
bool kt_active() { return g_kt != nullptr; }
void kt_mark(hipStream_t stream, int cls, bool begin) {
  KTimer *k = g_kt;
  if (!k || cls < 0 || cls >= KT_CLASSES) return;
  if (k->open.size() < (size_t)KT_CLASSES) k->open.resize(KT_CLASSES, nullptr);
  hipEvent_t ev = k->get();
  (void)hipEventRecord(ev, stream);
  if (begin) {
    if (k->open[cls]) k->pool.push_back(k->open[cls]);
    k->open[cls] = ev;
  } else if (k->open[cls]) {
    k->spans.push_back(KtSpan{cls, k->open[cls], ev});
    k->open[cls] = nullptr;
  } else {
    k->pool.push_back(ev);
  }
}
}  // namespace boom_amd

struct ba_engine {
  ba_config cfg{};
  bool kt_enabled = false;
  bool kt_overlap = false;   // (timing on, consecutive sweep launches still overlap: ba_set_kernel_timing(e, 2))
  KTimer kt;
  hipStream_t stream = nullptr;
  int p = 0;
  int cu_count = 256;
  size_t lds_per_cu = 160 * 1024;

  // ---- host copies (RegSuf + priors)
  bool have_suf = false, have_slab = false, have_spike = false,
       have_sigma = false;
  std::vector<double> xtx, xty, xsum;
  double yty = 0, n = 0, sumy = 0;
  std::vector<double> b, ominv, pi;
  int64_t max_model_size = -1;
  double prior_df = 0, prior_ss = 0, sigma_guess = 0;
  double sigma_max = std::numeric_limits<double>::infinity();
  int max_flips = -1;  // < 0: p
  double swap_threshold = 0.8;
  int draw_beta = 1, draw_sigma = 1;
  bool device_dirty = true;   // V/A/b/logpi/cm need (re)upload
  bool state_ready = false;

  // ---- device: shared
  DevBuf<double> dV, dA, db, dl1, dl0, dpi, dxty, dscal /* yty, n */;
  DevBuf<int32_t> dcm_start, dcm_idx;
  DevBuf<double> dcm_cor;
  bool cm_enabled = false;
  // ---- device: per chain
  DevBuf<uint8_t> dgamma;
  DevBuf<double> dbeta, dsigsq;
  DevBuf<uint16_t> dperm;
  DevBuf<uint64_t> dpos;
  DevBuf<int32_t> dstatus, dfail, dtodo, dmaxk, dtrace_idx;
  DevBuf<uint32_t> dinc;
  DevBuf<double> dbsum, dbsumsq, dacc, dsummary;
  DevBuf<double> dtr_sig, dtr_logp, dtr_k;
  DevBuf<uint16_t> drec_idx;  // recorded draws (ba_enable_draws)
  DevBuf<double> drec_beta;
  DevBuf<double> dmodel;  // per-chain model scratch (scalar-cache reads)
  DevBuf<double> dtab_lp;   // per-chain proposal table
  DevBuf<uint8_t> dtab_kind;
  DevBuf<int32_t> dtab_tag, dmodel_tag;
  DevBuf<int32_t> dran;  // catch-up launches of the state-space path: sweeps done per chain
  bool table_ok = false;  // nothing but ba_sweep launches since the tables were built
  bool model_ok = false;  // nothing that changes a model's factors since the last sweep launch
  int trace_stride = 0;
  // scratch for suf build
  DevBuf<double> dX, dy, dxtx, dxsum, dsufscal;

  int kcap = 0;
  int waves = 1;  // wavefronts per chain
  // ba_draw_next: the look-ahead batch.  la_avail draws are recorded on the
  // device, la_served of them have been handed out; the snapshot is the chains'
  // state (and the running summaries) at the start of the batch, which is what
  // a rewind restores before replaying the la_served draws already seen.
  int la_len = 1, la_avail = 0, la_served = 0;
  // host copies of the batch's record for the chains the caller reads (the
  // per-iteration loop reads chain 0 after every draw: one set of copies per
  // batch instead of per call); la_synced: the batch's launch has been waited for
  // and its chain statuses checked
  struct LaRows { std::vector<double> k, sig, beta; std::vector<uint16_t> idx; };
  std::unordered_map<int64_t, LaRows> la_cache;
  bool la_synced = false;
  // Overlapping look-ahead batches: the record holds two batches (halves la_slot and
  // la_slot ^ 1 of 2 la_len rows), the batch after the one being served is launched as soon
  // as serving starts (la_ahead) and the launches hand chains over (pipelined sweeps); each
  // batch's workgroups save their chain's state on entry (snapshot set = half) for rewinds.
  bool la_pipe = true;        // allowed (off for good after a batch had to be redone the old way)
  bool la_cur_piped = false;  // the batch being served was launched that way
  bool la_ahead = false;
  int la_slot = 0;
  hipEvent_t la_done[2] = {nullptr, nullptr};
  DevBuf<uint8_t> snap_gamma;
  DevBuf<double> snap_beta, snap_sigsq, snap_bsum, snap_bsumsq, snap_acc;
  DevBuf<uint16_t> snap_perm;
  DevBuf<uint64_t> snap_pos;
  DevBuf<int32_t> snap_fail;
  DevBuf<uint32_t> snap_inc;
  int rec_cap = 64;  // variables per recorded draw (ba_enable_draws)
  // HBM-resident path for models of more than 64 variables (ssvs_big_kernel.hip):
  // active once a chain has outgrown the LDS kernel, capacity grows on demand
  bool big_active = false;
  int big_kcap = 0;
  DevBuf<double> dbig_model, dbig_xs;
  // ba_set_tuning overrides (0 / -1: the engine chooses)
  int tune_waves = 0, tune_walk_policy = -1, tune_kcap_start = 0;
  // SpikeSlabSampler (sigma^2 given) mode
  int cur_mode = 0;          // mode of the launches in flight (0 BregVs, 1 SSS)
  int sss_slab_scales = 1;   // slab precision = Omega^{-1} / sigma^2
  int sss_max_flips = -1;    // limits only when > 0 (SpikeSlabSampler.cpp:77)
  double v_scale = 1.0;      // V = Omega^{-1} + v_scale * XtX currently on the device
  double v_scale_want = 1.0;
  DevBuf<uint64_t> dpos_sss;
  uint64_t seed = 0;
  // AdaptiveSpikeSlabRegressionSampler (mode 2): rates, iteration counts, options
  DevBuf<double> dada_birth, dada_death, dada_ws;
  DevBuf<uint64_t> dada_iter, dpos_ada;
  int ada_max_flips = 100;
  double ada_step = .001, ada_target = .345;

  // ---- state space (bsts local level + regression)
  bool ss_mode = false, ss_level_set = false, ss_initialized = false;
  int T = 0;
  DevBuf<double> dss_y, dss_X, dss_scratch;
  // The callers' loop is "one round, then read chain 0": what the accessors copy goes
  // through ONE pinned staging buffer per call (a batch of asynchronous copies, one
  // synchronisation) instead of one blocking copy per field, and a ba_sync() that follows
  // a clean ba_sync() with no call in between that could have enqueued or changed anything
  // is free (api_seq counts such calls, clean_seq remembers the last clean check).
  void *pinned = nullptr;
  size_t pinned_bytes = 0;
  uint64_t api_seq = 1, clean_seq = 0;
  int api_depth = 0;   // calls other than accessors in progress (they may enqueue after an inner ba_sync)
  // lane-major copies for kalman_lm_kernel (kalman_params.h): local level, T <= LM_TP
  DevBuf<double> dss_yt, dss_Xt;
  DevBuf<uint32_t> dss_obs_mask;
  DevBuf<uint8_t> dss_obs;
  DevBuf<double> dxty_c, dyty_c, dnobs_c;       // per-chain regression suf
  DevBuf<double> dlev_sigsq, dlev_n, dlev_sumsq;
  DevBuf<uint64_t> dpos_level, dpos_state, dpos_forecast;
  // kalman_prepare_kernel (level variance + normals of the next state draw) runs on a
  // second stream beside the X'e GEMM and the SSVS launch
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_state = nullptr, ev_prep[2] = {nullptr, nullptr};
  int ss_zbuf = 0;   // the normals buffer the next state draw reads
  // pipelined sweeps: consecutive ba_sweep launches alternate between `stream` and
  // pipe_stream and hand chains over through a ring of four queues (ssvs_kernel.hip)
  hipStream_t pipe_stream = nullptr;
  hipEvent_t pipe_ev[4] = {}, pipe_join_ev = nullptr;
  DevBuf<int32_t> dpipe_q, dpipe_err;
  bool pipe_on = false;      // the last thing enqueued was a pipelined sweep launch
  bool pipe_unchecked = false;   // a pipelined launch has gone out since the error word was last read
  int pipe_k = 0;            // launches in the current pipeline
  bool pipe_groups = false;  // ... which are the chain GROUPS of an engine of more chains than the machine holds
  DevBuf<int32_t> dprep_n;
  DevBuf<uint64_t> dprep_pos_state, dprep_pos_level;
  DevBuf<double> dprep_level;
  DevBuf<double> dxte_planes;   // split-K planes of the X'e GEMM
  double level_prior_df = 0, level_prior_ss = 0;
  double level_sigma_max = std::numeric_limits<double>::infinity();
  double ss_a0 = 0, ss_P0 = 1, ss_initial_level_sigsq = 1;
  // BinomialProbitSpikeSlabSampler (probit_kernel.hip): data on the device, the
  // latent sums z (chains x n), imputations done so far
  bool probit_mode = false;
  int64_t probit_n = 0;
  int probit_clt = 5;
  uint64_t probit_sweep = 0;
  DevBuf<double> dprob_X, dprob_y, dprob_nt, dprob_z;
  // BinomialLogitSpikeSlabSampler: the same buffers plus the observations' total
  // precisions (chains x n) and every chain's own V = slab precision + X'WX
  bool logit_mode = false;
  int logit_imputer = 0;           // 0: the reference's auxiliary mixture, 1: Polya-Gamma
  // PoissonRegressionSpikeSlabSampler: the logit path's machinery (every chain's own V a
  // vector at a time) with its own imputation kernel and SpikeSlabSampler's shuffle;
  // dprob_nt holds the exposures; the reference table's mixtures by count
  bool poisson_mode = false;
  bool poisson_mix_set = false;
  std::vector<int64_t> poisson_y;             // host copy of the counts (to map them to mixtures)
  DevBuf<int32_t> dpois_off, dpois_obs;
  DevBuf<double> dpois_mu, dpois_sigma, dpois_logw;
  int poisson_mix_one = -1;
  int slot_limit = 0;              // (ba_set_slot_limit)
  DevBuf<double> dlogit_w, dlogit_V;
  // ... V built a vector at a time (xtwx_cols_kernel.hip): the squared design matrix
  // (for the diagonal), the diagonals (chains x p), which vectors hold this sweep's
  // values (bits, logit_words words per chain), the request list (chain, variable) and
  // its length, the variable a parked chain waits for, the GEMM's split-K planes
  DevBuf<double> dlogit_Xsq, dlogit_vdiag, dlogit_planes;
  DevBuf<uint32_t> dlogit_valid;
  DevBuf<int32_t> dlogit_req, dlogit_cnt, dcol_request;
  int logit_words = 0;
  int64_t logit_req_batch = 0;     // requests per GEMM launch (bounds the planes)
  int64_t logit_cols_built = 0, logit_cols_requested = 0, logit_replays = 0;   // (diagnostics)
  // structural state (a list of state models, ssm_kernel.hip) instead of the local level
  bool ssm_set = false;
  SsgSpec ssg{};                   // the host's copy of the specification
  DevBuf<uint8_t> dssg_spec;       // ... and the device's
  double ssg_initial_sigsq[SSG_MAX_VAR] = {};
  double ssg_initial_phi[SSG_MAX_AR][AR_MAX] = {};
  // the template of ba_ss_set_structural (level / slope / seasonal -> variance index, -1: none)
  int ssg_template_var[3] = {-1, -1, -1};
  int ssg_template_ar = -1;        // ... and the block ba_ss_add_ar appended
  int ssg_kernel_choice = 1;       // (ba_ss_set_tuning: 0 general, 1 the default choice, 3 shape-specialised)
  // the local-level rounds of a call as one persistent launch (ss_round_kernel.hip); the
  // tile words of its X'e step, zeroed before every launch; how many chains' workgroups the
  // device holds at once, by launch capacity (0: not asked yet, < 0: the kernel does not fit)
  bool ss_round_enabled = true;    // (ba_ss_set_tuning 4 / 5: the separate launches of rounds 1-4 / this)
  DevBuf<int32_t> dround_ctl, dround_members, dround_reg;
  int ss_round_resident[4] = {0, 0, 0, 0};
  bool round_debug = false;        // (ba_ss_set_tuning 6 / 7: the round kernel's notes of a debugging session on / off)
  DevBuf<int32_t> dround_debug;    // ... on this engine's device
  DevBuf<double> dssm_sigsq, dssm_n, dssm_ss, dssm_work;   // chains x SSG_MAX_VAR (sigsq, n, ss)
  DevBuf<double> dar_phi, dar_suf;                         // chains x SSG_MAX_AR x (AR_MAX | AR_SUF_STRIDE)
  DevBuf<uint64_t> dpos_var;                               // chains x SSG_MAX_VAR
  // ---- look-ahead on the bsts path (ba_ss_set_lookahead / ba_ss_draw_next): a batch of
  // `len` sweep rounds per enqueue, every round's draw recorded on the device -- gamma,
  // beta, sigma^2, the state models' variances and coefficients for EVERY chain, the
  // state path for the registered chains -- and handed out one per call.  Two halves:
  // the batch after the one being served is enqueued as soon as serving starts.  Any
  // entry point that is not served from the record first puts the chains where the
  // caller has seen them (ss_la_settle: the snapshot of the batch's start, replayed up
  // to the draw being served), so the look-ahead is unobservable.
  struct SsLa {
    int len = 0;                // rounds per batch at most (<= 1: off)
    // rounds per batch NOW.  A caller whose loop reads something the record does not hold
    // (another chain's state path, sufficient statistics, a forecast) or changes something
    // (priors under the sampler) after every draw pays a rewind + replay of the batch each
    // time: so a settle halves the batch, a batch served to its end doubles it again (up to
    // len); at one round per call the look-ahead is off and is tried again after
    // `probe_wait` calm draws (16, doubling while the tries keep failing).
    int cur = 0, calm = 0, probe_wait = 16;
    bool clean = true;          // nothing has settled the batch being served
    int ahead_len = 0;          // rounds of the batch that is running ahead
    int avail = 0, served = 0, slot = 0;
    bool ahead = false;         // the next batch is enqueued (half slot ^ 1)
    bool synced = false;        // the batch being served is complete and its chains sound
    bool busy = false;          // (a settle in progress: entry points it calls do not settle again)
    hipEvent_t done[2] = {nullptr, nullptr};
    std::vector<int32_t> reg{0};            // chains whose state path is recorded (always the size of dreg / rstate's rows)
    std::vector<int32_t> want;              // ... and the ones asked for since: ss_la_alloc takes them into reg, as far as the record has room
    DevBuf<int32_t> dreg;
    DevBuf<double> lev_used;                // the level variance every chain's last state draw used
    size_t nvar = 0, nphi = 0, state_doubles = 0;   // per chain and round
    DevBuf<uint8_t> rgamma;                 // [slot][chain][round][p]
    DevBuf<double> rbeta, rsig, rvar, rphi; // [slot][chain][round][...]
    DevBuf<double> rstate;                  // [slot][registered chain][round][state_doubles]
    DevBuf<double> snap;                    // the state-space half of the chain state, two sets
    DevBuf<uint64_t> snap_pos;
    size_t snap_doubles = 0, snap_words = 0;
    struct Rows { std::vector<uint8_t> gamma; std::vector<double> beta, sig, var, phi, state; bool has_state = false; };
    std::unordered_map<int64_t, Rows> cache;
  } ssla;
};

namespace {

// CorrelationMap::fill, Models/Glm/PosteriorSamplers/CorrelationMap.cpp:41-59
void build_correlation_map(const ba_engine &e, std::vector<int32_t> &start,
                           std::vector<int32_t> &idx,
                           std::vector<double> &cor) {
  const int p = e.p;
  const double n = e.n;
  std::vector<double> xbar(p), sd(p);
  for (int i = 0; i < p; ++i) xbar[i] = e.xsum[i] / n;
  auto cov = [&](int i, int j) {
    return (e.xtx[(size_t)j * p + i] + (-n) * xbar[i] * xbar[j]) / (n - 1);
  };
  for (int i = 0; i < p; ++i) {
    sd[i] = std::sqrt(cov(i, i));
    if (!(sd[i] > 0.0)) sd[i] = 1.0;
  }
  start.assign(p + 1, 0);
  idx.clear();
  cor.clear();
  for (int i = 0; i < p; ++i) {
    start[i] = (int32_t)idx.size();
    for (int j = 0; j < p; ++j) {
      if (j == i) continue;
      const double c = std::fabs(cov(i, j) / (sd[i] * sd[j]));
      if (c >= e.swap_threshold) {
        idx.push_back(j);
        cor.push_back(c);
      }
    }
  }
  start[p] = (int32_t)idx.size();
}

// Working capacity of a chain's LDS set: 16, 32, 48 or 64 variables (the sweep
// kernel is instantiated per capacity; smaller is faster).  The capacity is
// ADAPTIVE: launches run with the current capacity, a chain that outgrows it
// stops at a sweep boundary and is resumed by ba_sync() with the next size
// (escalate()), and the capacity follows the largest model seen.  A
// max_model_size prior caps it; ba_config.max_model_size_hint pins it.
int lds_cap(const ba_engine &e) {   // largest capacity whose LDS set fits a CU
  int k = 64;
  while (k > 16 && ssvs_lds_layout(e.p, k).total > e.lds_per_cu) k -= 16;
  return k;
}
int cap_limit(const ba_engine &e) {
  int64_t need = std::min(64, e.p);
  if (e.max_model_size >= 0) need = std::min<int64_t>(need, std::max<int64_t>(1, e.max_model_size));
  const int k = (int)std::min<int64_t>(64, ((need + 15) / 16) * 16);
  return std::min(k, lds_cap(e));
}
int choose_kcap(const ba_engine &e) {
  const int limit = cap_limit(e);
  if (e.cfg.max_model_size_hint > 0) {
    const int k = (int)std::min<int64_t>(64, (((int64_t)e.cfg.max_model_size_hint + 15) / 16) * 16);
    return std::min(k, lds_cap(e));
  }
  int start = 32;  // (ba_set_tuning: tests force early escalations)
  if (e.tune_kcap_start > 0) start = std::max(16, (e.tune_kcap_start / 16) * 16);
  return std::min(start, limit);
}

// Wavefronts per chain.  The proposal batches scale with the number of waves
// (64 proposals each, evaluated speculatively), and several resident waves per
// SIMD hide the gather / scalar-load latencies; the register budget of the
// 4-wave kernels only exists for capacities <= 32.  ba_set_tuning overrides.
int choose_waves(const ba_engine &e, int kcap) {
  {
    const int w = e.tune_waves;
    if (w == 1 || w == 2 || (w == 4 && kcap <= 32)) return w;
  }
  // Two wavefronts per chain at every engine size: the helper wave is worth more than
  // the second resident chain it displaces (measured on the C2 workload, sweeps/s with
  // 1 / 2 wavefronts: 1024 chains 28 / 41 M, 2048 30 / 36 M, 4096 31 / 43 M, 8192
  // 32 / 46 M), so chains beyond 4 per CU simply run in further rounds.
  (void)kcap;
  // The state-space path alternates ONE sweep with the state draw: no table survives a round
  // and there is no quiet sweep to fork, the helper wave has little to do; one wavefront per
  // chain measured better at every size (T=2000, p=100, us per round with 1 / 2 wavefronts:
  // 512 chains 144 / 144, 1024 167 / 172, 2048 279 / 304, 4096 545 / 590).
  // (the local-level round only: with the structural kernel behind it the same choice makes
  // THAT kernel slower -- 5.72 vs 5.06 ms per round at m = 13 -- for reasons not understood)
  if (e.ss_mode && !e.ssm_set) return 1;
  return 2;
}

int upload_shared(ba_engine *e) {
  if (!e->device_dirty) return BA_OK;
  const int p = e->p;
  if (!e->have_suf) return fail(BA_E_STATE, "no regression data set");
  if (!e->have_slab || !e->have_spike || !e->have_sigma)
    return fail(BA_E_STATE, "priors (slab, spike, sigma) must be set before sampling");
  const size_t pp = (size_t)p * p;
  std::vector<double> V(pp), l1(p), l0(p), scal(2);
  for (size_t i = 0; i < pp; ++i) V[i] = e->ominv[i] + e->xtx[i] * e->v_scale_want;
  e->v_scale = e->v_scale_want;
  // VariableSelectionPrior::ensure_log_probabilities,
  // VariableSelectionPrior.cpp:310-317
  for (int j = 0; j < p; ++j) {
    l1[j] = std::log(e->pi[j]);
    l0[j] = std::log(1 - e->pi[j]);
  }
  scal[0] = e->yty;
  scal[1] = e->n;
  HIP_TRY(e->dV.resize(pp));
  HIP_TRY(e->dA.resize(pp));
  HIP_TRY(e->db.resize(p));
  HIP_TRY(e->dl1.resize(p));
  HIP_TRY(e->dl0.resize(p));
  HIP_TRY(e->dpi.resize(p));
  HIP_TRY(e->dxty.resize(p));
  HIP_TRY(e->dscal.resize(2));
  hipStream_t s = e->stream;
  HIP_TRY(hipMemcpyAsync(e->dV.ptr, V.data(), pp * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->dA.ptr, e->ominv.data(), pp * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->db.ptr, e->b.data(), p * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->dl1.ptr, l1.data(), p * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->dl0.ptr, l0.data(), p * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->dpi.ptr, e->pi.data(), p * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->dxty.ptr, e->xty.data(), p * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->dscal.ptr, scal.data(), 16, hipMemcpyHostToDevice, s));
  e->cm_enabled = e->swap_threshold < 1.0;
  if (e->cm_enabled) {
    std::vector<int32_t> start, idx;
    std::vector<double> cor;
    build_correlation_map(*e, start, idx, cor);
    HIP_TRY(e->dcm_start.resize(start.size()));
    HIP_TRY(e->dcm_idx.resize(std::max<size_t>(1, idx.size())));
    HIP_TRY(e->dcm_cor.resize(std::max<size_t>(1, cor.size())));
    HIP_TRY(hipMemcpyAsync(e->dcm_start.ptr, start.data(), start.size() * 4, hipMemcpyHostToDevice, s));
    if (!idx.empty()) {
      HIP_TRY(hipMemcpyAsync(e->dcm_idx.ptr, idx.data(), idx.size() * 4, hipMemcpyHostToDevice, s));
      HIP_TRY(hipMemcpyAsync(e->dcm_cor.ptr, cor.data(), cor.size() * 8, hipMemcpyHostToDevice, s));
    }
  }
  HIP_TRY(hipStreamSynchronize(s));
  e->kcap = choose_kcap(*e);
  e->waves = choose_waves(*e, e->kcap);
  e->device_dirty = false;
  e->table_ok = false;
  e->model_ok = false;
  return BA_OK;
}

int alloc_chain_state(ba_engine *e) {
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p;
  if (e->state_ready && e->dgamma.count == C * p) return BA_OK;
  HIP_TRY(e->dgamma.resize(C * p));
  HIP_TRY(e->dbeta.resize(C * p));
  HIP_TRY(e->dsigsq.resize(C));
  HIP_TRY(e->dperm.resize(C * p));
  HIP_TRY(e->dpos.resize(C));
  HIP_TRY(e->dpos_sss.resize(C));
  HIP_TRY(e->dstatus.resize(C));
  HIP_TRY(e->dfail.resize(C));
  HIP_TRY(e->dtodo.resize(C));
  HIP_TRY(e->dtab_lp.resize(2 * C * p));    // two slots per chain (ssvs_params.h)
  HIP_TRY(e->dtab_kind.resize(2 * C * p));
  HIP_TRY(e->dtab_tag.resize(C));
  HIP_TRY(e->dmodel_tag.resize(C));
  e->model_ok = false;
  HIP_TRY(e->dran.resize(C));
  e->table_ok = false;
  e->model_ok = false;
  HIP_TRY(e->dmaxk.resize(1));
  HIP_TRY(e->dtrace_idx.resize(C));
  HIP_TRY(e->dinc.resize(C * p));
  HIP_TRY(e->dbsum.resize(C * p));
  HIP_TRY(e->dbsumsq.resize(C * p));
  HIP_TRY(e->dacc.resize(C * ACC_COUNT));
  HIP_TRY(e->dsummary.resize(3 * p + SUMMARY_SCALARS));
  // defaults: gamma = 0, beta = 0, sigsq = 1 (RegressionModel(xdim)), perm =
  // identity (seq<uint>(0, p-1), BregVsSampler.cpp:186), stream position 0
  std::vector<uint16_t> perm(C * p);
  for (size_t c = 0; c < C; ++c)
    for (size_t j = 0; j < p; ++j) perm[c * p + j] = (uint16_t)j;
  std::vector<double> ones(C, 1.0);
  hipStream_t s = e->stream;
  HIP_TRY(hipMemsetAsync(e->dgamma.ptr, 0, C * p, s));
  HIP_TRY(hipMemsetAsync(e->dbeta.ptr, 0, C * p * 8, s));
  HIP_TRY(hipMemcpyAsync(e->dsigsq.ptr, ones.data(), C * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(e->dperm.ptr, perm.data(), C * p * 2, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(e->dpos.ptr, 0, C * 8, s));
  HIP_TRY(hipMemsetAsync(e->dpos_sss.ptr, 0, C * 8, s));
  HIP_TRY(hipMemsetAsync(e->dstatus.ptr, 0, C * 4, s));
  HIP_TRY(hipMemsetAsync(e->dfail.ptr, 0, C * 4, s));
  HIP_TRY(hipMemsetAsync(e->dtodo.ptr, 0, C * 4, s));
  HIP_TRY(hipMemsetAsync(e->dmaxk.ptr, 0, 4, s));
  HIP_TRY(hipMemsetAsync(e->dtrace_idx.ptr, 0, C * 4, s));
  HIP_TRY(hipStreamSynchronize(s));
  e->state_ready = true;
  return ba_reset_summaries(e);
}

void fill_params(ba_engine *e, SsvsParams &P) {
  std::memset(&P, 0, sizeof(P));
  P.p = e->p;
  P.chains = e->cfg.chains;
  P.chain_first = 0;
  P.chain_count = e->cfg.chains;
  P.chain_offset = e->cfg.chain_offset;
  P.kcap = e->kcap;
  P.waves = e->waves;
  P.V = e->dV.ptr;
  P.A = e->dA.ptr;
  P.b = e->db.ptr;
  P.l1 = e->dl1.ptr;
  P.l0 = e->dl0.ptr;
  P.pi = e->dpi.ptr;
  if (e->ss_mode) {
    P.xty = e->dxty_c.ptr;
    P.xty_stride = e->p;
    P.yty = e->dyty_c.ptr;
    P.nobs = e->dnobs_c.ptr;
    P.suf_stride = 1;
  } else if ((e->probit_mode || e->logit_mode) && e->dxty_c.count) {
    P.xty = e->dxty_c.ptr;      // X'z of every chain's own imputation
    P.xty_stride = e->p;
    P.yty = e->dscal.ptr;
    P.nobs = e->dscal.ptr + 1;
    P.suf_stride = 0;
  } else {
    P.xty = e->dxty.ptr;
    P.xty_stride = 0;
    P.yty = e->dscal.ptr;
    P.nobs = e->dscal.ptr + 1;
    P.suf_stride = 0;
  }
  P.prior_df = e->prior_df;
  P.prior_ss = e->prior_ss;
  P.sigma_max = e->sigma_max;
  P.swap_threshold = e->swap_threshold;
  P.max_model_size = e->max_model_size;
  const int mf = e->max_flips < 0 ? e->p : e->max_flips;
  P.max_flips = std::min(mf, e->p);
  P.draw_beta = e->draw_beta;
  P.draw_sigma = e->draw_sigma;
  P.cm_start = e->cm_enabled ? e->dcm_start.ptr : nullptr;
  P.cm_idx = e->dcm_idx.ptr;
  P.cm_cor = e->dcm_cor.ptr;
  P.gamma = e->dgamma.ptr;
  P.beta = e->dbeta.ptr;
  P.sigsq = e->dsigsq.ptr;
  P.perm = e->dperm.ptr;
  P.rng_pos = e->dpos.ptr;
  P.status = e->dstatus.ptr;
  P.failures = e->dfail.ptr;
  P.todo = e->dtodo.ptr;
  P.maxk = e->dmaxk.ptr;
  P.trace_idx = e->dtrace_idx.ptr;
  P.model_scratch = e->dmodel.ptr;
  P.table_lp = e->dtab_lp.ptr;
  P.table_kind = e->dtab_kind.ptr;
  P.table_tag = e->dtab_tag.ptr;
  P.table_keep = e->table_ok ? 1 : 0;
  P.model_tag = e->dmodel_tag.ptr;
  P.model_keep = e->model_ok ? 1 : 0;
  P.suf_changed = (e->ss_mode || e->probit_mode || e->logit_mode) ? 1 : 0;
  P.run_limit = 0;
  P.ran = nullptr;
  P.model_scratch_stride = (int64_t)ssvs_scalar_layout(64).total;
  P.big_kcap = e->big_kcap;
  P.big_model = e->dbig_model.ptr;
  P.big_model_stride = e->big_kcap > 0 ? (int64_t)ssvs_scalar_layout(e->big_kcap).total : 0;
  P.big_xs = e->dbig_xs.ptr;
  P.seed_lo = (uint32_t)e->seed;
  P.seed_hi = (uint32_t)(e->seed >> 32);
  P.stream = 0;
  P.mode = e->cur_mode;
  P.walk_policy = e->tune_walk_policy >= 0 ? e->tune_walk_policy : 1;  // (see ssvs_params.h)
  if (e->cur_mode == 1) {
    // SpikeSlabSampler: given sigma^2, no sigma draw, no swap move, own stream
    P.slab_scales = e->sss_slab_scales;
    P.stream = 3;
    P.rng_pos = e->dpos_sss.ptr;
    P.draw_sigma = 0;
    P.draw_beta = 1;
    P.cm_start = nullptr;
    P.max_flips = (e->sss_max_flips > 0) ? std::min(e->sss_max_flips, e->p) : e->p;
  }
  if (e->cur_mode == 1 && e->logit_mode && e->dlogit_V.count) {
    // BinomialLogitSpikeSlabSampler: the sampler's own shuffle, every chain's own V
    // (which moves with the latent data: factors and tables are rebuilt)
    P.mode = e->poisson_mode ? 1 : 2;   // (the Poisson sampler drives the plain SpikeSlabSampler)
    P.V = e->dlogit_V.ptr;
    P.v_chain_stride = (int64_t)e->p * e->p;
    P.model_keep = 0;
    P.table_keep = 0;
    P.col_valid = e->dlogit_valid.ptr;
    P.col_words = e->logit_words;
    P.col_request = e->dcol_request.ptr;
    P.v_diag = e->dlogit_vdiag.ptr;
  }
  if (e->cur_mode == 2) {
    // AdaptiveSpikeSlabRegressionSampler: own stream, no swap move
    P.mode = 0;
    P.stream = 4;
    P.rng_pos = e->dpos_ada.ptr;
    P.cm_start = nullptr;
  }
  P.ada_birth = e->dada_birth.ptr;
  P.ada_death = e->dada_death.ptr;
  P.ada_iter = e->dada_iter.ptr;
  P.ada_step = e->ada_step;
  P.ada_target = e->ada_target;
  P.ada_max_flips = e->ada_max_flips;
  P.q_in = P.q_out = nullptr;
  P.q_error = nullptr;
  P.trace_row0 = -1;
  P.snap_gamma = nullptr;
  P.adaptive = (e->cur_mode == 2) ? 1 : 0;
  P.ada_ws = e->dada_ws.ptr;
  P.inc_count = e->dinc.ptr;
  P.beta_sum = e->dbsum.ptr;
  P.beta_sumsq = e->dbsumsq.ptr;
  P.acc = e->dacc.ptr;
  P.trace_sigsq = e->dtr_sig.ptr;
  P.trace_logp = e->dtr_logp.ptr;
  P.trace_k = e->dtr_k.ptr;
  P.trace_stride = e->trace_stride;
  P.rec_idx = e->drec_idx.ptr;
  P.rec_beta = e->drec_beta.ptr;
  P.rec_cap = e->rec_cap;
}

// ---- models of more than 64 variables: the HBM-resident kernel -------------------
enum { BIG_KCAP_MAX = 1024 };
// largest capacity that can ever be needed (0: the LDS kernel covers everything)
int big_limit(const ba_engine &e) {
  int64_t need = e.p;
  if (e.max_model_size >= 0) need = std::min<int64_t>(need, e.max_model_size);
  if (need <= 64) return 0;
  return (int)std::min<int64_t>(BIG_KCAP_MAX, ((need + 63) / 64) * 64);
}
int ensure_big_buffers(ba_engine *e) {
  const size_t C = (size_t)e->cfg.chains, kc = (size_t)e->big_kcap;
  const size_t want_model = 2 * C * ssvs_scalar_layout(e->big_kcap).total;
  const size_t want_xs = C * 2 * kc * 64;
  if (e->dbig_model.count != want_model) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(e->dbig_model.resize(want_model));
    HIP_TRY(e->dbig_xs.resize(want_xs));
  }
  const SsvsBigLds lay = ssvs_big_lds_layout(e->p, e->big_kcap);
  if (lay.total > e->lds_per_cu)
    return fail(BA_E_MODEL_TOO_LARGE, "the model does not fit the large-model kernel's LDS working set");
  // the draw record holds rec_cap variables per draw: widen it (keeping what is recorded)
  if (e->drec_idx.count > 0 && e->rec_cap < e->big_kcap) {
    const size_t rows = C * (size_t)e->trace_stride, oc = (size_t)e->rec_cap;
    DevBuf<uint16_t> ni;
    DevBuf<double> nb;
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(ni.resize(rows * kc));
    HIP_TRY(nb.resize(rows * kc));
    HIP_TRY(hipMemcpy2DAsync(ni.ptr, kc * 2, e->drec_idx.ptr, oc * 2, oc * 2, rows, hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipMemcpy2DAsync(nb.ptr, kc * 8, e->drec_beta.ptr, oc * 8, oc * 8, rows, hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    std::swap(ni.ptr, e->drec_idx.ptr);
    std::swap(ni.count, e->drec_idx.count);
    std::swap(nb.ptr, e->drec_beta.ptr);
    std::swap(nb.count, e->drec_beta.count);
    e->rec_cap = e->big_kcap;
  }
  return BA_OK;
}
// one launch of the sweep: the LDS kernel for every chain it can hold, and --
// once any chain has outgrown it -- the HBM-resident kernel right behind it for
// the chains the first one parked (status CHAIN_MODEL_TOO_LARGE)
hipError_t launch_sweeps(ba_engine *e, const SsvsParams &P, int nsweeps) {
  hipError_t err = (e->cur_mode == 2) ? launch_ssvs_adaptive(e->stream, P, nsweeps)
                                      : launch_ssvs_sweep(e->stream, P, nsweeps);
  if (err == hipSuccess && e->big_active) err = launch_ssvs_big(e->stream, P, 0);
  return err;
}
// the chains parked by the last launches need (more) large-model capacity;
// returns 1 when nothing more can be done (the status stays an error)
int grow_big(ba_engine *e, int *stuck) {
  *stuck = 0;
  const int bl = big_limit(*e);
  if (bl == 0) { *stuck = 1; return BA_OK; }
  if (!e->big_active) {
    e->big_active = true;
    if (e->big_kcap == 0) e->big_kcap = std::min(bl, 128);
  } else {
    if (e->big_kcap >= bl) { *stuck = 1; return BA_OK; }
    e->big_kcap = std::min(bl, e->big_kcap * 2);
  }
  return ensure_big_buffers(e);
}

// the vectors of V named by dlogit_req[0, R), in batches the planes can hold
int build_columns(ba_engine *e, int64_t R) {
  const int64_t n = e->probit_n;
  for (int64_t r0 = 0; r0 < R; r0 += e->logit_req_batch) {
    const int64_t nr = std::min<int64_t>(e->logit_req_batch, R - r0);
    HIP_TRY(launch_xtwx_cols(e->stream, e->dprob_X.ptr, n, e->p, e->dlogit_w.ptr,
                             e->dlogit_req.ptr + 2 * r0, (int)nr, e->dA.ptr, e->dlogit_V.ptr,
                             e->dlogit_valid.ptr, e->logit_words, e->dlogit_planes.ptr));
  }
  return BA_OK;
}

// Chains of the logit sampler parked at "add variable j" for want of vector j of
// their V (CHAIN_NEED_COLUMN): the vectors are computed -- one GEMM for all of
// them -- and the chains replay the sweep they were in, from its start and with the
// same draws, now finding the vector.  *served: something was replayed (st is fresh).
int serve_columns(ba_engine *e, std::vector<int32_t> &st, bool *served) {
  *served = false;
  if (!e->logit_mode || !e->dcol_request.count) return BA_OK;
  const size_t C = (size_t)e->cfg.chains;
  bool any = false;
  for (size_t c = 0; c < C; ++c) any = any || st[c] == CHAIN_NEED_COLUMN || st[c] == CHAIN_NEED_COLUMN_BIG;
  if (!any) return BA_OK;
  std::vector<int32_t> want(C), req;
  HIP_TRY(hipMemcpy(want.data(), e->dcol_request.ptr, C * 4, hipMemcpyDeviceToHost));
  for (size_t c = 0; c < C; ++c) {
    if (st[c] != CHAIN_NEED_COLUMN && st[c] != CHAIN_NEED_COLUMN_BIG) continue;
    if (want[c] < 0 || want[c] >= e->p) return fail(BA_E_STATE, "a parked chain names no variable");
    req.push_back((int32_t)c);
    req.push_back(want[c]);
    // (the large-model kernel takes its chains back in the parked state)
    st[c] = (st[c] == CHAIN_NEED_COLUMN) ? CHAIN_OK : CHAIN_MODEL_TOO_LARGE;
  }
  const int64_t R = (int64_t)req.size() / 2;
  HIP_TRY(hipMemcpyAsync(e->dlogit_req.ptr, req.data(), req.size() * 4, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->dstatus.ptr, st.data(), C * 4, hipMemcpyHostToDevice, e->stream));
  int rc = build_columns(e, R);
  if (rc) return rc;
  e->logit_cols_requested += R;
  ++e->logit_replays;
  SsvsParams P;
  fill_params(e, P);
  HIP_TRY(launch_sweeps(e, P, 0));   // the sweep still owed
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(st.data(), e->dstatus.ptr, C * 4, hipMemcpyDeviceToHost));
  *served = true;
  return BA_OK;
}

// Resume chains that outgrew the capacity of the launch they were in, with the
// next larger capacity; then follow the largest model size seen.
int escalate(ba_engine *e, std::vector<int32_t> &st) {
  const size_t C = (size_t)e->cfg.chains;
  for (;;) {
    {
      bool served = false;
      int rc = serve_columns(e, st, &served);
      if (rc) return rc;
      if (served) continue;
    }
    bool any = false;
    for (size_t c = 0; c < C; ++c) any = any || (st[c] == CHAIN_MODEL_TOO_LARGE);
    if (!any) return BA_OK;
    if (e->cfg.max_model_size_hint > 0) return BA_OK;  // stays an error
    if (e->kcap >= cap_limit(*e)) {
      // beyond the LDS kernel: the parked chains go to the HBM-resident one
      int stuck = 0;
      int rc = grow_big(e, &stuck);
      if (rc) return rc;
      if (stuck) return BA_OK;
      SsvsParams P;
      fill_params(e, P);
      HIP_TRY(launch_ssvs_big(e->stream, P, 0));
      HIP_TRY(hipStreamSynchronize(e->stream));
      HIP_TRY(hipMemcpy(st.data(), e->dstatus.ptr, C * 4, hipMemcpyDeviceToHost));
      continue;
    }
    e->kcap += 16;
    e->waves = choose_waves(*e, e->kcap);
    for (size_t c = 0; c < C; ++c)
      if (st[c] == CHAIN_MODEL_TOO_LARGE) st[c] = CHAIN_OK;
    HIP_TRY(hipMemcpyAsync(e->dstatus.ptr, st.data(), C * 4, hipMemcpyHostToDevice, e->stream));
    SsvsParams P;
    fill_params(e, P);
    HIP_TRY(launch_sweeps(e, P, 0));   // runs the sweeps still owed
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(st.data(), e->dstatus.ptr, C * 4, hipMemcpyDeviceToHost));
  }
}

// per chain: K (m T) | state (m T) | smoothed disturbances (nvar T) | normals (<= (nvar + 1) T + m + 1)
int64_t ssm_work_stride(const ba_engine &e) {
  // (the template kernel keeps four disturbance series and up to five normals a step)
  // (general kernel: a smoothed-disturbance series per state-error row, and as many normals a step + 1)
  const int64_t per_step = std::max(2 * e.ssg.m + 2 * std::max(e.ssg.nvar, e.ssg.nerr) + 1, 2 * e.ssg.m + 9);
  return per_step * e.T + SSG_MAX_STATE + 72;
}

// does the block list have the shape the template kernel is compiled for?
// [local level | local linear trend] [seasonal, duration 1] [autoregression], m <= 16
void ssg_template_shape(const SsgSpec &q, int32_t *trend, int32_t *nseasons, int32_t *ar_lags) {
  *trend = *nseasons = *ar_lags = 0;
  if (q.nblocks < 1 || q.nblocks > 3 || q.m > 16) return;
  for (int i = 0; i < q.nblocks; ++i)
    if (q.blk[i].nvar == 0) return;   // (a static intercept: the general kernel)
  int b = 0, tr = 0, ns = 0, lags = 0;
  if (q.blk[0].kind == SSG_LOCAL_LEVEL) tr = 1;
  else if (q.blk[0].kind == SSG_LOCAL_LINEAR_TREND) tr = 2;
  else return;
  b = 1;
  if (b < q.nblocks && q.blk[b].kind == SSG_SEASONAL) {
    if (q.blk[b].duration != 1) return;
    ns = q.blk[b].nseasons;
    ++b;
  }
  if (b < q.nblocks && q.blk[b].kind == SSG_AR) {
    lags = q.blk[b].lags;
    ++b;
  }
  if (b != q.nblocks) return;
  *trend = tr;
  *nseasons = ns;
  *ar_lags = lags;
}

// the local-level path of a series of at most LM_TP steps runs lane-major
// (kalman_lm_kernel): its scratch arrays have pitch LM_TP
static bool ss_lane_major(const ba_engine &e) { return !e.ssm_set && e.T <= LM_TP; }
static size_t ss_pitch(const ba_engine &e) { return ss_lane_major(e) ? (size_t)LM_TP : (size_t)e.T; }

void fill_ss_params(ba_engine *e, SsParams &S) {
  std::memset(&S, 0, sizeof(S));  // (only_ran = nullptr: every chain)
  S.T = e->T;
  S.slot_limit = e->slot_limit;
  S.p = e->p;
  S.chains = e->cfg.chains;
  S.chain_first = 0;
  S.chain_count = e->cfg.chains;
  S.chain_offset = e->cfg.chain_offset;
  S.y = e->dss_y.ptr;
  S.X = e->dss_X.ptr;
  S.observed = e->dss_obs.ptr;
  S.Xt = e->dss_Xt.ptr;
  S.yt = e->dss_yt.ptr;
  S.obs_mask = e->dss_obs_mask.ptr;
  S.lane_major = ss_lane_major(*e) ? 1 : 0;
  S.TP = (int32_t)ss_pitch(*e);
  S.gamma = e->dgamma.ptr;
  S.beta = e->dbeta.ptr;
  S.sigsq = e->dsigsq.ptr;
  S.level_sigsq = e->dlev_sigsq.ptr;
  S.level_n = e->dlev_n.ptr;
  S.level_sumsq = e->dlev_sumsq.ptr;
  S.level_prior_df = e->level_prior_df;
  S.level_prior_ss = e->level_prior_ss;
  S.level_sigma_max = e->level_sigma_max;
  S.a0 = e->ss_a0;
  S.P0 = e->ss_P0;
  S.seed_lo = (uint32_t)e->seed;
  S.seed_hi = (uint32_t)(e->seed >> 32);
  S.pos_level = e->dpos_level.ptr;
  S.pos_state = e->dpos_state.ptr;
  S.status = e->dstatus.ptr;
  S.scratch = e->dss_scratch.ptr;
  S.scratch_stride = (int64_t)SS_SCRATCH_ARRAYS * (int64_t)ss_pitch(*e);
  S.xty = e->dxty_c.ptr;
  S.yty = e->dyty_c.ptr;
  S.nobs = e->dnobs_c.ptr;
  S.xte_planes = e->dxte_planes.ptr;
  S.prepared = 0;
  S.prep_n = e->dprep_n.ptr;
  S.prep_pos_state = e->dprep_pos_state.ptr;
  S.prep_pos_level = e->dprep_pos_level.ptr;
  S.prep_level_sigsq = e->dprep_level.ptr;
  S.zbuf = e->ss_zbuf;
  S.level_used = e->ssla.lev_used.ptr;
  if (e->ssm_set) {
    S.ssm.spec = reinterpret_cast<const SsgSpec *>(e->dssg_spec.ptr);
    S.ssm.m = e->ssg.m;
    S.ssm.nblocks = e->ssg.nblocks;
    S.ssm.nvar = e->ssg.nvar;
    S.ssm.nar = e->ssg.nar;
    S.ssm.ld = e->ssg.ld;
    S.ssm.bl = e->ssg.bl;
    S.ssm.nerr = e->ssg.nerr;
    if (e->ssg_kernel_choice == 1 || e->ssg_kernel_choice == 3)
      ssg_template_shape(e->ssg, &S.ssm.tpl_trend, &S.ssm.tpl_nseasons, &S.ssm.tpl_ar_lags);
    S.ssm.glob = 0;
    for (int b = 0; b < e->ssg.nblocks; ++b)
      if (e->ssg.blk[b].kind == SSG_TRIG || e->ssg.blk[b].kind == SSG_SEMILOCAL) S.ssm.glob = 1;
    S.ssm.var_sigsq = e->dssm_sigsq.ptr;
    S.ssm.var_n = e->dssm_n.ptr;
    S.ssm.var_ss = e->dssm_ss.ptr;
    S.ssm.pos_var = e->dpos_var.ptr;
    S.ssm.ar_phi = e->dar_phi.ptr;
    S.ssm.ar_suf = e->dar_suf.ptr;
    S.ssm.work = e->dssm_work.ptr;
    S.ssm.work_stride = ssm_work_stride(*e);
  }
}

// the state half of a state-space sweep: the structural kernel when a trend /
// seasonal specification is set, the local-level kernel otherwise
hipError_t launch_state_kernel(ba_engine *e, const SsParams &S, int draw) {
  return e->ssm_set ? launch_ssm_simsmooth(e->stream, S, draw)
                    : launch_kalman_simsmooth(e->stream, S, draw);
}

// The same for the state-space path, where a chain's sweeps alternate with the
// Kalman kernel: a chain that outgrew the capacity sat out the rest of the call
// (its SSVS launches booked the sweeps, its Kalman launches were skipped), so
// it is caught up one (SSVS, Kalman) pair at a time; chains that owe nothing
// leave both kernels at once.
int ss_escalate(ba_engine *e, std::vector<int32_t> &st) {
  const size_t C = (size_t)e->cfg.chains;
  for (;;) {
    bool any = false;
    for (size_t c = 0; c < C; ++c) any = any || (st[c] == CHAIN_MODEL_TOO_LARGE);
    if (!any) return BA_OK;
    if (e->cfg.max_model_size_hint > 0) return BA_OK;  // stays an error
    const bool to_big = e->kcap >= cap_limit(*e);
    if (to_big) {
      int stuck = 0;
      int rc = grow_big(e, &stuck);
      if (rc) return rc;
      if (stuck) return BA_OK;
    } else {
      e->kcap += 16;
      e->waves = choose_waves(*e, e->kcap);
    }
    std::vector<int32_t> todo(C);
    HIP_TRY(hipMemcpy(todo.data(), e->dtodo.ptr, C * 4, hipMemcpyDeviceToHost));
    int rounds = 0;
    for (size_t c = 0; c < C; ++c) {
      if (st[c] == CHAIN_MODEL_TOO_LARGE) {
        if (!to_big) st[c] = CHAIN_OK;   // (the large-model kernel takes parked chains as they are)
        rounds = std::max(rounds, (int)todo[c]);
      }
    }
    HIP_TRY(hipMemcpyAsync(e->dstatus.ptr, st.data(), C * 4, hipMemcpyHostToDevice, e->stream));
    SsvsParams P;
    fill_params(e, P);
    P.run_limit = 1;
    P.ran = e->dran.ptr;
    SsParams S;
    fill_ss_params(e, S);
    S.only_ran = e->dran.ptr;
    for (int r = 0; r < rounds; ++r) {
      HIP_TRY(launch_sweeps(e, P, 0));
      HIP_TRY(launch_state_kernel(e, S, 1));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(st.data(), e->dstatus.ptr, C * 4, hipMemcpyDeviceToHost));
  }
}

// the pinned staging buffer of the accessors' batched copies (engine struct)
hipError_t pinned_reserve(ba_engine *e, size_t bytes) {
  if (bytes <= e->pinned_bytes) return hipSuccess;
  if (e->pinned) (void)hipHostFree(e->pinned);
  e->pinned = nullptr;
  e->pinned_bytes = 0;
  const size_t want = std::max<size_t>(bytes, (size_t)1 << 16);
  hipError_t err = hipHostMalloc(&e->pinned, want, hipHostMallocDefault);
  if (err == hipSuccess) e->pinned_bytes = want;
  return err;
}

// (debugging sessions, ba_ss_set_tuning(e, 6): what THIS engine's round kernel noted, printed
// when one of its chains stops; the buffer lives on the engine's device)
int set_device(const ba_engine *e);
void dump_round_debug(ba_engine *e) {
  if (!e->round_debug || e->dround_debug.count == 0 || set_device(e)) return;
  DevBuf<int32_t> &g_round_debug = e->dround_debug;
  int32_t h[16 * 17];
  if (hipMemcpy(h, g_round_debug.ptr, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return;
  std::fprintf(stderr, "round kernel: %d sums that are not numbers\n", h[0]);
  for (int i = 0; i < std::min(h[0], 15); ++i) {
    const int32_t *o = h + 16 + i * 16;
    std::fprintf(stderr, "  chain %d round %d of %d variable %d row %d tile %d slot %d members %d bits %08x%08x\n", o[0], o[1], o[9],
                 o[2], o[3], o[4], o[5], o[6], (unsigned)o[7], (unsigned)o[8]);
  }
  {
    double d[32];
    if (hipMemcpy(d, g_round_debug.ptr + 16 * 17, sizeof d, hipMemcpyDeviceToHost) != hipSuccess) return;
    for (int i = 0; i < std::min(h[1], 4); ++i)
      std::fprintf(stderr, "  stopped in the state draw: chain %g round %g status %g sigsq %.17g level_sigsq %.17g prep_n %g level_sumsq %.17g level_n %g\n",
                   d[i * 8], d[i * 8 + 1], d[i * 8 + 2], d[i * 8 + 3], d[i * 8 + 4], d[i * 8 + 5], d[i * 8 + 6], d[i * 8 + 7]);
  }
  (void)hipMemset(g_round_debug.ptr, 0, (16 * 17 + 64) * 4);
}

int check_chain_status(ba_engine *e) {
  const size_t C = (size_t)e->cfg.chains;
  if (!e->state_ready) return BA_OK;
  std::vector<int32_t> st(C);
  // the status words and the launches' largest model in ONE round trip
  const bool follow = e->cfg.max_model_size_hint <= 0 && e->kcap > 0;
  HIP_TRY(pinned_reserve(e, C * 4 + 16));
  int32_t *hst = (int32_t *)e->pinned, *hmaxk = hst + C;
  HIP_TRY(hipMemcpyAsync(hst, e->dstatus.ptr, C * 4, hipMemcpyDeviceToHost, e->stream));
  if (follow) HIP_TRY(hipMemcpyAsync(hmaxk, e->dmaxk.ptr, 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  std::memcpy(st.data(), hst, C * 4);
  bool all_ok = true;
  for (size_t c = 0; c < C; ++c) all_ok = all_ok && st[c] == CHAIN_OK;
  {
    int rc = BA_OK;
    if (!all_ok) rc = e->ss_mode ? ss_escalate(e, st) : escalate(e, st);
    if (rc) return rc;
    // capacity follows the models: room for growth, no more
    if (follow) {
      int32_t maxk = *hmaxk;
      if (all_ok) {
        HIP_TRY(hipMemsetAsync(e->dmaxk.ptr, 0, 4, e->stream));   // (in order before the next launch)
      } else {   // (a catch-up has run since the copy above)
        HIP_TRY(hipMemcpyAsync(&maxk, e->dmaxk.ptr, 4, hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipMemsetAsync(e->dmaxk.ptr, 0, 4, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
      }
      const int want = std::min(cap_limit(*e), std::max(16, ((maxk + 8 + 15) / 16) * 16));
      if (maxk > 0 && want < e->kcap) {
        e->kcap = want;
        e->waves = choose_waves(*e, e->kcap);
      }
      if (maxk > 0 && maxk <= 56) e->big_active = false;  // every chain is back in the LDS kernel
    }
  }
  for (size_t c = 0; c < C; ++c) {
    if (st[c] != CHAIN_OK) {
      char buf[64];
      std::snprintf(buf, sizeof buf, " (chain %lld)",
                    (long long)(e->cfg.chain_offset + (int64_t)c));
      dump_round_debug(e);
      {  // (and which chains)
        int bad = 0;
        for (size_t d = 0; d < C; ++d) bad += st[d] != CHAIN_OK;
        if (e->round_debug && e->dround_debug.count) {
          std::fprintf(stderr, "  %d chains stopped:", bad);
          for (size_t d = 0; d < C; ++d) if (st[d] != CHAIN_OK) std::fprintf(stderr, " %zu(%d)", d, st[d]);
          std::fprintf(stderr, "\n");
        }
      }
      return fail(status_code(st[c]), std::string(status_message(st[c])) + buf);
    }
  }
  return BA_OK;
}

int set_device(const ba_engine *e) {
  HIP_TRY(hipSetDevice(e->cfg.device));
  return BA_OK;
}

// ---- a second stream that really runs beside the first --------------------------------
// HIP maps streams onto a few hardware queues; two streams on one queue run one after the
// other, whatever the program meant (measured: the bsts round 207 instead of 167 us when
// the engine's two streams happen to share a queue, which depends on how many streams the
// process created before).  So a candidate is TESTED: a kernel on the main stream waits
// (bounded: 20 ms) for a flag that a kernel on the candidate sets; the first candidate whose
// kernel gets through while the other is waiting is kept.
__global__ void stream_probe_wait_kernel(volatile int *flag, int *saw, long long ticks) {
  const long long t0 = wall_clock64();
  int v = 0;
  while ((v = *flag) == 0 && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  *saw = v;
}
__global__ void stream_probe_set_kernel(volatile int *flag) { *flag = 1; }

int concurrent_stream(ba_engine *e, hipStream_t *out) {
  DevBuf<int32_t> buf;
  HIP_TRY(buf.resize(2));
  hipStream_t tried[8];
  int ntried = 0;
  hipStream_t good = nullptr;
  for (; ntried < 8 && !good; ++ntried) {
    hipStream_t c = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    tried[ntried] = c;
    HIP_TRY(hipMemsetAsync(buf.ptr, 0, 8, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    // (wall_clock64 counts at 100 MHz: 2e6 ticks = 20 ms)
    hipLaunchKernelGGL(stream_probe_wait_kernel, dim3(1), dim3(1), 0, e->stream, (volatile int *)buf.ptr,
                       (int *)buf.ptr + 1, 2000000ll);
    hipLaunchKernelGGL(stream_probe_set_kernel, dim3(1), dim3(1), 0, c, (volatile int *)buf.ptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipStreamSynchronize(c));
    int32_t saw = 0;
    HIP_TRY(hipMemcpy(&saw, buf.ptr + 1, 4, hipMemcpyDeviceToHost));
    if (saw) good = c;
  }
  // (none ran beside the main stream: the last one serves, in sequence)
  if (!good) good = tried[ntried - 1];
  for (int i = 0; i < ntried; ++i)
    if (tried[i] != good) (void)hipStreamDestroy(tried[i]);
  *out = good;
  return BA_OK;
}

// ---- pipelined sweeps ---------------------------------------------------------------
// The end of a pipeline: the main stream waits for the other one, so that whatever is
// enqueued next comes after every sweep launch; a workgroup that waited for its chain in
// vain (it cannot happen while the chains fit the machine; bounded all the same) is an error.
int pipe_join(ba_engine *e) {
  if (!e->pipe_on) return BA_OK;
  e->pipe_on = false;
  e->pipe_k = 0;
  e->pipe_groups = false;
  HIP_TRY(hipEventRecord(e->pipe_join_ev, e->pipe_stream));
  HIP_TRY(hipStreamWaitEvent(e->stream, e->pipe_join_ev, 0));
  return BA_OK;
}
// (force: read the word whatever the flag says -- the look-ahead's batches, whose launches
// and checks interleave)
int pipe_check(ba_engine *e, bool force = false) {
  if (e->dpipe_err.count == 0 || (!force && !e->pipe_unchecked)) return BA_OK;
  e->pipe_unchecked = false;
  int32_t err = 0;
  HIP_TRY(hipMemcpy(&err, e->dpipe_err.ptr, 4, hipMemcpyDeviceToHost));
  if (err) {
    HIP_TRY(hipMemset(e->dpipe_err.ptr, 0, 4));
    return fail(BA_E_HIP, "a pipelined sweep launch waited for a chain in vain");
  }
  return BA_OK;
}

// ---- look-ahead serving (ba_draw_next) ------------------------------------------
int sweep_impl(ba_engine *e, int32_t nsweeps, bool record = true, int la_half = -1);
int read_record(ba_engine *e, int64_t c, int row0, int nrows, uint8_t *gamma,
                double *beta, double *sigsq);
int read_record_row_all(ba_engine *e, int row, uint8_t *gamma, double *beta, double *sigsq);

void la_discard(ba_engine *e) {
  e->la_avail = e->la_served = 0;
  e->la_cache.clear();
  e->la_synced = false;
}

int la_copy(ba_engine *e, bool save, int set = 0);
int la_redo_batch(ba_engine *e);
int ss_la_settle(ba_engine *e);

// the batch being served is complete and sound; a pipelined batch in which a chain stopped
// (capacity, an error) is run again the old way -- same draws -- where those are dealt with
int la_wait(ba_engine *e) {
  if (e->la_synced) return BA_OK;
  if (e->la_cur_piped) {
    HIP_TRY(hipEventSynchronize(e->la_done[e->la_slot]));
    int rc = pipe_check(e, true);
    if (rc) return rc;
    const size_t C = (size_t)e->cfg.chains;
    std::vector<int32_t> st(C);
    HIP_TRY(hipMemcpy(st.data(), e->dstatus.ptr, C * 4, hipMemcpyDeviceToHost));
    bool ok = true;
    for (size_t c = 0; c < C; ++c) ok = ok && st[c] == CHAIN_OK;
    if (!ok) {
      rc = la_redo_batch(e);
      if (rc) return rc;
    }
  } else {
    HIP_TRY(hipStreamSynchronize(e->stream));
    int rc = check_chain_status(e);
    if (rc) return rc;
  }
  e->la_synced = true;
  return BA_OK;
}

// the draw ba_draw_next is serving, for one chain: from the host copy of the
// chain's rows of the batch (fetched at the chain's first read in the batch)
int la_read(ba_engine *e, int64_t c, uint8_t *gamma, double *beta, double *sigsq) {
  {
    int rc = la_wait(e);
    if (rc) return rc;
  }
  const size_t p = (size_t)e->p, cap = (size_t)e->rec_cap, n = (size_t)e->la_avail;
  auto it = e->la_cache.find(c);
  if (it == e->la_cache.end()) {
    ba_engine::LaRows r;
    r.k.resize(n); r.sig.resize(n); r.beta.resize(n * cap); r.idx.resize(n * cap);
    const size_t base = (size_t)c * e->trace_stride + (size_t)e->la_slot * (size_t)e->la_len;
    HIP_TRY(hipMemcpy(r.k.data(), e->dtr_k.ptr + base, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r.sig.data(), e->dtr_sig.ptr + base, n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r.idx.data(), e->drec_idx.ptr + base * cap, n * cap * 2, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r.beta.data(), e->drec_beta.ptr + base * cap, n * cap * 8, hipMemcpyDeviceToHost));
    it = e->la_cache.emplace(c, std::move(r)).first;
  }
  const ba_engine::LaRows &r = it->second;
  const size_t row = (size_t)e->la_served - 1;
  const int k = (int)r.k[row];
  if (k < 0 || (size_t)k > cap) return fail(BA_E_STATE, "corrupt draw record");
  if (gamma) std::memset(gamma, 0, p);
  if (beta) std::memset(beta, 0, p * 8);
  for (int m = 0; m < k; ++m) {
    const size_t j = r.idx[row * cap + m];
    if (j >= p) return fail(BA_E_STATE, "corrupt draw record");
    if (gamma) gamma[j] = 1;
    if (beta) beta[j] = r.beta[row * cap + m];
  }
  if (sigsq) *sigsq = r.sig[row];
  return BA_OK;
}

// snapshot set `set` (0 / 1) <-> the live chain state
int la_snap_alloc(ba_engine *e) {
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p;
  HIP_TRY(e->snap_gamma.resize(2 * C * p));
  HIP_TRY(e->snap_beta.resize(2 * C * p));
  HIP_TRY(e->snap_sigsq.resize(2 * C));
  HIP_TRY(e->snap_perm.resize(2 * C * p));
  HIP_TRY(e->snap_pos.resize(2 * C));
  HIP_TRY(e->snap_fail.resize(2 * C));
  HIP_TRY(e->snap_inc.resize(2 * C * p));
  HIP_TRY(e->snap_bsum.resize(2 * C * p));
  HIP_TRY(e->snap_bsumsq.resize(2 * C * p));
  HIP_TRY(e->snap_acc.resize(2 * C * ACC_COUNT));
  return BA_OK;
}
int la_copy(ba_engine *e, bool save, int set) {
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p;
  hipStream_t s = e->stream;
  if (save) {
    int rc = la_snap_alloc(e);
    if (rc) return rc;
  }
  const size_t o1 = (size_t)set * C, op = (size_t)set * C * p;
#define LA_CP(snap, live, bytes)                                                      \
  HIP_TRY(hipMemcpyAsync(save ? (void *)(snap) : (void *)(live),                       \
                         save ? (const void *)(live) : (const void *)(snap), (bytes), \
                         hipMemcpyDeviceToDevice, s))
  LA_CP(e->snap_gamma.ptr + op, e->dgamma.ptr, C * p);
  LA_CP(e->snap_beta.ptr + op, e->dbeta.ptr, C * p * 8);
  LA_CP(e->snap_sigsq.ptr + o1, e->dsigsq.ptr, C * 8);
  LA_CP(e->snap_perm.ptr + op, e->dperm.ptr, C * p * 2);
  LA_CP(e->snap_pos.ptr + o1, e->dpos.ptr, C * 8);
  LA_CP(e->snap_fail.ptr + o1, e->dfail.ptr, C * 4);
  LA_CP(e->snap_inc.ptr + op, e->dinc.ptr, C * p * 4);
  LA_CP(e->snap_bsum.ptr + op, e->dbsum.ptr, C * p * 8);
  LA_CP(e->snap_bsumsq.ptr + op, e->dbsumsq.ptr, C * p * 8);
  LA_CP(e->snap_acc.ptr + (size_t)set * C * ACC_COUNT, e->dacc.ptr, C * ACC_COUNT * 8);
#undef LA_CP
  return BA_OK;
}

// every launch of the look-ahead has finished (the main stream has caught up with the other
// one) and nothing is ahead any more; *ahead_was: a batch beyond the one being served ran
int la_quiesce(ba_engine *e, bool *ahead_was) {
  *ahead_was = e->la_ahead;
  e->la_ahead = false;
  int rc = pipe_join(e);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  return BA_OK;
}

// A pipelined batch met a chain that cannot go on as it is (model beyond the launch's
// capacity, an error): everything in flight is dropped, the chains go back to the batch's
// start and the batch runs again the way batches ran before they overlapped -- the same
// draws, with the escalation / error report of that path.  Overlap stays off afterwards.
int la_redo_batch(ba_engine *e) {
  bool ahead = false;
  int rc = la_quiesce(e, &ahead);
  if (rc) return rc;
  const int served = e->la_served;
  rc = la_copy(e, false, e->la_slot);
  if (rc) return rc;
  {  // (the statuses and the sweeps booked as owed belong to the dropped launches)
    const size_t C = (size_t)e->cfg.chains;
    HIP_TRY(hipMemsetAsync(e->dstatus.ptr, 0, C * 4, e->stream));
    HIP_TRY(hipMemsetAsync(e->dtodo.ptr, 0, C * 4, e->stream));
  }
  e->la_pipe = false;
  e->la_cur_piped = false;
  e->la_slot = 0;
  e->table_ok = false;
  e->model_ok = false;
  rc = la_copy(e, true, 0);
  if (!rc) rc = sweep_impl(e, e->la_len);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->la_cache.clear();
  e->la_served = served;
  return check_chain_status(e);
}

// Something other than ba_draw_next is about to touch the engine while draws of
// the look-ahead batch are still unserved: put the chains where the caller has
// seen them -- the batch's start, replayed up to the last draw handed out (same
// stream positions, so the same draws).
int la_rewind(ba_engine *e) {
  if (e->la_served >= e->la_avail && !e->la_ahead) {
    la_discard(e);
    return BA_OK;
  }
  HIP_TRY(hipSetDevice(e->cfg.device));
  bool ahead = false;
  int rc = la_quiesce(e, &ahead);
  if (rc) return rc;
  const bool all_served = e->la_served >= e->la_avail;
  const int replay = all_served ? 0 : e->la_served;
  // where to go back to: the start of the batch being served, or -- every draw of it
  // served, the next one already run -- the start of that next one
  const int set = all_served ? (e->la_slot ^ 1) : e->la_slot;
  const bool piped = e->la_cur_piped;
  la_discard(e);
  if (piped) {
    rc = pipe_check(e, true);
    if (rc) return rc;
    const size_t C = (size_t)e->cfg.chains;
    // (a chain that stopped in the dropped launches stopped after the point we return to)
    HIP_TRY(hipMemsetAsync(e->dstatus.ptr, 0, C * 4, e->stream));
    HIP_TRY(hipMemsetAsync(e->dtodo.ptr, 0, C * 4, e->stream));
  } else {
    rc = check_chain_status(e);
    if (rc) return rc;
  }
  rc = la_copy(e, false, piped ? set : 0);
  if (rc) return rc;
  e->la_cur_piped = false;
  e->la_slot = 0;
  e->table_ok = false;
  e->model_ok = false;
  if (replay > 0) {
    rc = sweep_impl(e, replay, /*record=*/false);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    rc = check_chain_status(e);
  }
  return rc;
}

// ---- look-ahead on the bsts path --------------------------------------------------------
int ss_sweep_impl(ba_engine *e, int32_t nsweeps, int rec_slot);

// what a round leaves for the callers' loop, copied into the record: one workgroup per
// chain (gamma, beta, sigma^2, the state models' variances and coefficients), then one per
// registered chain (its state path)
struct SsRecParams {
  int32_t C, p, nvar, nphi, nreg, L, row;   // row: slot * L + round
  int64_t var_stride, phi_stride, state_stride, state_doubles;
  const uint8_t *gamma;
  const double *beta, *sigsq, *var, *phi, *state;
  const int32_t *reg;
  uint8_t *rgamma;
  double *rbeta, *rsig, *rvar, *rphi, *rstate;
};
__global__ __launch_bounds__(256) void ss_record_kernel(SsRecParams R) {
  const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
  const int slot = R.row / R.L, i = R.row % R.L;
  if (b < R.C) {
    const size_t at = ((size_t)slot * R.C + b) * R.L + i;
    const size_t p = (size_t)R.p;
    for (size_t j = tid; j < p; j += 256) {
      R.rgamma[at * p + j] = R.gamma[(size_t)b * p + j];
      R.rbeta[at * p + j] = R.beta[(size_t)b * p + j];
    }
    if (tid == 0) R.rsig[at] = R.sigsq[b];
    if (tid < R.nvar) R.rvar[at * R.nvar + tid] = R.var[(size_t)b * R.var_stride + tid];
    if (tid < R.nphi) R.rphi[at * R.nphi + tid] = R.phi[(size_t)b * R.phi_stride + tid];
  } else {
    const int r = b - R.C;
    const size_t c = (size_t)R.reg[r];
    const size_t at = ((size_t)slot * R.nreg + r) * R.L + i;
    const double *src = R.state + c * (size_t)R.state_stride;
    double *dst = R.rstate + at * (size_t)R.state_doubles;
    for (int64_t j = tid; j < R.state_doubles; j += 256) dst[j] = src[j];
  }
}

bool ss_la_on(const ba_engine *e) { return e->ssla.len > 1; }
bool ss_la_serving(const ba_engine *e) { return e->ssla.len > 1 && e->ssla.avail > 0 && !e->ssla.busy; }

hipError_t ss_la_record(ba_engine *e, int slot, int round) {
  ba_engine::SsLa &A = e->ssla;
  SsRecParams R{};
  R.C = e->cfg.chains;
  R.p = e->p;
  R.nvar = (int32_t)A.nvar;
  R.nphi = (int32_t)A.nphi;
  R.nreg = (int32_t)A.reg.size();
  R.L = A.len;
  R.row = slot * A.len + round;
  R.gamma = e->dgamma.ptr;
  R.beta = e->dbeta.ptr;
  R.sigsq = e->dsigsq.ptr;
  if (e->ssm_set) {
    R.var = e->dssm_sigsq.ptr;
    R.var_stride = SSG_MAX_VAR;
    R.phi = e->dar_phi.ptr;
    R.phi_stride = SSG_MAX_AR * AR_MAX;
    R.state = e->dssm_work.ptr + (size_t)e->ssg.m * e->T;
    R.state_stride = ssm_work_stride(*e);
  } else {
    R.var = e->ssla.lev_used.ptr;   // (the live value may be the NEXT round's: drawn ahead)
    R.var_stride = 1;
    R.phi = nullptr;
    R.phi_stride = 0;
    R.state = e->dss_scratch.ptr + (size_t)SS_STATE_ARRAY * ss_pitch(*e);
    R.state_stride = (int64_t)SS_SCRATCH_ARRAYS * (int64_t)ss_pitch(*e);
  }
  R.state_doubles = (int64_t)A.state_doubles;
  R.reg = A.dreg.ptr;
  R.rgamma = A.rgamma.ptr;
  R.rbeta = A.rbeta.ptr;
  R.rsig = A.rsig.ptr;
  R.rvar = A.rvar.ptr;
  R.rphi = A.rphi.ptr;
  R.rstate = A.rstate.ptr;
  hipLaunchKernelGGL(ss_record_kernel, dim3((unsigned)(R.C + R.nreg)), dim3(256), 0, e->stream, R);
  return hipGetLastError();
}

// the record's and the snapshots' buffers for the current specification
int ss_la_alloc(ba_engine *e) {
  ba_engine::SsLa &A = e->ssla;
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p, L = (size_t)A.len;
  A.nvar = e->ssm_set ? (size_t)e->ssg.nvar : 1;
  A.nphi = e->ssm_set ? (size_t)e->ssg.nar * AR_MAX : 0;
  A.state_doubles = e->ssm_set ? (size_t)e->ssg.m * e->T : ss_pitch(*e);
  {
    // chains whose state path was read since the last allocation join the record while their
    // rows fit in 2 GiB (m = 64, T = 2048, 256 rounds: half a gigabyte per chain); the others
    // keep being read by going back to the draw being served
    const double per_chain = 2.0 * (double)L * (double)A.state_doubles * 8.0;
    for (int32_t w : A.want)
      if (((double)A.reg.size() + 1.0) * per_chain <= 2147483648.0) A.reg.push_back(w);
    A.want.clear();
  }
  const size_t nreg = A.reg.size();
  HIP_TRY(A.rgamma.resize(2 * C * L * p));
  HIP_TRY(A.rbeta.resize(2 * C * L * p));
  HIP_TRY(A.rsig.resize(2 * C * L));
  HIP_TRY(A.rvar.resize(2 * C * L * A.nvar));
  HIP_TRY(A.rphi.resize(2 * C * L * std::max<size_t>(A.nphi, 1)));
  HIP_TRY(A.rstate.resize(2 * nreg * L * A.state_doubles));
  HIP_TRY(A.dreg.resize(nreg));
  HIP_TRY(A.lev_used.resize(C));
  HIP_TRY(hipMemcpy(A.dreg.ptr, A.reg.data(), nreg * 4, hipMemcpyHostToDevice));
  {  // (the round kernel's view of the same list: chain -> its index among the registered)
    std::vector<int32_t> of(C, -1);
    for (size_t r = 0; r < nreg; ++r) of[(size_t)A.reg[r]] = (int32_t)r;
    HIP_TRY(e->dround_reg.resize(C));
    HIP_TRY(hipMemcpy(e->dround_reg.ptr, of.data(), C * 4, hipMemcpyHostToDevice));
  }
  // snapshot: level (sigsq, n, sumsq) | xty | yty | nobs [| state models: sigsq, n, ss | phi | ar suf]
  A.snap_doubles = C * (3 + p + 2);
  A.snap_words = 2 * C;
  if (e->ssm_set) {
    A.snap_doubles += C * (3 * SSG_MAX_VAR + SSG_MAX_AR * AR_MAX + SSG_MAX_AR * AR_SUF_STRIDE);
    A.snap_words += C * SSG_MAX_VAR;
  }
  HIP_TRY(A.snap.resize(2 * A.snap_doubles));
  HIP_TRY(A.snap_pos.resize(2 * A.snap_words));
  for (int i = 0; i < 2; ++i)
    if (!A.done[i]) HIP_TRY(hipEventCreateWithFlags(&A.done[i], hipEventDisableTiming));
  return BA_OK;
}

// snapshot set `set` <-> the live chain state (both halves: the regression's by la_copy)
int ss_la_copy(ba_engine *e, bool save, int set) {
  ba_engine::SsLa &A = e->ssla;
  int rc = la_copy(e, save, set);
  if (rc) return rc;
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p;
  hipStream_t s = e->stream;
  double *d = A.snap.ptr + (size_t)set * A.snap_doubles;
  uint64_t *w = A.snap_pos.ptr + (size_t)set * A.snap_words;
#define SS_CP(live, n, T_)                                                                     \
  do {                                                                                          \
    const size_t bytes__ = (size_t)(n) * sizeof(T_);                                            \
    if ((live) && bytes__)                                                                      \
      HIP_TRY(hipMemcpyAsync(save ? (void *)cur__ : (void *)(live), save ? (const void *)(live) : (const void *)cur__, \
                             bytes__, hipMemcpyDeviceToDevice, s));                             \
    cur__ += (n);                                                                               \
  } while (0)
  {
    double *cur__ = d;
    SS_CP(e->dlev_sigsq.ptr, C, double);
    SS_CP(e->dlev_n.ptr, C, double);
    SS_CP(e->dlev_sumsq.ptr, C, double);
    SS_CP(e->dxty_c.ptr, C * p, double);
    SS_CP(e->dyty_c.ptr, C, double);
    SS_CP(e->dnobs_c.ptr, C, double);
    if (e->ssm_set) {
      SS_CP(e->dssm_sigsq.ptr, C * SSG_MAX_VAR, double);
      SS_CP(e->dssm_n.ptr, C * SSG_MAX_VAR, double);
      SS_CP(e->dssm_ss.ptr, C * SSG_MAX_VAR, double);
      SS_CP(e->dar_phi.ptr, e->ssg.nar > 0 ? C * SSG_MAX_AR * AR_MAX : 0, double);
      SS_CP(e->dar_suf.ptr, e->ssg.nar > 0 ? C * SSG_MAX_AR * AR_SUF_STRIDE : 0, double);
    }
  }
  {
    uint64_t *cur__ = w;
    SS_CP(e->dpos_level.ptr, C, uint64_t);
    SS_CP(e->dpos_state.ptr, C, uint64_t);
    if (e->ssm_set) SS_CP(e->dpos_var.ptr, C * SSG_MAX_VAR, uint64_t);
  }
#undef SS_CP
  return BA_OK;
}

// enqueue one batch into half `slot`: the snapshot of where it starts, then `len` rounds,
// each followed by its record
int ss_la_launch(ba_engine *e, int slot) {
  ba_engine::SsLa &A = e->ssla;
  A.busy = true;
  int rc = ss_la_copy(e, true, slot);
  if (!rc) rc = ss_sweep_impl(e, A.cur, slot);
  A.busy = false;
  if (rc) return rc;
  HIP_TRY(hipEventRecord(A.done[slot], e->stream));
  return BA_OK;
}

void ss_la_reset(ba_engine *e) {
  ba_engine::SsLa &A = e->ssla;
  A.avail = A.served = 0;
  A.slot = 0;
  A.ahead = false;
  A.synced = false;
  A.cache.clear();
}

// The chains as the caller has seen them: nothing of the look-ahead left in flight.  The
// batch being served is restored to its start and replayed up to the draw handed out
// last (same stream positions, so the same draws).
int ss_la_settle(ba_engine *e) {
  ba_engine::SsLa &A = e->ssla;
  if (A.len <= 1 || A.busy || A.avail == 0) return BA_OK;
  A.busy = true;
  struct Unbusy { ba_engine::SsLa &a; ~Unbusy() { a.busy = false; } } unbusy{A};
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->stream2) HIP_TRY(hipStreamSynchronize(e->stream2));
  const int served = A.served, slot = A.slot;
  const bool at_end = served >= A.avail && !A.ahead;   // the chains ARE at the draw served last
  // (With the whole batch handed out and the next one running, the next batch's own snapshot
  // IS the chains at the draw served last -- but not the state PATH of that draw, which a
  // forecast or another chain's state read asks for and only the replay brings back: the
  // batch is replayed then too.)
  ss_la_reset(e);
  if (at_end) return check_chain_status(e);   // (nothing dropped, nothing replayed: free)
  // a rewind and a replay follow (see SsLa::cur: whoever made this necessary may do so after
  // every draw, so the batches get shorter)
  A.clean = false;
  if (A.cur <= 2 && A.cur > 1) A.probe_wait = std::min(1024, A.probe_wait * 2);
  A.cur = std::max(1, A.cur / 2);
  A.calm = 0;
  int rc = ss_la_copy(e, false, slot);
  if (rc) return rc;
  {  // (a chain that stopped in the dropped rounds stopped after the point we return to)
    const size_t C = (size_t)e->cfg.chains;
    HIP_TRY(hipMemsetAsync(e->dstatus.ptr, 0, C * 4, e->stream));
    HIP_TRY(hipMemsetAsync(e->dtodo.ptr, 0, C * 4, e->stream));
  }
  e->table_ok = false;
  e->model_ok = false;
  if (served > 0) {
    rc = ss_sweep_impl(e, served, -1);
    if (rc) return rc;
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  return check_chain_status(e);
}

// the batch being served is complete and every chain went through it; a batch in which
// a chain stopped (capacity, an error) is run again round by round, with the stops dealt
// with where they happen -- the same draws
int ss_la_wait(ba_engine *e) {
  ba_engine::SsLa &A = e->ssla;
  if (A.synced) return BA_OK;
  HIP_TRY(hipEventSynchronize(A.done[A.slot]));
  const size_t C = (size_t)e->cfg.chains;
  std::vector<int32_t> st(C);
  HIP_TRY(hipMemcpy(st.data(), e->dstatus.ptr, C * 4, hipMemcpyDeviceToHost));
  bool ok = true;
  for (size_t c = 0; c < C; ++c) ok = ok && st[c] == CHAIN_OK;
  if (!ok) {
    A.busy = true;
    struct Unbusy { ba_engine::SsLa &a; ~Unbusy() { a.busy = false; } } unbusy{A};
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->stream2) HIP_TRY(hipStreamSynchronize(e->stream2));
    const int slot = A.slot;
    A.ahead = false;
    int rc = ss_la_copy(e, false, slot);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(e->dstatus.ptr, 0, C * 4, e->stream));
    HIP_TRY(hipMemsetAsync(e->dtodo.ptr, 0, C * 4, e->stream));
    e->table_ok = false;
    e->model_ok = false;
    rc = ss_la_copy(e, true, slot);   // (the same starting point, for a later settle)
    for (int i = 0; i < A.avail && !rc; ++i) {
      rc = ss_sweep_impl(e, 1, -1);
      if (!rc) HIP_TRY(hipStreamSynchronize(e->stream));
      if (!rc) rc = check_chain_status(e);   // (escalates, catches the chain up, reports errors)
      if (!rc) HIP_TRY(ss_la_record(e, slot, i));
    }
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    A.cache.clear();
  }
  A.synced = true;
  return BA_OK;
}

// one chain's rows of the batch being served, on the host (one set of copies per batch)
int ss_la_rows(ba_engine *e, int64_t c, bool want_state, const ba_engine::SsLa::Rows **out) {
  ba_engine::SsLa &A = e->ssla;
  int rc = ss_la_wait(e);
  if (rc) return rc;
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p, L = (size_t)A.len;
  auto it = A.cache.find(c);
  if (it == A.cache.end()) {
    ba_engine::SsLa::Rows r;
    r.gamma.resize(L * p); r.beta.resize(L * p); r.sig.resize(L); r.var.resize(L * A.nvar); r.phi.resize(L * A.nphi);
    const size_t at = ((size_t)A.slot * C + (size_t)c) * L;
    HIP_TRY(hipMemcpy(r.gamma.data(), A.rgamma.ptr + at * p, L * p, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r.beta.data(), A.rbeta.ptr + at * p, L * p * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r.sig.data(), A.rsig.ptr + at, L * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(r.var.data(), A.rvar.ptr + at * A.nvar, L * A.nvar * 8, hipMemcpyDeviceToHost));
    if (A.nphi) HIP_TRY(hipMemcpy(r.phi.data(), A.rphi.ptr + at * A.nphi, L * A.nphi * 8, hipMemcpyDeviceToHost));
    it = A.cache.emplace(c, std::move(r)).first;
  }
  if (want_state && !it->second.has_state) {
    size_t ri = 0;
    while (ri < A.reg.size() && A.reg[ri] != c) ++ri;
    if (ri == A.reg.size()) return fail(BA_E_STATE, "the chain's state path is not in the look-ahead's record");
    it->second.state.resize(L * A.state_doubles);
    const size_t at = ((size_t)A.slot * A.reg.size() + ri) * L;
    HIP_TRY(hipMemcpy(it->second.state.data(), A.rstate.ptr + at * A.state_doubles, L * A.state_doubles * 8,
                      hipMemcpyDeviceToHost));
    it->second.has_state = true;
  }
  *out = &it->second;
  return BA_OK;
}
bool ss_la_registered(const ba_engine *e, int64_t c) {
  for (int32_t r : e->ssla.reg)
    if (r == c) return true;
  return false;
}
// A chain whose state path was asked for and is not in the record: this read goes back to the
// draw being served (ss_la_settle), the batches from here on record the chain too -- a caller
// that reads chain c after every draw pays for it once, not every time.
// (the list the device buffers are sized by, `reg`, changes in ss_la_alloc only; the state
// record is bounded there)
void ss_la_want_state(ba_engine *e, int64_t c) {
  ba_engine::SsLa &A = e->ssla;
  if (ss_la_registered(e, c)) return;
  for (int32_t w : A.want)
    if (w == c) return;
  if (A.reg.size() + A.want.size() < 32) A.want.push_back((int32_t)c);
}

struct ApiScope {
  ba_engine *e;
  explicit ApiScope(ba_engine *en) : e(en) { e->api_seq++; e->api_depth++; }
  ~ApiScope() { e->api_depth--; }
  ApiScope(const ApiScope &) = delete;
};

// (a mutator changes what the next launch reads -- priors, data, options -- through copies
// on the main stream: a pipelined sweep launch still running on the other stream must be
// behind the main stream first, or it would see half of the new values)
#define MUTATE(e)                        \
  do {                                   \
    (e)->api_seq++;                      \
    int rc_m__ = set_device(e);          \
    if (!rc_m__) rc_m__ = pipe_join(e);  \
    if (!rc_m__) rc_m__ = ss_la_settle(e); \
    if (!rc_m__) rc_m__ = la_rewind(e);  \
    if (rc_m__) return rc_m__;           \
    (e)->table_ok = false;               \
    (e)->model_ok = false;               \
  } while (0)

// (accessors that enqueue nothing and change nothing: they do not count as a call
// between two ba_sync()s)
#define ENGINE_ACCESSOR_NOJOIN(e)                              \
  if (!(e)) return fail(BA_E_INVALID, "null engine");          \
  g_kt = (e)->kt_enabled ? &(e)->kt : nullptr;                 \
  {                                                            \
    int rc__ = set_device(e);                                  \
    if (rc__) return rc__;                                     \
  }
#define ENGINE_PROLOGUE_NOJOIN(e)                              \
  ENGINE_ACCESSOR_NOJOIN(e)                                    \
  ApiScope api_scope__(e);
// (everything but ba_sweep itself first lets the main stream catch up with a pipeline
// of sweep launches)
#define ENGINE_PROLOGUE(e)                                     \
  ENGINE_PROLOGUE_NOJOIN(e)                                    \
  {                                                            \
    int rc__ = pipe_join(e);                                   \
    if (!rc__) rc__ = ss_la_settle(e);                         \
    if (rc__) return rc__;                                     \
  }
#define ENGINE_ACCESSOR(e)                                     \
  ENGINE_ACCESSOR_NOJOIN(e)                                    \
  {                                                            \
    int rc__ = pipe_join(e);                                   \
    if (!rc__) rc__ = ss_la_settle(e);                         \
    if (rc__) return rc__;                                     \
  }
// (the entry points that serve the draw of ba_ss_draw_next from the device's record)
#define ENGINE_ACCESSOR_SERVED(e)                              \
  ENGINE_ACCESSOR_NOJOIN(e)                                    \
  {                                                            \
    int rc__ = pipe_join(e);                                   \
    if (rc__) return rc__;                                     \
  }

}  // namespace

extern "C" {

const char *ba_last_error(void) { return g_error.c_str(); }

int ba_engine_create(const ba_config *cfg, ba_engine **out) {
  if (!cfg || !out) return fail(BA_E_INVALID, "null argument");
  if (cfg->chains <= 0) return fail(BA_E_INVALID, "chains must be positive");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev <= 0)
    return fail(BA_E_HIP, "no HIP device visible: boom_amd has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(BA_E_INVALID, "device ordinal out of range");
  ba_engine *e = new ba_engine();
  e->cfg = *cfg;
  e->seed = cfg->seed;
  hipError_t err = hipSetDevice(cfg->device);
  if (err == hipSuccess) err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
  if (err != hipSuccess) {
    delete e;
    return fail(BA_E_HIP, std::string("stream creation: ") + hipGetErrorString(err));
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess) {
    e->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.maxSharedMemoryPerMultiProcessor > 0)
      e->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
  }
  // the shuffle's LDS-exchange search needs same-address exchanges resolved in
  // lane order; refuse a device that does not
  {
    int *dbad = nullptr, hbad = -1;
    err = hipMalloc((void **)&dbad, sizeof(int));
    if (err == hipSuccess) err = hipMemsetAsync(dbad, 0, sizeof(int), e->stream);
    if (err == hipSuccess) err = launch_lds_exchange_order(e->stream, dbad);
    if (err == hipSuccess) err = hipMemcpyAsync(&hbad, dbad, sizeof(int), hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (dbad) (void)hipFree(dbad);
    if (err != hipSuccess || hbad != 0) {
      (void)hipStreamDestroy(e->stream);
      delete e;
      if (err != hipSuccess) return fail(BA_E_HIP, std::string("device self-test: ") + hipGetErrorString(err));
      return fail(BA_E_HIP, "device self-test: LDS exchanges are not resolved in lane order on this device");
    }
  }
  *out = e;
  return BA_OK;
}

void ba_engine_destroy(ba_engine *e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  if (e->stream2) {
    (void)hipStreamSynchronize(e->stream2);
    (void)hipStreamDestroy(e->stream2);
  }
  for (int i = 0; i < 2; ++i) {
    if (e->la_done[i]) (void)hipEventDestroy(e->la_done[i]);
    if (e->ssla.done[i]) (void)hipEventDestroy(e->ssla.done[i]);
  }
  if (e->pipe_stream) {
    (void)hipStreamSynchronize(e->pipe_stream);
    (void)hipStreamDestroy(e->pipe_stream);
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(e->pipe_ev[i]);
    (void)hipEventDestroy(e->pipe_join_ev);
  }
  if (e->ev_state) (void)hipEventDestroy(e->ev_state);
  if (e->pinned) (void)hipHostFree(e->pinned);
  for (int i = 0; i < 2; ++i)
    if (e->ev_prep[i]) (void)hipEventDestroy(e->ev_prep[i]);
  if (e->stream) {
    (void)hipStreamSynchronize(e->stream);
    (void)hipStreamDestroy(e->stream);
  }
  delete e;
}

int ba_engine_info(const ba_engine *e, int32_t *device, int32_t *chains,
                   int32_t *p) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  if (device) *device = e->cfg.device;
  if (chains) *chains = e->cfg.chains;
  if (p) *p = e->p;
  return BA_OK;
}

// (work the caller puts on the stream after this call comes after every launch of the engine:
// sweeps that overlap on the engine's second stream are joined first)
void *ba_stream(ba_engine *e) {
  if (!e) return nullptr;
  if (hipSetDevice(e->cfg.device) != hipSuccess) return nullptr;
  if (pipe_join(e) != BA_OK) return nullptr;
  return (void *)e->stream;
}

// ---- measurement: device time per kernel class (ktimer.h) ------------------------
int32_t ba_kernel_classes(void) { return KT_CLASSES; }

const char *ba_kernel_class_name(int32_t cls) {
  static const char *const names[KT_CLASSES] = {
      "ssvs_sweep_kernel", "ssvs_big_kernel", "ssvs_adaptive_kernel", "kalman_simsmooth_kernel",
      "ssm_simsmooth_kernel", "atb_mfma_kernel", "probit_impute_kernel", "logit_impute_kernel",
      "xtwx_cols_kernel<false>+plain_reduce_kernel", "xtwx_cols_kernel<true>+xtwx_cols_reduce_kernel",
      "xtx_mfma_kernel+plane_sum_kernel+col_reduce_kernel", "poisson_impute_kernel",
      "kalman_prepare_kernel", "ss_round_kernel"};
  return (cls >= 0 && cls < KT_CLASSES) ? names[cls] : "";
}

int ba_set_kernel_timing(ba_engine *e, int32_t enabled) {
  ENGINE_PROLOGUE(e);
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->kt.collect();
  e->kt_enabled = enabled != 0;
  e->kt_overlap = enabled == 2;
  g_kt = e->kt_enabled ? &e->kt : nullptr;
  return BA_OK;
}

int ba_get_kernel_times(ba_engine *e, double *ms, int64_t *launches, int32_t reset) {
  ENGINE_PROLOGUE(e);
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->kt.collect();
  for (int c = 0; c < KT_CLASSES; ++c) {
    if (ms) ms[c] = e->kt.ms[c];
    if (launches) launches[c] = e->kt.launches[c];
    if (reset) { e->kt.ms[c] = 0; e->kt.launches[c] = 0; }
  }
  return BA_OK;
}

// ---------------------------------------------------------------- data
static int set_dimension(ba_engine *e, int p) {
  if (p <= 0 || p > 65535)
    return fail(BA_E_INVALID, "number of predictors must be in [1, 65535]");
  if (e->p != p) {
    e->p = p;
    e->have_slab = e->have_spike = false;
    e->state_ready = false;
  }
  return BA_OK;
}

int ba_upload_regression_suf(ba_engine *e, int32_t p, const double *xtx,
                             const double *xty, double yty, double n,
                             double ybar, const double *xbar) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (!xtx || !xty || !xbar) return fail(BA_E_INVALID, "null argument");
  int rc = set_dimension(e, p);
  if (rc) return rc;
  e->xtx.assign(xtx, xtx + (size_t)p * p);
  e->xty.assign(xty, xty + p);
  e->xsum.resize(p);
  for (int j = 0; j < p; ++j) e->xsum[j] = xbar[j] * n;
  e->yty = yty;
  e->n = n;
  e->sumy = ybar * n;
  e->have_suf = true;
  e->device_dirty = true;
  e->probit_mode = e->logit_mode = e->ss_mode = e->poisson_mode = false;   // (plain regression data now; the binomial, Poisson and state-space setters say otherwise after this)
  return BA_OK;
}

int ba_build_suf_from_xy_device(ba_engine *e, int64_t n, int32_t p,
                                const void *X_device, const void *y_device) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (!X_device || !y_device) return fail(BA_E_INVALID, "null argument");
  if (n <= 0) return fail(BA_E_INVALID, "n must be positive");
  int rc = set_dimension(e, p);
  if (rc) return rc;
  HIP_TRY(e->dxtx.resize((size_t)p * p));
  HIP_TRY(e->dxty.resize(p));
  HIP_TRY(e->dxsum.resize(p));
  HIP_TRY(e->dsufscal.resize(2));
  DevBuf<double> planes;  // split-K partial products (freed after the sync below)
  const int slices = suf_row_slices(n, p);
  if (slices > 1) HIP_TRY(planes.resize((size_t)slices * p * p));
  rc = launch_suf_from_xy(e->stream, n, p, (const double *)X_device,
                          (const double *)y_device, e->dxtx.ptr, e->dxty.ptr,
                          e->dsufscal.ptr, e->dxsum.ptr, planes.ptr);
  if (rc) return fail(BA_E_HIP, "suf kernel launch failed");
  e->xtx.resize((size_t)p * p);
  e->xty.resize(p);
  e->xsum.resize(p);
  double sc[2];
  HIP_TRY(hipMemcpyAsync(e->xtx.data(), e->dxtx.ptr, (size_t)p * p * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(e->xty.data(), e->dxty.ptr, (size_t)p * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(e->xsum.data(), e->dxsum.ptr, (size_t)p * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(sc, e->dsufscal.ptr, 16, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->yty = sc[0];
  e->sumy = sc[1];
  e->n = (double)n;
  e->have_suf = true;
  e->device_dirty = true;
  e->probit_mode = e->logit_mode = e->ss_mode = e->poisson_mode = false;   // (plain regression data now; the binomial, Poisson and state-space setters say otherwise after this)
  return BA_OK;
}

// ---- row-sharded build: partial statistics of a shard of rows, summed by the
// caller over its ranks (one all-reduce), then installed on every rank
size_t ba_suf_block_size(int32_t p) { return (size_t)p * p + 2 * (size_t)p + 2; }

int ba_suf_partial_device(ba_engine *e, int64_t n_rows, int32_t p, const void *X_device,
                          const void *y_device, void *block_device) {
  ENGINE_PROLOGUE(e);
  if (!X_device || !y_device || !block_device) return fail(BA_E_INVALID, "null argument");
  if (n_rows <= 0 || p <= 0 || p > 65535) return fail(BA_E_INVALID, "bad shard dimensions");
  double *blk = (double *)block_device;
  DevBuf<double> planes;
  const int slices = suf_row_slices(n_rows, p);
  if (slices > 1) HIP_TRY(planes.resize((size_t)slices * p * p));
  // block layout: [XtX p*p | Xty p | yty, sum y | column sums of X p]
  int rc = launch_suf_from_xy(e->stream, n_rows, p, (const double *)X_device,
                              (const double *)y_device, blk, blk + (size_t)p * p,
                              blk + (size_t)p * p + p, blk + (size_t)p * p + p + 2, planes.ptr);
  if (rc) return fail(BA_E_HIP, "suf kernel launch failed");
  HIP_TRY(hipStreamSynchronize(e->stream));
  return BA_OK;
}

int ba_set_suf_from_block_device(ba_engine *e, int64_t n_total, int32_t p,
                                 const void *block_device) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (!block_device) return fail(BA_E_INVALID, "null argument");
  if (n_total <= 0) return fail(BA_E_INVALID, "n must be positive");
  int rc = set_dimension(e, p);
  if (rc) return rc;
  const size_t pp = (size_t)p * p;
  std::vector<double> h(ba_suf_block_size(p));
  HIP_TRY(hipMemcpyAsync(h.data(), block_device, h.size() * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->xtx.assign(h.begin(), h.begin() + pp);
  e->xty.assign(h.begin() + pp, h.begin() + pp + p);
  e->yty = h[pp + p];
  e->sumy = h[pp + p + 1];
  e->xsum.assign(h.begin() + pp + p + 2, h.end());
  e->n = (double)n_total;
  e->have_suf = true;
  e->device_dirty = true;
  e->probit_mode = e->logit_mode = e->ss_mode = e->poisson_mode = false;   // (plain regression data now; the binomial, Poisson and state-space setters say otherwise after this)
  return BA_OK;
}

int ba_build_suf_from_xy(ba_engine *e, int64_t n, int32_t p, const double *X,
                         const double *y) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (!X || !y) return fail(BA_E_INVALID, "null argument");
  if (n <= 0 || p <= 0) return fail(BA_E_INVALID, "n and p must be positive");
  HIP_TRY(e->dX.resize((size_t)n * p));
  HIP_TRY(e->dy.resize((size_t)n));
  HIP_TRY(hipMemcpyAsync(e->dX.ptr, X, (size_t)n * p * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->dy.ptr, y, (size_t)n * 8, hipMemcpyHostToDevice, e->stream));
  int rc = ba_build_suf_from_xy_device(e, n, p, e->dX.ptr, e->dy.ptr);
  e->dX.release();
  e->dy.release();
  return rc;
}

int ba_get_regression_suf(ba_engine *e, double *xtx, double *xty, double *yty,
                          double *n, double *ybar, double *xbar) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  if (!e->have_suf) return fail(BA_E_STATE, "no regression data set");
  const int p = e->p;
  if (xtx) std::memcpy(xtx, e->xtx.data(), (size_t)p * p * 8);
  if (xty) std::memcpy(xty, e->xty.data(), (size_t)p * 8);
  if (yty) *yty = e->yty;
  if (n) *n = e->n;
  if (ybar) *ybar = e->sumy / e->n;
  if (xbar)
    for (int j = 0; j < p; ++j) xbar[j] = e->xsum[j] / e->n;
  return BA_OK;
}

// -------------------------------------------------------------- priors
int ba_set_slab(ba_engine *e, const double *prior_mean,
                const double *unscaled_prior_precision) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  if (!prior_mean || !unscaled_prior_precision) return fail(BA_E_INVALID, "null argument");
  if (e->p <= 0) return fail(BA_E_STATE, "set the regression data before the priors");
  const int p = e->p;
  e->b.assign(prior_mean, prior_mean + p);
  e->ominv.assign(unscaled_prior_precision, unscaled_prior_precision + (size_t)p * p);
  e->have_slab = true;
  e->device_dirty = true;
  return BA_OK;
}

int ba_set_spike(ba_engine *e, const double *pi, int64_t max_model_size) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  if (!pi) return fail(BA_E_INVALID, "null argument");
  if (e->p <= 0) return fail(BA_E_STATE, "set the regression data before the priors");
  for (int j = 0; j < e->p; ++j)
    if (!(pi[j] >= 0.0 && pi[j] <= 1.0))
      return fail(BA_E_INVALID, "prior inclusion probabilities must be in [0, 1]");
  e->pi.assign(pi, pi + e->p);
  e->max_model_size = max_model_size;
  e->have_spike = true;
  e->device_dirty = true;
  return BA_OK;
}

int ba_set_sigma_prior(ba_engine *e, double prior_df, double sigma_guess,
                       double sigma_upper_limit) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  if (sigma_upper_limit < 0) return fail(BA_E_INVALID, "sigma_max must be non-negative.");
  // ChisqModel(df, sigma): alpha = df/2, beta = df sigma^2/2 (ChisqModel.cpp:56-57)
  const double alpha = prior_df / 2.0;
  const double beta = prior_df * sigma_guess * sigma_guess / 2.0;
  e->prior_df = 2 * alpha;
  e->prior_ss = 2 * beta;
  e->sigma_guess = sigma_guess;
  e->sigma_max = sigma_upper_limit;
  e->have_sigma = true;
  return BA_OK;
}

int ba_set_priors_ctor1(ba_engine *e, double prior_nobs, double expected_rsq,
                        double expected_model_size,
                        int32_t first_term_is_intercept) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  if (!e->have_suf) return fail(BA_E_STATE, "no regression data set");
  if (!(expected_rsq > 0 && expected_rsq < 1)) return fail(BA_E_INVALID, "expected_rsq must be in (0, 1)");
  // BregVsSampler.cpp:37-44, 48-85
  const int p = e->p;
  const double n = e->n, ybar = e->sumy / n;
  const double sst = e->yty - n * ybar * ybar;
  const double sigma_guess = std::sqrt(sst / (n - 1) * (1 - expected_rsq));
  std::vector<double> b(p, 0.0), om((size_t)p * p), pi(p);
  if (first_term_is_intercept) b[0] = ybar;
  for (size_t i = 0; i < (size_t)p * p; ++i) om[i] = e->xtx[i] * (prior_nobs / n);
  double prob = expected_model_size / p;
  if (prob > 1) prob = 1.0;
  std::fill(pi.begin(), pi.end(), prob);
  if (first_term_is_intercept) pi[0] = 1.0;
  int rc = ba_set_slab(e, b.data(), om.data());
  if (!rc) rc = ba_set_spike(e, pi.data(), -1);
  if (!rc) rc = ba_set_sigma_prior(e, prior_nobs, sigma_guess, e->sigma_max);
  return rc;
}

int ba_set_priors_ctor2(ba_engine *e, double prior_sigma_nobs,
                        double prior_sigma_guess, double prior_beta_nobs,
                        double diagonal_shrinkage,
                        double prior_inclusion_probability,
                        int32_t force_intercept) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  if (!e->have_suf) return fail(BA_E_STATE, "no regression data set");
  // BregVsSampler.cpp:87-142
  if (prior_sigma_guess <= 0)
    return fail(BA_E_INVALID, "illegal value of prior_sigma_guess in constructor to BregVsSampler");
  const double alpha = diagonal_shrinkage;
  if (alpha > 1.0 || alpha < 0.0)
    return fail(BA_E_INVALID, "illegal value of 'diagonal_shrinkage' in BregVsSampler constructor.");
  const int p = e->p;
  const double n = e->n;
  std::vector<double> b(p, 0.0), om((size_t)p * p), pi(p, prior_inclusion_probability);
  b[0] = e->sumy / n;
  for (size_t i = 0; i < (size_t)p * p; ++i) om[i] = e->xtx[i] * (prior_beta_nobs / n);
  if (alpha < 1.0) {
    for (int j = 0; j < p; ++j) {
      const double d = om[(size_t)j * p + j];
      om[(size_t)j * p + j] = d + d * (alpha / (1 - alpha));
    }
    for (auto &v : om) v *= (1 - alpha);
  } else {
    for (int j = 0; j < p; ++j)
      for (int i = 0; i < p; ++i)
        if (i != j) om[(size_t)j * p + i] = 0.0;
  }
  if (force_intercept) pi[0] = 1.0;
  int rc = ba_set_slab(e, b.data(), om.data());
  if (!rc) rc = ba_set_spike(e, pi.data(), -1);
  if (!rc) rc = ba_set_sigma_prior(e, prior_sigma_nobs, prior_sigma_guess, e->sigma_max);
  return rc;
}

int ba_get_priors(ba_engine *e, double *prior_mean, double *ominv, double *pi,
                  double *prior_df, double *prior_ss) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  if (!e->have_slab || !e->have_spike || !e->have_sigma)
    return fail(BA_E_STATE, "priors not set");
  const int p = e->p;
  if (prior_mean) std::memcpy(prior_mean, e->b.data(), (size_t)p * 8);
  if (ominv) std::memcpy(ominv, e->ominv.data(), (size_t)p * p * 8);
  if (pi) std::memcpy(pi, e->pi.data(), (size_t)p * 8);
  if (prior_df) *prior_df = e->prior_df;
  if (prior_ss) *prior_ss = e->prior_ss;
  return BA_OK;
}

int ba_set_options(ba_engine *e, int32_t max_flips, double swap_threshold,
                   int32_t draw_beta, int32_t draw_sigma) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  e->max_flips = max_flips;
  if (swap_threshold != e->swap_threshold) e->device_dirty = true;
  e->swap_threshold = swap_threshold;
  e->draw_beta = draw_beta;
  e->draw_sigma = draw_sigma;
  return BA_OK;
}

int ba_set_tuning(ba_engine *e, int32_t waves_per_chain, int32_t walk_policy,
                  int32_t kcap_start) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  if (!(waves_per_chain == 0 || waves_per_chain == 1 || waves_per_chain == 2 || waves_per_chain == 4))
    return fail(BA_E_INVALID, "waves_per_chain must be 0, 1, 2 or 4");
  if (walk_policy < -1 || walk_policy > 3) return fail(BA_E_INVALID, "walk_policy must be in [-1, 3]");
  if (kcap_start < 0) return fail(BA_E_INVALID, "kcap_start must be non-negative");
  MUTATE(e);
  if (e->state_ready) {
    int rc = set_device(e);
    if (!rc) rc = ba_sync(e);
    if (rc) return rc;
  }
  e->tune_waves = waves_per_chain;
  e->tune_walk_policy = walk_policy;
  e->tune_kcap_start = kcap_start;
  e->device_dirty = true;  // capacity and waves are chosen again
  return BA_OK;
}

// --------------------------------------------------------------- state
int ba_set_state(ba_engine *e, int64_t chain, const uint8_t *gamma,
                 const double *beta, double sigsq) {
  ENGINE_PROLOGUE(e);
  if (e->p <= 0) return fail(BA_E_STATE, "set the regression data first");
  if (!gamma) return fail(BA_E_INVALID, "null argument");
  const int64_t C = e->cfg.chains;
  if (chain < -1 || chain >= C) return fail(BA_E_INVALID, "chain index out of range");
  if ((e->probit_mode || e->logit_mode) && sigsq != 1.0)
    return fail(BA_E_INVALID, "the binomial samplers' latent data have unit variance: sigsq must be 1");
  if (chain < 0) la_discard(e);  // every chain is overwritten: nothing to rewind to
  MUTATE(e);
  // launches in flight (and sweeps still owed after a capacity stop) belong to
  // the OLD state: resolve them before it is overwritten.  Setting every chain
  // also clears chain errors (the caller starts over).
  if (e->state_ready) {
    int rc = ba_sync(e);
    if (rc && chain >= 0) return rc;
    if (rc) {
      HIP_TRY(hipMemsetAsync(e->dstatus.ptr, 0, (size_t)C * 4, e->stream));
      HIP_TRY(hipMemsetAsync(e->dtodo.ptr, 0, (size_t)C * 4, e->stream));
    }
  }
  int rc = alloc_chain_state(e);
  if (rc) return rc;
  const size_t p = (size_t)e->p;
  std::vector<double> zeros;
  if (!beta) {
    zeros.assign(p, 0.0);
    beta = zeros.data();
  }
  hipStream_t s = e->stream;
  if (chain < 0) {
    std::vector<uint8_t> G((size_t)C * p);
    std::vector<double> B((size_t)C * p), S((size_t)C, sigsq);
    for (int64_t c = 0; c < C; ++c) {
      std::memcpy(&G[(size_t)c * p], gamma, p);
      std::memcpy(&B[(size_t)c * p], beta, p * 8);
    }
    HIP_TRY(hipMemcpyAsync(e->dgamma.ptr, G.data(), G.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->dbeta.ptr, B.data(), B.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->dsigsq.ptr, S.data(), S.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
  } else {
    HIP_TRY(hipMemcpyAsync(e->dgamma.ptr + (size_t)chain * p, gamma, p, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->dbeta.ptr + (size_t)chain * p, beta, p * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->dsigsq.ptr + chain, &sigsq, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return BA_OK;
}

int ba_get_state(ba_engine *e, int64_t chain, uint8_t *gamma, double *beta,
                 double *sigsq) {
  ENGINE_ACCESSOR_NOJOIN(e);
  if (!e->state_ready) return fail(BA_E_STATE, "no chain state yet");
  if (chain < 0 || chain >= e->cfg.chains) return fail(BA_E_INVALID, "chain index out of range");
  if (e->la_served > 0 && e->la_served <= e->la_avail)  // the draw ba_draw_next is serving
    return la_read(e, chain, gamma, beta, sigsq);
  {
    int rcj = pipe_join(e);
    if (rcj) return rcj;
  }
  if (ss_la_serving(e)) {   // the draw ba_ss_draw_next is serving
    const ba_engine::SsLa::Rows *r = nullptr;
    int rcr = ss_la_rows(e, chain, false, &r);
    if (rcr) return rcr;
    const size_t p = (size_t)e->p, row = (size_t)e->ssla.served - 1;
    if (gamma) std::memcpy(gamma, &r->gamma[row * p], p);
    if (beta) std::memcpy(beta, &r->beta[row * p], p * 8);
    if (sigsq) *sigsq = r->sig[row];
    return BA_OK;
  }
  int rc = ba_sync(e);
  if (rc) return rc;
  // (one batch through the pinned staging buffer: beta | sigsq | gamma)
  const size_t p = (size_t)e->p;
  HIP_TRY(pinned_reserve(e, p * 9 + 16));
  double *hb = (double *)e->pinned, *hs = hb + p;
  uint8_t *hg = (uint8_t *)(hs + 1);
  if (gamma) HIP_TRY(hipMemcpyAsync(hg, e->dgamma.ptr + (size_t)chain * p, p, hipMemcpyDeviceToHost, e->stream));
  if (beta) HIP_TRY(hipMemcpyAsync(hb, e->dbeta.ptr + (size_t)chain * p, p * 8, hipMemcpyDeviceToHost, e->stream));
  if (sigsq) HIP_TRY(hipMemcpyAsync(hs, e->dsigsq.ptr + chain, 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (gamma) std::memcpy(gamma, hg, p);
  if (beta) std::memcpy(beta, hb, p * 8);
  if (sigsq) *sigsq = *hs;
  return BA_OK;
}

// PosteriorSampler::logpri() of BregVsSampler (BregVsSampler.cpp:380-393) for one
// chain's current state: log p(gamma) + log p(sigma^2) + log N(beta_g | b_g,
// sigma^2 Omega_g).  Host arithmetic on the chain's state and the host copies of
// the priors (a k x k Cholesky): it is interface, not hot path.
int ba_logpri(ba_engine *e, int64_t chain, double *out) {
  ENGINE_ACCESSOR_SERVED(e);   // (ba_get_state below sees the draw being served)
  if (!out) return fail(BA_E_INVALID, "null argument");
  if (!e->have_slab || e->pi.empty()) return fail(BA_E_STATE, "priors are not set");
  const size_t p = (size_t)e->p;
  std::vector<uint8_t> g(p);
  std::vector<double> beta(p);
  double sigsq = 0.0;
  int rc = ba_get_state(e, chain, g.data(), beta.data(), &sigsq);
  if (rc) return rc;
  const double ninf = -std::numeric_limits<double>::infinity();
  std::vector<int> idx;
  for (size_t j = 0; j < p; ++j)
    if (g[j]) idx.push_back((int)j);
  const int k = (int)idx.size();
  // VariableSelectionPrior::logp (VariableSelectionPrior.cpp:271-285)
  double ans = 0.0;
  if (e->max_model_size >= 0 && k > e->max_model_size) ans = ninf;
  for (size_t j = 0; j < p && ans > ninf; ++j) {
    ans += g[j] ? std::log(e->pi[j]) : std::log(1.0 - e->pi[j]);
    if (!std::isfinite(ans)) ans = ninf;
  }
  if (!(ans > ninf)) {
    *out = ninf;
    return BA_OK;
  }
  // GenericGaussianVarianceSampler::log_prior: Gamma(df/2, ss/2) density of
  // 1/sigma^2 and the Jacobian of the reciprocal
  const double a = 0.5 * e->prior_df, b = 0.5 * e->prior_ss, x = 1.0 / sigsq;
  ans += a * std::log(b) - std::lgamma(a) + (a - 1.0) * std::log(x) - b * x - 2.0 * std::log(sigsq);
  if (k > 0) {
    // dmvn(beta_g, b_g, Omega^{-1}_g / sigma^2, log)
    std::vector<double> L((size_t)k * k, 0.0), d(k);
    for (int c = 0; c < k; ++c)
      for (int r = c; r < k; ++r) L[(size_t)r * k + c] = e->ominv[(size_t)idx[c] * p + idx[r]] / sigsq;
    for (int i = 0; i < k; ++i) d[i] = beta[idx[i]] - e->b[idx[i]];
    double quad = 0.0;  // Mdist on the matrix itself, before it is overwritten
    for (int c = 0; c < k; ++c) {
      quad += d[c] * d[c] * L[(size_t)c * k + c];
      for (int r = c + 1; r < k; ++r) quad += 2.0 * d[c] * d[r] * L[(size_t)r * k + c];
    }
    double ld = 0.0;
    for (int c = 0; c < k; ++c) {  // left-looking Cholesky, lower triangle in place
      double s = L[(size_t)c * k + c];
      for (int t = 0; t < c; ++t) s -= L[(size_t)c * k + t] * L[(size_t)c * k + t];
      if (!(s > 0.0)) { ld = ninf; break; }
      const double sd = std::sqrt(s);
      L[(size_t)c * k + c] = sd;
      ld += 2.0 * std::log(sd);
      for (int r = c + 1; r < k; ++r) {
        double v = L[(size_t)r * k + c];
        for (int t = 0; t < c; ++t) v -= L[(size_t)r * k + t] * L[(size_t)c * k + t];
        L[(size_t)r * k + c] = v / sd;
      }
    }
    ans += -0.5 * k * std::log(2.0 * M_PI) + 0.5 * ld - 0.5 * quad;
  }
  *out = ans;
  return BA_OK;
}

int ba_get_states(ba_engine *e, uint8_t *gamma, double *beta, double *sigsq) {
  ENGINE_ACCESSOR_SERVED(e);
  if (!e->state_ready) return fail(BA_E_STATE, "no chain state yet");
  if (ss_la_serving(e)) {
    // the draw being served, every chain: row `served - 1` of every chain's block of the record
    int rcw = ss_la_wait(e);
    if (rcw) return rcw;
    const ba_engine::SsLa &A = e->ssla;
    const size_t p = (size_t)e->p, C = (size_t)e->cfg.chains, L = (size_t)A.len;
    const size_t at = (size_t)A.slot * C * L + (size_t)A.served - 1;
    if (gamma) HIP_TRY(hipMemcpy2D(gamma, p, A.rgamma.ptr + at * p, L * p, p, C, hipMemcpyDeviceToHost));
    if (beta) HIP_TRY(hipMemcpy2D(beta, p * 8, A.rbeta.ptr + at * p, L * p * 8, p * 8, C, hipMemcpyDeviceToHost));
    if (sigsq) HIP_TRY(hipMemcpy2D(sigsq, 8, A.rsig.ptr + at, L * 8, 8, C, hipMemcpyDeviceToHost));
    return BA_OK;
  }
  if (e->la_served > 0 && e->la_served <= e->la_avail && (e->la_served < e->la_avail || e->la_ahead)) {
    // the draw being served, every chain: from the record (the chains themselves are ahead)
    int rcw = la_wait(e);
    if (rcw) return rcw;
    return read_record_row_all(e, e->la_slot * e->la_len + e->la_served - 1, gamma, beta, sigsq);
  }
  int rc = ba_sync(e);
  if (rc) return rc;
  const size_t p = (size_t)e->p, C = (size_t)e->cfg.chains;
  if (gamma) HIP_TRY(hipMemcpy(gamma, e->dgamma.ptr, C * p, hipMemcpyDeviceToHost));
  if (beta) HIP_TRY(hipMemcpy(beta, e->dbeta.ptr, C * p * 8, hipMemcpyDeviceToHost));
  if (sigsq) HIP_TRY(hipMemcpy(sigsq, e->dsigsq.ptr, C * 8, hipMemcpyDeviceToHost));
  return BA_OK;
}

int ba_seed(ba_engine *e, uint64_t seed) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  // sweeps in flight -- and sweeps still owed after a capacity stop -- belong
  // to the old key
  if (e->state_ready) {
    int rc = ba_sync(e);
    if (rc) return rc;
  }
  e->seed = seed;
  const size_t C = (size_t)e->cfg.chains;
  hipStream_t s = e->stream;
  // every sampler of every chain restarts at position 0 of its new stream
  if (e->dpos.ptr) HIP_TRY(hipMemsetAsync(e->dpos.ptr, 0, C * 8, s));
  if (e->dpos_sss.ptr) HIP_TRY(hipMemsetAsync(e->dpos_sss.ptr, 0, C * 8, s));
  if (e->dpos_ada.ptr) HIP_TRY(hipMemsetAsync(e->dpos_ada.ptr, 0, C * 8, s));
  if (e->dpos_level.ptr) HIP_TRY(hipMemsetAsync(e->dpos_level.ptr, 0, C * 8, s));
  if (e->dpos_var.ptr) HIP_TRY(hipMemsetAsync(e->dpos_var.ptr, 0, C * SSG_MAX_VAR * 8, s));
  e->probit_sweep = 0;
  if (e->dpos_state.ptr) HIP_TRY(hipMemsetAsync(e->dpos_state.ptr, 0, C * 8, s));
  if (e->dpos_forecast.ptr) HIP_TRY(hipMemsetAsync(e->dpos_forecast.ptr, 0, C * 8, s));
  HIP_TRY(hipStreamSynchronize(s));
  return BA_OK;
}

// A change of sampler (BregVs <-> SpikeSlab) or of the XtX scale inside V has
// to wait for the launches in flight (they may still be escalated).
static int switch_mode(ba_engine *e, int mode, double v_scale) {
  if (e->cur_mode == mode && e->v_scale_want == v_scale) return BA_OK;
  if (e->state_ready) {
    int rc = ba_sync(e);
    if (rc) return rc;
  }
  e->cur_mode = mode;
  e->table_ok = false;
  e->model_ok = false;
  if (e->v_scale_want != v_scale) {
    e->v_scale_want = v_scale;
    e->device_dirty = true;
  }
  return BA_OK;
}

// ------------------------------------------------------------ hot path
}  // extern "C"

namespace {
// la_half >= 0: a look-ahead batch that overlaps its neighbours -- recorded into that half of
// the draw record, the chains' state on entry saved into that snapshot set (the caller has
// checked la_can_overlap)
int sweep_impl(ba_engine *e, int32_t nsweeps, bool record, int la_half) {
  if (nsweeps < 0) return fail(BA_E_INVALID, "nsweeps must be non-negative");
  if (e->ss_mode) return fail(BA_E_STATE, "state-space data are set: use ba_ss_sweep");
  if (e->logit_mode || e->probit_mode)
    return fail(BA_E_STATE, "binomial data are set: use ba_logit_sweep / ba_probit_sweep (the regression sampler has no meaning on latent data)");
  int rc = switch_mode(e, 0, 1.0);
  if (rc) return rc;
  rc = upload_shared(e);
  if (rc) return rc;
  rc = alloc_chain_state(e);
  if (rc) return rc;
  if (record && e->trace_stride > 0 && nsweeps > e->trace_stride)
    return fail(BA_E_INVALID, "nsweeps exceeds the enabled trace length");
  HIP_TRY(e->dmodel.resize(2 * (size_t)e->cfg.chains * ssvs_scalar_layout(64).total));
  SsvsParams P;
  fill_params(e, P);
  if (!record) {  // (the record buffers belong to the look-ahead batches)
    P.trace_sigsq = P.trace_logp = P.trace_k = nullptr;
    P.rec_idx = nullptr;
    P.rec_beta = nullptr;
    P.trace_stride = 0;
    P.trace_idx = nullptr;   // ... and nothing of the record is valid any more (ba_predict checks)
    if (e->trace_stride > 0)
      HIP_TRY(hipMemsetAsync(e->dtrace_idx.ptr, 0, (size_t)e->cfg.chains * 4, e->stream));
  }
  const SsvsLds lay = ssvs_lds_layout(e->p, e->kcap);
  if (lay.total > e->lds_per_cu)
    return fail(BA_E_INVALID, "problem does not fit the LDS working set");
  if (record && e->trace_stride > 0 && la_half < 0)  // traces are those of the last ba_sweep call
    HIP_TRY(hipMemsetAsync(e->dtrace_idx.ptr, 0, (size_t)e->cfg.chains * 4, e->stream));
  if (la_half >= 0) {
    const size_t C = (size_t)e->cfg.chains, pp = (size_t)e->p;
    const size_t o1 = (size_t)la_half * C, op = (size_t)la_half * C * pp;
    P.trace_row0 = la_half * e->la_len;
    P.snap_gamma = e->snap_gamma.ptr + op;
    P.snap_beta = e->snap_beta.ptr + op;
    P.snap_sigsq = e->snap_sigsq.ptr + o1;
    P.snap_perm = e->snap_perm.ptr + op;
    P.snap_pos = e->snap_pos.ptr + o1;
    P.snap_fail = e->snap_fail.ptr + o1;
    P.snap_inc = e->snap_inc.ptr + op;
    P.snap_bsum = e->snap_bsum.ptr + op;
    P.snap_bsumsq = e->snap_bsumsq.ptr + op;
    P.snap_acc = e->snap_acc.ptr + (size_t)la_half * C * ACC_COUNT;
  }
#ifndef BA_PIPELINE
#define BA_PIPELINE 1
#endif
  // Consecutive ba_sweep calls with nothing in between: the launches alternate between two
  // streams and hand the chains over one by one (ssvs_kernel.hip), so that the next launch
  // fills the slots the current one's early finishers leave instead of waiting for its
  // slowest chain.  Only while every chain's workgroup is resident at once, no trace is
  // recorded (its cursor is reset per call) and no chain lives in the large-model kernel.
  const int resident_per_cu = (int)std::min<size_t>(4, e->lds_per_cu / lay.total);
  const bool pipelined = BA_PIPELINE && nsweeps > 0 && (e->trace_stride == 0 || la_half >= 0) && !e->big_active &&
                         e->cfg.chains <= resident_per_cu * e->cu_count && (!e->kt_enabled || e->kt_overlap);
  if (la_half >= 0 && !pipelined) return fail(BA_E_STATE, "look-ahead batch cannot overlap");
  if (!pipelined) {
    // More chains than the machine holds (round 4).  The chains go out in GROUPS of what fits
    // at once, the groups alternating between the engine's two streams: group g + 1's
    // workgroups move into the slots group g's early finishers leave -- different chains,
    // so no hand-over is needed -- and a group's next launch follows its last one on the
    // same stream.  (Rounds 1-3: one launch of more workgroups than fit -- a few workgroups
    // of the last round then start a whole round late on this machine, 53 ms per
    // 2048-chain launch where two launches of 1024 take 46 -- or, for exactly two or three
    // groups, separate launches one after the other, each as long as its slowest chain.)
    const int C = e->cfg.chains, group = resident_per_cu * e->cu_count;
    const bool groups = BA_PIPELINE && nsweeps > 0 && e->cur_mode != 2 && !e->big_active && C > group &&
                        e->trace_stride == 0 && (!e->kt_enabled || e->kt_overlap);
    if (groups) {
      if (!e->pipe_stream) {
        {
          int rcs = concurrent_stream(e, &e->pipe_stream);
          if (rcs) return rcs;
        }
        for (int i = 0; i < 4; ++i) HIP_TRY(hipEventCreateWithFlags(&e->pipe_ev[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->pipe_join_ev, hipEventDisableTiming));
      }
      if (!(e->pipe_on && e->pipe_groups)) {
        int rcj = pipe_join(e);
        if (rcj) return rcj;
        // (the other stream behind everything the main stream holds so far: uploads, mutators)
        HIP_TRY(hipEventRecord(e->pipe_ev[0], e->stream));
        HIP_TRY(hipStreamWaitEvent(e->pipe_stream, e->pipe_ev[0], 0));
      }
      SsvsParams Pg = P;
      int g = 0;
      for (int first = 0; first < C; first += group, ++g) {
        Pg.chain_first = first;
        Pg.chain_count = std::min(group, C - first);
        HIP_TRY(launch_ssvs_sweep((g & 1) ? e->pipe_stream : e->stream, Pg, (int)nsweeps));
      }
      e->pipe_on = true;
      e->pipe_groups = true;
    } else {
      int rcj = pipe_join(e);
      if (rcj) return rcj;
      HIP_TRY(launch_sweeps(e, P, (int)nsweeps));
    }
  } else {
    const size_t C = (size_t)e->cfg.chains, qlen = C + 2;
    if (!e->pipe_stream) {
      {
        int rcs = concurrent_stream(e, &e->pipe_stream);
        if (rcs) return rcs;
      }
      for (int i = 0; i < 4; ++i) HIP_TRY(hipEventCreateWithFlags(&e->pipe_ev[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->pipe_join_ev, hipEventDisableTiming));
    }
    if (e->dpipe_q.count != 4 * qlen) {
      HIP_TRY(hipStreamSynchronize(e->stream));
      HIP_TRY(e->dpipe_q.resize(4 * qlen));
      HIP_TRY(e->dpipe_err.resize(1));
      HIP_TRY(hipMemset(e->dpipe_err.ptr, 0, 4));
    }
    const int k = e->pipe_on ? e->pipe_k : 0;     // (a new pipeline starts on the main stream)
    hipStream_t st = (k & 1) ? e->pipe_stream : e->stream;
    int32_t *qout = e->dpipe_q.ptr + (size_t)(k & 3) * qlen;
    // the queue this launch fills: emptied on its own stream (after the launch two before
    // it, whose hand-over to the launch before it used the queue four back at the latest)
    HIP_TRY(hipMemsetAsync(qout, 0, 8, st));
    HIP_TRY(hipMemsetAsync(qout + 2, 0xFF, C * 4, st));
    HIP_TRY(hipEventRecord(e->pipe_ev[k & 3], st));
    P.q_out = qout;
    P.q_error = e->dpipe_err.ptr;
    if (k > 0) {
      P.q_in = e->dpipe_q.ptr + (size_t)((k - 1) & 3) * qlen;
      // (... which the previous launch's stream has emptied before that launch)
      HIP_TRY(hipStreamWaitEvent(st, e->pipe_ev[(k - 1) & 3], 0));
    }
    HIP_TRY(launch_ssvs_sweep(st, P, (int)nsweeps));
    if (la_half >= 0) HIP_TRY(hipEventRecord(e->la_done[la_half], st));
    e->pipe_on = true;
    e->pipe_unchecked = true;
    e->pipe_k = k + 1;
  }
  e->table_ok = true;  // until anything but another ba_sweep touches the engine
  e->model_ok = true;
  return BA_OK;
}
}  // namespace

extern "C" {

int ba_sweep(ba_engine *e, int32_t nsweeps) {
  ENGINE_PROLOGUE_NOJOIN(e);
  if (e->la_served < e->la_avail || e->ss_mode || e->logit_mode || e->probit_mode || e->cur_mode != 0) {
    int rcj = pipe_join(e);   // (anything but a plain continuation)
    if (rcj) return rcj;
  }
  // unserved look-ahead draws: the sweeps asked for here come after the last one served
  int rc = la_rewind(e);
  if (rc) return rc;
  return sweep_impl(e, nsweeps, /*record=*/e->la_len <= 1);
}

int ba_set_lookahead(ba_engine *e, int32_t lookahead) {
  ENGINE_PROLOGUE(e);
  if (lookahead < 1) return fail(BA_E_INVALID, "lookahead must be at least 1");
  MUTATE(e);
  if (lookahead > 1) {
    // (room for two batches: the one being served and the one launched ahead of it)
    int rc = ba_enable_draws(e, 2 * lookahead);
    if (rc) return rc;
  }
  e->la_len = lookahead;
  e->la_pipe = true;
  return BA_OK;
}

// can the next look-ahead batch overlap its neighbours (sweep_impl's conditions)
static bool la_can_overlap(const ba_engine *e) {
  if (!BA_PIPELINE || !e->la_pipe || e->big_active || (e->kt_enabled && !e->kt_overlap) || e->kcap <= 0) return false;
  const size_t lds = ssvs_lds_layout(e->p, e->kcap).total;
  if (lds > e->lds_per_cu) return false;
  const int resident_per_cu = (int)std::min<size_t>(4, e->lds_per_cu / lds);
  return e->cfg.chains <= resident_per_cu * e->cu_count;
}

int ba_draw_next(ba_engine *e) {
  ENGINE_PROLOGUE_NOJOIN(e);
  if (e->la_len <= 1) return ba_sweep(e, 1);
  if (e->la_served == e->la_avail) {
    // the record is used up: on to the next batch
    if (e->la_ahead) {
      // ... which is already running (or done): the other half of the record
      e->la_slot ^= 1;
      e->la_ahead = false;
      e->la_avail = e->la_served = 0;
      e->la_cache.clear();
      e->la_synced = false;
      e->la_cur_piped = true;
    } else {
      // ... from the chains' current state
      int rc = pipe_join(e);
      if (rc) return rc;
      la_discard(e);
      rc = switch_mode(e, 0, 1.0);
      if (!rc) rc = upload_shared(e);
      if (!rc) rc = alloc_chain_state(e);
      if (rc) return rc;
      if (!e->la_done[0]) {
        HIP_TRY(hipEventCreateWithFlags(&e->la_done[0], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->la_done[1], hipEventDisableTiming));
      }
      e->la_slot = 0;
      if (la_can_overlap(e)) {
        rc = la_snap_alloc(e);
        if (!rc) rc = sweep_impl(e, e->la_len, true, 0);
        e->la_cur_piped = true;
      } else {
        rc = la_copy(e, true, 0);
        if (!rc) rc = sweep_impl(e, e->la_len);
        e->la_cur_piped = false;
      }
      if (rc) return rc;
    }
    e->la_avail = e->la_len;
    // the batch after this one goes out now, into the other half
    if (e->la_cur_piped && la_can_overlap(e)) {
      int rc = sweep_impl(e, e->la_len, true, e->la_slot ^ 1);
      if (rc) return rc;
      e->la_ahead = true;
    }
  }
  ++e->la_served;
  return BA_OK;
}

int ba_sync(ba_engine *e) {
  ENGINE_ACCESSOR_SERVED(e);
  // (while ba_ss_draw_next serves a batch the chains run ahead on purpose: what the caller
  // waits for is the batch being served)
  if (ss_la_serving(e)) return ss_la_wait(e);
  HIP_TRY(hipStreamSynchronize(e->stream));   // (also the caller's own work on ba_stream())
  if (e->clean_seq == e->api_seq) return BA_OK;   // (no call since the last clean check: the status words are as they were)
  int rc = pipe_check(e);
  if (rc) return rc;
  const uint64_t seq = e->api_seq;   // (a catch-up inside the check is the check's own business)
  rc = check_chain_status(e);
  // (only at top level or from an accessor: a call in progress may enqueue more after this)
  if (rc == BA_OK && e->api_depth == 0) e->clean_seq = seq;
  return rc;
}

int ba_log_model_prob(ba_engine *e, int32_t ngamma, const uint8_t *gammas,
                      double *out) {
  ENGINE_PROLOGUE(e);
  if (!gammas || !out || ngamma <= 0) return fail(BA_E_INVALID, "bad argument");
  // the regression model's own sufficient statistics: in state-space mode they
  // are per chain and move every sweep, so there is no one answer
  if (e->ss_mode)
    return fail(BA_E_STATE, "ba_log_model_prob is not defined once state-space data are set (per-chain sufficient statistics)");
  // BregVsSampler's V = Omega^{-1} + XtX (a SpikeSlabSampler launch with a fixed
  // slab precision leaves XtX / sigma^2 in it)
  int rc = switch_mode(e, 0, 1.0);
  if (rc) return rc;
  rc = upload_shared(e);
  if (rc) return rc;
  rc = alloc_chain_state(e);
  if (rc) return rc;
  const size_t p = (size_t)e->p;
  DevBuf<uint8_t> dg;
  DevBuf<double> dout;
  DevBuf<int32_t> dst;
  HIP_TRY(dg.resize((size_t)ngamma * p));
  HIP_TRY(dout.resize(ngamma));
  HIP_TRY(dst.resize(ngamma));
  HIP_TRY(hipMemcpyAsync(dg.ptr, gammas, (size_t)ngamma * p, hipMemcpyHostToDevice, e->stream));
  SsvsParams P;
  fill_params(e, P);
  // this entry evaluates arbitrary models: use the largest working set
  P.kcap = 64;
  while (P.kcap > 8 && ssvs_lds_layout(e->p, P.kcap).total > e->lds_per_cu) P.kcap -= 8;
  HIP_TRY(launch_ssvs_logp(e->stream, P, (const uint8_t *)dg.ptr, (int)ngamma,
                           dout.ptr, dst.ptr));
  std::vector<int32_t> st(ngamma);
  HIP_TRY(hipMemcpyAsync(out, dout.ptr, (size_t)ngamma * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(st.data(), dst.ptr, (size_t)ngamma * 4, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  // vectors of more models than the LDS kernel holds: the large-model build, a few at
  // a time (each has a model block of two k x k factors in HBM)
  std::vector<int32_t> big;
  for (int i = 0; i < ngamma; ++i)
    if (st[i] == CHAIN_MODEL_TOO_LARGE) big.push_back(i);
  if (!big.empty()) {
    int kmax = 0;
    for (int32_t i : big) {
      int k = 0;
      for (size_t j = 0; j < p; ++j) k += gammas[(size_t)i * p + j] ? 1 : 0;
      kmax = std::max(kmax, k);
    }
    const int kcap = ((kmax + 63) / 64) * 64;
    if (kcap > BIG_KCAP_MAX || ssvs_big_lds_layout(e->p, kcap).total > e->lds_per_cu)
      return fail(BA_E_MODEL_TOO_LARGE, status_message(CHAIN_MODEL_TOO_LARGE));
    const size_t block = ssvs_scalar_layout(kcap).total, xs = (size_t)kcap * 64;
    const size_t batch = std::max<size_t>(1, std::min<size_t>(big.size(), ((size_t)256 << 20) / ((block + xs) * 8)));
    DevBuf<double> dmodel_ws, dxs_ws;
    DevBuf<int32_t> dwhich;
    HIP_TRY(dmodel_ws.resize(batch * block));
    HIP_TRY(dxs_ws.resize(batch * xs));
    HIP_TRY(dwhich.resize(big.size()));
    HIP_TRY(hipMemcpyAsync(dwhich.ptr, big.data(), big.size() * 4, hipMemcpyHostToDevice, e->stream));
    for (size_t b0 = 0; b0 < big.size(); b0 += batch) {
      const size_t nb = std::min(batch, big.size() - b0);
      HIP_TRY(launch_ssvs_big_logp(e->stream, P, kcap, (const uint8_t *)dg.ptr, dwhich.ptr + b0, (int)nb,
                                   dmodel_ws.ptr, dxs_ws.ptr, dout.ptr, dst.ptr));
    }
    HIP_TRY(hipMemcpyAsync(out, dout.ptr, (size_t)ngamma * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(st.data(), dst.ptr, (size_t)ngamma * 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  for (int i = 0; i < ngamma; ++i)
    if (st[i] != CHAIN_OK) return fail(status_code(st[i]), status_message(st[i]));
  return BA_OK;
}

// ----------------------------------------------------------- summaries
int ba_reset_summaries(ba_engine *e) {
  ENGINE_PROLOGUE(e);
  if (!e->state_ready) return BA_OK;
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p;
  hipStream_t s = e->stream;
  HIP_TRY(hipMemsetAsync(e->dinc.ptr, 0, C * p * 4, s));
  HIP_TRY(hipMemsetAsync(e->dbsum.ptr, 0, C * p * 8, s));
  HIP_TRY(hipMemsetAsync(e->dbsumsq.ptr, 0, C * p * 8, s));
  std::vector<double> acc(C * ACC_COUNT, 0.0);
  for (size_t c = 0; c < C; ++c)
    acc[c * ACC_COUNT + ACC_MIN_MARGIN] = std::numeric_limits<double>::infinity();
  HIP_TRY(hipMemcpyAsync(e->dacc.ptr, acc.data(), acc.size() * 8, hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  return BA_OK;
}

int ba_summaries_device(ba_engine *e, void *out_device) {
  ENGINE_PROLOGUE(e);
  if (!out_device) return fail(BA_E_INVALID, "null argument");
  if (!e->state_ready) return fail(BA_E_STATE, "no chain state yet");
  SsvsParams P;
  fill_params(e, P);
  HIP_TRY(launch_ssvs_reduce_summaries(e->stream, P, (double *)out_device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return BA_OK;
}

int ba_get_summaries(ba_engine *e, double *inclusion_count, double *beta_sum,
                     double *beta_sumsq, double *scalars) {
  ENGINE_PROLOGUE(e);
  int rc = ba_summaries_device(e, e->dsummary.ptr);
  if (rc) return rc;
  const size_t p = (size_t)e->p;
  std::vector<double> h(3 * p + SUMMARY_SCALARS);
  HIP_TRY(hipMemcpy(h.data(), e->dsummary.ptr, h.size() * 8, hipMemcpyDeviceToHost));
  if (inclusion_count) std::memcpy(inclusion_count, &h[0], p * 8);
  if (beta_sum) std::memcpy(beta_sum, &h[p], p * 8);
  if (beta_sumsq) std::memcpy(beta_sumsq, &h[2 * p], p * 8);
  if (scalars) std::memcpy(scalars, &h[3 * p], SUMMARY_SCALARS * 8);
  return BA_OK;
}

int ba_enable_traces(ba_engine *e, int32_t max_sweeps) {
  ENGINE_PROLOGUE(e);
  if (max_sweeps < 0) return fail(BA_E_INVALID, "max_sweeps must be non-negative");
  {  // (the recording buffers are the look-ahead's as well)
    int rc = la_rewind(e);
    if (rc) return rc;
    e->la_len = 1;
  }
  const size_t C = (size_t)e->cfg.chains;
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(e->dtr_sig.resize(C * max_sweeps));
  HIP_TRY(e->dtr_logp.resize(C * max_sweeps));
  HIP_TRY(e->dtr_k.resize(C * max_sweeps));
  e->drec_idx.release();
  e->drec_beta.release();
  e->trace_stride = max_sweeps;
  return BA_OK;
}

int ba_enable_draws(ba_engine *e, int32_t max_sweeps) {
  int rc = ba_enable_traces(e, max_sweeps);
  if (rc) return rc;
  const size_t C = (size_t)e->cfg.chains;
  e->rec_cap = std::max(64, e->big_active ? e->big_kcap : 0);
  HIP_TRY(e->drec_idx.resize(C * max_sweeps * e->rec_cap));
  HIP_TRY(e->drec_beta.resize(C * max_sweeps * e->rec_cap));
  return BA_OK;
}

}  // extern "C"

namespace {
// rows [row0, row0 + nrows) of one chain's record, expanded to dense gamma / beta
int read_record(ba_engine *e, int64_t c, int row0, int nrows, uint8_t *gamma,
                double *beta, double *sigsq) {
  const size_t p = (size_t)e->p, cap = (size_t)e->rec_cap;
  const size_t base = (size_t)c * e->trace_stride + row0;
  std::vector<double> ks(nrows), sig(nrows), b((size_t)nrows * cap);
  std::vector<uint16_t> idx((size_t)nrows * cap);
  HIP_TRY(hipMemcpy(ks.data(), e->dtr_k.ptr + base, (size_t)nrows * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(sig.data(), e->dtr_sig.ptr + base, (size_t)nrows * 8, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(idx.data(), e->drec_idx.ptr + base * cap, idx.size() * 2, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b.data(), e->drec_beta.ptr + base * cap, b.size() * 8, hipMemcpyDeviceToHost));
  if (gamma) std::memset(gamma, 0, (size_t)nrows * p);
  if (beta) std::memset(beta, 0, (size_t)nrows * p * 8);
  for (int s = 0; s < nrows; ++s) {
    const int k = (int)ks[s];
    if (k < 0 || (size_t)k > cap) return fail(BA_E_STATE, "corrupt draw record");
    for (int m = 0; m < k; ++m) {
      const size_t j = idx[(size_t)s * cap + m];
      if (j >= p) return fail(BA_E_STATE, "corrupt draw record");
      if (gamma) gamma[(size_t)s * p + j] = 1;
      if (beta) beta[(size_t)s * p + j] = b[(size_t)s * cap + m];
    }
    if (sigsq) sigsq[s] = sig[s];
  }
  return BA_OK;
}

// one row of EVERY chain's record (the draw ba_draw_next is serving)
int read_record_row_all(ba_engine *e, int row, uint8_t *gamma, double *beta, double *sigsq) {
  const size_t p = (size_t)e->p, cap = (size_t)e->rec_cap, C = (size_t)e->cfg.chains;
  const size_t stride = (size_t)e->trace_stride;
  std::vector<double> ks(C), sig(C), b(C * cap);
  std::vector<uint16_t> idx(C * cap);
  HIP_TRY(hipMemcpy2D(ks.data(), 8, e->dtr_k.ptr + row, stride * 8, 8, C, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy2D(sig.data(), 8, e->dtr_sig.ptr + row, stride * 8, 8, C, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy2D(idx.data(), cap * 2, e->drec_idx.ptr + (size_t)row * cap, stride * cap * 2,
                      cap * 2, C, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy2D(b.data(), cap * 8, e->drec_beta.ptr + (size_t)row * cap, stride * cap * 8,
                      cap * 8, C, hipMemcpyDeviceToHost));
  if (gamma) std::memset(gamma, 0, C * p);
  if (beta) std::memset(beta, 0, C * p * 8);
  for (size_t c = 0; c < C; ++c) {
    const int k = (int)ks[c];
    if (k < 0 || (size_t)k > cap) return fail(BA_E_STATE, "corrupt draw record");
    for (int m = 0; m < k; ++m) {
      const size_t j = idx[c * cap + m];
      if (j >= p) return fail(BA_E_STATE, "corrupt draw record");
      if (gamma) gamma[c * p + j] = 1;
      if (beta) beta[c * p + j] = b[c * cap + m];
    }
    if (sigsq) sigsq[c] = sig[c];
  }
  return BA_OK;
}
}  // namespace

extern "C" {

int ba_get_draws(ba_engine *e, int64_t chain, int32_t nsweeps, uint8_t *gamma,
                 double *beta, double *sigsq) {
  ENGINE_PROLOGUE(e);
  if (e->trace_stride <= 0 || e->drec_idx.count == 0)
    return fail(BA_E_STATE, "draw recording is not enabled");
  if (nsweeps <= 0 || nsweeps > e->trace_stride) return fail(BA_E_INVALID, "nsweeps out of range");
  if (chain < 0 || chain >= e->cfg.chains) return fail(BA_E_INVALID, "chain index out of range");
  int rc = ba_sync(e);
  if (rc) return rc;
  return read_record(e, chain, 0, nsweeps, gamma, beta, sigsq);
}

int ba_predict(ba_engine *e, int32_t first_draw, int32_t ndraws, int32_t nnew, const double *newX,
               double *out) {
  ENGINE_PROLOGUE(e);
  if (e->trace_stride <= 0 || e->drec_idx.count == 0)
    return fail(BA_E_STATE, "draw recording is not enabled");
  if (!newX || !out || nnew <= 0) return fail(BA_E_INVALID, "bad argument");
  if (first_draw < 0 || ndraws <= 0 || first_draw + ndraws > e->trace_stride)
    return fail(BA_E_INVALID, "draw range out of the record");
  if (ndraws > 65535 || e->cfg.chains > 65535) return fail(BA_E_INVALID, "too many draws or chains for one call");
  int rc = ba_sync(e);
  if (rc) return rc;
  const size_t C = (size_t)e->cfg.chains;
  {
    // only rows that the last recorded call actually wrote hold a draw
    std::vector<int32_t> rows(C);
    HIP_TRY(hipMemcpy(rows.data(), e->dtrace_idx.ptr, C * 4, hipMemcpyDeviceToHost));
    const int32_t have = *std::min_element(rows.begin(), rows.end());
    if (first_draw + ndraws > have)
      return fail(BA_E_INVALID, "draw range extends beyond the draws recorded by the last sweep call");
  }
  DevBuf<double> dX, dout;
  HIP_TRY(dX.resize((size_t)nnew * e->p));
  HIP_TRY(dout.resize(C * (size_t)ndraws * nnew));
  HIP_TRY(hipMemcpyAsync(dX.ptr, newX, (size_t)nnew * e->p * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(launch_predict(e->stream, e->dtr_k.ptr, e->drec_idx.ptr, e->drec_beta.ptr, e->trace_stride,
                         e->rec_cap, first_draw, ndraws, (int)C, e->p, dX.ptr, nnew, dout.ptr));
  HIP_TRY(hipMemcpyAsync(out, dout.ptr, C * (size_t)ndraws * nnew * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return BA_OK;
}

int ba_get_coefficient_traces(ba_engine *e, int32_t nsweeps, int32_t nvars,
                              const int32_t *vars, double *out) {
  ENGINE_PROLOGUE(e);
  if (e->trace_stride <= 0 || e->drec_idx.count == 0)
    return fail(BA_E_STATE, "draw recording is not enabled");
  if (nsweeps <= 0 || nsweeps > e->trace_stride) return fail(BA_E_INVALID, "nsweeps out of range");
  if (nvars <= 0 || !vars || !out) return fail(BA_E_INVALID, "bad argument");
  for (int v = 0; v < nvars; ++v)
    if (vars[v] < 0 || vars[v] >= e->p) return fail(BA_E_INVALID, "variable index out of range");
  int rc = ba_sync(e);
  if (rc) return rc;
  const size_t C = (size_t)e->cfg.chains, cap = (size_t)e->rec_cap, stride = (size_t)e->trace_stride;
  std::vector<int> slot(e->p, -1);
  for (int v = 0; v < nvars; ++v) slot[vars[v]] = v;
  std::vector<double> ks(nsweeps), b((size_t)nsweeps * cap);
  std::vector<uint16_t> idx((size_t)nsweeps * cap);
  std::memset(out, 0, C * (size_t)nvars * nsweeps * 8);
  for (size_t c = 0; c < C; ++c) {
    const size_t base = c * stride;
    HIP_TRY(hipMemcpy(ks.data(), e->dtr_k.ptr + base, (size_t)nsweeps * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(idx.data(), e->drec_idx.ptr + base * cap, idx.size() * 2, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b.data(), e->drec_beta.ptr + base * cap, b.size() * 8, hipMemcpyDeviceToHost));
    for (int s = 0; s < nsweeps; ++s) {
      const int k = (int)ks[s];
      for (int m = 0; m < k && (size_t)m < cap; ++m) {
        const int v = slot[idx[(size_t)s * cap + m] % (size_t)e->p];
        if (v >= 0) out[(c * nvars + v) * nsweeps + s] = b[(size_t)s * cap + m];
      }
    }
  }
  return BA_OK;
}

int ba_get_traces(ba_engine *e, int32_t nsweeps, double *sigsq, double *logp,
                  double *model_size) {
  ENGINE_PROLOGUE(e);
  if (e->trace_stride <= 0) return fail(BA_E_STATE, "traces are not enabled");
  if (nsweeps <= 0 || nsweeps > e->trace_stride) return fail(BA_E_INVALID, "nsweeps out of range");
  int rc = ba_sync(e);
  if (rc) return rc;
  const size_t C = (size_t)e->cfg.chains;
  auto fetch = [&](double *dst, const double *src) -> hipError_t {
    if (!dst) return hipSuccess;
    return hipMemcpy2D(dst, (size_t)nsweeps * 8, src, (size_t)e->trace_stride * 8,
                       (size_t)nsweeps * 8, C, hipMemcpyDeviceToHost);
  };
  HIP_TRY(fetch(sigsq, e->dtr_sig.ptr));
  HIP_TRY(fetch(logp, e->dtr_logp.ptr));
  HIP_TRY(fetch(model_size, e->dtr_k.ptr));
  return BA_OK;
}

// ------------------------------------------- SpikeSlabSampler (sigma^2 given)
int ba_set_sigsq(ba_engine *e, int64_t chain, double sigsq) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (!(sigsq > 0)) return fail(BA_E_INVALID, "sigsq must be positive");
  if ((e->probit_mode || e->logit_mode) && sigsq != 1.0)
    return fail(BA_E_INVALID, "the binomial samplers' latent data have unit variance: sigsq must be 1");
  int rc = alloc_chain_state(e);
  if (rc) return rc;
  const int64_t C = e->cfg.chains;
  if (chain < -1 || chain >= C) return fail(BA_E_INVALID, "chain index out of range");
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (chain < 0) {
    std::vector<double> v((size_t)C, sigsq);
    HIP_TRY(hipMemcpy(e->dsigsq.ptr, v.data(), (size_t)C * 8, hipMemcpyHostToDevice));
  } else {
    HIP_TRY(hipMemcpy(e->dsigsq.ptr + chain, &sigsq, 8, hipMemcpyHostToDevice));
  }
  return BA_OK;
}

int ba_sss_set_slab(ba_engine *e, const double *mu, const double *precision,
                    int32_t precision_scales_with_sigsq, int32_t max_flips) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  int rc = ba_set_slab(e, mu, precision);
  if (rc) return rc;
  e->sss_slab_scales = precision_scales_with_sigsq ? 1 : 0;
  e->sss_max_flips = max_flips;
  if (!e->have_sigma) {  // the sigma prior plays no role given sigma^2
    e->prior_df = 1.0;
    e->prior_ss = 1.0;
    e->have_sigma = true;
  }
  return BA_OK;
}

int ba_sss_sweep(ba_engine *e, int32_t nsweeps) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (nsweeps < 0) return fail(BA_E_INVALID, "nsweeps must be non-negative");
  if (e->ss_mode) return fail(BA_E_STATE, "state-space data are set: use ba_ss_sweep");
  if (e->logit_mode || e->probit_mode)
    return fail(BA_E_STATE, "binomial data are set: use ba_logit_sweep / ba_probit_sweep (a sweep without the imputation is not a draw of those samplers)");
  if (!e->have_slab) return fail(BA_E_STATE, "call ba_sss_set_slab first");
  int rc = alloc_chain_state(e);
  if (rc) return rc;
  double v_scale = 1.0;
  if (!e->sss_slab_scales) {
    // a slab precision that does not scale with sigma^2 makes V = P + XtX /
    // sigma^2 chain specific; the shared-matrix engine takes one sigma^2 for
    // all chains in this case (it is 1 for the logit / probit / Poisson users)
    const size_t C = (size_t)e->cfg.chains;
    rc = ba_sync(e);
    if (rc) return rc;
    std::vector<double> s2(C);
    HIP_TRY(hipMemcpy(s2.data(), e->dsigsq.ptr, C * 8, hipMemcpyDeviceToHost));
    for (size_t c = 1; c < C; ++c)
      if (s2[c] != s2[0])
        return fail(BA_E_INVALID, "a slab precision independent of sigma^2 needs the same sigma^2 in every chain");
    v_scale = 1.0 / s2[0];
  }
  rc = switch_mode(e, 1, v_scale);
  if (rc) return rc;
  rc = upload_shared(e);
  if (rc) return rc;
  HIP_TRY(e->dmodel.resize(2 * (size_t)e->cfg.chains * ssvs_scalar_layout(64).total));
  SsvsParams P;
  fill_params(e, P);
  if (e->trace_stride > 0)
    HIP_TRY(hipMemsetAsync(e->dtrace_idx.ptr, 0, (size_t)e->cfg.chains * 4, e->stream));
  HIP_TRY(launch_sweeps(e, P, (int)nsweeps));
  return BA_OK;
}

// ------------------------ BinomialProbitSpikeSlabSampler (data augmentation + SpikeSlabSampler)
int ba_probit_set_data(ba_engine *e, int64_t n, int32_t p, const double *X, const double *y,
                       const double *ntrials, int32_t clt_threshold) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (!X || !y || !ntrials) return fail(BA_E_INVALID, "null argument");
  if (n <= 0 || p <= 0) return fail(BA_E_INVALID, "n and p must be positive");
  // up to 2 * clt_threshold truncated-normal draws per observation read the
  // observation's substream of PROBIT_STRIDE uniforms (a draw takes 2 - 20 of them)
  if (clt_threshold < 0 || clt_threshold > 64)
    return fail(BA_E_INVALID, "clt_threshold must be between 0 and 64");
  for (int64_t i = 0; i < n; ++i) {
    if (y[i] < 0 || ntrials[i] < 0)
      return fail(BA_E_INVALID, "Negative values not allowed in BinomialProbitDataImputer::impute().");
    if (y[i] > ntrials[i])
      return fail(BA_E_INVALID, "Success count exceeds trial count in BinomialProbitDataImputer::impute.");
  }
  // refresh_xtx (BinomialProbitSpikeSlabSampler.cpp:71-77): X'NX, built on the
  // matrix cores from the rows scaled by sqrt(n_i).  Exact for Bernoulli data; for
  // trial counts that are not perfect squares sqrt(n_i)^2 differs from n_i by one
  // rounding, i.e. an element differs from the reference's sum_i n_i x x' by no more
  // than the two summation orders already differ (~1e-16 relative per term).  The
  // binomial cases of tests/test_probit_gpu.py (1 - 8 and 1 - 12 trials) hold the
  // inclusion indicators bit-exact against the oracle on this matrix.
  std::vector<double> Xs((size_t)n * p), zero((size_t)n, 0.0);
  for (int32_t j = 0; j < p; ++j)
    for (int64_t i = 0; i < n; ++i) Xs[(size_t)j * n + i] = X[(size_t)j * n + i] * std::sqrt(ntrials[i]);
  int rc = ba_build_suf_from_xy(e, n, p, Xs.data(), zero.data());
  if (rc) return rc;
  HIP_TRY(e->dprob_X.resize((size_t)n * p));
  HIP_TRY(e->dprob_y.resize((size_t)n));
  HIP_TRY(e->dprob_nt.resize((size_t)n));
  HIP_TRY(hipMemcpy(e->dprob_X.ptr, X, (size_t)n * p * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dprob_y.ptr, y, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dprob_nt.ptr, ntrials, (size_t)n * 8, hipMemcpyHostToDevice));
  e->probit_mode = true;
  e->logit_mode = false;
  e->dprob_z.release();
  e->probit_n = n;
  e->probit_clt = clt_threshold;
  e->probit_sweep = 0;
  e->ss_mode = false;
  return BA_OK;
}

int ba_probit_sweep(ba_engine *e, int32_t nsweeps) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (nsweeps < 0) return fail(BA_E_INVALID, "nsweeps must be non-negative");
  if (!e->probit_mode) return fail(BA_E_STATE, "call ba_probit_set_data first");
  if (!e->have_slab) return fail(BA_E_STATE, "call ba_sss_set_slab first");
  if (e->sss_slab_scales) return fail(BA_E_INVALID, "the probit sampler takes a fixed-precision slab (scales_with_sigsq = 0)");
  int rc = alloc_chain_state(e);
  if (rc) return rc;
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p, n = (size_t)e->probit_n;
  if (e->dprob_z.count != C * n) {
    HIP_TRY(e->dprob_z.resize(C * n));
    HIP_TRY(e->dxty_c.resize(C * p));
    HIP_TRY(e->dlogit_planes.resize((size_t)xtwx_cols_planes((int64_t)n) * C * p));   // (split-K planes of X'z)
  }
  {
    // the latent data have unit variance: sigma^2 = 1 in every chain, whatever a
    // caller left there before the binomial data were set
    std::vector<double> one(C, 1.0);
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->dsigsq.ptr, one.data(), C * 8, hipMemcpyHostToDevice));
  }
  rc = switch_mode(e, 1, 1.0);
  if (rc) return rc;
  rc = upload_shared(e);
  if (rc) return rc;
  HIP_TRY(e->dmodel.resize(2 * C * ssvs_scalar_layout(64).total));
  SsvsParams P;
  fill_params(e, P);
  ProbitParams Q;
  std::memset(&Q, 0, sizeof(Q));
  Q.n = (int32_t)n;
  Q.p = (int32_t)p;
  Q.chains = (int32_t)C;
  Q.clt_threshold = e->probit_clt;
  Q.slot_limit = e->slot_limit;
  Q.chain_offset = e->cfg.chain_offset;
  Q.X = e->dprob_X.ptr;
  Q.y = e->dprob_y.ptr;
  Q.ntrials = e->dprob_nt.ptr;
  Q.gamma = e->dgamma.ptr;
  Q.beta = e->dbeta.ptr;
  Q.z = e->dprob_z.ptr;
  Q.xtz = e->dxty_c.ptr;
  Q.seed_lo = (uint32_t)e->seed;
  Q.seed_hi = (uint32_t)(e->seed >> 32);
  Q.status = e->dstatus.ptr;
  // BinomialProbitSpikeSlabSampler::draw (BinomialProbitSpikeSlabSampler.cpp:40-46)
  for (int i = 0; i < nsweeps; ++i) {
    Q.sweep = e->probit_sweep++;
    HIP_TRY(launch_probit_impute(e->stream, Q, e->dlogit_planes.ptr));   // impute_latent_data, X'z
    HIP_TRY(launch_sweeps(e, P, 1));               // draw_model_indicators, draw_beta
    // (a chain that outgrew the launch's capacity replays THIS sweep's draws on
    // this sweep's latent data before the next imputation)
    HIP_TRY(hipStreamSynchronize(e->stream));
    rc = check_chain_status(e);
    if (rc) return rc;
    fill_params(e, P);
    P.model_keep = 1;
    e->model_ok = true;
  }
  return BA_OK;
}

// ------------------------ BinomialLogitSpikeSlabSampler (auxiliary-mixture augmentation)
int ba_logit_set_data(ba_engine *e, int64_t n, int32_t p, const double *X, const double *y,
                      const double *ntrials, int32_t clt_threshold) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (!X || !y || !ntrials) return fail(BA_E_INVALID, "null argument");
  if (n <= 0 || p <= 0) return fail(BA_E_INVALID, "n and p must be positive");
  // (per-trial imputation reads two uniforms per trial of the observation's substream
  // of LOGIT_STRIDE; the large-sample branch beyond the threshold a few dozen)
  if (clt_threshold < 1 || 4 * clt_threshold > LOGIT_STRIDE)
    return fail(BA_E_INVALID, "clt_threshold must be between 1 and 64");
  for (int64_t i = 0; i < n; ++i) {
    if (y[i] < 0 || ntrials[i] < 0)
      return fail(BA_E_INVALID, "The number of successes and the number of trials must both be non-negative in BinomialLogitPartialAugmentationDataImputer::impute().");
    if (y[i] > ntrials[i])
      return fail(BA_E_INVALID, "The number of successes must not exceed the number of trials in BinomialLogitPartialAugmentationDataImputer::impute().");
  }
  // (dimensions, the shared buffers and a placeholder X'X; the sweeps use every
  // chain's own X'WX)
  std::vector<double> zero((size_t)n, 0.0);
  int rc = ba_build_suf_from_xy(e, n, p, X, zero.data());
  if (rc) return rc;
  HIP_TRY(e->dprob_X.resize((size_t)n * p));
  HIP_TRY(e->dprob_y.resize((size_t)n));
  HIP_TRY(e->dprob_nt.resize((size_t)n));
  HIP_TRY(hipMemcpy(e->dprob_X.ptr, X, (size_t)n * p * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dprob_y.ptr, y, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dprob_nt.ptr, ntrials, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(e->dlogit_Xsq.resize((size_t)n * p));
  HIP_TRY(launch_square(e->stream, e->dprob_X.ptr, (size_t)n * p, e->dlogit_Xsq.ptr));
  e->logit_mode = true;
  e->poisson_mode = false;
  e->probit_mode = false;
  e->probit_n = n;
  e->probit_clt = clt_threshold;
  e->probit_sweep = 0;
  e->ss_mode = false;
  e->dprob_z.release();
  return BA_OK;
}

// ------------------------ PoissonRegressionSpikeSlabSampler
int ba_poisson_set_data(ba_engine *e, int64_t n, int32_t p, const double *X, const double *y,
                        const double *exposure) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (!X || !y || !exposure) return fail(BA_E_INVALID, "null argument");
  if (n <= 0 || p <= 0) return fail(BA_E_INVALID, "n and p must be positive");
  for (int64_t i = 0; i < n; ++i) {
    if (y[i] < 0 || y[i] != std::floor(y[i])) return fail(BA_E_INVALID, "counts must be non-negative integers");
    if (!(exposure[i] > 0)) return fail(BA_E_INVALID, "exposures must be positive");
  }
  std::vector<double> zero((size_t)n, 0.0);
  int rc = ba_build_suf_from_xy(e, n, p, X, zero.data());   // (dimensions and the shared buffers)
  if (rc) return rc;
  HIP_TRY(e->dprob_X.resize((size_t)n * p));
  HIP_TRY(e->dprob_y.resize((size_t)n));
  HIP_TRY(e->dprob_nt.resize((size_t)n));
  HIP_TRY(hipMemcpy(e->dprob_X.ptr, X, (size_t)n * p * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dprob_y.ptr, y, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dprob_nt.ptr, exposure, (size_t)n * 8, hipMemcpyHostToDevice));
  HIP_TRY(e->dlogit_Xsq.resize((size_t)n * p));
  HIP_TRY(launch_square(e->stream, e->dprob_X.ptr, (size_t)n * p, e->dlogit_Xsq.ptr));
  e->poisson_y.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) e->poisson_y[(size_t)i] = (int64_t)std::llround(y[i]);
  e->logit_mode = true;      // (the logit path's buffers and column service)
  e->poisson_mode = true;
  e->poisson_mix_set = false;
  e->probit_mode = false;
  e->probit_n = n;
  e->probit_clt = 0;
  e->probit_sweep = 0;
  e->ss_mode = false;
  e->dprob_z.release();
  return BA_OK;
}

int ba_poisson_set_mixtures(ba_engine *e, int32_t ncounts, const int64_t *counts, const int32_t *ncomp,
                            const double *mu, const double *sigma, const double *weight,
                            int64_t largest_index) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (!e->poisson_mode) return fail(BA_E_STATE, "call ba_poisson_set_data first");
  if (ncounts <= 0 || !counts || !ncomp || !mu || !sigma || !weight) return fail(BA_E_INVALID, "null argument");
  std::vector<int32_t> off((size_t)ncounts + 1, 0);
  for (int i = 0; i < ncounts; ++i) {
    if (i > 0 && counts[i] <= counts[i - 1]) return fail(BA_E_INVALID, "counts must be ascending and distinct");
    if (ncomp[i] <= 0 || ncomp[i] > POISSON_MAX_COMP) return fail(BA_E_INVALID, "a mixture has 1 .. 32 components");
    off[(size_t)i + 1] = off[(size_t)i] + ncomp[i];
  }
  const size_t tot = (size_t)off[(size_t)ncounts];
  std::vector<double> logw(tot);
  for (size_t c = 0; c < tot; ++c) {
    if (!(weight[c] > 0) || !(sigma[c] > 0)) return fail(BA_E_INVALID, "mixture weights and standard deviations must be positive");
    logw[c] = std::log(weight[c]);
  }
  auto find = [&](int64_t v) -> int {
    const int64_t *it = std::lower_bound(counts, counts + ncounts, v);
    return (it != counts + ncounts && *it == v) ? (int)(it - counts) : -2;
  };
  const size_t n = e->poisson_y.size();
  std::vector<int32_t> obs(n, -1);
  for (size_t i = 0; i < n; ++i) {
    const int64_t v = e->poisson_y[i];
    if (v <= 0) continue;
    if (v >= largest_index) { obs[i] = -1; continue; }   // the Gaussian limit (poisson_mixture_approximation_table.cpp:49-55)
    const int m = find(v);
    if (m < 0) return fail(BA_E_INVALID, "no mixture was given for a count that occurs in the data");
    obs[i] = m;
  }
  const int one = find(1);
  if (one < 0) return fail(BA_E_INVALID, "the mixture of count 1 (the event past the interval) is needed");
  HIP_TRY(e->dpois_off.resize(off.size()));
  HIP_TRY(e->dpois_mu.resize(tot));
  HIP_TRY(e->dpois_sigma.resize(tot));
  HIP_TRY(e->dpois_logw.resize(tot));
  HIP_TRY(e->dpois_obs.resize(n));
  HIP_TRY(hipMemcpy(e->dpois_off.ptr, off.data(), off.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dpois_mu.ptr, mu, tot * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dpois_sigma.ptr, sigma, tot * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dpois_logw.ptr, logw.data(), tot * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dpois_obs.ptr, obs.data(), n * 4, hipMemcpyHostToDevice));
  e->poisson_mix_one = one;
  e->poisson_mix_set = true;
  return BA_OK;
}

}  // extern "C"
static int logit_family_sweep(ba_engine *e, int32_t nsweeps);
extern "C" {

int ba_logit_set_imputer(ba_engine *e, int32_t kind) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  if (kind != 0 && kind != 1) return fail(BA_E_INVALID, "imputer must be 0 (auxiliary mixture) or 1 (Polya-Gamma)");
  MUTATE(e);
  e->logit_imputer = kind;
  return BA_OK;
}

int ba_logit_sweep(ba_engine *e, int32_t nsweeps) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (nsweeps < 0) return fail(BA_E_INVALID, "nsweeps must be non-negative");
  if (e->poisson_mode) return fail(BA_E_STATE, "Poisson data are set: use ba_poisson_sweep");
  if (!e->logit_mode) return fail(BA_E_STATE, "call ba_logit_set_data first");
  return logit_family_sweep(e, nsweeps);
}

int ba_poisson_sweep(ba_engine *e, int32_t nsweeps) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (nsweeps < 0) return fail(BA_E_INVALID, "nsweeps must be non-negative");
  if (!e->poisson_mode) return fail(BA_E_STATE, "call ba_poisson_set_data first");
  if (!e->poisson_mix_set) return fail(BA_E_STATE, "call ba_poisson_set_mixtures first");
  return logit_family_sweep(e, nsweeps);
}

}  // extern "C"

// the sweep loop shared by the logit and the Poisson samplers: imputation (per family),
// X'Wz and the diagonal, the vectors of V the sweep starts from, the inclusion /
// coefficient draws with park-and-replay for vectors requested mid-sweep
static int logit_family_sweep(ba_engine *e, int32_t nsweeps) {
  {
  if (!e->have_slab) return fail(BA_E_STATE, "call ba_sss_set_slab first");
  if (e->sss_slab_scales) return fail(BA_E_INVALID, "the logit sampler takes a fixed-precision slab (scales_with_sigsq = 0)");
  int rc = alloc_chain_state(e);
  if (rc) return rc;
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p, n = (size_t)e->probit_n;
  if (e->dprob_z.count != C * n || e->dlogit_V.count != C * p * p) {
    HIP_TRY(e->dprob_z.resize(C * n));
    HIP_TRY(e->dlogit_w.resize(C * n));
    HIP_TRY(e->dlogit_V.resize(C * p * p));
    HIP_TRY(e->dxty_c.resize(C * p));
    e->logit_words = (int)((p + 31) / 32);
    HIP_TRY(e->dlogit_vdiag.resize(C * p));
    HIP_TRY(e->dlogit_valid.resize(C * (size_t)e->logit_words));
    HIP_TRY(e->dlogit_req.resize(2 * C * p));
    HIP_TRY(e->dlogit_cnt.resize(1));
    HIP_TRY(e->dcol_request.resize(C));
    // the planes of one GEMM launch: at most 1 GiB, at least one request tile
    const size_t per_req = (size_t)xtwx_cols_planes((int64_t)n) * p * 8;
    e->logit_req_batch = (int64_t)std::min<size_t>(std::max<size_t>(((size_t)1 << 30) / per_req, 64), 32768);
    e->logit_req_batch = std::min<int64_t>(e->logit_req_batch, (int64_t)(C * p));
    HIP_TRY(e->dlogit_planes.resize((size_t)e->logit_req_batch * per_req / 8));
  }
  {
    std::vector<double> one(C, 1.0);   // (sigma^2 = 1: see ba_probit_sweep)
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->dsigsq.ptr, one.data(), C * 8, hipMemcpyHostToDevice));
  }
  rc = switch_mode(e, 1, 1.0);
  if (rc) return rc;
  rc = upload_shared(e);
  if (rc) return rc;
  HIP_TRY(e->dmodel.resize(2 * C * ssvs_scalar_layout(64).total));
  SsvsParams P;
  fill_params(e, P);
  ProbitParams Q;
  std::memset(&Q, 0, sizeof(Q));
  Q.n = (int32_t)n;
  Q.p = (int32_t)p;
  Q.chains = (int32_t)C;
  Q.clt_threshold = e->probit_clt;
  Q.slot_limit = e->slot_limit;
  Q.chain_offset = e->cfg.chain_offset;
  Q.X = e->dprob_X.ptr;
  Q.y = e->dprob_y.ptr;
  Q.ntrials = e->dprob_nt.ptr;
  Q.gamma = e->dgamma.ptr;
  Q.beta = e->dbeta.ptr;
  Q.z = e->dprob_z.ptr;
  Q.w = e->dlogit_w.ptr;
  Q.xtz = e->dxty_c.ptr;
  Q.seed_lo = (uint32_t)e->seed;
  Q.seed_hi = (uint32_t)(e->seed >> 32);
  Q.status = e->dstatus.ptr;
  Q.mix_off = e->dpois_off.ptr;
  Q.mix_mu = e->dpois_mu.ptr;
  Q.mix_sigma = e->dpois_sigma.ptr;
  Q.mix_logw = e->dpois_logw.ptr;
  Q.obs_mix = e->dpois_obs.ptr;
  Q.mix_one = e->poisson_mix_one;
  const int imputer = e->poisson_mode ? 2 : e->logit_imputer;
  // BinomialLogitSpikeSlabSampler::draw (BinomialLogitSpikeSlabSampler.cpp:50-54) /
  // PoissonRegressionSpikeSlabSampler::draw (PoissonRegressionSpikeSlabSampler.cpp:55-59)
  for (int i = 0; i < nsweeps; ++i) {
    Q.sweep = e->probit_sweep++;
    // impute_latent_data: z, w, X'Wz and the diagonal of V = slab precision + X'WX ...
    HIP_TRY(launch_logit_impute(e->stream, Q, e->dlogit_Xsq.ptr, e->dA.ptr, e->dlogit_vdiag.ptr,
                                e->dlogit_planes.ptr, imputer));
    // ... and the vectors of V the sweep starts from: those of the included variables
    HIP_TRY(launch_xtwx_cols_start(e->stream, e->dgamma.ptr, (int)C, (int)p, e->dlogit_req.ptr,
                                   e->dlogit_cnt.ptr, e->dlogit_valid.ptr, e->logit_words));
    int32_t R = 0;
    HIP_TRY(hipMemcpyAsync(&R, e->dlogit_cnt.ptr, 4, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    rc = build_columns(e, R);
    if (rc) return rc;
    e->logit_cols_built += R;
    HIP_TRY(launch_sweeps(e, P, 1));                                         // draw_model_indicators, draw_beta
    // (a chain that stopped for a missing vector of V, or outgrew the launch's
    // capacity, replays THIS sweep's draws on this sweep's latent data before the
    // next imputation: check_chain_status serves both)
    HIP_TRY(hipStreamSynchronize(e->stream));
    rc = check_chain_status(e);
    if (rc) return rc;
    fill_params(e, P);
  }
  e->table_ok = false;
  e->model_ok = false;
  return BA_OK;
  }
}

extern "C" {

// ------------------------ AdaptiveSpikeSlabRegressionSampler (birth / death)
static int ada_prepare(ba_engine *e) {
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p;
  if (e->dada_birth.count == C * p) return BA_OK;
  HIP_TRY(e->dada_birth.resize(C * p));
  HIP_TRY(e->dada_death.resize(C * p));
  HIP_TRY(e->dada_iter.resize(C));
  HIP_TRY(e->dada_ws.resize(C * 4 * p));   // (the large-model kernel's: cumulative rates, the sweep's undo copy)
  HIP_TRY(e->dpos_ada.resize(C));
  std::vector<double> ones(C * p, 1.0);   // birth_rates_, death_rates_ start at 1
  HIP_TRY(hipMemcpyAsync(e->dada_birth.ptr, ones.data(), C * p * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->dada_death.ptr, ones.data(), C * p * 8, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemsetAsync(e->dada_iter.ptr, 0, C * 8, e->stream));
  HIP_TRY(hipMemsetAsync(e->dpos_ada.ptr, 0, C * 8, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return BA_OK;
}

int ba_adaptive_set_options(ba_engine *e, int32_t max_flips, double step_size,
                            double target_acceptance_rate) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  if (step_size == 0 || step_size < -1) return fail(BA_E_INVALID, "Step size must be positive.");
  if (target_acceptance_rate == 0 || target_acceptance_rate >= 1 || target_acceptance_rate < -1)
    return fail(BA_E_INVALID, "Target acceptance rate must be strictly between 0 and 1.");
  MUTATE(e);
  if (max_flips >= 0) e->ada_max_flips = max_flips;
  if (step_size > 0) e->ada_step = step_size;
  if (target_acceptance_rate > 0) e->ada_target = target_acceptance_rate;
  return BA_OK;
}

int ba_adaptive_sweep(ba_engine *e, int32_t nsweeps) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (nsweeps < 0) return fail(BA_E_INVALID, "nsweeps must be non-negative");
  if (e->ss_mode) return fail(BA_E_STATE, "state-space data are set: use ba_ss_sweep");
  if (e->logit_mode || e->probit_mode)
    return fail(BA_E_STATE, "binomial data are set: use ba_logit_sweep / ba_probit_sweep (the regression sampler has no meaning on latent data)");
  int rc = alloc_chain_state(e);
  if (rc) return rc;
  rc = switch_mode(e, 2, 1.0);
  if (rc) return rc;
  rc = upload_shared(e);
  if (rc) return rc;
  rc = ada_prepare(e);
  if (rc) return rc;
  if (e->trace_stride > 0 && nsweeps > e->trace_stride)
    return fail(BA_E_INVALID, "nsweeps exceeds the enabled trace length");
  HIP_TRY(e->dmodel.resize(2 * (size_t)e->cfg.chains * ssvs_scalar_layout(64).total));
  SsvsParams P;
  fill_params(e, P);
  if (ssvs_ada_lds_layout(e->p, e->kcap).total > e->lds_per_cu)
    return fail(BA_E_INVALID, "problem does not fit the LDS working set of the adaptive kernel");
  if (e->trace_stride > 0)
    HIP_TRY(hipMemsetAsync(e->dtrace_idx.ptr, 0, (size_t)e->cfg.chains * 4, e->stream));
  HIP_TRY(launch_sweeps(e, P, (int)nsweeps));
  return BA_OK;
}

int ba_adaptive_get_rates(ba_engine *e, int64_t chain, double *birth_rates,
                          double *death_rates, uint64_t *iteration_count) {
  ENGINE_PROLOGUE(e);
  if (chain < 0 || chain >= e->cfg.chains) return fail(BA_E_INVALID, "chain index out of range");
  if (e->dada_birth.count == 0) return fail(BA_E_STATE, "no adaptive sweep yet");
  int rc = ba_sync(e);
  if (rc) return rc;
  const size_t p = (size_t)e->p;
  if (birth_rates) HIP_TRY(hipMemcpy(birth_rates, e->dada_birth.ptr + (size_t)chain * p, p * 8, hipMemcpyDeviceToHost));
  if (death_rates) HIP_TRY(hipMemcpy(death_rates, e->dada_death.ptr + (size_t)chain * p, p * 8, hipMemcpyDeviceToHost));
  if (iteration_count) HIP_TRY(hipMemcpy(iteration_count, e->dada_iter.ptr + chain, 8, hipMemcpyDeviceToHost));
  return BA_OK;
}

// --------------------------------------------------- state space (kalman)
static int ss_prepare(ba_engine *e) {
  if (!e->ss_mode) return fail(BA_E_STATE, "call ba_ss_set_data first");
  if (!e->ss_level_set && !e->ssm_set)
    return fail(BA_E_STATE, "call ba_ss_set_local_level or ba_ss_set_structural first");
  int rc = upload_shared(e);
  if (rc) return rc;
  rc = alloc_chain_state(e);
  if (rc) return rc;
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p, T = ss_pitch(*e);
  if (e->dss_scratch.count != C * SS_SCRATCH_ARRAYS * T) {
    HIP_TRY(e->dss_scratch.resize(C * SS_SCRATCH_ARRAYS * T));
    HIP_TRY(e->dxty_c.resize(C * p));
    HIP_TRY(e->dyty_c.resize(C));
    HIP_TRY(e->dnobs_c.resize(C));
    HIP_TRY(e->dlev_sigsq.resize(C));
    HIP_TRY(e->dlev_n.resize(C));
    HIP_TRY(e->dlev_sumsq.resize(C));
    HIP_TRY(e->dpos_level.resize(C));
    HIP_TRY(e->dpos_state.resize(C));
    HIP_TRY(e->dpos_forecast.resize(C));
    HIP_TRY(hipMemsetAsync(e->dpos_forecast.ptr, 0, C * 8, e->stream));
    HIP_TRY(e->dprep_n.resize(2 * C));
    HIP_TRY(e->dprep_pos_state.resize(2 * C));
    HIP_TRY(e->dprep_pos_level.resize(2 * C));
    HIP_TRY(e->dprep_level.resize(2 * C));
    HIP_TRY(hipMemsetAsync(e->dprep_n.ptr, 0, 2 * C * 4, e->stream));
    e->ss_zbuf = 0;
    HIP_TRY(e->dxte_planes.resize((size_t)xte_planes((int64_t)T) * C * p));
    // regression suf starts as the data's own (before the first impute_state)
    std::vector<double> xty(C * p), yty(C, e->yty), nobs(C, e->n),
        lev(C, e->ss_initial_level_sigsq);
    for (size_t c = 0; c < C; ++c) std::memcpy(&xty[c * p], e->xty.data(), p * 8);
    hipStream_t s = e->stream;
    HIP_TRY(hipMemcpyAsync(e->dxty_c.ptr, xty.data(), xty.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->dyty_c.ptr, yty.data(), C * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->dnobs_c.ptr, nobs.data(), C * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(e->dlev_sigsq.ptr, lev.data(), C * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(e->dlev_n.ptr, 0, C * 8, s));
    HIP_TRY(hipMemsetAsync(e->dlev_sumsq.ptr, 0, C * 8, s));
    HIP_TRY(hipMemsetAsync(e->dpos_level.ptr, 0, C * 8, s));
    HIP_TRY(hipMemsetAsync(e->dpos_state.ptr, 0, C * 8, s));
    HIP_TRY(hipMemsetAsync(e->dss_scratch.ptr, 0, C * SS_SCRATCH_ARRAYS * T * 8, s));
    if (e->ssm_set) {
      const size_t NV = SSG_MAX_VAR;
      HIP_TRY(e->dssm_sigsq.resize(C * NV));
      HIP_TRY(e->dssm_n.resize(C * NV));
      HIP_TRY(e->dssm_ss.resize(C * NV));
      HIP_TRY(e->dpos_var.resize(C * NV));
      HIP_TRY(e->dssm_work.resize(C * (size_t)ssm_work_stride(*e)));
      HIP_TRY(e->dssg_spec.resize(sizeof(SsgSpec)));
      HIP_TRY(hipMemcpy(e->dssg_spec.ptr, &e->ssg, sizeof(SsgSpec), hipMemcpyHostToDevice));
      std::vector<double> v0(C * NV);
      for (size_t c = 0; c < C; ++c)
        for (size_t i = 0; i < NV; ++i) v0[c * NV + i] = e->ssg_initial_sigsq[i];
      HIP_TRY(hipMemcpy(e->dssm_sigsq.ptr, v0.data(), C * NV * 8, hipMemcpyHostToDevice));
      HIP_TRY(hipMemsetAsync(e->dssm_n.ptr, 0, C * NV * 8, s));
      HIP_TRY(hipMemsetAsync(e->dssm_ss.ptr, 0, C * NV * 8, s));
      HIP_TRY(hipMemsetAsync(e->dpos_var.ptr, 0, C * NV * 8, s));
      HIP_TRY(hipMemsetAsync(e->dssm_work.ptr, 0, C * (size_t)ssm_work_stride(*e) * 8, s));
      if (e->ssg.nar > 0) {
        HIP_TRY(e->dar_phi.resize(C * SSG_MAX_AR * AR_MAX));
        HIP_TRY(e->dar_suf.resize(C * SSG_MAX_AR * AR_SUF_STRIDE));
        std::vector<double> ph(C * SSG_MAX_AR * AR_MAX, 0.0);
        for (size_t c = 0; c < C; ++c)
          for (int a = 0; a < SSG_MAX_AR; ++a)
            for (int i = 0; i < AR_MAX; ++i) ph[(c * SSG_MAX_AR + a) * AR_MAX + i] = e->ssg_initial_phi[a][i];
        HIP_TRY(hipMemcpy(e->dar_phi.ptr, ph.data(), ph.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemsetAsync(e->dar_suf.ptr, 0, C * SSG_MAX_AR * AR_SUF_STRIDE * 8, s));
      }
    }
    HIP_TRY(hipStreamSynchronize(s));  // (the host vectors above go out of scope)
    e->ss_initialized = false;
  }
  HIP_TRY(e->dmodel.resize(2 * (size_t)e->cfg.chains * ssvs_scalar_layout(64).total));
  return BA_OK;
}

int ba_ss_set_data(ba_engine *e, int32_t T, int32_t p, const double *y,
                   const double *X, const uint8_t *observed) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  if (!y || !X) return fail(BA_E_INVALID, "null argument");
  if (T <= 0 || p <= 0) return fail(BA_E_INVALID, "T and p must be positive");
  // The regression model's fixed XtX (and the initial Xty, ...) are over the
  // OBSERVED rows only: missing points never update the sufficient statistics
  // (StateSpaceRegressionModel.cpp:100-125, SufstatDataPolicy.hpp:166-167).
  std::vector<double> Xo((size_t)T * p), yo(T);
  std::vector<uint8_t> obs(T, 1);
  double nobs = 0;
  for (int t = 0; t < T; ++t) {
    if (observed) obs[t] = observed[t] ? 1 : 0;
    nobs += obs[t];
    yo[t] = obs[t] ? y[t] : 0.0;
  }
  for (int j = 0; j < p; ++j)
    for (int t = 0; t < T; ++t)
      Xo[(size_t)j * T + t] = obs[t] ? X[(size_t)j * T + t] : 0.0;
  int rc = ba_build_suf_from_xy(e, T, p, Xo.data(), yo.data());
  if (rc) return rc;
  e->n = nobs;
  e->T = T;
  HIP_TRY(e->dss_y.resize(T));
  HIP_TRY(e->dss_X.resize((size_t)T * p));
  HIP_TRY(e->dss_obs.resize(T));
  HIP_TRY(hipMemcpy(e->dss_y.ptr, y, (size_t)T * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dss_X.ptr, X, (size_t)T * p * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->dss_obs.ptr, obs.data(), T, hipMemcpyHostToDevice));
  if (T <= LM_TP) {
    std::vector<double> Xt((size_t)LM_TP * p, 0.0), yt(LM_TP, 0.0);
    std::vector<uint32_t> mask(LM_THREADS, 0u);
    for (int t = 0; t < T; ++t) {
      yt[lm_at(t)] = y[t];
      if (obs[t]) mask[t / LM_BS] |= 1u << (t % LM_BS);
    }
    for (int j = 0; j < p; ++j)
      for (int t = 0; t < T; ++t) Xt[(size_t)j * LM_TP + lm_at(t)] = X[(size_t)j * T + t];
    HIP_TRY(e->dss_yt.resize(LM_TP));
    HIP_TRY(e->dss_Xt.resize((size_t)LM_TP * p));
    HIP_TRY(e->dss_obs_mask.resize(LM_THREADS));
    HIP_TRY(hipMemcpy(e->dss_yt.ptr, yt.data(), (size_t)LM_TP * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->dss_Xt.ptr, Xt.data(), (size_t)LM_TP * p * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->dss_obs_mask.ptr, mask.data(), LM_THREADS * 4, hipMemcpyHostToDevice));
  } else {
    e->dss_yt.release();
    e->dss_Xt.release();
    e->dss_obs_mask.release();
  }
  e->ss_mode = true;
  e->ss_initialized = false;
  e->dss_scratch.release();
  e->device_dirty = true;
  return BA_OK;
}

int ba_ss_set_local_level(ba_engine *e, double level_df, double level_sigma_guess,
                          double level_sigma_upper_limit,
                          double initial_state_mean,
                          double initial_state_variance,
                          double initial_level_sigma) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  if (level_sigma_upper_limit < 0 || initial_state_variance < 0)
    return fail(BA_E_INVALID, "sigma_max must be non-negative.");
  // ChisqModel(df, sigma_guess): 2 alpha = df, 2 beta = df sigma^2
  e->level_prior_df = 2 * (level_df / 2.0);
  e->level_prior_ss = 2 * (level_df * level_sigma_guess * level_sigma_guess / 2.0);
  e->level_sigma_max = level_sigma_upper_limit;
  e->ss_a0 = initial_state_mean;
  e->ss_P0 = initial_state_variance;
  e->ss_initial_level_sigsq = initial_level_sigma * initial_level_sigma;
  e->ss_level_set = true;
  e->ssm_set = false;
  e->dss_scratch.release();
  return BA_OK;
}

}  // extern "C"

namespace {
// the scalars of the specification that follow from the block list
void ssg_finish(SsgSpec &q) {
  // P's leading dimension: odd (a lane per column and a lane per row both conflict-free), and
  // one of the four values ssg_simsmooth_kernel is compiled for
  q.ld = q.m <= 16 ? 17 : (q.m <= 32 ? 33 : (q.m <= 60 ? 61 : 65));
  // state-error rows: one per variance slot, except that a trig block's every component has one
  q.nerr = 0;
  for (int i = 0; i < q.nblocks; ++i) {
    q.blk[i].err0 = q.nerr;
    q.nerr += q.blk[i].kind == SSG_TRIG ? q.blk[i].dim
              : ((q.blk[i].kind == SSG_LOCAL_LINEAR_TREND || q.blk[i].kind == SSG_SEMILOCAL) ? 2 : 1);
  }
  // steps per block of the passes: the most that leaves FOUR workgroups to a CU's 160 KB of
  // LDS (all 1024 chains of a launch resident; at m = 59 sixteen steps left room for three,
  // and the launch ran as two rounds), never below 8; see ssg_pass_lds_doubles
  q.bl = 64;
  while (q.bl > 8 && (2 * q.bl * q.m + q.m * q.ld + q.bl * (q.nerr + 1) + SSG_MAX_STATE + 8 +
                      q.nar * AR_MAX * (AR_MAX + 1)) * 8 > 39 * 1024)
    q.bl /= 2;
}
// ArModel's constructor: "Attempt to initialize ArModel with an illegal value of the
// autoregression coefficients." (the quick bound, then the step-down recursion)
bool ar_stationary_host(const double *phi, int lags) {
  double a[AR_MAX], b[AR_MAX], sum = 0;
  for (int i = 0; i < lags; ++i) { a[i] = phi[i]; sum += std::fabs(a[i]); }
  if (sum < 1) return true;
  for (int k = lags; k >= 1; --k) {
    const double r = a[k - 1];
    if (!(std::fabs(r) < 1)) return false;
    for (int j = 0; j + 1 < k; ++j) b[j] = (a[j] + r * a[k - 2 - j]) / (1 - r * r);
    for (int j = 0; j + 1 < k; ++j) a[j] = b[j];
  }
  return true;
}
// the Philox sampler id of variance parameter v of the block about to be appended: level 1,
// slope 6, seasonal 7, autoregression 12 for the first block of its family (local level
// and local linear trend are one family), + 16 for every earlier block of the family
int ssg_stream_id(const SsgSpec &q, int kind, int v) {
  // (a semilocal trend's level variance is of the level family; its NonzeroMeanAr1Sampler a
  // family of its own, id 14)
  if (kind == SSG_SEMILOCAL && v == 1) {
    int occ = 0;
    for (int i = 0; i < q.nblocks; ++i) occ += q.blk[i].kind == SSG_SEMILOCAL;
    return 14 + 16 * occ;
  }
  auto family = [](int k) { return (k == SSG_LOCAL_LINEAR_TREND || k == SSG_SEMILOCAL) ? (int)SSG_LOCAL_LEVEL : k; };
  const int fam = family(kind);
  int occ = 0;
  for (int i = 0; i < q.nblocks; ++i) {
    const int k = q.blk[i].kind;
    if (q.blk[i].nvar == 0) continue;   // (a static intercept has no sampler: it is in no family)
    if (family(k) == fam) ++occ;
  }
  const int base = kind == SSG_SEASONAL ? 7 : (kind == SSG_AR ? 12 : (kind == SSG_TRIG ? 13 : (v == 0 ? 1 : 6)));
  return base + 16 * occ;
}
// model->add_state(...) on the engine's copy of the specification
int ssg_add(ba_engine *e, int32_t kind, const int32_t *iparams, const double *var_df,
            const double *var_sigma_guess, const double *var_sigma_upper_limit,
            const double *var_initial_sigma, const double *initial_phi,
            const double *initial_state_mean, const double *initial_state_variance) {
  SsgSpec &q = e->ssg;
  const bool is_static = kind == SSG_STATIC_INTERCEPT;   // (no parameter: the var_* arrays are not read)
  if ((!is_static && (!var_df || !var_sigma_guess || !var_sigma_upper_limit || !var_initial_sigma)) ||
      !initial_state_mean || !initial_state_variance)
    return fail(BA_E_INVALID, "null argument");
  if (q.nblocks >= SSG_MAX_BLOCKS) return fail(BA_E_INVALID, "more than 8 state models");
  SsgBlock k{};
  k.kind = kind;
  k.nvar = 1;
  k.duration = 1;
  k.ar_index = -1;
  switch (kind) {
    case SSG_LOCAL_LEVEL: k.dim = 1; break;
    case SSG_LOCAL_LINEAR_TREND: k.dim = 2; k.nvar = 2; break;
    case SSG_SEASONAL: {
      if (!iparams) return fail(BA_E_INVALID, "null argument");
      // SeasonalStateModelBase: "'nseasons' must be positive"; one season has no state
      if (iparams[0] < 2) return fail(BA_E_INVALID, "nseasons must be at least 2");
      if (iparams[1] < 1) return fail(BA_E_INVALID, "season_duration must be positive");
      // (the kernels keep a block's duration and its running phase in 16-bit fields)
      if (iparams[1] > 65535) return fail(BA_E_INVALID, "season_duration exceeds 65535");
      k.nseasons = iparams[0];
      k.duration = iparams[1];
      // new_season(t): (t - time_of_first_observation) is a multiple of the duration
      k.phase = ((iparams[2] % k.duration) + k.duration) % k.duration;
      k.dim = k.nseasons - 1;
      break;
    }
    case SSG_STATIC_INTERCEPT:
      // StaticInterceptStateModel: T = 1, RQR = 0, nothing to learn and no sampler -- a local
      // level whose variance (a slot of its own, never drawn) is 0
      k.kind = SSG_LOCAL_LEVEL;
      k.dim = 1;
      k.nvar = 0;
      break;
    case SSG_TRIG:
      // TrigStateModel: "At least one frequency needed ..."; the rotations as the transition
      // matrix holds them: (cos, sin) per frequency in initial_phi
      if (!iparams || !initial_phi) return fail(BA_E_INVALID, "null argument");
      if (iparams[0] < 1) return fail(BA_E_INVALID, "At least one frequency needed to initialize TrigStateModel.");
      if (2 * iparams[0] > SSG_MAX_STATE) return fail(BA_E_INVALID, "state dimension exceeds 64");
      k.nfreq = iparams[0];
      k.dim = 2 * k.nfreq;
      break;
    case SSG_SEMILOCAL:
      // SemilocalLinearTrendStateModel(level, slope): iparams = {force_stationary,
      // force_ar1_positive}; initial_phi = {slope mean prior mu, sigma, AR(1) coefficient prior mu,
      // sigma, initial mu, initial phi}
      if (!iparams || !initial_phi) return fail(BA_E_INVALID, "null argument");
      if (iparams[1] && !iparams[0])
        return fail(BA_E_INVALID, "force_ar1_positive without force_stationary (a one-sided truncation of the slope's "
                                  "AR(1) coefficient) is not built");
      if (!(initial_phi[1] > 0) || !(initial_phi[3] > 0)) return fail(BA_E_INVALID, "the slope's prior standard deviations must be positive");
      if (q.nar >= SSG_MAX_AR) return fail(BA_E_INVALID, "more than 4 autoregression / semilocal state models");
      k.dim = 3;
      k.nvar = 2;
      k.ar_index = q.nar;
      k.sl_truncate = iparams[0] != 0;
      k.sl_positive = iparams[1] != 0;
      break;
    case SSG_AR:
      if (!iparams) return fail(BA_E_INVALID, "null argument");
      if (iparams[0] < 1) return fail(BA_E_INVALID, "lags must be positive");
      if (iparams[0] > AR_MAX) return fail(BA_E_INVALID, "more than 16 lags");
      if (q.nar >= SSG_MAX_AR) return fail(BA_E_INVALID, "more than 4 autoregression state models");
      k.lags = iparams[0];
      k.dim = k.lags;
      k.ar_index = q.nar;
      break;
    default:
      return fail(BA_E_INVALID, "state model kind must be 1 (local level), 2 (local linear trend), 3 (seasonal), 4 (autoregression), 5 (static intercept), 6 (trig) or 7 (semilocal linear trend)");
  }
  const int nslot = is_static ? 1 : k.nvar;   // variance slots the block takes
  if (q.m + k.dim > SSG_MAX_STATE) return fail(BA_E_INVALID, "state dimension exceeds 64");
  if (q.nvar + nslot > SSG_MAX_VAR) return fail(BA_E_INVALID, "more than 16 variance parameters");
  for (int v = 0; v < k.nvar; ++v) {
    if (var_sigma_upper_limit[v] < 0) return fail(BA_E_INVALID, "sigma_max must be non-negative.");
    if ((kind == SSG_AR || kind == SSG_SEMILOCAL) && !(var_initial_sigma[v] > 0))
      return fail(BA_E_INVALID, "initial sigma must be positive");
  }
  for (int i = 0; i < k.dim; ++i) {
    // (a multivariate initial state goes through a Cholesky factor in the reference: it
    // needs a positive variance; the local level model alone does not)
    const bool ok = initial_state_variance[i] > 0.0 ||
                    ((kind == SSG_LOCAL_LEVEL || is_static) && initial_state_variance[i] == 0.0) ||
                    (kind == SSG_SEMILOCAL && i == 2);   // (the slope's long-run mean: a parameter, variance 0 whatever is passed)
    if (!ok) return fail(BA_E_INVALID, "initial state variances must be positive");
  }
  if (kind == SSG_AR && initial_phi && !ar_stationary_host(initial_phi, k.lags))
    return fail(BA_E_INVALID, "the initial autoregression coefficients are not stationary");
  k.first = q.m;
  k.var0 = q.nvar;
  for (int v = 0; v < k.nvar; ++v) {
    const int vi = k.var0 + v;
    // ChisqModel(df, sigma_guess): 2 alpha = df, 2 beta = df sigma^2
    q.prior_df[vi] = 2 * (var_df[v] / 2.0);
    q.prior_ss[vi] = 2 * (var_df[v] * var_sigma_guess[v] * var_sigma_guess[v] / 2.0);
    q.sigma_max[vi] = var_sigma_upper_limit[v];
    e->ssg_initial_sigsq[vi] = var_initial_sigma[v] * var_initial_sigma[v];
    k.sid[v] = ssg_stream_id(q, kind, v);
  }
  if (is_static) {
    // (the slot: variance 0, a sampler that is never run)
    q.prior_df[k.var0] = 0.0;
    q.prior_ss[k.var0] = 0.0;
    q.sigma_max[k.var0] = std::numeric_limits<double>::infinity();
    e->ssg_initial_sigsq[k.var0] = 0.0;
  }
  for (int i = 0; i < k.dim; ++i) {
    q.a0[k.first + i] = initial_state_mean[i];
    q.P0[k.first + i] = initial_state_variance[i];
    q.trig_c[k.first + i] = kind == SSG_TRIG ? initial_phi[2 * (i / 2)] : 0.0;
    q.trig_s[k.first + i] = kind == SSG_TRIG ? initial_phi[2 * (i / 2) + 1] : 0.0;
  }
  if (kind == SSG_AR) {
    for (int i = 0; i < AR_MAX; ++i)
      e->ssg_initial_phi[k.ar_index][i] = (initial_phi && i < k.lags) ? initial_phi[i] : 0.0;
    q.nar += 1;
  }
  if (kind == SSG_SEMILOCAL) {
    for (int i = 0; i < AR_MAX; ++i) e->ssg_initial_phi[k.ar_index][i] = 0.0;
    e->ssg_initial_phi[k.ar_index][0] = initial_phi[5];   // phi
    e->ssg_initial_phi[k.ar_index][1] = initial_phi[4];   // mu
    for (int i = 0; i < 4; ++i) q.sl_prior[k.ar_index][i] = initial_phi[i];
    // initial_state_mean()[2] = slope->mu() (per chain, per draw: the kernel's), variance 0
    q.a0[k.first + 2] = initial_phi[4];
    q.P0[k.first + 2] = 0.0;
    q.nar += 1;
  }
  q.blk[q.nblocks] = k;
  q.nblocks += 1;
  q.m += k.dim;
  q.nvar += nslot;
  ssg_finish(q);
  e->ssm_set = true;
  e->ss_level_set = false;
  e->dss_scratch.release();
  return BA_OK;
}
void ssg_clear(ba_engine *e) {
  e->ssg = SsgSpec{};
  for (int i = 0; i < 3; ++i) e->ssg_template_var[i] = -1;
  e->ssg_template_ar = -1;
  e->ssm_set = false;
  e->dss_scratch.release();
}
}  // namespace

extern "C" {

// for the tests of the spill streams (device_rng.h): a slot of a substream hands out `uniforms`
// numbers, not its whole stride (0: the default again).  Changes the draws.
int ba_set_slot_limit(ba_engine *e, int32_t uniforms) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  if (uniforms < 0 || (uniforms & 1)) return fail(BA_E_INVALID, "uniforms must be even and non-negative");
  MUTATE(e);
  e->slot_limit = uniforms;
  return BA_OK;
}

// diagnostic, changes no draw: 0 = the general kernel also where the shape-specialised one
// applies (the two are compared by the tests), 1 = the default
int ba_ss_set_tuning(ba_engine *e, int32_t kernel) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  if (kernel < 0 || kernel > 7 || kernel == 2)
    return fail(BA_E_INVALID, "kernel must be 0, 1, 3, 4, 5, 6 or 7 (2, four chains per wavefront, was removed: it never won)");
  MUTATE(e);
  // 4 / 5: the local-level rounds as the separate launches of rounds 1-4 / as the round
  // kernel (the default where it applies); the structural kernels' choice stays
  // 6 / 7: the round kernel's diagnostic notes (printed when a chain stops) on / off
  if (kernel >= 6) e->round_debug = kernel == 6;
  else if (kernel >= 4) e->ss_round_enabled = kernel == 5;
  else e->ssg_kernel_choice = kernel;
  return BA_OK;
}

int ba_ss_clear_state_models(ba_engine *e) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  ssg_clear(e);
  return BA_OK;
}

int ba_ss_add_state_model(ba_engine *e, int32_t kind, const int32_t *iparams, const double *var_df,
                          const double *var_sigma_guess, const double *var_sigma_upper_limit,
                          const double *var_initial_sigma, const double *initial_phi,
                          const double *initial_state_mean, const double *initial_state_variance) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  if (e->ss_level_set) ssg_clear(e);   // (a local-level specification is replaced, not extended)
  e->ss_level_set = false;
  return ssg_add(e, kind, iparams, var_df, var_sigma_guess, var_sigma_upper_limit, var_initial_sigma,
                 initial_phi, initial_state_mean, initial_state_variance);
}

int ba_ss_state_dimension(ba_engine *e, int32_t *state_dimension, int32_t *nblocks) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  if (state_dimension) *state_dimension = e->ssm_set ? e->ssg.m : (e->ss_level_set ? 1 : 0);
  if (nblocks) *nblocks = e->ssm_set ? e->ssg.nblocks : (e->ss_level_set ? 1 : 0);
  return BA_OK;
}

int ba_ss_set_structural(ba_engine *e, int32_t trend, int32_t nseasons, const double *var_df,
                         const double *var_sigma_guess, const double *var_sigma_upper_limit,
                         const double *var_initial_sigma, const double *initial_state_mean,
                         const double *initial_state_variance) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  if (!var_df || !var_sigma_guess || !var_sigma_upper_limit || !var_initial_sigma ||
      !initial_state_mean || !initial_state_variance)
    return fail(BA_E_INVALID, "null argument");
  if (trend != 1 && trend != 2) return fail(BA_E_INVALID, "trend must be 1 (local level) or 2 (local linear trend)");
  if (nseasons != 0 && nseasons < 2) return fail(BA_E_INVALID, "nseasons must be 0 or at least 2");
  const int m = trend + (nseasons > 0 ? nseasons - 1 : 0);
  if (m > SSG_MAX_STATE) return fail(BA_E_INVALID, "state dimension exceeds 64");
  for (int i = 0; i < 3; ++i) {
    const bool used = i == 0 || (i == 1 && trend == 2) || (i == 2 && nseasons > 0);
    if (used && var_sigma_upper_limit[i] < 0) return fail(BA_E_INVALID, "sigma_max must be non-negative.");
  }
  ssg_clear(e);
  int rc = ssg_add(e, trend == 2 ? SSG_LOCAL_LINEAR_TREND : SSG_LOCAL_LEVEL, nullptr, var_df, var_sigma_guess,
                   var_sigma_upper_limit, var_initial_sigma, nullptr, initial_state_mean, initial_state_variance);
  if (!rc && nseasons > 0) {
    const int32_t ip[3] = {nseasons, 1, 0};
    rc = ssg_add(e, SSG_SEASONAL, ip, var_df + 2, var_sigma_guess + 2, var_sigma_upper_limit + 2,
                 var_initial_sigma + 2, nullptr, initial_state_mean + trend, initial_state_variance + trend);
  }
  if (rc) {
    ssg_clear(e);
    return rc;
  }
  e->ssg_template_var[0] = 0;
  e->ssg_template_var[1] = trend == 2 ? 1 : -1;
  e->ssg_template_var[2] = nseasons > 0 ? trend : -1;
  return BA_OK;
}

int ba_ss_add_ar(ba_engine *e, int32_t lags, double prior_df, double sigma_guess,
                 double sigma_upper_limit, double initial_sigma, const double *initial_phi,
                 const double *initial_state_mean, const double *initial_state_variance) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  MUTATE(e);
  if (!e->ssm_set) return fail(BA_E_STATE, "call ba_ss_set_structural first");
  if (e->ssg_template_ar >= 0) return fail(BA_E_STATE, "the state already has an autoregression block");
  if (!initial_state_mean || !initial_state_variance) return fail(BA_E_INVALID, "null argument");
  if (lags < 1) return fail(BA_E_INVALID, "lags must be positive");
  const int32_t ip[3] = {lags, 0, 0};
  const int rc = ssg_add(e, SSG_AR, ip, &prior_df, &sigma_guess, &sigma_upper_limit, &initial_sigma, initial_phi,
                         initial_state_mean, initial_state_variance);
  if (rc) return rc;
  e->ssg_template_ar = e->ssg.nblocks - 1;
  return BA_OK;
}

// block `block` of one chain: its variance parameters, the state model's sufficient
// statistics of the last sweep, and for an autoregression block its coefficients and ArModel
// sufficient statistics
int ba_ss_get_state_model(ba_engine *e, int64_t chain, int32_t block, double *variances, double *suf_n,
                          double *suf_ss, double *phi, double *ar_xtx, double *ar_xty, double *ar_yty,
                          double *ar_n) {
  ENGINE_ACCESSOR_SERVED(e);
  if (!e->ss_mode || !e->ssm_set || e->dssm_work.count == 0)
    return fail(BA_E_STATE, "no structural state-space run yet");
  if (chain < 0 || chain >= e->cfg.chains) return fail(BA_E_INVALID, "chain index out of range");
  if (block < 0 || block >= e->ssg.nblocks) return fail(BA_E_INVALID, "state model index out of range");
  const SsgBlock &k = e->ssg.blk[block];
  const bool is_sl = k.kind == SSG_SEMILOCAL;
  const bool is_ar = k.kind == SSG_AR || is_sl;   // (both keep coefficients and statistics in an autoregression slot)
  if (!is_ar && (phi || ar_xtx || ar_xty || ar_yty || ar_n))
    return fail(BA_E_INVALID, "not an autoregression state model");
  if (is_sl && (ar_xty || ar_yty))
    return fail(BA_E_INVALID, "a semilocal linear trend's Ar1Suf comes back through ar_xtx (six doubles) and ar_n");
  if (ss_la_serving(e)) {
    if (!suf_n && !suf_ss && !ar_xtx && !ar_xty && !ar_yty && !ar_n) {
      // the draw ba_ss_draw_next is serving, from the record
      const ba_engine::SsLa::Rows *r = nullptr;
      int rcr = ss_la_rows(e, chain, false, &r);
      if (rcr) return rcr;
      const size_t row = (size_t)e->ssla.served - 1;
      if (variances)
        for (int v = 0; v < k.nvar; ++v) variances[v] = r->var[row * e->ssla.nvar + k.var0 + v];
      if (phi)
        for (int i = 0; i < (is_sl ? 2 : k.lags); ++i) phi[i] = r->phi[row * e->ssla.nphi + (size_t)k.ar_index * AR_MAX + i];
      return BA_OK;
    }
    int rcs = ss_la_settle(e);   // (sufficient statistics are not in the record)
    if (rcs) return rcs;
  }
  int rc = ba_sync(e);
  if (rc) return rc;
  HIP_TRY(pinned_reserve(e, (6 + AR_MAX + AR_SUF_STRIDE) * 8));
  double *hv = (double *)e->pinned, *hphi = hv + 6, *hsuf = hphi + AR_MAX;
  const size_t at = (size_t)chain * SSG_MAX_VAR + k.var0;
  HIP_TRY(hipMemcpyAsync(hv, e->dssm_sigsq.ptr + at, (size_t)k.nvar * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(hv + 2, e->dssm_n.ptr + at, (size_t)k.nvar * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(hv + 4, e->dssm_ss.ptr + at, (size_t)k.nvar * 8, hipMemcpyDeviceToHost, e->stream));
  if (is_ar) {
    const size_t slot = (size_t)chain * SSG_MAX_AR + k.ar_index;
    HIP_TRY(hipMemcpyAsync(hphi, e->dar_phi.ptr + slot * AR_MAX, AR_MAX * 8, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(hsuf, e->dar_suf.ptr + slot * AR_SUF_STRIDE, AR_SUF_STRIDE * 8, hipMemcpyDeviceToHost,
                           e->stream));
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int v = 0; v < k.nvar; ++v) {
    if (variances) variances[v] = hv[v];
    if (suf_n) suf_n[v] = hv[2 + v];
    if (suf_ss) suf_ss[v] = hv[4 + v];
  }
  if (k.kind == SSG_SEMILOCAL) {
    // (phi, mu) of the slope's NonzeroMeanAr1Model; its Ar1Suf -- sumsq, sum, cross, n, first,
    // last value -- through ar_xtx (six doubles)
    if (phi) { phi[0] = hphi[0]; phi[1] = hphi[1]; }
    if (ar_xtx) std::memcpy(ar_xtx, hsuf, 6 * 8);
    if (ar_n) *ar_n = hsuf[3];
  } else if (is_ar) {
    const int L = k.lags;
    if (phi) std::memcpy(phi, hphi, (size_t)L * 8);
    if (ar_xtx)
      for (int i = 0; i < L; ++i)
        for (int j = 0; j < L; ++j) ar_xtx[(size_t)j * L + i] = hsuf[(size_t)i * AR_MAX + j];
    if (ar_xty) std::memcpy(ar_xty, hsuf + AR_SUF_XTY, (size_t)L * 8);
    if (ar_yty) *ar_yty = hsuf[AR_SUF_YTY];
    if (ar_n) *ar_n = hsuf[AR_SUF_N];
  }
  return BA_OK;
}

int ba_ss_get_ar(ba_engine *e, int64_t chain, double *phi, double *sigsq, double *suf_xtx,
                 double *suf_xty, double *suf_yty, double *suf_n) {
  if (!e) return fail(BA_E_INVALID, "null engine");
  if (!e->ss_mode || !e->ssm_set || e->ssg_template_ar < 0 || e->dar_phi.count == 0)
    return fail(BA_E_STATE, "no structural run with an autoregression block yet");
  return ba_ss_get_state_model(e, chain, e->ssg_template_ar, sigsq, nullptr, nullptr, phi, suf_xtx, suf_xty,
                               suf_yty, suf_n);
}

// one chain's state draw, T x m (step t at [t * m, (t + 1) * m))
int ba_ss_get_state_draw(ba_engine *e, int64_t chain, double *state) {
  ENGINE_ACCESSOR_SERVED(e);
  if (!e->ss_mode || !e->ssm_set || e->dssm_work.count == 0)
    return fail(BA_E_STATE, "no structural state-space run yet");
  if (chain < 0 || chain >= e->cfg.chains) return fail(BA_E_INVALID, "chain index out of range");
  if (!state) return fail(BA_E_INVALID, "null argument");
  if (ss_la_serving(e)) {
    if (ss_la_registered(e, chain)) {
      const ba_engine::SsLa::Rows *r = nullptr;
      int rcr = ss_la_rows(e, chain, true, &r);
      if (rcr) return rcr;
      const size_t SD = e->ssla.state_doubles;
      std::memcpy(state, &r->state[((size_t)e->ssla.served - 1) * SD], SD * 8);
      return BA_OK;
    }
    ss_la_want_state(e, chain);
    int rcs = ss_la_settle(e);
    if (rcs) return rcs;
  }
  int rc = ba_sync(e);
  if (rc) return rc;
  const size_t T = (size_t)e->T, m = (size_t)e->ssg.m;
  HIP_TRY(pinned_reserve(e, m * T * 8));
  HIP_TRY(hipMemcpyAsync(e->pinned, e->dssm_work.ptr + (size_t)chain * ssm_work_stride(*e) + m * T, m * T * 8,
                         hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  std::memcpy(state, e->pinned, m * T * 8);
  return BA_OK;
}

int ba_ss_get_structural(ba_engine *e, int64_t chain, double *state, double *variances,
                         double *suf_n, double *suf_ss) {
  ENGINE_ACCESSOR_SERVED(e);
  if (!e->ss_mode || !e->ssm_set || e->dssm_work.count == 0)
    return fail(BA_E_STATE, "no structural state-space run yet");
  if (chain < 0 || chain >= e->cfg.chains) return fail(BA_E_INVALID, "chain index out of range");
  if ((variances || suf_n || suf_ss) && e->ssg_template_var[0] < 0)
    return fail(BA_E_STATE, "the state was not set with ba_ss_set_structural: use ba_ss_get_state_model");
  if (ss_la_serving(e)) {
    if (!suf_n && !suf_ss && (!state || ss_la_registered(e, chain))) {
      const ba_engine::SsLa::Rows *r = nullptr;
      int rcr = ss_la_rows(e, chain, state != nullptr, &r);
      if (rcr) return rcr;
      const size_t row = (size_t)e->ssla.served - 1, SD = e->ssla.state_doubles;
      if (state) std::memcpy(state, &r->state[row * SD], SD * 8);
      if (variances)
        for (int i = 0; i < 3; ++i) {
          const int vi = e->ssg_template_var[i];
          variances[i] = vi >= 0 ? r->var[row * e->ssla.nvar + vi] : 0.0;
        }
      return BA_OK;
    }
    if (state) ss_la_want_state(e, chain);
    int rcs = ss_la_settle(e);
    if (rcs) return rcs;
  }
  if (state) {
    const int rc = ba_ss_get_state_draw(e, chain, state);
    if (rc) return rc;
  }
  int rc = ba_sync(e);
  if (rc) return rc;
  // (one batch through the pinned staging buffer: variances | n | sums of squares)
  const size_t NV = SSG_MAX_VAR;
  HIP_TRY(pinned_reserve(e, 3 * NV * 8));
  double *hv = (double *)e->pinned;
  HIP_TRY(hipMemcpyAsync(hv, e->dssm_sigsq.ptr + chain * NV, NV * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(hv + NV, e->dssm_n.ptr + chain * NV, NV * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(hv + 2 * NV, e->dssm_ss.ptr + chain * NV, NV * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int i = 0; i < 3; ++i) {
    const int vi = e->ssg_template_var[i];
    // (an unused slot reports what it was given: the initial value, no statistics)
    if (variances) variances[i] = vi >= 0 ? hv[vi] : 0.0;
    if (suf_n) suf_n[i] = vi >= 0 ? hv[NV + vi] : 0.0;
    if (suf_ss) suf_ss[i] = vi >= 0 ? hv[2 * NV + vi] : 0.0;
  }
  return BA_OK;
}

int ba_ss_impute_state(ba_engine *e) {
  ENGINE_PROLOGUE(e);
  e->table_ok = false;
  int rc = ss_prepare(e);
  if (rc) return rc;
  SsParams S;
  fill_ss_params(e, S);
  HIP_TRY(launch_state_kernel(e, S, 0));
  e->ss_initialized = true;
  return BA_OK;
}

}  // extern "C"
namespace {
#ifdef BA_RSTAMPS
double *g_round_stamps = nullptr;
size_t g_round_stamps_n = 0;
#endif
// The round kernel (ss_round_kernel.hip) serves the local-level model on a series of at most
// LM_TP steps while every chain is in the LDS sweep kernel's range; how many chains one launch
// can take (0: the separate launches instead).
int ss_round_chains(ba_engine *e) {
  if (!e->ss_round_enabled || e->ssm_set || !ss_lane_major(*e) || e->big_active || e->cur_mode != 0) return 0;
  if (e->kcap != 16 && e->kcap != 32 && e->kcap != 48) return 0;
  int &res = e->ss_round_resident[e->kcap / 16];
  if (res == 0) {
    res = -1;
    if (ss_round_lds(e->p, e->kcap) <= (size_t)e->lds_per_cu / 2) {   // (two chains to a CU at least)
      SsvsParams P{};
      SsParams S{};
      SsRoundParams F{};
      P.kcap = e->kcap;
      P.p = e->p;
      int n = 0;
      if (launch_ss_round(e->stream, P, S, F, &n) == hipSuccess && n > 0) res = n;
      if (e->round_debug) std::fprintf(stderr, "round kernel: kcap %d lds %zu resident %d\n", e->kcap, ss_round_lds(e->p, e->kcap), n);
    }
  }
  return res > 0 ? std::min<int>(res, e->cfg.chains) : 0;
}

// `rounds` rounds of every chain: one launch per group of co-resident chains and per
// SS_ROUND_MAX_ROUNDS rounds; round i of the call goes to row rec_first + i of the record
int ss_round_launches(ba_engine *e, SsvsParams &P, SsParams &S, int rounds, int rec_slot) {
  const int per = ss_round_chains(e), C = e->cfg.chains;
  const size_t ctl = (size_t)SS_ROUND_MAX_ROUNDS * (1 + 2 * (size_t)per);   // (ticket | sizes | diagnostic variants: a done count per tile)
  const size_t mem = (size_t)SS_ROUND_MAX_ROUNDS * ((size_t)per * SS_ROUND_TILE + 2 * SS_ROUND_TILE);
  if (e->dround_ctl.count != ctl) HIP_TRY(e->dround_ctl.resize(ctl));
  if (e->dround_members.count != mem) HIP_TRY(e->dround_members.resize(mem));
  SsRoundParams F{};
  F.close_ticks = 100000;   // 1 ms: a tile is short of members only when another launch shares the machine (ss_round_kernel.hip)
  F.ticket = e->dround_ctl.ptr;
  F.sizes = F.ticket + SS_ROUND_MAX_ROUNDS;
  F.members = e->dround_members.ptr;
  F.planes = e->dxte_planes.ptr;
  if (e->round_debug) {
    if (e->dround_debug.count == 0) {
      HIP_TRY(e->dround_debug.resize(16 * 17 + 64));
      HIP_TRY(hipMemset(e->dround_debug.ptr, 0, (16 * 17 + 64) * 4));
    }
    F.debug = e->dround_debug.ptr;
  }
#ifdef BA_RSTAMPS
  {  // (diagnostic build: printed per call by tools/ss_round_phases.py through BA_RSTAMPS_DUMP)
    static DevBuf<double> stamps;
    if (stamps.count != (size_t)C * 16) {
      HIP_TRY(stamps.resize((size_t)C * 16));
      HIP_TRY(hipMemsetAsync(stamps.ptr, 0, (size_t)C * 16 * 8, e->stream));
    }
    F.stamps = stamps.ptr;
    g_round_stamps = stamps.ptr;
    g_round_stamps_n = (size_t)C * 16;
  }
#endif
  if (rec_slot >= 0) {
    ba_engine::SsLa &A = e->ssla;
    F.rgamma = A.rgamma.ptr;
    F.rbeta = A.rbeta.ptr;
    F.rsig = A.rsig.ptr;
    F.rvar = A.rvar.ptr;
    F.rstate = A.rstate.ptr;
    F.reg_of_chain = e->dround_reg.ptr;
    F.rec_slot = rec_slot;
    F.rec_len = A.len;
    F.nreg = (int32_t)A.reg.size();
  }
  for (int g0 = 0; g0 < C; g0 += per) {
    const int gc = std::min(per, C - g0);
    SsvsParams Pg = P;
    SsParams Sg = S;
    Pg.chain_first = Sg.chain_first = g0;
    Pg.chain_count = Sg.chain_count = gc;
    for (int r0 = 0; r0 < rounds; r0 += SS_ROUND_MAX_ROUNDS) {
      F.rounds = std::min<int>(SS_ROUND_MAX_ROUNDS, rounds - r0);
      F.rec_first = r0;
#ifdef BA_RSTAMPS
      { static int seq = 0; F.debug_seq = ++seq; }
#endif
      HIP_TRY(hipMemsetAsync(e->dround_ctl.ptr, 0, ctl * 4, e->stream));
      HIP_TRY(hipMemsetAsync(e->dround_members.ptr, 0xff, mem * 4, e->stream));
      HIP_TRY(launch_ss_round(e->stream, Pg, Sg, F, nullptr));
      Pg.model_keep = 1;   // (from here on the chains' model blocks are their own last launch's)
    }
  }
  P.model_keep = 1;
  e->model_ok = true;
  return BA_OK;
}

// nsweeps x StateSpacePosteriorSampler::draw on every chain; rec_slot >= 0: every round's
// draw goes to that half of the look-ahead's record
int ss_sweep_impl(ba_engine *e, int32_t nsweeps, int rec_slot) {
  e->table_ok = false;
  if (nsweeps < 0) return fail(BA_E_INVALID, "nsweeps must be non-negative");
  int rc = ss_prepare(e);
  if (rc) return rc;
  SsvsParams P;
  fill_params(e, P);
  SsParams S;
  fill_ss_params(e, S);
  // StateSpacePosteriorSampler::draw (StateSpacePosteriorSampler.cpp:42-64)
  if (!e->ss_initialized) {
    HIP_TRY(launch_state_kernel(e, S, 0));
    e->ss_initialized = true;
  }
  // The local-level state draw in two pieces: what does not depend on the round's
  // regression sweep (level variance, the normals) is done ahead on a second stream into
  // the chains' other normals buffer -- the step for round i + 1 goes out behind round
  // i's state draw, beside its X'e GEMM, its plane sum and the start of round i + 1's
  // SSVS launch (kalman_prepare_kernel).
  if (nsweeps > 0 && ss_round_chains(e) > 0) return ss_round_launches(e, P, S, nsweeps, rec_slot);
  const bool ahead = !e->ssm_set && nsweeps > 0;
  if (ahead && !e->stream2) {
    {
      int rcs = concurrent_stream(e, &e->stream2);
      if (rcs) return rcs;
    }
    HIP_TRY(hipEventCreateWithFlags(&e->ev_state, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&e->ev_prep[i], hipEventDisableTiming));
  }
  auto prepare_ahead = [&](int zbuf) -> hipError_t {
    // (after everything enqueued on the main stream so far: the chains' status words of the
    // sweep just launched, the buffer's last reader)
    hipError_t err = hipEventRecord(e->ev_state, e->stream);
    if (err != hipSuccess) return err;
    err = hipStreamWaitEvent(e->stream2, e->ev_state, 0);
    if (err != hipSuccess) return err;
    SsParams A = S;
    A.zbuf = zbuf;
    err = launch_kalman_prepare(e->stream2, A, 1);
    if (err != hipSuccess) return err;
    return hipEventRecord(e->ev_prep[zbuf], e->stream2);
  };
  if (ahead) {
    HIP_TRY(prepare_ahead(e->ss_zbuf));   // the call's first round: nothing to run beside
    S.prepared = 1;
  }
  for (int i = 0; i < nsweeps; ++i) {
    HIP_TRY(launch_sweeps(e, P, 1));                  // observation model
    if (ahead) {
      const int cur = e->ss_zbuf;
      HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_prep[cur], 0));
      S.zbuf = cur;
      HIP_TRY(launch_kalman_main(e->stream, S, 1));   // state models, state
      // (the state draw's wavefronts fill the register files -- 2 x 256 registers to a
      // SIMD -- so a prepare step launched beside it only delays it: it goes out behind)
      if (i + 1 < nsweeps) HIP_TRY(prepare_ahead(cur ^ 1));
      // ... and the regression's X'e: inside a call the plane sum is left to the next
      // round's sweep launch (one wave per chain on this path)
      const bool fold = i + 1 < nsweeps && !e->big_active && e->waves == 1 && e->cur_mode != 2;
      HIP_TRY(launch_kalman_xte(e->stream, S, fold));
      P.xty_planes = fold ? e->dxte_planes.ptr : nullptr;
      P.xty_nplanes = xte_planes((int64_t)S.TP);
      P.xty_plane_stride = (int64_t)e->cfg.chains * e->p;
      e->ss_zbuf = cur ^ 1;
    } else {
      HIP_TRY(launch_state_kernel(e, S, 1));          // state models, state
    }
    if (rec_slot >= 0) HIP_TRY(ss_la_record(e, rec_slot, i));
    P.model_keep = 1;  // from here on the chains' model blocks are their own last launch's
    e->model_ok = true;
  }
  return BA_OK;
}

}  // namespace
extern "C" {

int ba_ss_sweep(ba_engine *e, int32_t nsweeps) {
  ENGINE_PROLOGUE(e);   // (unserved look-ahead draws: the rounds asked for here come after the last one served)
  return ss_sweep_impl(e, nsweeps, -1);
}

// The callers' loop on the bsts path -- for (i in niter) { model.sample_posterior(); record }
// (Interfaces/R/bsts/src/bsts.cc:82-119) -- at the device's rate: rounds are enqueued
// `lookahead` at a time, every round's draw recorded on the device, and ba_ss_draw_next
// hands them out one per call; the accessors below see the draw being served.
int ba_ss_set_lookahead(ba_engine *e, int32_t lookahead) {
  ENGINE_PROLOGUE(e);
  if (lookahead < 1) return fail(BA_E_INVALID, "lookahead must be at least 1");
  MUTATE(e);
  {
    // the record: two halves x chains x rounds x (gamma + beta [+ variances, coefficients]);
    // a look-ahead it has no room for (p = 4096 with 1024 chains: 75 MB a round) is cut
    // down to what 2 GiB hold rather than failing in hipMalloc
    const double per_round = 2.0 * (double)e->cfg.chains * ((double)std::max(e->p, 1) * 9.0 + 8.0 * (SSG_MAX_VAR + SSG_MAX_AR * AR_MAX + 1));
    const double budget = 2147483648.0;
    if ((double)lookahead * per_round > budget) lookahead = std::max<int32_t>(1, (int32_t)(budget / per_round));
  }
  e->ssla.len = lookahead;
  e->ssla.cur = 0;
  e->ssla.calm = 0;
  e->ssla.probe_wait = 16;
  ss_la_reset(e);
  return BA_OK;
}

// the chains whose STATE PATH the look-ahead records (default: chain 0); the other chains'
// state is read by going back to the draw being served (correct, and slow)
int ba_ss_lookahead_chains(ba_engine *e, int32_t nchains, const int64_t *chains) {
  ENGINE_PROLOGUE(e);
  if (nchains < 0 || (nchains > 0 && !chains)) return fail(BA_E_INVALID, "bad argument");
  for (int i = 0; i < nchains; ++i)
    if (chains[i] < 0 || chains[i] >= e->cfg.chains) return fail(BA_E_INVALID, "chain index out of range");
  MUTATE(e);
  e->ssla.reg.clear();
  e->ssla.want.clear();
  for (int i = 0; i < nchains; ++i) e->ssla.reg.push_back((int32_t)chains[i]);
  ss_la_reset(e);
  return BA_OK;
}

int ba_ss_draw_next(ba_engine *e) {
  ENGINE_PROLOGUE_NOJOIN(e);
  {
    int rcj = pipe_join(e);
    if (rcj) return rcj;
  }
  ba_engine::SsLa &A = e->ssla;
  if (A.len <= 1) return ss_sweep_impl(e, 1, -1);
  if (A.cur <= 1) {
    // (every draw of late was followed by something the record could not serve: one round
    // per call, and another try with a batch of two after a while)
    if (A.cur < 1) A.cur = A.len;   // (first call after ba_ss_set_lookahead)
    else {
      if (++A.calm >= A.probe_wait) { A.cur = 2; A.calm = 0; }
      return ss_sweep_impl(e, 1, -1);
    }
  }
  if (A.served == A.avail) {
    if (A.avail > 0 && A.clean) {   // a batch served to its end in peace
      A.cur = std::min(A.len, A.cur * 2);
      A.probe_wait = 16;
    }
    A.clean = true;
    if (A.ahead) {
      // the record is used up: on to the batch that is already running (or done)
      A.slot ^= 1;
      A.ahead = false;
      A.avail = A.ahead_len;
    } else {
      // ... or from the chains' current state
      int rc = ss_la_settle(e);
      if (!rc) rc = la_rewind(e);
      if (!rc) rc = ss_prepare(e);
      if (!rc) rc = ss_la_alloc(e);
      // (the first impute_state of a run, if it is still to come, is not part of a batch:
      // a batch's snapshot is a point between two rounds)
      if (!rc) rc = ss_sweep_impl(e, 0, -1);
      if (rc) return rc;
      A.slot = 0;
      rc = ss_la_launch(e, 0);
      if (rc) return rc;
      A.avail = A.cur;
    }
    A.served = 0;
    A.synced = false;
    A.cache.clear();
    // the batch after this one goes out now, into the other half
    A.ahead_len = A.cur;
    int rc = ss_la_launch(e, A.slot ^ 1);
    if (rc) return rc;
    A.ahead = true;
  }
  ++A.served;
  return BA_OK;
}

int ba_ss_forecast(ba_engine *e, int32_t horizon, const double *newX, double *out) {
  ENGINE_PROLOGUE(e);
  if (!newX || !out || horizon <= 0) return fail(BA_E_INVALID, "bad argument");
  if (!e->ss_mode || e->dss_scratch.count == 0 || !e->ss_initialized)
    return fail(BA_E_STATE, "no state draw yet: run ba_ss_sweep or ba_ss_impute_state first");
  int rc = ba_sync(e);
  if (rc) return rc;
  const size_t C = (size_t)e->cfg.chains, p = (size_t)e->p, h = (size_t)horizon;
  DevBuf<double> dnx, dout;
  HIP_TRY(dnx.resize(h * p));
  HIP_TRY(dout.resize(C * h));
  HIP_TRY(hipMemcpyAsync(dnx.ptr, newX, h * p * 8, hipMemcpyHostToDevice, e->stream));
  SsParams S;
  fill_ss_params(e, S);
  if (e->ssm_set)
    HIP_TRY(launch_ssm_forecast(e->stream, S, horizon, dnx.ptr, e->dpos_forecast.ptr, dout.ptr));
  else
    HIP_TRY(launch_ss_forecast(e->stream, S, horizon, dnx.ptr, e->dpos_forecast.ptr, dout.ptr));
  HIP_TRY(hipMemcpyAsync(out, dout.ptr, C * h * 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return BA_OK;
}

int ba_ss_get_state(ba_engine *e, int64_t chain, double *state,
                    double *level_sigsq, double *level_n, double *level_sumsq) {
  ENGINE_ACCESSOR_SERVED(e);
  if (!e->ss_mode || e->dss_scratch.count == 0) return fail(BA_E_STATE, "no state-space run yet");
  if (e->ssm_set) return fail(BA_E_STATE, "a structural state is set: use ba_ss_get_structural");
  if (chain < 0 || chain >= e->cfg.chains) return fail(BA_E_INVALID, "chain index out of range");
  if (ss_la_serving(e)) {
    if (!level_n && !level_sumsq && (!state || ss_la_registered(e, chain))) {
      // the draw ba_ss_draw_next is serving, from the record
      const ba_engine::SsLa::Rows *r = nullptr;
      int rcr = ss_la_rows(e, chain, state != nullptr, &r);
      if (rcr) return rcr;
      const size_t row = (size_t)e->ssla.served - 1, T = (size_t)e->T, SD = e->ssla.state_doubles;
      if (level_sigsq) *level_sigsq = r->var[row];
      if (state) {
        const double *src = &r->state[row * SD];
        if (ss_lane_major(*e)) {
          for (size_t t = 0; t < T; ++t) state[t] = src[(size_t)lm_at((int)t)];
        } else {
          std::memcpy(state, src, T * 8);
        }
      }
      return BA_OK;
    }
    if (state) ss_la_want_state(e, chain);
    int rcs = ss_la_settle(e);   // (not in the record: the chains go back to the draw being served)
    if (rcs) return rcs;
  }
  int rc = ba_sync(e);
  if (rc) return rc;
  // (one batch through the pinned staging buffer: state | level variance | n | sum of squares)
  const size_t T = (size_t)e->T, TP = ss_pitch(*e);
  HIP_TRY(pinned_reserve(e, (TP + 3) * 8));
  double *hstate = (double *)e->pinned, *hl = hstate + TP;
  if (state)
    HIP_TRY(hipMemcpyAsync(hstate, e->dss_scratch.ptr + ((size_t)chain * SS_SCRATCH_ARRAYS + SS_STATE_ARRAY) * TP,
                           (ss_lane_major(*e) ? TP : T) * 8, hipMemcpyDeviceToHost, e->stream));
  if (level_sigsq) HIP_TRY(hipMemcpyAsync(hl, e->dlev_sigsq.ptr + chain, 8, hipMemcpyDeviceToHost, e->stream));
  if (level_n) HIP_TRY(hipMemcpyAsync(hl + 1, e->dlev_n.ptr + chain, 8, hipMemcpyDeviceToHost, e->stream));
  if (level_sumsq) HIP_TRY(hipMemcpyAsync(hl + 2, e->dlev_sumsq.ptr + chain, 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (state) {
    if (ss_lane_major(*e)) {
      for (size_t t = 0; t < T; ++t) state[t] = hstate[(size_t)lm_at((int)t)];
    } else {
      std::memcpy(state, hstate, T * 8);
    }
  }
  if (level_sigsq) *level_sigsq = hl[0];
  if (level_n) *level_n = hl[1];
  if (level_sumsq) *level_sumsq = hl[2];
  return BA_OK;
}

int ba_ss_set_level_sigsq(ba_engine *e, int64_t chain, double sigsq) {
  ENGINE_PROLOGUE(e);
  MUTATE(e);
  int rc = ss_prepare(e);
  if (rc) return rc;
  const int64_t C = e->cfg.chains;
  if (chain < -1 || chain >= C) return fail(BA_E_INVALID, "chain index out of range");
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (chain < 0) {
    std::vector<double> v((size_t)C, sigsq);
    HIP_TRY(hipMemcpy(e->dlev_sigsq.ptr, v.data(), (size_t)C * 8, hipMemcpyHostToDevice));
  } else {
    HIP_TRY(hipMemcpy(e->dlev_sigsq.ptr + chain, &sigsq, 8, hipMemcpyHostToDevice));
  }
  return BA_OK;
}

int ba_ss_get_chain_suf(ba_engine *e, int64_t chain, double *xty, double *yty,
                        double *n) {
  ENGINE_PROLOGUE(e);
  if (!e->ss_mode || e->dxty_c.count == 0) return fail(BA_E_STATE, "no state-space run yet");
  if (chain < 0 || chain >= e->cfg.chains) return fail(BA_E_INVALID, "chain index out of range");
  int rc = ba_sync(e);
  if (rc) return rc;
  const size_t p = (size_t)e->p;
  if (xty) HIP_TRY(hipMemcpy(xty, e->dxty_c.ptr + (size_t)chain * p, p * 8, hipMemcpyDeviceToHost));
  if (yty) HIP_TRY(hipMemcpy(yty, e->dyty_c.ptr + chain, 8, hipMemcpyDeviceToHost));
  if (n) HIP_TRY(hipMemcpy(n, e->dnobs_c.ptr + chain, 8, hipMemcpyDeviceToHost));
  return BA_OK;
}

}  // extern "C"

#ifdef BA_RSTAMPS
// diagnostic build only (tools/build/libboomamd_rstamps.so): the round kernel's phase ticks
// since the last call, chains x 2 x 8, and reset
extern "C" int ba_debug_round_stamps(double *out, int64_t n) {
  using namespace boom_amd;
  if (!g_round_stamps) return -1;
  const size_t m = std::min<size_t>((size_t)n, g_round_stamps_n);
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpy(out, g_round_stamps, m * 8, hipMemcpyDeviceToHost) != hipSuccess) return -2;
  if (hipMemset(g_round_stamps, 0, g_round_stamps_n * 8) != hipSuccess) return -2;
  return (int)m;
}
#endif
