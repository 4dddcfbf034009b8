// Several engines behind one handle: the multi-GPU form of the C-ABI (include/boom_amd.h,
// ba_group_*).  ONE process, one engine -- one HIP device, one stream -- per entry of the
// device list; chains are sharded over the engines by global id (engine i owns
// [i * chains_per_device, (i + 1) * chains_per_device)), so a chain draws the same
// numbers wherever it runs.  Nothing on the sampling path crosses devices.  The two
// collectives of SURVEY 8(e) go through librccl directly:
//   ba_group_build_suf_from_xy   rows of X sharded over the devices, local f64-MFMA syrk,
//                                ONE ncclAllReduce of (X'X | X'y | y'y, sum y | sum x)
//   ba_group_get_summaries       ONE ncclAllGather of the (3p + 16)-double summary blocks
// both issued for all devices inside ncclGroupStart / ncclGroupEnd on the engines' own
// streams.  librccl is loaded when a group spans more than one device (dlopen: the
// single-device library has no link-time dependency on it, and a process that already
// carries PyTorch's copy shares that one).  A group whose engines all sit on ONE device
// -- how the bookkeeping is tested on a one-GPU box -- needs no collective: the blocks
// are summed / copied on the device.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/boom_amd.h"

namespace {

// the subset of the RCCL API used (rccl.h: ncclResult_t is an int, 0 = success;
// ncclFloat64 = 8, ncclSum = 0; communicators are opaque pointers)
typedef void *rccl_comm_t;
struct Rccl {
  void *lib = nullptr;
  int (*CommInitAll)(rccl_comm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(rccl_comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  bool load(std::string *err) {
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) { *err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define RCCL_SYM(field, sym)                                              \
    field = reinterpret_cast<decltype(field)>(dlsym(lib, sym));            \
    if (!field) { *err = std::string("librccl lacks ") + sym; return false; }
    RCCL_SYM(CommInitAll, "ncclCommInitAll")
    RCCL_SYM(CommDestroy, "ncclCommDestroy")
    RCCL_SYM(GroupStart, "ncclGroupStart")
    RCCL_SYM(GroupEnd, "ncclGroupEnd")
    RCCL_SYM(AllReduce, "ncclAllReduce")
    RCCL_SYM(AllGather, "ncclAllGather")
    RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    return true;
  }
};
enum { RCCL_FLOAT64 = 8, RCCL_SUM = 0 };

thread_local std::string g_group_error;
int gfail(int code, const std::string &msg) {
  g_group_error = msg;
  return code;
}

// out[i] += in[i]
__global__ void add_block_kernel(double *out, const double *in, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] += in[i];
}

}  // namespace

struct ba_group {
  std::vector<ba_engine *> eng;
  std::vector<int32_t> dev;
  int32_t chains_per_device = 0;
  bool one_device = false;        // every engine on the same device: no collective needed
  Rccl rccl;
  std::vector<rccl_comm_t> comms;
  std::vector<double *> dbuf;     // per engine: a device buffer for the blocks that travel
  size_t dbuf_doubles = 0;
  int32_t p = 0;
};

namespace {
#define G_HIP(expr)                                                                        \
  do {                                                                                     \
    hipError_t e__ = (expr);                                                               \
    if (e__ != hipSuccess) return gfail(BA_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)
#define G_RCCL(g, expr)                                                                    \
  do {                                                                                     \
    int r__ = (expr);                                                                      \
    if (r__ != 0) return gfail(BA_E_HIP, std::string(#expr) + ": " + (g)->rccl.GetErrorString(r__)); \
  } while (0)
#define G_BA(expr)                                                       \
  do {                                                                   \
    int r__ = (expr);                                                    \
    if (r__ != BA_OK) return gfail(r__, ba_last_error());                \
  } while (0)

int ensure_buffers(ba_group *g, size_t doubles) {
  if (doubles <= g->dbuf_doubles) return BA_OK;
  for (size_t i = 0; i < g->eng.size(); ++i) {
    G_HIP(hipSetDevice(g->dev[i]));
    if (g->dbuf[i]) G_HIP(hipFree(g->dbuf[i]));
    g->dbuf[i] = nullptr;
    G_HIP(hipMalloc((void **)&g->dbuf[i], doubles * sizeof(double)));
  }
  g->dbuf_doubles = doubles;
  return BA_OK;
}
}  // namespace

extern "C" {

const char *ba_group_last_error(void) { return g_group_error.c_str(); }

// 1 when librccl can be loaded and exports every entry point a multi-device group uses
int32_t ba_group_rccl_available(void) {
  Rccl r;
  std::string err;
  const bool ok = r.load(&err);
  if (r.lib) dlclose(r.lib);
  if (!ok) g_group_error = err;
  return ok ? 1 : 0;
}

int ba_group_create(const int32_t *devices, int32_t ndevices, int32_t chains_per_device, uint64_t seed,
                    ba_group **out) {
  if (!devices || !out || ndevices <= 0) return gfail(BA_E_INVALID, "bad device list");
  if (chains_per_device <= 0) return gfail(BA_E_INVALID, "chains_per_device must be positive");
  bool all_same = true, all_distinct = true;
  for (int i = 0; i < ndevices; ++i)
    for (int j = 0; j < i; ++j) {
      if (devices[i] == devices[j]) all_distinct = false;
      else all_same = false;
    }
  if (ndevices > 1 && !all_same && !all_distinct)
    return gfail(BA_E_INVALID, "the device list must name distinct devices (or one device throughout)");
  ba_group *g = new ba_group();
  g->chains_per_device = chains_per_device;
  g->one_device = (ndevices == 1) || all_same;
  g->dev.assign(devices, devices + ndevices);
  g->dbuf.assign(ndevices, nullptr);
  for (int i = 0; i < ndevices; ++i) {
    ba_config cfg{devices[i], chains_per_device, (int64_t)i * chains_per_device, seed, 0, 0};
    ba_engine *e = nullptr;
    const int rc = ba_engine_create(&cfg, &e);
    if (rc != BA_OK) {
      g_group_error = ba_last_error();
      ba_group_destroy(g);
      return rc;
    }
    g->eng.push_back(e);
  }
  if (!g->one_device) {
    std::string err;
    if (!g->rccl.load(&err)) {
      ba_group_destroy(g);
      return gfail(BA_E_HIP, err);
    }
    g->comms.assign(ndevices, nullptr);
    const int r = g->rccl.CommInitAll(g->comms.data(), ndevices, g->dev.data());
    if (r != 0) {
      const std::string msg = std::string("ncclCommInitAll: ") + g->rccl.GetErrorString(r);
      g->comms.clear();
      ba_group_destroy(g);
      return gfail(BA_E_HIP, msg);
    }
  }
  *out = g;
  return BA_OK;
}

void ba_group_destroy(ba_group *g) {
  if (!g) return;
  for (size_t i = 0; i < g->comms.size(); ++i)
    if (g->comms[i]) (void)g->rccl.CommDestroy(g->comms[i]);
  for (size_t i = 0; i < g->dbuf.size(); ++i)
    if (g->dbuf[i]) {
      (void)hipSetDevice(g->dev[i]);
      (void)hipFree(g->dbuf[i]);
    }
  for (ba_engine *e : g->eng) ba_engine_destroy(e);
  if (g->rccl.lib) dlclose(g->rccl.lib);
  delete g;
}

int32_t ba_group_size(const ba_group *g) { return g ? (int32_t)g->eng.size() : 0; }

ba_engine *ba_group_engine(ba_group *g, int32_t i) {
  return (g && i >= 0 && i < (int32_t)g->eng.size()) ? g->eng[i] : nullptr;
}

int ba_group_locate(const ba_group *g, int64_t global_chain, int32_t *engine_index, int64_t *local_chain) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  const int64_t total = (int64_t)g->eng.size() * g->chains_per_device;
  if (global_chain < 0 || global_chain >= total) return gfail(BA_E_INVALID, "chain index out of range");
  if (engine_index) *engine_index = (int32_t)(global_chain / g->chains_per_device);
  if (local_chain) *local_chain = global_chain % g->chains_per_device;
  return BA_OK;
}

// rows [n i / G, n (i + 1) / G) of the n-row design matrix go to engine i
int ba_group_build_suf_from_xy(ba_group *g, int64_t n, int32_t p, const double *X, const double *y) {
  if (!g || !X || !y) return gfail(BA_E_INVALID, "null argument");
  const int G = (int)g->eng.size();
  if (n < G || p <= 0) return gfail(BA_E_INVALID, "fewer rows than devices");
  const size_t blk = ba_suf_block_size(p);
  int rc = ensure_buffers(g, 2 * blk);   // [the block that is reduced | a staging block]
  if (rc) return rc;
  // ---- every device: its rows up, partial statistics into its block (asynchronous
  // per device: the uploads and syrks of different devices overlap)
  std::vector<double *> dX(G, nullptr), dy(G, nullptr);
  // (freed on every way out, the error returns of the macros below included)
  struct Shards {
    ba_group *g;
    std::vector<double *> &dX, &dy;
    ~Shards() {
      for (size_t i = 0; i < dX.size(); ++i) {
        if (!dX[i] && !dy[i]) continue;
        (void)hipSetDevice(g->dev[i]);
        if (dX[i]) (void)hipFree(dX[i]);
        if (dy[i]) (void)hipFree(dy[i]);
      }
    }
  } shards{g, dX, dy};
  std::vector<std::vector<double>> hX(G);
  for (int i = 0; i < G; ++i) {
    const int64_t lo = n * i / G, hi = n * (i + 1) / G, ni = hi - lo;
    hX[i].resize((size_t)ni * p);
    for (int32_t j = 0; j < p; ++j)   // column-major shard
      std::memcpy(&hX[i][(size_t)j * ni], X + (size_t)j * n + lo, (size_t)ni * sizeof(double));
    G_HIP(hipSetDevice(g->dev[i]));
    G_HIP(hipMalloc((void **)&dX[i], (size_t)ni * p * sizeof(double)));
    G_HIP(hipMalloc((void **)&dy[i], (size_t)ni * sizeof(double)));
    hipStream_t s = (hipStream_t)ba_stream(g->eng[i]);
    G_HIP(hipMemcpyAsync(dX[i], hX[i].data(), (size_t)ni * p * sizeof(double), hipMemcpyHostToDevice, s));
    G_HIP(hipMemcpyAsync(dy[i], y + lo, (size_t)ni * sizeof(double), hipMemcpyHostToDevice, s));
  }
  for (int i = 0; i < G; ++i) {
    const int64_t lo = n * i / G, hi = n * (i + 1) / G;
    G_BA(ba_suf_partial_device(g->eng[i], hi - lo, p, dX[i], dy[i], g->dbuf[i]));
  }
  // every engine's block is complete before any other stream (or device) reads it
  for (int i = 0; i < G; ++i) {
    G_HIP(hipSetDevice(g->dev[i]));
    G_HIP(hipStreamSynchronize((hipStream_t)ba_stream(g->eng[i])));
  }
  // ---- ONE all-reduce of the blocks
  if (g->one_device) {
    G_HIP(hipSetDevice(g->dev[0]));
    hipStream_t s = (hipStream_t)ba_stream(g->eng[0]);
    for (int i = 1; i < G; ++i)   // in rank order: the sum a ring of two would give; G > 2 on one device is a test set-up
      hipLaunchKernelGGL(add_block_kernel, dim3((unsigned)((blk + 255) / 256)), dim3(256), 0, s, g->dbuf[0],
                         g->dbuf[i], blk);
    G_HIP(hipGetLastError());
    G_HIP(hipStreamSynchronize(s));
    for (int i = 1; i < G; ++i) G_HIP(hipMemcpy(g->dbuf[i], g->dbuf[0], blk * sizeof(double), hipMemcpyDeviceToDevice));
  } else {
    G_RCCL(g, g->rccl.GroupStart());
    for (int i = 0; i < G; ++i)
      G_RCCL(g, g->rccl.AllReduce(g->dbuf[i], g->dbuf[i], blk, RCCL_FLOAT64, RCCL_SUM, g->comms[i],
                                  (hipStream_t)ba_stream(g->eng[i])));
    G_RCCL(g, g->rccl.GroupEnd());
    for (int i = 0; i < G; ++i) {
      G_HIP(hipSetDevice(g->dev[i]));
      G_HIP(hipStreamSynchronize((hipStream_t)ba_stream(g->eng[i])));
    }
  }
  for (int i = 0; i < G; ++i) G_BA(ba_set_suf_from_block_device(g->eng[i], n, p, g->dbuf[i]));
  g->p = p;
  return BA_OK;
}

int ba_group_set_priors(ba_group *g, const double *prior_mean, const double *unscaled_prior_precision,
                        const double *prior_inclusion_probabilities, int64_t max_model_size, double prior_df,
                        double sigma_guess, double sigma_upper_limit) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  for (ba_engine *e : g->eng) {
    G_BA(ba_set_slab(e, prior_mean, unscaled_prior_precision));
    G_BA(ba_set_spike(e, prior_inclusion_probabilities, max_model_size));
    G_BA(ba_set_sigma_prior(e, prior_df, sigma_guess, sigma_upper_limit));
  }
  return BA_OK;
}

int ba_group_set_state(ba_group *g, const uint8_t *gamma, const double *beta, double sigsq) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  for (ba_engine *e : g->eng) G_BA(ba_set_state(e, -1, gamma, beta, sigsq));
  return BA_OK;
}

int ba_group_sweep(ba_group *g, int32_t nsweeps) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  for (ba_engine *e : g->eng) G_BA(ba_sweep(e, nsweeps));   // asynchronous: the devices run side by side
  return BA_OK;
}

// fn(engine, arg) on every engine of the group, in order; stops at the first error (whose
// text ba_group_last_error() then holds).  Everything that is the SAME on every device --
// the bsts / logit / Poisson data (replicated: SURVEY 8e), priors, state models, options,
// the look-ahead -- is set this way with the single-engine entry points.
int ba_group_call(ba_group *g, int (*fn)(ba_engine *, void *), void *arg) {
  if (!g || !fn) return gfail(BA_E_INVALID, "null argument");
  for (ba_engine *e : g->eng) G_BA(fn(e, arg));
  return BA_OK;
}

// the samplers' rounds on every device: each call only ENQUEUES (the single-engine entry
// points return once their launches are out), so the devices run side by side
int ba_group_ss_sweep(ba_group *g, int32_t nsweeps) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  for (ba_engine *e : g->eng) G_BA(ba_ss_sweep(e, nsweeps));
  return BA_OK;
}
int ba_group_ss_draw_next(ba_group *g) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  for (ba_engine *e : g->eng) G_BA(ba_ss_draw_next(e));
  return BA_OK;
}
int ba_group_logit_sweep(ba_group *g, int32_t nsweeps) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  for (ba_engine *e : g->eng) G_BA(ba_logit_sweep(e, nsweeps));
  return BA_OK;
}
int ba_group_poisson_sweep(ba_group *g, int32_t nsweeps) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  for (ba_engine *e : g->eng) G_BA(ba_poisson_sweep(e, nsweeps));
  return BA_OK;
}

int ba_group_sync(ba_group *g) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  for (ba_engine *e : g->eng) G_BA(ba_sync(e));
  return BA_OK;
}

int ba_group_reset_summaries(ba_group *g) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  for (ba_engine *e : g->eng) G_BA(ba_reset_summaries(e));
  return BA_OK;
}

// whole-job posterior summaries: every engine reduces its chains on its device, ONE
// all-gather of the blocks, the sum over devices on the host
int ba_group_get_summaries(ba_group *g, double *inclusion_count, double *beta_sum, double *beta_sumsq,
                           double *scalars, double *blocks) {
  if (!g) return gfail(BA_E_INVALID, "null group");
  const int G = (int)g->eng.size();
  int32_t p = 0;
  G_BA(ba_engine_info(g->eng[0], nullptr, nullptr, &p));
  if (p <= 0) return gfail(BA_E_STATE, "no data set");
  const size_t blk = 3 * (size_t)p + 16;
  int rc = ensure_buffers(g, (size_t)(G + 1) * blk);   // [gathered blocks | this engine's own]
  if (rc) return rc;
  for (int i = 0; i < G; ++i) G_BA(ba_summaries_device(g->eng[i], g->dbuf[i] + (size_t)G * blk));
  if (g->one_device) {
    G_HIP(hipSetDevice(g->dev[0]));
    for (int i = 0; i < G; ++i) G_HIP(hipStreamSynchronize((hipStream_t)ba_stream(g->eng[i])));
    for (int i = 0; i < G; ++i)
      G_HIP(hipMemcpy(g->dbuf[0] + (size_t)i * blk, g->dbuf[i] + (size_t)G * blk, blk * sizeof(double),
                      hipMemcpyDeviceToDevice));
  } else {
    G_RCCL(g, g->rccl.GroupStart());
    for (int i = 0; i < G; ++i)
      G_RCCL(g, g->rccl.AllGather(g->dbuf[i] + (size_t)G * blk, g->dbuf[i], blk, RCCL_FLOAT64, g->comms[i],
                                  (hipStream_t)ba_stream(g->eng[i])));
    G_RCCL(g, g->rccl.GroupEnd());
    G_HIP(hipSetDevice(g->dev[0]));
    G_HIP(hipStreamSynchronize((hipStream_t)ba_stream(g->eng[0])));
  }
  std::vector<double> h((size_t)G * blk);
  G_HIP(hipSetDevice(g->dev[0]));
  G_HIP(hipMemcpy(h.data(), g->dbuf[0], h.size() * sizeof(double), hipMemcpyDeviceToHost));
  if (blocks) std::memcpy(blocks, h.data(), h.size() * sizeof(double));
  std::vector<double> tot(blk, 0.0);
  tot[3 * (size_t)p + 6] = std::numeric_limits<double>::infinity();   // smallest decision margin: a minimum
  for (int i = 0; i < G; ++i)
    for (size_t k = 0; k < blk; ++k) {
      const double v = h[(size_t)i * blk + k];
      if (k == 3 * (size_t)p + 6) tot[k] = std::fmin(tot[k], v);
      else tot[k] += v;
    }
  if (inclusion_count) std::memcpy(inclusion_count, &tot[0], (size_t)p * sizeof(double));
  if (beta_sum) std::memcpy(beta_sum, &tot[p], (size_t)p * sizeof(double));
  if (beta_sumsq) std::memcpy(beta_sumsq, &tot[2 * (size_t)p], (size_t)p * sizeof(double));
  if (scalars) std::memcpy(scalars, &tot[3 * (size_t)p], 16 * sizeof(double));
  return BA_OK;
}

}  // extern "C"
