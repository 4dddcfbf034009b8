// Per-kernel-class device time of an engine's launches (ba_set_kernel_timing /
// ba_get_kernel_times): with timing enabled a launcher brackets its kernel with a
// pair of HIP events ON THE STREAM THE KERNEL IS LAUNCHED ON; disabled (the default)
// a KtScope is two loads of a thread-local pointer.  Measurement only -- no draw
// depends on it.  bench.py uses it for the roofline objects of the configurations
// whose sweep round is several kernels.
#ifndef BOOM_AMD_KTIMER_H
#define BOOM_AMD_KTIMER_H

#include <hip/hip_runtime.h>

namespace boom_amd {

enum KernelClass {
  KT_SSVS = 0,      // ssvs_sweep_kernel
  KT_SSVS_BIG,      // ssvs_big_kernel
  KT_SSVS_ADAPTIVE, // ssvs_adaptive_kernel
  KT_KALMAN,        // kalman_simsmooth_kernel
  KT_SSM,           // ssm_simsmooth_kernel
  KT_XTE_GEMM,      // atb_mfma_kernel (X'e of every chain)
  KT_PROBIT_IMPUTE, // probit_impute_kernel
  KT_LOGIT_IMPUTE,  // logit_impute_kernel
  KT_ROWS_GEMM,     // xtwx_cols_kernel<false> + plain_reduce_kernel (X'z, diagonal)
  KT_COLS_GEMM,     // xtwx_cols_kernel<true> + xtwx_cols_reduce_kernel (vectors of V)
  KT_SUF,           // xtx_mfma_kernel + plane_sum_kernel + col_reduce_kernel
  KT_POISSON_IMPUTE,// poisson_impute_kernel
  KT_KALMAN_PREPARE,// kalman_prepare_kernel (level variance + normals, second stream)
  KT_SS_ROUND,      // ss_round_kernel (the local-level bsts rounds of a call, one persistent launch)
  KT_CLASSES
};

void kt_mark(hipStream_t stream, int cls, bool begin);   // engine.hip
bool kt_active();

struct KtScope {
  hipStream_t s;
  int c;
  bool on;
  KtScope(hipStream_t stream, int cls) : s(stream), c(cls), on(kt_active()) {
    if (on) kt_mark(s, c, true);
  }
  ~KtScope() {
    if (on) kt_mark(s, c, false);
  }
};

}  // namespace boom_amd
#endif
