// Posterior predictive means from the recorded draws: what lm_spike.predict does on
// the host with the draws it kept (Interfaces/python/spikeslab/BayesBoom/spikeslab/
// spikeslab.py:530-546: coefficient_draws[burn:, :] @ predictors.T), for every chain at
// once and from the device's own record (ba_enable_draws) -- a draw is its k included
// variables and their coefficients, so a prediction costs k, not p, terms.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace boom_amd {

// grid = (ceil(nnew / 256), ndraws, chains).  out[chain][draw][i] = sum_m beta_m
// newX[i, var_m]; newX is column-major nnew x p.
__global__ __launch_bounds__(256) void predict_kernel(const double *__restrict__ trace_k,
                                                      const uint16_t *__restrict__ rec_idx,
                                                      const double *__restrict__ rec_beta, int stride,
                                                      int cap, int first_draw, int ndraws, int p,
                                                      const double *__restrict__ newX, int nnew,
                                                      double *__restrict__ out) {
  const int i = (int)(blockIdx.x * 256 + threadIdx.x), d = (int)blockIdx.y, c = (int)blockIdx.z;
  const size_t row = (size_t)c * stride + (size_t)(first_draw + d);
  int k = (int)trace_k[row];
  if (k > cap) k = cap;
  if (i >= nnew) return;
  const uint16_t *idx = rec_idx + row * cap;
  const double *b = rec_beta + row * cap;
  double acc = 0.0;
  for (int m = 0; m < k; ++m) acc += b[m] * newX[(size_t)(idx[m] % (unsigned)p) * nnew + i];
  out[((size_t)c * ndraws + d) * nnew + i] = acc;
}

hipError_t launch_predict(hipStream_t stream, const double *trace_k, const uint16_t *rec_idx,
                          const double *rec_beta, int stride, int cap, int first_draw, int ndraws,
                          int chains, int p, const double *newX, int nnew, double *out) {
  hipLaunchKernelGGL(predict_kernel, dim3((nnew + 255) / 256, ndraws, chains), dim3(256), 0, stream, trace_k,
                     rec_idx, rec_beta, stride, cap, first_draw, ndraws, p, newX, nnew, out);
  return hipGetLastError();
}

}  // namespace boom_amd
