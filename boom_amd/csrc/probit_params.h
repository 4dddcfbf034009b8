// Launch parameters of the probit data-augmentation kernel (probit_kernel.hip),
// shared with the host side (engine.hip).
#pragma once
#include <stdint.h>

#include "ssvs_params.h"

namespace boom_amd {

enum { PROBIT_STRIDE = 4096, PROBIT_KMAX = 1024, LOGIT_STRIDE = 256, PG_STRIDE = 4096, POISSON_STRIDE = 256,
       POISSON_MAX_COMP = 32 };

struct ProbitParams {
  int32_t n, p, chains, clt_threshold;
  int64_t chain_offset;
  const double *X;        // n x p column-major
  const double *y;        // successes
  const double *ntrials;  // trials
  const uint8_t *gamma;   // chains x p
  const double *beta;     // chains x p
  double *z;              // chains x n: the observations' sums of latent normals
  double *xtz;            // chains x p: X'z
  double *w;              // chains x n: the observations' total precision (logit only)
  uint32_t seed_lo, seed_hi;
  uint64_t sweep;         // imputations done so far (positions the substreams)
  int32_t *status;
  // Poisson regression (poisson_impute_kernel): ntrials holds the exposures; the
  // reference table's normal mixtures of NegLogGamma(count) -- mixture m has components
  // [mix_off[m], mix_off[m + 1]) of (mix_mu, mix_sigma, mix_logw); obs_mix[i] = the
  // mixture of observation i's count (-1: the Gaussian limit beyond the table; unused
  // for a zero count), mix_one = the mixture of count 1
  const int32_t *mix_off;
  const double *mix_mu, *mix_sigma, *mix_logw;
  const int32_t *obs_mix;
  int32_t mix_one;
  int32_t slot_limit;     // > 0: uniforms a substream slot serves before its spill stream (default: the stride)
};

}  // namespace boom_amd
