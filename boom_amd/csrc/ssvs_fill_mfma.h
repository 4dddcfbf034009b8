// The table fill's proposals on the f64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// A fill evaluates log_model_prob(gamma ^ {j}) for EVERY j against the current model
// (BregVsSampler::log_model_prob, BregVsSampler.cpp:216-239, restated as a change of the
// current model: ssvs_kernel.hip's header): for each j two triangular solves L x = rhs_j,
// L = chol(V_g) and chol(A_g).  eval_proposal / big_eval give one j to a lane, keep x in the
// lane's registers and take L's elements as SGPR operands -- 128 SGPRs a block, so half a
// block in flight at a time, each half a trip to L2 (the factors of a CU's eight chains do
// not fit the 16 KB scalar cache beyond k = 16): at k = 60 a solve took 60 k cycles where
// its 2 080 FMAs need 8 k.
//
// The 64 right-hand sides of a wavefront are a k x 64 MATRIX, and L^{-1} B is a blocked
// triangular solve: block row I (16 rows) of the solution is
//     X_I = inv(L_II) (B_I - sum_{J<I} L_IJ X_J),
// every product an MFMA with L's 16 x 4 chunks as the A operand (one 8-byte load per lane
// per chunk, from the chain's model block in HBM -- L2 resident --, used for all the
// column tiles) and the solution tiles as B operand AND accumulator: register q of lane l
// of a D tile holds row 4 q + (l >> 4), column l & 15, which is exactly the B operand of
// K-chunk q.  inv(L_II), the inverse of the factor's 16 x 16 diagonal blocks, is computed
// once per model (diag_inverses, when the model block is published) instead of
// substituting across lanes.  The arithmetic differs from the per-lane solve's in the
// order of its sums (and in inv(L_II) where that divides): same numbers to rounding, the
// same Markov chain -- the parity tests' bar is the oracle's draws, not this route.
//
// A pass holds NT = 2 column tiles (32 proposals, 16 doubles of x per 16 rows and lane):
// two passes serve a wavefront's 64 proposals; up to 128 rows (8 block rows) fit the
// registers.  Larger models keep the per-lane route.
#pragma once
// (included by ssvs_device.h, after its address-space types and bidx)

namespace boom_amd {

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

enum : int { MF_ROWS = 16, MF_NT = 2, MF_MAX_BLOCK_ROWS = 8,
             MF_MIN_NB = 6 };   // the LDS kernel's instances of capacity 8 MF_MIN_NB and more fill this way

// block rows the MFMA route is compiled for: the smallest of {2, 3, 4, 5, 6, 8} that holds k
// (never more than the capacity the model block was laid out for has)
__host__ __device__ inline int mf_block_rows(int k) {
  return k <= 32 ? 2 : (k <= 48 ? 3 : (k <= 64 ? 4 : (k <= 80 ? 5 : (k <= 96 ? 6 : 8))));
}

// inv(L_II) for nI block rows of 16, row-major 16 x 16 at inv + 256 I.  One wavefront.  L:
// the factor, block packed (rows < (krows + 7) & ~7 written), rd its reciprocal diagonal --
// read from LDS where the factorisation has just left them (the sweep kernel's factors, the
// large-model kernel's 64 x 64 tile: `L` and `rd` are then tile-local), since reading them
// back from the model block cost a dozen dependent trips to memory per build.  Rows >= krows
// give zero rows.
template <class LP, class RP>
__device__ __forceinline__ void diag_inverses(LP L, RP rd, double *inv, int krows, int nI, int lane) {
  const int kpad8 = (krows + 7) & ~7;
  const int b = lane >> 4, c = lane & 15;
  for (int I0 = 0; I0 < nI; I0 += 4) {
    const int I = I0 + b;
    if (I < nI) {
      double x[MF_ROWS];
#pragma unroll
      for (int r = 0; r < MF_ROWS; ++r) {
        const int R = MF_ROWS * I + r;
        double acc = 0.0;
#pragma unroll
        for (int s = 0; s < r; ++s) {
          const double l = L[bidx(R, MF_ROWS * I + s)];
          acc += ((R < kpad8) ? l : 0.0) * x[s];
        }
        const double rdr = rd[R];
        const double rdv = (R < krows) ? rdr : 0.0;
        x[r] = (r == c) ? rdv : ((r < c) ? 0.0 : -acc * rdv);
      }
#pragma unroll
      for (int r = 0; r < MF_ROWS; ++r) inv[I * (MF_ROWS * MF_ROWS) + r * MF_ROWS + c] = x[r];
    }
  }
}

// X <- L^{-1} X for NI block rows of 16, MF_NT column tiles; Lg: the block-packed factor,
// Linv: its inverse diagonal blocks (global memory).  Rows >= kpad8 of L are not there.
// The A operands are a fixed list of groups of four chunks -- (1, 0), diag 1, (2, 0),
// (2, 1), diag 2, ... in the order the products need them -- and they do not depend on X:
// group n + D is asked for before group n's products are issued (a ring of D + 1 groups,
// eight registers each), so a trip to L2 or beyond hides behind 8 D MFMAs instead of
// stalling each block row.  (Left to itself the compiler hoisted every load of the solve to the top: 148
// spilled registers beside the 128 of X.)
constexpr int mf_group_row(int n) { int I = 0; while ((I + 1) * (I + 2) / 2 <= n) ++I; return I; }
template <int NI>
__device__ __forceinline__ void mf_tri_solve(const double *__restrict__ Lg, const double *__restrict__ Linv,
                                             int kpad8, int lane, v4d (&X)[NI][MF_NT]) {
  const int g = lane >> 4, r = lane & 15;
  constexpr int G = NI * (NI + 1) / 2;
  constexpr int D = NI <= 5 ? 6 : (NI <= 6 ? 4 : 3);   // groups asked for ahead of the one in use
  double ring[D + 1][4];
  // group n = (I, J), J <= I (J == I: the diagonal block's inverse)
  auto fetch = [&](int n, double (&a)[4]) {
    const int I = mf_group_row(n), J = n - I * (I + 1) / 2;
    if (J == I) {
      const double *ip = Linv + I * (MF_ROWS * MF_ROWS) + r * MF_ROWS + g;
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] = ip[4 * c];
    } else {
      // element (16 I + r, 16 J + 4 c + g): 8 x 8 block (2 I + (r >> 3), 2 J + (c >> 1))
      // (as loaded: the sign and the mask of the rows that are not there go on where the group
      // is used -- applied here, every load was waited for in turn, through one pair of
      // temporary registers, a chain of four to eleven round trips per group of the prologue)
      const int bi = 2 * I + (r >> 3);
      const double *rowp = Lg + ((bi * (bi + 1)) / 2) * 64 + (r & 7) * 8 + g;
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] = rowp[(2 * J + (c >> 1)) * 64 + 4 * (c & 1)];
    }
  };
#pragma unroll
  for (int n = 0; n < D; ++n)
    if (n < G) fetch(n, ring[n]);
#pragma unroll
  for (int n = 0; n < G; ++n) {
    if (n + D < G) fetch(n + D, ring[(n + D) % (D + 1)]);
    __builtin_amdgcn_sched_barrier(0);
    const int I = mf_group_row(n), J = n - I * (I + 1) / 2;
    double a[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = ring[n % (D + 1)][c];
    if (J != I) {
      const bool rowok = MF_ROWS * I + r < kpad8;
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] = rowok ? -a[c] : 0.0;
    }
    if (J == I) {
      v4d out[MF_NT];
#pragma unroll
      for (int t = 0; t < MF_NT; ++t) out[t] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < MF_NT; ++t)
          out[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c], X[I][t][c], out[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < MF_NT; ++t) X[I][t] = out[t];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < MF_NT; ++t)
          X[I][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c], X[J][t][c], X[I][t], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ double mf_group_sum(double x) {   // over the four lanes l & 15 == const
  x += __shfl_xor(x, 16);
  x += __shfl_xor(x, 32);
  return x;
}

// What a proposal's two solves leave: |x_V|^2, x_V . w, |x_A|^2, rhs_A . b_g
struct MfSums { double nv, dv, na, ab; };

// One factor of one pass: the columns' right-hand sides gathered, solved, reduced.  SF: 0
// the factor of V_g (sum2 = x . w), 1 that of A_g (sum2 = rhs . b_g).  A template
// parameter, not a loop variable: the weights are read where they are used only (as one
// rolled loop over the two factors the compiler loaded them once for both uses and kept
// 64 registers of them across the solve).
template <int NI, int SF>
__device__ __forceinline__ void mf_factor(const double *__restrict__ Mat, const double (&addf)[MF_NT],
                                          const double (&dropf)[MF_NT], const int (&jt)[MF_NT], int p,
                                          const double *__restrict__ Lg, const double *__restrict__ Linv,
                                          const double *__restrict__ wv, lds_u16 *glist, int k, int lane_in,
                                          double (&n2)[MF_NT], double (&sum2)[MF_NT]) {
  // (the lane number made opaque per call: every address below derives from it, and what the
  // compiler can prove invariant it hoists out of the caller's loop over passes and rounds
  // and keeps in registers -- ~120 of them -- across everything)
  int lane = lane_in;
  asm volatile("" : "+v"(lane));
  const int g = lane >> 4, kpad8 = (k + 7) & ~7;
  v4d X[NI][MF_NT];
#pragma unroll
  for (int t = 0; t < MF_NT; ++t) { n2[t] = 0.0; sum2[t] = 0.0; }
  // every gather of the factor goes out before the first is looked at: the loads land in
  // X's own registers, so what is in flight costs no more than X does
  int gmv[NI][4];
#pragma unroll
  for (int I = 0; I < NI; ++I) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = MF_ROWS * I + 4 * q + g;
      const int gm = (int)glist[m < k ? m : 0];
      gmv[I][q] = gm;
#pragma unroll
      for (int t = 0; t < MF_NT; ++t) X[I][t][q] = Mat[(size_t)gm * p + jt[t]];
    }
  }
  // (the weights of a factor are asked for together, like the gathers -- read one at a time
  // where each is used, every one of up to 32 was a round trip of its own --, four block rows
  // (sixteen loads, 32 registers) at a time: all 32 at once cost the capacity-48 / 64 instances of
  // the sweep kernel 56-72 B/lane more scratch beside X's 96-128 registers)
  constexpr int WG = 4;
#pragma unroll
  for (int I0 = 0; I0 < NI; I0 += WG) {
    double wr[WG][4];
    if (SF) {
#pragma unroll
      for (int i = 0; i < WG; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) wr[i][q] = (I0 + i < NI) ? wv[MF_ROWS * (I0 + i) + 4 * q + g] : 0.0;
#pragma unroll
      for (int i = 0; i < WG; ++i) asm volatile("" : "+v"(wr[i][0]), "+v"(wr[i][1]), "+v"(wr[i][2]), "+v"(wr[i][3]));
    }
#pragma unroll
    for (int i = 0; i < WG; ++i) {
      const int I = I0 + i;
      if (I < NI) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = MF_ROWS * I + 4 * q + g;
          double wm = 0.0;
          if (SF) wm = (m < k) ? wr[i][q] : 0.0;
#pragma unroll
          for (int t = 0; t < MF_NT; ++t) {
            const double e = (gmv[I][q] == jt[t]) ? dropf[t] : 0.0;
            const double val = fma(X[I][t][q], addf[t], e);
            X[I][t][q] = val;
            if (SF) sum2[t] += val * wm;   // A[j, g] . b_g
          }
        }
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  mf_tri_solve<NI>(Lg, Linv, kpad8, lane, X);
#pragma unroll
  for (int I0 = 0; I0 < NI; I0 += WG) {
    double wr[WG][4];
    if (!SF) {
#pragma unroll
      for (int i = 0; i < WG; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) wr[i][q] = (I0 + i < NI) ? wv[MF_ROWS * (I0 + i) + 4 * q + g] : 0.0;
#pragma unroll
      for (int i = 0; i < WG; ++i) asm volatile("" : "+v"(wr[i][0]), "+v"(wr[i][1]), "+v"(wr[i][2]), "+v"(wr[i][3]));
    }
#pragma unroll
    for (int i = 0; i < WG; ++i) {
      const int I = I0 + i;
      if (I < NI) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = MF_ROWS * I + 4 * q + g;
          double wm = 0.0;
          if (!SF) wm = (m < k) ? wr[i][q] : 0.0;
#pragma unroll
          for (int t = 0; t < MF_NT; ++t) {
            const double x = X[I][t][q];
            n2[t] += x * x;
            if (!SF) sum2[t] += x * wm;   // x_V . w
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < MF_NT; ++t) {
    n2[t] = mf_group_sum(n2[t]);
    sum2[t] = mf_group_sum(sum2[t]);
  }
}

// One pass: the 32 proposals jbase + 32 ps + 16 t + (lane & 15), t < 2.  flags: bit 0 fast,
// bit 1 add, of THIS lane's proposal jbase + lane (lane = proposal layout); the pass's
// lanes (lane >> 5 == ps) get their proposal's sums in `out`.
//   gblock  the chain's model block (global), S its layout; glist: the sorted index list
//           (LDS); V, A: the shared matrices (NAT reads: element (g_m, j))
template <int NI>
__device__ __forceinline__ void mf_pass(const double *__restrict__ V, const double *__restrict__ A, int p,
                                        double sv, double sa, const double *__restrict__ gblock,
                                        const SsvsScalarLayout &S, uint32_t inv_v, uint32_t inv_a,
                                        lds_u16 *glist, int k, int jbase, int ps, int flags, int lane,
                                        MfSums &out) {
  const int c = lane & 15;
  // Column t's right-hand side is V[g, j] (add) or e_i (drop), written as a blend --
  // Mat * addf + e, addf = the matrix's scale or 0 -- so that no load hangs on a per-lane
  // condition (a conditional load is a branch, and branches between the gathers would
  // serialise their round trips).  Rows >= k and the columns of proposals that are not
  // `fast` carry finite numbers nobody reads: inv(L_II)'s rows >= k are zero, the weights
  // w and b_g are zero there, and the caller uses a column's sums only if it is fast.
  int jt[MF_NT];
  double addv[MF_NT], adda[MF_NT], dropf[MF_NT];
#pragma unroll
  for (int t = 0; t < MF_NT; ++t) {
    const int src = 32 * ps + 16 * t + c;
    const int fl = __shfl(flags, src);
    const bool fast = (fl & 1) != 0, add = (fl & 2) != 0;
    jt[t] = fast ? jbase + src : 0;
    addv[t] = add ? sv : 0.0;
    adda[t] = add ? sa : 0.0;
    dropf[t] = add ? 0.0 : 1.0;
  }
  double nv[MF_NT], dv[MF_NT], na[MF_NT], ab[MF_NT];
  mf_factor<NI, 0>(V, addv, dropf, jt, p, gblock + S.Lv, gblock + inv_v, gblock + S.w, glist, k, lane, nv, dv);
  mf_factor<NI, 1>(A, adda, dropf, jt, p, gblock + S.La, gblock + inv_a, gblock + S.bg, glist, k, lane, na, ab);
  if ((lane >> 5) == ps) {
    const bool hi = ((lane >> 4) & 1) != 0;
    out.nv = hi ? nv[1] : nv[0];
    out.dv = hi ? dv[1] : dv[0];
    out.na = hi ? na[1] : na[0];
    out.ab = hi ? ab[1] : ab[0];
  }
}

// The sums of the wavefront's 64 proposals jbase + lane (k <= 128).  MAXNI: the block rows
// the caller's capacity can reach (its kernel carries no code for more).
template <int MAXNI>
__device__ __forceinline__ MfSums mf_proposal_sums(const double *V, const double *A, int p, double sv, double sa,
                                                   const double *gblock, const SsvsScalarLayout &S,
                                                   uint32_t inv_v, uint32_t inv_a, lds_u16 *glist, int k,
                                                   int jbase, int flags, int lane) {
  MfSums out{0.0, 0.0, 0.0, 0.0};
  const int nI = mf_block_rows(k);
#pragma nounroll
  for (int ps = 0; ps < 2; ++ps) {
    if (nI == 2) mf_pass<2>(V, A, p, sv, sa, gblock, S, inv_v, inv_a, glist, k, jbase, ps, flags, lane, out);
    if constexpr (MAXNI >= 3) { if (nI == 3) mf_pass<3>(V, A, p, sv, sa, gblock, S, inv_v, inv_a, glist, k, jbase, ps, flags, lane, out); }
    if constexpr (MAXNI >= 4) { if (nI == 4) mf_pass<4>(V, A, p, sv, sa, gblock, S, inv_v, inv_a, glist, k, jbase, ps, flags, lane, out); }
    if constexpr (MAXNI >= 5) { if (nI == 5) mf_pass<5>(V, A, p, sv, sa, gblock, S, inv_v, inv_a, glist, k, jbase, ps, flags, lane, out); }
    if constexpr (MAXNI >= 6) { if (nI == 6) mf_pass<6>(V, A, p, sv, sa, gblock, S, inv_v, inv_a, glist, k, jbase, ps, flags, lane, out); }
    if constexpr (MAXNI >= 8) { if (nI == 8) mf_pass<8>(V, A, p, sv, sa, gblock, S, inv_v, inv_a, glist, k, jbase, ps, flags, lane, out); }
  }
  return out;
}

}  // namespace

}  // namespace boom_amd
