// Blocked right-looking Cholesky factorisation (lower, in place, row-major) for gfx950.
//
// Replaces numpy.linalg.cholesky at regression.py:241 (fit), :537 (marginal_likelihood) and :555
// (marginal_likelihood_gradient).  Two-level blocking:
//   outer panels of OB columns  -> one trailing SYRK update with K = OB on the MFMA GEMM (the
//                                   compute-bound kernel: intensity OB/8 FLOP per byte of C traffic)
//   inner blocks of 128 columns -> potrf_diag (one workgroup: factor the 128 x 128 diagonal block
//                                   and invert it), panel TRSM as a product with the inverse
//                                   (MFMA), and the update of the rest of the outer panel (K = 128).
// A non-positive or non-finite pivot is reported LAPACK-style through `info` (first failing
// column + 1); the factorisation then continues with a unit pivot so that the launch sequence stays
// asynchronous — the host inspects `info` once at the end (regression.py:540-542 behaviour).
#include "gpmi_internal.h"

namespace {

constexpr int NB = GPMI_NB;
constexpr int SP = NB + 1;  // LDS row pitch (conflict-free row and column walks)

// One workgroup: L = chol(A_blk) in place (lower part), invD = L^-1 (dense 128 x 128, zero above
// the diagonal).
__global__ __launch_bounds__(256) void potrf_diag_kernel(double* __restrict__ A, int64_t ld,
                                                         double* __restrict__ invD,
                                                         int* __restrict__ info, int col0) {
  __shared__ double S[NB * SP];
  __shared__ double dinv[NB];
  const int tid = threadIdx.x;
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx >> 7, c = idx & 127;
    S[r * SP + c] = A[(int64_t)r * ld + c];
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  for (int j = 0; j < NB; ++j) {
    double ajj = S[j * SP + j];
    if (!(ajj > 0.0) || !(ajj < 1.79e308)) {
      if (tid == 0 && *info == 0) *info = col0 + j + 1;
      ajj = 1.0;
    }
    const double dj = sqrt(ajj);
    __syncthreads();  // everyone has read the pivot before it is overwritten
    if (tid == j) {
      S[j * SP + j] = dj;
      dinv[j] = 1.0 / dj;
    }
    if (tid > j && tid < NB) S[tid * SP + j] = S[tid * SP + j] / dj;
    __syncthreads();
    // rank-1 update of the trailing lower triangle
    for (int i = j + 1 + ty; i < NB; i += 16) {
      const double lij = S[i * SP + j];
      for (int k = j + 1 + tx; k <= i; k += 16) S[i * SP + k] -= lij * S[k * SP + j];
    }
    // the next pivot S[j+1][j+1] is final only after the update
    __syncthreads();
  }
  // write L back (lower triangle incl. diagonal; the upper triangle of A is left untouched)
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx >> 7, c = idx & 127;
    if (c <= r) A[(int64_t)r * ld + c] = S[r * SP + c];
  }
  // X = L^-1 by forward substitution, thread c owns column c.  X[i][c] (i > c) is kept in the
  // upper triangle of S (transposed: S[c][i]) so that both L and X stay in LDS.
  __syncthreads();
  if (tid < NB) {
    const int c = tid;
    for (int i = 1; i < NB; ++i) {
      // all lanes walk k uniformly so that L[i][k] is a broadcast read
      double acc = 0.0;
      for (int k = 0; k < i; ++k) {
        const double lik = S[i * SP + k];
        double xk = 0.0;
        if (k == c) xk = dinv[c];
        else if (k > c) xk = S[c * SP + k];
        acc = fma(lik, xk, acc);
      }
      if (i > c) S[c * SP + i] = -acc * dinv[i];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < NB * NB; idx += 256) {
    const int r = idx >> 7, c = idx & 127;
    double v = 0.0;
    if (c == r) v = dinv[r];
    else if (c < r) v = S[c * SP + r];
    invD[r * NB + c] = v;
  }
}

}  // namespace

void launch_potrf_diag(hipStream_t s, double* Ablk, int64_t ld, double* invD, int* info, int col0) {
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), 0, s, Ablk, ld, invD, info, col0);
}

void potrf_lower(gpmi_ctx* c, hipStream_t s, double* A, int64_t np, int64_t ld, double* invD,
                 int* info) {
  const int nt = (int)(np / NB);
  const int OBT = 4;  // outer panel = 4 inner blocks = 512 columns
  for (int J = 0; J < nt; J += OBT) {
    const int Je = (J + OBT < nt) ? J + OBT : nt;
    for (int j = J; j < Je; ++j) {
      double* Ajj = A + (int64_t)j * NB * ld + (int64_t)j * NB;
      double* invDj = invD + (int64_t)j * NB * NB;
      const int below = nt - j - 1;
      {
        ProfScope ps(c, s, GPMI_PROF_PANEL, (double)NB * NB * NB / 3.0 + 2.0 * below * NB * NB * NB,
                     8.0 * NB * NB * (2.0 + 2.0 * below));
        launch_potrf_diag(s, Ajj, ld, invDj, info, j * NB);
        if (below > 0) {
          // panel TRSM: A21 <- A21 * L11^-T  (in place: one tile column, see gemm_f64.hip)
          double* A21 = Ajj + (int64_t)NB * ld;
          launch_gemm_nt(s, TILES_RECT, OP_ASSIGN, A21, ld, A21, ld, invDj, NB, below, 1, NB);
        }
      }
      const int pc = Je - j - 1;  // remaining block columns of the outer panel
      if (below > 0 && pc > 0) {
        // inner update of the rest of the outer panel, rows below: tiles (ti >= tj, tj < pc)
        double* A21 = Ajj + (int64_t)NB * ld;
        double* C = A21 + NB;
        const double tiles = pc * (pc + 1) / 2.0 + (double)(below - pc) * pc;
        ProfScope ps(c, s, GPMI_PROF_PANEL, tiles * 2.0 * NB * NB * NB, tiles * 16.0 * NB * NB);
        launch_gemm_nt(s, TILES_LOWER, OP_SUB, C, ld, A21, ld, A21, ld, below, pc, NB);
      }
    }
    const int rem = nt - Je;
    if (rem > 0) {
      // trailing update: A22 -= P P^T with P = A[Je.., J..Je) (K = (Je - J) * 128), lower tiles only
      const int kw = (Je - J) * NB;
      double* P = A + (int64_t)Je * NB * ld + (int64_t)J * NB;
      double* C = A + (int64_t)Je * NB * ld + (int64_t)Je * NB;
      const double tiles = rem * (rem + 1) / 2.0;
      ProfScope ps(c, s, GPMI_PROF_SYRK, tiles * 2.0 * NB * NB * kw,
                   tiles * 16.0 * NB * NB + 8.0 * rem * NB * kw);
      launch_gemm_nt(s, TILES_LOWER, OP_SUB, C, ld, P, ld, P, ld, rem, rem, kw);
    }
  }
}
