// Probe (round 4): candidates for the 16 x 16 elimination on the chain of potrf_diag, one wave each, timed with
// s_memtime and checked against a host Cholesky; plus the instruction latencies the design rests on.
// build: hipcc --offload-arch=gfx950 -O3 -o build/factor16_probe factor16_probe.hip
//
// Layout of the candidates ("column per lane"): lane (g = lane >> 4, k = lane & 15) holds ALL 16 rows of column k of
// the block (x[0..15], the four lane groups redundantly) and the entries E[k][4 q + g] (q = 0..3) of row k of the
// accumulated row operations.  A step needs no LDS and no cross-row traffic:
//   x[i] += bcast_C(x[i]) * nt      nt = -(x[C] / p_C)   (the lane's own pivot-row element)
//   e[q] += bcast_C(e[q]) * nte     nte = nt for the rows below the pivot, 0 elsewhere
// where bcast_C = DPP row_newbcast:C.  gfx950 has the fused form v_fmac_f64_dpp.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

template <int C>
__device__ inline double bc(double v) {
  long long x = __builtin_bit_cast(long long, v);
  x = __builtin_amdgcn_mov_dpp(x, 0x150 + C, 0xf, 0xf, false);
  return __builtin_bit_cast(double, x);
}
__device__ inline double rcp_newton(double p) {
  double y = __builtin_amdgcn_rcp(p);
  double e = fma(-p, y, 1.0);
  y = fma(y, e, y);
  e = fma(-p, y, 1.0);
  return fma(y, e, y);
}

struct St {
  double x[16], e[4];
  double p, ip, myp;
  int badcol, k, g;
};

// ---------------------------------------------------------------- V1: builtins only (compiler handles the hazards)
template <int C>
struct StepV1 {
  static __device__ inline void run(St& s) {
    if (s.k == C) s.myp = s.p;
    if constexpr (C < 15) {
      const double nt = -(s.x[C] * s.ip);
      const double nte = (s.k > C) ? nt : 0.0;
      s.x[C + 1] = fma(bc<C>(s.x[C + 1]), nt, s.x[C + 1]);
      double pn = bc<C + 1>(s.x[C + 1]);
      if (!(pn > 0.0) || !(pn < 1.79e308)) {
        if (s.badcol < 0) s.badcol = C + 1;
        pn = 1.0;
      }
      const double ipn = rcp_newton(pn);
#pragma unroll
      for (int i = C + 2; i < 16; ++i) s.x[i] = fma(bc<C>(s.x[i]), nt, s.x[i]);
#pragma unroll
      for (int q = 0; q <= (C >> 2); ++q) s.e[q] = fma(bc<C>(s.e[q]), nte, s.e[q]);
      s.p = pn;
      s.ip = ipn;
      StepV1<C + 1>::run(s);
    }
  }
};

// ---------------------------------------------------------------- V2: fused v_fmac_f64_dpp, fixed issue order
#define FMAC_DPP(dst, mul, C) \
  asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(dst) : "v"(mul), "n"(C))
template <int C>
struct StepV2 {
  static __device__ inline void run(St& s) {
    if (s.k == C) s.myp = s.p;
    if constexpr (C < 15) {
      double nt;
      asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(nt) : "v"(s.x[C]), "v"(s.ip));
      const double nte = (s.k > C) ? nt : 0.0;
      FMAC_DPP(s.x[C + 1], nt, C);
      // two issue slots between the write of x[C+1] and its DPP read
      if constexpr (C + 2 < 16) FMAC_DPP(s.x[C + 2], nt, C);
      if constexpr (C + 3 < 16)
        FMAC_DPP(s.x[C + 3], nt, C);
      else
        asm volatile("s_nop 1");
      double y0, pb;
      asm volatile("v_rcp_f64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(y0) : "v"(s.x[C + 1]), "n"(C + 1));
      asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(pb) : "v"(s.x[C + 1]), "n"(C + 1));
      // Newton on the reciprocal with the remaining updates in the shadows of its dependent steps
      double e1, y1, e2, y2;
      asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e1) : "v"(pb), "v"(y0));
      if constexpr (C + 4 < 16) FMAC_DPP(s.x[C + 4], nt, C);
      if constexpr (C + 5 < 16) FMAC_DPP(s.x[C + 5], nt, C);
      asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(y1) : "v"(y0), "v"(e1));
      if constexpr (C + 6 < 16) FMAC_DPP(s.x[C + 6], nt, C);
      if constexpr (C + 7 < 16) FMAC_DPP(s.x[C + 7], nt, C);
      asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e2) : "v"(pb), "v"(y1));
      if constexpr (C + 8 < 16) FMAC_DPP(s.x[C + 8], nt, C);
      if constexpr (C + 9 < 16) FMAC_DPP(s.x[C + 9], nt, C);
      asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(y2) : "v"(y1), "v"(e2));
#pragma unroll
      for (int i = C + 10; i < 16; ++i) FMAC_DPP(s.x[i], nt, C);
#pragma unroll
      for (int q = 0; q <= (C >> 2); ++q) FMAC_DPP(s.e[q], nte, C);
      if (__builtin_expect(!((pb > 0.0) && (pb < 1.79e308)), 0)) {  // wave-uniform, never taken on a healthy matrix
        asm volatile("; non-positive pivot");                      // (keeps the branch a branch: off the chain)
        if (s.badcol < 0) s.badcol = C + 1;
        pb = 1.0;
        y2 = 1.0;
      }
      s.p = pb;
      s.ip = y2;
      StepV2<C + 1>::run(s);
    }
  }
};

// ---------------------------------------------------------------- V3: V2 with the next pivot formed one step ahead
// (p_{C+1} = s2 - (s1 s3) / p_C from three broadcasts taken before step C's updates: one fma between ip_C and the rcp)
template <int C>
struct StepV3 {
  // s13, s2: A[C+1][C] * A[C][C+1] and A[C+1][C+1] before step C (wave-uniform per 16-lane row)
  static __device__ inline void run(St& s, double s13, double s2) {
    if (s.k == C) s.myp = s.p;
    if constexpr (C < 15) {
      double pb, nt, y0;
      asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(pb) : "v"(s13), "v"(s.ip), "v"(s2));
      asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(nt) : "v"(s.x[C]), "v"(s.ip));
      asm volatile("v_rcp_f64 %0, %1" : "=v"(y0) : "v"(pb));
      const double nte = (s.k > C) ? nt : 0.0;
      FMAC_DPP(s.x[C + 1], nt, C);
      if constexpr (C + 2 < 16) FMAC_DPP(s.x[C + 2], nt, C);
      double e1, y1, e2, y2;
      asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e1) : "v"(pb), "v"(y0));
      if constexpr (C + 3 < 16) FMAC_DPP(s.x[C + 3], nt, C);
      // the broadcasts of the NEXT step's look-ahead: x[C+1], x[C+2] are final for step C from here on
      double n1 = 0, n2 = 0, n3 = 0, n13 = 0;
      if constexpr (C + 2 < 16) {
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(n1) : "v"(s.x[C + 2]), "n"(C + 1));
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(n3) : "v"(s.x[C + 1]), "n"(C + 2));
      }
      asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(y1) : "v"(y0), "v"(e1));
      if constexpr (C + 2 < 16) {
        asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(n2) : "v"(s.x[C + 2]), "n"(C + 2));
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(n13) : "v"(n1), "v"(n3));
      }
      asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e2) : "v"(pb), "v"(y1));
      if constexpr (C + 4 < 16) FMAC_DPP(s.x[C + 4], nt, C);
      if constexpr (C + 5 < 16) FMAC_DPP(s.x[C + 5], nt, C);
      asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(y2) : "v"(y1), "v"(e2));
#pragma unroll
      for (int i = C + 6; i < 16; ++i) FMAC_DPP(s.x[i], nt, C);
#pragma unroll
      for (int q = 0; q <= (C >> 2); ++q) FMAC_DPP(s.e[q], nte, C);
      if (__builtin_expect(!((pb > 0.0) && (pb < 1.79e308)), 0)) {
        asm volatile("; non-positive pivot");
        if (s.badcol < 0) s.badcol = C + 1;
        pb = 1.0;
        y2 = 1.0;
      }
      s.p = pb;
      s.ip = y2;
      StepV3<C + 1>::run(s, n13, n2);
    }
  }
};


// ---------------------------------------------------------------- V4: V2's order without v_rcp_f64_dpp, one cubic
// refinement step instead of two Newton steps (rcp: 2^-24.4 -> 1 ulp), pivots checked once at the end
#define MOV_DPP(dst, src, L) \
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src), "n"(L))
template <int C>
struct StepV4 {
  static __device__ inline void run(St& s) {
    if (s.k == C) s.myp = s.p;
    if constexpr (C < 15) {
      double nt;
      asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(nt) : "v"(s.x[C]), "v"(s.ip));
      const double nte = (s.k > C) ? nt : 0.0;
      FMAC_DPP(s.x[C + 1], nt, C);
      if constexpr (C + 2 < 16) FMAC_DPP(s.x[C + 2], nt, C);
      if constexpr (C + 3 < 16)
        FMAC_DPP(s.x[C + 3], nt, C);
      else
        asm volatile("s_nop 1");
      double y0, pb, e1, e2, y;
      MOV_DPP(pb, s.x[C + 1], C + 1);
      if constexpr (C + 4 < 16) FMAC_DPP(s.x[C + 4], nt, C);
      asm volatile("v_rcp_f64 %0, %1" : "=v"(y0) : "v"(pb));
      if constexpr (C + 5 < 16) FMAC_DPP(s.x[C + 5], nt, C);
      if constexpr (C + 6 < 16) FMAC_DPP(s.x[C + 6], nt, C);
      if constexpr (C + 7 < 16) FMAC_DPP(s.x[C + 7], nt, C);
      asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e1) : "v"(pb), "v"(y0));
      if constexpr (C + 8 < 16) FMAC_DPP(s.x[C + 8], nt, C);
      asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(e2) : "v"(e1));
      if constexpr (C + 9 < 16) FMAC_DPP(s.x[C + 9], nt, C);
      asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(y) : "v"(y0), "v"(e2));
#pragma unroll
      for (int i = C + 10; i < 16; ++i) FMAC_DPP(s.x[i], nt, C);
#pragma unroll
      for (int q = 0; q <= (C >> 2); ++q) FMAC_DPP(s.e[q], nte, C);
      s.p = pb;
      s.ip = y;
      StepV4<C + 1>::run(s);
    }
  }
};

// ---------------------------------------------------------------- V5: V4 with the pivot one step ahead (V3's scheme)
template <int C>
struct StepV5 {
  static __device__ inline void run(St& s, double s13, double s2) {
    if (s.k == C) s.myp = s.p;
    if constexpr (C < 15) {
      double pb, nt, y0, e1, e2, y;
      asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(pb) : "v"(s13), "v"(s.ip), "v"(s2));
      asm volatile("v_mul_f64 %0, -%1, %2" : "=v"(nt) : "v"(s.x[C]), "v"(s.ip));
      asm volatile("v_rcp_f64 %0, %1" : "=v"(y0) : "v"(pb));
      const double nte = (s.k > C) ? nt : 0.0;
      FMAC_DPP(s.x[C + 1], nt, C);
      if constexpr (C + 2 < 16) FMAC_DPP(s.x[C + 2], nt, C);
      if constexpr (C + 3 < 16) FMAC_DPP(s.x[C + 3], nt, C);
      asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e1) : "v"(pb), "v"(y0));
      double n1 = 0, n2 = 0, n3 = 0, n13 = 0;
      if constexpr (C + 2 < 16) {
        if constexpr (C + 3 >= 16) asm volatile("s_nop 0");
        MOV_DPP(n1, s.x[C + 2], C + 1);
        MOV_DPP(n3, s.x[C + 1], C + 2);
      }
      asm volatile("v_fma_f64 %0, %1, %1, %1" : "=v"(e2) : "v"(e1));
      if constexpr (C + 2 < 16) {
        MOV_DPP(n2, s.x[C + 2], C + 2);
        asm volatile("v_mul_f64 %0, %1, %2" : "=v"(n13) : "v"(n1), "v"(n3));
      }
      asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(y) : "v"(y0), "v"(e2));
#pragma unroll
      for (int i = C + 4; i < 16; ++i) FMAC_DPP(s.x[i], nt, C);
#pragma unroll
      for (int q = 0; q <= (C >> 2); ++q) FMAC_DPP(s.e[q], nte, C);
      s.p = pb;
      s.ip = y;
      StepV5<C + 1>::run(s, n13, n2);
    }
  }
};

// ---------------------------------------------------------------- V6: the generated one-block-per-step form that
// potrf.hip uses (csrc/factor16_steps.h, tools/gen_factor16.py)
using Elim16 = St;
#include "../../inference-tools_amd/csrc/factor16_steps.h"
template <int C>
struct StepV6 {
  static __device__ inline void run(St& s) {
    if (s.k == C) s.myp = s.p;
    if constexpr (C < 15) {
      const double nt = -(s.x[C] * s.ip);
      const double nte = (s.k > C) ? nt : 0.0;
      double pb, y;
      ElimStepAsm<C>::run(s, nt, nte, pb, y);
      s.p = pb;
      s.ip = y;
      StepV6<C + 1>::run(s);
    }
  }
};

template <int V>
__device__ inline void eliminate(St& s) {
  s.p = bc<0>(s.x[0]);
  if (!(s.p > 0.0) || !(s.p < 1.79e308)) {
    s.badcol = 0;
    s.p = 1.0;
  }
  s.ip = rcp_newton(s.p);
  if constexpr (V == 1) StepV1<0>::run(s);
  if constexpr (V == 2) StepV2<0>::run(s);
  if constexpr (V == 3) {
    const double s1 = bc<0>(s.x[1]), s3 = bc<1>(s.x[0]), s2 = bc<1>(s.x[1]);
    StepV3<0>::run(s, s1 * s3, s2);
  }
  if constexpr (V == 4) StepV4<0>::run(s);
  if constexpr (V == 6) StepV6<0>::run(s);
  if constexpr (V == 5) {
    const double s1 = bc<0>(s.x[1]), s3 = bc<1>(s.x[0]), s2 = bc<1>(s.x[1]);
    StepV5<0>::run(s, s1 * s3, s2);
  }
  if constexpr (V >= 4) {  // pivots checked once: lane k holds p_k
    const unsigned long long bad = __ballot(!((s.myp > 0.0) && (s.myp < 1.79e308))) & 0xffffull;
    if (bad) s.badcol = __builtin_ctzll(bad);
  }
}

// out: L (16 x 16 row-major, lower), W = L^-1 (16 x 16 row-major), cycles per call
template <int V>
__global__ __launch_bounds__(64) void probe_kernel(const double* __restrict__ Ain, double* __restrict__ Lout,
                                                   double* __restrict__ Wout, unsigned long long* cyc, int reps) {
  __shared__ __attribute__((aligned(16))) double T[16 * 18];
  const int lane = threadIdx.x, k = lane & 15, g = lane >> 4;
  d4_t blk;
#pragma unroll
  for (int j = 0; j < 4; ++j) blk[j] = Ain[(4 * j + g) * 16 + k];  // the MFMA D layout the kernel hands over
  St s;
  unsigned long long t0 = 0, t1 = 0;
  for (int rep = 0; rep < reps + 1; ++rep) {
    if (rep == 1) t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(blk[j]));
    // D layout -> full columns in every lane, through LDS (the block is symmetric: column k = row k, contiguous)
#pragma unroll
    for (int j = 0; j < 4; ++j) T[(4 * j + g) * 18 + k] = blk[j];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const d2_t v = *reinterpret_cast<const d2_t*>(&T[k * 18 + 2 * m]);
      s.x[2 * m] = v[0];
      s.x[2 * m + 1] = v[1];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) s.e[q] = (4 * q + g == k) ? 1.0 : 0.0;
    s.k = k;
    s.g = g;
    s.myp = 1.0;
    s.badcol = -1;
    eliminate<V>(s);
#pragma unroll
    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(s.e[q]));
  }
  t1 = __builtin_amdgcn_s_memtime();
  const double rs = 1.0 / sqrt(s.myp);
#pragma unroll
  for (int q = 0; q < 4; ++q) Wout[k * 16 + 4 * q + g] = s.e[q] * rs;
  if (g == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double rsi = __shfl(rs, i, 64);
      Lout[k * 16 + i] = (i <= k) ? s.x[i] * rsi : 0.0;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) (void)__shfl(rs, i, 64);
  }
  if (lane == 0) {
    cyc[0] = (t1 - t0) / (unsigned long long)reps;
    cyc[1] = (unsigned long long)(s.badcol + 1);
  }
}

// ---------------------------------------------------------------- latencies: dependent chains / independent streams
template <int OP>
__global__ __launch_bounds__(64) void lat_kernel(double* io, unsigned long long* cyc, int n) {
  double a = io[threadIdx.x], b = io[64 + threadIdx.x];
  double r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = a + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));
      if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));
      if (OP == 2) asm volatile("v_rcp_f64 %0, %0\n s_nop 0" : "+v"(a));
      if (OP == 3) asm volatile("s_nop 1\n v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a));
      if (OP == 4) asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b));
      if (OP == 5) asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(r[u & 7]) : "v"(b));
      if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[u & 7]) : "v"(b));
      if (OP == 7) asm volatile("v_rsq_f64 %0, %0\n s_nop 0" : "+v"(a));
      if (OP == 8) asm volatile("v_rcp_f64 %0, %0" : "+v"(r[u & 7]));
      if (OP == 9) asm volatile("v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(r[u & 7]));
      if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(((int*)&a)[0]) : "v"(((int*)&b)[0]));
    }
  }
  d4_t m0 = {a, a, a, a}, m1 = m0, m2 = m0, m3 = m0;
  if (OP == 11 || OP == 12 || OP == 13) {
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (OP == 11) {  // one dependent chain
          m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m0, 0, 0, 0);
          m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m0, 0, 0, 0);
          m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m0, 0, 0, 0);
          m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m0, 0, 0, 0);
        } else if (OP == 12) {  // four independent accumulators
          m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m0, 0, 0, 0);
          m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m1, 0, 0, 0);
          m2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m2, 0, 0, 0);
          m3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m3, 0, 0, 0);
        } else {  // two
          m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m0, 0, 0, 0);
          m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m1, 0, 0, 0);
          m0 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m0, 0, 0, 0);
          m1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, m1, 0, 0, 0);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  a += m0[0] + m1[1] + m2[2] + m3[3];
#pragma unroll
  for (int i = 0; i < 8; ++i) a += r[i];
  io[128 + threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// accuracy of v_rcp_f64 / v_rsq_f64: max |1 - x y| (resp. |1 - x y^2|) over a sweep of mantissas
__global__ void acc_kernel(double* out) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  double mr = 0.0, ms = 0.0, m3 = 0.0;
  for (int i = 0; i < 4096; ++i) {
    const double x = (1.0 + (tid * 4096.0 + i) / (65536.0 * 4096.0) * 3.0) * 1.37e-3;
    const double y = __builtin_amdgcn_rcp(x);
    mr = fmax(mr, fabs(fma(-x, y, 1.0)));
    const double e = fma(-x, y, 1.0);
    const double y3 = fma(y, fma(e, e, e), y);  // cubic step
    m3 = fmax(m3, fabs(fma(-x, y3, 1.0)));
    const double z = __builtin_amdgcn_rsq(x);
    ms = fmax(ms, fabs(fma(-x * z, z, 1.0)));
  }
  out[tid * 3] = mr;
  out[tid * 3 + 1] = ms;
  out[tid * 3 + 2] = m3;
}

#define CK(x)                                                              \
  do {                                                                     \
    hipError_t e_ = (x);                                                   \
    if (e_ != hipSuccess) {                                                \
      printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);     \
      exit(1);                                                             \
    }                                                                      \
  } while (0)

int main() {
  const int n = 16;
  std::vector<double> A(n * n), L(n * n, 0.0), W(n * n, 0.0);
  srand(7);
  // SPD test block: squared-exponential covariance of 16 points + noise (the kind of block the chain sees)
  double pts[16];
  for (int i = 0; i < n; ++i) pts[i] = rand() / (double)RAND_MAX;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) A[i * n + j] = exp(-0.5 * (pts[i] - pts[j]) * (pts[i] - pts[j]) / 0.09) + (i == j ? 0.01 : 0.0);
  // host reference
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    L[j * n + j] = sqrt(d);
    for (int i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = v / L[j * n + j];
    }
  }
  for (int j = 0; j < n; ++j) {  // W = L^-1 by forward substitution on the identity
    for (int i = 0; i < n; ++i) {
      double v = (i == j) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) v -= L[i * n + k] * W[k * n + j];
      W[i * n + j] = v / L[i * n + i];
    }
  }
  double *dA, *dL, *dW, *dio;
  unsigned long long* dc;
  CK(hipMalloc(&dA, n * n * 8));
  CK(hipMalloc(&dL, n * n * 8));
  CK(hipMalloc(&dW, n * n * 8));
  CK(hipMalloc(&dc, 64));
  CK(hipMalloc(&dio, 192 * 8));
  CK(hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice));
  auto run = [&](int v) {
    CK(hipMemset(dL, 0, n * n * 8));
    CK(hipMemset(dW, 0, n * n * 8));
    for (int pass = 0; pass < 2; ++pass) {
      if (v == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(1), dim3(64), 0, 0, dA, dL, dW, dc, 2000);
      if (v == 2) hipLaunchKernelGGL(probe_kernel<2>, dim3(1), dim3(64), 0, 0, dA, dL, dW, dc, 2000);
      if (v == 3) hipLaunchKernelGGL(probe_kernel<3>, dim3(1), dim3(64), 0, 0, dA, dL, dW, dc, 2000);
      if (v == 4) hipLaunchKernelGGL(probe_kernel<4>, dim3(1), dim3(64), 0, 0, dA, dL, dW, dc, 2000);
      if (v == 5) hipLaunchKernelGGL(probe_kernel<5>, dim3(1), dim3(64), 0, 0, dA, dL, dW, dc, 2000);
      if (v == 6) hipLaunchKernelGGL(probe_kernel<6>, dim3(1), dim3(64), 0, 0, dA, dL, dW, dc, 2000);
      CK(hipDeviceSynchronize());
    }
    std::vector<double> gl(n * n), gw(n * n);
    unsigned long long c[2];
    CK(hipMemcpy(gl.data(), dL, n * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gw.data(), dW, n * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost));
    double el = 0, ew = 0, ml = 0, mw = 0;
    for (int i = 0; i < n * n; ++i) {
      el = fmax(el, fabs(gl[i] - L[i]));
      ml = fmax(ml, fabs(L[i]));
      ew = fmax(ew, fabs(gw[i] - W[i]));
      mw = fmax(mw, fabs(W[i]));
    }
    if (getenv("F16_DUMP") && (v == 6))
      for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) printf("%9.2e ", gw[i * n + j] - W[i * n + j]);
        printf("\n");
      }
    printf("V%d: %llu cycles per 16x16 block (prologue through LDS included), info %llu, rel err L %.2e, W %.2e\n", v,
           c[0], c[1], el / ml, ew / mw);
  };
  run(1);
  run(2);
  run(3);
  run(4);
  run(5);
  run(6);
  std::vector<double> io(192, 1.000001);
  for (int i = 64; i < 128; ++i) io[i] = 1e-9;
  CK(hipMemcpy(dio, io.data(), 192 * 8, hipMemcpyHostToDevice));
  const char* names[] = {"v_fma_f64 dependent", "v_mul_f64 dependent", "v_rcp_f64 dependent (+s_nop 0)",
                         "v_mov_b64_dpp dependent (+s_nop 1)", "v_fmac_f64_dpp dependent (+s_nop 1)",
                         "v_fmac_f64_dpp 8 independent", "v_fma_f64 8 independent", "v_rsq_f64 dependent (+s_nop 0)",
                         "v_rcp_f64 8 independent", "v_mov_b64_dpp 8 independent", "v_cndmask_b32 dependent",
                         "v_mfma_f64_16x16x4 one dependent chain", "v_mfma_f64_16x16x4 four accumulators", "v_mfma_f64_16x16x4 two accumulators"};
  for (int op = 0; op <= 13; ++op) {
    const int it = 256;
    for (int pass = 0; pass < 2; ++pass) {
      switch (op) {
        case 0: hipLaunchKernelGGL(lat_kernel<0>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 1: hipLaunchKernelGGL(lat_kernel<1>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 2: hipLaunchKernelGGL(lat_kernel<2>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 3: hipLaunchKernelGGL(lat_kernel<3>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 4: hipLaunchKernelGGL(lat_kernel<4>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 5: hipLaunchKernelGGL(lat_kernel<5>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 6: hipLaunchKernelGGL(lat_kernel<6>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 7: hipLaunchKernelGGL(lat_kernel<7>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 8: hipLaunchKernelGGL(lat_kernel<8>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 9: hipLaunchKernelGGL(lat_kernel<9>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 10: hipLaunchKernelGGL(lat_kernel<10>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 11: hipLaunchKernelGGL(lat_kernel<11>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 12: hipLaunchKernelGGL(lat_kernel<12>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
        case 13: hipLaunchKernelGGL(lat_kernel<13>, dim3(1), dim3(64), 0, 0, dio, dc, it); break;
      }
      CK(hipDeviceSynchronize());
    }
    unsigned long long c;
    CK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost));
    printf("%-40s %.2f cycles per instruction\n", names[op], (double)c / (it * 16.0));
  }
  double* dacc;
  CK(hipMalloc(&dacc, 65536 * 3 * 8));
  hipLaunchKernelGGL(acc_kernel, dim3(256), dim3(256), 0, 0, dacc);
  CK(hipDeviceSynchronize());
  std::vector<double> acc(65536 * 3);
  CK(hipMemcpy(acc.data(), dacc, acc.size() * 8, hipMemcpyDeviceToHost));
  double mr = 0, ms = 0, m3 = 0;
  for (int i = 0; i < 65536; ++i) {
    mr = fmax(mr, acc[3 * i]);
    ms = fmax(ms, acc[3 * i + 1]);
    m3 = fmax(m3, acc[3 * i + 2]);
  }
  printf("v_rcp_f64: max |1 - x y| = %.3e (2^%.1f); after one cubic step %.3e; v_rsq_f64: max |1 - x y^2| = %.3e (2^%.1f)\n", mr,
         log2(mr), m3, ms, log2(ms));
  return 0;
}
