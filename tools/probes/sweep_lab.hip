// Playground for the single-launch triangular sweeps: variants of the two kernels selected by template flags, each timed
// with HIP events (min of 10 launches, the fill outside the timed region) and check-summed - the product kernels of
// solve.hip ("product") run beside them.  Variants that survive are ported into solve.hip by hand; nothing here ships.
//   bash tools/probes/sweep_lab.sh [n ...]
// Flags: NOBAR1 (fold without the first barrier), DPP (lane exchanges), UPFRONT (LDS reads issued before their use), PAD (the
// conflict-free layouts - the form solve.hip ships), BACKOFF (far pollers sleep longer), WARM (a dummy pass through the
// tail code), SKIP_FOLD / SKIP_SOLVE (ablations: wrong results by design), RING4 / RING8 (several polls in flight: the
// check sums DIFFER - the compiler may copy a ring register between the load and its wait, and loads still in flight at
// the exit are only drained, not owned; kept as the record of the timing experiment, not as code to port), and
// fwd2_lab (two block rows per workgroup: 179 spilled registers); POLL1 / POLL2W (one / two waves poll and hand v_j on
// through LDS - POLL2W is what solve.hip ships since R6.13), PUB2 / PUB1 (the step's 128 values published by two waves / one wave
// instead of by sixteen lanes of each of the eight: slower by the extra barrier).  profiles/HISTORY.md R6.8, R6.13.
#include "../../inference-tools_amd/csrc/solve.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
// Two neighbouring elements in ONE memory round trip: both loads are issued before either is looked at (measured flat
// against two polls one after the other; the product kernel now polls with two waves, one element per lane).
__device__ inline void flow_poll2(const double* p, int* err, bool& dead, double& a, double& b) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  unsigned long long x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while ((x == FLOW_SENTINEL || y == FLOW_SENTINEL) && !dead) {
    __builtin_amdgcn_s_sleep(1);
    const unsigned long long x2 = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long y2 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x = x2;
    y = y2;
    ++spins;
    if ((spins & 4095) == 0 && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) dead = true;
    if (spins > FLOW_SPIN_LIMIT) {
      if (err) atomicCAS(err, 0, GPMI_ERR_INTERNAL);
      dead = true;
    }
  }
  a = (x == FLOW_SENTINEL) ? 0.0 : __longlong_as_double((long long)x);
  b = (y == FLOW_SENTINEL) ? 0.0 : __longlong_as_double((long long)y);
}

enum { F_NOBAR1 = 1, F_DPP = 2, F_UPFRONT = 4, F_SKIP_FOLD = 8, F_SKIP_SOLVE = 16, F_BACKOFF = 32, F_UPLOOP = 64, F_RING4 = 128, F_RING8 = 256, F_WARM = 512, F_PAD = 1024, F_POLL1 = 2048, F_POLL2W = 4096, F_PUB2 = 8192, F_PUB1 = 16384 };

template <int F>
__device__ inline double poll_far(const double* p, int* err, bool& dead, int far) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  unsigned long long bits = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while (bits == FLOW_SENTINEL && !dead) {
    __builtin_amdgcn_s_sleep(1);
    if (F & F_BACKOFF)
      for (int w = 0; w < far; ++w) __builtin_amdgcn_s_sleep(8);
    bits = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (++spins > FLOW_SPIN_LIMIT) {
      if (err) atomicCAS(err, 0, GPMI_ERR_INTERNAL);
      dead = true;
    }
  }
  return (bits == FLOW_SENTINEL) ? 0.0 : __longlong_as_double((long long)bits);
}

template <int F>
__device__ inline void poll2_far(const double* p, int* err, bool& dead, double& a, double& b, int far) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  unsigned long long x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while ((x == FLOW_SENTINEL || y == FLOW_SENTINEL) && !dead) {
    __builtin_amdgcn_s_sleep(1);
    if (F & F_BACKOFF)
      for (int w = 0; w < far; ++w) __builtin_amdgcn_s_sleep(8);
    x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (++spins > FLOW_SPIN_LIMIT) {
      if (err) atomicCAS(err, 0, GPMI_ERR_INTERNAL);
      dead = true;
    }
  }
  a = (x == FLOW_SENTINEL) ? 0.0 : __longlong_as_double((long long)x);
  b = (y == FLOW_SENTINEL) ? 0.0 : __longlong_as_double((long long)y);
}

// ---- ring polling: D loads of the same address in flight, issued a fraction of the round trip apart, so a value is seen
// a fraction of a round trip after it became visible (a blocking poll loop sees it up to a whole round trip late)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
#define RING_ISSUE16(r, p) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(r) : "v"(p) : "memory")
#define RING_ISSUE8(r, p) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(r) : "v"(p) : "memory")
#define RING_WAIT(n, r) asm volatile("s_waitcnt vmcnt(" #n ")" : "+v"(r) : : "memory")
__device__ inline bool ring_done(u64x2 g) { return __builtin_amdgcn_ballot_w64(g.x == FLOW_SENTINEL || g.y == FLOW_SENTINEL) == 0; }
__device__ inline bool ring_done(unsigned long long g) { return __builtin_amdgcn_ballot_w64(g == FLOW_SENTINEL) == 0; }

#define RING_STEP(n, r, ISSUE)  \
  RING_WAIT(n, r);              \
  got = r;                      \
  if (ring_done(got)) break;    \
  ISSUE(r, p);

template <typename T, int D>
__device__ inline T ring_poll(const void* p, int* err, bool& dead) {
  T r0, r1, r2, r3, r4, r5, r6, r7, got;
  constexpr bool wide = sizeof(T) == 16;
#define ISS(r, p)              \
  do {                         \
    if constexpr (wide)        \
      RING_ISSUE16(r, p);      \
    else                       \
      RING_ISSUE8(r, p);       \
  } while (0)
  if constexpr (D == 4) {
    ISS(r0, p); __builtin_amdgcn_s_sleep(6); ISS(r1, p); __builtin_amdgcn_s_sleep(6); ISS(r2, p); __builtin_amdgcn_s_sleep(6); ISS(r3, p);
    for (int spins = 0;; ++spins) {
      RING_STEP(3, r0, ISS) RING_STEP(3, r1, ISS) RING_STEP(3, r2, ISS) RING_STEP(3, r3, ISS)
      if (spins > FLOW_SPIN_LIMIT) { if (err) atomicCAS(err, 0, GPMI_ERR_INTERNAL); dead = true; break; }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
  } else {
    ISS(r0, p); __builtin_amdgcn_s_sleep(3); ISS(r1, p); __builtin_amdgcn_s_sleep(3); ISS(r2, p); __builtin_amdgcn_s_sleep(3); ISS(r3, p);
    __builtin_amdgcn_s_sleep(3); ISS(r4, p); __builtin_amdgcn_s_sleep(3); ISS(r5, p); __builtin_amdgcn_s_sleep(3); ISS(r6, p);
    __builtin_amdgcn_s_sleep(3); ISS(r7, p);
    for (int spins = 0;; ++spins) {
      RING_STEP(7, r0, ISS) RING_STEP(7, r1, ISS) RING_STEP(7, r2, ISS) RING_STEP(7, r3, ISS)
      RING_STEP(7, r4, ISS) RING_STEP(7, r5, ISS) RING_STEP(7, r6, ISS) RING_STEP(7, r7, ISS)
      if (spins > FLOW_SPIN_LIMIT) { if (err) atomicCAS(err, 0, GPMI_ERR_INTERNAL); dead = true; break; }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : : "memory");
  }
#undef ISS
  return got;
}
__device__ inline double ring_value(unsigned long long bits) { return bits == FLOW_SENTINEL ? 0.0 : __longlong_as_double((long long)bits); }

template <int F>
__global__ __launch_bounds__(FLOW_THREADS) void fwd_lab(const double* __restrict__ L, int64_t ld, const double* __restrict__ invD,
                                                        const double* __restrict__ r, double* __restrict__ v, int* __restrict__ err, int nt) {
  const int k = blockIdx.x;
  invD += (int64_t)k * NB * NB;
  __shared__ __attribute__((aligned(16))) double part[NB][66];
  __shared__ __attribute__((aligned(16))) double u[NB + 8];
  __shared__ __attribute__((aligned(16))) double vin[2][NB];  // F_POLL1 / F_POLL2W: v_j through LDS, one / two waves poll
  __shared__ __attribute__((aligned(16))) double vout[NB];    // F_PUB1 / F_PUB2: v_k on its way to the publishing wave(s)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row4 = tid >> 2, q4 = tid & 3;
  // F_PAD layout: a row is 32 slots of 16 B (+1 pad); the partial of source lane 16 q + c sits in slot 8 q + ((c/2 + 4 (q/2)) % 8)
  const int wq = lane >> 4, wc = lane & 15;
  const int wpos = (F & F_PAD) ? 2 * (8 * wq + (((wc >> 1) + 4 * (wq >> 1)) & 7)) + (wc & 1) : lane;
  double xi[32];
  {
    const double* p = invD + row4 * NB + q4 * 32;
#pragma unroll
    for (int c = 0; c < 32; c += 2) {
      const d2_t a = *reinterpret_cast<const d2_t*>(p + c);
      xi[c] = a[0];
      xi[c + 1] = a[1];
    }
  }
  const double rk = r[(int64_t)k * NB + row4];
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.0;
  const int nmain = k;
#pragma nounroll
  for (int pass = (F & F_WARM) ? 0 : 1; pass < 2; ++pass) {
  if (pass == 1 && nmain > 0) {
    const double* base = L + (int64_t)(k * NB + wave * 16) * ld + 2 * lane;
    d2_t ha[8], hb[8];
    bool dead = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) ha[i] = *reinterpret_cast<const d2_t*>(base + (int64_t)i * ld);
    for (int j = 0; j < nmain; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) hb[i] = *reinterpret_cast<const d2_t*>(base + (int64_t)(8 + i) * ld + (int64_t)j * NB);
      double v0, v1;
      if (F & (F_POLL1 | F_POLL2W)) {
        double* vj = vin[j & 1];
        if (F & F_POLL1) {
          if (wave == 0) {
            double a, b;
            poll2_far<F>(v + (int64_t)j * NB + 2 * lane, err, dead, a, b, 0);
            *reinterpret_cast<d2_t*>(&vj[2 * lane]) = d2_t{a, b};
          }
        } else if (tid < NB) {
          vj[tid] = poll_far<F>(v + (int64_t)j * NB + tid, err, dead, 0);
        }
        __syncthreads();
        const d2_t x = *reinterpret_cast<const d2_t*>(&vj[2 * lane]);
        v0 = x[0];
        v1 = x[1];
      } else if ((F & (F_RING4 | F_RING8)) && j == nmain - 1 && !dead) {
        const u64x2 g = ring_poll<u64x2, (F & F_RING8) ? 8 : 4>(v + (int64_t)j * NB + 2 * lane, err, dead);
        v0 = ring_value(g.x);
        v1 = ring_value(g.y);
      } else {
        poll2_far<F>(v + (int64_t)j * NB + 2 * lane, err, dead, v0, v1, min(nmain - 1 - j, 15));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fma(ha[i][0], v0, fma(ha[i][1], v1, acc[i]));
      if (j + 1 < nmain) {
#pragma unroll
        for (int i = 0; i < 8; ++i) ha[i] = *reinterpret_cast<const d2_t*>(base + (int64_t)i * ld + (int64_t)(j + 1) * NB);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[8 + i] = fma(hb[i][0], v0, fma(hb[i][1], v1, acc[8 + i]));
    }
  }
  if (F & F_SKIP_FOLD) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (q4 == 0) u[(F & F_PAD) ? row4 + 2 * (row4 >> 5) : row4] = rk - s;
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) part[wave * 16 + i][wpos] = acc[i];
    if (F & F_NOBAR1) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();
    }
    double s = 0.0;
    if (F & F_PAD) {
      double pv[16];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const d2_t x = *reinterpret_cast<const d2_t*>(&part[row4][2 * (8 * q4 + ((t + 4 * (q4 >> 1)) & 7))]);
        pv[2 * t] = x[0];
        pv[2 * t + 1] = x[1];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 16; ++c) s += pv[c];
    } else if (F & F_UPFRONT) {
      double pv[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) pv[c] = part[row4][q4 * 16 + c];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 16; ++c) s += pv[c];
    } else {
#pragma unroll
      for (int c = 0; c < 16; ++c) s += part[row4][q4 * 16 + c];
    }
    if (F & F_DPP) {
      s += quad_swap<QUAD_XOR1>(s);
      s += quad_swap<QUAD_XOR2>(s);
    } else {
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
    }
    if (q4 == 0) u[(F & F_PAD) ? row4 + 2 * (row4 >> 5) : row4] = rk - s;
  }
  __syncthreads();
  if (F & F_SKIP_SOLVE) {
    if (q4 == 0 && pass == 1) flow_publish(v + (int64_t)k * NB + row4, u[(F & F_PAD) ? row4 + 2 * (row4 >> 5) : row4] * xi[0]);
    continue;
  }
  {
    double s = 0.0;
    if (F & F_UPFRONT) {
      double uv[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) uv[c] = u[q4 * ((F & F_PAD) ? 34 : 32) + c];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 32; ++c) s = fma(xi[c], uv[c], s);
    } else {
#pragma unroll
      for (int c = 0; c < 32; ++c) s = fma(xi[c], u[q4 * ((F & F_PAD) ? 34 : 32) + c], s);
    }
    if (F & F_DPP) {
      s += quad_swap<QUAD_XOR1>(s);
      s += quad_swap<QUAD_XOR2>(s);
    } else {
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
    }
    if (F & (F_PUB2 | F_PUB1)) {
      // publish from two waves / one wave instead of from sixteen lanes of each of the eight
      double* vo = vout;
      if (q4 == 0) vo[row4] = s;
      __syncthreads();
      if (pass == 1) {
        if ((F & F_PUB2) && tid < NB) flow_publish(v + (int64_t)k * NB + tid, vo[tid]);
        if ((F & F_PUB1) && tid < 64) {
          flow_publish(v + (int64_t)k * NB + 2 * tid, vo[2 * tid]);
          flow_publish(v + (int64_t)k * NB + 2 * tid + 1, vo[2 * tid + 1]);
        }
      }
    } else if (q4 == 0 && pass == 1) flow_publish(v + (int64_t)k * NB + row4, s);
  }
  }
}

// backward: F_DPP = quad layout of the last step (no fourth barrier), F_UPLOOP = a_j reads up front in the loop,
// F_UPFRONT = fold / u reads up front, F_BACKOFF
template <int F>
__global__ __launch_bounds__(FLOW_THREADS) void bwd_lab(const double* __restrict__ L, int64_t ld, const double* __restrict__ invD,
                                                        const double* __restrict__ w, double* __restrict__ a, int* __restrict__ err, int nt) {
  const int step = blockIdx.x;
  const int k = nt - 1 - step;
  invD += (int64_t)k * NB * NB;
  __shared__ __attribute__((aligned(16))) double part[8][NB];
  __shared__ __attribute__((aligned(16))) double u[NB + 8];
  __shared__ __attribute__((aligned(16))) double ain[2][NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = tid & 127, rg = tid >> 7;
  const int colq = tid >> 2, q4 = tid & 3;
  constexpr int ustride = (F & F_PAD) ? 34 : 32;
  double xi[32];
  {
    const double* p = (F & F_DPP) ? invD + (q4 * 32) * NB + colq : invD + (rg * 32) * NB + col;
#pragma unroll
    for (int i = 0; i < 32; ++i) xi[i] = p[i * NB];
  }
  const double wk = w[(int64_t)k * NB + col];
  d2_t acc = d2_t{0.0, 0.0};
  const int nmain = nt - k - 1;
#pragma nounroll
  for (int pass = (F & F_WARM) ? 0 : 1; pass < 2; ++pass) {
  if (pass == 1 && nmain > 0) {
    const double* base = L + (int64_t)wave * ld + (int64_t)k * NB + 2 * lane;
    d2_t ha[8], hb[8];
    bool dead = false;
    {
      const double* pj = base + (int64_t)(nt - 1) * NB * ld;
#pragma unroll
      for (int i = 0; i < 8; ++i) ha[i] = *reinterpret_cast<const d2_t*>(pj + (int64_t)(8 * i) * ld);
    }
    for (int t = 0; t < nmain; ++t) {
      const int j = nt - 1 - t;
      {
        const double* pj = base + (int64_t)j * NB * ld;
#pragma unroll
        for (int i = 0; i < 8; ++i) hb[i] = *reinterpret_cast<const d2_t*>(pj + (int64_t)(64 + 8 * i) * ld);
      }
      double* aj = ain[t & 1];
      if (tid < NB) {
        if ((F & (F_RING4 | F_RING8)) && t == nmain - 1 && !dead)
          aj[tid] = ring_value(ring_poll<unsigned long long, (F & F_RING8) ? 8 : 4>(a + (int64_t)j * NB + tid, err, dead));
        else
          aj[tid] = poll_far<F>(a + (int64_t)j * NB + tid, err, dead, min(nmain - 1 - t, 15));
      }
      __syncthreads();
      if (F & F_UPLOOP) {
        double xa[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) xa[i] = aj[wave + 8 * i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[0] = fma(ha[i][0], xa[i], acc[0]);
          acc[1] = fma(ha[i][1], xa[i], acc[1]);
        }
        if (t + 1 < nmain) {
          const double* pj = base + (int64_t)(j - 1) * NB * ld;
#pragma unroll
          for (int i = 0; i < 8; ++i) ha[i] = *reinterpret_cast<const d2_t*>(pj + (int64_t)(8 * i) * ld);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          acc[0] = fma(hb[i][0], xa[8 + i], acc[0]);
          acc[1] = fma(hb[i][1], xa[8 + i], acc[1]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const double x = aj[wave + 8 * i];
          acc[0] = fma(ha[i][0], x, acc[0]);
          acc[1] = fma(ha[i][1], x, acc[1]);
        }
        if (t + 1 < nmain) {
          const double* pj = base + (int64_t)(j - 1) * NB * ld;
#pragma unroll
          for (int i = 0; i < 8; ++i) ha[i] = *reinterpret_cast<const d2_t*>(pj + (int64_t)(8 * i) * ld);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const double x = aj[wave + 64 + 8 * i];
          acc[0] = fma(hb[i][0], x, acc[0]);
          acc[1] = fma(hb[i][1], x, acc[1]);
        }
      }
    }
  }
  part[wave][2 * lane] = acc[0];
  part[wave][2 * lane + 1] = acc[1];
  __syncthreads();
  if (tid < NB) {
    double s = 0.0;
    if (F & F_UPFRONT) {
      double pv[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) pv[g] = part[g][tid];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < 8; ++g) s += pv[g];
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) s += part[g][tid];
    }
    u[(F & F_PAD) ? tid + 2 * (tid >> 5) : tid] = wk - s;
  }
  __syncthreads();
  if (F & F_DPP) {
    double s = 0.0;
    if (F & F_UPFRONT) {
      double uv[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) uv[i] = u[q4 * ustride + i];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 32; ++i) s = fma(xi[i], uv[i], s);
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) s = fma(xi[i], u[q4 * ustride + i], s);
    }
    s += quad_swap<QUAD_XOR1>(s);
    s += quad_swap<QUAD_XOR2>(s);
    if (q4 == 0 && pass == 1) flow_publish(a + (int64_t)k * NB + colq, s);
  } else {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s = fma(xi[i], u[rg * ustride + i], s);
    part[4 + rg][col] = s;
    __syncthreads();
    if (tid < NB && pass == 1)
      flow_publish(a + (int64_t)k * NB + tid, (part[4][tid] + part[5][tid]) + (part[6][tid] + part[7][tid]));
    __syncthreads();  // (the next pass writes part[] again)
  }
  }
}

// ---- two block rows per workgroup: 256 rows per cross-workgroup hand-off.  The workgroup is the virtual workgroups 2p and
// 2p + 1 of the 128-row kernel (same lane partials in the same order, same fold, same solve: bit-identical); the second one's
// last input, v_{2p}, is handed over through LDS instead of through memory.
__global__ __launch_bounds__(FLOW_THREADS) void fwd2_lab(const double* __restrict__ L, int64_t ld, const double* __restrict__ invD,
                                                         const double* __restrict__ r, double* __restrict__ v, int* __restrict__ err, int nt) {
  const int p = blockIdx.x, kA = 2 * p, kB = 2 * p + 1;
  const bool hasB = kB < nt;
  __shared__ __attribute__((aligned(16))) double part[NB][66];
  __shared__ __attribute__((aligned(16))) double u[NB + 8];
  __shared__ __attribute__((aligned(16))) double vl[NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row4 = tid >> 2, q4 = tid & 3;
  const int wpos = 2 * (8 * (lane >> 4) + ((((lane & 15) >> 1) + 4 * (lane >> 5)) & 7)) + (lane & 1);
  double xiA[32], xiB[32];
  {
    const double* pa = invD + (int64_t)kA * NB * NB + row4 * NB + q4 * 32;
    const double* pb = invD + (int64_t)(hasB ? kB : kA) * NB * NB + row4 * NB + q4 * 32;
#pragma unroll
    for (int c = 0; c < 32; c += 2) {
      const d2_t a = *reinterpret_cast<const d2_t*>(pa + c);
      xiA[c] = a[0];
      xiA[c + 1] = a[1];
      const d2_t b = *reinterpret_cast<const d2_t*>(pb + c);
      xiB[c] = b[0];
      xiB[c + 1] = b[1];
    }
  }
  const double rkA = r[(int64_t)kA * NB + row4];
  const double rkB = hasB ? r[(int64_t)kB * NB + row4] : 0.0;
  double accA[16], accB[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) accA[i] = accB[i] = 0.0;
  const int nA = kA;
  const double* baseA = L + (int64_t)(kA * NB + wave * 16) * ld + 2 * lane;
  const double* baseB = baseA + (hasB ? (int64_t)NB * ld : 0);
  d2_t a0[8], a1[8], b0[8], b1[8];
  bool dead = false;
  // first tiles: block 0 of both halves (for p = 0, B's only block is the diagonal-adjacent one, block kA = 0)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    b0[i] = *reinterpret_cast<const d2_t*>(baseB + (int64_t)i * ld);
    b1[i] = *reinterpret_cast<const d2_t*>(baseB + (int64_t)(8 + i) * ld);
  }
  if (nA > 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a0[i] = *reinterpret_cast<const d2_t*>(baseA + (int64_t)i * ld);
      a1[i] = *reinterpret_cast<const d2_t*>(baseA + (int64_t)(8 + i) * ld);
    }
    for (int j = 0; j < nA; ++j) {
      double v0, v1;
      flow_poll2(v + (int64_t)j * NB + 2 * lane, err, dead, v0, v1);
#pragma unroll
      for (int i = 0; i < 8; ++i) accA[i] = fma(a0[i][0], v0, fma(a0[i][1], v1, accA[i]));
      if (j + 1 < nA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a0[i] = *reinterpret_cast<const d2_t*>(baseA + (int64_t)i * ld + (int64_t)(j + 1) * NB);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) accA[8 + i] = fma(a1[i][0], v0, fma(a1[i][1], v1, accA[8 + i]));
      if (j + 1 < nA) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a1[i] = *reinterpret_cast<const d2_t*>(baseA + (int64_t)(8 + i) * ld + (int64_t)(j + 1) * NB);
      }
      if (hasB) {
#pragma unroll
        for (int i = 0; i < 8; ++i) accB[i] = fma(b0[i][0], v0, fma(b0[i][1], v1, accB[i]));
#pragma unroll
        for (int i = 0; i < 8; ++i) b0[i] = *reinterpret_cast<const d2_t*>(baseB + (int64_t)i * ld + (int64_t)(j + 1) * NB);
#pragma unroll
        for (int i = 0; i < 8; ++i) accB[8 + i] = fma(b1[i][0], v0, fma(b1[i][1], v1, accB[8 + i]));
#pragma unroll
        for (int i = 0; i < 8; ++i) b1[i] = *reinterpret_cast<const d2_t*>(baseB + (int64_t)(8 + i) * ld + (int64_t)(j + 1) * NB);
      }
    }
  }
  // ---- stage A
  double sA;
  {
#pragma unroll
    for (int i = 0; i < 16; ++i) part[wave * 16 + i][wpos] = accA[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double pv[16];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const d2_t x = *reinterpret_cast<const d2_t*>(&part[row4][2 * (8 * q4 + ((t + 4 * (q4 >> 1)) & 7))]);
      pv[2 * t] = x[0];
      pv[2 * t + 1] = x[1];
    }
    __builtin_amdgcn_sched_barrier(0);
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += pv[c];
    s += quad_swap<QUAD_XOR1>(s);
    s += quad_swap<QUAD_XOR2>(s);
    if (q4 == 0) u[row4 + 2 * (row4 >> 5)] = rkA - s;
    __syncthreads();
    double uv[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) uv[c] = u[q4 * 34 + c];
    __builtin_amdgcn_sched_barrier(0);
    s = 0.0;
#pragma unroll
    for (int c = 0; c < 32; ++c) s = fma(xiA[c], uv[c], s);
    s += quad_swap<QUAD_XOR1>(s);
    s += quad_swap<QUAD_XOR2>(s);
    sA = s;
    if (q4 == 0) {
      flow_publish(v + (int64_t)kA * NB + row4, sA);
      vl[row4] = sA;
    }
  }
  if (!hasB) return;
  __syncthreads();  // v_A in LDS; every read of u by stage A is done
  {
    const d2_t va = *reinterpret_cast<const d2_t*>(&vl[2 * lane]);
#pragma unroll
    for (int i = 0; i < 8; ++i) accB[i] = fma(b0[i][0], va[0], fma(b0[i][1], va[1], accB[i]));
#pragma unroll
    for (int i = 0; i < 8; ++i) accB[8 + i] = fma(b1[i][0], va[0], fma(b1[i][1], va[1], accB[8 + i]));
#pragma unroll
    for (int i = 0; i < 16; ++i) part[wave * 16 + i][wpos] = accB[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double pv[16];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const d2_t x = *reinterpret_cast<const d2_t*>(&part[row4][2 * (8 * q4 + ((t + 4 * (q4 >> 1)) & 7))]);
      pv[2 * t] = x[0];
      pv[2 * t + 1] = x[1];
    }
    __builtin_amdgcn_sched_barrier(0);
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += pv[c];
    s += quad_swap<QUAD_XOR1>(s);
    s += quad_swap<QUAD_XOR2>(s);
    if (q4 == 0) u[row4 + 2 * (row4 >> 5)] = rkB - s;
    __syncthreads();
    double uv[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) uv[c] = u[q4 * 34 + c];
    __builtin_amdgcn_sched_barrier(0);
    s = 0.0;
#pragma unroll
    for (int c = 0; c < 32; ++c) s = fma(xiB[c], uv[c], s);
    s += quad_swap<QUAD_XOR1>(s);
    s += quad_swap<QUAD_XOR2>(s);
    if (q4 == 0) flow_publish(v + (int64_t)kB * NB + row4, s);
  }
}

struct Bufs {
  double *L, *invD, *r, *v;
  int* err;
  int64_t n, ld;
  int nt;
  hipStream_t s;
};

template <typename Launch>
void run(const char* name, const Bufs& b, Launch launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 10; ++rep) {
    hipLaunchKernelGGL(flow_fill_kernel, dim3((unsigned)((b.n + 255) / 256), 1, 1), dim3(256), 0, b.s, b.v, b.n, (int64_t)0);
    (void)hipEventRecord(e0, b.s);
    launch();
    (void)hipEventRecord(e1, b.s);
    (void)hipStreamSynchronize(b.s);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  std::vector<double> h((size_t)b.n);
  (void)hipMemcpy(h.data(), b.v, sizeof(double) * b.n, hipMemcpyDeviceToHost);
  double sum = 0;
  for (double x : h) sum += x;
  int herr = 0;
  (void)hipMemcpy(&herr, b.err, sizeof(int), hipMemcpyDeviceToHost);
  printf("  %-44s %8.1f us  %.3f us/step  checksum %a%s\n", name, best * 1e3, best * 1e3 / b.nt, sum, herr ? "  ERR" : "");
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
}

#define FWD(name, F) run("forward  " name, b, [&] { hipLaunchKernelGGL(fwd_lab<F>, dim3(b.nt), dim3(FLOW_THREADS), 0, b.s, b.L, b.ld, b.invD, b.r, b.v, b.err, b.nt); })
#define BWD(name, F) run("backward " name, b, [&] { hipLaunchKernelGGL(bwd_lab<F>, dim3(b.nt), dim3(FLOW_THREADS), 0, b.s, b.L, b.ld, b.invD, b.r, b.v, b.err, b.nt); })
}  // namespace

int main(int argc, char** argv) {
  for (int a = 1; a < (argc > 1 ? argc : 2); ++a) {
    Bufs b;
    b.n = argc > 1 ? atoll(argv[a]) : 16384;
    b.nt = (int)(b.n / NB);
    b.ld = b.n;
    (void)hipMalloc(&b.L, sizeof(double) * b.n * b.ld);
    (void)hipMalloc(&b.invD, sizeof(double) * b.nt * NB * NB);
    (void)hipMalloc(&b.r, sizeof(double) * b.n);
    (void)hipMalloc(&b.v, sizeof(double) * b.n);
    (void)hipMalloc(&b.err, sizeof(int));
    (void)hipMemset(b.err, 0, sizeof(int));
    {
      std::vector<double> hl((size_t)b.n * 64);
      for (size_t i = 0; i < hl.size(); ++i) hl[i] = 1e-6 * (double)((i * 2654435761u) % 1000);
      for (int64_t off = 0; off < b.n * b.ld; off += (int64_t)hl.size())
        (void)hipMemcpy(b.L + off, hl.data(), sizeof(double) * hl.size(), hipMemcpyHostToDevice);
      for (int64_t off = 0; off < (int64_t)b.nt * NB * NB; off += (int64_t)hl.size())
        (void)hipMemcpy(b.invD + off, hl.data(), sizeof(double) * std::min<int64_t>((int64_t)hl.size(), (int64_t)b.nt * NB * NB - off),
                        hipMemcpyHostToDevice);
      (void)hipMemcpy(b.r, hl.data(), sizeof(double) * b.n, hipMemcpyHostToDevice);
    }
    (void)hipStreamCreate(&b.s);
    printf("n = %lld (%d steps)\n", (long long)b.n, b.nt);
    run("forward  product (solve.hip)", b, [&] {
      hipLaunchKernelGGL(trsv_fwd_flow_kernel, dim3(b.nt), dim3(FLOW_THREADS), 0, b.s, b.L, b.ld, b.invD, b.r, b.v, b.err, (int64_t)0,
                         (int64_t)0, (int64_t)0, b.nt, 1);
    });
    run("forward  TWO block rows per workgroup", b, [&] {
      hipLaunchKernelGGL(fwd2_lab, dim3((b.nt + 1) / 2), dim3(FLOW_THREADS), 0, b.s, b.L, b.ld, b.invD, b.r, b.v, b.err, b.nt);
    });
    FWD("round-5 form", 0);
    FWD("no first barrier", F_NOBAR1);
    FWD("dpp", F_DPP);
    FWD("upfront", F_UPFRONT);
    FWD("nobar+dpp", F_NOBAR1 | F_DPP);
    FWD("nobar+dpp+upfront", F_NOBAR1 | F_DPP | F_UPFRONT);
    FWD("nobar+dpp+backoff", F_NOBAR1 | F_DPP | F_BACKOFF);
    FWD("nobar+dpp+upfront+backoff", F_NOBAR1 | F_DPP | F_UPFRONT | F_BACKOFF);
    FWD("pad", F_PAD);
    FWD("pad+upfront", F_PAD | F_UPFRONT);
    FWD("pad+nobar+dpp+upfront", F_PAD | F_NOBAR1 | F_DPP | F_UPFRONT);
    FWD("pad+nobar+dpp", F_PAD | F_NOBAR1 | F_DPP);
    FWD("pad+nobar+dpp+upfront+backoff", F_PAD | F_NOBAR1 | F_DPP | F_UPFRONT | F_BACKOFF);
    FWD("pad+nobar+dpp+upfront+POLL1 (one wave polls)", F_PAD | F_NOBAR1 | F_DPP | F_UPFRONT | F_POLL1);
    FWD("pad+nobar+dpp+upfront+POLL2W (two waves poll)", F_PAD | F_NOBAR1 | F_DPP | F_UPFRONT | F_POLL2W);
    FWD("POLL2W + publish from two waves", F_PAD | F_NOBAR1 | F_DPP | F_UPFRONT | F_POLL2W | F_PUB2);
    FWD("POLL2W + publish from one wave", F_PAD | F_NOBAR1 | F_DPP | F_UPFRONT | F_POLL2W | F_PUB1);
    FWD("warm (round-5 form)", F_WARM);
    FWD("nobar+dpp+upfront+warm", F_NOBAR1 | F_DPP | F_UPFRONT | F_WARM);
    FWD("nobar+dpp+warm", F_NOBAR1 | F_DPP | F_WARM);
    FWD("nobar+dpp+upfront+warm+backoff", F_NOBAR1 | F_DPP | F_UPFRONT | F_WARM | F_BACKOFF);
    FWD("ABLATION warm, no fold", F_NOBAR1 | F_DPP | F_SKIP_FOLD | F_WARM);
    FWD("ABLATION warm, no solve", F_NOBAR1 | F_DPP | F_SKIP_SOLVE | F_WARM);
    FWD("ring4 (round-5 form)", F_RING4);
    FWD("ring8 (round-5 form)", F_RING8);
    FWD("nobar+dpp+upfront+ring4", F_NOBAR1 | F_DPP | F_UPFRONT | F_RING4);
    FWD("nobar+dpp+upfront+ring8", F_NOBAR1 | F_DPP | F_UPFRONT | F_RING8);
    FWD("nobar+dpp+upfront+ring4+backoff", F_NOBAR1 | F_DPP | F_UPFRONT | F_RING4 | F_BACKOFF);
    FWD("nobar+dpp+upfront+ring8+backoff", F_NOBAR1 | F_DPP | F_UPFRONT | F_RING8 | F_BACKOFF);
    FWD("ABLATION no fold (nobar+dpp)", F_NOBAR1 | F_DPP | F_SKIP_FOLD);
    FWD("ABLATION no solve (nobar+dpp)", F_NOBAR1 | F_DPP | F_SKIP_SOLVE);
    FWD("ABLATION neither", F_NOBAR1 | F_DPP | F_SKIP_FOLD | F_SKIP_SOLVE);
    run("backward product (solve.hip)", b, [&] {
      hipLaunchKernelGGL(trsv_bwd_flow_kernel, dim3(b.nt), dim3(FLOW_THREADS), 0, b.s, b.L, b.ld, b.invD, b.r, b.v, b.err, b.nt, (int64_t)0,
                         (int64_t)0, (int64_t)0, 1);
    });
    BWD("round-5 form", 0);
    BWD("quad+dpp last step", F_DPP);
    BWD("upfront fold", F_UPFRONT);
    BWD("upfront loop", F_UPLOOP);
    BWD("quad+dpp+upfront", F_DPP | F_UPFRONT);
    BWD("quad+dpp+upfront+uploop", F_DPP | F_UPFRONT | F_UPLOOP);
    BWD("backoff", F_BACKOFF);
    BWD("quad+dpp+backoff", F_DPP | F_BACKOFF);
    BWD("pad+quad+dpp", F_PAD | F_DPP);
    BWD("pad+quad+dpp+upfront", F_PAD | F_DPP | F_UPFRONT);
    BWD("pad+quad+dpp+upfront+uploop", F_PAD | F_DPP | F_UPFRONT | F_UPLOOP);
    BWD("pad+quad+dpp+upfront+backoff", F_PAD | F_DPP | F_UPFRONT | F_BACKOFF);
    BWD("warm (round-5 form)", F_WARM);
    BWD("warm+upfront", F_WARM | F_UPFRONT | F_UPLOOP);
    BWD("warm+quad+dpp", F_WARM | F_DPP);
    BWD("warm+quad+dpp+upfront", F_WARM | F_DPP | F_UPFRONT);
    BWD("warm+backoff", F_WARM | F_BACKOFF);
    BWD("ring4", F_RING4);
    BWD("ring8", F_RING8);
    BWD("ring4+backoff", F_RING4 | F_BACKOFF);
    BWD("ring8+backoff", F_RING8 | F_BACKOFF);
    BWD("ring4+upfront", F_RING4 | F_UPFRONT | F_UPLOOP);
    BWD("quad+dpp+upfront+ring4+backoff", F_DPP | F_UPFRONT | F_RING4 | F_BACKOFF);
    BWD("quad+dpp+upfront+ring8+backoff", F_DPP | F_UPFRONT | F_RING8 | F_BACKOFF);
    (void)hipFree(b.L); (void)hipFree(b.invD); (void)hipFree(b.r); (void)hipFree(b.v); (void)hipFree(b.err);
  }
  return 0;
}
