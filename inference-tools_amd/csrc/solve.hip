// Triangular solves and reductions of the GP path (gfx950).
//
// Replaces scipy.linalg.solve_triangular at regression.py:213, 242-244, 410, 447, 538 and the
// reductions of regression.py:214, 539.  All solves use the inverted 128 x 128 diagonal blocks
// produced by potrf_diag, so every step is a matrix-vector / matrix-matrix product:
//   single right-hand side  -> HBM-bound GEMV sweeps (L is read exactly once per solve)
//   many right-hand sides    -> the fp64 MFMA GEMM (right-hand sides stored as rows, "NT" form)
#include <cstdlib>
#include <mutex>

#include "gpmi_internal.h"

namespace {

constexpr int NB = GPMI_NB;

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// ---- single-launch sweeps (dataflow over the 128-blocks) ----------------------------------------
// One launch per 128-block (256 per fit) is bound by the kernel boundary (~10 us per step, 2.6 ms for the two
// sweeps of a fit at N = 16384, although L is only 1 GB).  Here ONE launch does a whole sweep:
// workgroup k owns block row k (forward) / block column k (backward), accumulates the contributions
// of the blocks whose solution is already published, and publishes its own 128 values.  The output
// vector doubles as the flag array: it is pre-filled with a NaN payload no computation produces and a
// consumer polls the elements it needs (agent-scope loads, which bypass the per-XCD L2; the
// producer's agent-scope stores write through), so one memory round trip per step is all that is
// left on the critical path.  The summation order is fixed, so results are bit-reproducible.
//
// Progress.  Workgroup k only ever waits for workgroups with a lower index of the SAME launch.  HIP promises
// no dispatch order, so the first argument is residency, not order: a sweep of up to `ncu` workgroups (N <= 32768 on 256
// CUs; a workgroup's 8 waves x ~200 registers and ~68 KiB of LDS leave room for exactly one per CU) has every workgroup of
// the launch - the producers a poller waits for included - resident or already finished, whatever order the dispatcher
// (one per XCD) picked.  The host side keeps it that way across streams: sweeps_in_flight() below serialises sweeps of
// different lanes once their workgroups would not all fit.  Larger sweeps, and batched ones (a whole batch in one
// launch), rest on what the hardware does: linear workgroup ids are handed out in ascending order, each XCD starting its
// share in order, so the lowest unfinished index is always resident or next to be dispatched (flow_batched_id below
// spells this out for batches).  Should a poll still time out (a bug, not a wait) the workgroup gives up for good:
// every later poll of that lane returns at once, it publishes what it has, and its consumers - which then
// read a non-sentinel value - do the same, so the launch drains in O(1) polls per workgroup and the host
// reports GPMI_ERR_INTERNAL instead of a hung GPU.
constexpr unsigned long long FLOW_SENTINEL = 0xFFF8DEADBEEF0000ull;
constexpr int FLOW_THREADS = 512;
constexpr int FLOW_SPIN_LIMIT = 1 << 22;  // ~1 s of s_sleep + L2-bypassing load per lane

// `dead` (per lane, starts false): set once a poll has timed out or another workgroup reported a failure
__device__ inline double flow_poll(const double* p, int* err, bool& dead) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  unsigned long long bits = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while (bits == FLOW_SENTINEL && !dead) {
    __builtin_amdgcn_s_sleep(1);
    bits = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ++spins;
    // every 4096 spins: has another workgroup of this sweep already given up?
    if ((spins & 4095) == 0 && err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) dead = true;
    if (spins > FLOW_SPIN_LIMIT) {
      if (err) atomicCAS(err, 0, GPMI_ERR_INTERNAL);
      dead = true;
    }
  }
  return (bits == FLOW_SENTINEL) ? 0.0 : __longlong_as_double((long long)bits);
}

// Which (step, problem) a workgroup of a BATCHED sweep is.  One problem's workgroups form a dependency chain, and a
// resident workgroup must never wait for one that has not been dispatched: the dispatcher hands linear workgroup id n to
// XCD n % 8 and every XCD starts its share in order, so problem z lives entirely on XCD z % 8, its steps in increasing
// n - a resident step's predecessor was dispatched before it on the same XCD, whatever else (a second stream, another
// process) competes for the CUs.  (With the steps of a problem dealt over the XCDs, two processes sharing a device could
// each hold the slots the other's missing predecessors needed: a time-out after 1 s.)  grid.x = 8 nt ceil(batch / 8).
// It is also faster: a problem's factor and its hand-offs stay in one XCD's L2 - config 5's batched likelihood 14 100 ->
// 15 200 evaluations/s.  (The same idea for a SINGLE sweep - consecutive steps on one XCD - changes nothing: 0.430 ms per
// sweep pair at N = 16384 either way.)
__device__ inline bool flow_batched_id(int nt, int batch, int& step, int& z) {
  const int n = (int)blockIdx.x, xcd = n & 7, m = n >> 3;
  step = m % nt;
  z = (m / nt) * 8 + xcd;
  return z < batch;
}

// -DGPMI_SWEEP_STAMPS (tools/probes/sweep_hops.hip only): 10 ns wall-clock stamps of wave 0 at the phases of a step
#ifdef GPMI_SWEEP_STAMPS
__device__ unsigned long long g_sweep_stamp[8][1024];
#define SWEEP_STAMP_DECL unsigned long long sweep_t[5] = {0, 0, 0, 0, 0}
#define SWEEP_STAMP(slot, k) sweep_t[slot] = __builtin_amdgcn_s_memrealtime()  // kept in registers: no store on the path
#define SWEEP_STAMP_FLUSH(k)                                                  \
  do {                                                                        \
    if (threadIdx.x == 0)                                                     \
      for (int q_ = 0; q_ < 5; ++q_) g_sweep_stamp[q_][k] = sweep_t[q_];      \
  } while (0)
#else
#define SWEEP_STAMP_DECL
#define SWEEP_STAMP(slot, k)
#define SWEEP_STAMP_FLUSH(k)
#endif

// x of lane (l ^ 1) / (l ^ 2) of the same quad, by DPP (a couple of cycles; __shfl_xor goes through ds_bpermute, an LDS
// round trip, twice per fold, on every step of a sweep's critical path)
template <int CTRL>
__device__ inline double quad_swap(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}
constexpr int QUAD_XOR1 = 0xB1;  // quad_perm [1, 0, 3, 2]
constexpr int QUAD_XOR2 = 0x4E;  // quad_perm [2, 3, 0, 1]

__device__ inline void flow_publish(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void flow_fill_kernel(double* __restrict__ v, int64_t np, int64_t sVec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < np)
    reinterpret_cast<unsigned long long*>(v + (int64_t)blockIdx.z * sVec)[i] = FLOW_SENTINEL;
}

// forward  L v = r.  Workgroup k: u = r_k - sum_{j<k} L_kj v_j ;  v_k = invD_k u.
// Wave w streams rows 16 w .. 16 w + 15 of the block row, lane l columns 2 l, 2 l + 1 of every block (1 KiB
// coalesced per row, requested before the block's v_j is polled: L is static); the partial sums stay in
// the lane across blocks and are folded once through LDS.  invD_k is preloaded four threads per row, so
// after v_{k-1} arrives 32 + 32 FMAs per thread, the fold, two barriers and the store remain.
__global__ __launch_bounds__(FLOW_THREADS) void trsv_fwd_flow_kernel(
    const double* __restrict__ L, int64_t ld, const double* __restrict__ invD,
    const double* __restrict__ r, double* __restrict__ v, int* __restrict__ err, int64_t sMat,
    int64_t sInv, int64_t sVec, int nt, int batch) {
  int k = blockIdx.x, z = 0;
  if (batch > 1 && !flow_batched_id(nt, batch, k, z)) return;
  SWEEP_STAMP_DECL;
  L += (int64_t)z * sMat;
  invD += (int64_t)z * sInv + (int64_t)k * NB * NB;
  r += (int64_t)z * sVec;
  v += (int64_t)z * sVec;
  if (err) err += z;
  // LDS layouts chosen against the bank rules of gfx950 (ds_write_b64: 16 contiguous lanes on 32 banks; ds_read_b128: groups of
  // 16 lanes = 4 rows x 4 quarter-rows on 64 banks).  A row of `part` is 32 slots of 16 B plus one of padding; the partial sum
  // of source lane 16 q + c lives in slot 8 q + ((c / 2 + 4 (q / 2)) % 8), half c % 2: the writers of a group hit 16 different
  // bank pairs, and the four quarter-rows x four rows a reader group fetches fall into 16 different slots.  `u` carries one
  // slot of padding per 32 values for the same reason.  (Rounds 3-5: part[NB][65] read as 16 doubles per lane and u[NB] - both
  // 4-way conflicts, ~0.6 us each on every step of the sweep; the summation order is the same, the results bit-identical.)
  __shared__ __attribute__((aligned(16))) double part[NB][66];
  __shared__ __attribute__((aligned(16))) double u[NB + 8];
  __shared__ __attribute__((aligned(16))) double vin[2][NB];  // v_j on its way from the two polling waves to all eight
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row4 = tid >> 2, q4 = tid & 3;  // four threads per row, 32 columns each
  const int wpos = 2 * (8 * (lane >> 4) + ((((lane & 15) >> 1) + 4 * (lane >> 5)) & 7)) + (lane & 1);
  // invD_k, requested first (static data: no dependence on the sweep), four threads per row
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
  const int nmain = k;  // blocks 0 .. k - 1; only the last one is on the critical path
  if (nmain > 0) {
    // two 8-row buffers per lane, alternating: half A (rows 0..7 of the wave) of block j is consumed while
    // half B is in flight, then half A of block j + 1 is requested before half B is consumed
    const double* base = L + (int64_t)(k * NB + wave * 16) * ld + 2 * lane;
    d2_t ha[8], hb[8];
    bool dead = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) ha[i] = *reinterpret_cast<const d2_t*>(base + (int64_t)i * ld);
    for (int j = 0; j < nmain; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        hb[i] = *reinterpret_cast<const d2_t*>(base + (int64_t)(8 + i) * ld + (int64_t)j * NB);
      // Two waves poll (one element per lane) and hand v_j on through LDS, as the backward sweep does: the step's fold
      // waits for the LAST wave that has seen the values, and the last of eight independent poll loops (each one memory
      // round trip per turn) is later than the one loop of a wave pair by a third of a round trip - 0.25 us per step at
      // N = 16384, where all eight waves polling for themselves also put eight times the requests on v_j's lines
      // (tools/probes/sweep_lab.hip: POLL2W; rounds 3-6a polled per wave).  Double-buffered: a slow wave may still read v_{j-1}.
      double* vj = vin[j & 1];
      if (tid < NB) vj[tid] = flow_poll(v + (int64_t)j * NB + tid, err, dead);
      if (j == nmain - 1) SWEEP_STAMP(0, k);
      __syncthreads();
      const d2_t vv = *reinterpret_cast<const d2_t*>(&vj[2 * lane]);
      const double v0 = vv[0], v1 = vv[1];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = fma(ha[i][0], v0, fma(ha[i][1], v1, acc[i]));
      if (j + 1 < nmain) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          ha[i] = *reinterpret_cast<const d2_t*>(base + (int64_t)i * ld + (int64_t)(j + 1) * NB);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[8 + i] = fma(hb[i][0], v0, fma(hb[i][1], v1, acc[8 + i]));
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave * 16 + i][wpos] = acc[i];
  SWEEP_STAMP(1, k);
  // Rows 16 w .. 16 w + 15 are written AND folded by wave w (row4 = tid >> 2): the LDS operations of one wave execute in
  // order, so no workgroup barrier stands between the partial sums and their fold (rounds 3-5 had one: ~0.2 us per step).
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  SWEEP_STAMP(2, k);
  {
    // fold the 64 lane partials of every row: 4 threads per row, 16 each (eight 16-byte reads), fixed order; all reads
    // are issued before the first add
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
    if (q4 == 0) u[row4 + 2 * (row4 >> 5)] = rk - s;
  }
  __syncthreads();
  SWEEP_STAMP(3, k);
  {
    double uv[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) uv[c] = u[q4 * 34 + c];
    __builtin_amdgcn_sched_barrier(0);
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 32; ++c) s = fma(xi[c], uv[c], s);
    s += quad_swap<QUAD_XOR1>(s);
    s += quad_swap<QUAD_XOR2>(s);
    if (q4 == 0) flow_publish(v + (int64_t)k * NB + row4, s);
    SWEEP_STAMP(4, k);
    SWEEP_STAMP_FLUSH(k);
  }
}

// backward  L^T a = w.  Workgroup b owns block column k = nt - 1 - b:
// u = w_k - sum_{j>k} L_jk^T a_j ;  a_k = invD_k^T u.  Wave w takes rows i = w, w + 8, .. of every block,
// lane l the columns 2 l, 2 l + 1 (1 KiB coalesced per row); the partial sums of a lane's two columns
// persist across blocks and the eight waves are folded once through LDS.
__global__ __launch_bounds__(FLOW_THREADS) void trsv_bwd_flow_kernel(
    const double* __restrict__ L, int64_t ld, const double* __restrict__ invD,
    const double* __restrict__ w, double* __restrict__ a, int* __restrict__ err, int nt, int64_t sMat,
    int64_t sInv, int64_t sVec, int batch) {
  int step = blockIdx.x, z = 0;
  if (batch > 1 && !flow_batched_id(nt, batch, step, z)) return;
  SWEEP_STAMP_DECL;
  const int k = nt - 1 - step;
  L += (int64_t)z * sMat;  // batch (lockstep evaluations)
  invD += (int64_t)z * sInv;
  w += (int64_t)z * sVec;
  a += (int64_t)z * sVec;
  if (err) err += z;
  invD += (int64_t)k * NB * NB;
  __shared__ double part[8][NB];
  __shared__ double u[NB];
  __shared__ double ain[2][NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = tid & 127, rg = tid >> 7;  // last step: thread = (column, group of 32 rows)
  double xi[32];
  {
    const double* p = invD + (rg * 32) * NB + col;
#pragma unroll
    for (int i = 0; i < 32; ++i) xi[i] = p[i * NB];
  }
  const double wk = w[(int64_t)k * NB + col];
  d2_t acc = d2_t{0.0, 0.0};
  const int nmain = nt - k - 1;  // blocks j = nt - 1 .. k + 1; only the last one is on the critical path
  if (nmain > 0) {
    // rows wave + 8 i of a block: i = 0..7 in buffer ha, i = 8..15 in hb (alternating as in the forward sweep)
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
      // a_j into LDS (double-buffered: a slow wave may still read the previous block's values)
      double* aj = ain[t & 1];
      if (tid < NB) aj[tid] = flow_poll(a + (int64_t)j * NB + tid, err, dead);
      if (t == nmain - 1) SWEEP_STAMP(0, step);
      __syncthreads();
      if (t == nmain - 1) SWEEP_STAMP(1, step);
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
  part[wave][2 * lane] = acc[0];
  part[wave][2 * lane + 1] = acc[1];
  __syncthreads();
  SWEEP_STAMP(2, step);
  if (tid < NB) {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += part[g][tid];
    u[tid] = wk - s;
  }
  __syncthreads();
  SWEEP_STAMP(3, step);
  {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s = fma(xi[i], u[rg * 32 + i], s);
    part[4 + rg][col] = s;
    __syncthreads();
    if (tid < NB)
      flow_publish(a + (int64_t)k * NB + tid, (part[4][tid] + part[5][tid]) + (part[6][tid] + part[7][tid]));
    SWEEP_STAMP(4, step);
    SWEEP_STAMP_FLUSH(step);
  }
}

// inv2 slot b (512 x 512, ld 512) <- blockdiag(invD_{4b}, .., invD_{4b+3}), zeros elsewhere
__global__ void inv2_seed_kernel(const double* __restrict__ invD, double* __restrict__ inv2, int nt) {
  const int b = blockIdx.z, r = blockIdx.y, cpair = (blockIdx.x * 64 + threadIdx.x) * 2;
  const int t = r >> 7;
  d2_t v = d2_t{0.0, 0.0};
  if ((cpair >> 7) == t && 4 * b + t < nt)
    v = *reinterpret_cast<const d2_t*>(invD + ((int64_t)(4 * b + t) * NB + (r & 127)) * NB + (cpair & 127));
  *reinterpret_cast<d2_t*>(inv2 + ((int64_t)b * GPMI_OB + r) * GPMI_OB + cpair) = v;
}

__global__ void copy_panel_kernel(const double* __restrict__ src, int64_t lds, double* __restrict__ dst,
                                  int64_t ldd) {
  const int64_t r = blockIdx.y;
  const int c = (blockIdx.x * 64 + threadIdx.x) * 2;
  *reinterpret_cast<d2_t*>(dst + r * ldd + c) = *reinterpret_cast<const d2_t*>(src + r * lds + c);
}

__global__ void set_identity_kernel(double* __restrict__ Q, int64_t ld, int64_t np, int64_t sMat) {
  const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const int64_t i = blockIdx.y;
  Q += (int64_t)blockIdx.z * sMat;
  if (j < np)
    *reinterpret_cast<d2_t*>(Q + i * ld + j) = d2_t{j == i ? 1.0 : 0.0, j + 1 == i ? 1.0 : 0.0};
}

// `height` rows of `width` doubles between a strided and a packed layout (grid: x over the row, y = row)
__global__ void copy_rows_kernel(const double* __restrict__ src, int64_t spitch, double* __restrict__ dst, int64_t dpitch,
                                 int64_t width) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < width) dst[(int64_t)blockIdx.y * dpitch + j] = src[(int64_t)blockIdx.y * spitch + j];
}

__global__ void copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

__global__ void residual_batched_kernel(const double* __restrict__ y, const double* __restrict__ mus,
                                        const double* __restrict__ mu_consts, double* __restrict__ r,
                                        int64_t n, int64_t np, int64_t sVec) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const int64_t z = blockIdx.z;
  const double m = mus ? mus[z * n + (i < n ? i : 0)] : mu_consts[z];
  r[z * sVec + i] = (i < n) ? y[i] - m : 0.0;
}

__global__ void residual_kernel(const double* __restrict__ y, const double* __restrict__ mu,
                                double mu_const, double* __restrict__ r, int64_t n, int64_t np) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  r[i] = (i < n) ? y[i] - (mu ? mu[i] : mu_const) : 0.0;
}

// deterministic two-value reduction: red[0] = sum v^2, red[1] = sum ln L_ii  (fixed tree order)
__global__ __launch_bounds__(1024) void lml_reduce_kernel(const double* __restrict__ v,
                                                          const double* __restrict__ L, int64_t ld,
                                                          int64_t np, double* __restrict__ red,
                                                          int64_t sMat, int64_t sVec) {
  v += (int64_t)blockIdx.z * sVec;
  L += (int64_t)blockIdx.z * sMat;
  red += 2 * blockIdx.z;
  __shared__ double s0[16], s1[16];
  double a = 0.0, b = 0.0;
  for (int64_t i = threadIdx.x; i < np; i += 1024) {
    const double vi = v[i];
    a = fma(vi, vi, a);
    b += log(L[i * ld + i]);
  }
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s0[wave] = a;
    s1[wave] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < 16; ++w) {
      ta += s0[w];
      tb += s1[w];
    }
    red[0] = ta;
    red[1] = tb;
  }
}

// one wave per row: out[m] = sum_n Q[m][n] * a[n]
__global__ __launch_bounds__(256) void rows_dot_kernel(const double* __restrict__ Q, int64_t ld,
                                                       int64_t mp, int64_t np,
                                                       const double* __restrict__ a,
                                                       double* __restrict__ out, int64_t sQ, int64_t sVec) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= mp) return;
  a += (int64_t)blockIdx.z * sVec;    // batch: problem z, vectors sVec apart
  out += (int64_t)blockIdx.z * sVec;
  const double* q = Q + (int64_t)blockIdx.z * sQ + row * ld;
  double s = 0.0;
  // (one wave per row, one chain of sums in column order; unrolled so that eight row pieces are in flight per wave: with
  // four waves per CU the loop otherwise reads at 2.3 TB/s)
#pragma unroll 8
  for (int64_t j = lane * 2; j < np; j += 128) {
    const d2_t qv = *reinterpret_cast<const d2_t*>(q + j);
    const d2_t av = *reinterpret_cast<const d2_t*>(a + j);
    s = fma(qv[0], av[0], s);
    s = fma(qv[1], av[1], s);
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

__global__ __launch_bounds__(256) void rows_sumsq_kernel(const double* __restrict__ Q, int64_t ld,
                                                         int64_t mp, int64_t np, double base,
                                                         double* __restrict__ out, int64_t sQ, int64_t sVec,
                                                         double sign) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= mp) return;
  out += (int64_t)blockIdx.z * sVec;
  const double* q = Q + (int64_t)blockIdx.z * sQ + row * ld;
  double s = 0.0;
#pragma unroll 8
  for (int64_t j = lane * 2; j < np; j += 128) {
    const d2_t qv = *reinterpret_cast<const d2_t*>(q + j);
    s = fma(qv[0], qv[0], s);
    s = fma(qv[1], qv[1], s);
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = sign * (base - s);
}

}  // namespace

// Serialise sweeps of different streams so that their workgroups are always all resident together (see the
// progress argument above): the chip holds one flow workgroup per CU; a sweep of nt >= FLOW_GATE_MIN workgroups
// (N > 4096: the sizes that run one lane per evaluation, gpmi_lml_batch) queues behind the previous such sweep of
// the context through one event, whichever lane or context issues it, so at most one of them (<= ncu workgroups for
// N <= 32768) is in flight beside any number of small ones.  They are HBM-bound and sub-millisecond: nothing is
// lost by not overlapping them.
// The gate is per DEVICE, process-wide (one event per device id under a mutex), not per context: a regressor, the
// engine its covariance object keeps for cross-covariances, a second regressor or another thread each have a context
// of their own, and their sweeps share the same CUs.
constexpr int64_t FLOW_GATE_MIN = 32;
namespace {
std::mutex g_sweep_mu;
hipEvent_t g_sweep_ev[64] = {nullptr};
bool g_sweep_used[64] = {false};
}  // namespace
static void flow_gate_enter(gpmi_ctx* c, hipStream_t s, int64_t workgroups) {
  if (workgroups <= FLOW_GATE_MIN || c->device < 0 || c->device >= 64) return;
  std::lock_guard<std::mutex> lk(g_sweep_mu);
  const int d = c->device;
  if (!g_sweep_ev[d] && hipEventCreateWithFlags(&g_sweep_ev[d], hipEventDisableTiming) != hipSuccess) {
    g_sweep_ev[d] = nullptr;
    (void)hipGetLastError();
    return;
  }
  if (g_sweep_used[d]) (void)hipStreamWaitEvent(s, g_sweep_ev[d], 0);  // (a no-op behind the same stream's own record)
}
static void flow_gate_leave(gpmi_ctx* c, hipStream_t s, int64_t workgroups) {
  if (workgroups <= FLOW_GATE_MIN || c->device < 0 || c->device >= 64) return;
  std::lock_guard<std::mutex> lk(g_sweep_mu);
  const int d = c->device;
  if (!g_sweep_ev[d]) return;
  (void)hipEventRecord(g_sweep_ev[d], s);
  g_sweep_used[d] = true;
}

void lane_run_early(Lane& lane, hipStream_t s) {
  Lane::EarlyWork& e = lane.early;
  if (!e.pending) return;
  e.pending = false;
  launch_residual(s, e.y, e.mu, e.mu_const, e.r, e.n, e.np);
  for (double* f : e.fill)
    if (f) hipLaunchKernelGGL(flow_fill_kernel, dim3((unsigned)((e.np + 255) / 256), 1, 1), dim3(256), 0, s, f, e.np, (int64_t)0);
  if (e.ident) {
    launch_set_identity(s, e.ident, e.ident_ld, e.np);
    lane.identity_ready = true;
  }
}

void trsv_forward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                  const double* invD, const double* r, double* out, int* err, const BatchShape& bs, bool prefilled) {
  const int nt = (int)(np / NB);
  flow_gate_enter(c, s, (int64_t)nt);
  ProfScope ps(c, s, GPMI_PROF_SOLVE, (double)np * np, 4.0 * np * np);
  if (!prefilled)
    hipLaunchKernelGGL(flow_fill_kernel, dim3((unsigned)((np + 255) / 256), 1, (unsigned)bs.count), dim3(256), 0, s,
                       out, np, bs.sVec);
  const unsigned grid = bs.count > 1 ? 8u * (unsigned)nt * (unsigned)((bs.count + 7) / 8) : (unsigned)nt;
  hipLaunchKernelGGL(trsv_fwd_flow_kernel, dim3(grid), dim3(FLOW_THREADS), 0, s, L, ld, invD, r, out, err, bs.sMat,
                     bs.sInv, bs.sVec, nt, bs.count);
  flow_gate_leave(c, s, (int64_t)nt);
}

void trsv_backward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                   const double* invD, const double* r, double* out, int* err, const BatchShape& bs, bool prefilled) {
  const int nt = (int)(np / NB);
  flow_gate_enter(c, s, (int64_t)nt);
  ProfScope ps(c, s, GPMI_PROF_SOLVE, (double)np * np, 4.0 * np * np);
  if (!prefilled)
    hipLaunchKernelGGL(flow_fill_kernel, dim3((unsigned)((np + 255) / 256), 1, (unsigned)bs.count), dim3(256), 0, s, out,
                       np, bs.sVec);
  const unsigned grid = bs.count > 1 ? 8u * (unsigned)nt * (unsigned)((bs.count + 7) / 8) : (unsigned)nt;
  hipLaunchKernelGGL(trsv_bwd_flow_kernel, dim3(grid), dim3(FLOW_THREADS), 0, s, L, ld, invD, r, out, err, nt, bs.sMat,
                     bs.sInv, bs.sVec, bs.count);
  flow_gate_leave(c, s, (int64_t)nt);
}

void build_inv2(hipStream_t s, const double* L, int64_t np, int64_t ld, const double* invD, double* inv2,
                double* tmp) {
  // inv([[A, 0], [C, B]]) = [[A^-1, 0], [-B^-1 C A^-1, B^-1]], applied twice: 128 -> 256 -> 512.
  // Every product is one batched launch over the outer blocks (uniform strides).
  const int nt = (int)(np / NB);
  const int nob = (nt + 3) / 4;
  const int64_t slot = (int64_t)GPMI_OB * GPMI_OB, tslot = 256 * 256;
  hipLaunchKernelGGL(inv2_seed_kernel, dim3(GPMI_OB / 2 / 64, GPMI_OB, (unsigned)nob), dim3(64), 0, s, invD, inv2,
                     nt);
  auto offdiag = [&](int r0, int c0, int h2, int h1, int count) {
    // X[r0.., c0..] = -X[r0.., r0..] * L[r0.., c0..] * X[c0.., c0..]   (h2 x h1 tiles), `count` outer blocks
    if (count <= 0 || h2 <= 0) return;
    const GemmBatch b1{count, tslot, 4 * NB * ld + 4 * NB, slot};
    launch_gemm(s, TILES_RECT, OP_ASSIGN, true, 0, tmp, 256, L + (int64_t)r0 * NB * ld + (int64_t)c0 * NB, ld,
                inv2 + (int64_t)c0 * NB * GPMI_OB + (int64_t)c0 * NB, GPMI_OB, h2, h1, h1 * NB, nullptr, b1);
    const GemmBatch b2{count, slot, slot, tslot};
    launch_gemm(s, TILES_RECT, OP_SUB, true, 0, inv2 + (int64_t)r0 * NB * GPMI_OB + (int64_t)c0 * NB, GPMI_OB,
                inv2 + (int64_t)r0 * NB * GPMI_OB + (int64_t)r0 * NB, GPMI_OB, tmp, 256, h2, h1, h2 * NB, nullptr,
                b2);
  };
  // outer blocks b with tile 4 b + t inside the matrix: (nt - t + 3) / 4
  offdiag(1, 0, 1, 1, (nt - 1 + 3) / 4);
  offdiag(3, 2, 1, 1, (nt - 3 + 3) / 4);
  const int full = nt / 4;
  offdiag(2, 0, 2, 2, full);
  if (nt % 4 == 3) {  // last outer block has three tiles: a 1 x 2 corner
    const int64_t b = full;
    const GemmBatch one{};
    const double* Lb = L + (b * 4 * NB) * ld + b * 4 * NB;
    double* Xb = inv2 + b * slot;
    double* Tb = tmp + b * tslot;
    launch_gemm(s, TILES_RECT, OP_ASSIGN, true, 0, Tb, 256, Lb + (int64_t)2 * NB * ld, ld, Xb, GPMI_OB, 1, 2, 2 * NB,
                nullptr, one);
    launch_gemm(s, TILES_RECT, OP_SUB, true, 0, Xb + (int64_t)2 * NB * GPMI_OB, GPMI_OB,
                Xb + (int64_t)2 * NB * GPMI_OB + 2 * NB, GPMI_OB, Tb, 256, 1, 2, NB, nullptr, one);
  }
}

void trsm_rows_forward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                       const double* inv2, double* Q, int64_t mp, bool upper_rhs, double* Qout,
                       double* panel) {
  // Right-looking sweep over the 512-wide outer blocks: the block's solution is ONE product with the
  // inverted diagonal block (K <= 512, cut at the diagonal: the inverse is lower triangular), then
  // ONE update with K = 512 carries it to all remaining columns (throughput bound).  32 dependent
  // pairs of launches at N = 16384 instead of 128 x 2 + 32 with the 128-wide inverses.
  const int nt = (int)(np / NB), mt_all = (int)(mp / NB);
  const int OBT = GPMI_OB / NB;
  const int64_t ldp = GPMI_OB + 32;
  ProfScope ps(c, s, GPMI_PROF_TRSM, (double)mp * np * np, 4.0 * np * np);
  for (int J = 0; J < nt; J += OBT) {
    const int Je = (J + OBT < nt) ? J + OBT : nt;
    const int w = Je - J;
    // upper_rhs: Q is upper triangular (e.g. the identity): rows below block Je - 1 are still zero here
    const int mt = upper_rhs ? (Je < mt_all ? Je : mt_all) : mt_all;
    double* QJ = Q + (int64_t)J * NB;
    double* X = Qout ? Qout + (int64_t)J * NB : panel;
    const int64_t ldx = Qout ? ld : ldp;
    launch_gemm(s, TILES_RECT, OP_ASSIGN, false, 2, X, ldx, QJ, ld, inv2 + (int64_t)(J / OBT) * GPMI_OB * GPMI_OB,
                GPMI_OB, mt, w, w * NB);
    const int rest = nt - Je;
    if (rest > 0)  // Q[:, Je:] -= X * L[Je:, J:Je]^T
      launch_gemm_nt_split(s, TILES_RECT, OP_SUB, Q + (int64_t)Je * NB, ld, X, ldx,
                           L + (int64_t)Je * NB * ld + (int64_t)J * NB, ld, mt, rest, w * NB,
                           gemm_split_point((int64_t)mt * rest, c->ncu, w * NB));
    if (!Qout) {
      const int64_t rows = (int64_t)mt * NB;
      hipLaunchKernelGGL(copy_panel_kernel, dim3((unsigned)(w * NB / 2 / 64), (unsigned)rows), dim3(64), 0, s, panel,
                         ldp, QJ, ld);
    }
  }
}

void trsm_rows_backward(gpmi_ctx* c, hipStream_t s, const double* L, int64_t np, int64_t ld,
                        const double* invD, double* Q, int64_t mp, const double* inv2, double* panel) {
  const int nt = (int)(np / NB), mt = (int)(mp / NB);
  ProfScope ps(c, s, GPMI_PROF_TRSM, (double)mp * np * np, 4.0 * np * np);
  // GPMI_BACKWARD_OB=0: the 128-wide steps of rounds 1-6a (two launches per tile column)
  static const bool wide = [] {
    const char* e = std::getenv("GPMI_BACKWARD_OB");
    return !e || std::atoi(e) != 0;
  }();
  if (wide && inv2 && panel) {
    // Right-looking over the 512-wide outer blocks from the last to the first, the mirror image of trsm_rows_forward: the
    // block's solution is ONE product with the inverted diagonal block (untransposed: B is k-major, the contraction of
    // tile column j starts at its diagonal), then ONE update with K = 512 carries it to all columns to its left.  At
    // N = 4096 that is 8 dependent pairs of launches instead of 32 (round 6; the spatial derivatives of config 4's
    // acquisition gradient spent 45 % of their time in those 64 launches).
    const int OBT = GPMI_OB / NB;
    const int64_t ldp = GPMI_OB + 32;
    const int nob = (nt + OBT - 1) / OBT;
    for (int b = nob - 1; b >= 0; --b) {
      const int J = b * OBT, Je = (J + OBT < nt) ? J + OBT : nt, w = Je - J;
      double* QJ = Q + (int64_t)J * NB;
      launch_gemm(s, TILES_RECT, OP_ASSIGN, true, 4, panel, ldp, QJ, ld, inv2 + (int64_t)b * GPMI_OB * GPMI_OB, GPMI_OB, mt, w,
                  w * NB);
      if (J > 0)  // Q[:, 0:J] -= X * L[J:Je, 0:J]     (B = block rows J .. Je of L, k-major)
        launch_gemm(s, TILES_RECT, OP_SUB, true, 0, Q, ld, panel, ldp, L + (int64_t)J * NB * ld, ld, mt, J, w * NB);
      hipLaunchKernelGGL(copy_panel_kernel, dim3((unsigned)(w * NB / 2 / 64), (unsigned)((int64_t)mt * NB)), dim3(64), 0, s, panel,
                         ldp, QJ, ld);
    }
    return;
  }
  for (int k = nt - 1; k >= 0; --k) {
    double* Qk = Q + (int64_t)k * NB;
    // Q[:, k] <- Q[:, k] * invD_k          (B = invD_k is k-major here)
    launch_gemm(s, TILES_RECT, OP_ASSIGN, true, false, Qk, ld, Qk, ld, invD + (int64_t)k * NB * NB,
                NB, mt, 1, NB);
    // Q[:, 0:k] -= Q[:, k] * L[k, 0:k]     (B = block row k of L, k-major)
    if (k > 0)
      launch_gemm(s, TILES_RECT, OP_SUB, true, false, Q, ld, Qk, ld, L + (int64_t)k * NB * ld, ld, mt,
                  k, NB);
  }
}

void launch_set_identity(hipStream_t s, double* Q, int64_t ld, int64_t np, int batch, int64_t sMat) {
  dim3 grid((unsigned)((np / 2 + 255) / 256), (unsigned)np, (unsigned)batch);
  hipLaunchKernelGGL(set_identity_kernel, grid, dim3(256), 0, s, Q, ld, np, sMat);
}

// B lockstep problems (np <= 4096): Q_z <- L_z^-T, rows = columns of L_z^-1 (Q_z starts as the identity), by forward
// substitution over the 128-blocks with the inverses of the diagonal blocks - two-level like potrf_lower_batched: inside
// an outer panel of 4 tile columns the K = 128 steps touch the panel's own columns only, the columns to its right take
// the panel at once with K = 512.  Q is upper triangular throughout (row r has non-zeros from column r on), so block
// row t only joins at its own panel.  Every launch carries the whole batch (blockIdx.z).
void trsm_identity_batched(hipStream_t s, const double* L, int64_t np, int64_t ld, const double* invD, double* Q,
                           const BatchShape& bs) {
  const int nt = (int)(np / NB);
  const int OBT = 4;
  launch_set_identity(s, Q, ld, np, bs.count, bs.sMat);
  GemmBatch inplace{bs.count, bs.sMat, bs.sMat, bs.sInv};
  inplace.b_lower_tri = true;  // products with the inverses of the diagonal blocks
  GemmBatch upd{bs.count, bs.sMat, bs.sMat, bs.sMat};
  upd.ring_order_only = true;  // the K = 512 updates of a batch of one must sum like those of a batch of many
  for (int J = 0; J < nt; J += OBT) {
    const int Je = (J + OBT < nt) ? J + OBT : nt;
    for (int j = J; j < Je; ++j) {
      const int rows = j + 1;  // block rows 0 .. j have entries in column block j
      double* Qj = Q + (int64_t)j * NB;
      // Q[:, j] <- Q[:, j] invD_j^T   (in place, one tile column)
      launch_gemm_nt(s, TILES_RECT, OP_ASSIGN, Qj, ld, Qj, ld, invD + (int64_t)j * NB * NB, NB, rows, 1, NB, nullptr,
                     inplace);
      const int pc = Je - j - 1;
      if (pc > 0)  // Q[:, j+1 .. Je) -= Q[:, j] L[j+1 .. Je, j]^T
        launch_gemm_nt(s, TILES_RECT, OP_SUB, Qj + NB, ld, Qj, ld, L + (int64_t)(j + 1) * NB * ld + (int64_t)j * NB, ld,
                       rows, pc, NB, nullptr, upd);
    }
    const int rest = nt - Je;
    if (rest > 0)  // Q[:, Je ..) -= Q[:, J .. Je) L[Je .., J .. Je)^T
      launch_gemm_nt(s, TILES_RECT, OP_SUB, Q + (int64_t)Je * NB, ld, Q + (int64_t)J * NB, ld,
                     L + (int64_t)Je * NB * ld + (int64_t)J * NB, ld, Je, rest, (Je - J) * NB, nullptr, upd);
  }
}

void launch_copy_rows(hipStream_t s, const double* src, int64_t spitch, double* dst, int64_t dpitch, int64_t width,
                      int64_t height) {
  if (width <= 0 || height <= 0) return;
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((width + 255) / 256), (unsigned)height), dim3(256), 0, s, src, spitch,
                     dst, dpitch, width);
}

void launch_copy(hipStream_t s, const double* src, double* dst, int64_t n) {
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
}

void launch_residual(hipStream_t s, const double* y, const double* mu, double mu_const, double* r,
                     int64_t n, int64_t np) {
  hipLaunchKernelGGL(residual_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, y, mu,
                     mu_const, r, n, np);
}

void launch_lml_reduce(hipStream_t s, const double* v, const double* L, int64_t ld, int64_t np,
                       double* red, const BatchShape& bs) {
  hipLaunchKernelGGL(lml_reduce_kernel, dim3(1, 1, (unsigned)bs.count), dim3(1024), 0, s, v, L, ld, np, red,
                     bs.sMat, bs.sVec);
}

void launch_residual_batched(hipStream_t s, const double* y, const double* mus, const double* mu_consts,
                             double* r, int64_t n, int64_t np, const BatchShape& bs) {
  hipLaunchKernelGGL(residual_batched_kernel, dim3((unsigned)((np + 255) / 256), 1, (unsigned)bs.count),
                     dim3(256), 0, s, y, mus, mu_consts, r, n, np, bs.sVec);
}

void launch_rows_dot(hipStream_t s, const double* Q, int64_t ld, int64_t mp, int64_t np,
                     const double* a, double* out, int batch, int64_t sQ, int64_t sVec) {
  hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)((mp + 3) / 4), 1, (unsigned)batch), dim3(256), 0, s, Q, ld, mp, np,
                     a, out, sQ, sVec);
}

void launch_rows_sumsq(hipStream_t s, const double* Q, int64_t ld, int64_t mp, int64_t np,
                       double base, double* out, int batch, int64_t sQ, int64_t sVec, double sign) {
  hipLaunchKernelGGL(rows_sumsq_kernel, dim3((unsigned)((mp + 3) / 4), 1, (unsigned)batch), dim3(256), 0, s, Q, ld, mp,
                     np, base, out, sQ, sVec, sign);
}
