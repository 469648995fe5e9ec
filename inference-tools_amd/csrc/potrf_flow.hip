// Flag-ordered ("dataflow") tile Cholesky for the part of the factorisation that is bound by the panel chain:
// the last tile rows of a large matrix and the whole of a matrix of N <= ~8000 (numpy.linalg.cholesky at
// regression.py:241, 537, 555).
//
// In stream order that part costs chain + updates: potrf_diag (29.5 us) -> panel TRSM -> inner update per 128 columns
// with the chip nearly idle, then a trailing update with the chain idle.  Here the two run side by side:
//
//   chain stream (the lane's CU-masked panel stream, ordinary launches in stream order, nothing but the critical path):
//       D(k)  = potrf_diag of tile (k, k)
//       Tc(k) = tile (k+1, k) <- tile (k+1, k) invD(k)^T              (4 workgroups of 32 rows)
//       Uc(k) = tile (k+1, k+1) -= L(k+1, k) L(k+1, k)^T               (3 workgroups of 64 x 64)
//   update stream (the other CUs): ONE persistent launch whose workgroups pull tile tasks from in-order queues, a task
//   being claimed only when the flags of its inputs are set:
//       T(i, k), i >= k + 2    panel TRSM of tile (i, k), four 32-row slabs
//       U(i, j, k)             tile (i, j) -= L(i, k) L(j, k)^T for ONE column k (K = 128), four 64 x 64 sub-tiles;
//                              columns of the tile's own outer panel (and, near the diagonal, of the panel before it)
//       Z(i, j, q)             tile (i, j) -= L(i, 4q..4q+3) L(j, 4q..4q+3)^T  (K = 512, the throughput kernel), every
//                              earlier outer panel
//   i.e. the two-level blocking of potrf_lower, tile by tile, with the same kernels' tile bodies (gemm_tiles.h) and
//   the same order of summation for every element: the factor is bit-identical to the stream-ordered schedule's.
//
// Flags (ints, zeroed per factorisation): Ddone = number of diagonal blocks factored; Lcnt[i] = TRSM slabs finished
// in tile row i (4 per column, columns in order); F[i][j] = sub-updates applied to tile (i, j) (4 per column).
// A chain launch publishes the results of the launch before it at its own start (the kernel boundary has made them
// visible) and waits - normally not at all - for the flags of the tile it is about to touch.  A task kernel workgroup
// releases its stores (agent-scope release fence) before it bumps a flag and acquires after it has claimed a task.
//
// Progress: every queue is sorted by a virtual time under which each task comes after all its inputs, a task is
// claimed only when its inputs are complete, and claimed tasks run to completion - so the earliest unclaimed task of
// the earliest queue head always becomes ready, whatever the dispatch order and however few workgroups are resident.
// The only cross-launch dependency is between the chain stream and the task kernel, which run on disjoint CU masks.
// Every poll is bounded: a time-out (a bug or a serialising profiler, never a wait) sets the abort word, every
// poller gives up, and the host reports GPMI_ERR_INTERNAL through `info` instead of hanging the GPU.
// tools/sim/flow_sim.py holds the same task list as an executable model (NumPy replay in random admissible order
// + a discrete-event timing model that chose the queue layout).
#include <algorithm>
#include <cstdlib>
#include <tuple>
#include <vector>

#include "gemm_tiles.h"

using namespace gemm_tiles;

namespace {

constexpr int NB = GPMI_NB;
constexpr int OBT = 4;        // tile columns per outer panel (K = 512 chunks)
constexpr int FLOW_NQ = 4;    // queues: rows within 3 / within 8 tile rows of the column being applied / the rest / Z
constexpr int FT_T = 0, FT_U = 1, FT_Z = 2;

struct FlowTask {
  uint8_t type, s, fadd, pad;
  uint16_t i, j, k, pad2;  // k: column (T, U) or outer panel q (Z)
};
static_assert(sizeof(FlowTask) == 12, "FlowTask layout");

// hot words on lines of their own
constexpr int FL_DDONE = 0, FL_ABORT = 32, FL_HEAD = 64, FL_LCNT = 64 + 32 * FLOW_NQ;
__host__ __device__ inline int flow_f_off(int m) { return FL_LCNT + ((m + 31) / 32) * 32; }
inline int flow_claim_off(int m) { return flow_f_off(m) + ((m * m + 31) / 32) * 32; }  // one claim word per task
inline int flow_flag_ints(int m, int64_t ntasks) { return flow_claim_off(m) + (int)ntasks; }

struct FlowArgs {
  double* A;           // tile (0, 0) of the tail
  const double* invD;  // inverse of diagonal block 0 of the tail
  int64_t ld;
  int m;
  const FlowTask* tasks[FLOW_NQ];
  int* claim[FLOW_NQ];  // claim words of queue q
  int count[FLOW_NQ];
  int* flags;
  int* info;
  unsigned long long* stamp;
};

__device__ __forceinline__ bool flow_ready(const FlowTask& t, const int* __restrict__ fl, int m) {
  const int* Lcnt = fl + FL_LCNT;
  const int* F = fl + flow_f_off(m);
  if (t.type == FT_T) return flow_ld(fl + FL_DDONE) >= t.k + 1 && flow_ld(F + t.i * m + t.k) == 4 * t.k;
  if (t.type == FT_U)
    return flow_ld(Lcnt + t.i) >= 4 * (t.k + 1) && flow_ld(Lcnt + t.j) >= 4 * (t.k + 1) &&
           flow_ld(F + t.i * m + t.j) >= 4 * t.k;
  return flow_ld(Lcnt + t.i) >= 4 * OBT * (t.k + 1) && flow_ld(Lcnt + t.j) >= 4 * OBT * (t.k + 1) &&
         flow_ld(F + t.i * m + t.j) == 4 * OBT * t.k;
}

// The two small tile bodies are kept out of line: with all three inlined into one loop the register allocation ran out
// (256 VGPRs and 196 bytes of scratch per lane, spills inside the K = 512 body); as functions they keep their own
// and the K = 512 body, inlined, has the kernel's to itself.  (The LDS block travels as an address-space-3 pointer
// and the matrices as address-space-1 pointers so that the accesses stay ds_ / global_ instructions, not flat_.)
typedef __attribute__((address_space(3))) double lds_double_t;
typedef __attribute__((address_space(1))) double glb_double_t;  // (generic pointers would turn every access into flat_)
// PROTO (GPMI_FLOW_PROTO): how a tile changes hands between CUs.  0: write-through (sc1) stores, acquire on the
// consumer; 1: the same + an agent release in front of the flag; 2: sc1 stores and sc1 loads of every matrix byte;
// 3: non-temporal stores + agent release (whole-L2 write-back per task).
template <int PROTO>
__device__ __attribute__((noinline)) void flow_do_T(glb_double_t* C, const glb_double_t* invDk, int64_t ld,
                                                    lds_double_t* smem) {
  staged_tile<OP_ASSIGN, 0, 32, 128, PROTO == 3 ? CST_NT : CST_SC1, PROTO == 2 ? LD_SC1 : LD_PLAIN>(
      (const double*)C, (const double*)invDk, (double*)C, ld, NB, ld, NB / BK, (double*)smem);
}
template <int PROTO>
__device__ __attribute__((noinline)) void flow_do_U(const glb_double_t* Ai, const glb_double_t* Bj, glb_double_t* C,
                                                    int64_t ld, lds_double_t* smem) {
  dma64_tile<OP_SUB, PROTO == 3 ? CST_NT : CST_SC1, PROTO == 2 ? LD_SC1 : LD_PLAIN>(
      (const double*)Ai, (const double*)Bj, (double*)C, ld, ld, ld, NB / DMA_BK, (double*)smem);
}
template <int PROTO>
__global__ __launch_bounds__(256, 2) void flow_task_kernel(FlowArgs a) {
  __shared__ double smem[DMA128_LDS_DOUBLES];
  __shared__ int sh[4];
  static_assert(staged_lds_doubles<0, 32, 128>() <= DMA128_LDS_DOUBLES, "LDS of the TRSM slab body");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* fl = a.flags;
  int* Lcnt = fl + FL_LCNT;
  int* F = fl + flow_f_off(a.m);
  if (a.stamp && tid == 0 && blockIdx.x < 8) a.stamp[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    if (wave == 0) {
      // Queues in priority order; of each, the wave examines the 64 tasks from the queue's head on at once (lane l:
      // claim word and input flags of task head + l) and takes a ready one with a CAS on ITS claim word - claims of
      // different tasks do not serialise on one word.  Which of the ready ones: the (workgroup id mod 4)-th, so
      // that workgroups arriving together spread over the first few instead of all racing for the first.  The head
      // (every task before it is claimed) is advanced by whoever sees claimed tasks at it.
      int got = 0;
      int spins = 0;
      for (;;) {
        bool any_live = false;
        for (int q = 0; q < FLOW_NQ && got == 0; ++q) {
          const FlowTask* tq = q == 0 ? a.tasks[0] : q == 1 ? a.tasks[1] : q == 2 ? a.tasks[2] : a.tasks[3];
          int* cq = q == 0 ? a.claim[0] : q == 1 ? a.claim[1] : q == 2 ? a.claim[2] : a.claim[3];
          const int nq = q == 0 ? a.count[0] : q == 1 ? a.count[1] : q == 2 ? a.count[2] : a.count[3];
          const int h = __builtin_amdgcn_readfirstlane(flow_ld(fl + FL_HEAD + 32 * q));
          if (h >= nq) continue;
          any_live = true;
          const int x = h + lane;
          bool taken = true, rdy = false;
          FlowTask mine{};
          if (x < nq) {
            taken = flow_ld(cq + x) != 0;
            if (!taken) {
              mine = tq[x];
              rdy = flow_ready(mine, fl, a.m);
            }
          }
          const unsigned long long tb = __ballot(taken);
          const int lead = tb == ~0ull ? 64 : __ffsll((long long)~tb) - 1;  // claimed tasks at the head
          if (lead > 0 && lane == 0) atomicMax(fl + FL_HEAD + 32 * q, h + lead);
          unsigned long long rb = __ballot(rdy);
          int skip = (int)(blockIdx.x & 3);
          while (rb != 0ull) {
            unsigned long long pick = rb;
            for (int s = 0; s < skip && (pick & (pick - 1)) != 0ull; ++s) pick &= pick - 1;
            const int sel = __ffsll((long long)pick) - 1;
            int ok = 0;
            if (lane == sel) ok = atomicCAS(cq + x, 0, 1) == 0;
            ok = __shfl(ok, sel, 64);
            if (ok) {
              const int w0 = __shfl(*reinterpret_cast<const int*>(&mine), sel, 64);
              const int w1 = __shfl(*(reinterpret_cast<const int*>(&mine) + 1), sel, 64);
              const int w2 = __shfl(*(reinterpret_cast<const int*>(&mine) + 2), sel, 64);
              if (lane == 0) {
                sh[1] = w0;
                sh[2] = w1;
                sh[3] = w2;
              }
              got = 1;
              break;
            }
            rb &= ~(1ull << sel);
            skip = 0;
          }
        }
        if (got != 0) break;
        if (!any_live) {  // every queue is exhausted
          got = -1;
          break;
        }
        __builtin_amdgcn_s_sleep(16);
        ++spins;
        if ((spins & 63) == 0 && flow_ld(fl + FL_ABORT)) {
          got = -1;
          break;
        }
        if (spins > FLOW_SPIN_LIMIT) {
          if (lane == 0) {
            __hip_atomic_store(fl + FL_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.info) atomicCAS(a.info, 0, GPMI_ERR_INTERNAL);
          }
          got = -1;
          break;
        }
      }
      if (lane == 0) sh[0] = got;
      if (got > 0) {
        // the inputs were written by other CUs: drop what this CU's L1 holds of them
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
    if (sh[0] < 0) break;
    FlowTask t;
    {
      int* w = reinterpret_cast<int*>(&t);
      w[0] = sh[1];
      w[1] = sh[2];
      w[2] = sh[3];
    }
    int* flag;
    const int64_t ld = a.ld;
    if (t.type == FT_T) {
      // rows [32 s, 32 s + 32) of tile (i, k) <- the same rows times invD(k)^T, in place
      double* C = a.A + ((int64_t)t.i * NB + 32 * t.s) * ld + (int64_t)t.k * NB;
      flow_do_T<PROTO>((glb_double_t*)C, (const glb_double_t*)(a.invD + (int64_t)t.k * NB * NB), ld, (lds_double_t*)smem);
      flag = Lcnt + t.i;
    } else if (t.type == FT_U) {
      const int r0 = 64 * (t.s >> 1), c0 = 64 * (t.s & 1);
      double* C = a.A + ((int64_t)t.i * NB + r0) * ld + (int64_t)t.j * NB + c0;
      const double* Ai = a.A + ((int64_t)t.i * NB + r0) * ld + (int64_t)t.k * NB;
      const double* Bj = a.A + ((int64_t)t.j * NB + c0) * ld + (int64_t)t.k * NB;
      flow_do_U<PROTO>((const glb_double_t*)Ai, (const glb_double_t*)Bj, (glb_double_t*)C, ld, (lds_double_t*)smem);
      flag = F + t.i * a.m + t.j;
    } else {
      double* C = a.A + (int64_t)t.i * NB * ld + (int64_t)t.j * NB;
      const double* Ai = a.A + (int64_t)t.i * NB * ld + (int64_t)t.k * OBT * NB;
      const double* Bj = a.A + (int64_t)t.j * NB * ld + (int64_t)t.k * OBT * NB;
      dma128_tile<OP_SUB, PROTO == 3 ? CST_NT : CST_SC1, PROTO == 2 ? LD_SC1 : LD_PLAIN>(Ai, Bj, C, ld, ld, ld, OBT * NB / DMA_BK, smem);
      flag = F + t.i * a.m + t.j;
    }
    // publish: the tile was stored write-through (sc1); every wave's stores acknowledged, workgroup barrier, flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (PROTO == 1 || PROTO == 3) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __hip_atomic_fetch_add(flag, (int)t.fadd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (a.stamp && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a.stamp[8 + (__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 7)] = __builtin_amdgcn_s_memrealtime();
  }
}

// Task lists of a tail of m tile rows (tools/sim/flow_sim.py: build()).  Tile (i, j), P = j / 4: the columns of the
// outer panels before `lazy` are applied as K = 512 chunks, the columns [4 lazy, j) one at a time; lazy = P - 1 for
// the tiles within NEAR rows below their own panel's diagonal block (the K = 512 chunk of the panel just finished
// would sit on the chain: potrf_diag(4P) waits for it), lazy = P elsewhere.
constexpr int FLOW_NEAR = 4;
inline int flow_lazy_panels(int i, int j) {
  const int P = j / OBT;
  const int e = (i < OBT * P + OBT + FLOW_NEAR) ? P - 1 : P;
  return e > 0 ? e : 0;
}

struct FlowQueues {
  std::vector<FlowTask> q[FLOW_NQ];
  double flops_update = 0.0, flops_trsm = 0.0;
};

FlowQueues flow_build(int m) {
  FlowQueues out;
  // virtual time: T(i, k) at k, U(i, j, k) at k + 1/2, Z(i, j, q) at 4 q + 3 + 1/4; queues sorted by it, then by row
  typedef std::tuple<int, int, int, int, int> Key;  // (4 x virtual time, i, j, s, index)
  std::vector<std::pair<Key, FlowTask>> h[FLOW_NQ];
  for (int k = 0; k < m; ++k)
    for (int i = k + 2; i < m; ++i) {
      const int d = i - k;
      const int cls = d <= 3 ? 0 : (d <= 8 ? 1 : 2);
      for (int s = 0; s < 4; ++s) {
        FlowTask t{(uint8_t)FT_T, (uint8_t)s, 1, 0, (uint16_t)i, 0, (uint16_t)k, 0};
        h[cls].push_back({Key(4 * k, i, 0, s, 0), t});
      }
      out.flops_trsm += (double)NB * NB * NB;
      for (int j = k + 1; j <= i; ++j) {
        if (OBT * flow_lazy_panels(i, j) > k) continue;
        // a diagonal tile keeps only its lower triangle current: the upper-right 64 x 64 sub-tile is skipped (as the
        // lower-tile launches of potrf_lower do) and sub-tile 0 counts for two
        for (int s = 0; s < 4; ++s) {
          if (i == j && s == 1) continue;
          FlowTask t{(uint8_t)FT_U, (uint8_t)s, (uint8_t)((i == j && s == 0) ? 2 : 1), 0, (uint16_t)i, (uint16_t)j,
                     (uint16_t)k, 0};
          h[cls].push_back({Key(4 * k + 2, i, j, s, 0), t});
        }
        out.flops_update += (i == j ? 0.75 : 1.0) * 2.0 * NB * NB * NB;
      }
    }
  for (int i = 0; i < m; ++i)
    for (int j = 0; j <= i; ++j)
      for (int q = 0; q < flow_lazy_panels(i, j); ++q) {
        FlowTask t{(uint8_t)FT_Z, 0, (uint8_t)(4 * OBT), 0, (uint16_t)i, (uint16_t)j, (uint16_t)q, 0};
        h[FLOW_NQ - 1].push_back({Key(4 * (OBT * q + OBT - 1) + 1, i, j, 0, 0), t});
        out.flops_update += 2.0 * NB * NB * NB * OBT;
      }
  for (int c = 0; c < FLOW_NQ; ++c) {
    std::sort(h[c].begin(), h[c].end(), [](const std::pair<Key, FlowTask>& x, const std::pair<Key, FlowTask>& y) {
      return x.first < y.first;
    });
    out.q[c].reserve(h[c].size());
    for (auto& e : h[c]) out.q[c].push_back(e.second);
  }
  return out;
}

}  // namespace

bool ensure_masked_pair(gpmi_ctx* c, Lane& L, int k);  // api.hip

// GPMI_FLOW=0 keeps the stream-ordered schedule everywhere; GPMI_FLOW_MIN=<tile rows> (default 8) is the smallest
// tail worth a task kernel
bool potrf_flow_enabled(gpmi_ctx* c, Lane& lane, int m) {
  static const bool on = [] {
    const char* e = std::getenv("GPMI_FLOW");
    return !e || std::atoi(e) != 0;
  }();
  static const int min_rows = [] {
    const char* e = std::getenv("GPMI_FLOW_MIN");
    return e ? std::atoi(e) : 8;
  }();
  return on && m >= min_rows && m < 4096 && ensure_masked_pair(c, lane, 0);
}

void potrf_flow_free(Lane& lane) {
  if (lane.flow_tasks) (void)hipFree(lane.flow_tasks);
  if (lane.flow_flags) (void)hipFree(lane.flow_flags);
  lane.flow_tasks = nullptr;
  lane.flow_flags = nullptr;
  lane.flow_m = 0;
}

// Factor the trailing tile rows [t0, nt) of A (every update by the columns before t0 applied) in place.  The caller
// has ordered lane.stream behind whatever produced that state; on return lane.stream is ordered behind the factor.
// Returns false (nothing enqueued) if the task lists cannot be set up.
bool potrf_flow_tail(gpmi_ctx* c, Lane& lane, double* A, int64_t ld, double* invD, int* info, int nt, int t0) {
  const int m = nt - t0;
  hipStream_t sf = lane.stream, sp = lane.sp[0], su = lane.su[0];
  if (lane.flow_m != m) {
    potrf_flow_free(lane);
    const FlowQueues fq = flow_build(m);
    size_t total = 0;
    for (int q = 0; q < FLOW_NQ; ++q) total += fq.q[q].size();
    if (hipMalloc(&lane.flow_tasks, sizeof(FlowTask) * (total ? total : 1)) != hipSuccess ||
        hipMalloc(&lane.flow_flags, sizeof(int) * flow_flag_ints(m, (int64_t)total)) != hipSuccess) {
      (void)hipGetLastError();
      potrf_flow_free(lane);
      return false;
    }
    size_t off = 0;
    for (int q = 0; q < FLOW_NQ; ++q) {
      lane.flow_off[q] = (int64_t)off;
      lane.flow_count[q] = (int)fq.q[q].size();
      if (!fq.q[q].empty() &&
          hipMemcpy(static_cast<FlowTask*>(lane.flow_tasks) + off, fq.q[q].data(), sizeof(FlowTask) * fq.q[q].size(),
                    hipMemcpyHostToDevice) != hipSuccess) {
        potrf_flow_free(lane);
        return false;
      }
      off += fq.q[q].size();
    }
    lane.flow_flops_update = fq.flops_update;
    lane.flow_flops_trsm = fq.flops_trsm;
    lane.flow_ntasks = (int64_t)total;
    lane.flow_m = m;
  }
  int* fl = lane.flow_flags;
  (void)hipMemsetAsync(fl, 0, sizeof(int) * flow_flag_ints(m, lane.flow_ntasks), sf);
  (void)hipEventRecord(lane.ev_join, sf);
  (void)hipStreamWaitEvent(sp, lane.ev_join, 0);
  (void)hipStreamWaitEvent(su, lane.ev_join, 0);

  double* A0 = A + (int64_t)t0 * NB * ld + (int64_t)t0 * NB;
  double* invD0 = invD + (int64_t)t0 * NB * NB;
  FlowArgs fa{};
  fa.A = A0;
  fa.invD = invD0;
  fa.ld = ld;
  fa.m = m;
  for (int q = 0; q < FLOW_NQ; ++q) {
    fa.tasks[q] = static_cast<const FlowTask*>(lane.flow_tasks) + lane.flow_off[q];
    fa.claim[q] = fl + flow_claim_off(m) + lane.flow_off[q];
    fa.count[q] = lane.flow_count[q];
  }
  fa.flags = fl;
  fa.info = info;
  // one stamped "launch" for the bench's accounting of the trailing updates (class SYRK_REST: not the dominant kernel's
  // name in rocprof's tables): FLOPs of every U and Z task
  fa.stamp = prof_stamp_slot(c, lane.flow_flops_update, 0.0, GPMI_PROF_SYRK_REST);
  const int ncu_u = c->ncu - c->pair_cus[0];
  static const int wgs_per_cu = [] {
    const char* e = std::getenv("GPMI_FLOW_WGS");
    const int v = e ? std::atoi(e) : 2;
    return v > 0 ? v : 2;
  }();
  static const int proto = [] {
    const char* e = std::getenv("GPMI_FLOW_PROTO");
    return e ? std::atoi(e) : 0;
  }();
  const dim3 grid((unsigned)(wgs_per_cu * ncu_u));
  if (proto == 1) hipLaunchKernelGGL(flow_task_kernel<1>, grid, dim3(256), 0, su, fa);
  else if (proto == 2) hipLaunchKernelGGL(flow_task_kernel<2>, grid, dim3(256), 0, su, fa);
  else if (proto == 3) hipLaunchKernelGGL(flow_task_kernel<3>, grid, dim3(256), 0, su, fa);
  else hipLaunchKernelGGL(flow_task_kernel<0>, grid, dim3(256), 0, su, fa);

  // the chain
  int* Lcnt = fl + FL_LCNT;
  int* F = fl + flow_f_off(m);
  {
    ProfScope ps(c, sp, GPMI_PROF_PANEL, (double)m * NB * NB * NB / 3.0 + lane.flow_flops_trsm, 0.0);
    for (int k = 0; k < m; ++k) {
      double* Akk = A0 + (int64_t)k * NB * ld + (int64_t)k * NB;
      double* invDk = invD0 + (int64_t)k * NB * NB;
      launch_potrf_diag(sp, Akk, ld, invDk, info, (t0 + k) * NB);
      if (k + 1 < m) {
        double* A21 = Akk + (int64_t)NB * ld;
        GemmBatch tc;
        tc.ncu_hint = c->pair_cus[0];
        tc.hook.pub = fl + FL_DDONE;
        tc.hook.pub_val = k + 1;
        tc.hook.wait = F + (k + 1) * m + k;
        tc.hook.wait_val = 4 * k;
        tc.hook.abort = fl + FL_ABORT;
        tc.hook.info = info;
        launch_gemm_nt(sp, TILES_RECT, OP_ASSIGN, A21, ld, A21, ld, invDk, NB, 1, 1, NB, nullptr, tc);
        GemmBatch uc;
        uc.hook.pub = Lcnt + k + 1;
        uc.hook.pub_val = 4 * (k + 1);
        uc.hook.wait = F + (k + 1) * m + (k + 1);
        uc.hook.wait_val = 4 * k;
        uc.hook.abort = fl + FL_ABORT;
        uc.hook.info = info;
        launch_gemm_nt(sp, TILES_LOWER, OP_SUB, A21 + NB, ld, A21, ld, A21, ld, 1, 1, NB, nullptr, uc);
      }
    }
  }
  (void)hipEventRecord(lane.ev_panel, sp);
  (void)hipEventRecord(lane.ev_main, su);
  (void)hipStreamWaitEvent(sf, lane.ev_panel, 0);
  (void)hipStreamWaitEvent(sf, lane.ev_main, 0);
  return true;
}
