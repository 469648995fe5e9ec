// Shared by the translation units of the C-ABI (api.hip: handle, lanes, workspaces, lifecycle, instrumentation;
// api_regression.hip: the GpRegressor entry points; api_mix.hip: ChangePoint mixtures and per-point noise; api_linv.hip:
// GpLinearInverter; api_dense.hip: plugin covariance functions and the rank-one append): error macros and the helpers
// every entry point is built from.  Internal - the public interface is include/gpmi.h.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "gpmi_internal.h"

#define HIPCHK(ctx, expr)                                                                   \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess) {                                                                \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                      \
      return (e__ == hipErrorOutOfMemory) ? GPMI_ERR_NOMEM : GPMI_ERR_HIP;                  \
    }                                                                                       \
  } while (0)

#define ARGCHK(ctx, cond, msg) \
  do {                         \
    if (!(cond)) {             \
      (ctx)->err = (msg);      \
      return GPMI_ERR_ARG;     \
    }                          \
  } while (0)

// hipMemset on device memory returns once the fill is ENQUEUED in the null stream, and the library's streams are
// non-blocking: they do not wait for the null stream.  A buffer whose zeros later kernels rely on (the upper 16-blocks
// of the inverse diagonal blocks, the padding of x / y / noise, the zero vectors) is therefore zeroed AND waited for.
// (Round 5: with eight processes time-slicing one device the fill of `invD` arrived milliseconds late and wiped inverse
// blocks that potrf_diag had already written - pivot failures in the second or third diagonal block, a few per
// thousand evaluations; with the device to itself the fill always won the race.)
#define ZERO_SYNC(ctx, ptr, bytes)                          \
  do {                                                      \
    HIPCHK(ctx, hipMemset((ptr), 0, (bytes)));              \
    HIPCHK(ctx, hipStreamSynchronize(nullptr));             \
  } while (0)

// a negative `info` is written by the flag-ordered kernels when a poll timed out: GPMI_INFO_FLOW_TIMEOUT by the tile-task
// factorisation (potrf_flow.hip) - the one case the stream-ordered schedule (GPMI_OPT_NO_FLOW) cures, marked "[flow-tail]"
// in the error text for the caller that wants to repeat the call -, GPMI_ERR_INTERNAL by the triangular sweeps
#define INFOCHK(ctx, inf)                                                                                     \
  do {                                                                                                        \
    if ((inf) != 0 && std::getenv("GPMI_DEBUG_INFO"))                                                         \
      std::fprintf(stderr, "[gpmi] %s: info = %d (n = %lld)\n", __func__, (int)(inf), (long long)(ctx)->n);    \
    if ((inf) < 0) {                                                                                          \
      (ctx)->err = (inf) == GPMI_INFO_FLOW_TIMEOUT                                                            \
                       ? "internal error: the tile-task factorisation timed out [flow-tail]"                  \
                       : "internal error: a flag-ordered triangular sweep timed out";                         \
      return GPMI_ERR_INTERNAL;                                                                               \
    }                                                                                                         \
  } while (0)

constexpr int RED_SLOTS = 8192;  // per-lane result slots for batched evaluations

// one evaluation of the mixture covariance K = sum_m diag(g_m) K_m diag(g_m) + extra I (+ data errors)
struct MixEval {
  int nk;
  const KParams* p;   // nk sub-kernels (their extra_diag is ignored)
  const double* g;    // nk x np device weights (row m: g_m; padding 1 for m = 0, else 0)
  double extra;       // WhiteNoise variance
  double* scratch;    // np x ld
  const double* zero; // np zeros
};

extern thread_local std::string g_create_err;

// Inputs and outputs of a lockstep chunk between the CALLER's (pageable) host arrays and the device: everything is laid
// out in the handle's pinned, device-mapped staging buffer (allocated once per handle) and moved by a copy kernel that
// reads / writes that buffer over the bus - strided rows packed on the fly, no hipMemcpy2DAsync, no pageable memory handed
// to the runtime, and no per-call host allocation (the std::vector pads of round 5 were 0.4 - 0.8 MB of malloc / free per
// call: part of the heap churn behind the 28 ms batch calls of profiles/r06_search_regression.txt; the cure of that
// stall itself is keep_heap_top_once() in api.hip).
// Usage: reserve() the bytes of a chunk; host() + put() / put_copy() / up() for inputs; down() for outputs;
// hipStreamSynchronize; finish() (pinned -> caller).
struct RowStage {
  gpmi_ctx* c;
  hipStream_t s;
  size_t used = 0;
  struct Item {
    char* dst;
    size_t dpitch, off, width, height;
  };
  std::vector<Item> items;
  RowStage(gpmi_ctx* ctx, hipStream_t stream) : c(ctx), s(stream) {}
  // capacity for everything staged until the next finish(); only while nothing is staged
  int reserve(size_t bytes) {
    if ((int64_t)bytes <= c->h_stage_bytes) return GPMI_OK;
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr;
    c->h_stage_bytes = 0;
    HIPCHK(c, hipHostMalloc(&c->h_stage, bytes, hipHostMallocMapped | hipHostMallocPortable));
    c->h_stage_bytes = (int64_t)bytes;
    return GPMI_OK;
  }
  size_t take(size_t bytes) {
    const size_t off = used;
    used += (bytes + 255) & ~(size_t)255;
    return off;
  }
  // pinned scratch of `bytes` (a multiple of 8) the caller fills in place, then put()
  double* host(size_t bytes) { return reinterpret_cast<double*>(reinterpret_cast<char*>(c->h_stage) + take(bytes)); }
  int put(void* dst_dev, const double* pinned, size_t bytes) {
    ARGCHK(c, used <= (size_t)c->h_stage_bytes && bytes % 8 == 0, "internal: staging buffer too small");
    launch_copy_rows(s, pinned, 0, static_cast<double*>(dst_dev), 0, (int64_t)(bytes / 8), 1);
    return GPMI_OK;
  }
  // a contiguous caller array -> device
  int put_copy(void* dst_dev, const void* src, size_t bytes) {
    const size_t padded = (bytes + 7) & ~(size_t)7;
    ARGCHK(c, used + padded + 256 <= (size_t)c->h_stage_bytes, "internal: staging buffer too small");
    double* p = host(padded);
    std::memcpy(p, src, bytes);
    return put(dst_dev, p, padded);
  }
  // caller rows (pitch `spitch` bytes) -> device rows (pitch `dpitch` bytes)
  int up(void* dst_dev, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height) {
    ARGCHK(c, used + width * height + 256 <= (size_t)c->h_stage_bytes, "internal: staging buffer too small");
    double* p = host(width * height);
    for (size_t r = 0; r < height; ++r)
      std::memcpy(reinterpret_cast<char*>(p) + r * width, static_cast<const char*>(src) + r * spitch, width);
    launch_copy_rows(s, p, (int64_t)(width / 8), static_cast<double*>(dst_dev), (int64_t)(dpitch / 8), (int64_t)(width / 8),
                     (int64_t)height);
    return GPMI_OK;
  }
  // `height` device rows of `width` bytes (pitch `spitch`) -> caller rows (pitch `dpitch`), after the sync and finish()
  int down(void* dst, size_t dpitch, const void* src_dev, size_t spitch, size_t width, size_t height) {
    ARGCHK(c, used + width * height + 256 <= (size_t)c->h_stage_bytes, "internal: staging buffer too small");
    const size_t off = take(width * height);
    launch_copy_rows(s, static_cast<const double*>(src_dev), (int64_t)(spitch / 8),
                     reinterpret_cast<double*>(reinterpret_cast<char*>(c->h_stage) + off), (int64_t)(width / 8),
                     (int64_t)(width / 8), (int64_t)height);
    items.push_back({static_cast<char*>(dst), dpitch, off, width, height});
    return GPMI_OK;
  }
  int flush() { return GPMI_OK; }
  // after the stream has been synchronised
  void finish() {
    const char* base = reinterpret_cast<const char*>(c->h_stage);
    for (const Item& it : items)
      for (size_t r = 0; r < it.height; ++r) std::memcpy(it.dst + r * it.dpitch, base + it.off + r * it.width, it.width);
    items.clear();
    used = 0;
  }
};

// api.hip
int lane_streams(gpmi_ctx* c, Lane& L);
bool ensure_masked_pair(gpmi_ctx* c, Lane& L, int k);
int lane_masked_streams(gpmi_ctx* c, Lane& L);
int lane_alloc(gpmi_ctx* c, Lane& L);
void lane_free(Lane& L);
int ensure_lanes(gpmi_ctx* c, size_t count);
void free_data(gpmi_ctx* c);
int make_params(gpmi_ctx* c, int kernel, const double* theta, int n_theta, double extra, KParams& p);
int set_device(gpmi_ctx* c);
void build_mix_square(gpmi_ctx* c, hipStream_t s, const MixEval& mx, double* dst, bool lower_only);
int enqueue_factor_and_forward(gpmi_ctx* c, Lane& L, const KParams& p, const double* mu_dev, double mu_const, int slot, bool allow_lookahead = true,
                               const MixEval* mix = nullptr, bool prebuild_inv2 = false, double* backward_out = nullptr,
                               double* early_identity = nullptr);
int ensure_second_matrix(gpmi_ctx* c, Lane& L);
int ensure_inv2(gpmi_ctx* c, Lane& F, hipStream_t s);
int ensure_trsm_panel(gpmi_ctx* c, int64_t rows);
int enqueue_inverse_factor(gpmi_ctx* c, Lane& L, Lane& F);
int ensure_batch_ws(gpmi_ctx* c, int want);
int ensure_batch_grad_ws(gpmi_ctx* c, int want, int n_theta);
int ensure_query_ws(gpmi_ctx* c, int64_t mp);
void linv_free(LinvState& S);  // api_linv.hip
// v <- -v;  out_i = alpha_i^2 - iK_ii (one problem / problem z of a batch)
void launch_negate(hipStream_t s, double* v, int64_t n);
void launch_qdiag(hipStream_t s, const double* iK, int64_t ld, const double* alpha, double* out, int64_t n);
void launch_qdiag_batched(hipStream_t s, int batch, const double* iK, int64_t ld, const double* alpha, double* out, int64_t n,
                          int64_t sMat, int64_t sVec, int64_t sOut);
