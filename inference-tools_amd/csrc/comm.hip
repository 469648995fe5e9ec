// RCCL (over xGMI) result gather for multi-GPU sweeps: the one collective of the GP path.
//
// Reference counterpart: results returned by multiprocessing.Pool.map (regression.py:600-601) and the
// (theta, log-prob) tuples sent over Pipes by the tempering processes (mcmc/parallel.py:195-201).
// One process per GPU; rank 0 creates the RCCL unique id, the host bootstrap (any CPU channel, e.g.
// torch.distributed gloo) hands it to the other ranks, and every rank calls gpmi_comm_init.
// librccl is loaded lazily with dlopen so that single-GPU users never pay for it.
#include <dlfcn.h>

#include <cstring>

#include "gpmi_internal.h"

namespace {

struct ncclUniqueIdLike {
  char internal[128];
};
typedef int (*fn_get_unique_id)(ncclUniqueIdLike*);
typedef int (*fn_comm_init_rank)(void**, int, ncclUniqueIdLike, int);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_comm_count)(void*, int*);
typedef int (*fn_comm_destroy)(void*);
typedef const char* (*fn_get_error_string)(int);

struct Rccl {
  void* lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_broadcast broadcast = nullptr;   // optional symbols: absent ones make their entry points fail, not the library
  fn_comm_count comm_count = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_get_error_string get_error_string = nullptr;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.get_unique_id = (fn_get_unique_id)dlsym(r.lib, "ncclGetUniqueId");
      r.comm_init_rank = (fn_comm_init_rank)dlsym(r.lib, "ncclCommInitRank");
      r.all_gather = (fn_all_gather)dlsym(r.lib, "ncclAllGather");
      r.broadcast = (fn_broadcast)dlsym(r.lib, "ncclBroadcast");
      r.comm_count = (fn_comm_count)dlsym(r.lib, "ncclCommCount");
      r.comm_destroy = (fn_comm_destroy)dlsym(r.lib, "ncclCommDestroy");
      r.get_error_string = (fn_get_error_string)dlsym(r.lib, "ncclGetErrorString");
      if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy) r.lib = nullptr;
    }
  }
  return r.lib ? &r : nullptr;
}

constexpr int NCCL_DOUBLE = 8;  // ncclFloat64 (rccl.h)

int hip_fail(gpmi_ctx* c, const char* what, hipError_t e) {
  c->err = std::string(what) + ": " + hipGetErrorString(e);
  return GPMI_ERR_HIP;
}

int fail(gpmi_ctx* c, const char* what, int code) {
  Rccl* r = rccl();
  c->err = std::string(what) + ": " +
           ((r && r->get_error_string) ? r->get_error_string(code) : "RCCL error");
  return GPMI_ERR_HIP;
}

}  // namespace

extern "C" {

int gpmi_comm_unique_id(char* id_out) {
  Rccl* r = rccl();
  if (!r || !id_out) return GPMI_ERR_ARG;
  ncclUniqueIdLike id;
  if (r->get_unique_id(&id) != 0) return GPMI_ERR_HIP;
  std::memcpy(id_out, id.internal, 128);
  return GPMI_OK;
}

int gpmi_comm_init(gpmi_ctx* c, int rank, int world, const char* id_bytes) {
  if (!c) return GPMI_ERR_ARG;
  Rccl* r = rccl();
  if (!r) {
    c->err = "librccl.so could not be loaded";
    return GPMI_ERR_NODEVICE;
  }
  if (!id_bytes || world < 1 || rank < 0 || rank >= world) {
    c->err = "bad rank / world / id";
    return GPMI_ERR_ARG;
  }
  if (hipError_t e = hipSetDevice(c->device); e != hipSuccess) return hip_fail(c, "hipSetDevice", e);
  if (c->comm) {
    r->comm_destroy(c->comm);
    c->comm = nullptr;
  }
  ncclUniqueIdLike id;
  std::memcpy(id.internal, id_bytes, 128);
  int rc = r->comm_init_rank(&c->comm, world, id, rank);
  if (rc != 0) return fail(c, "ncclCommInitRank", rc);
  c->comm_rank = rank;
  c->comm_world = world;
  if (!c->comm_stream) {
    if (hipError_t e = hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking); e != hipSuccess)
      return hip_fail(c, "hipStreamCreateWithFlags (gather stream)", e);
  }
  return GPMI_OK;
}

int gpmi_comm_allgather(gpmi_ctx* c, const double* send_host, double* recv_host, int64_t count) {
  if (!c) return GPMI_ERR_ARG;
  Rccl* r = rccl();
  if (!r || !c->comm) {
    c->err = "gpmi_comm_init has not been called";
    return GPMI_ERR_ARG;
  }
  if (!send_host || !recv_host || count <= 0) {
    c->err = "bad buffers";
    return GPMI_ERR_ARG;
  }
  if (hipError_t e = hipSetDevice(c->device); e != hipSuccess) return hip_fail(c, "hipSetDevice", e);
  const int64_t need = count * (1 + (int64_t)c->comm_world);
  if (c->comm_buf_doubles < need) {
    if (c->comm_buf) (void)hipFree(c->comm_buf);
    c->comm_buf = nullptr;
    c->comm_buf_doubles = 0;
    if (hipMalloc(&c->comm_buf, sizeof(double) * need) != hipSuccess) {
      c->err = "out of device memory for the gather buffer";
      return GPMI_ERR_NOMEM;
    }
    c->comm_buf_doubles = need;
  }
  hipStream_t s = c->comm_stream;
  double* send = c->comm_buf;
  double* recv = c->comm_buf + count;
  if (hipError_t e = hipMemcpyAsync(send, send_host, sizeof(double) * count, hipMemcpyHostToDevice, s); e != hipSuccess)
    return hip_fail(c, "gather upload", e);
  int rc = r->all_gather(send, recv, (size_t)count, NCCL_DOUBLE, c->comm, s);
  if (rc != 0) return fail(c, "ncclAllGather", rc);
  if (hipError_t e = hipMemcpyAsync(recv_host, recv, sizeof(double) * count * c->comm_world, hipMemcpyDeviceToHost, s);
      e != hipSuccess)
    return hip_fail(c, "gather download", e);
  if (hipError_t e = hipStreamSynchronize(s); e != hipSuccess) return hip_fail(c, "gather synchronise", e);
  return GPMI_OK;
}

// Start-up distribution of the data set: rank `root`'s buffer to every rank, in place.  The
// reference ships x, y, y_err to its workers by pickling the whole GpRegressor into them (regression.py:597-601,
// mcmc/parallel.py:127-136); here it is one ncclBroadcast of a few hundred kilobytes over xGMI.
int gpmi_comm_broadcast(gpmi_ctx* c, double* buf_host, int64_t count, int root) {
  if (!c) return GPMI_ERR_ARG;
  Rccl* r = rccl();
  if (!r || !c->comm) {
    c->err = "gpmi_comm_init has not been called";
    return GPMI_ERR_ARG;
  }
  if (!r->broadcast) {
    c->err = "this librccl has no ncclBroadcast";
    return GPMI_ERR_NODEVICE;
  }
  if (!buf_host || count <= 0 || root < 0 || root >= c->comm_world) {
    c->err = "bad buffer / count / root";
    return GPMI_ERR_ARG;
  }
  if (hipError_t e = hipSetDevice(c->device); e != hipSuccess) return hip_fail(c, "hipSetDevice", e);
  if (c->comm_buf_doubles < count) {
    if (c->comm_buf) (void)hipFree(c->comm_buf);
    c->comm_buf = nullptr;
    c->comm_buf_doubles = 0;
    if (hipMalloc(&c->comm_buf, sizeof(double) * count) != hipSuccess) {
      c->err = "out of device memory for the broadcast buffer";
      return GPMI_ERR_NOMEM;
    }
    c->comm_buf_doubles = count;
  }
  hipStream_t s = c->comm_stream;
  if (c->comm_rank == root)
    if (hipError_t e = hipMemcpyAsync(c->comm_buf, buf_host, sizeof(double) * count, hipMemcpyHostToDevice, s);
        e != hipSuccess)
      return hip_fail(c, "broadcast upload", e);
  int rc = r->broadcast(c->comm_buf, c->comm_buf, (size_t)count, NCCL_DOUBLE, root, c->comm, s);
  if (rc != 0) return fail(c, "ncclBroadcast", rc);
  if (c->comm_rank != root)
    if (hipError_t e = hipMemcpyAsync(buf_host, c->comm_buf, sizeof(double) * count, hipMemcpyDeviceToHost, s);
        e != hipSuccess)
      return hip_fail(c, "broadcast download", e);
  if (hipError_t e = hipStreamSynchronize(s); e != hipSuccess) return hip_fail(c, "broadcast synchronise", e);
  return GPMI_OK;
}

// The number of ranks RCCL itself sees in the communicator (ncclCommCount): what a scaling run prints beside the
// launcher's WORLD_SIZE.
int gpmi_comm_count(gpmi_ctx* c, int* ranks) {
  if (!c || !ranks) return GPMI_ERR_ARG;
  Rccl* r = rccl();
  if (!r || !c->comm || !r->comm_count) {
    c->err = "no communicator (gpmi_comm_init), or this librccl has no ncclCommCount";
    return GPMI_ERR_ARG;
  }
  int rc = r->comm_count(c->comm, ranks);
  if (rc != 0) return fail(c, "ncclCommCount", rc);
  return GPMI_OK;
}

int gpmi_comm_destroy(gpmi_ctx* c) {
  if (!c) return GPMI_ERR_ARG;
  if (c->comm) {  // librccl is only ever loaded by a handle that asked for a communicator
    Rccl* r = rccl();
    if (r) r->comm_destroy(c->comm);
  }
  c->comm = nullptr;
  if (c->comm_buf) (void)hipFree(c->comm_buf);
  c->comm_buf = nullptr;
  c->comm_buf_doubles = 0;
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  c->comm_stream = nullptr;
  return GPMI_OK;
}

}  // extern "C"
