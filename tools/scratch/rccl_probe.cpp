#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstdio>
int main() {
  int n = 0; hipGetDeviceCount(&n); printf("devices %d\n", n);
  hipSetDevice(0);
  double* d; hipMalloc(&d, 64);
  ncclUniqueId id; ncclResult_t r = ncclGetUniqueId(&id); printf("getUniqueId %d\n", (int)r);
  ncclComm_t comm; r = ncclCommInitRank(&comm, 1, id, 0); printf("commInitRank %d %s\n", (int)r, ncclGetErrorString(r));
  if (r == ncclSuccess) {
    r = ncclAllGather(d, d + 4, 4, ncclDouble, comm, 0); printf("allgather %d\n", (int)r);
    hipDeviceSynchronize(); ncclCommDestroy(comm);
  }
  return 0;
}
