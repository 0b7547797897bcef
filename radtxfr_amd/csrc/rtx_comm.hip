// Single-process collectives over the GPUs of one node: rtx_comm_init_all / rtx_allgather /
// rtx_comm_destroy. One host process drives several devices -- the reference's scripts are plain Python programs with no
// launcher (Generate_LWIR_TUD.py:117-150) -- and the ONE exchange of the wavenumber-sharded path, the all-gather of the
// packed [tau, L-up, L-down] blocks, runs either through RCCL (ncclCommInitAll + a grouped ncclAllGather over xGMI) or
// through a peer-copy fan-out (every device pulls its peers' blocks with hipMemcpyPeerAsync on its own stream): the
// payloads are small (8 MB per rank at C3 on 8 GPUs) and xGMI is point-to-point, so 7 direct copies per device on 7
// independent links are as good a schedule as a ring. RCCL is resolved at run time (dlopen: the library the process
// already holds -- PyTorch ships one -- else librccl.so), so libradtxfr_hip.so has no link-time dependency on it.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "rtx_common.h"

typedef void* nccl_comm_t;
struct NcclApi {
  void* lib;
  int (*CommInitAll)(nccl_comm_t*, int, const int*);
  int (*CommDestroy)(nccl_comm_t);
  int (*GroupStart)(void);
  int (*GroupEnd)(void);
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
  const char* (*GetErrorString)(int);
};

static bool load_nccl(NcclApi* api) {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;  // the copy the process already uses (torch's), if any
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  if (!h) return false;
  api->lib = h;
  api->CommInitAll = (int (*)(nccl_comm_t*, int, const int*))dlsym(h, "ncclCommInitAll");
  api->CommDestroy = (int (*)(nccl_comm_t))dlsym(h, "ncclCommDestroy");
  api->GroupStart = (int (*)(void))dlsym(h, "ncclGroupStart");
  api->GroupEnd = (int (*)(void))dlsym(h, "ncclGroupEnd");
  api->AllGather = (int (*)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t))dlsym(h, "ncclAllGather");
  api->GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  return api->CommInitAll && api->CommDestroy && api->GroupStart && api->GroupEnd && api->AllGather;
}

struct rtx_comm {
  int ndev;
  std::vector<int> devs;
  int backend;  // 0 peer copies, 1 RCCL
  NcclApi nccl;
  std::vector<nccl_comm_t> comms;
  std::vector<hipEvent_t> ready;  // per rank: its send block is complete (peer backend)
  std::vector<hipEvent_t> done;   // per rank: it has pulled every peer's block (peer backend)
};

// puts the calling thread's current device back on every way out of a function that walks the devices
struct DeviceRestore {
  int dev = -1;
  DeviceRestore() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceRestore() { if (dev >= 0) (void)hipSetDevice(dev); }
};

extern "C" int rtx_comm_destroy(rtx_comm* c) {
  if (!c) return 0;
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (int i = 0; i < c->ndev; ++i) {
    (void)hipSetDevice(c->devs[i]);
    if (c->backend == 1 && i < (int)c->comms.size() && c->comms[i]) c->nccl.CommDestroy(c->comms[i]);
    if (i < (int)c->ready.size() && c->ready[i]) (void)hipEventDestroy(c->ready[i]);
    if (i < (int)c->done.size() && c->done[i]) (void)hipEventDestroy(c->done[i]);
  }
  (void)hipSetDevice(cur);
  delete c;
  return 0;
}

extern "C" int rtx_comm_backend(const rtx_comm* c) { return c ? c->backend : -1; }

// backend: -1 = RCCL when it loads and the devices are distinct, else peer copies; 0 = peer copies; 1 = RCCL or fail.
extern "C" int rtx_comm_init_all(int ndev, const int* devs_h, int backend, rtx_comm** out) {
  if (!out) RTX_FAIL("out is NULL");
  *out = nullptr;
  if (ndev < 1 || ndev > 64 || !devs_h) RTX_FAIL("ndev=%d", ndev);
  int n_vis = 0;
  RTX_HIP(hipGetDeviceCount(&n_vis));
  bool distinct = true;
  for (int i = 0; i < ndev; ++i) {
    if (devs_h[i] < 0 || devs_h[i] >= n_vis) RTX_FAIL("device %d of the list is %d; %d devices are visible", i, devs_h[i], n_vis);
    for (int j = 0; j < i; ++j) distinct = distinct && devs_h[j] != devs_h[i];
  }
  if (const char* e = getenv("RADTXFR_COMM")) {
    if (!strcmp(e, "peer")) backend = 0;
    if (!strcmp(e, "rccl")) backend = 1;
  }
  DeviceRestore restore;
  const int cur = restore.dev;
  rtx_comm* c = new rtx_comm();
  c->ndev = ndev;
  c->devs.assign(devs_h, devs_h + ndev);
  c->backend = 0;
  memset(&c->nccl, 0, sizeof(c->nccl));
  if (backend != 0) {
    if (!distinct) {
      if (backend == 1) { delete c; RTX_FAIL("RCCL needs distinct devices (a device may repeat only with the peer-copy backend)"); }
    } else if (load_nccl(&c->nccl)) {
      c->comms.assign(ndev, nullptr);
      const int rc = c->nccl.CommInitAll(c->comms.data(), ndev, c->devs.data());
      if (rc == 0) c->backend = 1;
      else if (backend == 1) {
        rtx_set_error("ncclCommInitAll failed: %s", c->nccl.GetErrorString ? c->nccl.GetErrorString(rc) : "?");
        delete c;
        return 1;
      } else c->comms.clear();
    } else if (backend == 1) { delete c; RTX_FAIL("librccl.so could not be loaded: %s", dlerror()); }
  }
  c->ready.assign(ndev, nullptr);
  c->done.assign(ndev, nullptr);
  for (int i = 0; i < ndev; ++i) {
    hipError_t e = hipSetDevice(c->devs[i]);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming);
    if (e != hipSuccess) {
      rtx_set_error("event on device %d: %s", c->devs[i], hipGetErrorString(e));
      if (cur >= 0) (void)hipSetDevice(cur);
      rtx_comm_destroy(c);
      return 1;
    }
  }
  *out = c;
  return 0;
}

// Every rank i contributes sendbufs_h[i][0 .. count) (float32, on device devs[i]); afterwards recvbufs_h[j] on every
// device j holds the ndev blocks in rank order: recvbufs_h[j][i*count + t] = sendbufs_h[i][t]. streams_h[i] is rank i's
// stream: the send block must be complete in its order, and the gathered block is complete in its order afterwards.
// Asynchronous with respect to the host. A send buffer may alias its own slot of the receive buffer (in-place).
extern "C" int rtx_allgather(rtx_comm* c, const void* const* sendbufs_h, void* const* recvbufs_h, int64_t count, void* const* streams_h) {
  if (!c || !sendbufs_h || !recvbufs_h || !streams_h) RTX_FAIL("a required pointer is NULL");
  if (count < 0) RTX_FAIL("count=%lld", (long long)count);
  if (count == 0) return 0;
  DeviceRestore restore;
  if (c->backend == 1) {
    int rc = c->nccl.GroupStart();
    hipError_t he = hipSuccess;
    for (int i = 0; i < c->ndev && rc == 0 && he == hipSuccess; ++i) {
      he = hipSetDevice(c->devs[i]);
      if (he == hipSuccess)
        rc = c->nccl.AllGather(sendbufs_h[i], recvbufs_h[i], (size_t)count, 7 /* ncclFloat32 */, c->comms[i], (hipStream_t)streams_h[i]);
    }
    const int rc2 = c->nccl.GroupEnd();  // the group is closed on every path
    if (he != hipSuccess) RTX_FAIL("hipSetDevice: %s", hipGetErrorString(he));
    if (rc != 0 || rc2 != 0) RTX_FAIL("ncclAllGather failed: %s", c->nccl.GetErrorString ? c->nccl.GetErrorString(rc ? rc : rc2) : "?");
    return 0;
  }
  const size_t bytes = (size_t)count * sizeof(float);
  for (int i = 0; i < c->ndev; ++i) {
    RTX_HIP(hipSetDevice(c->devs[i]));
    RTX_HIP(hipEventRecord(c->ready[i], (hipStream_t)streams_h[i]));
  }
  for (int j = 0; j < c->ndev; ++j) {  // device j pulls every block on its own stream
    RTX_HIP(hipSetDevice(c->devs[j]));
    hipStream_t st = (hipStream_t)streams_h[j];
    for (int i = 0; i < c->ndev; ++i) {
      if (i != j) RTX_HIP(hipStreamWaitEvent(st, c->ready[i], 0));
      char* dst = (char*)recvbufs_h[j] + (size_t)i * bytes;
      if ((const void*)dst == sendbufs_h[i]) continue;  // in place
      RTX_HIP(hipMemcpyPeerAsync(dst, c->devs[j], sendbufs_h[i], c->devs[i], bytes, st));
    }
    RTX_HIP(hipEventRecord(c->done[j], st));
  }
  // a rank's later work on its stream (the next step's kernels write its send block again) waits until every peer has
  // read that block: as with a collective, the call is complete in a stream's order only when all ranks are through it
  for (int i = 0; i < c->ndev; ++i) {
    hipStream_t st = (hipStream_t)streams_h[i];
    for (int j = 0; j < c->ndev; ++j)
      if (j != i) RTX_HIP(hipStreamWaitEvent(st, c->done[j], 0));
  }
  return 0;
}
