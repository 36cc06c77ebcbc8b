// rm_gather.cpp — the multi-GPU gather of the C ABI: one host process drives the GPUs of a node, every GPU renders its
// interleaved row tiles (rm_render_tiles) and the tiles travel to one root GPU over RCCL (xGMI, point to point: every
// peer has its own link to the root), where rm_deinterleave puts them into frame order.
//
// No reference counterpart: the reference renders whole frames on one GPU.  One communicator per device from
// ncclCommInitAll; a gather is ONE group of ncclSend (on each peer's stream) / ncclRecv (on the root's stream) pairs — no
// collective over all ranks, no staging through the host.  librccl is loaded at run time (dlopen) so that the library
// has no link-time dependency on it and shares the process's RCCL with whoever loaded one first (PyTorch ships its own).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>
#include <vector>

#include "../../include/raymarcher_amd.h"
#include "rm_internal.h"

using namespace rm;

namespace {
struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
std::mutex g_rcclMu;
Rccl g_rccl;

int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rcclMu);
  if (g_rccl.handle) return RM_OK;
  void *h = nullptr;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) { set_error(std::string("librccl could not be loaded: ") + dlerror()); return RM_ERR_UNSUPPORTED; }
  Rccl r;
  r.handle = h;
#define RM_SYM(field, name)                                                                    \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                                 \
  if (!r.field) { set_error(std::string("librccl lacks ") + name); dlclose(h); return RM_ERR_UNSUPPORTED; }
  RM_SYM(CommInitAll, "ncclCommInitAll")
  RM_SYM(CommDestroy, "ncclCommDestroy")
  RM_SYM(GroupStart, "ncclGroupStart")
  RM_SYM(GroupEnd, "ncclGroupEnd")
  RM_SYM(Send, "ncclSend")
  RM_SYM(Recv, "ncclRecv")
  RM_SYM(GetErrorString, "ncclGetErrorString")
#undef RM_SYM
  g_rccl = r;
  return RM_OK;
}

#define HIP_OK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                               \
      return RM_ERR_DEVICE;                                                                       \
    }                                                                                             \
  } while (0)
#define NCCL_OK(expr)                                                                             \
  do {                                                                                            \
    ncclResult_t r_ = (expr);                                                                     \
    if (r_ != ncclSuccess) {                                                                      \
      set_error(std::string(#expr) + ": " + g_rccl.GetErrorString(r_));                           \
      return RM_ERR_DEVICE;                                                                       \
    }                                                                                             \
  } while (0)
}  // namespace

struct RmGather {
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;  // empty for a single device (nothing to communicate)
};

extern "C" {

int rm_gather_slot_rows(int H, int tileRows, int numShards) {
  if (H <= 0 || tileRows <= 0 || numShards <= 0) return 0;
  return shard_rows(H, tileRows, 0, numShards);  // shard 0 owns the most rows: equal slots of that size hold every shard
}

int rm_gather_create(const int *devices, int numDevices, RmGather **out) {
  if (!devices || !out || numDevices <= 0 || numDevices > 64) { set_error("bad device list"); return RM_ERR_INVALID_ARGUMENT; }
  int have = 0;
  HIP_OK(hipGetDeviceCount(&have));
  for (int i = 0; i < numDevices; i++) {
    if (devices[i] < 0 || devices[i] >= have) { set_error("device index out of range"); return RM_ERR_INVALID_ARGUMENT; }
    for (int j = 0; j < i; j++)
      if (devices[j] == devices[i]) { set_error("a device appears twice (one shard per GPU)"); return RM_ERR_INVALID_ARGUMENT; }
  }
  RmGather *g = new RmGather;
  g->devices.assign(devices, devices + numDevices);
  if (numDevices > 1) {
    if (int st = load_rccl()) { delete g; return st; }
    g->comms.resize(numDevices);
    ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), numDevices, devices);
    if (r != ncclSuccess) {
      set_error(std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
      delete g;
      return RM_ERR_DEVICE;
    }
  }
  *out = g;
  return RM_OK;
}

void rm_gather_destroy(RmGather *g) {
  if (!g) return;
  for (ncclComm_t c : g->comms) (void)g_rccl.CommDestroy(c);
  delete g;
}

int rm_gather_tiles(RmGather *g, const float *const *d_tiles, float *d_gathered, int W, int H, int tileRows, int root,
                    void *const *streams) {
  if (!g || !d_tiles || !d_gathered || W <= 0 || H <= 0 || tileRows <= 0) { set_error("bad gather arguments"); return RM_ERR_INVALID_ARGUMENT; }
  const int n = (int)g->devices.size();
  if (root < 0 || root >= n) { set_error("root out of range"); return RM_ERR_INVALID_ARGUMENT; }
  const size_t slotFloats = (size_t)rm_gather_slot_rows(H, tileRows, n) * W * 4;
  int caller = 0;
  HIP_OK(hipGetDevice(&caller));
  auto stream = [&](int k) { return static_cast<hipStream_t>(streams ? streams[k] : nullptr); };
  // the root's own tiles: a copy inside its memory, on its stream
  {
    const size_t count = (size_t)shard_rows(H, tileRows, root, n) * W * 4;
    if (count && !d_tiles[root]) { set_error("null tile buffer"); return RM_ERR_INVALID_ARGUMENT; }
    HIP_OK(hipSetDevice(g->devices[root]));
    if (count) HIP_OK(hipMemcpyAsync(d_gathered + root * slotFloats, d_tiles[root], count * sizeof(float), hipMemcpyDeviceToDevice, stream(root)));
  }
  if (n > 1) {
    // a failure inside the group still closes it (an open group would swallow the process's next RCCL calls) and restores
    // the caller's device
    int rc = RM_OK;
    std::string why;
    ncclResult_t r = g_rccl.GroupStart();
    if (r != ncclSuccess) { set_error(std::string("ncclGroupStart: ") + g_rccl.GetErrorString(r)); (void)hipSetDevice(caller); return RM_ERR_DEVICE; }
    for (int k = 0; k < n && rc == RM_OK; k++) {
      if (k == root) continue;
      const size_t count = (size_t)shard_rows(H, tileRows, k, n) * W * 4;
      if (!count) continue;
      if (!d_tiles[k]) { rc = RM_ERR_INVALID_ARGUMENT; why = "null tile buffer"; break; }
      // the send is ordered behind shard k's render on ITS stream; the receive lands in slot k of the root's buffer
      hipError_t e = hipSetDevice(g->devices[k]);
      if (e == hipSuccess) {
        r = g_rccl.Send(d_tiles[k], count, ncclFloat, root, g->comms[k], stream(k));
        if (r != ncclSuccess) { rc = RM_ERR_DEVICE; why = std::string("ncclSend: ") + g_rccl.GetErrorString(r); break; }
        e = hipSetDevice(g->devices[root]);
      }
      if (e != hipSuccess) { rc = RM_ERR_DEVICE; why = std::string("hipSetDevice: ") + hipGetErrorString(e); break; }
      r = g_rccl.Recv(d_gathered + k * slotFloats, count, ncclFloat, k, g->comms[root], stream(root));
      if (r != ncclSuccess) { rc = RM_ERR_DEVICE; why = std::string("ncclRecv: ") + g_rccl.GetErrorString(r); break; }
    }
    r = g_rccl.GroupEnd();
    if (rc == RM_OK && r != ncclSuccess) { rc = RM_ERR_DEVICE; why = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r); }
    if (rc != RM_OK) { set_error(why); (void)hipSetDevice(caller); return rc; }
  }
  HIP_OK(hipSetDevice(caller));
  return RM_OK;
}

}  // extern "C"
