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
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
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
  RM_SYM(CommAbort, "ncclCommAbort")
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
  std::vector<ncclComm_t> comms;  // empty for a single device without RM_GATHER_FORCE_COMM (nothing to communicate)
  bool selfSendRecv = false;      // RM_GATHER_FORCE_COMM: the root's own tiles also travel through ncclSend / ncclRecv
  bool broken = false;            // a grouped call failed: the communicators were aborted
};

namespace {
// Restores the calling thread's current device on every exit path.
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
  ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

// One gather of `count[k]` elements of `type` from src[k] (device k) into dst + k·slotElems (device root).
int gather_elems(RmGather *g, const void *const *src, void *dst, const std::vector<size_t> &count, size_t slotElems, size_t elemBytes,
                 ncclDataType_t type, int root, void *const *streams) {
  const int n = (int)g->devices.size();
  if (g->broken) { set_error("this RmGather was aborted by an earlier failure; destroy and re-create it"); return RM_ERR_DEVICE; }
  // everything that can be checked is checked BEFORE the group opens: a failure inside it would leave unmatched sends behind
  for (int k = 0; k < n; k++)
    if (count[k] && !src[k]) { set_error("null tile buffer"); return RM_ERR_INVALID_ARGUMENT; }
  for (int k = 0; k < n; k++)
    if (count[k] && !device_accessible(src[k])) { set_error("a tile buffer is not device-accessible memory"); return RM_ERR_INVALID_ARGUMENT; }
  if (!device_accessible(dst)) { set_error("d_gathered is not device-accessible memory"); return RM_ERR_INVALID_ARGUMENT; }
  DeviceGuard guard;
  auto stream = [&](int k) { return static_cast<hipStream_t>(streams ? streams[k] : nullptr); };
  char *out = static_cast<char *>(dst);
  const bool grouped = !g->comms.empty();
  if (!(grouped && g->selfSendRecv) && count[root]) {  // the root's own tiles: a copy inside its memory, on its stream
    HIP_OK(hipSetDevice(g->devices[root]));
    HIP_OK(hipMemcpyAsync(out + (size_t)root * slotElems * elemBytes, src[root], count[root] * elemBytes, hipMemcpyDeviceToDevice, stream(root)));
  }
  if (!grouped) return RM_OK;
  std::string why;
  ncclResult_t r = g_rccl.GroupStart();
  if (r != ncclSuccess) { set_error(std::string("ncclGroupStart: ") + g_rccl.GetErrorString(r)); return RM_ERR_DEVICE; }
  for (int k = 0; k < n && why.empty(); k++) {
    if ((k == root && !g->selfSendRecv) || !count[k]) continue;
    // the send is ordered behind shard k's render on ITS stream; the receive lands in slot k of the root's buffer
    hipError_t e = hipSetDevice(g->devices[k]);
    if (e != hipSuccess) { why = std::string("hipSetDevice: ") + hipGetErrorString(e); break; }
    r = g_rccl.Send(src[k], count[k], type, root, g->comms[k], stream(k));
    if (r != ncclSuccess) { why = std::string("ncclSend: ") + g_rccl.GetErrorString(r); break; }
    e = hipSetDevice(g->devices[root]);
    if (e != hipSuccess) { why = std::string("hipSetDevice: ") + hipGetErrorString(e); break; }
    r = g_rccl.Recv(out + (size_t)k * slotElems * elemBytes, count[k], type, k, g->comms[root], stream(root));
    if (r != ncclSuccess) { why = std::string("ncclRecv: ") + g_rccl.GetErrorString(r); break; }
  }
  r = g_rccl.GroupEnd();
  if (why.empty() && r != ncclSuccess) why = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(r);
  if (!why.empty()) {
    // a send whose receive was never posted (or the reverse) would hang its stream: abort the communicators, which fails
    // the outstanding operations, and refuse further use of this object
    for (ncclComm_t c : g->comms) (void)g_rccl.CommAbort(c);
    g->comms.clear();
    g->broken = true;
    set_error(why + " (communicators aborted)");
    return RM_ERR_DEVICE;
  }
  return RM_OK;
}
}  // namespace

extern "C" {

int rm_gather_slot_rows(int H, int tileRows, int numShards) {
  if (H <= 0 || tileRows <= 0 || numShards <= 0) return 0;
  return max_shard_rows(H, tileRows, numShards, root_relief());  // equal slots of the largest shard's size hold every shard
}

int rm_gather_create_ex(const int *devices, int numDevices, unsigned flags, RmGather **out) {
  if (!devices || !out || numDevices <= 0 || numDevices > 64) { set_error("bad device list"); return RM_ERR_INVALID_ARGUMENT; }
  if (flags & ~(unsigned)RM_GATHER_FORCE_COMM) { set_error("unknown gather flag"); return RM_ERR_INVALID_ARGUMENT; }
  int have = 0;
  HIP_OK(hipGetDeviceCount(&have));
  for (int i = 0; i < numDevices; i++) {
    if (devices[i] < 0 || devices[i] >= have) { set_error("device index out of range"); return RM_ERR_INVALID_ARGUMENT; }
    for (int j = 0; j < i; j++)
      if (devices[j] == devices[i]) { set_error("a device appears twice (one shard per GPU)"); return RM_ERR_INVALID_ARGUMENT; }
  }
  RmGather *g = new RmGather;
  g->devices.assign(devices, devices + numDevices);
  g->selfSendRecv = (flags & RM_GATHER_FORCE_COMM) != 0;
  if (numDevices > 1 || g->selfSendRecv) {
    DeviceGuard guard;  // ncclCommInitAll switches devices
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
int rm_gather_create(const int *devices, int numDevices, RmGather **out) { return rm_gather_create_ex(devices, numDevices, 0u, out); }

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
  std::vector<size_t> count(n);
  for (int k = 0; k < n; k++) count[k] = (size_t)shard_rows(H, tileRows, k, n, root_relief()) * W * 4;
  return gather_elems(g, reinterpret_cast<const void *const *>(d_tiles), d_gathered, count, (size_t)rm_gather_slot_rows(H, tileRows, n) * W * 4,
                      sizeof(float), ncclFloat, root, streams);
}

int rm_gather_tiles_rgba8(RmGather *g, const uint8_t *const *d_tiles8, uint8_t *d_gathered8, int W, int H, int tileRows, int root,
                          void *const *streams) {
  if (!g || !d_tiles8 || !d_gathered8 || W <= 0 || H <= 0 || tileRows <= 0) { set_error("bad gather arguments"); return RM_ERR_INVALID_ARGUMENT; }
  const int n = (int)g->devices.size();
  if (root < 0 || root >= n) { set_error("root out of range"); return RM_ERR_INVALID_ARGUMENT; }
  std::vector<size_t> count(n);
  for (int k = 0; k < n; k++) count[k] = (size_t)shard_rows(H, tileRows, k, n, root_relief()) * W * 4;  // bytes: 4 per pixel
  return gather_elems(g, reinterpret_cast<const void *const *>(d_tiles8), d_gathered8, count, (size_t)rm_gather_slot_rows(H, tileRows, n) * W * 4,
                      1, ncclUint8, root, streams);
}

}  // extern "C"
