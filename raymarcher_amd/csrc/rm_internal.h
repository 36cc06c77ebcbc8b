// rm_internal.h — helpers shared by the launcher (rm_kernels.hip) and the host side (rm_host.cpp).
#pragma once
#include <initializer_list>
#include <string>
#include <utility>
#include <vector>
#include <cstdint>

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#else
#define __host__
#define __device__
#endif

namespace rm {

// Records the text returned by rm_last_error() for the calling thread.
void set_error(const std::string &msg);

// True if `p` is memory a kernel may dereference (device, managed or pinned host) according to the HIP runtime.
// Every "device pointer" argument of the ABI is checked with it before a launch: a kernel that touches plain host
// memory faults the GPU.
bool device_accessible(const void *p);
// RM_ERR_INVALID_ARGUMENT (+ rm_last_error text) unless every non-null pointer of the list is device-accessible.
int require_device_pointers(std::initializer_list<std::pair<const char *, const void *>> ptrs);

#ifdef __HIPCC__
// Grow-only device scratch memory owned by the library, one buffer per (current device, stream, user tag): calls on
// different streams of one device may run concurrently on the GPU and therefore never share scratch.  Growing
// synchronises `stream` (nothing else uses the old buffer) and reallocates.  Returns an rm_status.
enum { kWsPost = 2, kWsTileOrder = 3, kWsWavefront = 4 };
int stream_workspace(int tag, hipStream_t stream, size_t need, void **out);
#endif

// Baseline JPEG → RGBA8, top row first (rm_jpeg.cpp).
int jpeg_decode(const std::vector<uint8_t> &file, std::vector<uint8_t> &rgba, int &W, int &H);
// First frame of a GIF → RGBA8, top row first (rm_gif.cpp).
int gif_decode(const std::vector<uint8_t> &file, std::vector<uint8_t> &rgba, int &W, int &H);

// Rows owned by `shard` when an H-row frame is cut into tiles of tileRows rows dealt round-robin.
__host__ __device__ inline int shard_rows(int H, int tileRows, int shard, int numShards) {
  const int tiles = (H + tileRows - 1) / tileRows;
  if (shard >= tiles) return 0;
  const int owned = (tiles - shard + numShards - 1) / numShards;
  int rows = owned * tileRows;
  const int lastRows = H - (tiles - 1) * tileRows;  // rows of the (possibly partial) last tile
  if ((tiles - 1) % numShards == shard) rows -= tileRows - lastRows;
  return rows;
}

}  // namespace rm
