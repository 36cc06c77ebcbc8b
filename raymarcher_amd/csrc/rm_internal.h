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
enum { kWsPost = 2, kWsTileOrder = 3, kWsWavefront = 4, kWsLightSplit = 5 };
int stream_workspace(int tag, hipStream_t stream, size_t need, void **out);
#endif

// Baseline JPEG → RGBA8, top row first (rm_jpeg.cpp).
int jpeg_decode(const std::vector<uint8_t> &file, std::vector<uint8_t> &rgba, int &W, int &H);
// First frame of a GIF → RGBA8, top row first (rm_gif.cpp).
int gif_decode(const std::vector<uint8_t> &file, std::vector<uint8_t> &rgba, int &W, int &H);

// ---- the row-tile partition of a frame over the shards of a multi-GPU render ------------------------------------------------
// A frame of H rows is cut into tiles of tileRows rows.  relief = 0 (the classic deal): tile t belongs to shard t mod N.
// relief = K >= 2 ("root relief"): the deal runs in cycles of N·K − 1 tiles — K − 1 full rounds over shards 0 … N−1, then one
// round that leaves shard 0 out — so shard 0 (the gather's root, which also receives N − 1 slots and de-interleaves the whole
// frame every frame) owns (K − 1)/K of a peer's tiles.  A shard's tiles are packed in frame order either way.
// The process-wide setting (rm_set_root_relief; every rank of a job must use the same) is read by every entry point that deals
// tiles; the functions below take it as a parameter so that kernels receive it by value.
int root_relief();

// global tile → (shard, the shard's local tile ordinal)
__host__ __device__ inline void tile_owner(int t, int N, int K, int &shard, int &local) {
  if (K < 2 || N < 2) { shard = t % N; local = t / N; return; }
  const int L = N * K - 1, q = t / L, c = t % L;
  if (c < N * (K - 1)) { shard = c % N; local = q * (shard == 0 ? K - 1 : K) + c / N; }
  else { shard = c - N * (K - 1) + 1; local = q * K + (K - 1); }
}
// a shard's local tile ordinal → global tile (increasing in j)
__host__ __device__ inline int tile_of(int shard, int j, int N, int K) {
  if (K < 2 || N < 2) return j * N + shard;
  const int L = N * K - 1, m = (shard == 0) ? K - 1 : K, q = j / m, i = j % m;
  return q * L + ((i < K - 1) ? i * N + shard : N * (K - 1) + shard - 1);
}
// Rows owned by `shard`.
__host__ __device__ inline int shard_rows(int H, int tileRows, int shard, int numShards, int relief = 0) {
  const int tiles = (H + tileRows - 1) / tileRows;
  int owned;
  if (relief < 2 || numShards < 2) {
    if (shard >= tiles) return 0;
    owned = (tiles - shard + numShards - 1) / numShards;
  } else {
    const int N = numShards, K = relief, L = N * K - 1, m = (shard == 0) ? K - 1 : K, rem = tiles % L;
    const int full = rem < N * (K - 1) ? rem : N * (K - 1);  // positions of the remainder that lie in the full rounds
    owned = (tiles / L) * m + (full > shard ? (full - shard + N - 1) / N : 0) + ((shard >= 1 && rem > N * (K - 1) + shard - 1) ? 1 : 0);
  }
  if (owned <= 0) return 0;
  int rows = owned * tileRows;
  const int lastRows = H - (tiles - 1) * tileRows;  // rows of the (possibly partial) last tile
  int s, l;
  tile_owner(tiles - 1, numShards, relief, s, l);
  if (s == shard) rows -= tileRows - lastRows;
  return rows;
}
// rows of the largest shard (the size of one gather slot): shard 0 without relief, shard 1 with it
__host__ __device__ inline int max_shard_rows(int H, int tileRows, int numShards, int relief = 0) {
  const int a = shard_rows(H, tileRows, 0, numShards, relief);
  const int b = numShards > 1 ? shard_rows(H, tileRows, 1, numShards, relief) : 0;
  return a > b ? a : b;
}

}  // namespace rm
