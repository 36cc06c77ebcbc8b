// mgpu_host.cpp — one process drives every visible MI355X of a node through the C ABI (no Python, no torch): the
// single-process counterpart of `bench.py --gpus N`.  Every device renders its interleaved row tiles of each frame
// (rm_render_tiles) with `depth` frames in flight on `depth` streams per device, the tiles travel to device 0 over RCCL
// (rm_gather_tiles: grouped ncclSend / ncclRecv, float4; or --rgba8: rm_tiles_to_rgba8 + rm_gather_tiles_rgba8, 4 B/pixel),
// device 0 de-interleaves.  Prints Mpixels/s of whole frames assembled on device 0.
//
//   hipcc -std=c++17 -O2 -I include scripts/mgpu_host.cpp -o mgpu_host -L raymarcher_amd/lib -lraymarcher_amd -Wl,-rpath,$PWD/raymarcher_amd/lib
//   ./mgpu_host tests/golden/scenes/simple/unit_mandelbulb.json [--size 3840 2160] [--frames 200] [--gpus N] [--rgba8]
//               [--iters 12] [--levels 5 --bounces 2 --reflection] [--depth 3] [--force-comm]
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "raymarcher_amd.h"

#define CHECK(call)                                                                                   \
  do {                                                                                                \
    int st_ = (call);                                                                                 \
    if (st_ != RM_OK) { std::fprintf(stderr, "%s: %s (%s)\n", #call, rm_status_string(st_), rm_last_error()); return 1; } \
  } while (0)
#define HIP(call)                                                                                     \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: mgpu_host scenefile.json [options]\n"); return 2; }
  int W = 3840, H = 2160, frames = 200, gpus = 0, depth = 3, T = 8;
  bool rgba8 = false, forceComm = false;
  RmSettings s;
  rm_settings_default(&s);
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i];
    auto num = [&](int k) { return std::atoi(argv[i + k]); };
    if (a == "--size" && i + 2 < argc) { W = num(1); H = num(2); i += 2; }
    else if (a == "--frames" && i + 1 < argc) { frames = num(1); i++; }
    else if (a == "--gpus" && i + 1 < argc) { gpus = num(1); i++; }
    else if (a == "--depth" && i + 1 < argc) { depth = num(1); i++; }
    else if (a == "--iters" && i + 1 < argc) { s.fractalIters = num(1); i++; }
    else if (a == "--levels" && i + 1 < argc) { s.mengerLevels = num(1); i++; }
    else if (a == "--bounces" && i + 1 < argc) { s.numReflection = num(1); i++; }
    else if (a == "--reflection") s.enableReflection = 1;
    else if (a == "--soft") s.enableSoftShadow = 1;
    else if (a == "--ao") s.enableAmbientOcclusion = 1;
    else if (a == "--rgba8") rgba8 = true;
    else if (a == "--force-comm") forceComm = true;
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  int have = rm_device_count();
  if (have < 1) { std::fprintf(stderr, "no HIP device\n"); return 1; }
  const int N = (gpus > 0 && gpus <= have) ? gpus : have;
  if (depth < 1 || depth > 8) depth = 3;

  RmScene *sc = nullptr;
  CHECK(rm_scene_load(argv[1], &sc));
  RmCameraData cd; RmCamera cam; RmGlobals g;
  CHECK(rm_scene_camera_data(sc, &cd));
  CHECK(rm_camera_build(&cd, W, H, 0.1f, 100.0f, nullptr, nullptr, &cam));
  CHECK(rm_scene_globals(sc, nullptr, &g));
  const RmObject *objs = rm_scene_objects(sc);
  const RmLight *lights = rm_scene_lights(sc);
  const int no = rm_scene_num_objects(sc), nl = rm_scene_num_lights(sc);

  std::vector<int> devs(N);
  for (int k = 0; k < N; k++) devs[k] = k;
  RmGather *ga = nullptr;
  CHECK(rm_gather_create_ex(devs.data(), N, forceComm ? RM_GATHER_FORCE_COMM : 0u, &ga));
  const int slot = rm_gather_slot_rows(H, T, N);
  const size_t px = rgba8 ? 4 : 16;
  // per device and frame slot: a stream, the float4 tiles, (rgba8) their 8-bit copy; on device 0 per frame slot: gather buffer + frame
  std::vector<std::vector<hipStream_t>> stream(N, std::vector<hipStream_t>(depth));
  std::vector<std::vector<float *>> tiles(N, std::vector<float *>(depth));
  std::vector<std::vector<uint8_t *>> tiles8(N, std::vector<uint8_t *>(depth, nullptr));
  std::vector<void *> gathered(depth), frame(depth);
  for (int k = 0; k < N; k++) {
    CHECK(rm_set_device(devs[k]));
    for (int f = 0; f < depth; f++) {
      HIP(hipStreamCreateWithFlags(&stream[k][f], hipStreamNonBlocking));
      HIP(hipMalloc(reinterpret_cast<void **>(&tiles[k][f]), size_t(slot) * W * 16));
      if (rgba8) HIP(hipMalloc(reinterpret_cast<void **>(&tiles8[k][f]), size_t(slot) * W * 4));
    }
  }
  CHECK(rm_set_device(devs[0]));
  for (int f = 0; f < depth; f++) {
    HIP(hipMalloc(&gathered[f], size_t(N) * slot * W * px));
    HIP(hipMalloc(&frame[f], size_t(W) * H * px));
  }
  auto submit = [&](int i) -> int {
    const int f = i % depth;
    std::vector<const float *> src(N);
    std::vector<const uint8_t *> src8(N);
    std::vector<void *> st(N);
    for (int k = 0; k < N; k++) {
      CHECK(rm_set_device(devs[k]));
      CHECK(rm_render_tiles(&cam, objs, no, lights, nl, &g, &s, W, H, T, k, N, tiles[k][f], nullptr, stream[k][f]));
      if (rgba8) CHECK(rm_tiles_to_rgba8(tiles[k][f], tiles8[k][f], W, rm_shard_rows(H, T, k, N), stream[k][f]));
      src[k] = tiles[k][f]; src8[k] = tiles8[k][f]; st[k] = stream[k][f];
    }
    CHECK(rm_set_device(devs[0]));
    if (rgba8) {
      CHECK(rm_gather_tiles_rgba8(ga, src8.data(), static_cast<uint8_t *>(gathered[f]), W, H, T, 0, st.data()));
      CHECK(rm_deinterleave_rgba8(static_cast<uint8_t *>(gathered[f]), static_cast<uint8_t *>(frame[f]), W, H, T, N, slot, 1, stream[0][f]));
    } else {
      CHECK(rm_gather_tiles(ga, src.data(), static_cast<float *>(gathered[f]), W, H, T, 0, st.data()));
      CHECK(rm_deinterleave(static_cast<float *>(gathered[f]), static_cast<float *>(frame[f]), W, H, T, N, slot, stream[0][f]));
    }
    return 0;
  };
  auto drain = [&]() -> int {
    for (int k = 0; k < N; k++) {
      CHECK(rm_set_device(devs[k]));
      for (int f = 0; f < depth; f++) HIP(hipStreamSynchronize(stream[k][f]));
    }
    return 0;
  };
  const int warm = frames / 10 + depth;
  for (int i = 0; i < warm; i++) if (submit(i)) return 1;
  if (drain()) return 1;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < frames; i++) if (submit(i)) return 1;
  if (drain()) return 1;
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("{\"metric\": \"Mpixels/s\", \"value\": %.2f, \"n_gpus\": %d, \"frames\": %d, \"ms_per_frame\": %.4f, \"size\": [%d, %d], "
              "\"gather\": \"%s\", \"frames_in_flight\": %d, \"host\": \"one process, rm_gather_tiles over RCCL\"}\n",
              double(W) * H * frames / secs / 1e6, N, frames, secs / frames * 1e3, W, H, rgba8 ? "rgba8" : "float4", depth);
  rm_gather_destroy(ga);
  rm_scene_free(sc);
  return 0;
}
