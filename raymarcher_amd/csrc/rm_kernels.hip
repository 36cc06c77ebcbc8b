// rm_kernels.hip — HIP kernels and the C-ABI launcher of the per-pixel raymarch (gfx950 only).
//
// Replaces Realtime::rayMarch() of the reference (src/realtimerender.cpp:53-87): instead of uploading
// ~600 uniforms by name and drawing a full-screen quad through resources/raymarch.{vert,frag}, the
// launcher copies one constant SceneBlock to the device and launches one lane per pixel.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "rm_device.hip.h"
#include "rm_wavefront.hip.h"
#include "rm_internal.h"

namespace rm {

bool device_accessible(const void *p) {
  hipPointerAttribute_t a{};
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // unknown (plain host) pointer: clear the sticky error
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeHost;  // Host = pinned
}
int require_device_pointers(std::initializer_list<std::pair<const char *, const void *>> ptrs) {
  for (const auto &p : ptrs)
    if (p.second && !device_accessible(p.second)) {
      set_error(std::string(p.first) + " is not device-accessible memory");
      return RM_ERR_INVALID_ARGUMENT;
    }
  return RM_OK;
}

// Block = 4 waves side by side, each wave an 8×8 pixel tile → the block covers 32×8 pixels.
#ifndef RM_TILE_W
#define RM_TILE_W 8   // pixels per wave tile, horizontally (RM_TILE_W × RM_TILE_H = 64)
#endif
static_assert(RM_TILE_W == 4 || RM_TILE_W == 8 || RM_TILE_W == 16, "tile width: 4, 8 or 16 pixels");

// COUNT: 0 production, 1 reference-work counters, 2 executed-work counters (rm_device.hip.h), 3 production code plus clock
// stamps: every wave adds its (s_memtime, s_memrealtime) spans to counters[3], counters[4] — shader cycles and 100 MHz
// ticks — from which rm_render_clocked derives the clock the chip held under this kernel's own load.  The stamps go to a
// buffer of their own and no output value depends on them.
// Register budget = waves per SIMD (second launch bound), MEASURED per kernel class (profiles/r02_m_occupancy.md): the
// compiler's own choice for these kernels is 121-219 VGPRs (2-4 waves); bounding them tighter spills 29-90 registers, but
// the spills land outside the march / iteration loops (shading prologues and epilogues) and the extra resident waves hide the
// serial latency of an evaluation (scalar loads per object, dependent transcendental chains): at 3840x2160 a 5-object Phong
// scene gains 26 %, bump + reflection 28 %, textured / sky-box scenes 83-90 %, the 8K Menger frame 18 %, the terrain and
// sea frames 5-7 %, the headline bulb frame 2.3 % (5 waves; its hot loops stay spill-free).  Re-tuned on the final code: 6 / 5 / 6 / 6.  Frames too small to fill the
// chip (256x256, 1080p tails) lose 1-2 %.  -DRM_*_WAVES=n overrides, for the experiment script scripts/gpu_variants.sh.
#ifndef RM_GENERIC_WAVES
#define RM_GENERIC_WAVES 6
#endif
#ifndef RM_BULB_WAVES
#define RM_BULB_WAVES 5
#endif
// the instantiations without main's secondary rays (SEC = false: no reflection / refraction anywhere in the frame) need far fewer
// registers — the bulb kernel 90 without a single spill — and take their own budgets (profiles/r04_d_secondary_rays.md)
#ifndef RM_BULB_NOSEC_WAVES
#define RM_BULB_NOSEC_WAVES 6
#endif
#ifndef RM_GENERIC_NOSEC_WAVES
#define RM_GENERIC_NOSEC_WAVES 6
#endif
#ifndef RM_ENV_NOSEC_WAVES
#define RM_ENV_NOSEC_WAVES 6
#endif
#ifndef RM_TEX_NOSEC_WAVES
#define RM_TEX_NOSEC_WAVES 6
#endif
#ifndef RM_ENV_WAVES
#define RM_ENV_WAVES 6
#endif
#ifndef RM_TEX_WAVES
#define RM_TEX_WAVES 6
#endif
// SPLIT ("light split", launch_render): 1 = the launch of a frame whose heaviest tiles are rendered one light per workgroup — a 1-D
// grid: workgroups 0 … splitTiles·numLights − 1 are those tiles' partial workgroups (tile = tileOrder[b / numLights], light b mod
// numLights: primary march, surface, THAT light's shadow march, its result to splitStore), the last of which to arrive finishes
// the tile's pixels from the stored results (shadePixel's mode 2: no march); the rest render the other tiles whole.
template <bool BULB, int COUNT, bool ENV, bool TEX, bool SEC = true, int SPLIT = 0>
__global__ __launch_bounds__(256, (TEX ? (SEC ? RM_TEX_WAVES : RM_TEX_NOSEC_WAVES) : (ENV ? (SEC ? RM_ENV_WAVES : RM_ENV_NOSEC_WAVES) : (BULB ? (SEC ? RM_BULB_WAVES : RM_BULB_NOSEC_WAVES) : (SEC ? RM_GENERIC_WAVES : RM_GENERIC_NOSEC_WAVES))))) void render_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                      int nRows, float4 *__restrict__ out,
                                                      float4 *__restrict__ bright,
                                                      unsigned long long *__restrict__ counters) {
  constexpr int CM = (COUNT == 3) ? 0 : COUNT;  // counting mode of the device code
  unsigned long long t0 = 0, r0 = 0;
  if (COUNT == 3) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  // LDS copy of the object table for per-lane (divergent) material lookups; the single-bulb class reads one entry.
  __shared__ RmObject s_objs[BULB ? 1 : RM_MAX_OBJECTS];
  {
    const int nd = sb->numObjects * (int)(sizeof(RmObject) / 4);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sb->objs);
    uint32_t *dst = reinterpret_cast<uint32_t *>(s_objs);
    for (int i = threadIdx.x; i < nd; i += blockDim.x) dst[i] = src[i];
  }
  // the launches that read samplers build the byte→unorm table (wave-uniform condition; ends with a barrier)
  if (TEX || (ENV && (sb->s.features & (RM_FEAT_NIGHTSKY_BACKGROUND | RM_FEAT_SEA)))) initUnormTable();
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // tile order: the workgroups of a grid start in blockIdx order; sb->tileOrder (if any) says which tile each one renders
  int tile = blockIdx.y * gridDim.x + blockIdx.x;
  const int32_t *order = sb->tileOrder;
  const int tsh = sb->tileShift, tw = 1 << tsh;  // wave-uniform (scalar): the tile is tw pixels wide, 64 / tw tall
  int tilesX = (int)gridDim.x;
  LightSplit split{-1, nullptr, 0};
  if (SPLIT) {  // one-wave workgroups, 1-D grid (launch_render)
    const int nl = sb->numLights, K = sb->splitTiles, b = (int)blockIdx.x;
    tilesX = (W + tw - 1) / tw;
    int h = b;  // position of the tile in tileOrder
    if (SPLIT == 1) {
      if (b < K * nl) { h = b / nl; split.part = b - h * nl; }
      else h = K + (b - K * nl);
    }
    tile = order[h];
    if (h < K) {  // K arrival counters, then per tile 64 pixels × (nl shadow results + the primary march's)
      split.tileIndex = h;
      split.slot = sb->splitStore + (((size_t)K + 63) & ~(size_t)63) + ((size_t)h * 64 + lane) * (size_t)(2 * nl + 6);
    }
  } else if (order && sb->tileCount == (int)(gridDim.x * gridDim.y)) {
    tile = order[tile];
  }
  const int tbx = tile % tilesX, tby = tile / tilesX;
  // the wave's start stamp waits in LDS (not in two scalar registers across the whole kernel — the register budget is tight)
  __shared__ unsigned long long s_c0[4];
  if (!SPLIT && sb->tileCost && lane == 0) s_c0[wave] = __builtin_amdgcn_s_memtime();
  const int x = (tbx * (blockDim.x >> 6) + wave) * tw + (lane & (tw - 1));
  const int r = tby * (64 >> tsh) + (lane >> tsh);
  if (x >= W || r >= nRows) return;
  const int y = map.frameRow(r);
  V4 col, br;
  Counters cnt{0, 0, 0, 0, 0, 0};
  bool hit;
  shadePixel<BULB, CM, ENV, TEX, SEC, SPLIT>(sb, s_objs, x, y, W, H, col, br, cnt, hit, split);
  if (SPLIT == 1 && split.part >= 0) {
    // A partial workgroup: its results are in memory.  The LAST of the tile's numLights workgroups to get here finishes the tile —
    // surface point, AO and the light sum from the stored results, no march (shadePixel in mode 2) — the others are done.  Release /
    // acquire at device scope around a counter per tile: the stores of the others are visible to the one that reads old == nl − 1.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this workgroup's records are written back before its arrival counts
    uint32_t *arrived = reinterpret_cast<uint32_t *>(sb->splitStore) + split.tileIndex;  // the counters precede the records (launch_render zeroes them)
    const int leader = __builtin_ctzll(__ballot(1));
    uint32_t old = 0;
    if ((int)__lane_id() == leader) old = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = __shfl(old, leader);
    if ((int)old != sb->numLights - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // only the finisher invalidates its view before it reads the others' records
    split.part = -1;
    shadePixel<BULB, CM, ENV, TEX, SEC, 2>(sb, s_objs, x, y, W, H, col, br, cnt, hit, split);
  }
  const size_t o = (size_t)r * W + x;
  out[o] = make_float4(col.x, col.y, col.z, col.w);
  if (bright) bright[o] = make_float4(br.x, br.y, br.z, br.w);
  if (CM) {
    atomicAdd(&counters[0], cnt.evals);
    atomicAdd(&counters[1], cnt.iters);
    if (hit) atomicAdd(&counters[2], 1ull);
    if (cnt.shades) atomicAdd(&counters[6], cnt.shades);
    if (cnt.fbm9) atomicAdd(&counters[7], cnt.fbm9);
    if (cnt.fbmd8) atomicAdd(&counters[8], cnt.fbmd8);
    if (cnt.shapes) atomicAdd(&counters[9], cnt.shapes);
  }
  if (!SPLIT && sb->tileCost && sb->tileCount == (int)(gridDim.x * gridDim.y)) {  // wave-uniform
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    int t2 = blockIdx.y * gridDim.x + blockIdx.x;  // recomputed rather than kept live
    if (sb->tileOrder) t2 = sb->tileOrder[t2];
    if ((int)__lane_id() == __builtin_ctzll(__ballot(1))) atomicAdd(&sb->tileCost[t2], (uint32_t)((c1 - s_c0[wave]) >> 6));
  }
  if (COUNT == 3) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((int)__lane_id() == __builtin_ctzll(__ballot(1))) {  // first live lane of the wave
      atomicAdd(&counters[3], t1 - t0);
      atomicAdd(&counters[4], r1 - r0);
      // optional per-wave life span (100 MHz ticks) for occupancy timelines: counters[5] holds a device pointer or 0
      unsigned long long *spans = reinterpret_cast<unsigned long long *>(counters[5]);
      if (spans) {
        int t3 = blockIdx.y * gridDim.x + blockIdx.x;
        if (sb->tileOrder && sb->tileCount == (int)(gridDim.x * gridDim.y)) t3 = sb->tileOrder[t3];
        const size_t w = (size_t)t3 * (blockDim.x >> 6) + wave;
        spans[2 * w] = r0;
        spans[2 * w + 1] = r1;
      }
    }
  }
}

// sdMengerSponge's uniform prologue (frag:1052-1053), once per launch, with the device's own rm_math (bit-exactness with the
// oracle needs the contract's sin / cos / smoothstep, which only the device and the oracle implement).
__global__ void scene_prep_kernel(SceneBlock *sb) {
  sb->mengerAni = smoothstep_(-0.2f, 0.2f, -cos_(0.5f * sb->g.iTime));
  sb->mengerOff = 1.5f * sin_(0.01f * sb->g.iTime);
}

// ---- tile order ------------------------------------------------------------------------------------------
// The cost of a tile (one workgroup of render_kernel) spans three orders of magnitude: background tiles end after one
// evaluation, while an interior tile may hold ONE ray that creeps through a crevice for all 256 steps without ever
// converging — a sequential chain of ≈0.5 M instructions, ≈1 ms on its own.  Workgroups start in blockIdx order; with
// tiles in raster order such stragglers start at random times, the last of them late, and the kernel ends in a tail of a
// few lonely waves (profiles/r02_e_wave_timeline.md: the last 15 % of the kernel's life had ≤ 3 waves resident).
// Starting the heavy tiles first removes the tail: 3.11 → 2.43 ms on the 4K bulb frame with measured costs
// (profiles/r02_f_tile_order.md).  Nothing about a pixel changes, only when its tile starts.
// Where the straggler pixels are cannot be told from a sparse pre-pass (tried: 8 sample rays per tile, 0.24 ms, no gain);
// what does know is the previous frame.  render_kernel adds every wave's shader-cycle span to tileCost[tile]; the next
// frame of the same size on the same stream starts its tiles in descending order of those costs (a renderer's consecutive
// frames are nearly the same picture; a frame with no history, or after a change of size, runs in raster order).
// The order is a two-launch bucket sort by log2(cost): tile_hist_kernel counts, tile_scatter_kernel places.
constexpr int kOrderBuckets = 16;
RM_DEV int orderBucket(uint32_t cost) {  // heaviest = bucket 0; costs are shader cycles / 64, i.e. ≈2^5 … 2^17
  const int lg = 31 - __builtin_clz(cost | 1u);
  const int b = 17 - lg;
  return b < 0 ? 0 : (b >= kOrderBuckets ? kOrderBuckets - 1 : b);
}
__global__ __launch_bounds__(256) void tile_hist_kernel(const uint32_t *__restrict__ cost, int n, uint32_t *__restrict__ hist) {
  __shared__ uint32_t s_cnt[kOrderBuckets];
  if (threadIdx.x < kOrderBuckets) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = i < n ? orderBucket(cost[i]) : -1;
  for (int k = 0; k < kOrderBuckets; k++) {  // wave-aggregated: one LDS atomic per wave and bucket present
    const unsigned long long m = __ballot(b == k);
    if (m && (int)__lane_id() == __builtin_ctzll(m)) atomicAdd(&s_cnt[k], (uint32_t)__builtin_popcountll(m));
  }
  __syncthreads();
  if (threadIdx.x < kOrderBuckets && s_cnt[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_cnt[threadIdx.x]);
}
// hist[0..15] = bucket sizes (from tile_hist_kernel), hist[16..31] = cursors (zero on entry).  Consumes (clears) cost[] unless
// `keep` (the last sort of a settled picture: the costs stay as the stale costs of whatever picture comes next).
__global__ __launch_bounds__(256) void tile_scatter_kernel(uint32_t *__restrict__ cost, int n, uint32_t *__restrict__ hist,
                                                            int32_t *__restrict__ order, int keep) {
  __shared__ uint32_t s_cnt[kOrderBuckets], s_base[kOrderBuckets];
  if (threadIdx.x < kOrderBuckets) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = i < n ? orderBucket(cost[i]) : -1;
  uint32_t rank = 0;  // position inside the block's share of bucket b
  for (int k = 0; k < kOrderBuckets; k++) {
    const unsigned long long m = __ballot(b == k);
    if (!m) continue;
    uint32_t waveBase = 0;
    if ((int)__lane_id() == __builtin_ctzll(m)) waveBase = atomicAdd(&s_cnt[k], (uint32_t)__builtin_popcountll(m));
    waveBase = __shfl(waveBase, __builtin_ctzll(m));
    if (b == k) rank = waveBase + (uint32_t)__builtin_popcountll(m & ((1ull << __lane_id()) - 1ull));
  }
  __syncthreads();
  if (threadIdx.x < kOrderBuckets) {
    uint32_t start = 0;  // exclusive scan of the bucket sizes, heaviest bucket first
    for (int k = 0; k < (int)threadIdx.x; k++) start += hist[k];
    s_base[threadIdx.x] = start + (s_cnt[threadIdx.x] ? atomicAdd(&hist[kOrderBuckets + threadIdx.x], s_cnt[threadIdx.x]) : 0u);
  }
  __syncthreads();
  if (i < n) {
    order[s_base[b] + rank] = i;
    if (!keep) cost[i] = 0u;  // the next frame accumulates afresh
  }
}

// A frame WITHOUT usable history (the first of its size on a stream, or any frame whose scene or camera differs from the one that
// recorded the costs): stand-in costs from geometry alone.  One thread per tile: the tile centre's primary ray against every
// object's world-space bounding ball (SceneBlock::objBall) — closest approach inside [0.6·R, R + the tile's footprint] is a
// silhouette candidate (the rays that graze an object march longest), inside 0.6·R an interior tile, anything else background;
// the bucket sort above then starts rings first, interiors next, background last, raster order within a class.  Measured on cold
// 4K frames (scripts/cold_order_probe.py, profiles/r04_k_geometric_order.md): bulb 2.85 → 2.26 ms (measured costs: 1.99),
// directional_light_2.json 1.89 → 1.77-1.81, reflections_complex.json 8.70 → 8.39-8.44.  Same pixels.
__global__ __launch_bounds__(256) void tile_geom_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H, int nRows, int tilesX,
                                                        int tileW, int tileH, int n, const uint32_t *__restrict__ stale,
                                                        uint32_t *__restrict__ cost, int combine, int ringLog2, int dilate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int x = (i % tilesX) * tileW + tileW / 2, r = (i / tilesX) * tileH + tileH / 2;
  x = x < W ? x : W - 1;
  r = r < nRows ? r : nRows - 1;
  V3 ro, rd;
  primaryRay(sb, x, map.frameRow(r), W, H, ro, rd);
  // angle of one tile seen from the eye ≈ the distance between neighbouring tile centres' directions
  V3 ro2, rd2;
  const int span = tileW > tileH ? tileW : tileH;  // the tile's larger side, in pixels
  primaryRay(sb, x < W - span ? x + span : x - span, map.frameRow(r), W, H, ro2, rd2);
  const float foot = len(sub(rd2, rd));
  int cls = 0;
  const int no = sb->numObjects;
  for (int k = 0; k < no; k++) {
    const V3 v = v3(sb->objBall[k][0] - ro.x, sb->objBall[k][1] - ro.y, sb->objBall[k][2] - ro.z);
    const float R = sb->objBall[k][3], tca = dot(v, rd);
    const float q2 = dot(v, v) - tca * tca, hi = fma(foot, tca, R), lo = 0.6f * R;
    if (tca > 0.0f && q2 <= hi * hi) cls = (q2 >= lo * lo) ? 2 : (cls > 1 ? cls : 1);
  }
  const uint32_t gv = cls == 2 ? (1u << ringLog2) : (cls == 1 ? (1u << 11) : (1u << 4));
  // combine: a frame of the same size rendered a DIFFERENT picture before (a moving camera) — its measured costs are stale but near;
  // the heavier of the two estimates decides.  dilate: the stale estimate of a tile is the heaviest within that many tiles of it (a
  // silhouette that the camera's motion shifted by a few tiles is still where its heavy tiles are looked for)
  uint32_t sv = 0u;
  if (combine) {
    const int tx = i % tilesX, ty = i / tilesX, tilesY = (n + tilesX - 1) / tilesX;
    for (int dy = -dilate; dy <= dilate; dy++)
      for (int dx = -dilate; dx <= dilate; dx++) {
        const int nx = tx + dx, ny = ty + dy;
        if (nx < 0 || ny < 0 || nx >= tilesX || ny >= tilesY || ny * tilesX + nx >= n) continue;
        const uint32_t c = stale[ny * tilesX + nx];
        sv = c > sv ? c : sv;
      }
  }
  cost[i] = sv > gv ? sv : gv;
}

__global__ void probe_math_kernel(int fn, const float *x, const float *y, const float *z, float *out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = x[i], b = y ? y[i] : 0.0f, c = z ? z[i] : 0.0f, r = 0.0f;
  switch (fn) {
    case RM_FN_SIN: r = sin_(a); break;
    case RM_FN_COS: r = cos_(a); break;
    case RM_FN_ACOS: r = acos_(a); break;
    case RM_FN_ATAN2: r = atan2_(a, b); break;
    case RM_FN_LOG2: r = log2_(a); break;
    case RM_FN_EXP2: r = exp2_(a); break;
    case RM_FN_POW: r = pow_(a, b); break;
    case RM_FN_SQRT: r = sqrt_(a); break;
    case RM_FN_DIV: r = a / b; break;
    case RM_FN_PNOISE3: r = pnoise(v3(a, b, c)); break;
    case RM_FN_ASIN: r = asin_(a); break;
    case RM_FN_Q16: r = __half2float(__float2half_rn(a)); break;
    case RM_FN_SQRT_FAST: r = sqrt_fast_(a); break;
    case RM_FN_DIVR: r = divr_(a, b); break;
    case RM_FN_RCP: r = rcp_(a); break;
    case RM_FN_SMOOTHSTEP: r = smoothstep_(a, b, c); break;
    case RM_FN_MIN: r = min_(a, b); break;
    case RM_FN_MAX: r = max_(a, b); break;
    case RM_FN_FRACT: r = fract_(a); break;
    case RM_FN_MEDIAN_ABS: r = __builtin_amdgcn_fmed3f(fabs_(a), fabs_(b), fabs_(c)); break;
  }
  out[i] = r;
}

__global__ void probe_sdscene_kernel(const SceneBlock *__restrict__ sb, const float *pts, float *out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Counters cnt{0, 0, 0, 0, 0, 0};
  SceneMin m = sdScene<false, 0>(sb, v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), cnt);
  out[4 * i] = m.d;
  out[4 * i + 1] = (float)m.idx;
  out[4 * i + 2] = m.trap.y;
  out[4 * i + 3] = m.trap.z;
}

// clamp → ×255 → round-half-up, vertical flip (src/realtime.cpp:337-338 + GL's RGBA8 conversion).
__global__ void to_rgba8_kernel(const float4 *__restrict__ in, uchar4 *__restrict__ out, int W, int H) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  float4 c = in[(size_t)y * W + x];
  auto q = [](float v) { v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); return (unsigned char)(v * 255.0f + 0.5f); };
  out[(size_t)(H - 1 - y) * W + x] = make_uchar4(q(c.x), q(c.y), q(c.z), q(c.w));
}

// packed tiles → RGBA8, rows as they are (the multi-GPU shard's share of to_rgba8_kernel's conversion)
__global__ void tiles_to_rgba8_kernel(const float4 *__restrict__ in, uchar4 *__restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 c = in[i];
  auto q = [](float v) { v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); return (unsigned char)(v * 255.0f + 0.5f); };
  out[i] = make_uchar4(q(c.x), q(c.y), q(c.z), q(c.w));
}
// gathered RGBA8 slots → frame rows (flip: row 0 of the output is the top of the image, as rm_frame_to_rgba8 writes it)
__global__ void deinterleave_rgba8_kernel(const uchar4 *__restrict__ in, uchar4 *__restrict__ out, int W, int H, int tileRows,
                                          int numShards, int strideRows, int flip, int relief) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;  // y = frame row (0 = bottom)
  if (x >= W) return;
  int shard, localTile;
  tile_owner(y / tileRows, numShards, relief, shard, localTile);
  int before = shard * strideRows;
  if (strideRows == 0)
    for (int s = 0; s < shard; s++) before += shard_rows(H, tileRows, s, numShards, relief);
  int local = localTile * tileRows + (y % tileRows);
  out[(size_t)(flip ? H - 1 - y : y) * W + x] = in[(size_t)(before + local) * W + x];
}
// gathered[shard-major packed rows] → frame rows
__global__ void deinterleave_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, int W, int H, int tileRows,
                                    int numShards, int strideRows, int relief) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;  // y = frame row
  if (x >= W) return;
  int shard, localTile;
  tile_owner(y / tileRows, numShards, relief, shard, localTile);
  // rows owned by shards < shard, plus this shard's rows before frame row y
  int before = shard * strideRows;
  if (strideRows == 0)
    for (int s = 0; s < shard; s++) before += shard_rows(H, tileRows, s, numShards, relief);
  int local = localTile * tileRows + (y % tileRows);
  out[(size_t)y * W + x] = in[(size_t)(before + local) * W + x];
}

// ---- launcher state -------------------------------------------------------------------------------------
// Everything is per device: a host thread driving GPU k never takes a lock that a thread driving GPU j holds, and no lock
// is held across a blocking HIP call on the launch path.  Scratch memory is per (device, stream): two calls on different
// streams of one device may overlap on the GPU, so they must not share ping-pong buffers or hit lists.
namespace {
constexpr int kSlotsInit = 8, kSlotsMax = 64;
struct Slot {
  SceneBlock *host = nullptr;  // pinned
  SceneBlock *dev = nullptr;
  hipEvent_t done = nullptr;
  bool used = false;
};
struct TileOrderState {
  int tileCount = 0, W = 0, nRows = 0, nw = 0, tileShift = 3;
  void *mem = nullptr;
  unsigned long long sceneKey = 0;  // hash of the scene + camera + row map of the frame that recorded the costs in `mem`
  int sorts = 0;                     // consecutive frames of that picture whose order came from measured costs
};
// "Tile shape": which of the two tile shapes a picture renders faster with is scene-dependent (upright objects: 4 wide × 16 tall
// tiles straddle fewer vertical silhouettes, so whole waves agree on the table walk's shortcuts more often — C2 at 1080p 0.866 →
// 0.792 ms — while reflections_complex.json loses 5 % that way; profiles/r04_m_tile_shape.txt).  So the launcher MEASURES, per
// stream and picture: frames 0-1 of a picture run 8×8 (frame 1, ordered by frame 0's costs, is timed with HIP events), frames 2-3
// run 4×16 (frame 3 timed), frames 4-7 repeat that (clocks ramp up over a process's first frames: one round would favour the later
// candidate), and from then on the shape with the smaller best time is used.  Same pixels whatever the shape.  Single-bulb class: 8×8
// always (measured: 4×16 +5 %).  RM_TILE_SHAPE / rm_debug_set_tile_shape: 0 tune, 3 always 8×8, 2 always 4×16.
struct ShapeTune {
  unsigned long long key = 0;
  int W = 0, nRows = 0;
  int frame = 0;    // frames of this picture enqueued so far
  int chosen = -1;  // the decided tile shift, -1 while measuring
  hipEvent_t ev[4][2] = {};  // [timed launch k: candidate k & 1 (0 = 8×8, 1 = 4×16), round k >> 1][start, stop]
  bool timed[4] = {false, false, false, false};
  void drop() {
    for (auto &c : ev) for (auto &e : c) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (bool &t : timed) t = false;
  }
};
// "Light split" (launch_render) helps frames that are bound by the life of their heaviest waves when those waves are shadow marches
// (C2 at 1080p: −35 %) and costs others a few per cent (redundant primary marches, cache write-backs: 4K frames +3…+9 %,
// depth_of_field.json +10 %; profiles/r04_s_light_split.md).  So it is MEASURED per stream and settled picture like the tile shape:
// settled frames 0-1 plain (frame 1 timed), 2-3 split (frame 3 timed), then the split stays only if it won by 3 %; plain while the
// timings are outstanding.  A decision is shared by the device's other streams.
struct SplitTune {
  unsigned long long key = 0;
  int W = 0, nRows = 0, tileShift = 0, div = 0;
  int frame = 0;    // settled frames of this picture enqueued so far
  int chosen = -1;  // -1 measuring, 0 plain, 1 split
  hipEvent_t ev[2][2] = {};  // [0 plain, 1 split][start, stop]
  bool timed[2] = {false, false};
  void drop() {
    for (auto &p : ev) for (auto &e : p) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    timed[0] = timed[1] = false;
  }
};
struct TimedLaunch { hipEvent_t ev[5]; int n; };  // n = 2 (one stage) or 3 (tile-order sort + render kernel)
struct DeviceState {
  std::mutex mu;                 // guards everything below; held for the host-side enqueue of ONE launch on this device
  std::vector<Slot> slots;       // ring of scene-table slots; grows (to kSlotsMax) instead of waiting for a busy slot
  size_t next = 0;
  unsigned long long *dCounters = nullptr;  // 10 words: evals, iterations, hits, clock stamps (2), span pointer, shades, fbm9, fbmd8, shapes
  std::vector<TimedLaunch> timed;           // rm_set_timing / rm_get_timing, per device
  int numCUs = 0;
  std::map<hipStream_t, TileOrderState> tileOrder;  // what the feedback costs of each stream belong to
  std::map<hipStream_t, ShapeTune> shapeTune;  // the tile-shape tuner's state per stream
  std::map<std::tuple<unsigned long long, int, int>, int> shapeChoice;  // decisions by (picture, W, rows): other streams adopt them
  std::map<hipStream_t, size_t> wfDenied;  // smallest wavefront workspace (bytes) that could not be had on a stream
  const int32_t *dbgTileOrder = nullptr;  // rm_debug_set_tile_order (experiments): overrides the modes below
  uint32_t *dbgTileCost = nullptr;
  int dbgTileCount = 0;
  int lastPath = 0;  // rm_debug_last_path: the schedule of the most recent render launch on this device
  int lastSplit = 0; // rm_debug_last_split: tiles that launch rendered one light per workgroup (0: none)
  std::map<hipStream_t, SplitTune> splitTune;  // the light split's tuner per stream
  std::map<std::tuple<unsigned long long, int, int, int, int>, int> splitChoice;  // decisions by (picture, W, rows, tile shape, divisor)
};
std::atomic<int> g_tileOrderMode{-1};  // rm_set_tile_order: -1 = take RM_TILE_ORDER or the default
constexpr int kDefaultTileOrder = 1;
DeviceState g_dev[64];
std::atomic<bool> g_timing{false};
std::atomic<int> g_tileShape{-1};  // rm_debug_set_tile_shape: -1 = the RM_TILE_SHAPE environment variable (default 0 = tune), 0 tune, 3 8×8, 2 4×16
std::atomic<bool> g_lightSplitForce{false};  // rm_debug_set_light_split with a divisor: split without measuring
std::atomic<int> g_lightSplit{-1};  // rm_debug_set_light_split: -1 = the RM_LIGHT_SPLIT environment variable (default 256), 0 off, n: the heaviest 1/n of the tiles
std::atomic<int> g_kernelPath{0};  // rm_set_kernel_path: 0 auto, 1 one lane per pixel, 5 wavefront pipeline

#define HIP_OK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                               \
      return RM_ERR_DEVICE;                                                                       \
    }                                                                                             \
  } while (0)

int new_slot(Slot *s) {
  HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&s->host), sizeof(SceneBlock), hipHostMallocDefault));
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&s->dev), sizeof(SceneBlock)));
  HIP_OK(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
  return RM_OK;
}
// Caller holds ds.mu.  Returns a slot whose previous launch (if any) has finished: the next slot of the ring if its event
// has fired, otherwise a fresh one — the enqueue path never blocks on the GPU while it holds the device lock.  Only when
// kSlotsMax launches are in flight does it wait (hipEventSynchronize on the oldest), which bounds pinned memory.
int acquire_slot(DeviceState &ds, Slot **out) {
  if (ds.slots.empty()) {
    // built aside and swapped in only when every allocation has succeeded: a failure part-way (out of memory on the first
    // call) leaves the device state empty, so the next call tries again instead of handing out half-made slots
    std::vector<Slot> fresh(kSlotsInit);
    unsigned long long *counters = nullptr;
    int st = RM_OK;
    for (auto &s : fresh)
      if ((st = new_slot(&s)) != RM_OK) break;
    if (st == RM_OK && hipMalloc(reinterpret_cast<void **>(&counters), 10 * sizeof(unsigned long long)) != hipSuccess) {
      set_error("hipMalloc of the counter block failed");
      st = RM_ERR_DEVICE;
    }
    if (st != RM_OK) {
      for (auto &s : fresh) {
        if (s.host) (void)hipHostFree(s.host);
        if (s.dev) (void)hipFree(s.dev);
        if (s.done) (void)hipEventDestroy(s.done);
      }
      return st;
    }
    ds.slots.swap(fresh);
    ds.dCounters = counters;
  }
  Slot *s = &ds.slots[ds.next];
  if (s->used) {
    const hipError_t q = hipEventQuery(s->done);
    if (q == hipErrorNotReady) {
      if ((int)ds.slots.size() < kSlotsMax) {
        ds.slots.insert(ds.slots.begin() + (long)ds.next, Slot{});  // a fresh slot in front of the busy one keeps ring order
        s = &ds.slots[ds.next];
        int st = new_slot(s);
        if (st != RM_OK) { ds.slots.erase(ds.slots.begin() + (long)ds.next); return st; }
      } else {
        HIP_OK(hipEventSynchronize(s->done));
      }
    } else if (q != hipSuccess) {
      set_error(std::string("hipEventQuery: ") + hipGetErrorString(q));
      return RM_ERR_DEVICE;
    }
  }
  ds.next = (ds.next + 1) % ds.slots.size();
  s->used = true;
  *out = s;
  return RM_OK;
}

// Grow-only scratch memory of one (device, stream, user): see rm_internal.h.
struct WsKey { int dev; hipStream_t stream; int tag; bool operator<(const WsKey &o) const { return std::tie(dev, stream, tag) < std::tie(o.dev, o.stream, o.tag); } };
struct WsBuf { void *mem = nullptr; size_t bytes = 0; };
std::mutex g_wsMu;
std::map<WsKey, WsBuf> g_ws;
}  // namespace

// Largest single workspace buffer the library may allocate (0 = no limit): rm_set_workspace_limit / RM_WF_MAX_WORKSPACE_BYTES.
std::atomic<unsigned long long> g_wsLimit{~0ull};  // ~0 = not set yet: the environment variable decides
unsigned long long workspace_limit() {
  unsigned long long v = g_wsLimit.load();
  if (v == ~0ull) {
    const char *e = std::getenv("RM_WF_MAX_WORKSPACE_BYTES");
    v = e ? std::strtoull(e, nullptr, 10) : 0ull;
    g_wsLimit.store(v);
  }
  return v;
}

int stream_workspace(int tag, hipStream_t stream, size_t need, void **out) {
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  WsBuf *b;
  {
    std::lock_guard<std::mutex> lock(g_wsMu);
    b = &g_ws[WsKey{dev, stream, tag}];  // std::map nodes are stable: the pointer outlives the lock
  }
  // only work enqueued on `stream` uses this buffer, and one host thread enqueues on a stream at a time
  if (b->bytes < need) {
    const unsigned long long limit = workspace_limit();
    if (limit && need > limit) {
      set_error("workspace of " + std::to_string(need) + " bytes exceeds the limit of " + std::to_string(limit) + " (rm_set_workspace_limit)");
      return RM_ERR_DEVICE;
    }
    HIP_OK(hipStreamSynchronize(stream));
    if (b->mem) HIP_OK(hipFree(b->mem));
    b->mem = nullptr; b->bytes = 0;
    const hipError_t e = hipMalloc(&b->mem, need);
    if (e != hipSuccess) {
      (void)hipGetLastError();  // an allocation failure is not sticky for the caller: later HIP calls on this thread start clean
      b->mem = nullptr;
      set_error("hipMalloc of a " + std::to_string(need) + "-byte workspace: " + hipGetErrorString(e));
      return RM_ERR_DEVICE;
    }
    b->bytes = need;
  }
  *out = b->mem;
  return RM_OK;
}
// Frees every grow-only buffer of the current device (after the device has drained); rm_release_workspaces.
int release_workspaces(size_t *freedOut) {
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  HIP_OK(hipDeviceSynchronize());
  size_t freed = 0;
  std::lock_guard<std::mutex> lock(g_wsMu);
  for (auto it = g_ws.begin(); it != g_ws.end();) {
    if (it->first.dev == dev) {
      if (it->second.mem) { HIP_OK(hipFree(it->second.mem)); freed += it->second.bytes; }
      it = g_ws.erase(it);
    } else {
      ++it;
    }
  }
  if (freedOut) *freedOut = freed;
  return RM_OK;
}
namespace {

// Carve the wavefront pipeline's records for `cap` hit slots and `nl` lights out of the stream's workspace.
constexpr int kWfBuffers = 13;
size_t wavefront_sizes(size_t cap, int nl, size_t sizes[kWfBuffers]) {
  auto align = [](size_t v) { return (v + 255) & ~size_t(255); };
  const size_t nlq = (size_t)(nl > 0 ? nl : 1);
  const size_t sz[kWfBuffers] = {align(WF_STRIDE * (kWfMaxBounces + 2) * 4), align(cap * 16), align(cap * 16), align(cap * 16), align(cap * 16),
                                 align(cap * 16), align(cap * 16), align(cap * 16), align(cap * nlq * 4), align(cap * 8), align(cap * 16),
                                 align(cap * 16), align(cap * 8)};
  size_t total = 0;
  for (int i = 0; i < kWfBuffers; i++) { sizes[i] = sz[i]; total += sz[i]; }
  return total;
}
size_t wavefront_bytes(size_t cap, int nl) { size_t sizes[kWfBuffers]; return wavefront_sizes(cap, nl, sizes); }
int wavefront_workspace(size_t cap, int nl, hipStream_t stream, WfWs *ws) {
  size_t sizes[kWfBuffers];
  const size_t total = wavefront_sizes(cap, nl, sizes);
  void *mem = nullptr;
  if (int st = stream_workspace(kWsWavefront, stream, total, &mem)) return st;
  char *q = static_cast<char *>(mem);
  int k = 0;
  auto take = [&]() { char *r = q; q += sizes[k++]; return r; };
  ws->counters = reinterpret_cast<uint32_t *>(take());
  ws->rayO[0] = reinterpret_cast<float4 *>(take()); ws->rayO[1] = reinterpret_cast<float4 *>(take());
  ws->rayD[0] = reinterpret_cast<float4 *>(take()); ws->rayD[1] = reinterpret_cast<float4 *>(take());
  ws->hit = reinterpret_cast<int4 *>(take());
  ws->surfP = reinterpret_cast<float4 *>(take());
  ws->surfN = reinterpret_cast<float4 *>(take());
  ws->shadow = reinterpret_cast<float *>(take());
  ws->pathPix = reinterpret_cast<int2 *>(take());
  ws->pathA = reinterpret_cast<float4 *>(take());
  ws->pathB = reinterpret_cast<float4 *>(take());
  ws->pathC = reinterpret_cast<float2 *>(take());
  ws->cap = (uint32_t)cap;
  return RM_OK;
}

bool tex_ok(const RmTexture &t) { return t.pixels && t.width > 0 && t.height > 0; }

int check_device_pointers(const RmResources &res, const float *d_rgba, const float *d_bright) {
  auto bad = [](const char *what) { set_error(std::string(what) + " is not device-accessible memory"); return RM_ERR_INVALID_ARGUMENT; };
  int st = require_device_pointers({{"d_rgba", d_rgba}, {"d_bright", d_bright}});
  if (st != RM_OK) return st;
  for (int i = 0; i < res.numTextures; i++)
    if (res.textures[i].pixels && !device_accessible(res.textures[i].pixels)) return bad("a texture's pixels");
  if (res.noise.pixels && !device_accessible(res.noise.pixels)) return bad("RmResources.noise.pixels");
  for (int f = 0; f < 6; f++)
    if (res.skybox[f].pixels && !device_accessible(res.skybox[f].pixels)) return bad("a sky-box face");
  if (res.ltc1 && !device_accessible(res.ltc1)) return bad("RmResources.ltc1");
  if (res.ltc2 && !device_accessible(res.ltc2)) return bad("RmResources.ltc2");
  return RM_OK;
}
const RmResources kNoResources{};

int validate_scene(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                   const RmGlobals *g, const RmSettings *s, const RmResources &res) {
  const RmTexture *tex = res.textures;
  const int numTex = res.numTextures;
  if (numTex < 0 || (numTex > 0 && !tex)) { set_error("bad texture table"); return RM_ERR_INVALID_ARGUMENT; }
  if (numTex > RM_MAX_TEXTURES) { set_error("more than RM_MAX_TEXTURES textures"); return RM_ERR_CAPACITY; }
  if (!cam || !g || !s || (numObjects > 0 && !objs) || (numLights > 0 && !lights) || numObjects < 0 || numLights < 0) {
    set_error("null scene pointer or negative count");
    return RM_ERR_INVALID_ARGUMENT;
  }
  if (numObjects > RM_MAX_OBJECTS || numLights > RM_MAX_LIGHTS) {
    set_error("scene exceeds RM_MAX_OBJECTS / RM_MAX_LIGHTS");
    return RM_ERR_CAPACITY;
  }
  if (s->maxSteps < 0 || s->fractalIters < 0 || s->mengerLevels < 0 || s->numReflection < 0) {
    set_error("negative loop bound in RmSettings");
    return RM_ERR_INVALID_ARGUMENT;
  }
  if ((s->features & (RM_FEAT_NIGHTSKY_BACKGROUND | RM_FEAT_SEA)) && !tex_ok(res.noise)) {
    set_error("NIGHTSKY_BACKGROUND / SEA read the noise texture: supply RmResources.noise (rm_render_res)");
    return RM_ERR_UNSUPPORTED;
  }
  if (s->enableSkyBox) {
    for (int f = 0; f < 6; f++)
      if (!tex_ok(res.skybox[f])) {
        set_error("enableSkyBox without six cube-map faces in RmResources.skybox (rm_render_res)");
        return RM_ERR_UNSUPPORTED;
      }
  }
  for (int i = 0; i < numObjects; i++) {
    if (objs[i].type < 0 || objs[i].type >= RM_CUSTOM) {
      set_error("object " + std::to_string(i) + ": CUSTOM / unknown type (the reference's sdCUSTOM returns an unset value)");
      return RM_ERR_UNSUPPORTED;
    }
    if (objs[i].texLoc != -1) {
      const int t = objs[i].texLoc, ty = objs[i].type;
      if (t < 0 || t >= numTex) {
        set_error("object " + std::to_string(i) + ": texLoc without a matching texture (use rm_render_ex)");
        return RM_ERR_UNSUPPORTED;
      }
      if (ty != RM_CUBE && ty != RM_CONE && ty != RM_CYLINDER && ty != RM_SPHERE) {
        set_error("object " + std::to_string(i) + ": textures are only defined for cube, cone, cylinder, sphere");
        return RM_ERR_UNSUPPORTED;
      }
      if (!tex_ok(tex[t])) {
        set_error("texture " + std::to_string(t) + ": null pixels or empty size");
        return RM_ERR_INVALID_ARGUMENT;
      }
    }
  }
  for (int i = 0; i < numLights; i++) {
    if (lights[i].type < 0 || lights[i].type > RM_LIGHT_AREA) {
      set_error("light " + std::to_string(i) + ": unknown light type");
      return RM_ERR_UNSUPPORTED;
    }
    if (lights[i].type == RM_LIGHT_AREA && (!res.ltc1 || !res.ltc2)) {
      set_error("light " + std::to_string(i) + ": area lights read the LTC tables: supply RmResources.ltc1/ltc2 (rm_render_res)");
      return RM_ERR_UNSUPPORTED;
    }
  }
  return RM_OK;
}

// A world-space ball that contains every object, grown by a margin δ such that outside it every object's distance value
// exceeds the hit threshold by a wide factor (so a march out there can only miss).  Per object: unit-shape radius r in
// object space (sdMatch's sizes, frag:1262-1293), world centre c = −A⁻¹b and extent r·σ(A⁻¹) of the ball's image under
// the model matrix (A, b = linear part and translation of invModel; σ = largest singular value), and κ = scaleFactor / σ(A⁻¹), a lower bound of
// (distance value) / (world distance to the object's ball) for the exact SDFs.  The Mandelbulb (power 8, |seed| <= 2)
// enters with r = 2.1: beyond it the estimate is >= 0.68·scaleFactor.  Scenes with a type that has no bound here
// (2-D Mandelbrot, Sierpinski) get cullOk = 0.
// Largest singular value of a 3×3 matrix: the largest eigenvalue of the symmetric M·Mᵀ in closed form, padded.
double sigma_max3(const double m[3][3]) {
  double B[3][3];
  for (int r0 = 0; r0 < 3; r0++)
    for (int c0 = 0; c0 < 3; c0++) B[r0][c0] = m[r0][0] * m[c0][0] + m[r0][1] * m[c0][1] + m[r0][2] * m[c0][2];
  const double p1 = B[0][1] * B[0][1] + B[0][2] * B[0][2] + B[1][2] * B[1][2];
  const double q = (B[0][0] + B[1][1] + B[2][2]) / 3.0;
  const double p2 = (B[0][0] - q) * (B[0][0] - q) + (B[1][1] - q) * (B[1][1] - q) + (B[2][2] - q) * (B[2][2] - q) + 2.0 * p1;
  double lmax;
  if (!(p2 > 1e-300)) lmax = q;
  else {
    const double pp = std::sqrt(p2 / 6.0);
    double C3[3][3];
    for (int r0 = 0; r0 < 3; r0++)
      for (int c0 = 0; c0 < 3; c0++) C3[r0][c0] = (B[r0][c0] - (r0 == c0 ? q : 0.0)) / pp;
    double hd = (C3[0][0] * (C3[1][1] * C3[2][2] - C3[1][2] * C3[2][1]) - C3[0][1] * (C3[1][0] * C3[2][2] - C3[1][2] * C3[2][0]) +
                 C3[0][2] * (C3[1][0] * C3[2][1] - C3[1][1] * C3[2][0])) / 2.0;
    hd = hd < -1.0 ? -1.0 : (hd > 1.0 ? 1.0 : hd);
    lmax = q + 2.0 * pp * std::cos(std::acos(hd) / 3.0);
  }
  return std::sqrt(lmax > 0.0 ? lmax : 0.0) * (1.0 + 1e-6);
}

void scene_cull_ball(SceneBlock *h) {
  h->cullOk = 0;
  h->objBallOk = 0;
  h->cullC[0] = h->cullC[1] = h->cullC[2] = 0.0f;
  h->cullR2 = 0.0f;
  h->cullR2Soft = 0.0f;
  h->cullBoxOk = 0;
  for (int k = 0; k < 3; k++) h->cullLo[k] = h->cullHi[k] = 0.0f;
  {  // Lipschitz bound of the distance values per unit of world length (the skip test's seed, rm_device.hip.h nextMinBound):
     // scaleFactor × the stretch of invModel's linear part, for the shapes whose SDF is 1-Lipschitz in object space
    double lip = 0.0;
    for (int i = 0; i < h->numObjects; i++) {
      const RmObject &o = h->objs[i];
      const bool lipschitz = (o.type >= RM_CUBE && o.type <= RM_RECTANGLE) || o.type == RM_MENGERSPONGE;
      const float *M = o.invModel;
      const double a[3][3] = {{M[0], M[4], M[8]}, {M[1], M[5], M[9]}, {M[2], M[6], M[10]}};
      const double li = lipschitz ? std::fabs((double)o.scaleFactor) * sigma_max3(a) : INFINITY;
      lip = (li > lip || !(li == li)) ? li : lip;
    }
    h->cullLip = (std::isfinite(lip) && lip < 1e6) ? (float)(lip * (1.0 + 1e-5)) : INFINITY;
    bool prim = h->numObjects > 0;
    for (int i = 0; i < h->numObjects; i++) prim = prim && h->objs[i].type >= RM_CUBE && h->objs[i].type <= RM_RECTANGLE;
    h->cullOneOk = (prim && std::isfinite(h->cullLip)) ? 1 : 0;
  }
  const int n = h->numObjects;
  if (n <= 0) return;
  // half-extents of the unit shapes' object-space bounding boxes (sdMatch's sizes; the capsule's segment runs from 0 to 0.5 in y)
  static const double kExtent[][3] = {{.5, .5, .5}, {.5, .5, .5}, {.5, .5, .5}, {.5, .5, .5}, {.5, .5, .5}, {.625, .125, .625},
                                      {.1, .6, .1}, {.5, .5, .5}, {.5, .5, 0.0}};  // cube … rectangle
  double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
  static const double kRadius[] = {0.8661, 0.7072, 0.7072, 0.5001, 0.5001, 0.6251, 0.6001, 0.5001, 0.7072};  // cube … rectangle
  double cx[RM_MAX_OBJECTS], cy[RM_MAX_OBJECTS], cz[RM_MAX_OBJECTS], rad[RM_MAX_OBJECTS];
  double kappa = 1e30, kappaSoft = 1e30, C[3] = {0, 0, 0};
  for (int i = 0; i < n; i++) {
    const RmObject &o = h->objs[i];
    double r;
    if (o.type >= RM_CUBE && o.type <= RM_RECTANGLE) r = kRadius[o.type];
    else if (o.type == RM_MENGERSPONGE) r = 1.7322;
    else if (o.type == RM_MANDELBULB) {
      const double jx = h->g.juliaSeed[0], jy = h->g.juliaSeed[1];
      if (!(h->g.power == 8.0f) || !(jx * jx + jy * jy <= 4.0) || !(o.scaleFactor >= 0.01f)) return;
      r = 2.1;
    } else return;
    const float *M = o.invModel;
    const double a[3][3] = {{M[0], M[4], M[8]}, {M[1], M[5], M[9]}, {M[2], M[6], M[10]}};  // a[row][col]
    const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                       a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    if (!(std::fabs(det) > 1e-12) || !std::isfinite(det)) return;
    double inv[3][3];
    inv[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / det; inv[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / det;
    inv[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / det; inv[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / det;
    inv[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / det; inv[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / det;
    inv[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / det; inv[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / det;
    inv[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / det;
    // nf = the largest singular value of A⁻¹ (how much the model matrix can stretch a length): the largest eigenvalue of the
    // symmetric B = A⁻¹·A⁻¹ᵀ in closed form, with a relative safety margin.  (The Frobenius norm used before is an upper bound
    // too, but √3 too large for a uniform scale: every ball was 1.7× wider than it had to be.)
    const double nf = sigma_max3(inv);
    const double b[3] = {M[12], M[13], M[14]};
    cx[i] = -(inv[0][0] * b[0] + inv[0][1] * b[1] + inv[0][2] * b[2]);
    cy[i] = -(inv[1][0] * b[0] + inv[1][1] * b[1] + inv[1][2] * b[2]);
    cz[i] = -(inv[2][0] * b[0] + inv[2][1] * b[1] + inv[2][2] * b[2]);
    rad[i] = r * nf;
    {
      double e[3] = {r, r, r};  // Menger sponge: the box of half-size 1 (r = √3 is its corner); Mandelbulb: the ball's box
      if (o.type >= RM_CUBE && o.type <= RM_RECTANGLE) for (int k = 0; k < 3; k++) e[k] = kExtent[o.type][k] + 1e-4;
      else if (o.type == RM_MENGERSPONGE) e[0] = e[1] = e[2] = 1.0001;
      const double c[3] = {cx[i], cy[i], cz[i]};
      for (int k = 0; k < 3; k++) {
        const double w = std::fabs(inv[k][0]) * e[0] + std::fabs(inv[k][1]) * e[1] + std::fabs(inv[k][2]) * e[2];
        lo[k] = std::fmin(lo[k], c[k] - w);
        hi[k] = std::fmax(hi[k], c[k] + w);
      }
    }
    const double ki = (double)o.scaleFactor / nf;
    if (!(ki > 1e-6) || !std::isfinite(rad[i]) || !std::isfinite(cx[i] + cy[i] + cz[i])) return;
    // hard bound: the bulb's constant 0.68·scaleFactor needs no δ; soft bound: beyond ρ = 2.1 its estimate ≈ 0.5·ρ·ln ρ has
    // slope >= 0.87 in object space
    if (o.type != RM_MANDELBULB) kappa = ki < kappa ? ki : kappa;
    const double ksi = (o.type == RM_MANDELBULB) ? 0.8 * ki : ki;
    kappaSoft = ksi < kappaSoft ? ksi : kappaSoft;
    C[0] += cx[i] / n; C[1] += cy[i] / n; C[2] += cz[i] / n;
  }
  for (int i = 0; i < n; i++) {  // the per-object balls, for the geometric tile order (the bulb's tight radius where it holds)
    const RmObject &o = h->objs[i];
    double r = rad[i];
    if (o.type == RM_MANDELBULB) {
      const double jx = h->g.juliaSeed[0], jy = h->g.juliaSeed[1];
      if (o.scaleFactor >= 0.05f && jx * jx + jy * jy <= 1.2996) r = rad[i] * (1.15 / 2.1);
    }
    h->objBall[i][0] = (float)cx[i]; h->objBall[i][1] = (float)cy[i]; h->objBall[i][2] = (float)cz[i]; h->objBall[i][3] = (float)r;
  }
  h->objBallOk = 1;
  double R = 0.0;
  for (int i = 0; i < n; i++) {
    const double d = std::sqrt((cx[i] - C[0]) * (cx[i] - C[0]) + (cy[i] - C[1]) * (cy[i] - C[1]) + (cz[i] - C[2]) * (cz[i] - C[2])) + rad[i];
    R = d > R ? d : R;
  }
  if (kappa > 1e29) kappa = 1.0;                         // only Mandelbulbs: any margin does
  const double delta = std::fmax(0.05, 4.0e-3 / kappa);  // κ·δ >= 4× the hit threshold
  R = (R + delta) * 1.001;
  if (!std::isfinite(R) || R > 1e6) return;
  h->cullC[0] = (float)C[0]; h->cullC[1] = (float)C[1]; h->cullC[2] = (float)C[2];
  h->cullR2 = (float)(R * R);
  h->cullOk = 1;
  // The same argument for the axis-aligned box around the objects' bounding boxes, grown by the same margin δ: a point outside
  // it is at least δ away from every object's box, so every distance value there exceeds 4× the hit threshold.  Hard-shadow,
  // primary and bounce marches end where their ray leaves ball ∩ box (flat or elongated scenes: the box is much tighter).
  bool boxOk = true;
  for (int k = 0; k < 3; k++) {
    const double m = delta * 1.001 + 1e-3 * std::fmax(std::fabs(lo[k]), std::fabs(hi[k]));
    lo[k] -= m; hi[k] += m;
    boxOk = boxOk && std::isfinite(lo[k]) && std::isfinite(hi[k]) && hi[k] > lo[k] && std::fabs(lo[k]) < 1e6 && std::fabs(hi[k]) < 1e6;
  }
  static const bool boxOn = [] { const char *e = getenv("RM_CULL_BOX"); return !e || atoi(e) != 0; }();
  // Only where the box is much tighter than the ball (flat or elongated scenes: a floor slab, a row of objects): for a compact
  // scene — the lone Menger cube of C5: box / ball volume 0.39 — the three reciprocals per ray cost more than the 5 % of
  // evaluations they save (measured: 21.9 -> 22.3 ms), while directional_light_2.json (0.07) executes 16 % fewer evaluations.
  const double volBox = (hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]), volBall = 4.18879 * R * R * R;
  boxOk = boxOk && volBox < 0.3 * volBall;
  if (boxOk && boxOn) {
    for (int k = 0; k < 3; k++) { h->cullLo[k] = (float)lo[k]; h->cullHi[k] = (float)hi[k]; }
    h->cullBoxOk = 1;
  }
  // Soft shadows: a shadow ray starts on a surface, i.e. inside the ball (radius R), and at distance ρ from the centre has
  // travelled t <= ρ + R while every distance value is >= κ·(ρ − R).  8·κ·(ρ − R) >= ρ + R  ⇔  ρ >= R·(8κ + 1)/(8κ − 1):
  // past that radius min(pen, 8·d/t) is settled.
  const double ks = kappaSoft;
  if (ks > 0.2 && ks < 1e29) {
    const double Rs = R * (8.0 * ks + 1.0) / (8.0 * ks - 1.0) * 1.001;
    if (std::isfinite(Rs) && Rs < 1e6) h->cullR2Soft = (float)(Rs * Rs);
  }
}

// nearClip / farClip (raymarch.vert:23-24) at the corners of the full-screen quad, as the vertex shader computes them, per
// triangle: P0, P1 − P0, P2 − P0 with P0 = (sg, sg), P1 = (−sg, sg), P2 = (sg, −sg), sg = −1 below the TL-BR diagonal and
// +1 above it.  invProjView·(x, y, z, 1) = ((M0·x + M1·y) + M2·z) + M3, fused — the oracle's mat4_mul_v4, on the host's
// binary32 FMA (the same bits on any IEEE machine).
void ray_planes(SceneBlock *h) {
  const float *M = h->cam.invProjView;
  auto corner = [&](float x, float y, float z, float out[4]) {
    for (int c = 0; c < 4; c++) out[c] = std::fmaf(M[12 + c], 1.0f, std::fmaf(M[8 + c], z, std::fmaf(M[4 + c], y, M[c] * x)));
  };
  for (int tri = 0; tri < 2; tri++) {
    const float sg = tri ? 1.0f : -1.0f;
    for (int k = 0; k < 2; k++) {
      const float z = k ? 1.0f : -1.0f;
      float p0[4], p1[4], p2[4];
      corner(sg, sg, z, p0); corner(-sg, sg, z, p1); corner(sg, -sg, z, p2);
      for (int c = 0; c < 4; c++) {
        h->rayPlane[tri][k][0][c] = p0[c];
        h->rayPlane[tri][k][1][c] = p1[c] - p0[c];
        h->rayPlane[tri][k][2][c] = p2[c] - p0[c];
      }
    }
  }
}

// What an evaluation reads of an object (SceneBlock::evalRec), incl. the bound of the table walk's pass-over test, from h->objs.
void scene_eval_records(SceneBlock *h) {
  for (int i = 0; i < h->numObjects; i++) {
    const RmObject &o = h->objs[i];
    EvalRecord &e = h->evalRec[i];
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 3; r++) e.m[c * 3 + r] = o.invModel[c * 4 + r];
    e.scaleFactor = o.scaleFactor;
    e.type = o.type;
    // the skip test's bound (rm_device.hip.h, sdScene<…, SKIP>): radius of the unit shape's bounding ball, with a margin
    static const float kBound[] = {0.8662f, 0.7073f, 0.7073f, 0.5001f, 0.5001f, 0.6252f, 0.6002f, 0.5001f, 0.7073f};  // cube … rectangle
    const float sf = o.scaleFactor;
    const bool ok = std::isfinite(sf) && sf > 1e-6f && sf < 1e6f;
    e.invScale = ok ? 1.0f / sf : 0.0f;
    // the primitives only: a fractal's evaluation also writes the orbit trap that sdScene returns — the trap of the LAST
    // evaluated fractal in table order, nearest or not — so passing over one would change it
    e.boundR = (ok && o.type >= RM_CUBE && o.type <= RM_RECTANGLE) ? kBound[o.type] : INFINITY;
  }
}

int stage_scene(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                const RmGlobals *g, const RmSettings *s, hipStream_t stream, DeviceState &ds, Slot **slotOut,
                const RmResources &res, const int32_t *tileOrder = nullptr, uint32_t *tileCost = nullptr,
                int tileCount = 0, int tileShift = 3, int splitTiles = 0, float *splitStore = nullptr) {  // caller holds ds.mu
  Slot *slot;
  int st = acquire_slot(ds, &slot);
  if (st != RM_OK) return st;
  SceneBlock *h = slot->host;
  h->cam = *cam; h->g = *g; h->s = *s;
  h->numObjects = numObjects; h->numLights = numLights;
  for (int i = 0; i < numObjects; i++) h->objs[i] = objs[i];
  scene_eval_records(h);
  for (int i = 0; i < numLights; i++) h->lights[i] = lights[i];
  h->numTextures = res.numTextures;
  for (int i = 0; i < res.numTextures; i++) h->tex[i] = res.textures[i];
  h->noise = res.noise;
  for (int f = 0; f < 6; f++) h->skybox[f] = res.skybox[f];
  h->ltc1 = res.ltc1; h->ltc2 = res.ltc2;
  scene_cull_ball(h);
  ray_planes(h);
  h->tileOrder = tileOrder; h->tileCost = tileCost; h->tileCount = tileCount;
  h->tileShift = tileShift;
  h->splitTiles = splitTiles; h->splitStore = splitStore;
  h->mengerAni = 0.0f; h->mengerOff = 0.0f;
  HIP_OK(hipMemcpyAsync(slot->dev, h, sizeof(SceneBlock), hipMemcpyHostToDevice, stream));
  bool menger = false;
  for (int i = 0; i < numObjects; i++) menger = menger || objs[i].type == RM_MENGERSPONGE;
  if (menger) {  // stream-ordered between the upload and the kernels that read the block
    hipLaunchKernelGGL(scene_prep_kernel, dim3(1), dim3(1), 0, stream, slot->dev);
    HIP_OK(hipGetLastError());
  }
  *slotOut = slot;
  return RM_OK;
}

// Whether the wavefront pipeline is expected to beat the one-lane-per-pixel kernel on this scene (measured).
// Measured (profiles/r03_b_wavefront.md): with reflection bounces the regrouping wins from 4K frames up (8K Menger frame
// with two bounces 39.0 -> 24.4 ms, the same scene at 4K 12.0 -> 9.2 ms, reflections_complex.json at 4K with two bounces
// 25.4 -> 20.2 ms, with one 16.8 -> 16.4 ms); at 1080p its dozen launches of persistent waves cost more than the idle lanes
// they remove (4.5 -> 5.2 ms, 5.4 -> 7.2 ms), and without secondary rays the one-lane-per-pixel kernel keeps 89-95 % of its
// lanes busy by itself (directional_light_2.json: 1.3 ms against 3.5 ms).
// Round 3, after the table walk learnt to pass over far objects (sdScene<…, SKIP>) and to follow a single object (march()'s
// fast path, all-primitive tables): for all-primitive tables the one-lane-per-pixel kernel is ahead at every bounce count
// (reflections_complex.json 4K: 7.5 ms against 12.4 with one bounce, 12.4 against 15.7 with two) — in the wavefront kernels a
// wave's lanes are unrelated rays, and both tests need the whole wave to agree.  Mixed tables (primitives and a fractal): the
// pass-over test applies, the fast path does not; two or more bounces as measured before the fast path.
bool skip_applies(const RmObject *objs, int numObjects) {
  bool prim = false;
  for (int i = 0; i < numObjects; i++) prim = prim || (objs[i].type >= RM_CUBE && objs[i].type <= RM_RECTANGLE);
  return prim && numObjects >= 2;
}
bool all_primitives(const RmObject *objs, int numObjects) {
  bool prim = numObjects > 0;
  for (int i = 0; i < numObjects; i++) prim = prim && objs[i].type >= RM_CUBE && objs[i].type <= RM_RECTANGLE;
  return prim;
}
// Size threshold: whole frames and row ranges from 2^22 pixels (one launch after the other on a stream: 4K and up).  Row-TILE
// shards (rm_render_tiles with numShards > 1) come from multi-GPU hosts, which keep several frames in flight per GPU
// (dist.FramePipeline, scripts/mgpu_host.cpp): the pipeline's dozen launches per frame then overlap those of its neighbours and
// it pays from 2^21 pixels — measured on shards of the C5 scene with three frames in flight (profiles/r04_j_c5_shards.md):
// 4.18 M pixels (1/8 of the 8K frame) 3.41 against 4.34 ms per frame, 2.09 M 2.47 against 2.73, 1.04 M 1.70 against 1.44.
bool wavefront_pays(const RmObject *objs, int numObjects, int bounces, size_t pixels, bool tileShard) {
  if (all_primitives(objs, numObjects)) return false;
  return bounces >= (skip_applies(objs, numObjects) ? 2 : 1) && pixels >= (size_t(1) << (tileShard ? 21 : 22));
}

int launch_render(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                  const RmGlobals *g, const RmSettings *s, int W, int H, RowMap map, int nRows, float *d_rgba,
                  float *d_bright, hipStream_t stream, int count, RmCounters *countersOut,
                  const RmResources &res = kNoResources, double *clockMHz = nullptr, unsigned long long *d_waveSpans = nullptr) {
  // count: 0 production launch, 1 / 2 counted (reference work / executed work; synchronises), 3 production code with clock stamps
  int st = validate_scene(cam, objs, numObjects, lights, numLights, g, s, res);
  if (st != RM_OK) return st;
  if (W <= 0 || H <= 0 || nRows < 0) { set_error("bad frame size"); return RM_ERR_INVALID_ARGUMENT; }
  if (nRows == 0) return RM_OK;  // empty row range: nothing to write, a null buffer is fine
  if (!d_rgba) { set_error("null output buffer"); return RM_ERR_INVALID_ARGUMENT; }
  if ((st = check_device_pointers(res, d_rgba, d_bright)) != RM_OK) return st;
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { set_error("device index out of range"); return RM_ERR_DEVICE; }
  DeviceState &ds = g_dev[dev];
  std::lock_guard<std::mutex> lock(ds.mu);  // this device only; nothing below blocks on the GPU unless `count` asks for numbers back
  const dim3 block(256);
  const bool bulb = (numObjects == 1 && objs[0].type == RM_MANDELBULB);
  auto nonzero3 = [](const float *v) { return v[0] != 0.0f || v[1] != 0.0f || v[2] != 0.0f; };
  // Two schedules of the same per-ray arithmetic, identical bits: rm::render_kernel (one lane per pixel; every class, and the
  // counted variants) and, for table-walk classes with bounces, the wavefront pipeline.
  static const int envPath = std::getenv("RM_KERNEL_PATH") ? std::atoi(std::getenv("RM_KERNEL_PATH")) : 0;
  const int pathReq = g_kernelPath.load() ? g_kernelPath.load() : envPath;  // 0 = the measured-fastest schedule of the scene's class
  const bool envFeatures = (s->features & (RM_FEAT_TERRAIN | RM_FEAT_CLOUD | RM_FEAT_SKY_BACKGROUND | RM_FEAT_NIGHTSKY_BACKGROUND | RM_FEAT_SEA)) != 0;
  // anything that reads a sampler or takes the area-light branches: object textures, sky box, emissive rectangles, area lights
  bool textured = s->enableSkyBox != 0;
  for (int i = 0; i < numObjects; i++) textured = textured || objs[i].texLoc != -1 || objs[i].isEmissive;
  for (int i = 0; i < numLights; i++) textured = textured || lights[i].type == RM_LIGHT_AREA;
  // The wavefront pipeline (rm_wavefront.hip.h) covers the table-walk classes whose evaluations cost the same on every
  // lane: no Mandelbulb / 2-D Mandelbrot in the table, no samplers or procedural layers, no refraction.
  bool wfOk = !bulb && !count && !envFeatures && !textured && !g->isTwoD && s->maxSteps >= 1 && s->numReflection <= kWfMaxBounces;
  bool anyReflective = false;
  for (int i = 0; i < numObjects; i++) {
    if (objs[i].type == RM_MANDELBULB || objs[i].type == RM_MANDELBROT) wfOk = false;
    if (s->enableRefraction && nonzero3(objs[i].cTransparent)) wfOk = false;
    anyReflective = anyReflective || nonzero3(objs[i].cReflective);
  }
  const int wfBounces = (s->enableReflection && anyReflective) ? s->numReflection : 0;
  // whether main's secondary rays (frag:2491-2570) can fire for any pixel of this frame: a reflective object with reflection on and
  // at least one bounce, or a transparent one with refraction on — otherwise the plain instantiations compile them out (SEC = false)
  bool anyTransparent = false;
  for (int i = 0; i < numObjects; i++) anyTransparent = anyTransparent || nonzero3(objs[i].cTransparent);
  const bool secondary = (s->enableReflection && anyReflective && s->numReflection > 0) || (s->enableRefraction && anyTransparent);
  const bool wfSkip = skip_applies(objs, numObjects);
  bool wavefront = wfOk && (pathReq == 5 || (pathReq == 0 && wavefront_pays(objs, numObjects, wfBounces, (size_t)nRows * W, map.numShards > 1)));
  // The wavefront pipeline's knobs and records, settled BEFORE anything below depends on `wavefront`: if its workspace
  // (≈(160 + 4·numLights) B per hit slot, grow-only per (device, stream): 5.8 GB for an 8K frame) cannot be had, the
  // auto-selected launch falls back to render_kernel — identical bits, no workspace — and only an explicit path-5 request
  // reports the failure.  A (device, stream) that was refused once is not asked again for as much or more, so a frame
  // sequence does not pay a failing allocation (and the stream synchronisation in front of it) per frame.
  // Tuning knobs for A/B runs (defaults are the measured best); chunk sizes are clamped so that slot and ray ids stay 32-bit.
  auto envInt = [](const char *name) { const char *e = std::getenv(name); return e ? std::atoi(e) : 0; };
  static const int wavesPerSimd = envInt("RM_WF_WAVES_PER_SIMD"), envFlush = envInt("RM_WF_FLUSH"), envSlotChunk = envInt("RM_WF_SLOT_CHUNK"),
                   envRayChunk = envInt("RM_WF_RAY_CHUNK"), envPixelChunk = envInt("RM_WF_PIXEL_CHUNK"), envMaxChunk = envInt("RM_WF_MAX_CHUNK");
  constexpr int kWfChunkMax = 4096;
  auto clampChunk = [](int v) { return (uint32_t)(v > kWfChunkMax ? kWfChunkMax : v); };
  const uint32_t slotChunk = envSlotChunk >= 64 ? clampChunk(envSlotChunk) : kWfSlotChunk;  // >= 64: one trip's hits fit one fresh chunk
  const uint32_t maxChunk = envMaxChunk > 0 ? clampChunk(envMaxChunk) : 0u;  // 0: fixed chunks (guided chunks measured slower)
  const uint32_t rayChunk = envRayChunk > 0 ? clampChunk(envRayChunk) : wfRayChunk(1), pixelChunk = envPixelChunk > 0 ? clampChunk(envPixelChunk) : wfRayChunk(0);
  WfWs wfWs{};
  int wfPrimaryWaves = 0, wfShadowWaves = 0;
  if (wavefront) {
    if (ds.numCUs == 0) {
      hipDeviceProp_t prop;
      HIP_OK(hipGetDeviceProperties(&prop, dev));
      ds.numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // persistent waves: as many as are resident at once (4 SIMDs per CU x the kernel's register budget)
    auto waves = [&](int kind) { return ds.numCUs * 4 * (wavesPerSimd > 0 && wavesPerSimd < wfMarchWaves(kind) ? wavesPerSimd : wfMarchWaves(kind)); };
    wfPrimaryWaves = waves(0); wfShadowWaves = waves(2);
    const int marchWaves = wfShadowWaves > wfPrimaryWaves ? wfShadowWaves : wfPrimaryWaves;
    // hit-slot capacity: every ray may hit, plus one partly used chunk of slots per persistent wave
    const size_t cap = (size_t)nRows * W + (size_t)slotChunk * marchWaves;
    // 32-bit ids: hit slots x lights (shadow rays) and the striped cursors' padding (one chunk per stripe) stay below 2^32
    const size_t chunkMax = rayChunk > pixelChunk ? rayChunk : pixelChunk;
    const bool idsFit = (double)(cap + (size_t)kWfStripes * chunkMax) * (numLights > 0 ? numLights : 1) < 4.0e9;
    int wst = RM_ERR_DEVICE;
    auto denied = ds.wfDenied.find(stream);
    const size_t wfBytes = wavefront_bytes(cap, numLights);
    if (!idsFit) set_error("frame too large for the wavefront pipeline's 32-bit ray ids");
    else if (denied != ds.wfDenied.end() && wfBytes >= denied->second) set_error("wavefront workspace was refused on this stream before");
    else if ((wst = wavefront_workspace(cap, numLights, stream, &wfWs)) != RM_OK) ds.wfDenied[stream] = wfBytes;
    if (wst != RM_OK) {
      if (pathReq == 5 && idsFit) return wst;  // an explicit request reports the workspace failure; a frame the ids cannot cover "does not apply"
      wavefront = false;
    }
  }
  // Waves (8×8 tiles, side by side) per workgroup.  A workgroup's registers and LDS come free only when its LAST wave
  // ends, and march lengths differ a lot between neighbouring tiles, so small workgroups keep more waves resident: one wave
  // per workgroup for every class (measured at the register budgets above: the 4K bulb frame 2.31 / 2.34 / 2.58 ms at
  // 1 / 2 / 4 waves, the 8K Menger frame 51.0 / 51.8 / 58.6 ms, bump + reflection at 4K 19.4 / 19.8 / 22.1 ms; at the
  // compiler's own budgets two waves were best for the bulb, profiles/r02_c_waves_per_block.md).  RM_WAVES_PER_BLOCK overrides.
  static const int wpb = std::getenv("RM_WAVES_PER_BLOCK") ? std::atoi(std::getenv("RM_WAVES_PER_BLOCK")) : 0;
  const int nw = (wpb == 1 || wpb == 2 || wpb == 4) ? wpb : 1;
  // the picture this launch renders: everything that decides a pixel (FNV-1a over the caller's tables and the row map) — what the
  // tile-order feedback and the tile-shape tuner key their measurements by
  unsigned long long key = 1469598103934665603ull;
  {
    auto mix = [&](const void *p, size_t nb) {
      const unsigned char *b8 = static_cast<const unsigned char *>(p);
      for (size_t k = 0; k < nb; k++) key = (key ^ b8[k]) * 1099511628211ull;
    };
    mix(cam, sizeof(*cam)); mix(g, sizeof(*g)); mix(s, sizeof(*s)); mix(&map, sizeof(map));
    mix(objs, sizeof(RmObject) * (size_t)numObjects); mix(lights, sizeof(RmLight) * (size_t)numLights);
  }
  // Tile shape ("tile shape" above): 8×8 unless the tuner is measuring or has chosen 4×16 for this picture on this stream
  static const int envShape = std::getenv("RM_TILE_SHAPE") ? std::atoi(std::getenv("RM_TILE_SHAPE")) : 0;
  const int shapeReq = g_tileShape.load() >= 0 ? g_tileShape.load() : envShape;
  const bool bigFrame = (size_t)nRows * W >= (size_t)2048 * 64;
  int tileShift = (RM_TILE_W == 8) ? 3 : (RM_TILE_W == 4 ? 2 : (RM_TILE_W == 16 ? 4 : 3));
  ShapeTune *tune = nullptr;  // non-null: this launch is one of the tuner's (frames 0-3 of a picture) or follows its choice
  int tuneTimed = -1;         // 0..3: time this launch as the tuner's candidate (k & 1: 0 = 8×8, 1 = 4×16) of round k >> 1
  if ((shapeReq == 2 || shapeReq == 3) && count == 0) tileShift = shapeReq;  // the counted / stamped diagnostic builds keep 8×8 (their callers size per-wave arrays by it)
  else if (RM_TILE_W == 8 && !bulb && !wavefront && !g->isTwoD && count == 0 && bigFrame && !ds.dbgTileOrder && !ds.dbgTileCost) {
    tune = &ds.shapeTune[stream];
    if (tune->key != key || tune->W != W || tune->nRows != nRows) {
      tune->drop();
      tune->key = key; tune->W = W; tune->nRows = nRows; tune->frame = 0; tune->chosen = -1;
      const auto known = ds.shapeChoice.find(std::make_tuple(key, W, nRows));  // another stream of this device measured this picture
      if (known != ds.shapeChoice.end()) tune->chosen = known->second;
    }
    if (tune->chosen < 0 && tune->frame >= 8) {
      bool ready = true;
      for (int k = 0; k < 4; k++) ready = ready && tune->timed[k] && hipEventQuery(tune->ev[k][1]) == hipSuccess;
      if (ready) {
        float best[2] = {1e30f, 1e30f};
        bool ok = true;
        for (int k = 0; k < 4; k++) {
          float ms = 0.0f;
          ok = ok && hipEventElapsedTime(&ms, tune->ev[k][0], tune->ev[k][1]) == hipSuccess;
          best[k & 1] = ms < best[k & 1] ? ms : best[k & 1];
        }
        tune->chosen = (ok && best[1] < 0.97f * best[0]) ? 2 : 3;  // 4×16 must win by 3 %
        tune->drop();
        if (ds.shapeChoice.size() >= 256) ds.shapeChoice.clear();
        ds.shapeChoice[std::make_tuple(key, W, nRows)] = tune->chosen;
      }
    }
    (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is not an error
    if (tune->chosen >= 0) tileShift = tune->chosen;
    else {
      // frames 0-1, 4-5: 8×8; 2-3, 6-7: 4×16; from frame 8 until the timings are in (a host that enqueues far ahead of the GPU): 8×8 —
      // the enqueue path never waits for them
      const int f = tune->frame;
      tileShift = (f < 8 && ((f >> 1) & 1)) ? 2 : 3;
      if (f < 8 && (f & 1)) tuneTimed = ((f >> 1) & 1) | ((f >> 2) << 1);  // candidate | round << 1
    }
    tune->frame++;
  }
  const int tileW = 1 << tileShift, tileH = 64 >> tileShift;
  const dim3 rgrid((W + nw * tileW - 1) / (nw * tileW), (nRows + tileH - 1) / tileH), rblock(64 * nw);
  // Tile order ("tile order" above): 0 raster order, 1 feedback — tiles start heaviest-first by the costs the previous frame
  // of this size on this stream recorded.  Small frames are not worth the two extra launches.
  static const int envOrder = std::getenv("RM_TILE_ORDER") ? std::atoi(std::getenv("RM_TILE_ORDER")) : kDefaultTileOrder;
  const int orderMode = g_tileOrderMode.load() >= 0 ? g_tileOrderMode.load() : envOrder;
  const int tileCount = (int)(rgrid.x * rgrid.y);
  // every class of the one-lane-per-pixel kernel (round 3: the layer and sampler kernels too — area light + point light 1080p
  // 0.80 -> 0.59 ms, textured floor / sky box at 4K 2.5 -> 2.3 ms, terrain + cloud horizon view 4.31 -> 4.04 ms, sea unchanged)
  const bool ordered = orderMode > 0 && !wavefront && !g->isTwoD && count == 0 &&
                       tileCount >= 2048 && !ds.dbgTileOrder && !ds.dbgTileCost;
  uint32_t *oCost = nullptr, *oHist = nullptr, *oCost2 = nullptr;
  int32_t *oOrder = nullptr;
  bool haveCost = false, samePicture = false;
  // A picture that repeats SETTLES: its first frames re-sort by the costs the frame before measured (each under a better order than
  // the last); the kSettle-th such sort keeps its costs (they are the stale costs of whatever picture comes next) and from then on
  // the same order is reused — no ordering launches (memset + two kernels, ≈25 µs a frame: 1 % of the 4K bulb frame, 8 % of its
  // 1/8 shard) and no cost atomics in the render.  RM_TILE_ORDER_SETTLE=0: re-sort every frame (rounds 2-3).
  static const int kSettle = [] { const char *e = std::getenv("RM_TILE_ORDER_SETTLE"); const int v = e ? std::atoi(e) : 3; return v < 0 ? 0 : (v > 1000 ? 1000 : v); }();
  const int kSettleHold = kSettle + 1;
  int costSorts = 0;  // this frame is the costSorts-th consecutive cost-ordered frame of its picture (0: not cost-ordered)
  if (ordered) {
    void *mem = nullptr;
    if ((st = stream_workspace(kWsTileOrder, stream, (size_t)tileCount * 12 + 256, &mem)) != RM_OK) return st;
    oHist = static_cast<uint32_t *>(mem);
    oCost = oHist + 64;
    oOrder = reinterpret_cast<int32_t *>(oCost + tileCount);
    oCost2 = oCost + 2 * (size_t)tileCount;  // a new picture's estimates (tile_geom_kernel), so that it can read its neighbours' stale costs
    TileOrderState &ts = ds.tileOrder[stream];
    const TileOrderState now{tileCount, W, nRows, nw, tileShift, mem, key};
    haveCost = ts.tileCount == now.tileCount && ts.W == W && ts.nRows == nRows && ts.nw == nw && ts.tileShift == tileShift && ts.mem == mem;
    if (!haveCost) HIP_OK(hipMemsetAsync(oCost, 0, (size_t)tileCount * 4, stream));
    samePicture = haveCost && ts.sceneKey == key;
    costSorts = samePicture ? ts.sorts + 1 : 0;
    ts = now;
    ts.sorts = costSorts > kSettleHold ? kSettleHold : costSorts;
  }
  // Which order this frame's tiles start in: the previous frame's measured costs when it was the same picture; otherwise — no
  // history, or the scene / camera moved — the geometric classification (tile_geom_kernel), where the scene has per-object balls
  // and no procedural layers (their cost is not where the objects are); otherwise raster order.
  static const int geomMode = [] { const char *e = std::getenv("RM_TILE_ORDER_GEOMETRIC"); return e ? std::atoi(e) : 2; }();  // 0 off (raster), 1 geometry alone, 2 geometry + stale costs (measured best, default)
  const bool geomOn = geomMode != 0;
  static const int ringCombined = [] { const char *e = std::getenv("RM_GEOM_RING_LOG2"); const int v = e ? std::atoi(e) : 16; return v < 5 ? 5 : (v > 17 ? 17 : v); }();
  static const int geomDilate = [] { const char *e = std::getenv("RM_GEOM_DILATE"); const int v = e ? std::atoi(e) : 0; return v < 0 ? 0 : (v > 16 ? 16 : v); }();
  const bool byCost = ordered && samePicture;
  const bool lastSort = byCost && kSettle > 0 && costSorts == kSettle, settled = byCost && kSettle > 0 && costSorts > kSettle;
  bool byGeom = ordered && !samePicture && geomOn && !envFeatures && numObjects > 0;
  // "Light split": a settled picture of the plain table-walk class (no secondary rays, samplers or layers) with several lights is
  // bound by the life of its heaviest waves, and those are whole tiles whose every pixel runs one long shadow march per light back to
  // back (C2: 26-40 evaluations of primary march, then three soft-shadow marches of 256 — profiles/r04_r_c2_chain_sim.txt).  The
  // first tileCount / kSplitDiv tiles of the settled order are therefore rendered by numLights workgroups each — every one repeats
  // the primary march and the surface point and marches ONE light, its result going to memory (the first one's primary result too)
  // — and the last of them to arrive finishes the tile from the stored results: surface point, AO and the light sum, no march.  The
  // same marches, the same sums in the same order: the same pixels.  Whether it pays is measured per picture (SplitTune above).
  // RM_LIGHT_SPLIT=0: off; =n: the heaviest 1/n of the tiles.
  static const int envSplitDiv = [] { const char *e = std::getenv("RM_LIGHT_SPLIT"); const int v = e ? std::atoi(e) : 256; return v < 0 ? 0 : v; }();
  const int kSplitDiv = g_lightSplit.load() >= 0 ? g_lightSplit.load() : envSplitDiv;  // rm_debug_set_light_split
  int splitK = 0, splitTimed = -1;  // splitTimed: 0 / 1 = time this launch as the tuner's plain / split candidate
  float *splitStore = nullptr;
  SplitTune *splitTune = nullptr;
  if (settled && kSplitDiv > 0 && !bulb && !envFeatures && !textured && !secondary && count == 0 && nw == 1 && numLights >= 2 &&
      numLights <= RM_MAX_LIGHTS && tuneTimed < 0) {
    splitK = tileCount / kSplitDiv;
    if (splitK > 0 && !g_lightSplitForce.load()) {  // measured, unless a test forces it (rm_debug_set_light_split)
      splitTune = &ds.splitTune[stream];
      SplitTune &tn = *splitTune;
      if (tn.key != key || tn.W != W || tn.nRows != nRows || tn.tileShift != tileShift || tn.div != kSplitDiv) {
        tn.drop();
        tn.key = key; tn.W = W; tn.nRows = nRows; tn.tileShift = tileShift; tn.div = kSplitDiv; tn.frame = 0; tn.chosen = -1;
        const auto known = ds.splitChoice.find(std::make_tuple(key, W, nRows, tileShift, kSplitDiv));
        if (known != ds.splitChoice.end()) tn.chosen = known->second;
      }
      if (tn.chosen < 0 && tn.frame >= 4 && tn.timed[0] && tn.timed[1] && hipEventQuery(tn.ev[0][1]) == hipSuccess &&
          hipEventQuery(tn.ev[1][1]) == hipSuccess) {
        float plainMs = 0.0f, splitMs = 0.0f;
        const bool ok = hipEventElapsedTime(&plainMs, tn.ev[0][0], tn.ev[0][1]) == hipSuccess &&
                        hipEventElapsedTime(&splitMs, tn.ev[1][0], tn.ev[1][1]) == hipSuccess;
        tn.chosen = (ok && splitMs < 0.97f * plainMs) ? 1 : 0;
        tn.drop();
        if (ds.splitChoice.size() >= 256) ds.splitChoice.clear();
        ds.splitChoice[std::make_tuple(key, W, nRows, tileShift, kSplitDiv)] = tn.chosen;
      }
      (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is not an error
      bool useSplit = tn.chosen == 1;
      if (tn.chosen < 0) {
        const int f = tn.frame;
        useSplit = f == 2 || f == 3;
        if (f == 1 || f == 3) splitTimed = f >> 1;
      }
      tn.frame++;
      if (!useSplit) splitK = 0;
    }
    if (splitK > 0) {
      void *mem = nullptr;
      if (stream_workspace(kWsLightSplit, stream, ((((size_t)splitK + 63) & ~(size_t)63) + (size_t)splitK * 64 * (2 * numLights + 6)) * sizeof(float), &mem) == RM_OK) splitStore = static_cast<float *>(mem);
      else splitK = 0;  // no memory for it: the plain launch
    }
  }
  Slot *slot;
  st = stage_scene(cam, objs, numObjects, lights, numLights, g, s, stream, ds, &slot, res,
                   ordered ? ((byCost || byGeom) ? oOrder : nullptr) : ds.dbgTileOrder,
                   ordered ? ((lastSort || settled) ? nullptr : oCost) : ds.dbgTileCost,
                   ordered ? tileCount : ds.dbgTileCount, tileShift, splitK, splitStore);
  if (st != RM_OK) return st;
  if (byGeom && !slot->host->objBallOk) {  // an object without a bounding ball (Sierpinski, 2-D Mandelbrot as an object): raster order
    byGeom = false;
    slot->host->tileOrder = nullptr;
    HIP_OK(hipMemcpyAsync(&slot->dev->tileOrder, &slot->host->tileOrder, sizeof(slot->host->tileOrder), hipMemcpyHostToDevice, stream));
  }
  unsigned long long *dc = ds.dCounters;
  if (count) {
    HIP_OK(hipMemsetAsync(dc, 0, 10 * sizeof(unsigned long long), stream));
    if (d_waveSpans) HIP_OK(hipMemcpyAsync(dc + 5, &d_waveSpans, sizeof(d_waveSpans), hipMemcpyHostToDevice, stream));
  }
  TimedLaunch tl{};
  const bool timing = g_timing.load();
  // the events of a launch that fails part-way are destroyed on the way out (kept = handed to ds.timed below)
  struct TimedGuard {
    TimedLaunch &t; bool kept = false;
    ~TimedGuard() { if (!kept) for (int i = 0; i < t.n; i++) (void)hipEventDestroy(t.ev[i]); }
  } timedGuard{tl};
  auto stamp = [&](int i) -> int {
    if (!timing) return RM_OK;
    HIP_OK(hipEventCreate(&tl.ev[i]));
    tl.n = i + 1;
    HIP_OK(hipEventRecord(tl.ev[i], stream));
    return RM_OK;
  };
  float4 *o = reinterpret_cast<float4 *>(d_rgba), *b = reinterpret_cast<float4 *>(d_bright);
  if (wavefront) {
    const WfWs &ws = wfWs;
    HIP_OK(hipMemsetAsync(ws.counters, 0, WF_STRIDE * (kWfMaxBounces + 2) * sizeof(uint32_t), stream));
    const dim3 pgrid(wfPrimaryWaves), mgrid(wfShadowWaves), mblock(64), dense(ds.numCUs * 16);
    const int thr = envFlush > 0 && envFlush <= 64 ? envFlush : 16;
    if ((st = stamp(0)) != RM_OK) return st;
    for (int gen = 0; gen <= wfBounces; gen++) {
      if (wfSkip) {
        if (gen == 0) hipLaunchKernelGGL((wf_march_kernel<0, true>), pgrid, mblock, 0, stream, slot->dev, map, W, H, nRows, o, b, ws, gen, thr, pixelChunk, maxChunk, slotChunk);
        else hipLaunchKernelGGL((wf_march_kernel<1, true>), pgrid, mblock, 0, stream, slot->dev, map, W, H, nRows, o, b, ws, gen, thr, rayChunk, maxChunk, slotChunk);
        hipLaunchKernelGGL(wf_surface_kernel<true>, dense, block, 0, stream, slot->dev, map, W, H, ws, gen);
        if (numLights > 0) hipLaunchKernelGGL((wf_march_kernel<2, true>), mgrid, mblock, 0, stream, slot->dev, map, W, H, nRows, o, b, ws, gen, thr, rayChunk, maxChunk, slotChunk);
      } else {
        if (gen == 0) hipLaunchKernelGGL((wf_march_kernel<0, false>), pgrid, mblock, 0, stream, slot->dev, map, W, H, nRows, o, b, ws, gen, thr, pixelChunk, maxChunk, slotChunk);
        else hipLaunchKernelGGL((wf_march_kernel<1, false>), pgrid, mblock, 0, stream, slot->dev, map, W, H, nRows, o, b, ws, gen, thr, rayChunk, maxChunk, slotChunk);
        hipLaunchKernelGGL(wf_surface_kernel<false>, dense, block, 0, stream, slot->dev, map, W, H, ws, gen);
        if (numLights > 0) hipLaunchKernelGGL((wf_march_kernel<2, false>), mgrid, mblock, 0, stream, slot->dev, map, W, H, nRows, o, b, ws, gen, thr, rayChunk, maxChunk, slotChunk);
      }
      hipLaunchKernelGGL(wf_light_kernel, dense, block, 0, stream, slot->dev, map, W, H, o, b, ws, gen, wfBounces);
    }
    if ((st = stamp(1)) != RM_OK) return st;
  } else {
    if ((st = stamp(0)) != RM_OK) return st;
    // instantiations <BULB, COUNT, ENV, TEX>: the bulb class and the generic table walk, plain and counted, without
    // procedural layers or textures; the generic kernel with either or both.  Features a launch does not need are
    // compiled out so the common kernels keep their register budget.
    if ((byCost || byGeom) && !settled) {  // this frame's launch order — from the previous frame's tile costs or from geometry — ahead of the render
      const dim3 sgrid((tileCount + 255) / 256);
      uint32_t *sortCost = oCost;
      if (byGeom) {  // estimates into their own array (the kernel reads the stale costs of a tile's neighbourhood), stale costs cleared after
        hipLaunchKernelGGL(tile_geom_kernel, sgrid, dim3(256), 0, stream, slot->dev, map, W, H, nRows, (int)rgrid.x, nw * tileW, tileH, tileCount, oCost, oCost2,
                           (geomMode == 2 && haveCost) ? 1 : 0, ((geomMode == 2 && haveCost) ? ringCombined : 16), geomDilate);
        HIP_OK(hipMemsetAsync(oCost, 0, (size_t)tileCount * sizeof(uint32_t), stream));
        sortCost = oCost2;
      }
      HIP_OK(hipMemsetAsync(oHist, 0, 2 * kOrderBuckets * sizeof(uint32_t), stream));
      hipLaunchKernelGGL(tile_hist_kernel, sgrid, dim3(256), 0, stream, sortCost, tileCount, oHist);
      hipLaunchKernelGGL(tile_scatter_kernel, sgrid, dim3(256), 0, stream, sortCost, tileCount, oHist, oOrder, lastSort ? 1 : 0);
      if ((st = stamp(1)) != RM_OK) return st;  // stage 0 = the ordering launches, stage 1 = the render
    }
    if (splitTimed >= 0 && splitTune) {
      if (hipEventCreate(&splitTune->ev[splitTimed][0]) == hipSuccess && hipEventCreate(&splitTune->ev[splitTimed][1]) == hipSuccess)
        HIP_OK(hipEventRecord(splitTune->ev[splitTimed][0], stream));
      else splitTimed = -1;
    }
    if (tuneTimed >= 0 && tune) {  // the tuner's timed launch of this candidate shape: events around the render kernel alone
      if (hipEventCreate(&tune->ev[tuneTimed][0]) == hipSuccess && hipEventCreate(&tune->ev[tuneTimed][1]) == hipSuccess)
        HIP_OK(hipEventRecord(tune->ev[tuneTimed][0], stream));
      else tuneTimed = -1;
    }
#define RM_LAUNCH(B, C, E, T) hipLaunchKernelGGL((render_kernel<B, C, E, T>), rgrid, rblock, 0, stream, slot->dev, map, W, H, nRows, o, b, dc)
    if (envFeatures || textured) {
      // counting instantiations of the layer / sampler kernels: the reference's work only (they have no shortcuts to count apart)
#define RM_LAUNCH_NOSEC(B, E, T) hipLaunchKernelGGL((render_kernel<B, 0, E, T, false>), rgrid, rblock, 0, stream, slot->dev, map, W, H, nRows, o, b, dc)
      if (envFeatures && textured) { if (count) RM_LAUNCH(false, 1, true, true); else if (secondary) RM_LAUNCH(false, 0, true, true); else RM_LAUNCH_NOSEC(false, true, true); }
      else if (envFeatures) { if (count) RM_LAUNCH(false, 1, true, false); else if (secondary) RM_LAUNCH(false, 0, true, false); else RM_LAUNCH_NOSEC(false, true, false); }
      else { if (count) RM_LAUNCH(false, 1, false, true); else if (secondary) RM_LAUNCH(false, 0, false, true); else RM_LAUNCH_NOSEC(false, false, true); }
    } else if (bulb) {
      if (count == 1) RM_LAUNCH(true, 1, false, false);
      else if (count == 2) RM_LAUNCH(true, 2, false, false);
      else if (count == 3) RM_LAUNCH(true, 3, false, false);
      else if (secondary) RM_LAUNCH(true, 0, false, false);
      else RM_LAUNCH_NOSEC(true, false, false);
    } else {
      if (count == 1) RM_LAUNCH(false, 1, false, false);
      else if (count == 2) RM_LAUNCH(false, 2, false, false);
      else if (count == 3) RM_LAUNCH(false, 3, false, false);
      else if (secondary) RM_LAUNCH(false, 0, false, false);
      else if (splitK > 0) {
        // light split: the heavy tiles one light per workgroup first, every other tile behind them in the same grid; the last of a
        // tile's workgroups to finish its march finishes the tile.  (A second launch for the finish cost 35-45 µs per frame —
        // more than the split gains on throughput-bound frames; the same launch on a side stream gained nothing.)
        HIP_OK(hipMemsetAsync(splitStore, 0, (size_t)splitK * sizeof(uint32_t), stream));  // the tiles' arrival counters
        hipLaunchKernelGGL((render_kernel<false, 0, false, false, false, 1>), dim3((unsigned)(splitK * numLights + tileCount - splitK)), dim3(64), 0, stream,
                           slot->dev, map, W, H, nRows, o, b, dc);
      } else RM_LAUNCH_NOSEC(false, false, false);
    }
#undef RM_LAUNCH
#undef RM_LAUNCH_NOSEC
    if (tuneTimed >= 0 && tune) {
      HIP_OK(hipEventRecord(tune->ev[tuneTimed][1], stream));
      tune->timed[tuneTimed] = true;
    }
    if (splitTimed >= 0 && splitTune) {
      HIP_OK(hipEventRecord(splitTune->ev[splitTimed][1], stream));
      splitTune->timed[splitTimed] = true;
    }
    if ((st = stamp(((byCost || byGeom) && !settled) ? 2 : 1)) != RM_OK) return st;
  }
  HIP_OK(hipGetLastError());
  ds.lastPath = wavefront ? 5 : 1;
  ds.lastSplit = wavefront ? 0 : splitK;
  if (timing) { ds.timed.push_back(tl); timedGuard.kept = true; }
  HIP_OK(hipEventRecord(slot->done, stream));
  if (count) {
    unsigned long long hc[10];
    HIP_OK(hipMemcpyAsync(hc, dc, sizeof(hc), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (countersOut) {
      countersOut->sceneEvals = hc[0]; countersOut->bulbIters = hc[1]; countersOut->hitPixels = hc[2];
      countersOut->shadedPoints = hc[6]; countersOut->terrainEvals = hc[7]; countersOut->cloudEvals = hc[8];
      countersOut->shapeEvals = hc[9];
    }
    if (clockMHz) *clockMHz = hc[4] ? 100.0 * (double)hc[3] / (double)hc[4] : 0.0;
  }
  return RM_OK;
}
// the device the calling thread has current
int current_device_state(DeviceState **out) {
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { set_error("device index out of range"); return RM_ERR_DEVICE; }
  *out = &g_dev[dev];
  return RM_OK;
}
}  // namespace
}  // namespace rm

using namespace rm;

extern "C" {

int rm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}
int rm_set_device(int device) {
  HIP_OK(hipSetDevice(device));
  return RM_OK;
}

int rm_render(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
              const RmGlobals *g, const RmSettings *s, int W, int H, int rowBegin, int rowEnd, float *d_rgba,
              float *d_bright, void *stream) {
  if (rowBegin < 0 || rowEnd > H || rowBegin > rowEnd) { set_error("rows out of range"); return RM_ERR_INVALID_ARGUMENT; }
  int n = rowEnd - rowBegin;
  RowMap map{rowBegin, n > 0 ? n : 1, 0, 1, 0};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, n, d_rgba, d_bright,
                       static_cast<hipStream_t>(stream), 0, nullptr);
}

int rm_render_ex(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                 const RmGlobals *g, const RmSettings *s, const RmTexture *textures, int numTextures, int W, int H,
                 int rowBegin, int rowEnd, float *d_rgba, float *d_bright, void *stream) {
  if (rowBegin < 0 || rowEnd > H || rowBegin > rowEnd) { set_error("rows out of range"); return RM_ERR_INVALID_ARGUMENT; }
  int n = rowEnd - rowBegin;
  RowMap map{rowBegin, n > 0 ? n : 1, 0, 1, 0};
  RmResources res{};
  res.textures = textures; res.numTextures = numTextures;
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, n, d_rgba, d_bright,
                       static_cast<hipStream_t>(stream), 0, nullptr, res);
}

int rm_render_res(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                  const RmGlobals *g, const RmSettings *s, const RmResources *res, int W, int H, int rowBegin, int rowEnd,
                  float *d_rgba, float *d_bright, void *stream) {
  if (rowBegin < 0 || rowEnd > H || rowBegin > rowEnd) { set_error("rows out of range"); return RM_ERR_INVALID_ARGUMENT; }
  int n = rowEnd - rowBegin;
  RowMap map{rowBegin, n > 0 ? n : 1, 0, 1, 0};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, n, d_rgba, d_bright,
                       static_cast<hipStream_t>(stream), 0, nullptr, res ? *res : kNoResources);
}

int rm_render_counted_ex(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                         const RmGlobals *g, const RmSettings *s, int W, int H, int rowBegin, int rowEnd, float *d_rgba,
                         float *d_bright, int mode, RmCounters *out) {
  if (rowBegin < 0 || rowEnd > H || rowBegin > rowEnd) { set_error("rows out of range"); return RM_ERR_INVALID_ARGUMENT; }
  if (mode != RM_COUNT_REFERENCE && mode != RM_COUNT_EXECUTED) { set_error("bad counting mode"); return RM_ERR_INVALID_ARGUMENT; }
  int n = rowEnd - rowBegin;
  RowMap map{rowBegin, n > 0 ? n : 1, 0, 1, 0};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, n, d_rgba, d_bright, nullptr, mode, out);
}
int rm_render_counted_res(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                          const RmGlobals *g, const RmSettings *s, const RmResources *res, int W, int H, int rowBegin, int rowEnd,
                          float *d_rgba, float *d_bright, int mode, RmCounters *out) {
  if (rowBegin < 0 || rowEnd > H || rowBegin > rowEnd) { set_error("rows out of range"); return RM_ERR_INVALID_ARGUMENT; }
  if (mode != RM_COUNT_REFERENCE && mode != RM_COUNT_EXECUTED) { set_error("bad counting mode"); return RM_ERR_INVALID_ARGUMENT; }
  int n = rowEnd - rowBegin;
  RowMap map{rowBegin, n > 0 ? n : 1, 0, 1, 0};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, n, d_rgba, d_bright, nullptr, mode, out,
                       res ? *res : kNoResources);
}
int rm_render_counted(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                      const RmGlobals *g, const RmSettings *s, int W, int H, int rowBegin, int rowEnd, float *d_rgba,
                      float *d_bright, RmCounters *out) {
  return rm_render_counted_ex(cam, objs, numObjects, lights, numLights, g, s, W, H, rowBegin, rowEnd, d_rgba, d_bright,
                              RM_COUNT_REFERENCE, out);
}
int rm_render_clocked(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                      const RmGlobals *g, const RmSettings *s, int W, int H, float *d_rgba, double *shaderMHz,
                      unsigned long long *d_waveSpans) {
  if (!shaderMHz) { set_error("null shaderMHz"); return RM_ERR_INVALID_ARGUMENT; }
  if (d_waveSpans && !device_accessible(d_waveSpans)) { set_error("d_waveSpans is not device-accessible memory"); return RM_ERR_INVALID_ARGUMENT; }
  // the stamped build exists for the single-Mandelbulb class and for the plain table walk (no samplers, no procedural layers)
  bool plain = objs != nullptr && s != nullptr && numObjects >= 1 &&
               (s->features & (RM_FEAT_TERRAIN | RM_FEAT_CLOUD | RM_FEAT_SEA | RM_FEAT_SKY_BACKGROUND | RM_FEAT_NIGHTSKY_BACKGROUND)) == 0 && !s->enableSkyBox;
  for (int i = 0; plain && i < numObjects; i++) plain = objs[i].texLoc < 0 && !objs[i].isEmissive;
  for (int i = 0; plain && i < numLights; i++) plain = lights[i].type != RM_LIGHT_AREA;
  if (!plain) {
    set_error("rm_render_clocked covers the single-Mandelbulb class and the plain table walk (no samplers, no procedural layers)");
    return RM_ERR_UNSUPPORTED;
  }
  RowMap map{0, H > 0 ? H : 1, 0, 1, 0};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, H, d_rgba, nullptr, nullptr, 3, nullptr,
                       kNoResources, shaderMHz, d_waveSpans);
}

int rm_render_tiles(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                    const RmGlobals *g, const RmSettings *s, int W, int H, int tileRows, int shard, int numShards,
                    float *d_rgba, float *d_bright, void *stream) {
  if (tileRows <= 0 || numShards <= 0 || shard < 0 || shard >= numShards) {
    set_error("bad tile partition");
    return RM_ERR_INVALID_ARGUMENT;
  }
  RowMap map{0, tileRows, shard, numShards, root_relief()};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, shard_rows(H, tileRows, shard, numShards, root_relief()),
                       d_rgba, d_bright, static_cast<hipStream_t>(stream), 0, nullptr);
}

int rm_render_tiles_res(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                        const RmGlobals *g, const RmSettings *s, const RmResources *res, int W, int H, int tileRows,
                        int shard, int numShards, float *d_rgba, float *d_bright, void *stream) {
  if (tileRows <= 0 || numShards <= 0 || shard < 0 || shard >= numShards) {
    set_error("bad tile partition");
    return RM_ERR_INVALID_ARGUMENT;
  }
  RowMap map{0, tileRows, shard, numShards, root_relief()};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, shard_rows(H, tileRows, shard, numShards, root_relief()),
                       d_rgba, d_bright, static_cast<hipStream_t>(stream), 0, nullptr, res ? *res : kNoResources);
}

int rm_deinterleave(const float *d_gathered, float *d_frame, int W, int H, int tileRows, int numShards,
                    int shardStrideRows, void *stream) {
  if (!d_gathered || !d_frame || W <= 0 || H <= 0 || tileRows <= 0 || numShards <= 0 || numShards > 64 ||
      (shardStrideRows != 0 && shardStrideRows < max_shard_rows(H, tileRows, numShards, root_relief()))) {
    set_error("bad deinterleave arguments");
    return RM_ERR_INVALID_ARGUMENT;
  }
  if (int st = require_device_pointers({{"d_gathered", d_gathered}, {"d_frame", d_frame}})) return st;
  dim3 grid((W + 255) / 256, H), block(256);
  hipLaunchKernelGGL(deinterleave_kernel, grid, block, 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4 *>(d_gathered), reinterpret_cast<float4 *>(d_frame), W, H, tileRows,
                     numShards, shardStrideRows, root_relief());
  HIP_OK(hipGetLastError());
  return RM_OK;
}

int rm_tiles_to_rgba8(const float *d_tiles, uint8_t *d_tiles8, int W, int rows, void *stream) {
  if (W <= 0 || rows < 0) { set_error("bad tile arguments"); return RM_ERR_INVALID_ARGUMENT; }
  if (rows == 0) return RM_OK;
  if (!d_tiles || !d_tiles8) { set_error("null tile buffer"); return RM_ERR_INVALID_ARGUMENT; }
  if (int st = require_device_pointers({{"d_tiles", d_tiles}, {"d_tiles8", d_tiles8}})) return st;
  const size_t n = (size_t)rows * W;
  hipLaunchKernelGGL(tiles_to_rgba8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4 *>(d_tiles), reinterpret_cast<uchar4 *>(d_tiles8), n);
  HIP_OK(hipGetLastError());
  return RM_OK;
}
int rm_deinterleave_rgba8(const uint8_t *d_gathered8, uint8_t *d_frame8, int W, int H, int tileRows, int numShards,
                          int shardStrideRows, int flip, void *stream) {
  if (!d_gathered8 || !d_frame8 || W <= 0 || H <= 0 || tileRows <= 0 || numShards <= 0 || numShards > 64 ||
      (shardStrideRows != 0 && shardStrideRows < max_shard_rows(H, tileRows, numShards, root_relief()))) {
    set_error("bad deinterleave arguments");
    return RM_ERR_INVALID_ARGUMENT;
  }
  if (int st = require_device_pointers({{"d_gathered8", d_gathered8}, {"d_frame8", d_frame8}})) return st;
  dim3 grid((W + 255) / 256, H), block(256);
  hipLaunchKernelGGL(deinterleave_rgba8_kernel, grid, block, 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const uchar4 *>(d_gathered8), reinterpret_cast<uchar4 *>(d_frame8), W, H, tileRows, numShards,
                     shardStrideRows, flip, root_relief());
  HIP_OK(hipGetLastError());
  return RM_OK;
}

int rm_frame_to_rgba8(const float *d_rgba, uint8_t *d_out, int W, int H, void *stream) {
  if (!d_rgba || !d_out || W <= 0 || H <= 0) { set_error("bad frame arguments"); return RM_ERR_INVALID_ARGUMENT; }
  if (int st = require_device_pointers({{"d_rgba", d_rgba}, {"d_out", d_out}})) return st;
  dim3 grid((W + 255) / 256, H), block(256);
  hipLaunchKernelGGL(to_rgba8_kernel, grid, block, 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4 *>(d_rgba), reinterpret_cast<uchar4 *>(d_out), W, H);
  HIP_OK(hipGetLastError());
  return RM_OK;
}

int rm_set_timing(int on) {
  g_timing.store(on != 0);
  DeviceState *ds;
  if (int st = current_device_state(&ds)) return st;
  std::lock_guard<std::mutex> lock(ds->mu);
  for (auto &t : ds->timed)
    for (int i = 0; i < t.n; i++) (void)hipEventDestroy(t.ev[i]);
  ds->timed.clear();
  return RM_OK;
}
int rm_get_timing(double *avgKernelMs, int *launches) {
  double stages[4];
  return rm_get_stage_timing(avgKernelMs, stages, launches);
}
int rm_get_stage_timing(double *avgTotalMs, double avgStageMs[4], int *launches) {
  DeviceState *ds;
  if (int st = current_device_state(&ds)) return st;
  std::vector<TimedLaunch> timed;
  {
    std::lock_guard<std::mutex> lock(ds->mu);
    timed.swap(ds->timed);  // waiting for the events happens outside the device lock
  }
  double total = 0.0, stage[4] = {0, 0, 0, 0};
  int rc = RM_OK;
  for (auto &t : timed) {
    if (t.n < 2 || rc != RM_OK) continue;
    float ms = 0.0f;
    if (hipEventSynchronize(t.ev[t.n - 1]) != hipSuccess || hipEventElapsedTime(&ms, t.ev[0], t.ev[t.n - 1]) != hipSuccess) {
      set_error("timing events could not be read");
      rc = RM_ERR_DEVICE;
      continue;
    }
    total += ms;
    // by role, whatever the launch was made of: the last interval is the render (one kernel or the wavefront pipeline's), the one
    // before it — present only in a launch that sorted its tiles — the ordering launches
    if (hipEventElapsedTime(&ms, t.ev[t.n - 2], t.ev[t.n - 1]) == hipSuccess) stage[1] += ms;
    if (t.n >= 3 && hipEventElapsedTime(&ms, t.ev[0], t.ev[t.n - 2]) == hipSuccess) stage[0] += ms;
  }
  const double n = timed.empty() ? 1.0 : (double)timed.size();
  if (launches) *launches = (int)timed.size();
  if (avgTotalMs) *avgTotalMs = total / n;
  if (avgStageMs) for (int i = 0; i < 4; i++) avgStageMs[i] = stage[i] / n;
  for (auto &t : timed)
    for (int i = 0; i < t.n; i++) (void)hipEventDestroy(t.ev[i]);
  return rc;
}
// The cheap exact forms against the IEEE operations for EVERY binary32 input (NaN = NaN): out[0] = inputs where rcp_(y) !=
// 1.0f / y, out[1] = inputs of the fast range 2^-126 <= |y| < 2^126 where the bare v_rcp_f32 + Newton form differs, out[2] =
// inputs where sqrt_fast_(x) != sqrtf(x), out[3] = inputs of sqrt_noscale_'s domain (±0, |x| >= 2^-96, ±inf, NaN)
// where it differs from sqrtf(x), out[4] = inputs where fract_(x) (v_fract_f32) != x − floor(x) kept below 1.  All must be 0.
__global__ void check_math_kernel(unsigned long long *out) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  unsigned long long bad = 0, badFast = 0, badSqrt = 0, badNoscale = 0, badFract = 0;
  for (uint64_t u = tid; u < (1ull << 32); u += stride) {
    const float y = u2f((uint32_t)u), ref = 1.0f / y, got = rcp_(y);
    if (f2u(got) != f2u(ref) && !(got != got && ref != ref)) bad++;
    const float ay = fabs_(y);
    if (ay >= 1.17549435e-38f && ay < 8.50705917e37f) {
      const float r = __builtin_amdgcn_rcpf(y), f = rm::fma(rm::fma(-y, r, 1.0f), r, r);
      if (f2u(f) != f2u(ref)) badFast++;
    }
    const float fd = y - __builtin_floorf(y), fref = (fd >= 1.0f) ? 0.99999994f : fd, fg = fract_(y);
    if (f2u(fg) != f2u(fref) && !(fg != fg && fref != fref)) badFract++;
    const float sref = sqrt_(y), sf = sqrt_fast_(y);
    if (f2u(sf) != f2u(sref) && !(sf != sf && sref != sref)) badSqrt++;
    if (!(ay > 0.0f && ay < 1.262177448e-29f)) {
      const float sn = sqrt_noscale_(y);
      if (f2u(sn) != f2u(sref) && !(sn != sn && sref != sref)) badNoscale++;

    }
  }
  if (bad) atomicAdd(&out[0], bad);
  if (badFast) atomicAdd(&out[1], badFast);
  if (badSqrt) atomicAdd(&out[2], badSqrt);
  if (badNoscale) atomicAdd(&out[3], badNoscale);
  if (badFract) atomicAdd(&out[4], badFract);
}
int rm_debug_check_math(unsigned long long *mismatches5) {
  if (!mismatches5) { set_error("null pointer"); return RM_ERR_INVALID_ARGUMENT; }
  unsigned long long *d = nullptr;
  HIP_OK(hipMalloc(reinterpret_cast<void **>(&d), 5 * sizeof(unsigned long long)));
  hipError_t e = hipMemset(d, 0, 5 * sizeof(unsigned long long));
  if (e == hipSuccess) {
    check_math_kernel<<<4096, 256>>>(d);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(mismatches5, d, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { set_error(std::string("rm_debug_check_math: ") + hipGetErrorString(e)); return RM_ERR_DEVICE; }
  return RM_OK;
}
int rm_debug_ray_planes(const RmCamera *cam, float *out48) {
  if (!cam || !out48) { set_error("null pointer"); return RM_ERR_INVALID_ARGUMENT; }
  static SceneBlock blk;  // host-only scratch; the planes are a pure function of the camera
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  blk.cam = *cam;
  ray_planes(&blk);
  std::memcpy(out48, blk.rayPlane, sizeof(blk.rayPlane));
  return RM_OK;
}
int rm_debug_cull_bounds(const RmObject *objs, int numObjects, const RmGlobals *g, float *out14) {
  if ((!objs && numObjects > 0) || !g || !out14) { set_error("null pointer"); return RM_ERR_INVALID_ARGUMENT; }
  if (numObjects < 0 || numObjects > RM_MAX_OBJECTS) { set_error("numObjects out of range"); return RM_ERR_INVALID_ARGUMENT; }
  static SceneBlock blk;  // host-only scratch; the bounds are a pure function of the object table and the globals
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  blk.g = *g;
  blk.numObjects = numObjects;
  for (int i = 0; i < numObjects; i++) blk.objs[i] = objs[i];
  scene_cull_ball(&blk);
  out14[0] = (float)blk.cullOk;
  for (int k = 0; k < 3; k++) { out14[1 + k] = blk.cullC[k]; out14[7 + k] = blk.cullLo[k]; out14[10 + k] = blk.cullHi[k]; }
  out14[4] = blk.cullR2; out14[5] = blk.cullR2Soft; out14[6] = (float)blk.cullBoxOk;
  out14[13] = blk.cullLip;
  return RM_OK;
}
int rm_debug_set_tile_order(const int32_t *d_order, uint32_t *d_cost, int tileCount) {
  DeviceState *ds;
  if (int st = current_device_state(&ds)) return st;
  std::lock_guard<std::mutex> lock(ds->mu);
  ds->dbgTileOrder = d_order; ds->dbgTileCost = d_cost; ds->dbgTileCount = tileCount;
  return RM_OK;
}
int rm_set_tile_order(int mode) {
  if (mode < -1 || mode > 1) { set_error("tile order mode must be -1, 0 or 1"); return RM_ERR_INVALID_ARGUMENT; }
  g_tileOrderMode.store(mode);
  return RM_OK;
}
int rm_debug_last_path(void) {
  DeviceState *ds;
  if (current_device_state(&ds)) return -1;
  std::lock_guard<std::mutex> lock(ds->mu);
  return ds->lastPath;
}
int rm_debug_set_light_split(int div) {
  if (div < -1) { set_error("light split: -1 (default), 0 (off) or the divisor n >= 1"); return RM_ERR_INVALID_ARGUMENT; }
  g_lightSplit.store(div);
  g_lightSplitForce.store(div > 0);  // an explicit divisor splits without measuring (tests); -1 / the environment variable: measured
  return RM_OK;
}
int rm_debug_last_split(void) {
  DeviceState *ds;
  if (current_device_state(&ds) != RM_OK) return -1;
  std::lock_guard<std::mutex> lock(ds->mu);
  return ds->lastSplit;
}
int rm_set_kernel_path(int path) {
  if (path != 0 && path != 1 && path != 5) { set_error("kernel path must be 0, 1 or 5 (2-4, the bulb pipelines, were removed in round 4)"); return RM_ERR_INVALID_ARGUMENT; }
  g_kernelPath.store(path);
  return RM_OK;
}
int rm_debug_set_tile_shape(int mode) {
  if (mode != -1 && mode != 0 && mode != 2 && mode != 3) { set_error("tile shape mode must be -1, 0, 2 or 3"); return RM_ERR_INVALID_ARGUMENT; }
  g_tileShape.store(mode);
  return RM_OK;
}
int rm_set_workspace_limit(unsigned long long bytes) {
  g_wsLimit.store(bytes == ~0ull ? ~0ull - 1 : bytes);
  for (DeviceState &ds : g_dev) {  // what was refused under the old limit may be asked for again
    std::lock_guard<std::mutex> lock(ds.mu);
    ds.wfDenied.clear();
  }
  return RM_OK;
}
int rm_release_workspaces(unsigned long long *freedBytes) {
  DeviceState *ds;
  if (int st = current_device_state(&ds)) return st;
  std::lock_guard<std::mutex> lock(ds->mu);  // no launch is being enqueued on this device meanwhile
  size_t freed = 0;
  if (int st = release_workspaces(&freed)) return st;
  ds->tileOrder.clear();  // the feedback costs lived in the buffers just freed
  for (auto &kv : ds->shapeTune) kv.second.drop();
  ds->shapeTune.clear();
  ds->shapeChoice.clear();
  for (auto &kv : ds->splitTune) kv.second.drop();
  ds->splitTune.clear();
  ds->splitChoice.clear();
  ds->wfDenied.clear();
  if (freedBytes) *freedBytes = freed;
  return RM_OK;
}

int rm_probe_math(int fn, const float *d_x, const float *d_y, const float *d_z, float *d_out, int n, void *stream) {
  if (fn < 0 || fn >= RM_FN_COUNT || !d_x || !d_out || n < 0) { set_error("bad probe arguments"); return RM_ERR_INVALID_ARGUMENT; }
  if (int st = require_device_pointers({{"d_x", d_x}, {"d_y", d_y}, {"d_z", d_z}, {"d_out", d_out}})) return st;
  if (n == 0) return RM_OK;
  hipLaunchKernelGGL(probe_math_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), fn, d_x,
                     d_y, d_z, d_out, n);
  HIP_OK(hipGetLastError());
  return RM_OK;
}

int rm_probe_sdscene(const RmObject *objs, int numObjects, const RmGlobals *g, const RmSettings *s, const float *d_pts,
                     float *d_out, int n, void *stream) {
  RmCamera cam{};
  int st = validate_scene(&cam, objs, numObjects, nullptr, 0, g, s, kNoResources);
  if (st != RM_OK) return st;
  if (!d_pts || !d_out || n < 0) { set_error("bad probe arguments"); return RM_ERR_INVALID_ARGUMENT; }
  if (int st2 = require_device_pointers({{"d_pts", d_pts}, {"d_out", d_out}})) return st2;
  if (n == 0) return RM_OK;
  DeviceState *ds;
  if ((st = current_device_state(&ds)) != RM_OK) return st;
  std::lock_guard<std::mutex> lock(ds->mu);
  Slot *slot;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  st = stage_scene(&cam, objs, numObjects, nullptr, 0, g, s, hs, *ds, &slot, kNoResources);
  if (st != RM_OK) return st;
  hipLaunchKernelGGL(probe_sdscene_kernel, dim3((n + 255) / 256), dim3(256), 0, hs, slot->dev, d_pts, d_out, n);
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(slot->done, hs));
  return RM_OK;
}

}  // extern "C"
