// rm_kernels.hip — HIP kernels and the C-ABI launcher of the per-pixel raymarch (gfx950 only).
//
// Replaces Realtime::rayMarch() of the reference (src/realtimerender.cpp:53-87): instead of uploading
// ~600 uniforms by name and drawing a full-screen quad through resources/raymarch.{vert,frag}, the
// launcher copies one constant SceneBlock to the device and launches one lane per pixel.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "rm_bulb_pipeline.hip.h"
#include "rm_device.hip.h"
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
constexpr int kTileW = RM_TILE_W, kTileH = 64 / RM_TILE_W;
constexpr int kBlockW = 4 * kTileW, kBlockH = kTileH;

template <bool BULB, bool COUNT, bool ENV, bool TEX>
__global__ __launch_bounds__(256) void render_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                      int nRows, float4 *__restrict__ out,
                                                      float4 *__restrict__ bright,
                                                      unsigned long long *__restrict__ counters) {
  // LDS copy of the object table for per-lane (divergent) material lookups.
  __shared__ RmObject s_objs[RM_MAX_OBJECTS];
  {
    const int nd = sb->numObjects * (int)(sizeof(RmObject) / 4);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sb->objs);
    uint32_t *dst = reinterpret_cast<uint32_t *>(s_objs);
    for (int i = threadIdx.x; i < nd; i += 256) dst[i] = src[i];
  }
  // the launches that read samplers build the byte→unorm table (wave-uniform condition; ends with a barrier)
  if (TEX || (ENV && (sb->s.features & (RM_FEAT_NIGHTSKY_BACKGROUND | RM_FEAT_SEA)))) initUnormTable();
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x = blockIdx.x * kBlockW + wave * kTileW + (lane % kTileW);
  const int r = blockIdx.y * kBlockH + (lane / kTileW);
  if (x >= W || r >= nRows) return;
  const int y = map.frameRow(r);
  V4 col, br;
  Counters cnt{0, 0};
  bool hit;
  shadePixel<BULB, COUNT, ENV, TEX>(sb, s_objs, x, y, W, H, col, br, cnt, hit);
  const size_t o = (size_t)r * W + x;
  out[o] = make_float4(col.x, col.y, col.z, col.w);
  if (bright) bright[o] = make_float4(br.x, br.y, br.z, br.w);
  if (COUNT) {
    atomicAdd(&counters[0], cnt.evals);
    atomicAdd(&counters[1], cnt.iters);
    if (hit) atomicAdd(&counters[2], 1ull);
  }
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
  }
  out[i] = r;
}

__global__ void probe_sdscene_kernel(const SceneBlock *__restrict__ sb, const float *pts, float *out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Counters cnt{0, 0};
  SceneMin m = sdScene<false, false>(sb, v3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), cnt);
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

// gathered[shard-major packed rows] → frame rows
__global__ void deinterleave_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, int W, int H, int tileRows,
                                    int numShards, int strideRows) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;  // y = frame row
  if (x >= W) return;
  int tile = y / tileRows, shard = tile % numShards;
  // rows owned by shards < shard, plus this shard's rows before frame row y
  int before = shard * strideRows;
  if (strideRows == 0)
    for (int s = 0; s < shard; s++) before += shard_rows(H, tileRows, s, numShards);
  int local = (tile / numShards) * tileRows + (y % tileRows);
  out[(size_t)y * W + x] = in[(size_t)(before + local) * W + x];
}

// ---- launcher state -------------------------------------------------------------------------------------
namespace {
constexpr int kSlots = 8;
constexpr int kAutoBulbPath = 1;  // what rm_set_kernel_path(0) picks for the single-Mandelbulb class (measured best)
struct Slot {
  SceneBlock *host = nullptr;  // pinned
  SceneBlock *dev = nullptr;
  hipEvent_t done = nullptr;
  bool used = false;
};
struct DeviceState {
  Slot slots[kSlots];
  int next = 0;
  unsigned long long *dCounters = nullptr;
  bool init = false;
  // wavefront-pipeline workspace (grow-only; allocated outside any capture, on first use / growth)
  void *wsMem = nullptr;
  size_t wsBytes = 0;
  int numCUs = 0;
};
std::mutex g_mu;
DeviceState g_dev[64];
bool g_timing = false;
int g_kernelPath = 0;  // rm_set_kernel_path: 0 auto, 1 one-lane-per-pixel, 2 pipeline A (state machine), 3 pipeline B (plain loops)
struct TimedLaunch { hipEvent_t ev[5]; int n; };  // n = 2 (single kernel) or 5 (pipeline K1..K4 boundaries)
std::vector<TimedLaunch> g_timed;

#define HIP_OK(expr)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                               \
      return RM_ERR_DEVICE;                                                                       \
    }                                                                                             \
  } while (0)

int acquire_slot(Slot **out) {
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) { set_error("device index out of range"); return RM_ERR_DEVICE; }
  DeviceState &ds = g_dev[dev];
  if (!ds.init) {
    for (auto &s : ds.slots) {
      HIP_OK(hipHostMalloc(reinterpret_cast<void **>(&s.host), sizeof(SceneBlock), hipHostMallocDefault));
      HIP_OK(hipMalloc(reinterpret_cast<void **>(&s.dev), sizeof(SceneBlock)));
      HIP_OK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    HIP_OK(hipMalloc(reinterpret_cast<void **>(&ds.dCounters), 3 * sizeof(unsigned long long)));
    ds.init = true;
  }
  Slot &s = ds.slots[ds.next];
  ds.next = (ds.next + 1) % kSlots;
  if (s.used) HIP_OK(hipEventSynchronize(s.done));  // only blocks with > kSlots launches in flight
  s.used = true;
  *out = &s;
  return RM_OK;
}

// Carve the pipeline workspace for `pixels` pixels and `nl` lights out of the device allocation.
int bulb_workspace(DeviceState &ds, size_t pixels, int nl, hipStream_t stream, BulbWs *ws, BulbWsB *wsB, BulbWsC *wsC,
                   bool withQueues) {
  auto align = [](size_t v) { return (v + 255) & ~size_t(255); };
  const size_t nlq = (size_t)(nl > 0 ? nl : 1);
  const size_t oCnt = 0, oPix = align(256), oRec = oPix + align(pixels * 4), oP = oRec + align(pixels * 16),
               oN = oP + align(pixels * 16), oSh = oN + align(pixels * 16), oRay = oSh + align(pixels * nlq * 8),
               oQ = oRay + align(pixels * nlq * 4),
               qBytes = withQueues ? 2 * (3 * align(pixels * 4) + 4 * align(pixels * nlq * 4)) : 0, total = oQ + qBytes;
  if (total > ds.wsBytes) {
    HIP_OK(hipStreamSynchronize(stream));
    if (ds.wsMem) HIP_OK(hipFree(ds.wsMem));
    ds.wsMem = nullptr; ds.wsBytes = 0;
    HIP_OK(hipMalloc(&ds.wsMem, total));
    ds.wsBytes = total;
  }
  char *b = static_cast<char *>(ds.wsMem);
  ws->counters = reinterpret_cast<uint32_t *>(b + oCnt);
  ws->hitPix = reinterpret_cast<int *>(b + oPix);
  ws->hitRec = reinterpret_cast<float4 *>(b + oRec);
  ws->surfP = reinterpret_cast<float4 *>(b + oP);
  ws->surfN = reinterpret_cast<float4 *>(b + oN);
  ws->shadow = reinterpret_cast<int2 *>(b + oSh);
  wsB->counters = ws->counters; wsB->hitPix = ws->hitPix; wsB->hitRec = ws->hitRec; wsB->surfP = ws->surfP;
  wsB->surfN = ws->surfN; wsB->shadow = ws->shadow;
  wsB->rayHit = reinterpret_cast<uint32_t *>(b + oRay);
  wsB->cap = (uint32_t)pixels;
  wsC->b = *wsB;
  if (withQueues) {
    char *q = b + oQ;
    auto take = [&](size_t bytes) { char *r = q; q += align(bytes); return r; };
    for (int k = 0; k < 2; k++) {
      wsC->pPix[k] = reinterpret_cast<int *>(take(pixels * 4));
      wsC->pT[k] = reinterpret_cast<float *>(take(pixels * 4));
      wsC->pSteps[k] = reinterpret_cast<int *>(take(pixels * 4));
      wsC->sRay[k] = reinterpret_cast<uint32_t *>(take(pixels * nlq * 4));
      wsC->sT[k] = reinterpret_cast<float *>(take(pixels * nlq * 4));
      wsC->sPen[k] = reinterpret_cast<float *>(take(pixels * nlq * 4));
      wsC->sSteps[k] = reinterpret_cast<int *>(take(pixels * nlq * 4));
    }
  }
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
// object space (sdMatch's sizes, frag:1262-1293), world centre c = −A⁻¹b and extent r·‖A⁻¹‖_F of the ball's image under
// the model matrix (A, b = linear part and translation of invModel), and κ = scaleFactor / ‖A⁻¹‖_F, a lower bound of
// (distance value) / (world distance to the object's ball) for the exact SDFs.  The Mandelbulb (power 8, |seed| <= 2)
// enters with r = 2.1: beyond it the estimate is >= 0.68·scaleFactor.  Scenes with a type that has no bound here
// (2-D Mandelbrot, Sierpinski) get cullOk = 0.
void scene_cull_ball(SceneBlock *h) {
  h->cullOk = 0;
  h->cullC[0] = h->cullC[1] = h->cullC[2] = 0.0f;
  h->cullR2 = 0.0f;
  h->cullR2Soft = 0.0f;
  const int n = h->numObjects;
  if (n <= 0) return;
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
    double nf = 0.0;
    for (int r0 = 0; r0 < 3; r0++)
      for (int c0 = 0; c0 < 3; c0++) nf += inv[r0][c0] * inv[r0][c0];
    nf = std::sqrt(nf);
    const double b[3] = {M[12], M[13], M[14]};
    cx[i] = -(inv[0][0] * b[0] + inv[0][1] * b[1] + inv[0][2] * b[2]);
    cy[i] = -(inv[1][0] * b[0] + inv[1][1] * b[1] + inv[1][2] * b[2]);
    cz[i] = -(inv[2][0] * b[0] + inv[2][1] * b[1] + inv[2][2] * b[2]);
    rad[i] = r * nf;
    const double ki = (double)o.scaleFactor / nf;
    if (!(ki > 1e-6) || !std::isfinite(rad[i]) || !std::isfinite(cx[i] + cy[i] + cz[i])) return;
    // hard bound: the bulb's constant 0.68·scaleFactor needs no δ; soft bound: beyond ρ = 2.1 its estimate ≈ 0.5·ρ·ln ρ has
    // slope >= 0.87 in object space
    if (o.type != RM_MANDELBULB) kappa = ki < kappa ? ki : kappa;
    const double ksi = (o.type == RM_MANDELBULB) ? 0.8 * ki : ki;
    kappaSoft = ksi < kappaSoft ? ksi : kappaSoft;
    C[0] += cx[i] / n; C[1] += cy[i] / n; C[2] += cz[i] / n;
  }
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
  // Soft shadows: a shadow ray starts on a surface, i.e. inside the ball (radius R), and at distance ρ from the centre has
  // travelled t <= ρ + R while every distance value is >= κ·(ρ − R).  8·κ·(ρ − R) >= ρ + R  ⇔  ρ >= R·(8κ + 1)/(8κ − 1):
  // past that radius min(pen, 8·d/t) is settled.
  const double ks = kappaSoft;
  if (ks > 0.2 && ks < 1e29) {
    const double Rs = R * (8.0 * ks + 1.0) / (8.0 * ks - 1.0) * 1.001;
    if (std::isfinite(Rs) && Rs < 1e6) h->cullR2Soft = (float)(Rs * Rs);
  }
}

int stage_scene(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                const RmGlobals *g, const RmSettings *s, hipStream_t stream, Slot **slotOut, const RmResources &res) {
  Slot *slot;
  int st = acquire_slot(&slot);
  if (st != RM_OK) return st;
  SceneBlock *h = slot->host;
  h->cam = *cam; h->g = *g; h->s = *s;
  h->numObjects = numObjects; h->numLights = numLights;
  for (int i = 0; i < numObjects; i++) h->objs[i] = objs[i];
  for (int i = 0; i < numLights; i++) h->lights[i] = lights[i];
  h->numTextures = res.numTextures;
  for (int i = 0; i < res.numTextures; i++) h->tex[i] = res.textures[i];
  h->noise = res.noise;
  for (int f = 0; f < 6; f++) h->skybox[f] = res.skybox[f];
  h->ltc1 = res.ltc1; h->ltc2 = res.ltc2;
  scene_cull_ball(h);
  HIP_OK(hipMemcpyAsync(slot->dev, h, sizeof(SceneBlock), hipMemcpyHostToDevice, stream));
  *slotOut = slot;
  return RM_OK;
}

int launch_render(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                  const RmGlobals *g, const RmSettings *s, int W, int H, RowMap map, int nRows, float *d_rgba,
                  float *d_bright, hipStream_t stream, bool count, RmCounters *countersOut,
                  const RmResources &res = kNoResources) {
  std::lock_guard<std::mutex> lock(g_mu);
  int st = validate_scene(cam, objs, numObjects, lights, numLights, g, s, res);
  if (st != RM_OK) return st;
  if (W <= 0 || H <= 0 || nRows < 0) { set_error("bad frame size"); return RM_ERR_INVALID_ARGUMENT; }
  if (nRows == 0) return RM_OK;  // empty row range: nothing to write, a null buffer is fine
  if (!d_rgba) { set_error("null output buffer"); return RM_ERR_INVALID_ARGUMENT; }
  if ((st = check_device_pointers(res, d_rgba, d_bright)) != RM_OK) return st;
  Slot *slot;
  st = stage_scene(cam, objs, numObjects, lights, numLights, g, s, stream, &slot, res);
  if (st != RM_OK) return st;
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
  unsigned long long *dc = g_dev[dev].dCounters;
  if (count) HIP_OK(hipMemsetAsync(dc, 0, 3 * sizeof(unsigned long long), stream));
  dim3 grid((W + kBlockW - 1) / kBlockW, (nRows + kBlockH - 1) / kBlockH), block(256);
  const bool bulb = (numObjects == 1 && objs[0].type == RM_MANDELBULB);
  auto nonzero3 = [](const float *v) { return v[0] != 0.0f || v[1] != 0.0f || v[2] != 0.0f; };
  // The wavefront pipeline covers the single-Mandelbulb class without secondary rays; everything else (and the
  // counted variant) runs the one-lane-per-pixel kernel.  Both produce the same bits.
  static const int envPath = std::getenv("RM_KERNEL_PATH") ? std::atoi(std::getenv("RM_KERNEL_PATH")) : 0;
  int path = g_kernelPath ? g_kernelPath : envPath;
  if (path == 0) path = kAutoBulbPath;
  const bool envFeatures = (s->features & (RM_FEAT_TERRAIN | RM_FEAT_CLOUD | RM_FEAT_SKY_BACKGROUND | RM_FEAT_NIGHTSKY_BACKGROUND | RM_FEAT_SEA)) != 0;
  // anything that reads a sampler or takes the area-light branches: object textures, sky box, emissive rectangles, area lights
  bool textured = s->enableSkyBox != 0;
  for (int i = 0; i < numObjects; i++) textured = textured || objs[i].texLoc != -1 || objs[i].isEmissive;
  for (int i = 0; i < numLights; i++) textured = textured || lights[i].type == RM_LIGHT_AREA;
  const bool pipeline = bulb && !count && path != 1 && !envFeatures && !textured && !g->isTwoD && s->maxSteps >= 1 && s->fractalIters >= 1 &&
                        !(s->enableReflection && nonzero3(objs[0].cReflective)) &&
                        !(s->enableRefraction && nonzero3(objs[0].cTransparent));
  TimedLaunch tl{};
  auto stamp = [&](int i) -> int {
    if (!g_timing) return RM_OK;
    HIP_OK(hipEventCreate(&tl.ev[i]));
    HIP_OK(hipEventRecord(tl.ev[i], stream));
    tl.n = i + 1;
    return RM_OK;
  };
  float4 *o = reinterpret_cast<float4 *>(d_rgba), *b = reinterpret_cast<float4 *>(d_bright);
  if (pipeline) {
    DeviceState &ds = g_dev[dev];
    if (ds.numCUs == 0) {
      hipDeviceProp_t prop;
      HIP_OK(hipGetDeviceProperties(&prop, dev));
      ds.numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    BulbWs ws;
    BulbWsB wsB;
    BulbWsC wsC;
    // hit-list capacity: every pixel may hit, plus one partly used 64-slot chunk per persistent wave
    const size_t slots = (size_t)nRows * W + (size_t)kSlotChunk * ds.numCUs * 8 * 4;
    st = bulb_workspace(ds, slots, numLights, stream, &ws, &wsB, &wsC, path == 4);
    if (st != RM_OK) return st;
    HIP_OK(hipMemsetAsync(ws.counters, 0, 256, stream));
    // tuning knobs for A/B runs (defaults are the measured best)
    static const int blocksPerCU = std::getenv("RM_PIPE_BLOCKS_PER_CU") ? std::atoi(std::getenv("RM_PIPE_BLOCKS_PER_CU")) : 8;
    static const int flushThr = std::getenv("RM_PIPE_FLUSH") ? std::atoi(std::getenv("RM_PIPE_FLUSH")) : kDefaultFlushThreshold;
    const dim3 persistent(ds.numCUs * (blocksPerCU > 0 ? blocksPerCU : 8)), dense(ds.numCUs * 16);
    if ((st = stamp(0)) != RM_OK) return st;
    if (path == 2) {
      hipLaunchKernelGGL(bulb_primary_kernel, persistent, block, 0, stream, slot->dev, map, W, H, nRows, o, b, ws, flushThr);
      if ((st = stamp(1)) != RM_OK) return st;
      hipLaunchKernelGGL(bulb_surface_kernel, dense, block, 0, stream, slot->dev, map, W, H, ws);
      if ((st = stamp(2)) != RM_OK) return st;
      hipLaunchKernelGGL(bulb_shadow_kernel, persistent, block, 0, stream, slot->dev, ws, flushThr);
      if ((st = stamp(3)) != RM_OK) return st;
      hipLaunchKernelGGL(bulb_shade_kernel, dense, block, 0, stream, slot->dev, map, W, H, o, b, ws);
    } else if (path == 4) {
      // step budgets per pass; the last pass always runs to the end of the march
      static const int kBudgets[] = {16, 16, 32, 64, 1 << 30};
      const int nPass = (int)(sizeof(kBudgets) / sizeof(kBudgets[0]));
      hipLaunchKernelGGL((bulbC_primary_kernel<true>), grid, block, 0, stream, slot->dev, map, W, H, nRows, o, b, wsC, 0, kBudgets[0]);
      for (int p = 1; p < nPass; p++)
        hipLaunchKernelGGL((bulbC_primary_kernel<false>), persistent, block, 0, stream, slot->dev, map, W, H, nRows, o, b, wsC, p, kBudgets[p]);
      if ((st = stamp(1)) != RM_OK) return st;
      hipLaunchKernelGGL(bulbB_surface_kernel, dense, block, 0, stream, slot->dev, map, W, H, wsB);
      if ((st = stamp(2)) != RM_OK) return st;
      hipLaunchKernelGGL((bulbC_shadow_kernel<true>), dense, block, 0, stream, slot->dev, wsC, 0, kBudgets[0]);
      for (int p = 1; p < nPass; p++)
        hipLaunchKernelGGL((bulbC_shadow_kernel<false>), persistent, block, 0, stream, slot->dev, wsC, p, kBudgets[p]);
      if ((st = stamp(3)) != RM_OK) return st;
      hipLaunchKernelGGL(bulbB_shade_kernel, dense, block, 0, stream, slot->dev, map, W, H, o, b, wsB);
    } else {
      hipLaunchKernelGGL(bulbB_primary_kernel, grid, block, 0, stream, slot->dev, map, W, H, nRows, o, b, wsB);
      if ((st = stamp(1)) != RM_OK) return st;
      hipLaunchKernelGGL(bulbB_surface_kernel, dense, block, 0, stream, slot->dev, map, W, H, wsB);
      if ((st = stamp(2)) != RM_OK) return st;
      hipLaunchKernelGGL(bulbB_shadow_kernel, dense, block, 0, stream, slot->dev, wsB);
      if ((st = stamp(3)) != RM_OK) return st;
      hipLaunchKernelGGL(bulbB_shade_kernel, dense, block, 0, stream, slot->dev, map, W, H, o, b, wsB);
    }
    if ((st = stamp(4)) != RM_OK) return st;
  } else {
    if ((st = stamp(0)) != RM_OK) return st;
    // instantiations <BULB, COUNT, ENV, TEX>: the bulb class and the generic table walk, plain and counted, without
    // procedural layers or textures; the generic kernel with either or both.  Features a launch does not need are
    // compiled out so the common kernels keep their register budget.
#define RM_LAUNCH(B, C, E, T) hipLaunchKernelGGL((render_kernel<B, C, E, T>), grid, block, 0, stream, slot->dev, map, W, H, nRows, o, b, dc)
    if (envFeatures || textured) {
      if (envFeatures && textured) RM_LAUNCH(false, false, true, true);
      else if (envFeatures) RM_LAUNCH(false, false, true, false);
      else RM_LAUNCH(false, false, false, true);
    } else if (bulb) {
      if (count) RM_LAUNCH(true, true, false, false);
      else RM_LAUNCH(true, false, false, false);
    } else {
      if (count) RM_LAUNCH(false, true, false, false);
      else RM_LAUNCH(false, false, false, false);
    }
#undef RM_LAUNCH
    if ((st = stamp(1)) != RM_OK) return st;
  }
  HIP_OK(hipGetLastError());
  if (g_timing) g_timed.push_back(tl);
  HIP_OK(hipEventRecord(slot->done, stream));
  if (count) {
    unsigned long long hc[3];
    HIP_OK(hipMemcpyAsync(hc, dc, sizeof(hc), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    if (countersOut) { countersOut->sceneEvals = hc[0]; countersOut->bulbIters = hc[1]; countersOut->hitPixels = hc[2]; }
  }
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
  RowMap map{rowBegin, n > 0 ? n : 1, 0, 1};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, n, d_rgba, d_bright,
                       static_cast<hipStream_t>(stream), false, nullptr);
}

int rm_render_ex(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                 const RmGlobals *g, const RmSettings *s, const RmTexture *textures, int numTextures, int W, int H,
                 int rowBegin, int rowEnd, float *d_rgba, float *d_bright, void *stream) {
  if (rowBegin < 0 || rowEnd > H || rowBegin > rowEnd) { set_error("rows out of range"); return RM_ERR_INVALID_ARGUMENT; }
  int n = rowEnd - rowBegin;
  RowMap map{rowBegin, n > 0 ? n : 1, 0, 1};
  RmResources res{};
  res.textures = textures; res.numTextures = numTextures;
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, n, d_rgba, d_bright,
                       static_cast<hipStream_t>(stream), false, nullptr, res);
}

int rm_render_res(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                  const RmGlobals *g, const RmSettings *s, const RmResources *res, int W, int H, int rowBegin, int rowEnd,
                  float *d_rgba, float *d_bright, void *stream) {
  if (rowBegin < 0 || rowEnd > H || rowBegin > rowEnd) { set_error("rows out of range"); return RM_ERR_INVALID_ARGUMENT; }
  int n = rowEnd - rowBegin;
  RowMap map{rowBegin, n > 0 ? n : 1, 0, 1};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, n, d_rgba, d_bright,
                       static_cast<hipStream_t>(stream), false, nullptr, res ? *res : kNoResources);
}

int rm_render_counted(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                      const RmGlobals *g, const RmSettings *s, int W, int H, int rowBegin, int rowEnd, float *d_rgba,
                      float *d_bright, RmCounters *out) {
  if (rowBegin < 0 || rowEnd > H || rowBegin > rowEnd) { set_error("rows out of range"); return RM_ERR_INVALID_ARGUMENT; }
  int n = rowEnd - rowBegin;
  RowMap map{rowBegin, n > 0 ? n : 1, 0, 1};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, n, d_rgba, d_bright, nullptr, true, out);
}

int rm_render_tiles(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                    const RmGlobals *g, const RmSettings *s, int W, int H, int tileRows, int shard, int numShards,
                    float *d_rgba, float *d_bright, void *stream) {
  if (tileRows <= 0 || numShards <= 0 || shard < 0 || shard >= numShards) {
    set_error("bad tile partition");
    return RM_ERR_INVALID_ARGUMENT;
  }
  RowMap map{0, tileRows, shard, numShards};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, shard_rows(H, tileRows, shard, numShards),
                       d_rgba, d_bright, static_cast<hipStream_t>(stream), false, nullptr);
}

int rm_render_tiles_res(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                        const RmGlobals *g, const RmSettings *s, const RmResources *res, int W, int H, int tileRows,
                        int shard, int numShards, float *d_rgba, float *d_bright, void *stream) {
  if (tileRows <= 0 || numShards <= 0 || shard < 0 || shard >= numShards) {
    set_error("bad tile partition");
    return RM_ERR_INVALID_ARGUMENT;
  }
  RowMap map{0, tileRows, shard, numShards};
  return launch_render(cam, objs, numObjects, lights, numLights, g, s, W, H, map, shard_rows(H, tileRows, shard, numShards),
                       d_rgba, d_bright, static_cast<hipStream_t>(stream), false, nullptr, res ? *res : kNoResources);
}

int rm_deinterleave(const float *d_gathered, float *d_frame, int W, int H, int tileRows, int numShards,
                    int shardStrideRows, void *stream) {
  if (!d_gathered || !d_frame || W <= 0 || H <= 0 || tileRows <= 0 || numShards <= 0 || numShards > 64 ||
      (shardStrideRows != 0 && shardStrideRows < shard_rows(H, tileRows, 0, numShards))) {
    set_error("bad deinterleave arguments");
    return RM_ERR_INVALID_ARGUMENT;
  }
  if (int st = require_device_pointers({{"d_gathered", d_gathered}, {"d_frame", d_frame}})) return st;
  dim3 grid((W + 255) / 256, H), block(256);
  hipLaunchKernelGGL(deinterleave_kernel, grid, block, 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4 *>(d_gathered), reinterpret_cast<float4 *>(d_frame), W, H, tileRows,
                     numShards, shardStrideRows);
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
  std::lock_guard<std::mutex> lock(g_mu);
  g_timing = on != 0;
  for (auto &t : g_timed)
    for (int i = 0; i < t.n; i++) (void)hipEventDestroy(t.ev[i]);
  g_timed.clear();
  return RM_OK;
}
int rm_get_timing(double *avgKernelMs, int *launches) {
  double stages[4];
  return rm_get_stage_timing(avgKernelMs, stages, launches);
}
int rm_get_stage_timing(double *avgTotalMs, double avgStageMs[4], int *launches) {
  std::lock_guard<std::mutex> lock(g_mu);
  double total = 0.0, stage[4] = {0, 0, 0, 0};
  for (auto &t : g_timed) {
    if (t.n < 2) continue;
    HIP_OK(hipEventSynchronize(t.ev[t.n - 1]));
    float ms = 0.0f;
    HIP_OK(hipEventElapsedTime(&ms, t.ev[0], t.ev[t.n - 1]));
    total += ms;
    for (int i = 0; i + 1 < t.n && i < 4; i++) {
      HIP_OK(hipEventElapsedTime(&ms, t.ev[i], t.ev[i + 1]));
      stage[i] += ms;
    }
  }
  const double n = g_timed.empty() ? 1.0 : (double)g_timed.size();
  if (launches) *launches = (int)g_timed.size();
  if (avgTotalMs) *avgTotalMs = total / n;
  if (avgStageMs) for (int i = 0; i < 4; i++) avgStageMs[i] = stage[i] / n;
  for (auto &t : g_timed)
    for (int i = 0; i < t.n; i++) (void)hipEventDestroy(t.ev[i]);
  g_timed.clear();
  return RM_OK;
}
int rm_set_kernel_path(int path) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (path < 0 || path > 4) { set_error("kernel path must be 0..4"); return RM_ERR_INVALID_ARGUMENT; }
  g_kernelPath = path;
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
  std::lock_guard<std::mutex> lock(g_mu);
  RmCamera cam{};
  int st = validate_scene(&cam, objs, numObjects, nullptr, 0, g, s, kNoResources);
  if (st != RM_OK) return st;
  if (!d_pts || !d_out || n < 0) { set_error("bad probe arguments"); return RM_ERR_INVALID_ARGUMENT; }
  if (int st2 = require_device_pointers({{"d_pts", d_pts}, {"d_out", d_out}})) return st2;
  if (n == 0) return RM_OK;
  Slot *slot;
  hipStream_t hs = static_cast<hipStream_t>(stream);
  st = stage_scene(&cam, objs, numObjects, nullptr, 0, g, s, hs, &slot, kNoResources);
  if (st != RM_OK) return st;
  hipLaunchKernelGGL(probe_sdscene_kernel, dim3((n + 255) / 256), dim3(256), 0, hs, slot->dev, d_pts, d_out, n);
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(slot->done, hs));
  return RM_OK;
}

}  // extern "C"
