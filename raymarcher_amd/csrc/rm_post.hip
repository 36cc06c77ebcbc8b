// rm_post.hip — the post passes the reference runs after the raymarch draw call, as HIP image kernels:
// bloom (separable 9-tap Gaussian ping-pong over BrightColor), HDR tone map / gamma, FXAA.
// Reference: Realtime::applyBloom / applyLightEffects / applyFXAA (src/realtimerender.cpp:92-165),
// resources/blur.frag, hdr.frag, fxaa.frag; FBO formats from initCustomFBO (src/realtimerender.cpp:479-552).
//
// These are HBM-streaming kernels (8-16 B read + 4-16 B written per pixel; the 9 taps of a blur pass and the
// FXAA neighbourhood hit L1/L2), nowhere near the raymarch's cost.  Intermediate images use the reference's storage
// formats — binary16 for the HDR / bright / ping-pong targets, 8-bit for the FXAA source — so values are rounded
// exactly where the reference's framebuffers round them, and the result is bit-reproducible against the oracle.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <mutex>
#include <string>

#include "rm_device.hip.h"
#include "rm_internal.h"

namespace rm {
namespace {

struct alignas(8) half4 { __half x, y, z, w; };  // 8-byte aligned: one 64-bit global / LDS access per texel

__device__ __forceinline__ float q16(float v) { return __half2float(__float2half_rn(v)); }
__device__ __forceinline__ unsigned char to8(float v) {
  v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
  return (unsigned char)(int)fma(v, 255.0f, 0.5f);
}
__device__ __forceinline__ int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// blur.frag:9-31, CLAMP_TO_EDGE, taps on texel centres
__global__ void blur_kernel(const half4 *__restrict__ src, half4 *__restrict__ dst, int W, int H, int horizontal) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const float w[5] = {0.2270270270f, 0.1945945946f, 0.1216216216f, 0.0540540541f, 0.0162162162f};
  half4 c = src[(size_t)y * W + x];
  float r = __half2float(c.x) * w[0], g = __half2float(c.y) * w[0], b = __half2float(c.z) * w[0];
#pragma unroll
  for (int i = 1; i < 5; i++) {
    const int xp = horizontal ? clampi(x + i, W) : x, yp = horizontal ? y : clampi(y + i, H);
    const int xm = horizontal ? clampi(x - i, W) : x, ym = horizontal ? y : clampi(y - i, H);
    half4 p = src[(size_t)yp * W + xp], m = src[(size_t)ym * W + xm];
    r = fma(__half2float(p.x), w[i], r); g = fma(__half2float(p.y), w[i], g); b = fma(__half2float(p.z), w[i], b);
    r = fma(__half2float(m.x), w[i], r); g = fma(__half2float(m.y), w[i], g); b = fma(__half2float(m.z), w[i], b);
  }
  // The sums must be rounded to binary32 BEFORE the binary16 store (that is what a GL fragment shader writing an
  // RGBA16F target does).  Without the barrier hipcc fuses the last fma with the conversion (v_fma_mixlo_f16, a
  // single rounding), which differs from the two-step rounding in rare near-tie cases.
  asm volatile("" : "+v"(r), "+v"(g), "+v"(b));
  dst[(size_t)y * W + x] = half4{__float2half_rn(r), __float2half_rn(g), __float2half_rn(b), __float2half_rn(1.0f)};
}

// One horizontal pass followed by one vertical pass (blur.frag twice) in a single launch: the source tile with its
// 4-texel apron is staged in LDS once, the horizontal result — rounded to binary16 exactly as the ping-pong target
// would store it — stays in LDS, and the vertical taps read it from there.  Same arithmetic, same order, same two
// roundings per pass as two blur_kernel launches; global traffic per pair drops from 2×(9 L2 reads + 1 write) per
// pixel to 1.4 reads + 1 write.  Rows/columns outside the image replicate the edge (CLAMP_TO_EDGE) by clamping the
// GLOBAL coordinate when the tile is staged, which is what clamping each tap does.
constexpr int kBlurTW = 64, kBlurTH = 32, kBlurR = 4;
// F32SRC: `srcv` is the float BrightColor plane (pass 1): the staging step rounds it to binary16 as the reference's RGBA16F
// colour attachment does, so no separate conversion pass (and its 8 B/pixel round trip) is needed.
template <bool F32SRC>
__global__ __launch_bounds__(256) void blur_pair_kernel(const void *__restrict__ srcv, half4 *__restrict__ dst, int W, int H) {
  constexpr int SW = kBlurTW + 2 * kBlurR, SH = kBlurTH + 2 * kBlurR;  // staged source: 72 × 40
  constexpr int kStage = SW * SH, kPer = (kStage + 255) / 256;
  __shared__ half4 s_src[SH][SW];
  __shared__ half4 s_h[SH][kBlurTW];
  const float w[5] = {0.2270270270f, 0.1945945946f, 0.1216216216f, 0.0540540541f, 0.0162162162f};
  const int x0 = blockIdx.x * kBlurTW, y0 = blockIdx.y * kBlurTH;
  {  // every load of the thread is issued before the first LDS write waits for one (a load-wait-write loop is latency-bound)
    using Texel = typename std::conditional<F32SRC, float4, half4>::type;
    const Texel *src = static_cast<const Texel *>(srcv);
    Texel t[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int i = threadIdx.x + k * 256, lx = i % SW, ly = i / SW;
      if (i < kStage) t[k] = src[(size_t)clampi(y0 - kBlurR + ly, H) * W + clampi(x0 - kBlurR + lx, W)];
    }
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int i = threadIdx.x + k * 256, lx = i % SW, ly = i / SW;
      if (i < kStage) {
        if constexpr (F32SRC) s_src[ly][lx] = half4{__float2half_rn(t[k].x), __float2half_rn(t[k].y), __float2half_rn(t[k].z), __float2half_rn(1.0f)};
        else s_src[ly][lx] = t[k];
      }
    }
  }
  __syncthreads();
  static_assert((kBlurTW * SH) % 256 == 0 && (kBlurTW * kBlurTH) % 256 == 0, "whole rounds of the block");
#pragma unroll 2
  for (int k0 = 0; k0 < kBlurTW * SH / 256; k0++) {  // horizontal pass on every staged row
    const int i = threadIdx.x + k0 * 256, lx = i % kBlurTW, ly = i / kBlurTW;
    const half4 c = s_src[ly][lx + kBlurR];
    float r = __half2float(c.x) * w[0], g = __half2float(c.y) * w[0], b = __half2float(c.z) * w[0];
#pragma unroll
    for (int k = 1; k < 5; k++) {
      const half4 p = s_src[ly][lx + kBlurR + k], m = s_src[ly][lx + kBlurR - k];
      r = fma(__half2float(p.x), w[k], r); g = fma(__half2float(p.y), w[k], g); b = fma(__half2float(p.z), w[k], b);
      r = fma(__half2float(m.x), w[k], r); g = fma(__half2float(m.y), w[k], g); b = fma(__half2float(m.z), w[k], b);
    }
    asm volatile("" : "+v"(r), "+v"(g), "+v"(b));  // binary32 first, then binary16 (see blur_kernel)
    s_h[ly][lx] = half4{__float2half_rn(r), __float2half_rn(g), __float2half_rn(b), __float2half_rn(1.0f)};
  }
  __syncthreads();
#pragma unroll 2
  for (int k0 = 0; k0 < kBlurTW * kBlurTH / 256; k0++) {  // vertical pass on the tile
    const int i = threadIdx.x + k0 * 256, lx = i % kBlurTW, ly = i / kBlurTW;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= W || y >= H) continue;
    const half4 c = s_h[ly + kBlurR][lx];
    float r = __half2float(c.x) * w[0], g = __half2float(c.y) * w[0], b = __half2float(c.z) * w[0];
#pragma unroll
    for (int k = 1; k < 5; k++) {
      const half4 p = s_h[ly + kBlurR + k][lx], m = s_h[ly + kBlurR - k][lx];
      r = fma(__half2float(p.x), w[k], r); g = fma(__half2float(p.y), w[k], g); b = fma(__half2float(p.z), w[k], b);
      r = fma(__half2float(m.x), w[k], r); g = fma(__half2float(m.y), w[k], g); b = fma(__half2float(m.z), w[k], b);
    }
    asm volatile("" : "+v"(r), "+v"(g), "+v"(b));
    dst[(size_t)y * W + x] = half4{__float2half_rn(r), __float2half_rn(g), __float2half_rn(b), __float2half_rn(1.0f)};
  }
}

// hdr.frag:13-35 without bloom (with it: light_bloom_kernel).  Writes either the float frame (no FXAA afterwards) or the RGBA8 FXAA
// source.
__global__ void light_kernel(const float4 *__restrict__ frag, float4 *__restrict__ outF, uchar4 *__restrict__ out8, int n, int hdr,
                             float exposure) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 f = frag[i];
  const float c[3] = {q16(f.x), q16(f.y), q16(f.z)};
  float r[3];
#pragma unroll
  for (int k = 0; k < 3; k++) r[k] = hdr ? 1.0f - exp_((-c[k]) * exposure) : pow_(c[k], 1.0f / 2.2f);
  if (out8) out8[i] = make_uchar4(to8(r[0]), to8(r[1]), to8(r[2]), 255);
  else outF[i] = make_float4(r[0], r[1], r[2], 1.0f);
}

// applyBloom's last pass (pass 9, horizontal) and the hdr.frag composite in one launch: a block owns 256 pixels of one row, stages
// the 264 source texels in LDS, blurs them — the sum rounded to binary32 and then to binary16, the value the ping-pong target would
// have held — and composites.  Saves the 8 B/pixel written and re-read between the two and a pass whose nine taps per pixel
// came through L1.
__global__ __launch_bounds__(256) void light_bloom_kernel(const float4 *__restrict__ frag, const half4 *__restrict__ src, float4 *__restrict__ outF,
                                                          uchar4 *__restrict__ out8, int W, int H, float exposure) {
  __shared__ half4 s_row[256 + 2 * kBlurR];
  const float w[5] = {0.2270270270f, 0.1945945946f, 0.1216216216f, 0.0540540541f, 0.0162162162f};
  const int x0 = blockIdx.x * 256, y = blockIdx.y, x = x0 + threadIdx.x;
  const half4 *row = src + (size_t)y * W;
  s_row[threadIdx.x] = row[clampi(x0 - kBlurR + (int)threadIdx.x, W)];
  if (threadIdx.x < 2 * kBlurR) s_row[256 + threadIdx.x] = row[clampi(x0 - kBlurR + 256 + (int)threadIdx.x, W)];
  float4 f = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (x < W) f = frag[(size_t)y * W + x];
  __syncthreads();
  if (x >= W) return;
  const int lx = threadIdx.x + kBlurR;
  const half4 c0 = s_row[lx];
  float r = __half2float(c0.x) * w[0], g = __half2float(c0.y) * w[0], b = __half2float(c0.z) * w[0];
#pragma unroll
  for (int k = 1; k < 5; k++) {
    const half4 p = s_row[lx + k], m = s_row[lx - k];
    r = fma(__half2float(p.x), w[k], r); g = fma(__half2float(p.y), w[k], g); b = fma(__half2float(p.z), w[k], b);
    r = fma(__half2float(m.x), w[k], r); g = fma(__half2float(m.y), w[k], g); b = fma(__half2float(m.z), w[k], b);
  }
  asm volatile("" : "+v"(r), "+v"(g), "+v"(b));  // binary32 first, then binary16 (see blur_kernel)
  const float c[3] = {q16(f.x) + q16(r), q16(f.y) + q16(g), q16(f.z) + q16(b)};
  float o[3];
#pragma unroll
  for (int k = 0; k < 3; k++) o[k] = 1.0f - exp_((-c[k]) * exposure);
  const size_t i = (size_t)y * W + x;
  if (out8) out8[i] = make_uchar4(to8(o[0]), to8(o[1]), to8(o[2]), 255);
  else outF[i] = make_float4(o[0], o[1], o[2], 1.0f);
}

__global__ void quant8_kernel(const float4 *__restrict__ frag, uchar4 *__restrict__ out8, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 f = frag[i];
  out8[i] = make_uchar4(to8(f.x), to8(f.y), to8(f.z), 255);
}

// GL_LINEAR / GL_REPEAT fetch of the RGBA8 FXAA source at normalised (u, v).  `unorm` is the 256-entry table byte/255
// (each entry the correctly rounded quotient, so a lookup equals the division bit for bit): twelve IEEE divisions per
// fetch — ≈130 VALU instructions — become twelve LDS reads.
__device__ __forceinline__ V3 fetch8(const uchar4 *__restrict__ img, const float *unorm, int W, int H, float u, float v) {
  float fx = fma(u, (float)W, -0.5f), fy = fma(v, (float)H, -0.5f);
  float x0 = floor_(fx), y0 = floor_(fy);
  float a = fx - x0, b = fy - y0;
  int i0 = wrapIndex(x0, W), j0 = wrapIndex(y0, H);
  int i1 = (i0 + 1 == W) ? 0 : i0 + 1, j1 = (j0 + 1 == H) ? 0 : j0 + 1;
  uchar4 p00 = img[(size_t)j0 * W + i0], p10 = img[(size_t)j0 * W + i1], p01 = img[(size_t)j1 * W + i0], p11 = img[(size_t)j1 * W + i1];
  auto f = [&](unsigned char q) { return unorm[q]; };
  return v3(mix_(mix_(f(p00.x), f(p10.x), a), mix_(f(p01.x), f(p11.x), a), b),
            mix_(mix_(f(p00.y), f(p10.y), a), mix_(f(p01.y), f(p11.y), a), b),
            mix_(mix_(f(p00.z), f(p10.z), a), mix_(f(p01.z), f(p11.z), a), b));
}
__device__ __forceinline__ float rgb2luma(V3 c) { return sqrt_(dot(c, v3(0.299f, 0.587f, 0.114f))); }  // fxaa.frag:18-20

// fxaa.frag:22-166
__global__ void fxaa_kernel(const uchar4 *__restrict__ img, float4 *__restrict__ out, int W, int H) {
  __shared__ float s_unorm[256];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;  // 256 threads in a 2-D block (16 × 16 pixels by default)
  s_unorm[tid] = (float)tid / 255.0f;
  __syncthreads();
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= W || y >= H) return;
  const float quality[12] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.5f, 2.0f, 2.0f, 2.0f, 2.0f, 4.0f, 8.0f};
  const float invW = 1.0f / (float)W, invH = 1.0f / (float)H;
  const float tu = ((float)x + 0.5f) / (float)W, tv = ((float)y + 0.5f) / (float)H;
  auto texOff = [&](int dx, int dy) { return fetch8(img, s_unorm, W, H, tu + (float)dx * invW, tv + (float)dy * invH); };
  V3 colorCenter = texOff(0, 0);
  V3 result = colorCenter;
  float lumaCenter = rgb2luma(colorCenter);
  float lumaDown = rgb2luma(texOff(0, -1)), lumaUp = rgb2luma(texOff(0, 1));
  float lumaLeft = rgb2luma(texOff(-1, 0)), lumaRight = rgb2luma(texOff(1, 0));
  float lumaMin = min_(lumaCenter, min_(min_(lumaDown, lumaUp), min_(lumaLeft, lumaRight)));
  float lumaMax = max_(lumaCenter, max_(max_(lumaDown, lumaUp), max_(lumaLeft, lumaRight)));
  float lumaRange = lumaMax - lumaMin;
  if (!(lumaRange < max_(0.0312f, lumaMax * 0.125f))) {
    float lumaDownLeft = rgb2luma(texOff(-1, -1)), lumaUpRight = rgb2luma(texOff(1, 1));
    float lumaUpLeft = rgb2luma(texOff(-1, 1)), lumaDownRight = rgb2luma(texOff(1, -1));
    float lumaDownUp = lumaDown + lumaUp, lumaLeftRight = lumaLeft + lumaRight;
    float lumaLeftCorners = lumaDownLeft + lumaUpLeft, lumaDownCorners = lumaDownLeft + lumaDownRight;
    float lumaRightCorners = lumaDownRight + lumaUpRight, lumaUpCorners = lumaUpRight + lumaUpLeft;
    float edgeHorizontal = (fabs_(fma(-2.0f, lumaLeft, lumaLeftCorners)) + fabs_(fma(-2.0f, lumaCenter, lumaDownUp)) * 2.0f) +
                           fabs_(fma(-2.0f, lumaRight, lumaRightCorners));
    float edgeVertical = (fabs_(fma(-2.0f, lumaUp, lumaUpCorners)) + fabs_(fma(-2.0f, lumaCenter, lumaLeftRight)) * 2.0f) +
                         fabs_(fma(-2.0f, lumaDown, lumaDownCorners));
    bool isHorizontal = edgeHorizontal >= edgeVertical;
    float luma1 = isHorizontal ? lumaDown : lumaLeft, luma2 = isHorizontal ? lumaUp : lumaRight;
    float gradient1 = luma1 - lumaCenter, gradient2 = luma2 - lumaCenter;
    bool is1Steepest = fabs_(gradient1) >= fabs_(gradient2);
    float gradientScaled = 0.25f * max_(fabs_(gradient1), fabs_(gradient2));
    float stepLength = isHorizontal ? invH : invW;
    float lumaLocalAverage;
    if (is1Steepest) { stepLength = -stepLength; lumaLocalAverage = 0.5f * (luma1 + lumaCenter); }
    else lumaLocalAverage = 0.5f * (luma2 + lumaCenter);
    float cu = tu, cv = tv;
    if (isHorizontal) cv = fma(stepLength, 0.5f, cv); else cu = fma(stepLength, 0.5f, cu);
    float ox = isHorizontal ? invW : 0.0f, oy = isHorizontal ? 0.0f : invH;
    float u1 = cu - ox, v1 = cv - oy, u2 = cu + ox, v2 = cv + oy;
    float lumaEnd1 = rgb2luma(fetch8(img, s_unorm, W, H, u1, v1)) - lumaLocalAverage;
    float lumaEnd2 = rgb2luma(fetch8(img, s_unorm, W, H, u2, v2)) - lumaLocalAverage;
    bool reached1 = fabs_(lumaEnd1) >= gradientScaled, reached2 = fabs_(lumaEnd2) >= gradientScaled;
    bool reachedBoth = reached1 && reached2;
    if (!reached1) { u1 -= ox; v1 -= oy; }
    if (!reached2) { u2 += ox; v2 += oy; }
    if (!reachedBoth) {
      for (int i = 2; i < 12; i++) {
        if (!reached1) lumaEnd1 = rgb2luma(fetch8(img, s_unorm, W, H, u1, v1)) - lumaLocalAverage;
        if (!reached2) lumaEnd2 = rgb2luma(fetch8(img, s_unorm, W, H, u2, v2)) - lumaLocalAverage;
        reached1 = fabs_(lumaEnd1) >= gradientScaled;
        reached2 = fabs_(lumaEnd2) >= gradientScaled;
        reachedBoth = reached1 && reached2;
        if (!reached1) { u1 = fma(-ox, quality[i], u1); v1 = fma(-oy, quality[i], v1); }
        if (!reached2) { u2 = fma(ox, quality[i], u2); v2 = fma(oy, quality[i], v2); }
        if (reachedBoth) break;
      }
    }
    float distance1 = isHorizontal ? (tu - u1) : (tv - v1), distance2 = isHorizontal ? (u2 - tu) : (v2 - tv);
    bool isDirection1 = distance1 < distance2;
    float distanceFinal = min_(distance1, distance2);
    float edgeThickness = distance1 + distance2;
    float pixelOffset = -distanceFinal / edgeThickness + 0.5f;
    bool isLumaCenterSmaller = lumaCenter < lumaLocalAverage;
    bool correctVariation = ((isDirection1 ? lumaEnd1 : lumaEnd2) < 0.0f) != isLumaCenterSmaller;
    float finalOffset = correctVariation ? pixelOffset : 0.0f;
    float lumaAverage = (1.0f / 12.0f) * (fma(2.0f, lumaDownUp + lumaLeftRight, lumaLeftCorners) + lumaRightCorners);
    float sub1 = clamp_(fabs_(lumaAverage - lumaCenter) / lumaRange, 0.0f, 1.0f);
    float sub2 = (fma(-2.0f, sub1, 3.0f) * sub1) * sub1;
    float subFinal = (sub2 * sub2) * 0.875f;
    finalOffset = max_(finalOffset, subFinal);
    float fu = tu, fv = tv;
    if (isHorizontal) fv = fma(finalOffset * stepLength, 1.0f, fv); else fu = fma(finalOffset * stepLength, 1.0f, fu);
    result = fetch8(img, s_unorm, W, H, fu, fv);
  }
  out[(size_t)y * W + x] = make_float4(result.x, result.y, result.z, 1.0f);
}


}  // namespace
}  // namespace rm

using namespace rm;

#define HIP_OK(expr)                                                                               \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                                \
      return RM_ERR_DEVICE;                                                                        \
    }                                                                                              \
  } while (0)

extern "C" int rm_post_process(const float *d_frag, const float *d_bright, float *d_out, int W, int H,
                               const RmPostSettings *ps, void *stream) {
  if (!d_frag || !d_out || !ps || W <= 0 || H <= 0) { set_error("bad post-process arguments"); return RM_ERR_INVALID_ARGUMENT; }
  if (ps->enableBloom && !d_bright) { set_error("bloom needs the BrightColor plane"); return RM_ERR_INVALID_ARGUMENT; }
  if (int rc = require_device_pointers({{"d_frag", d_frag}, {"d_bright", d_bright}, {"d_out", d_out}})) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t n = (size_t)W * H;
  // ping-pong + 8-bit staging images of THIS stream: post-processing two frames on two streams never shares them
  void *wsMem = nullptr;
  if (int rc = stream_workspace(kWsPost, st, n * (8 + 8 + 4), &wsMem)) return rc;
  half4 *pa = static_cast<half4 *>(wsMem), *pb = pa + n;
  uchar4 *stage8 = reinterpret_cast<uchar4 *>(pb + n);
  const float4 *frag = reinterpret_cast<const float4 *>(d_frag);
  float4 *out = reinterpret_cast<float4 *>(d_out);
  const dim3 lin((unsigned)((n + 255) / 256)), blk(256), grid2((W + 255) / 256, H);
  const bool light = ps->enableHDR || ps->enableGammaCorrection || ps->enableBloom;
  if (!light && !ps->enableFXAA) {
    if (d_out != d_frag) HIP_OK(hipMemcpyAsync(d_out, d_frag, n * 16, hipMemcpyDeviceToDevice, st));
    return RM_OK;
  }
  if (light) {
    if (ps->enableBloom) {  // applyBloom: 10 passes H,V,H,…; the composite reads the buffer pass 9 wrote
      half4 *src = pa, *dst = pb;
      const dim3 tiles((W + kBlurTW - 1) / kBlurTW, (H + kBlurTH - 1) / kBlurTH);
      // passes 1-8 as four horizontal+vertical pairs through LDS; the first reads the float BrightColor plane directly
      hipLaunchKernelGGL(blur_pair_kernel<true>, tiles, blk, 0, st, static_cast<const void *>(d_bright), src, W, H);
      for (int i = 1; i < 4; i++) {
        hipLaunchKernelGGL(blur_pair_kernel<false>, tiles, blk, 0, st, static_cast<const void *>(src), dst, W, H);
        half4 *t = src; src = dst; dst = t;
      }
      // pass 9 (horizontal), the one composited, inside the composite
      hipLaunchKernelGGL(light_bloom_kernel, grid2, blk, 0, st, frag, src, out, ps->enableFXAA ? stage8 : nullptr, W, H, ps->exposure);
    } else {
      hipLaunchKernelGGL(light_kernel, lin, blk, 0, st, frag, out, ps->enableFXAA ? stage8 : nullptr, (int)n, ps->enableHDR, ps->exposure);
    }
  } else {
    hipLaunchKernelGGL(quant8_kernel, lin, blk, 0, st, frag, stage8, (int)n);
  }
  if (ps->enableFXAA) {
    static const int fbw = std::getenv("RM_FXAA_BLOCK_W") ? std::atoi(std::getenv("RM_FXAA_BLOCK_W")) : 16;
    // 16 × 16 pixels per block (measured at 4K: 256×1 0.269 ms, 64×4 0.263, 32×8 0.257, 16×16 0.252 — the pass is bound by its divergent
    // edge searches and bilinear fetches, not by cache lines)
    const int bw = (fbw == 256 || fbw == 64 || fbw == 32 || fbw == 16) ? fbw : 16, bh = 256 / bw;
    hipLaunchKernelGGL(fxaa_kernel, dim3((W + bw - 1) / bw, (H + bh - 1) / bh), dim3(bw, bh), 0, st, stage8, out, W, H);
  }
  HIP_OK(hipGetLastError());
  return RM_OK;
}
