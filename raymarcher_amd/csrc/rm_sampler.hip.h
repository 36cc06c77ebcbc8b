// rm_sampler.hip.h — the texture units of the reference as plain global-memory reads: every sampler it binds is an
// RGBA8 image filtered with GL_LINEAR (objTextures/noise: GL_REPEAT, realtimerender.cpp:295-300, 378-395; skybox
// faces and the LTC tables: GL_CLAMP_TO_EDGE, :557-589, :902-930).  Weights are binary32
// (mix(mix(t00,t10,a), mix(t01,t11,a), b), texel = byte/255) so the CPU oracle reproduces every fetch bit for bit;
// the hardware sampler's 8-bit fixed-point weights would not.  The images are tiny (≤ a few MB) and L2-resident.
#pragma once
#include "rm_math.hip.h"

namespace rm {

// byte → [0,1]: the 256 correctly rounded quotients q/255, built per workgroup (initUnormTable) so that a texel
// costs an LDS read instead of an IEEE division (≈11 VALU instructions, sixteen of them per bilinear fetch).
// A lookup equals the division bit for bit.  Kernels that sample must call initUnormTable() first.
__shared__ float s_unorm[256];
RM_DEV void initUnormTable() {
  for (int i = threadIdx.x; i < 256; i += blockDim.x) s_unorm[i] = RM_DIVC((float)i, 255.0f);  // = i / 255 (exact sequence, rm_math.hip.h)
  __syncthreads();
}
RM_DEV float unorm8(unsigned char q) { return s_unorm[q]; }

RM_DEV int wrapIndex(float f, int n) {
  f = (fabs_(f) < 1.0e9f) ? f : 0.0f;
  int i = (int)f % n;
  return i < 0 ? i + n : i;
}
RM_DEV int clampIndex(int i, int n) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); }

// texture(sampler2D, (su, sv)) (GL 3.3 §3.8.11).  CLAMP = false: GL_REPEAT, true: GL_CLAMP_TO_EDGE.
template <bool CLAMP>
RM_DEV V4 sampleRGBA8(const uint8_t *pixels, int W, int H, float su, float sv) {
  float u = fma(su, (float)W, -0.5f), v = fma(sv, (float)H, -0.5f);
  float fu = floor_(u), fv = floor_(v);
  float a = u - fu, b = v - fv;
  int i0, i1, j0, j1;
  if (CLAMP) {
    fu = (fabs_(fu) < 1.0e9f) ? fu : 0.0f;
    fv = (fabs_(fv) < 1.0e9f) ? fv : 0.0f;
    int iu = (int)fu, iv = (int)fv;
    i0 = clampIndex(iu, W); i1 = clampIndex(iu + 1, W);
    j0 = clampIndex(iv, H); j1 = clampIndex(iv + 1, H);
  } else {
    i0 = wrapIndex(fu, W); j0 = wrapIndex(fv, H);
    i1 = (i0 + 1 == W) ? 0 : i0 + 1; j1 = (j0 + 1 == H) ? 0 : j0 + 1;
  }
  const uchar4 *px = reinterpret_cast<const uchar4 *>(pixels);
  uchar4 p00 = px[(size_t)j0 * W + i0], p10 = px[(size_t)j0 * W + i1], p01 = px[(size_t)j1 * W + i0], p11 = px[(size_t)j1 * W + i1];
  V4 lo = v4(mix_(unorm8(p00.x), unorm8(p10.x), a), mix_(unorm8(p00.y), unorm8(p10.y), a),
             mix_(unorm8(p00.z), unorm8(p10.z), a), mix_(unorm8(p00.w), unorm8(p10.w), a));
  V4 hi = v4(mix_(unorm8(p01.x), unorm8(p11.x), a), mix_(unorm8(p01.y), unorm8(p11.y), a),
             mix_(unorm8(p01.z), unorm8(p11.z), a), mix_(unorm8(p01.w), unorm8(p11.w), a));
  return v4(mix_(lo.x, hi.x, b), mix_(lo.y, hi.y, b), mix_(lo.z, hi.z, b), mix_(lo.w, hi.w, b));
}
// objTextures[i]: RGB of a GL_REPEAT fetch.
RM_DEV V3 sampleTexture(const RmTexture &t, float su, float sv) {
  V4 c = sampleRGBA8<false>(t.pixels, t.width, t.height, su, sv);
  return v3(c.x, c.y, c.z);
}
// texture(samplerCube, r): face and (s,t) of GL 3.3 §3.8.10 table 3.19, ties x before y before z; the fetch
// stays inside the face (no seamless filtering: the reference never enables it).
RM_DEV V3 sampleCube(const RmTexture *faces, V3 r) {
  float ax = fabs_(r.x), ay = fabs_(r.y), az = fabs_(r.z), sc, tc, ma;
  int face;
  if (ax >= ay && ax >= az) { ma = ax; face = (r.x >= 0.0f) ? 0 : 1; sc = (r.x >= 0.0f) ? -r.z : r.z; tc = -r.y; }
  else if (ay >= az)        { ma = ay; face = (r.y >= 0.0f) ? 2 : 3; sc = r.x; tc = (r.y >= 0.0f) ? r.z : -r.z; }
  else                      { ma = az; face = (r.z >= 0.0f) ? 4 : 5; sc = (r.z >= 0.0f) ? r.x : -r.x; tc = -r.y; }
  const RmTexture &t = faces[face];
  const float ima = rcp_(ma);  // contract: rm_divr
  V4 c = sampleRGBA8<true>(t.pixels, t.width, t.height, fma(sc * ima, 0.5f, 0.5f), fma(tc * ima, 0.5f, 0.5f));
  return v3(c.x, c.y, c.z);
}

}  // namespace rm
