// rm_env.hip.h — procedural layers of the reference shader for gfx950: FBM value-noise terrain (400-step height
// field march + 32-step shadow), volumetric clouds (128-step front-to-back march, two density samples per dense
// step), analytic sky and fog.  frag:464-746, 1519-1584, 1950-2158; the shader's compile-time
// #defines TERRAIN / CLOUD / SKY_BACKGROUND are runtime feature bits here.  No textures are involved: all
// noise is arithmetic hashing, so the work is pure VALU.  Numeric contract as in rm_math.hip.h; note that GLSL
// evaluates `f*m3*x` as (f*m3)*x, so the scaled constant matrices are formed first.
#pragma once
#include "rm_math.hip.h"
#include "rm_sampler.hip.h"

namespace rm {

struct M3 { float c[3][3]; };  // c[col][row]
RM_DEV V3 mulMV(const M3 &M, V3 v) {
  return v3(fma(M.c[2][0], v.z, fma(M.c[1][0], v.y, M.c[0][0] * v.x)), fma(M.c[2][1], v.z, fma(M.c[1][1], v.y, M.c[0][1] * v.x)),
            fma(M.c[2][2], v.z, fma(M.c[1][2], v.y, M.c[0][2] * v.x)));
}
RM_DEV M3 mulMM(const M3 &A, const M3 &B) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    V3 col = mulMV(A, v3(B.c[c][0], B.c[c][1], B.c[c][2]));
    R.c[c][0] = col.x; R.c[c][1] = col.y; R.c[c][2] = col.z;
  }
  return R;
}
RM_DEV M3 scaleM(const M3 &A, float f) {
  M3 R;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) R.c[c][r] = f * A.c[c][r];
  return R;
}

RM_DEV float hash1f(float n) { return fract_((n * 17.0f) * fract_(n * 0.3183099f)); }  // frag:467-469
RM_DEV float hash1v2(float px, float py) {                                              // frag:472-475
  px = 50.0f * fract_(px * 0.3183099f);
  py = 50.0f * fract_(py * 0.3183099f);
  return fract_((px * py) * (px + py));
}
RM_DEV float quintic(float w) { return ((w * w) * w) * fma(w, fma(w, 6.0f, -15.0f), 10.0f); }
RM_DEV float noiseT(float x, float y) {  // frag:493-502
  float px = floor_(x), py = floor_(y);
  float ux = quintic(fract_(x)), uy = quintic(fract_(y));
  // the shader's "+ 0.0" only turns a −0 lattice coordinate into +0, and the hash starts with fract(·) of a multiple of it,
  // which is +0 for either zero (v_fract_f32(−0) = +0, rm_debug_check_math): dropped
  float a = hash1v2(px, py), b = hash1v2(px + 1.0f, py);
  float c = hash1v2(px, py + 1.0f), d = hash1v2(px + 1.0f, py + 1.0f);
  float t = fma(b - a, ux, a);
  t = fma(c - a, uy, t);
  t = fma((((a - b) - c) + d) * ux, uy, t);
  return fma(2.0f, t, -1.0f);
}
RM_DEV V4 noised3(V3 x) {  // frag:536-567
  V3 p = v3(floor_(x.x), floor_(x.y), floor_(x.z));
  V3 w = v3(fract_(x.x), fract_(x.y), fract_(x.z));
  V3 u = v3(quintic(w.x), quintic(w.y), quintic(w.z));
  V3 du = v3(((30.0f * w.x) * w.x) * fma(w.x, w.x - 2.0f, 1.0f), ((30.0f * w.y) * w.y) * fma(w.y, w.y - 2.0f, 1.0f),
             ((30.0f * w.z) * w.z) * fma(w.z, w.z - 2.0f, 1.0f));
  float n = fma(157.0f, p.z, fma(317.0f, p.y, p.x));
  float a = hash1f(n), b = hash1f(n + 1.0f), c = hash1f(n + 317.0f), d = hash1f(n + 318.0f);  // "+ 0.0" dropped: hash1f(−0) = hash1f(+0)
  float e = hash1f(n + 157.0f), f = hash1f(n + 158.0f), g = hash1f(n + 474.0f), h = hash1f(n + 475.0f);
  float k0 = a, k1 = b - a, k2 = c - a, k3 = e - a;
  float k4 = ((a - b) - c) + d, k5 = ((a - c) - e) + g, k6 = ((a - b) - e) + f;
  float k7 = ((((((-a + b) + c) - d) + e) - f) - g) + h;
  float v = fma(k1, u.x, k0);
  v = fma(k2, u.y, v);
  v = fma(k3, u.z, v);
  v = fma(k4 * u.x, u.y, v);
  v = fma(k5 * u.y, u.z, v);
  v = fma(k6 * u.z, u.x, v);
  v = fma((k7 * u.x) * u.y, u.z, v);
  float dx = fma(k7 * u.y, u.z, fma(k6, u.z, fma(k4, u.y, k1)));
  float dy = fma(k7 * u.z, u.x, fma(k4, u.x, fma(k5, u.z, k2)));
  float dz = fma(k7 * u.x, u.y, fma(k5, u.y, fma(k6, u.x, k3)));
  return v4(fma(2.0f, v, -1.0f), (2.0f * du.x) * dx, (2.0f * du.y) * dy, (2.0f * du.z) * dz);
}
RM_DEV float fbm_9(float x, float y, Counters &cnt) {  // frag:630-644
  cnt.fbm9++;
  const float m00 = 1.9f * 0.80f, m01 = 1.9f * 0.60f, m10 = 1.9f * -0.60f, m11 = 1.9f * 0.80f;
  float a = 0.0f, b = 0.5f;
#pragma unroll 1
  for (int i = 0; i < 9; i++) {
    float n = noiseT(x, y);
    a = fma(b, n, a);
    b = b * 0.55f;
    float nx = fma(m10, y, m00 * x), ny = fma(m11, y, m01 * x);
    x = nx; y = ny;
  }
  return a;
}
RM_DEV V4 fbmd_8(V3 x, Counters &cnt) {  // frag:647-667
  cnt.fbmd8++;
  const M3 m3c = {{{0.00f, 0.80f, 0.60f}, {-0.80f, 0.36f, -0.48f}, {-0.60f, -0.48f, 0.64f}}};   // frag:118-120
  const M3 m3ic = {{{0.00f, -0.80f, -0.60f}, {0.80f, 0.36f, -0.48f}, {0.60f, -0.48f, 0.64f}}};  // frag:121-123
  const M3 fm3 = scaleM(m3c, 2.0f), fm3i = scaleM(m3ic, 2.0f);
  float a = 0.0f, b = 0.5f;
  V3 d = v3(0.0f, 0.0f, 0.0f);
  M3 m = {{{1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f}}};
#pragma unroll
  for (int i = 0; i < 8; i++) {  // unrolled: `m` stays in registers (no runtime-indexed arrays)
    V4 n = noised3(x);
    a = fma(b, n.x, a);
    if (i < 4) d = add(d, mulMV(scaleM(m, b), v3(n.y, n.z, n.w)));
    b = b * 0.65f;
    x = mulMV(fm3, x);
    m = mulMM(fm3i, m);
  }
  return v4(a, d.x, d.y, d.z);
}
RM_DEV void sdTerrain(float px, float pz, float &hgt, float &slope, Counters &cnt) {  // frag:737-746
  float e = fbm_9(RM_DIVR_CONST(px, 2000.0f) + 1.0f, RM_DIVR_CONST(pz, 2000.0f) + -2.0f, cnt);
  slope = 1.0f - smoothstep_(0.12f, 0.13f, fabs_(e + 0.12f));
  e = fma(600.0f, e, 600.0f);
  hgt = fma(90.0f, smoothstep_(552.0f, 594.0f, e), e);
}
// frag:1529-1584, timeOfDay = 0.1
RM_DEV V3 getSunDir() {
  float ang = mix_(0.0f, 3.14f, 0.1f);
  return normalize(v3(cos_(ang), sin_(ang), -0.577f));
}
RM_DEV V3 getSkyColor() {
  V3 c = mix(v3(1.0f, 0.5f, 0.2f), v3(0.8f, 0.9f, 1.1f), smoothstep_(0.0f, 0.2f, 0.1f));
  return mix(c, v3(1.0f, 0.8f, 0.5f), smoothstep_(0.8f, 1.0f, 0.1f));
}
RM_DEV V3 getSunColor() {
  V3 c = mix(v3(1.0f, 0.5f, 0.2f), v3(1.0f, 1.0f, 0.8f), smoothstep_(0.0f, 0.2f, 0.1f));
  return mix(c, v3(1.0f, 0.8f, 0.5f), smoothstep_(0.8f, 1.0f, 0.1f));
}
RM_DEV V3 getSky(V3 rd) {
  V3 col = scale(getSkyColor(), fma(0.4f, rd.y, 0.6f));
  float s = pow_(clamp_(dot(rd, getSunDir()), 0.0f, 1.0f), 32.0f);
  return madd(getSunColor(), s, col);
}
RM_DEV V3 fog(V3 col, float t) {  // frag:1519-1523
  float k = (-t) * 0.00025f;
  V3 ext = v3(exp2_(k * 1.0f), exp2_(k * 1.5f), exp2_(k * 4.0f));
  return v3(fma(1.0f - ext.x, 0.55f, col.x * ext.x), fma(1.0f - ext.y, 0.55f, col.y * ext.y), fma(1.0f - ext.z, 0.58f, col.z * ext.z));
}
RM_DEV V4 cloudsFbm(float iTime, V3 pos, Counters &cnt) {  // frag:1950-1952
  V3 q = v3(fma(0.07f, iTime, fma(pos.x, 0.0015f, 2.0f)), fma(0.07f, 0.5f * iTime, fma(pos.y, 0.0015f, 1.1f)),
            fma(0.07f, -0.15f * iTime, fma(pos.z, 0.0015f, 1.0f)));
  return fbmd_8(q, cnt);
}
RM_DEV float cloudsShadowFlat(float iTime, V3 ro, V3 rd, Counters &cnt) {  // frag:1954-1959
  float t = (900.0f - ro.y) / rd.y;
  if (t < 0.0f) return 1.0f;
  return cloudsFbm(iTime, madd(rd, t, ro), cnt).x;
}
// frag:1961-1974; nnd = −d always (contract decision UB10, iq's original order).  Returns (density, gra.y).
RM_DEV void cloudsMap(float iTime, V3 pos, float &den, float &gy, float &nnd, Counters &cnt) {
  float d = fabs_(pos.y - 900.0f) - 4.0f;
  gy = (pos.y - 900.0f > 0.0f) ? 1.0f : ((pos.y - 900.0f < 0.0f) ? -1.0f : 0.0f);
  V4 n = cloudsFbm(iTime, pos, cnt);
  d = fma(400.0f * n.x, fma(0.3f, gy, 0.7f), d);
  nnd = -d;
  den = (d > 0.0f) ? -d : min_(RM_DIVR_CONST(-d, 100.0f), 0.25f);
  gy = (d > 0.0f) ? 0.0f : gy;
}
RM_DEV bool cloudMarch(float iTime, int steps, V3 ro, V3 rd, float minT, float maxT, V4 &sum, Counters &cnt) {  // frag:1976-2026
  bool hasHit = false;
  float t = minT, thickness = 0.0f;
  const V3 sunColor = getSunColor(), sunDir = getSunDir();
#pragma unroll 1
  for (int i = 0; i < steps; i++) {
    V3 pos = madd(rd, t, ro);
    float den, gy, nnd;
    cloudsMap(iTime, pos, den, gy, nnd, cnt);
    float dt = max_(0.3f, 0.011f * t);
    if (den > 0.001f) {
      hasHit = true;
      float den2, gy2, kk;
      cloudsMap(iTime, madd(sunDir, 70.0f, pos), den2, gy2, kk, cnt);
      float sha = 1.0f - smoothstep_(-200.0f, 200.0f, kk);
      sha = sha * 1.5f;
      V3 nor = normalize(v3(0.0f, gy, 0.0f));
      float dif = clamp_(fma(0.6f, dot(nor, sunDir), 0.4f), 0.0f, 1.0f) * sha;
      float occ = fma(0.1f, 1.0f - den, fma(0.7f, max_(1.0f - RM_DIVR_CONST(kk, 200.0f), 0.0f), 0.2f));
      float up = fma(0.5f, nor.y, 0.5f), dn = fma(-0.5f, nor.y, 0.5f);
      V3 lin = v3(0.0f, 0.0f, 0.0f);
      lin = v3(lin.x + ((0.70f * 1.0f) * up) * occ, lin.y + ((0.80f * 1.0f) * up) * occ, lin.z + ((1.00f * 1.0f) * up) * occ);
      lin = v3(lin.x + ((0.10f * 1.0f) * dn) * occ, lin.y + ((0.40f * 1.0f) * dn) * occ, lin.z + ((0.20f * 1.0f) * dn) * occ);
      lin = v3(lin.x + fma((sunColor.x * 3.0f) * dif, occ, 0.1f), lin.y + fma((sunColor.y * 3.0f) * dif, occ, 0.1f),
               lin.z + fma((sunColor.z * 3.0f) * dif, occ, 0.1f));
      V3 col = v3(0.8f * 0.45f, 0.8f * 0.45f, 0.8f * 0.45f);
      col = mul(col, lin);
      col = fog(col, t);
      float alp = clamp_(((den * 0.5f) * 0.125f) * dt, 0.0f, 1.0f);
      col = scale(col, alp);
      float om = 1.0f - sum.w;
      sum = v4(fma(col.x, om, sum.x), fma(col.y, om, sum.y), fma(col.z, om, sum.z), fma(alp, om, sum.w));
      thickness = fma(dt, den, thickness);
    } else {
      dt = fabs_(den) + 0.2f;
    }
    t = t + dt;
    if (sum.w > 0.995f || t > maxT) break;
  }
  float glow = pow_(clamp_(dot(sunDir, rd), 0.0f, 1.0f), 32.0f);
  float mx = max_(0.0f, fma(-0.0125f, thickness, 1.0f));
  sum.x = sum.x + ((mx * sunColor.x) * 0.3f) * glow;
  sum.y = sum.y + ((mx * sunColor.y) * 0.3f) * glow;
  sum.z = sum.z + ((mx * sunColor.z) * 0.3f) * glow;
  return hasHit;
}
// frag:2031-2057; blue-noise sample = 0 (texture blob missing from the reference checkout), FRAME = 1.
RM_DEV V3 cloudRender(float iTime, V3 ro, V3 rd, V3 bg, bool &hit, float maxT, Counters &cnt) {
  float minT = 0.0f;
  float tl = (600.0f - ro.y) / rd.y, th = (1200.0f - ro.y) / rd.y;
  hit = false;
  if (!(tl > 0.0f)) return bg;
  minT = max_(minT, tl);
  if (th > 0.0f) maxT = min_(maxT, th);
  V4 sum = v4(0.0f, 0.0f, 0.0f, 0.0f);
  float off = (float)(1 % 64) + 0.61803398875f;
  minT = fma(0.3f, fract_(off + 0.0f), minT);
  hit = cloudMarch(iTime, 128, ro, rd, minT, maxT, sum, cnt);
  sum = v4(clamp_(sum.x, 0.0f, 1.0f), clamp_(sum.y, 0.0f, 1.0f), clamp_(sum.z, 0.0f, 1.0f), clamp_(sum.w, 0.0f, 1.0f));
  float om = 1.0f - sum.w;
  return v3(fma(bg.x, om, sum.x), fma(bg.y, om, sum.y), fma(bg.z, om, sum.z));
}
RM_DEV float raymarchTerrain(V3 ro, V3 rd, float tmin, float tmax, Counters &cnt) {  // frag:2060-2090
  float tp = (700.0f - ro.y) / rd.y;
  if (tp > 0.0f) tmax = min_(tmax, tp);
  float dis = 0.0f, th = 0.0f, t = tmin, ot = t, odis = 0.0f;
#pragma unroll 1
  for (int i = 0; i < 400; i++) {
    th = 0.001f * t;
    V3 pos = madd(rd, t, ro);
    float hgt, slope;
    sdTerrain(pos.x, pos.z, hgt, slope, cnt);
    dis = pos.y - hgt;
    if (dis < th) break;
    ot = t;
    odis = dis;
    t = fma(dis * 0.8f, fma(-0.75f, slope, 1.0f), t);
    if (t > tmax) break;
  }
  if (t > tmax) return -1.0f;
  return ot + ((th - odis) * (t - ot)) / (dis - odis);
}
RM_DEV float terrainHeight(float x, float z, Counters &cnt) { float h, s; sdTerrain(x, z, h, s, cnt); return h; }
RM_DEV V3 terrainNormal(float px, float pz, Counters &cnt) {  // frag:2106-2111
  const float e = 0.03f;
  return normalize(v3(terrainHeight(px - e, pz - 0.0f, cnt) - terrainHeight(px + e, pz + 0.0f, cnt), 2.0f * e,
                      terrainHeight(px - 0.0f, pz - e, cnt) - terrainHeight(px + 0.0f, pz + e, cnt)));
}
RM_DEV float terrainShadow(V3 ro, V3 rd, float mint, Counters &cnt) {  // frag:2113-2125
  float res = 1.0f, t = mint;
#pragma unroll 1
  for (int i = 0; i < 32; i++) {
    V3 pos = madd(rd, t, ro);
    float hei = pos.y - terrainHeight(pos.x, pos.z, cnt);
    res = min_(res, divr_(32.0f * hei, t));
    if (res < 0.0001f || pos.y > 700.0f) break;
    t = t + clamp_(hei, fma(t, 0.1f, 2.0f), 100.0f);
  }
  return clamp_(res, 0.0f, 1.0f);
}
RM_DEV bool terrainRender(float iTime, V3 ro, V3 rd, float maxT, V3 bg, V3 &colOut, float &dOut, Counters &cnt) {  // frag:2128-2158
  colOut = bg; dOut = maxT;
  float res = raymarchTerrain(ro, rd, 15.0f, maxT, cnt);
  if (!(res > 0.0f)) return false;
  dOut = res;
  V3 p = madd(rd, res, ro);
  V3 pn = terrainNormal(p.x, p.z, cnt);
  V3 epos = v3(p.x + 0.0f, p.y + 4.8f, p.z + 0.0f);
  const V3 sunColor = getSunColor(), sunDir = getSunDir();
  float sha1 = terrainShadow(v3(p.x + 0.0f, p.y + 0.02f, p.z + 0.0f), sunDir, 0.02f, cnt);
  sha1 = sha1 * smoothstep_(-0.325f, -0.075f, cloudsShadowFlat(iTime, epos, sunDir, cnt));
  V4 fb = fbmd_8(v3(((p.x - 0.0f) * 0.15f) * 1.0f, ((p.y - 600.0f) * 0.15f) * 0.2f, ((p.z - 0.0f) * 0.15f) * 1.0f), cnt);
  float k = (0.8f * (1.0f - fabs_(pn.y))) * 0.8f;
  V3 nor = normalize(v3(fma(k, fb.y, pn.x), fma(k, fb.z, pn.y), fma(k, fb.w, pn.z)));
  V3 col = v3(0.18f * 0.85f, 0.12f * 0.85f, 0.10f * 0.85f);
  col = mix(col, v3(0.1f * 0.2f, 0.1f * 0.2f, 0.0f * 0.2f), smoothstep_(0.7f, 0.9f, nor.y));
  float dif = clamp_(dot(nor, sunDir), 0.0f, 1.0f) * sha1;
  float bac = clamp_(dot(normalize(v3(-sunDir.x, 0.0f, -sunDir.z)), nor), 0.0f, 1.0f);
  float foc = clamp_(RM_DIVR_CONST(p.y / 2.0f - 180.0f, 130.0f), 0.0f, 1.0f);
  float dom = clamp_(fma(0.5f, nor.y, 0.5f), 0.0f, 1.0f);
  V3 lin = mix(v3(0.1f * 0.1f, 0.1f * 0.2f, 0.1f * 0.1f), scale(sunColor, 3.0f), dom);
  lin = v3((0.2f * lin.x) * foc, (0.2f * lin.y) * foc, (0.2f * lin.z) * foc);
  lin = v3(fma(8.5f * sunColor.x, dif, lin.x), fma(8.5f * sunColor.y, dif, lin.y), fma(8.5f * sunColor.z, dif, lin.z));
  lin = v3(fma((0.27f * sunColor.x) * bac, foc, lin.x), fma((0.27f * sunColor.y) * bac, foc, lin.y),
           fma((0.27f * sunColor.z) * bac, foc, lin.z));
  colOut = mul(col, lin);
  return true;
}
// ---- night sky and sea (frag:476-516, 591-598, 1562-1573, 2160-2310): both read the `noise` texture -------------
RM_DEV float hashSin2(float px, float py) {  // frag:481-483
  return fract_(sin_(dot2(px, py, 12.9898f, 78.233f)) * 43758.5453f);
}
RM_DEV float hermite(float f) { return (f * f) * fma(-2.0f, f, 3.0f); }
RM_DEV float noiseW(float px, float py) {  // frag:504-518
  float ix = floor_(px), iy = floor_(py);
  float ux = hermite(fract_(px)), uy = hermite(fract_(py));
  float a = hashSin2(ix + 0.0f, iy + 0.0f), b = hashSin2(ix + 1.0f, iy + 0.0f);
  float c = hashSin2(ix + 0.0f, iy + 1.0f), d = hashSin2(ix + 1.0f, iy + 1.0f);
  return fma(2.0f, mix_(mix_(a, b, ux), mix_(c, d, ux), uy), -1.0f);
}
RM_DEV float noiseV(const RmTexture &noise, V3 x) {  // frag:591-598; textureLod(noise, ·, 0).yx
  V3 p = v3(floor_(x.x), floor_(x.y), floor_(x.z));
  V3 f = v3(hermite(fract_(x.x)), hermite(fract_(x.y)), hermite(fract_(x.z)));
  float u = fma(37.0f, p.z, p.x) + f.x, v = fma(239.0f, p.z, p.y) + f.y;
  V4 t = sampleRGBA8<false>(noise.pixels, noise.width, noise.height, (u + 0.5f) / 256.0f, (v + 0.5f) / 256.0f);
  return fma(mix_(t.y, t.x, f.z), 2.0f, -1.0f);
}
RM_DEV V3 getMoonColor(const RmTexture &noise, float iTime, V3 rd) {  // frag:1562-1573, MOON of frag:107
  const V3 MOON = normalize(v3(-0.4f, 0.4f, 0.3f));
  float ms = noiseV(noise, scale(rd, 20.0f));
  float q = (0.1f * ms) * ms;
  V3 mCol = v3(fma(-q, ms, 0.5f), fma(-q, ms, 0.5f), fma(-q, ms, 0.3f));
  float moonDot = dot(MOON, rd);
  V3 col = scale(mCol, smoothstep_(0.9985f, 0.999f, moonDot));
  float halo = smoothstep_(0.91f, 0.9985f, moonDot);
  col = v3(fma(0.15f, halo, col.x), fma(0.15f, halo, col.y), fma(0.15f, halo, col.z));
  float sh = 6.0f * sin_(iTime / 2.0f);
  float star = smoothstep_(0.99f, 0.999f, noiseV(noise, v3(floor_(fma(rd.x, 202.0f, -sh)), floor_(fma(rd.y, 202.0f, -sh)),
                                                          floor_(fma(rd.z, 202.0f, -sh)))));
  float sc = clamp_(star, 0.0f, 1.0f);
  return v3(fma(sc, 0.4f, col.x), fma(sc, 0.4f, col.y), fma(sc, 0.4f, col.z));
}
constexpr float kSeaHeight = 0.2f, kSeaChoppy = 1.0f, kSeaSpeed = 0.5f, kSeaFreq = 0.16f;  // frag:96-99
RM_DEV float sea_octave(float ux, float uy, float choppy) {  // frag:2162-2169
  float n = noiseW(ux, uy);
  ux += n; uy += n;
  float sx, cx, sy, cy;
  sincos_(ux, sx, cx);
  sincos_(uy, sy, cy);
  float wx = 1.0f - fabs_(sx), wy = 1.0f - fabs_(sy);
  wx = mix_(wx, fabs_(cx), wx); wy = mix_(wy, fabs_(cy), wy);
  return pow_(1.0f - pow_(wx * wy, 0.65f), choppy);
}
// seaMap (ITER_GEOMETRY = 3) / seaMapD (ITER_FRAGMENT = 5), frag:2195-2241; SEA_TIME of frag:2192.
RM_DEV float seaMap(float iTime, V3 p, int iters) {
  const float seaTime = fma(iTime, kSeaSpeed, 1.0f);
  float freq = kSeaFreq, amp = kSeaHeight, choppy = kSeaChoppy, ux = p.x, uy = p.z, h = 0.0f;
  for (int i = 0; i < iters; i++) {
    float d = sea_octave((ux + seaTime) * freq, (uy + seaTime) * freq, choppy);
    d += sea_octave((ux - seaTime) * freq, (uy - seaTime) * freq, choppy);
    h = fma(d, amp, h);
    float nx = dot2(ux, uy, 1.6f, 1.2f), ny = dot2(ux, uy, -1.2f, 1.6f);  // uv *= octave_m (row vector × mat2)
    ux = nx; uy = ny;
    freq *= 2.0f; amp *= 0.2f;
    choppy = mix_(choppy, 1.0f, 0.2f);
  }
  return p.y - h;
}
RM_DEV V3 getSeaNormal(float iTime, V3 p, float eps) {  // frag:2243-2250
  float ny = seaMap(iTime, p, 5);
  float nx = seaMap(iTime, v3(p.x + eps, p.y, p.z), 5) - ny;
  float nz = seaMap(iTime, v3(p.x, p.y, p.z + eps), 5) - ny;
  return normalize(v3(nx, eps, nz));
}
RM_DEV float seaMapHeight(float iTime, V3 ro, V3 rd, V3 &p, float maxT) {  // frag:2252-2282
  float tm = 0.0f, tx = 1000.0f;
  float hx = seaMap(iTime, madd(rd, tx, ro), 3);
  if (hx > 0.0f) { p = v3(0.0f, 0.0f, 0.0f); return tx; }
  float hm = seaMap(iTime, madd(rd, tm, ro), 3);
  float tmid = 0.0f;
  for (int i = 0; i < 8; i++) {
    float f = divr_(hm, hm - hx);
    tmid = mix_(tm, tx, f);
    p = madd(rd, tmid, ro);
    if (tmid > maxT) return -1.0f;
    float hmid = seaMap(iTime, p, 3);
    if (hmid < 0.0f) { tx = tmid; hx = hmid; } else { tm = tmid; hm = hmid; }
  }
  return tmid;
}
RM_DEV V3 getSeaColor(const RmTexture &noise, float iTime, V3 p, V3 n, V3 l, V3 eye, V3 dist) {  // frag:2171-2190
  const V3 base = v3(0.4f, 0.49f, 0.48f), water = v3(0.8f, 0.9f, 0.6f);  // frag:101-102
  float fresnel = clamp_(1.0f - dot(n, neg(eye)), 0.0f, 1.0f);
  fresnel = pow_(fresnel, 3.0f) * 0.65f;
  V3 refl = reflect(eye, n);
  V3 reflected = getMoonColor(noise, iTime, refl);
  float pw = pow_(fma(dot(n, l), 0.4f, 0.6f), 80.0f);
  V3 refracted = v3(fma(pw * water.x, 0.12f, base.x), fma(pw * water.y, 0.12f, base.y), fma(pw * water.z, 0.12f, base.z));
  V3 color = mix(refracted, reflected, fresnel);
  float atten = max_(fma(-dot(dist, dist), 0.001f, 1.0f), 0.0f);
  float dh = p.y - kSeaHeight;
  color = v3(fma((water.x * dh) * 0.18f, atten, color.x), fma((water.y * dh) * 0.18f, atten, color.y),
             fma((water.z * dh) * 0.18f, atten, color.z));
  const float nrm = (60.0f + 8.0f) / (3.14159265f * 8.0f);
  float spec = pow_(max_(dot(refl, l), 0.0f), 60.0f) * nrm;
  return v3(color.x + spec, color.y + spec, color.z + spec);
}
// frag:2284-2310; the bare `return;` of the miss path (UB7) returns colour = bg, d = maxT, no hit.
RM_DEV bool seaRender(const RmTexture &noise, float iTime, int W, V3 ro, V3 rd, float maxT, V3 bg, V3 &colOut, float &dOut) {
  colOut = bg; dOut = maxT;
  V3 p;
  float t = seaMapHeight(iTime, ro, rd, p, maxT);
  if (len(p) == 0.0f || t == -1.0f) return false;
  dOut = t;
  V3 d = sub(p, ro);
  V3 n = getSeaNormal(iTime, p, (dot(d, d) * 0.1f) / (float)W);
  V3 s = getSky(rd);
  V3 sc = getSeaColor(noise, iTime, p, n, getSunDir(), rd, d);
  float t2 = pow_(smoothstep_(0.0f, -0.05f, rd.y), 0.3f);
  colOut = fog(mix(s, sc, t2), t);
  return true;
}
// Sea, terrain, then cloud after a render() (frag:2444-2456, 2506-2518, 2555-2567).  The cloud layer is bounded by
// tr.d, which starts at the render's d — not at the sea's — when TERRAIN is off.
struct EnvOut { bool terrainHit, cloudHit, seaHit; V3 tcol, ccol, scol; };
RM_DEV EnvOut envLayers(uint32_t features, const RmTexture &noise, float iTime, int W, V3 ro, V3 rd, float d, V3 bg, Counters &cnt) {
  EnvOut e;
  float sd = d, td = d;
  e.terrainHit = false; e.cloudHit = false; e.seaHit = false;
  e.tcol = bg; e.ccol = bg; e.scol = bg;
  if (features & RM_FEAT_SEA) e.seaHit = seaRender(noise, iTime, W, ro, rd, d, bg, e.scol, sd);
  if (features & RM_FEAT_TERRAIN) e.terrainHit = terrainRender(iTime, ro, rd, sd, bg, e.tcol, td, cnt);
  if (features & RM_FEAT_CLOUD) e.ccol = cloudRender(iTime, ro, rd, bg, e.cloudHit, td, cnt);
  return e;
}

}  // namespace rm
