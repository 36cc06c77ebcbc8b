// rm_bulb_pipeline.hip.h — the single-Mandelbulb scene class as a wavefront pipeline of four kernels.
//
// Why: in the one-lane-per-pixel kernel (rm::render_kernel) a wave runs max-over-lanes of every nested loop
// (march steps × Mandelbulb iterations, then normals, then one shadow march per light); on the north-star
// frame only ≈27 % of the issued lane-slots did useful work.  Here the same per-pixel arithmetic (bit for
// bit — every formula is shared with rm_device.hip.h) is regrouped so that each kernel has one kind of work:
//
//   K1 bulb_primary_kernel  primary rays.  Persistent waves; each lane is a small state machine whose loop
//                           body is ONE Mandelbulb iteration (frag:786-798), so lanes at different march
//                           steps / iteration counts never wait for each other.  Lanes whose ray ended park;
//                           when ≥16 are parked (__ballot/popcount) the wave flushes them — misses store the
//                           background, hits are appended to the hit list — and refills them with the next
//                           pixels.  Pixel indices and hit-list slots are reserved per wave in chunks (512
//                           indices / 64 slots per atomic): a single device-scope counter sustains only ≈88
//                           atomics/µs, and one atomic per flush (≈10⁶ per 4K frame) made the first version
//                           of this kernel atomic-bound (7.2 ms).  Unused slots of a wave's last chunk are
//                           marked invalid (pix = −1) and skipped downstream.
//   K2 bulb_surface_kernel  one lane per HIT (dense waves): hit point, 4-tap normal, Perlin bump, AO.
//   K3 bulb_shadow_kernel   one lane per (hit, light) shadow ray, same iteration-level state machine and
//                           refill as K1.  Rays of lights with N·L <= 0.005 are not marched: getPhong
//                           discards their result (frag:1908-1912), so the frame is unchanged.
//   K4 bulb_shade_kernel    one lane per hit: Phong sum in light order, orbit-trap colour, float4 store.
//
// Intermediate records live in a per-device workspace in HBM (≤ 52 B per pixel + 8 B per shadow ray; a few
// hundred MB of traffic per 4K frame ≈ 0.1 ms at HBM speed).  Wave-coherent 8×8 pixel tiles are still the unit
// in which pixels are handed out (tile-major pixel cursor), so neighbouring lanes start on neighbouring rays.
#pragma once
#include "rm_device.hip.h"

namespace rm {

struct RowMap {
  int rowBegin, tileRows, shard, numShards;
  __host__ __device__ int frameRow(int r) const {
    return rowBegin + ((r / tileRows) * numShards + shard) * tileRows + (r % tileRows);
  }
};

// Device workspace of the pipeline (pointers into one allocation owned by the launcher).
struct BulbWs {
  uint32_t *counters;  // [0] pixel cursor, [1] number of hits, [2] shadow-ray cursor
  int *hitPix;         // packed output index r·W + x of each hit
  float4 *hitRec;      // (depth res.d, trap.y, trap.z, trap.w)           frag:1477, 800
  float4 *surfP;       // (p.xyz, ao)
  float4 *surfN;       // (bumped normal, unused)
  int2 *shadow;        // per ray [light·nHits + hit]: (intersectObj or -1, bits of the penumbra factor)
};

constexpr int kDefaultFlushThreshold = 16;
constexpr uint32_t kPixelChunk = 512;  // pixel indices (= 8 tiles) reserved per atomic on the pixel cursor
constexpr uint32_t kSlotChunk = 64;    // hit-list slots reserved per atomic
constexpr uint32_t kRayChunk = 1024;   // shadow rays reserved per atomic

// Wave-uniform broadcast of lane `src`'s value.
RM_DEV uint32_t bcast(uint32_t v, int src) { return (uint32_t)__shfl((int)v, src); }
// Reserve `n` units from a device counter, once per wave (lane 0 issues the atomic).
RM_DEV uint32_t waveReserve(uint32_t *counter, uint32_t n) {
  uint32_t base = 0;
  if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, n);
  return bcast(base, 0);
}

RM_DEV unsigned long long laneMaskLt() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// ---- Mandelbulb distance estimator as a resumable state (frag:775-803) ---------------------------------------
struct BulbDE {
  V3 w, c;
  float dz, m, ty, tz, tw;
  int it;
};
struct BulbParams {  // wave-uniform; read ONCE before the persistent loop so the loop body has no scalar loads
  float power, pexp, jx, jy, scale;
  float m00, m01, m02, m10, m11, m12, m20, m21, m22, m30, m31, m32;  // rows 0-2 of invModel, by column
  int iters;
  bool julia, angleSafe;
};
RM_DEV BulbParams bulbParams(const SceneBlock *sb) {
  BulbParams k;
  k.power = sb->g.power;
  k.pexp = (k.power - 1.0f) / 2.0f;
  k.jx = sb->g.juliaSeed[0];
  k.jy = sb->g.juliaSeed[1];
  k.julia = len2(k.jx, k.jy) != 0.0f;
  k.angleSafe = fabs_(k.power) < 1.0e6f;
  k.iters = sb->s.fractalIters;
  k.scale = sb->objs[0].scaleFactor;
  const float *M = sb->objs[0].invModel;
  k.m00 = M[0]; k.m01 = M[1]; k.m02 = M[2];
  k.m10 = M[4]; k.m11 = M[5]; k.m12 = M[6];
  k.m20 = M[8]; k.m21 = M[9]; k.m22 = M[10];
  k.m30 = M[12]; k.m31 = M[13]; k.m32 = M[14];
  return k;
}
RM_DEV void deStart(BulbDE &s, const BulbParams &k, V3 p) {
  V3 po = v3(fma(k.m20, p.z, fma(k.m10, p.y, fma(k.m00, p.x, k.m30))), fma(k.m21, p.z, fma(k.m11, p.y, fma(k.m01, p.x, k.m31))),
             fma(k.m22, p.z, fma(k.m12, p.y, fma(k.m02, p.x, k.m32))));  // frag:1417
  s.w = po;
  s.m = dot(po, po);
  s.ty = fabs_(po.y); s.tz = fabs_(po.z); s.tw = s.m;  // trap = vec4(abs(w), m), frag:778 (trap.x is never read)
  s.dz = 1.0f;
  s.c = k.julia ? v3(k.jx, k.jy, 0.0f) : po;
  s.it = 0;
}
// One iteration of frag:786-798; true when the loop ends (bailout or iteration cap).
RM_DEV bool deStep(BulbDE &s, const BulbParams &k) {
  s.dz = fma(k.power * pow_(s.m, k.pexp), s.dz, 1.0f);
  float r = sqrt_fast_(s.m);
  float b = k.power * acos_(divr_(s.w.y, r));
  float a = k.power * atan2_(s.w.x, s.w.z);
  float pr = pow_(r, k.power);
  float sb_, cb_, sa_, ca_;
  if (k.angleSafe) { sincos_inrange_(b, sb_, cb_); sincos_inrange_(a, sa_, ca_); }  // wave-uniform
  else { sincos_(b, sb_, cb_); sincos_(a, sa_, ca_); }
  s.w = v3(fma(pr, sb_ * sa_, s.c.x), fma(pr, cb_, s.c.y), fma(pr, sb_ * ca_, s.c.z));
  s.ty = min_(s.ty, fabs_(s.w.y));
  s.tz = min_(s.tz, fabs_(s.w.z));
  s.tw = min_(s.tw, s.m);
  s.m = dot(s.w, s.w);
  s.it++;
  return (s.m > 2.0f) || (s.it >= k.iters);
}
// frag:802 then sdScene's scale and nearest-object select (frag:1419-1423) for a one-object table.
RM_DEV float deDistance(const BulbDE &s, const BulbParams &k) {
  float d = divr_((0.25f * log_(s.m)) * sqrt_fast_(s.m), s.dz);
  float cur = d * k.scale;
  return (cur < 1000000.0f) ? cur : 1000000.0f;
}

// Pixel handed out by the tile-major cursor: 64 consecutive indices = one 8×8 tile.
RM_DEV bool decodePixel(uint32_t idx, int tilesX, int W, int nRows, int &x, int &r) {
  uint32_t tile = idx >> 6, l = idx & 63u;
  x = (int)(tile % (uint32_t)tilesX) * 8 + (int)(l & 7u);
  r = (int)(tile / (uint32_t)tilesX) * 8 + (int)(l >> 3);
  return x < W && r < nRows;
}

enum { ST_NEED = 0, ST_MARCH = 1, ST_HIT = 2, ST_MISS = 3, ST_DONE = 4 };

// ---- K1 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bulb_primary_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                            int nRows, float4 *__restrict__ out,
                                                            float4 *__restrict__ bright, BulbWs ws, int flushThreshold) {
  const int tilesX = (W + 7) >> 3, tilesY = (nRows + 7) >> 3;
  const uint32_t totalIdx = (uint32_t)tilesX * (uint32_t)tilesY * 64u;
  const BulbParams k = bulbParams(sb);
  const int maxSteps = sb->s.maxSteps;
  const float far = sb->cam.initialFar;
  const V3 bg = backgroundColor(sb);
  const unsigned long long lt = laneMaskLt();

  int st = ST_NEED, pix = -1, steps = 0;
  V3 ro = v3(0, 0, 0), rd = v3(0, 0, 0);
  float t = 0.0f, hitD = 0.0f;
  BulbDE de{};
  // wave-uniform cursors into the reserved chunks
  uint32_t pixCur = 0, pixEnd = 0, slotCur = 0, slotEnd = 0;
  bool exhausted = false;

  for (;;) {
    const unsigned long long mMarch = __ballot(st == ST_MARCH);
    const unsigned long long mWait = __ballot(st == ST_NEED || st == ST_HIT || st == ST_MISS);
    if (mMarch == 0 ? (mWait != 0) : (__popcll(mWait) >= flushThreshold)) {
      // ---- flush: write results of parked lanes, then refill waiting lanes ----
      if (st == ST_MISS) {  // frag:2325, 2465
        out[pix] = make_float4(bg.x, bg.y, bg.z, 1.0f);
        if (bright) bright[pix] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        st = ST_NEED;
      }
      const unsigned long long mHit = __ballot(st == ST_HIT);
      if (mHit) {
        const uint32_t n = (uint32_t)__popcll(mHit), avail = slotEnd - slotCur;
        uint32_t fresh = 0;
        if (n > avail) fresh = waveReserve(&ws.counters[1], kSlotChunk);  // old chunk gets exactly filled
        if (st == ST_HIT) {
          const uint32_t rank = (uint32_t)__popcll(mHit & lt);
          const uint32_t slot = (rank < avail) ? (slotCur + rank) : (fresh + (rank - avail));
          ws.hitPix[slot] = pix;
          ws.hitRec[slot] = make_float4(hitD, de.ty, de.tz, de.tw);
          st = ST_NEED;
        }
        if (n > avail) { slotCur = fresh + (n - avail); slotEnd = fresh + kSlotChunk; }
        else slotCur += n;
      }
      // refill from the wave's pixel chunk
      if (pixCur == pixEnd && !exhausted) {
        pixCur = waveReserve(&ws.counters[0], kPixelChunk);
        pixEnd = pixCur + kPixelChunk;
        if (pixCur >= totalIdx) { exhausted = true; pixEnd = pixCur; }
        else if (pixEnd > totalIdx) pixEnd = totalIdx;
      }
      {
        const unsigned long long mNeed = __ballot(st == ST_NEED);
        const uint32_t avail = pixEnd - pixCur, n = (uint32_t)__popcll(mNeed);
        if (st == ST_NEED) {
          const uint32_t rank = (uint32_t)__popcll(mNeed & lt);
          if (rank < avail) {
            int x, r;
            if (decodePixel(pixCur + rank, tilesX, W, nRows, x, r)) {
              pix = r * W + x;
              primaryRay(sb, x, map.frameRow(r), W, H, ro, rd);
              t = 0.0f;
              steps = 0;
              deStart(de, k, madd(rd, t, ro));
              st = ST_MARCH;
            }  // else: padding lane of an edge tile, stays NEED
          } else if (exhausted) {
            st = ST_DONE;
          }
        }
        pixCur += (n < avail) ? n : avail;
      }
      continue;
    }
    if (mMarch == 0) break;  // nothing marching, nothing waiting: every lane is DONE
    if (st == ST_MARCH) {
      if (deStep(de, k)) {  // this lane's sdScene evaluation is complete → one step of frag:1459-1470
        const float d = deDistance(de, k);
        const bool hit = fabs_(d) < kSurfaceDist;
        if (hit || t > far) {
          hitD = t - d;  // frag:1477
          st = hit ? ST_HIT : ST_MISS;
        } else {
          t = fma(d, 1.0f, t);
          steps++;
          if (steps >= maxSteps) st = ST_MISS;
          else deStart(de, k, madd(rd, t, ro));
        }
      }
    }
  }
  // unused slots of this wave's last chunk are holes
  for (uint32_t sl = slotCur + (threadIdx.x & 63); sl < slotEnd; sl += 64) ws.hitPix[sl] = -1;
}

// ---- K2 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bulb_surface_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                            BulbWs ws) {
  const uint32_t nHits = ws.counters[1];  // reserved slots (multiple of kSlotChunk); holes have pix < 0
  Counters cnt{0, 0, 0, 0, 0, 0};
  for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < nHits; h += gridDim.x * blockDim.x) {
    const int pix = ws.hitPix[h];
    if (pix < 0) continue;
    const float4 rec = ws.hitRec[h];
    const int r = pix / W, x = pix - r * W;
    V3 ro, rd;
    primaryRay(sb, x, map.frameRow(r), W, H, ro, rd);
    const V3 p = madd(rd, rec.x, ro);                    // frag:2333
    V3 n = getNormal<true, false>(sb, p, cnt);           // frag:1436-1444
    if (sb->s.features & RM_FEAT_PERLIN_BUMP) n = bumpNormal(n, p);  // frag:2334-2336
    float ao = 1.0f;
    if (sb->s.enableAmbientOcclusion) ao = calcAO<true, false>(sb, p, n, cnt);  // frag:1859
    ws.surfP[h] = make_float4(p.x, p.y, p.z, ao);
    ws.surfN[h] = make_float4(n.x, n.y, n.z, 0.0f);
  }
}

// ---- K3 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bulb_shadow_kernel(const SceneBlock *__restrict__ sb, BulbWs ws, int flushThreshold) {
  const uint32_t nHits = ws.counters[1];
  const uint32_t nl = (uint32_t)sb->numLights;
  const uint32_t totalRays = nHits * nl;
  const BulbParams k = bulbParams(sb);
  const int maxSteps = sb->s.maxSteps;
  const float far = sb->cam.initialFar;
  const unsigned long long lt = laneMaskLt();

  int st = ST_NEED, steps = 0;
  uint32_t ray = 0;
  V3 so = v3(0, 0, 0), L = v3(0, 0, 0);
  float t = 0.0f, maxT = 0.0f, pen = 1.0f;
  BulbDE de{};
  uint32_t rayCur = 0, rayEnd = 0;
  bool exhausted = false;

  for (;;) {
    const unsigned long long mMarch = __ballot(st == ST_MARCH);
    const unsigned long long mWait = __ballot(st == ST_NEED);
    if (mMarch == 0 ? (mWait != 0) : (__popcll(mWait) >= flushThreshold)) {
      if (rayCur == rayEnd && !exhausted) {
        rayCur = waveReserve(&ws.counters[2], kRayChunk);
        rayEnd = rayCur + kRayChunk;
        if (rayCur >= totalRays) { exhausted = true; rayEnd = rayCur; }
        else if (rayEnd > totalRays) rayEnd = totalRays;
      }
      const uint32_t avail = rayEnd - rayCur, n = (uint32_t)__popcll(mWait);
      if (st == ST_NEED) {
        const uint32_t rank = (uint32_t)__popcll(mWait & lt);
        if (rank < avail) {
          ray = rayCur + rank;
          const uint32_t li = ray / nHits, h = ray - li * nHits;  // light-major: a wave marches toward one light
          if (ws.hitPix[h] >= 0) {
            const float4 P = ws.surfP[h], Nn = ws.surfN[h];
            const V3 p = v3(P.x, P.y, P.z), N = v3(Nn.x, Nn.y, Nn.z);
            // li differs across a wave only at the boundary between two lights' ray ranges → per-lane table read
            const LightGeom g = lightSetup(sb->lights[li], p, far);
            if (dot(N, g.L) <= 0.005f) {
              // getPhong skips this light whatever the shadow march returns (frag:1912): do not march.
              ws.shadow[ray] = make_int2(-1, (int)f2u(1.0f));
            } else {
              so = shadowOrigin(p, N);
              L = g.L;
              maxT = g.maxT;
              t = 0.0f;
              pen = 1.0f;
              steps = 0;
              deStart(de, k, madd(L, t, so));
              st = ST_MARCH;
            }
          }  // hole: nothing to do, stays NEED
        } else if (exhausted) {
          st = ST_DONE;
        }
      }
      rayCur += (n < avail) ? n : avail;
      continue;
    }
    if (mMarch == 0) break;
    if (st == ST_MARCH) {
      if (deStep(de, k)) {  // one step of softshadow, frag:1708-1714
        const float d = deDistance(de, k);
        const bool hit = fabs_(d) < kSurfaceDist;
        bool end = hit || t > maxT;
        if (!end) {
          pen = min_(pen, divr_(8.0f * d, t));
          t = t + fabs_(d);
          steps++;
          if (steps >= maxSteps) end = true;
          else deStart(de, k, madd(L, t, so));
        }
        if (end) {
          ws.shadow[ray] = make_int2(hit ? 0 : -1, (int)f2u(pen));
          st = ST_NEED;
        }
      }
    }
  }
}

// ---- K4 ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bulb_shade_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                          float4 *__restrict__ out, float4 *__restrict__ bright,
                                                          BulbWs ws) {
  const uint32_t nHits = ws.counters[1];
  const int nl = sb->numLights;
  const float ka = sb->g.ka, kd = sb->g.kd, ks = sb->g.ks;
  (void)kd;
  const float far = sb->cam.initialFar;
  const bool soft = sb->s.enableSoftShadow != 0;
  const RmObject &o = sb->objs[0];
  Material mat;
  mat.amb = v3(o.cAmbient[0], o.cAmbient[1], o.cAmbient[2]);
  mat.dif = v3(sb->g.kd * o.cDiffuse[0], sb->g.kd * o.cDiffuse[1], sb->g.kd * o.cDiffuse[2]);  // getDiffuse, untextured
  mat.spec = v3(o.cSpecular[0], o.cSpecular[1], o.cSpecular[2]);
  mat.shininess = o.shininess;
  for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < nHits; h += gridDim.x * blockDim.x) {
    const int pix = ws.hitPix[h];
    if (pix < 0) continue;
    const float4 rec = ws.hitRec[h], P = ws.surfP[h], Nn = ws.surfN[h];
    const int r = pix / W, x = pix - r * W;
    V3 ro, rd;
    primaryRay(sb, x, map.frameRow(r), W, H, ro, rd);
    const V3 p = v3(P.x, P.y, P.z), N = v3(Nn.x, Nn.y, Nn.z);
    const float ao = P.w;
    V3 total = v3((mat.amb.x * ka) * ao, (mat.amb.y * ka) * ao, (mat.amb.z * ka) * ao);  // frag:1860
    const V3 V = normalize(neg(rd));
    for (int i = 0; i < nl; i++) {
      const RmLight &li = sb->lights[i];
      const LightGeom g = lightSetup(li, p, far);
      const int2 sh = ws.shadow[(uint32_t)i * nHits + h];
      V3 cur;
      if (lightTerm(li, g, mat, N, V, ks, sh.x, u2f((uint32_t)sh.y), soft, cur)) total = add(total, cur);
    }
    const V3 c = bulbTrapColor(rec.y, rec.z, rec.w);  // frag:2356-2360
    const V3 col = v3(c.x * (total.x * 8.0f), c.y * (total.y * 8.0f), c.z * (total.z * 8.0f));  // frag:2361
    // frag:2572: phong + refl + refr with refl = refr = 0
    const V3 fc = v3((col.x + 0.0f) + 0.0f, (col.y + 0.0f) + 0.0f, (col.z + 0.0f) + 0.0f);
    out[pix] = make_float4(fc.x, fc.y, fc.z, (1.0f + 0.0f) + 0.0f);
    if (bright) {
      const float lum = dot(fc, v3(0.2126f, 0.7152f, 0.0722f));  // frag:1938-1946
      bright[pix] = (lum > 1.0f) ? make_float4(fc.x, fc.y, fc.z, 1.0f) : make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    }
  }
}

// =====================================================================================================
// Variant B: the same four stages with plain nested loops (no per-lane state machine).  Stage 1 keeps the
// one-lane-per-pixel 8×8 tile mapping of rm::render_kernel but only marches the primary ray; hits are
// compacted (block-aggregated atomic), so stages 2-4 run on dense waves, and stage 2 also compacts, per
// light, the shadow rays that can matter (N·L > 0.005).
// =====================================================================================================
struct BulbWsB {
  uint32_t *counters;  // [1] hits, [4+i] shadow rays of light i
  int *hitPix;
  float4 *hitRec, *surfP, *surfN;
  int2 *shadow;        // [light·cap + hit]
  uint32_t *rayHit;    // [light·cap + k] → hit index of the k-th marched ray of that light
  uint32_t cap;        // hit capacity (pixels of the launch)
};

// Block-aggregated append: returns this lane's slot (valid only where `want`), one atomic per block.
RM_DEV uint32_t blockAppend(bool want, uint32_t *counter, uint32_t *ldsScratch /* >= 5 words */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(want);
  if (lane == 0) ldsScratch[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t total = ldsScratch[0] + ldsScratch[1] + ldsScratch[2] + ldsScratch[3];
    ldsScratch[4] = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  uint32_t base = ldsScratch[4];
  for (int w = 0; w < wave; w++) base += ldsScratch[w];
  const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();  // scratch may be reused by the next call
  return slot;
}

__global__ __launch_bounds__(256) void bulbB_primary_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                             int nRows, float4 *__restrict__ out,
                                                             float4 *__restrict__ bright, BulbWsB ws) {
  __shared__ uint32_t scratch[8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int x = blockIdx.x * 32 + wave * 8 + (lane & 7);
  const int r = blockIdx.y * 8 + (lane >> 3);
  const bool inside = x < W && r < nRows;
  bool hit = false;
  MarchRes res;
  res.obj = -1; res.d = 0.0f; res.trap = v4(0.0f, 0.0f, 0.0f, 0.0f);
  if (inside) {
    V3 ro, rd;
    primaryRay(sb, x, map.frameRow(r), W, H, ro, rd);
    Counters cnt{0, 0, 0, 0, 0, 0};
    res = march<true, false, false>(sb, ro, rd, sb->cam.initialFar, 1.0f, cnt);  // frag:2322
    hit = res.obj != -1;
    if (!hit) {
      const V3 bg = backgroundColor(sb);
      const size_t o = (size_t)r * W + x;
      out[o] = make_float4(bg.x, bg.y, bg.z, 1.0f);
      if (bright) bright[o] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    }
  }
  const uint32_t slot = blockAppend(hit, &ws.counters[1], scratch);
  if (hit) {
    ws.hitPix[slot] = r * W + x;
    ws.hitRec[slot] = make_float4(res.d, res.trap.y, res.trap.z, res.trap.w);
  }
}

__global__ __launch_bounds__(256) void bulbB_surface_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                             BulbWsB ws) {
  __shared__ uint32_t scratch[8];
  const uint32_t nHits = ws.counters[1];
  const int nl = sb->numLights;
  const float far = sb->cam.initialFar;
  Counters cnt{0, 0, 0, 0, 0, 0};
  const uint32_t stride = gridDim.x * blockDim.x;
  // every thread of a block runs the same number of trips so that the block-level appends stay collective
  for (uint32_t h0 = blockIdx.x * blockDim.x; h0 < nHits; h0 += stride) {
    const uint32_t h = h0 + threadIdx.x;
    const bool live = h < nHits;
    V3 p = v3(0, 0, 0), n = v3(0, 1, 0);
    if (live) {
      const int pix = ws.hitPix[h];
      const float4 rec = ws.hitRec[h];
      const int r = pix / W, x = pix - r * W;
      V3 ro, rd;
      primaryRay(sb, x, map.frameRow(r), W, H, ro, rd);
      p = madd(rd, rec.x, ro);                                             // frag:2333
      n = getNormal<true, false>(sb, p, cnt);                              // frag:1436-1444
      if (sb->s.features & RM_FEAT_PERLIN_BUMP) n = bumpNormal(n, p);      // frag:2334-2336
      float ao = 1.0f;
      if (sb->s.enableAmbientOcclusion) ao = calcAO<true, false>(sb, p, n, cnt);  // frag:1859
      ws.surfP[h] = make_float4(p.x, p.y, p.z, ao);
      ws.surfN[h] = make_float4(n.x, n.y, n.z, 0.0f);
    }
    for (int i = 0; i < nl; i++) {
      bool want = false;
      if (live) {
        const LightGeom g = lightSetup(sb->lights[i], p, far);
        want = !(dot(n, g.L) <= 0.005f);  // otherwise getPhong drops the light whatever the march finds (frag:1912)
        if (!want) ws.shadow[(uint32_t)i * ws.cap + h] = make_int2(-1, (int)f2u(1.0f));
      }
      const uint32_t slot = blockAppend(want, &ws.counters[4 + i], scratch);
      if (want) ws.rayHit[(uint32_t)i * ws.cap + slot] = h;
    }
  }
}

__global__ __launch_bounds__(256) void bulbB_shadow_kernel(const SceneBlock *__restrict__ sb, BulbWsB ws) {
  const int nl = sb->numLights;
  const float far = sb->cam.initialFar;
  Counters cnt{0, 0, 0, 0, 0, 0};
  const uint32_t stride = gridDim.x * blockDim.x;
  for (int i = 0; i < nl; i++) {
    const uint32_t nRays = ws.counters[4 + i];
    const RmLight &li = sb->lights[i];  // wave-uniform
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < nRays; j += stride) {
      const uint32_t h = ws.rayHit[(uint32_t)i * ws.cap + j];
      const float4 P = ws.surfP[h], Nn = ws.surfN[h];
      const V3 p = v3(P.x, P.y, P.z), N = v3(Nn.x, Nn.y, Nn.z);
      const LightGeom g = lightSetup(li, p, far);
      const MarchRes sh = march<true, false, true>(sb, shadowOrigin(p, N), g.L, g.maxT, 1.0f, cnt);  // frag:1908
      ws.shadow[(uint32_t)i * ws.cap + h] = make_int2(sh.obj, (int)f2u(sh.d));
    }
  }
}

__global__ __launch_bounds__(256) void bulbB_shade_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                           float4 *__restrict__ out, float4 *__restrict__ bright,
                                                           BulbWsB ws) {
  const uint32_t nHits = ws.counters[1];
  const int nl = sb->numLights;
  const float ka = sb->g.ka, kd = sb->g.kd, ks = sb->g.ks;
  (void)kd;
  const float far = sb->cam.initialFar;
  const bool soft = sb->s.enableSoftShadow != 0;
  const RmObject &o = sb->objs[0];
  Material mat;
  mat.amb = v3(o.cAmbient[0], o.cAmbient[1], o.cAmbient[2]);
  mat.dif = v3(sb->g.kd * o.cDiffuse[0], sb->g.kd * o.cDiffuse[1], sb->g.kd * o.cDiffuse[2]);  // getDiffuse, untextured
  mat.spec = v3(o.cSpecular[0], o.cSpecular[1], o.cSpecular[2]);
  mat.shininess = o.shininess;
  for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < nHits; h += gridDim.x * blockDim.x) {
    const int pix = ws.hitPix[h];
    const float4 rec = ws.hitRec[h], P = ws.surfP[h], Nn = ws.surfN[h];
    const int r = pix / W, x = pix - r * W;
    V3 ro, rd;
    primaryRay(sb, x, map.frameRow(r), W, H, ro, rd);
    const V3 p = v3(P.x, P.y, P.z), N = v3(Nn.x, Nn.y, Nn.z);
    const float ao = P.w;
    V3 total = v3((mat.amb.x * ka) * ao, (mat.amb.y * ka) * ao, (mat.amb.z * ka) * ao);  // frag:1860
    const V3 V = normalize(neg(rd));
    for (int i = 0; i < nl; i++) {
      const RmLight &li = sb->lights[i];
      const LightGeom g = lightSetup(li, p, far);
      const int2 sh = ws.shadow[(uint32_t)i * ws.cap + h];
      V3 cur;
      if (lightTerm(li, g, mat, N, V, ks, sh.x, u2f((uint32_t)sh.y), soft, cur)) total = add(total, cur);
    }
    const V3 c = bulbTrapColor(rec.y, rec.z, rec.w);  // frag:2356-2360
    const V3 col = v3(c.x * (total.x * 8.0f), c.y * (total.y * 8.0f), c.z * (total.z * 8.0f));  // frag:2361
    const V3 fc = v3((col.x + 0.0f) + 0.0f, (col.y + 0.0f) + 0.0f, (col.z + 0.0f) + 0.0f);      // frag:2572
    out[pix] = make_float4(fc.x, fc.y, fc.z, (1.0f + 0.0f) + 0.0f);
    if (bright) {
      const float lum = dot(fc, v3(0.2126f, 0.7152f, 0.0722f));  // frag:1938-1946
      bright[pix] = (lum > 1.0f) ? make_float4(fc.x, fc.y, fc.z, 1.0f) : make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    }
  }
}

// =====================================================================================================
// Variant C: variant B with the two march stages cut into passes with a step budget.  A ray that has not
// ended when its budget is used up is appended (block-aggregated) to a continuation queue with its state
// (depth, step count, penumbra factor) and resumes in the next launch, packed densely with the other survivors.
// The march state between two steps of frag:1459-1470 / 1708-1714 is exactly (t, i[, res]), so resuming is exact.
// Lanes therefore idle for at most `budget` steps per pass instead of for the longest march of their wave.
// =====================================================================================================
struct BulbWsC {
  BulbWsB b;
  // primary continuation queues (ping-pong): packed output index, depth, steps done
  int *pPix[2]; float *pT[2]; int *pSteps[2];
  // shadow continuation queues (ping-pong): ray = light·cap + hit, depth, penumbra factor, steps done
  uint32_t *sRay[2]; float *sT[2]; float *sPen[2]; int *sSteps[2];
};
constexpr int kCntPrimary = 8;   // counters[8 + pass]  : rays queued FOR primary pass `pass` (pass >= 1)
constexpr int kCntShadow = 24;   // counters[24 + pass] : rays queued FOR shadow pass `pass` (pass >= 1)

// One budgeted stretch of raymarch() (frag:1459-1470).  Returns 0 = budget used up, 1 = ended without a hit,
// 2 = hit (res filled).
RM_DEV int marchBudget(const SceneBlock *sb, V3 ro, V3 rd, float end, int maxSteps, int budget, float &t, int &steps,
                       MarchRes &res) {
  Counters cnt{0, 0, 0, 0, 0, 0};
  for (int local = 0; ; ) {
    if (steps >= maxSteps) return 1;
    const SceneMin c = sdScene<true, false>(sb, madd(rd, t, ro), cnt);
    const bool hit = fabs_(c.d) < kSurfaceDist;
    if (hit || t > end) {
      res.obj = hit ? c.idx : -1;
      res.d = t - c.d;  // frag:1477
      res.trap = c.trap;
      return hit ? 2 : 1;
    }
    t = fma(c.d, 1.0f, t);
    steps++;
    if (++local >= budget) return 0;
  }
}

template <bool FIRST>
__global__ __launch_bounds__(256) void bulbC_primary_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                             int nRows, float4 *__restrict__ out,
                                                             float4 *__restrict__ bright, BulbWsC ws, int pass, int budget) {
  __shared__ uint32_t scratch[8];
  const int maxSteps = sb->s.maxSteps;
  const float far = sb->cam.initialFar;
  const int inQ = (pass + 1) & 1, outQ = pass & 1;  // pass p reads queue (p+1)&1 (written by pass p-1) and writes p&1
  const uint32_t nIn = FIRST ? 0u : ws.b.counters[kCntPrimary + pass];
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t trips = FIRST ? 1u : (nIn + stride - 1) / stride;
  for (uint32_t trip = 0; trip < trips; trip++) {
    bool live;
    int pix = 0, steps = 0;
    float t = 0.0f;
    if (FIRST) {
      const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
      const int x = blockIdx.x * 32 + wave * 8 + (lane & 7), r = blockIdx.y * 8 + (lane >> 3);
      live = x < W && r < nRows;
      pix = r * W + x;
    } else {
      const uint32_t j = trip * stride + blockIdx.x * blockDim.x + threadIdx.x;
      live = j < nIn;
      if (live) { pix = ws.pPix[inQ][j]; t = ws.pT[inQ][j]; steps = ws.pSteps[inQ][j]; }
    }
    int state = 1;
    MarchRes res;
    res.obj = -1; res.d = 0.0f; res.trap = v4(0.0f, 0.0f, 0.0f, 0.0f);
    if (live) {
      const int r = pix / W, x = pix - r * W;
      V3 ro, rd;
      primaryRay(sb, x, map.frameRow(r), W, H, ro, rd);
      state = marchBudget(sb, ro, rd, far, maxSteps, budget, t, steps, res);
      if (state == 1) {
        const V3 bg = backgroundColor(sb);
        out[pix] = make_float4(bg.x, bg.y, bg.z, 1.0f);
        if (bright) bright[pix] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
      }
    }
    const bool isHit = live && state == 2, cont = live && state == 0;
    const uint32_t hs = blockAppend(isHit, &ws.b.counters[1], scratch);
    if (isHit) {
      ws.b.hitPix[hs] = pix;
      ws.b.hitRec[hs] = make_float4(res.d, res.trap.y, res.trap.z, res.trap.w);
    }
    const uint32_t cs = blockAppend(cont, &ws.b.counters[kCntPrimary + pass + 1], scratch);
    if (cont) { ws.pPix[outQ][cs] = pix; ws.pT[outQ][cs] = t; ws.pSteps[outQ][cs] = steps; }
  }
}

// One budgeted stretch of softshadow() (frag:1708-1714).  Returns 0 = budget used up, 1 = ended, hit flag in `hit`.
RM_DEV int shadowBudget(const SceneBlock *sb, V3 so, V3 L, float maxT, int maxSteps, int budget, float &t, float &pen,
                        int &steps, bool &hit) {
  Counters cnt{0, 0, 0, 0, 0, 0};
  for (int local = 0; ; ) {
    hit = false;
    if (steps >= maxSteps) return 1;
    const SceneMin c = sdScene<true, false>(sb, madd(L, t, so), cnt);
    hit = fabs_(c.d) < kSurfaceDist;
    if (hit || t > maxT) return 1;
    pen = min_(pen, divr_(8.0f * c.d, t));
    t = t + fabs_(c.d);
    steps++;
    if (++local >= budget) return 0;
  }
}

template <bool FIRST>
__global__ __launch_bounds__(256) void bulbC_shadow_kernel(const SceneBlock *__restrict__ sb, BulbWsC ws, int pass, int budget) {
  __shared__ uint32_t scratch[8];
  const int maxSteps = sb->s.maxSteps;
  const float far = sb->cam.initialFar;
  const int nl = sb->numLights;
  const int inQ = (pass + 1) & 1, outQ = pass & 1;
  const uint32_t stride = gridDim.x * blockDim.x;
  // FIRST: the per-light lists stage 2 emitted; otherwise the continuation queue of the previous pass
  const int lists = FIRST ? nl : 1;
  for (int li0 = 0; li0 < lists; li0++) {
    const uint32_t nIn = FIRST ? ws.b.counters[4 + li0] : ws.b.counters[kCntShadow + pass];
    const uint32_t trips = (nIn + stride - 1) / stride;
    for (uint32_t trip = 0; trip < trips; trip++) {
      const uint32_t j = trip * stride + blockIdx.x * blockDim.x + threadIdx.x;
      const bool live = j < nIn;
      uint32_t ray = 0;
      float t = 0.0f, pen = 1.0f;
      int steps = 0, state = 1;
      bool hit = false;
      if (live) {
        if (FIRST) ray = (uint32_t)li0 * ws.b.cap + ws.b.rayHit[(uint32_t)li0 * ws.b.cap + j];
        else { ray = ws.sRay[inQ][j]; t = ws.sT[inQ][j]; pen = ws.sPen[inQ][j]; steps = ws.sSteps[inQ][j]; }
        const uint32_t li = ray / ws.b.cap, h = ray - li * ws.b.cap;
        const float4 P = ws.b.surfP[h], Nn = ws.b.surfN[h];
        const V3 p = v3(P.x, P.y, P.z), N = v3(Nn.x, Nn.y, Nn.z);
        const LightGeom g = lightSetup(sb->lights[li], p, far);
        state = shadowBudget(sb, shadowOrigin(p, N), g.L, g.maxT, maxSteps, budget, t, pen, steps, hit);
        if (state == 1) ws.b.shadow[ray] = make_int2(hit ? 0 : -1, (int)f2u(pen));
      }
      const bool cont = live && state == 0;
      const uint32_t cs = blockAppend(cont, &ws.b.counters[kCntShadow + pass + 1], scratch);
      if (cont) { ws.sRay[outQ][cs] = ray; ws.sT[outQ][cs] = t; ws.sPen[outQ][cs] = pen; ws.sSteps[outQ][cs] = steps; }
    }
  }
}

}  // namespace rm
