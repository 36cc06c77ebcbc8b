// rm_wavefront.hip.h — the table-walk scene classes (primitives, Menger sponge, Sierpinski) as a wavefront pipeline.
//
// Why.  In the one-lane-per-pixel kernel (rm::render_kernel) a wave is an 8×8 pixel tile and every march of main()'s
// call tree — primary ray, one shadow ray per light, and the same again for every reflection bounce (frag:2491-2524) —
// runs for as long as its LONGEST lane: on the 8K Menger frame with two bounces (BASELINE configs[4]) 35 % of the lanes of
// an issued instruction are live, on lighting/reflections_complex.json 64 % (profiles/r02_n_configs.md).  An evaluation of
// these scenes costs the same on every lane (no data-dependent inner loop, unlike the Mandelbulb's iteration count), so
// regrouping rays loses nothing.  The pipeline keeps the per-ray arithmetic of rm_device.hip.h bit for bit and regroups it:
//
//   per generation g (0 = primary rays, g >= 1 = the g-th reflection bounce):
//     wf_march_kernel<0|1>  persistent waves; a lane is a ray, the loop body is ONE sdScene evaluation = one step of
//                           raymarch (frag:1453-1484).  A lane whose ray ends stores its result at once — a miss completes
//                           its pixel, a hit is appended to the generation's hit list — and waits; when >= T lanes wait
//                           (__ballot / popcount) the wave refills them from the source (pixel cursor / ray queue).
//     wf_surface_kernel     one lane per hit (dense): hit point, 4-tap normal, Perlin bump, ambient occlusion.
//     wf_march_kernel<2>    the same state machine over the shadow rays (hit × light) of softshadow (frag:1703-1725);
//                           rays of lights that N·L <= 0.005 drops are never marched (frag:1912 ignores their result).
//     wf_light_kernel       one lane per hit: Phong sum in light order, fractal palette; generation 0 opens the pixel's
//                           path record, later ones add into its reflection term; a path that bounces again appends its
//                           next ray, otherwise the pixel is complete and stored.
//
// Records live in a per-(device, stream) workspace in HBM.  Cursors are reserved per wave in chunks (one device atomic per
// 256 rays / pixels / hit slots, RM_WF_*_CHUNK below); unused slots of a wave's last chunk are holes (src < 0) that the dense
// kernels skip.
// Not covered (the launcher keeps those on rm::render_kernel): procedural layers, samplers (textures, sky box, area
// lights), refraction, scenes with a Mandelbulb or the 2-D Mandelbrot (data-dependent evaluation cost), counting modes.
#pragma once
#include "rm_device.hip.h"

namespace rm {

// ---- wave / block helpers of the queues ----------------------------------------------------------------------------------
// Reserve `n` units from a device counter, once per wave (lane 0 issues the atomic, the base is broadcast).
RM_DEV uint32_t waveReserve(uint32_t *counter, uint32_t n) {
  uint32_t base = 0;
  if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, n);
  return (uint32_t)__shfl((int)base, 0);
}
RM_DEV unsigned long long laneMaskLt() { return (1ull << (threadIdx.x & 63)) - 1ull; }
enum { ST_NEED = 0, ST_MARCH = 1, ST_HIT = 2, ST_MISS = 3, ST_DONE = 4 };  // a lane of a persistent march wave
// Pixel handed out by the tile-major cursor: 64 consecutive indices = one 8×8 tile (neighbouring lanes start on neighbouring rays).
RM_DEV bool decodePixel(uint32_t idx, int tilesX, int W, int nRows, int &x, int &r) {
  uint32_t tile = idx >> 6, l = idx & 63u;
  x = (int)(tile % (uint32_t)tilesX) * 8 + (int)(l & 7u);
  r = (int)(tile / (uint32_t)tilesX) * 8 + (int)(l >> 3);
  return x < W && r < nRows;
}
// Block-aggregated append (256-thread blocks): returns this lane's slot (valid only where `want`), one atomic per block.
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

constexpr int kWfMaxBounces = 7;
constexpr uint32_t kWfShadowHit = 0x7fc00001u;  // shadow result of a ray that hit: a NaN (no penumbra factor is one)
enum { WF_SRC = 0, WF_HITS = 1, WF_SHADOW = 2, WF_NEXT = 3, WF_STRIDE = 8 };  // counters of one generation

struct WfWs {
  uint32_t *counters;  // [WF_STRIDE·g + …]: source cursor, hit slots reserved, shadow-ray cursor, rays appended for g + 1
  float4 *rayO[2];     // rays of generation g >= 1 live in buffer g & 1: (origin, path id bits) …
  float4 *rayD[2];     // … (direction, unused)
  int4 *hit;           // per hit slot: (src = pixel index (g = 0) or ray index; < 0 = hole, bits of res.d, object, bits of trap.z)
  float4 *surfP;       // (p, ambient occlusion)
  float4 *surfN;       // (bumped normal, unused)
  float *shadow;       // [light·cap + hit slot]: penumbra factor of a ray that missed, kWfShadowHit of one that hit
  int2 *pathPix;       // paths are indexed by the generation-0 hit slot: (pixel index, object of the primary hit)
  float4 *pathA;       // (phong.xyz, refl.w)
  float4 *pathB;       // (refl.xyz, fil.x)
  float2 *pathC;       // (fil.y, fil.z)
  uint32_t cap;        // hit-slot capacity
};

// frag:2572-2574 with refr = 0: fragColor = phong + refl + refr, bright pass (frag:1938-1946)
RM_DEV void wfStorePixel(float4 *__restrict__ out, float4 *__restrict__ bright, int pix, V3 phong, V4 refl) {
  const V3 c = v3((phong.x + refl.x) + 0.0f, (phong.y + refl.y) + 0.0f, (phong.z + refl.z) + 0.0f);
  out[pix] = make_float4(c.x, c.y, c.z, (1.0f + refl.w) + 0.0f);
  if (bright) {
    const float lum = dot(c, v3(0.2126f, 0.7152f, 0.0722f));
    bright[pix] = (lum > 1.0f) ? make_float4(c.x, c.y, c.z, 1.0f) : make_float4(0.0f, 0.0f, 0.0f, 1.0f);
  }
}
// The ray a hit record belongs to: generation 0 recomputes the primary ray of its pixel (the same code, the same bits).
RM_DEV void wfRayOf(const SceneBlock *sb, const RowMap &map, int W, int H, const WfWs &ws, int gen, int src, V3 &ro, V3 &rd,
                    uint32_t &path, uint32_t slot) {
  if (gen == 0) {
    const int r = src / W, x = src - r * W;
    primaryRay(sb, x, map.frameRow(r), W, H, ro, rd);
    path = slot;
  } else {
    const float4 O = ws.rayO[gen & 1][src], D = ws.rayD[gen & 1][src];
    ro = v3(O.x, O.y, O.z);
    rd = v3(D.x, D.y, D.z);
    path = f2u(O.w);
  }
}
// Direction and march range of a light's shadow ray: lightSetup's L and maxT (frag:1864-1880) without the falloff terms.
RM_DEV void wfLightRay(const RmLight &li, V3 p, float far, V3 &L, float &maxT) {
  if (li.type == RM_LIGHT_DIRECTIONAL) {
    L = normalize(v3(-li.dir[0], -li.dir[1], -li.dir[2]));
    maxT = far;
  } else {
    const V3 toL = sub(v3(li.pos[0], li.pos[1], li.pos[2]), p);
    L = normalize(toL);
    maxT = len(toL);
  }
}

// Register budgets (waves per SIMD) of the march kernels: the shadow kernel fits 64 VGPRs; the primary / bounce kernels carry
// the ray set-up of their refill path (primaryRay's IEEE divisions) and spill 15 / 5 registers at that budget.
#ifndef RM_WF_MARCH_WAVES
#define RM_WF_MARCH_WAVES 8
#endif
#ifndef RM_WF_PRIMARY_WAVES
#define RM_WF_PRIMARY_WAVES 6
#endif
constexpr int wfMarchWaves(int kind) { return kind == 2 ? RM_WF_MARCH_WAVES : RM_WF_PRIMARY_WAVES; }
// Cursor granularity.  One device counter sustains ≈88 atomics per µs (measured in round 1): with 64-slot hit chunks the 20 M
// primary hits of the 8K Menger frame were 311 k atomics ≈ 3.5 ms of a 5.0 ms kernel (the bounce kernels likewise), so hit
// slots, rays and pixels are reserved 256 at a time (a wave's unused remainder becomes holes; 64 was atomic-bound, 1024 and
// guided chunks measured slower: profiles/r03_b_wavefront.md).
#ifndef RM_WF_SLOT_CHUNK
#define RM_WF_SLOT_CHUNK 256
#endif
#ifndef RM_WF_RAY_CHUNK
#define RM_WF_RAY_CHUNK 256
#endif
#ifndef RM_WF_PIXEL_CHUNK
#define RM_WF_PIXEL_CHUNK 256
#endif
constexpr uint32_t kWfSlotChunk = RM_WF_SLOT_CHUNK;
constexpr uint32_t kWfStripes = 64;  // power of two
constexpr uint32_t wfRayChunk(int kind) { return kind == 0 ? RM_WF_PIXEL_CHUNK : RM_WF_RAY_CHUNK; }

// KIND 0: primary rays from the tile-major pixel cursor; 1: bounce rays of generation `gen` from the ray queue;
// 2: shadow rays of generation `gen`, ray id = light·(hit slots) + hit slot.
// SKIP: the table walk's skip test (tables with primitives among two or more objects; the launcher decides — a lone Menger
// sponge has nothing to pass over and the test's registers cost it 5 %).
template <int KIND, bool SKIP = false>
__global__ __launch_bounds__(64, wfMarchWaves(KIND)) void wf_march_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                                          int nRows, float4 *__restrict__ out,
                                                                          float4 *__restrict__ bright, WfWs ws, int gen,
                                                                          int flushThreshold, uint32_t rayChunk, uint32_t maxChunk, uint32_t slotChunk) {
  uint32_t *cnt = ws.counters + WF_STRIDE * gen;
  const int tilesX = (W + 7) >> 3, tilesY = (nRows + 7) >> 3;
  const uint32_t nSlots = (KIND == 2) ? cnt[WF_HITS] : 0u;
  const uint32_t total = (KIND == 0) ? (uint32_t)tilesX * (uint32_t)tilesY * 64u
                         : (KIND == 1) ? ws.counters[WF_STRIDE * (gen - 1) + WF_NEXT] : nSlots * (uint32_t)sb->numLights;
  // Expensive rays cluster in the source order (a region of grazing shadow rays that all run 256 steps is contiguous in the
  // hit list): a wave's chunk of consecutive ids would be all cheap or all expensive, and the waves that drew the expensive
  // chunks end the kernel alone.  So consecutive 64-id blocks of the cursor map to kWfStripes stripes of the source, and every
  // chunk samples the whole source.
  const uint32_t stripeBlocks = (((total + 63u) >> 6) + kWfStripes - 1u) / kWfStripes, totalPad = stripeBlocks * kWfStripes * 64u;
  const int maxSteps = sb->s.maxSteps;
  const float far = sb->cam.initialFar;
  const bool soft = sb->s.enableSoftShadow != 0;
  const float cullR2 = (KIND == 2 && soft) ? sb->cullR2Soft : sb->cullR2;
  const V3 bg = backgroundColor(sb);
  const float ks = sb->g.ks;
  const unsigned long long lt = laneMaskLt();
  const int cur = gen & 1;
  Counters none{0, 0, 0, 0, 0, 0};

  int st = ST_NEED, steps = 0;
  uint32_t src = 0;
  V3 ro = v3(0, 0, 0), rd = v3(0, 0, 0);
  float depth = 0.0f, end = 0.0f, pen = 1.0f;
  // the table walk's skip test (rm_device.hip.h, sdScene<…, SKIP>): ub = upper bound of the next evaluation's minimum; every ray
  // direction here is a normalised vector (|rd| within a few ulp of 1), hence one padded constant for all lanes
  float ub = __builtin_inff();
  const float lipLen = sb->cullLip * 1.001f;
  uint32_t srcCur = 0, srcEnd = 0, slotCur = 0, slotEnd = 0;  // wave-uniform cursors into the reserved chunks
  bool exhausted = false;

  for (;;) {
    const unsigned long long mMarch = __ballot(st == ST_MARCH);
    const unsigned long long mWait = __ballot(st == ST_NEED);
    if (mMarch == 0 ? (mWait != 0) : ((int)__popcll(mWait) >= flushThreshold)) {
      // ---- refill the waiting lanes from the wave's chunk of the source ----
      if (srcCur == srcEnd && !exhausted) {
        // guided self-scheduling: a chunk is the work still unclaimed (as of this wave's previous reservation) split over
        // four rounds of all waves, between `rayChunk` (the floor, at the end of the kernel) and `maxChunk`
        uint32_t want = (totalPad - srcEnd) / (4u * gridDim.x);
        const uint32_t hi = maxChunk > rayChunk ? maxChunk : rayChunk;  // maxChunk <= rayChunk: fixed chunks
        want = want < rayChunk ? rayChunk : (want > hi ? hi : want);
        srcCur = waveReserve(&cnt[KIND == 2 ? WF_SHADOW : WF_SRC], want);
        srcEnd = srcCur + want;
        if (srcCur >= totalPad) { exhausted = true; srcEnd = srcCur; }
        else if (srcEnd > totalPad) srcEnd = totalPad;
      }
      const uint32_t avail = srcEnd - srcCur, n = (uint32_t)__popcll(mWait);
      if (st == ST_NEED) {
        const uint32_t rank = (uint32_t)__popcll(mWait & lt);
        if (rank < avail) {
          // cursor position → ray id: 64-id blocks dealt round-robin into kWfStripes far-apart stripes of the source
          const uint32_t cid = srcCur + rank, blk = cid >> 6;
          const uint32_t id = (((blk & (kWfStripes - 1u)) * stripeBlocks + blk / kWfStripes) << 6) | (cid & 63u);
          if (id >= total) {
            // padding of the striped id space: nothing to do, stays NEED
          } else if (KIND == 0) {
            int x, r;
            if (decodePixel(id, tilesX, W, nRows, x, r)) {  // else: padding lane of an edge tile, stays NEED
              src = (uint32_t)(r * W + x);
              primaryRay(sb, x, map.frameRow(r), W, H, ro, rd);
              end = sceneCullEnd<true>(sb, ro, rd, far, cullR2);
              if (far >= 0.0f && end < 0.0f) {
                // the ray starts outside the scene's bounding ball and never enters it: its one evaluation cannot hit (the
                // ball's margin keeps every distance value above the hit threshold) and then depth 0 > end — the background
                out[src] = make_float4(bg.x, bg.y, bg.z, 1.0f);
                if (bright) bright[src] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
              } else {
                depth = 0.0f; steps = 0;
                st = ST_MARCH; if (SKIP) ub = __builtin_inff();
              }
            }
          } else if (KIND == 1) {
            const float4 O = ws.rayO[cur][id], D = ws.rayD[cur][id];
            src = id;
            ro = v3(O.x, O.y, O.z);
            rd = v3(D.x, D.y, D.z);
            end = sceneCullEnd<true>(sb, ro, rd, far, cullR2);
            depth = 0.0f; steps = 0;
            st = ST_MARCH; if (SKIP) ub = __builtin_inff();
          } else {
            const uint32_t li = id / nSlots, h = id - li * nSlots;  // light-major: a wave marches toward one light
            if (ws.hit[h].x >= 0) {                                   // else a hole: stays NEED
              const float4 P = ws.surfP[h], Nn = ws.surfN[h];
              const V3 p = v3(P.x, P.y, P.z), N = v3(Nn.x, Nn.y, Nn.z);
              V3 L;
              float maxT;
              wfLightRay(sb->lights[li], p, far, L, maxT);  // li differs across a wave only where two lights' ranges meet
              if (!(dot(N, L) <= 0.005f)) {                  // frag:1912 drops the light otherwise: its ray is not marched
                ro = shadowOrigin(p, N);
                rd = L;
                end = soft ? sceneCullEnd<false>(sb, ro, rd, maxT, cullR2) : sceneCullEnd<true>(sb, ro, rd, maxT, cullR2);
                depth = 0.0f; pen = 1.0f; steps = 0;
                src = li * ws.cap + h;
                st = ST_MARCH; if (SKIP) ub = __builtin_inff();
              }
            }
          }
        } else if (exhausted) {
          st = ST_DONE;
        }
      }
      srcCur += (n < avail) ? n : avail;
      continue;
    }
    if (mMarch == 0) break;  // nothing marching, nothing waiting: every lane is DONE

    // ---- one step of raymarch (frag:1459-1470) / softshadow (frag:1708-1714) on every marching lane ----
    bool hitNow = false;
    float hitD = 0.0f, hitTz = 0.0f;
    int hitObj = -1;
    if (st == ST_MARCH) {
      // trap mode 2: a hit record carries trap.z alone (the sponge's palette index) and this pipeline's tables hold no bulb
      const SceneMin c = sdScene<false, 0, KIND != 2 ? 2 : 0, SKIP>(sb, madd(rd, depth, ro), none, ub);
      const bool hit = fabs_(c.d) < kSurfaceDist;
      bool fin = hit || depth > end;
      if (!fin) {
        if (KIND == 2) {
          if (soft) pen = min_(pen, divr_(8.0f * c.d, depth));
          depth = depth + fabs_(c.d);
        } else {
          depth = fma(c.d, 1.0f, depth);
        }
        if (SKIP) ub = nextMinBound(c.d, lipLen, depth);
        steps++;
        fin = steps >= maxSteps;  // the loop runs out: a miss
      }
      if (fin) {
        st = ST_NEED;
        if (KIND == 2) {
          ws.shadow[src] = hit ? u2f(kWfShadowHit) : pen;
        } else if (hit) {
          hitNow = true;
          hitD = depth - c.d;  // frag:1477
          hitObj = c.idx;
          hitTz = c.trap.z;
        } else if (KIND == 0) {  // frag:2325, 2459-2465: the pixel is the background
          out[src] = make_float4(bg.x, bg.y, bg.z, 1.0f);
          if (bright) bright[src] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        } else {  // frag:2519-2523: the bounce sees the background and ends the path
          const uint32_t path = f2u(ws.rayO[cur][src].w);
          const int2 pp = ws.pathPix[path];
          const float4 A = ws.pathA[path], B = ws.pathB[path];
          const float2 Cc = ws.pathC[path];
          V4 refl = v4(B.x, B.y, B.z, A.w);
          refl.x += (ks * B.w) * bg.x;
          refl.y += (ks * Cc.x) * bg.y;
          refl.z += (ks * Cc.y) * bg.z;
          refl.w += 1.0f;
          wfStorePixel(out, bright, pp.x, v3(A.x, A.y, A.z), refl);
        }
      }
    }
    if (KIND != 2) {  // append this trip's hits to the generation's hit list (slots reserved per wave in chunks)
      const unsigned long long mHit = __ballot(hitNow);
      if (mHit) {
        const uint32_t n = (uint32_t)__popcll(mHit), avail = slotEnd - slotCur;
        uint32_t fresh = 0;
        if (n > avail) fresh = waveReserve(&cnt[WF_HITS], slotChunk);  // the old chunk gets exactly filled
        if (hitNow) {
          const uint32_t rank = (uint32_t)__popcll(mHit & lt);
          const uint32_t slot = (rank < avail) ? (slotCur + rank) : (fresh + (rank - avail));
          ws.hit[slot] = make_int4((int)src, (int)f2u(hitD), hitObj, (int)f2u(hitTz));
        }
        if (n > avail) { slotCur = fresh + (n - avail); slotEnd = fresh + slotChunk; }
        else slotCur += n;
      }
    }
  }
  if (KIND != 2)  // unused slots of this wave's last chunk are holes
    for (uint32_t sl = slotCur + (threadIdx.x & 63); sl < slotEnd; sl += 64) ws.hit[sl] = make_int4(-1, 0, 0, 0);
}

// One lane per hit of generation `gen`: frag:2333-2336 and getPhong's ambient-occlusion term (frag:1859).
template <bool SKIP>
__global__ __launch_bounds__(256) void wf_surface_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H, WfWs ws,
                                                          int gen) {
  const uint32_t nSlots = ws.counters[WF_STRIDE * gen + WF_HITS];
  Counters none{0, 0, 0, 0, 0, 0};
  for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < nSlots; h += gridDim.x * blockDim.x) {
    const int4 rec = ws.hit[h];
    if (rec.x < 0) continue;
    V3 ro, rd;
    uint32_t path;
    wfRayOf(sb, map, W, H, ws, gen, rec.x, ro, rd, path, h);
    const V3 p = madd(rd, u2f((uint32_t)rec.y), ro);
    // seeds of the skip test as in render(): the march stopped within SURFACE_DIST of a surface, p at most that far (× |rd|) from there
    const float ubP = fma(kSurfaceDist, (sb->cullLip * len(rd)) * 1.0001f, kSurfaceDist) * 1.001f + fma(fabs_(u2f((uint32_t)rec.y)), 1.0e-6f, 1.0e-5f);
    V3 n = getNormal<false, 0, SKIP>(sb, p, none, fma(0.0005f, sb->cullLip * 1.001f, ubP));
    if (sb->s.features & RM_FEAT_PERLIN_BUMP) n = bumpNormal(n, p);
    float ao = 1.0f;
    if (sb->s.enableAmbientOcclusion) ao = calcAO<false, 0, SKIP>(sb, p, n, none, ubP);
    ws.surfP[h] = make_float4(p.x, p.y, p.z, ao);
    ws.surfN[h] = make_float4(n.x, n.y, n.z, 0.0f);
  }
}

// One lane per hit of generation `gen`: the rest of getPhong (frag:1860-1931), render's fractal palette (frag:2354-2365)
// and main's reflection bookkeeping (frag:2481-2524).  numBounces = the generations that follow a primary hit.
__global__ __launch_bounds__(256) void wf_light_kernel(const SceneBlock *__restrict__ sb, RowMap map, int W, int H,
                                                        float4 *__restrict__ out, float4 *__restrict__ bright, WfWs ws, int gen,
                                                        int numBounces) {
  __shared__ uint32_t s_scratch[8];
  const uint32_t nSlots = ws.counters[WF_STRIDE * gen + WF_HITS];
  const int nl = sb->numLights;
  const float ka = sb->g.ka, ks = sb->g.ks, kd = sb->g.kd;
  const float far = sb->cam.initialFar;
  const bool soft = sb->s.enableSoftShadow != 0;
  const int nxt = (gen + 1) & 1;
  for (uint32_t base = blockIdx.x * blockDim.x; base < nSlots; base += gridDim.x * blockDim.x) {  // block-uniform trip count
    const uint32_t h = base + threadIdx.x;
    int4 rec = make_int4(-1, 0, 0, 0);
    if (h < nSlots) rec = ws.hit[h];
    bool bounce = false;
    V3 nro = v3(0, 0, 0), nrd = v3(0, 0, 0);
    uint32_t path = 0;
    if (rec.x >= 0) {
      V3 ro, rd;
      wfRayOf(sb, map, W, H, ws, gen, rec.x, ro, rd, path, h);
      const float4 P = ws.surfP[h], Nn = ws.surfN[h];
      const V3 p = v3(P.x, P.y, P.z), N = v3(Nn.x, Nn.y, Nn.z);
      const RmObject &o = sb->objs[rec.z];  // per-lane index: vector loads from the constant block
      Material mat;
      mat.amb = v3(o.cAmbient[0], o.cAmbient[1], o.cAmbient[2]);
      mat.dif = v3(kd * o.cDiffuse[0], kd * o.cDiffuse[1], kd * o.cDiffuse[2]);  // getDiffuse, untextured
      mat.spec = v3(o.cSpecular[0], o.cSpecular[1], o.cSpecular[2]);
      mat.shininess = o.shininess;
      V3 total = v3((mat.amb.x * ka) * P.w, (mat.amb.y * ka) * P.w, (mat.amb.z * ka) * P.w);  // frag:1860
      const V3 V = normalize(neg(rd));
      for (int i = 0; i < nl; i++) {
        const RmLight &li = sb->lights[i];  // uniform index → scalar loads
        const LightGeom g = lightSetup(li, p, far);
        int shObj = -1;
        float pen = 1.0f;
        if (!(dot(N, g.L) <= 0.005f)) {  // the shadow ray was marched
          const float s = ws.shadow[(uint32_t)i * ws.cap + h];
          shObj = (f2u(s) == kWfShadowHit) ? 0 : -1;
          pen = s;
        }
        V3 c;
        if (lightTerm(li, g, mat, N, V, ks, shObj, pen, soft, c)) total = add(total, c);
      }
      V3 col = total;
      if (o.type == RM_MENGERSPONGE) {  // frag:2362-2365
        const float tz = u2f((uint32_t)rec.w);
        const V3 c = v3(fma(0.5f, cos_(fma(2.0f, tz, 0.0f)), 0.5f), fma(0.5f, cos_(fma(2.0f, tz, 1.0f)), 0.5f),
                        fma(0.5f, cos_(fma(2.0f, tz, 2.0f)), 0.5f));
        col = mul(c, total);
      }
      V3 phong, fil;
      V4 refl;
      int pix, obj0;
      if (gen == 0) {
        phong = col;
        refl = v4(0.0f, 0.0f, 0.0f, 0.0f);
        fil = v3(1.0f, 1.0f, 1.0f);
        pix = rec.x;
        obj0 = rec.z;
      } else {
        const int2 pp = ws.pathPix[path];
        const float4 A = ws.pathA[path], B = ws.pathB[path];
        const float2 Cc = ws.pathC[path];
        pix = pp.x; obj0 = pp.y;
        phong = v3(A.x, A.y, A.z);
        fil = v3(B.w, Cc.x, Cc.y);
        refl = v4(B.x, B.y, B.z, A.w);
        refl.x += (ks * fil.x) * col.x;  // frag:2519-2520
        refl.y += (ks * fil.y) * col.y;
        refl.z += (ks * fil.z) * col.z;
        refl.w += 1.0f;
      }
      const RmObject &o0 = sb->objs[obj0];
      const V3 cRefl = v3(o0.cReflective[0], o0.cReflective[1], o0.cReflective[2]);
      bounce = gen < numBounces && sb->s.enableReflection && len(cRefl) != 0.0f;  // frag:2491-2492
      if (bounce) {
        nrd = reflect(rd, N);  // frag:2496-2497: this generation's ray and bumped normal
        nro = v3(fma(nrd.x * kSurfaceDist, 3.0f, p.x), fma(nrd.y * kSurfaceDist, 3.0f, p.y), fma(nrd.z * kSurfaceDist, 3.0f, p.z));
        fil = mul(fil, cRefl);
        ws.pathPix[path] = make_int2(pix, obj0);
        ws.pathA[path] = make_float4(phong.x, phong.y, phong.z, refl.w);
        ws.pathB[path] = make_float4(refl.x, refl.y, refl.z, fil.x);
        ws.pathC[path] = make_float2(fil.y, fil.z);
      } else {
        wfStorePixel(out, bright, pix, phong, refl);
      }
    }
    const uint32_t slot = blockAppend(bounce, &ws.counters[WF_STRIDE * gen + WF_NEXT], s_scratch);
    if (bounce) {
      ws.rayO[nxt][slot] = make_float4(nro.x, nro.y, nro.z, u2f(path));
      ws.rayD[nxt][slot] = make_float4(nrd.x, nrd.y, nrd.z, 0.0f);
    }
  }
}

}  // namespace rm
