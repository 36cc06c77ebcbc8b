// rm_device.hip.h — per-pixel sphere tracing for gfx950: SDF table, march loops, normals, Perlin bump,
// Phong + shadow + AO, reflection / refraction compositing.
//
// MI355X-native replacement of resources/raymarch.frag (reference line numbers are cited as frag:N).
// Numeric contract: every value follows DESIGN.md §3 — binary32, fused multiply-add only where fma()
// is written (-ffp-contract=off), built-ins from rm_math.hip.h — so the frame is bit-reproducible and
// is checked bit-for-bit against the separately written CPU oracle.
//
// Execution shape: one lane per pixel, a wave = an 8×8 pixel tile (coherent rays).  The object loop of
// the scene union is wave-uniform: object records are read with a uniform index from the constant
// scene block (scalar loads → SGPR operands, no per-lane table traffic) and the switch on the
// primitive type is a scalar branch.  Per-lane (divergent) material lookups after a hit read the LDS
// copy of the table.
#pragma once
#include "../../include/raymarcher_amd.h"
#include <type_traits>

#include "rm_math.hip.h"
#include "rm_internal.h"

namespace rm {

struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

RM_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RM_DEV V4 v4(float x, float y, float z, float w) { return V4{x, y, z, w}; }
RM_DEV V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RM_DEV V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RM_DEV V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
RM_DEV V3 scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RM_DEV V3 neg(V3 a) { return v3(-a.x, -a.y, -a.z); }
RM_DEV V3 madd(V3 a, float s, V3 b) { return v3(fma(a.x, s, b.x), fma(a.y, s, b.y), fma(a.z, s, b.z)); }
RM_DEV float dot2(float ax, float ay, float bx, float by) { return fma(ay, by, ax * bx); }
RM_DEV float dot(V3 a, V3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
RM_DEV float len2(float x, float y) { return sqrt_(dot2(x, y, x, y)); }
RM_DEV float len(V3 a) { return sqrt_(dot(a, a)); }  // the guarded cheap form is no faster here (measured: bulb frame +1.4 %)
RM_DEV V3 normalize(V3 a) { float inv = rcp_(len(a)); return scale(a, inv); }
RM_DEV V3 reflect(V3 I, V3 N) { float k = 2.0f * dot(N, I); return madd(N, -k, I); }
RM_DEV V3 refract(V3 I, V3 N, float eta) {
  float d = dot(N, I);
  float k = fma(-(eta * eta), fma(-d, d, 1.0f), 1.0f);
  float t = fma(eta, d, sqrt_(k));
  V3 r = v3(fma(-t, N.x, eta * I.x), fma(-t, N.y, eta * I.y), fma(-t, N.z, eta * I.z));
  return (k < 0.0f) ? v3(0.0f, 0.0f, 0.0f) : r;
}
RM_DEV V3 mix(V3 a, V3 b, float t) { return v3(mix_(a.x, b.x, t), mix_(a.y, b.y, t), mix_(a.z, b.z, t)); }

constexpr float kSurfaceDist = 0.001f;  // frag:32

// Per-lane work counters (only live in the counting instantiations; where nothing reads them the increments are dead code).
// COUNT is a mode: 0 = no counters (production), 1 = count the REFERENCE's work (no bounding-ball culls, shadow rays of
// dropped lights still marched — the algorithmic figure of the roofline), 2 = count the work the production kernel really
// executes (culls and skips honoured).  evals = sdScene evaluations, iters = Mandelbulb iterations, shades = surface points
// shaded by render() (normal + bump + Phong; primary hits and bounce hits), fbm9 / fbmd8 = terrain-height and cloud-noise
// evaluations of the procedural layers (frag:630-667), shapes = sdMatch evaluations (objects really evaluated by the table walk).
struct Counters { unsigned long long evals, iters, shades, fbm9, fbmd8, shapes; };

}  // namespace rm
#include "rm_sampler.hip.h"
#include "rm_env.hip.h"
namespace rm {


// Everything a frame needs, in one constant block (uploaded once per launch by the launcher).
struct EvalRecord {
  float m[12];  // invModel[0..2], [4..6], [8..10], [12..14]
  float scaleFactor;
  int32_t type;
  // object-space bound for the table walk's skip test: the unit shape lies inside the ball |p| <= boundR (sdMatch's sizes,
  // +inf for types without one), so its distance value is >= (|p_object| − boundR)·scaleFactor; invScale = 1 / scaleFactor
  float invScale, boundR;
};
static_assert(sizeof(EvalRecord) == 64, "one cache line");
struct SceneBlock {
  RmCamera cam;
  RmGlobals g;
  RmSettings s;
  int32_t numObjects;
  int32_t numLights;
  RmObject objs[RM_MAX_OBJECTS];
  RmLight lights[RM_MAX_LIGHTS];
  RmTexture tex[RM_MAX_TEXTURES];  // device pixel pointers
  int32_t numTextures;
  RmTexture noise;                 // `noise` (night sky, sea)
  RmTexture skybox[6];             // cube-map faces +X,-X,+Y,-Y,+Z,-Z
  const uint8_t *ltc1, *ltc2;      // RM_LTC_SIZE² RGBA8 tables of the area lights
  // World-space ball outside which no object can be hit (computed by the launcher, see scene_cull_ball); cullOk = 0
  // when the scene holds an object without a known bound.
  // What an evaluation reads of an object, packed into one 64-byte line (ONE s_load_dwordx16 instead of seven scattered
  // loads from RmObject): the three rows of invModel that sdScene uses, scaleFactor, type.  Filled by the launcher.
  alignas(64) EvalRecord evalRec[RM_MAX_OBJECTS];
  // nearClip / farClip at the corners of the full-screen quad, per triangle: [below / above the TL-BR diagonal][near, far]
  // [P0, P1 − P0, P2 − P0][xyzw]; filled by the launcher (ray_planes), interpolated per pixel by primaryRay.
  float rayPlane[2][2][3][4];
  float cullC[3];
  float cullR2;
  float cullR2Soft;  // larger ball for soft-shadow rays (0 = none): beyond it 8·d/t >= 1, so the penumbra min() is settled
  int32_t cullOk;
  int32_t cullOneOk;  // 1 = every object is a primitive (cube … rectangle): the march loops may take the single-object fast path
  float cullLip;  // Lipschitz bound of every object's distance value per unit of world length (+inf with a fractal in the table)
  float cullLo[3], cullHi[3];  // axis-aligned box with the same property (see scene_cull_ball); cullBoxOk = 0: none
  int32_t cullBoxOk;
  // Shape of a wave's pixel tile: 2^tileShift pixels wide, 64 >> tileShift tall (3 = 8×8, the default; 2 = 4 wide × 16 tall,
  // which the launcher's tuner picks for pictures it measures faster that way: rm_kernels.hip, "tile shape").  Same pixels.
  int32_t tileShift;
  // World-space bounding ball of every object (centre xyz, radius; filled by scene_cull_ball with the balls it derives anyway),
  // objBallOk = 1 when every object has one: tile_geom_kernel classifies the tiles of a frame WITHOUT cost history by their
  // centre ray's closest approach to these balls (rm_kernels.hip, "tile order").  Never read by the render kernels.
  float objBall[RM_MAX_OBJECTS][4];
  int32_t objBallOk;
  // Launch order of the workgroups (see rm_kernels.hip, "tile order"): workgroup b renders tile tileOrder[b] (a permutation
  // of 0..tileCount-1, heaviest tiles first) or tile b if null; tileCost (or null) accumulates every tile's shader-cycle cost.
  const int32_t *tileOrder;
  uint32_t *tileCost;
  int32_t tileCount;
  // "Light split" (rm_kernels.hip): the first splitTiles tiles of tileOrder — the heaviest of a settled picture — are rendered by
  // numLights workgroups each, one shadow march per pixel apiece (results in splitStore: per tile 64 pixels × (numLights × (object
  // bits, penumbra / distance) + the primary march's result, 6 words)), and finished by a second launch that reads them instead of
  // marching.  0 / null otherwise.
  int32_t splitTiles;
  float *splitStore;
  // Uniforms of sdMengerSponge's prologue (frag:1052-1053: ani = smoothstep(−0.2, 0.2, −cos(0.5·iTime)), off = 1.5·sin(0.01·iTime)),
  // evaluated ONCE per launch by scene_prep_kernel with the contract's own sin / cos instead of once per evaluation per lane
  // (≈45 of the ≈230 vector instructions of a 5-level evaluation); only read when the table holds a Menger sponge.
  float mengerAni, mengerOff;
};

// Which frame row a launch's local row r is: a plain row range (numShards = 1) or the row tiles of one shard of a multi-GPU
// frame (tiles of tileRows rows dealt round-robin, include/raymarcher_amd.h rm_render_tiles).
struct RowMap {
  int rowBegin, tileRows, shard, numShards;
  int relief;  // the partition's root relief (rm_internal.h: 0 = tile t belongs to shard t mod numShards)
  __host__ __device__ int frameRow(int r) const {
    return rowBegin + tile_of(shard, r / tileRows, numShards, relief) * tileRows + (r % tileRows);
  }
};

// What a kernel of the light split hands down to getPhong: part >= 0 — march light `part` only and store its result in slot
// (SPLIT = 1, a heavy tile's partial workgroup); SPLIT = 2 — read every light's result (and the primary march's) from slot
// instead of marching: what the last of a tile's partial workgroups runs to finish it.
struct LightSplit { int part; float *slot; int tileIndex; };
struct SceneMin { int idx; float d; V4 trap; };
struct MarchRes { int obj; float d; V4 trap; };
struct Hit { V3 rd, p, n; int obj; };
struct RenderOut { V3 col; int isEnv; float d; };


// ---- primitives (frag:832-894, 991-1019), unit sizes of sdMatch (frag:1262-1293) -------------------
RM_DEV float sdBox(V3 p, float bx, float by, float bz) {
  float qx = fabs_(p.x) - bx, qy = fabs_(p.y) - by, qz = fabs_(p.z) - bz;
  V3 qm = v3(max_(qx, 0.0f), max_(qy, 0.0f), max_(qz, 0.0f));
  return len(qm) + min_(max_(qx, max_(qy, qz)), 0.0f);
}
RM_DEV float sdCone(V3 p, float r, float h) {
  float pox = len2(p.x, p.z) - r, poy = p.y + h;
  float ex = -r, ey = 2.0f * h;
  float t = clamp_(RM_DIVR_CONST(dot2(pox, poy, ex, ey), dot2(ex, ey, ex, ey)), 0.0f, 1.0f);
  float qx = fma(-ex, t, pox), qy = fma(-ey, t, poy);
  float d = len2(qx, qy);
  return (max_(qx, qy) > 0.0f) ? d : -min_(d, poy);
}
RM_DEV float sdCylinder(V3 p, float h, float r) {
  float dx = len2(p.x, p.z) - r, dy = fabs_(p.y) - h;
  return min_(max_(dx, dy), 0.0f) + len2(max_(dx, 0.0f), max_(dy, 0.0f));
}
RM_DEV float sdOctahedron(V3 p, float s) {
  p = v3(fabs_(p.x), fabs_(p.y), fabs_(p.z));
  float m = ((p.x + p.y) + p.z) - s;
  float rx = fma(3.0f, p.x, -m), ry = fma(3.0f, p.y, -m), rz = fma(3.0f, p.z, -m);
  bool cx = rx < 0.0f, cy = ry < 0.0f, cz = rz < 0.0f;
  V3 q = cx ? p : (cy ? v3(p.y, p.z, p.x) : v3(p.z, p.x, p.y));
  float k = clamp_(0.5f * ((q.z - q.y) + s), 0.0f, s);
  float dl = len(v3(q.x, (q.y - s) + k, q.z - k));
  return (cx || cy || cz) ? dl : (m * 0.57735027f);
}
RM_DEV float sdTorus(V3 p, float tx, float ty) { return len2(len2(p.x, p.z) - tx, p.y) - ty; }
RM_DEV float sdCapsule(V3 p, float h, float r) {
  p.y = p.y - clamp_(p.y, 0.0f, h);
  return len(p) - r;
}
RM_DEV float sdDeathStar(V3 p2, float ra, float rb, float d) {
  float px = p2.x, py = len2(p2.y, p2.z);
  float a = (((ra * ra) - (rb * rb)) + (d * d)) / (2.0f * d);
  float b = sqrt_(max_((ra * ra) - (a * a), 0.0f));
  float inner = len2(px - a, py - b);
  float outer = max_(len2(px, py) - ra, -(len2(px - d, py) - rb));
  return (fma(px, b, -(py * a)) > d * max_(b - py, 0.0f)) ? inner : outer;
}

// ---- fractals ---------------------------------------------------------------------------------------
// frag:751-769
RM_DEV float sdMandelBrot(const SceneBlock *sb, float px, float py) {
  float ltime = fma(-0.5f, cos_(sb->g.iTime * 0.06f), 0.5f);
  float zoom = pow_(0.9f, 50.0f * ltime);
  float k = (0.045f * zoom) * fma(-ltime, 0.5f, 1.0f);
  float cx = -0.745f - k, cy = 0.186f - k;
  float ld2 = 1.0f;
  float lz2 = dot2(px, py, px, py);
  const int n = sb->s.maxSteps;
  for (int i = 0; i < n; i++) {
    ld2 = ld2 * (4.0f * lz2);
    float nx = fma(px, px, -(py * py)) + cx;
    float ny = fma(2.0f * px, py, cy);
    px = nx; py = ny;
    lz2 = dot2(px, py, px, py);
    if (lz2 > 200.0f) break;
  }
  float d = sqrt_(lz2 / ld2) * log_(lz2);
  return sqrt_(clamp_((150.0f / zoom) * d, 0.0f, 1.0f));
}

// frag:775-803.  r = length(w) is sqrt of the same dot product that produced m, so r = sqrt(m) bit for bit.
// The iteration loop is compiled three times and chosen once per call by wave-uniform tests on `power`:
//   BULB_GENERIC    any power: pow through the contract's general evaluation (plan classified once, outside the loop);
//   BULB_TRIG8      power == 8: the same contract, with the two pows written out — binary exponentiation of 8 is
//                   ((r²)²)², of 3.5 is (m·(m·m))·√m — as straight-line code (identical bits, no bit loop);
//   BULB_ALGEBRAIC8 power == 8 and RM_FEAT_BULB_POWER8_ALGEBRAIC: the step by complex squarings (see the header).
//
// TRAPMIN: the orbit-trap minima (frag:795) with the |·| folded into the v_min_f32 (source modifier).  Since the contract's
// min IS v_min_f32 (rm_math), both spellings below produce the same bits; TRAPMIN only picks the one-instruction form.
enum BulbMode { BULB_GENERIC = 0, BULB_TRIG8 = 1, BULB_ALGEBRAIC8 = 2 };
// TRAP = false drops the orbit trap altogether: shadow marches, normal taps and AO taps never read it (with the select
// form the compiler removed it there by itself; the v_min_f32 form is inline asm, so it is spelled out).
template <int COUNT, int MODE, bool TRAPMIN, bool TRAP>
RM_DEV float bulbIterate(const SceneBlock *sb, V3 pos, V4 &resColor, Counters &cnt) {
  // the power-8 instantiations are only entered with sb->g.power == 8.0f: the literal (an inline constant of the multiplies)
  // instead of a scalar-register operand, which halves a VALU instruction's issue rate (profiles/r03_c_valu_microbench.md)
  const float power = (MODE == BULB_GENERIC) ? sb->g.power : 8.0f;
  const float pexp = (power - 1.0f) / 2.0f;
  const int iters = sb->s.fractalIters;
  const bool julia = len2(sb->g.juliaSeed[0], sb->g.juliaSeed[1]) != 0.0f;  // frag:782
  // angles are power·acos(·) ∈ [0, power·pi] and power·atan(·,·) ∈ [−power·pi, power·pi], always finite: for
  // |power| < 1e6 they stay inside the contract range of sin/cos and the range guard can be dropped
  const bool angleSafe = (MODE != BULB_GENERIC) || fabs_(power) < 1.0e6f;
  const PowPlan planPower = powPlanUniform(power), planPexp = powPlanUniform(pexp);  // uniforms: classified once
  V3 w = pos;
  float m = dot(w, w);
  V4 trap = v4(fabs_(w.x), fabs_(w.y), fabs_(w.z), m);
  float dz = 1.0f;
  V3 c = julia ? v3(sb->g.juliaSeed[0], sb->g.juliaSeed[1], 0.0f) : pos;
  for (int i = 0; i < iters; i++) {
    if (COUNT) cnt.iters++;
    // ONE range check per iteration for its square root and its two reciprocals (trigonometric forms): with
    // 2^-96 <= m < inf the unscaled sqrt is exact and r = sqrt(m) lies in [2^-48, 2^64), inside the reciprocal's fast range;
    // max(|w.x|, |w.z|) <= r, so it only needs its lower bound.  Any lane outside (the exact origin, a point on the y axis,
    // non-finite input): the whole wave takes the individually guarded forms — the same bits either way.
    const bool raw = (MODE != BULB_ALGEBRAIC8) &&
                     __ballot(!(m >= 1.262177448e-29f) || !(m < __builtin_inff()) || !(max_(fabs_(w.x), fabs_(w.z)) >= 1.17549435e-38f)) == 0;
    // the iteration's arithmetic, instantiated twice: RAW (unguarded fast forms) and guarded — a wave-uniform branch
    auto step = [&](auto rawTag) __attribute__((always_inline)) {
      constexpr bool RAW = decltype(rawTag)::value;
      float r = RAW ? sqrt_noscale_(m) : sqrt_fast_(m);  // frag:789
      if (MODE == BULB_ALGEBRAIC8) {
        // (y + iρ)^8 = r^8·(cos 8θ + i sin 8θ), ((z + ix)/ρ)^8 = cos 8φ + i sin 8φ by three complex squarings each,
        // m^3.5 = m³·√m — no acos/atan/sin/cos/pow in the loop
        dz = fma(8.0f * (((m * m) * m) * r), dz, 1.0f);
        float rho = sqrt_fast_(dot2(w.x, w.z, w.x, w.z));
        float inv = rcp_(rho);
        float cz = (rho == 0.0f) ? 1.0f : w.z * inv, sx = (rho == 0.0f) ? 0.0f : w.x * inv;
        float re = w.y, im = rho;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          float t = fma(re, re, -(im * im));
          im = 2.0f * (re * im);
          re = t;
          t = fma(cz, cz, -(sx * sx));
          sx = 2.0f * (cz * sx);
          cz = t;
        }
        w = v3(fma(im, sx, c.x), re + c.y, fma(im, cz, c.z));
      } else {
        float pm, pr;
        if (MODE == BULB_TRIG8) {
          pm = (m * (m * m)) * r;             // pow_(m, 3.5): p = m, b = m·m, p = p·b, then ·sqrt(m)
          float r2 = r * r, r4 = r2 * r2;
          pr = r4 * r4;                       // pow_(r, 8): three squarings
        } else {
          pm = powApply(m, pexp, planPexp);
          pr = powApply(r, power, planPower);
        }
        dz = fma(power * pm, dz, 1.0f);       // frag:787
        float b = power * acos_(RAW ? w.y * rcp_raw_(r) : divr_(w.y, r));  // frag:790
        float a = power * atan2_<RAW>(w.x, w.z);                            // frag:791
        float sb_, cb_, sa_, ca_;
        if (angleSafe) { sincos_inrange_(b, sb_, cb_); sincos_inrange_(a, sa_, ca_); }  // wave-uniform
        else { sincos_(b, sb_, cb_); sincos_(a, sa_, ca_); }
        w = v3(fma(pr, sb_ * sa_, c.x), fma(pr, cb_, c.y), fma(pr, sb_ * ca_, c.z));  // frag:792-793
      }
    };
    if (raw) step(std::true_type{}); else step(std::false_type{});
    if (TRAP) {  // trap.x is never read (resColor below)
      if (TRAPMIN) trap = v4(trap.x, hwmin_abs_(trap.y, w.y), hwmin_abs_(trap.z, w.z), hwmin_(trap.w, m));
      else trap = v4(trap.x, min_(trap.y, fabs_(w.y)), min_(trap.z, fabs_(w.z)), min_(trap.w, m));
    }
    m = dot(w, w);
    if (m > 2.0f) break;  // frag:798 (FRACTALS_BAILOUT)
  }
  resColor = v4(m, trap.y, trap.z, trap.w);
  return divr_((0.25f * log_(m)) * sqrt_fast_(m), dz);  // frag:802
}
template <int COUNT, bool TRAPMIN, bool TRAP>
RM_DEV float sdMandelBulb(const SceneBlock *sb, V3 pos, V4 &resColor, Counters &cnt) {
  if (sb->g.power == 8.0f) {  // wave-uniform
    if (sb->s.features & RM_FEAT_BULB_POWER8_ALGEBRAIC) return bulbIterate<COUNT, BULB_ALGEBRAIC8, TRAPMIN, TRAP>(sb, pos, resColor, cnt);
    return bulbIterate<COUNT, BULB_TRIG8, TRAPMIN, TRAP>(sb, pos, resColor, cnt);
  }
  return bulbIterate<COUNT, BULB_GENERIC, TRAPMIN, TRAP>(sb, pos, resColor, cnt);
}

// frag:808-827
RM_DEV float sdSierpinski(V3 p) {
  const float Scale = 1.85f, Offset = 2.0f;
  const float k = Offset * (Scale - 1.0f);
  for (int n = 0; n < 14; n++) {
    if (p.x + p.y < 0.0f) { float t = p.x; p.x = -p.y; p.y = -t; }
    if (p.x + p.z < 0.0f) { float t = p.x; p.x = -p.z; p.z = -t; }
    if (p.y + p.z < 0.0f) { float t = p.z; p.z = -p.y; p.y = -t; }
    p = v3(fma(p.x, Scale, -k), fma(p.y, Scale, -k), fma(p.z, Scale, -k));
  }
  return len(p) * pow_(Scale, -14.0f);
}

// frag:1049-1071 (ma = frag:124-126, column-major)
// Two bit-identical shortcuts, both on wave-uniform conditions:
//  * ani == 0 (always at iTime = 0): p ← mix(p, ma·(p + off), 0) = ma·(…)·0 + p·1 is p itself — up to the sign of a zero
//    component, which nothing downstream can see (|p| in the box, mod(p·s, 2) gives +0 for either zero) — or NaN where p is
//    not finite, where the level's candidate c is NaN with or without the mix; so the rotation is skipped (15 of ≈45
//    instructions per level);
//  * the division by s = 3^(m+1) uses the exact constant-divisor sequence (RM_DIVC) for the first eight levels: its
//    numerator is min(…) − 1 with min(…) a number near 1, i.e. 0 or at least 2^-24 in magnitude.
// Inside a level every operand of the three maxima and of the minimum is a freshly computed |·| (canonical, >= +0), so the
// compiler's own v_max_f32 with |·| source modifiers / v_min3_f32 give the contract's bits (same comparator as the v_min_f32 /
// v_max_f32 pair of rm_math, no NaN-quieting prologue needed) — and, unlike the inline-asm spellings, carry no hazard s_nops.
// TRAP = 0 (shadow marches, normal and AO taps) drops the orbit-trap bookkeeping (res is then unspecified); 1 is the shader's
// vec4; 2 keeps res.z alone (see the level below).
// LEVELS > 0 / STILL = true: the level count and ani == 0 as COMPILE-TIME facts — the levels are then one basic block with no
// scalar branch between them (their candidates c are independent of one another; only the final compare / select chain is
// serial), which is worth more than the branches look: see sdMengerSponge below.  LEVELS = 0: both read at run time.
template <int TRAP, int LEVELS, bool STILL>
RM_DEV float mengerImpl(const SceneBlock *sb, V3 p, V4 &res) {
  float d = sdBox(p, 1.0f, 1.0f, 1.0f);
  float ty = 1.0f, tz = 0.0f;
  const float ani = sb->mengerAni, off = sb->mengerOff;  // scene_prep_kernel (rm_kernels.hip): the prologue's uniforms
  const int levels = LEVELS > 0 ? LEVELS : sb->s.mengerLevels;
  const bool still = STILL || ani == 0.0f;  // wave-uniform (scalar loads)
  // one level (frag:1057-1069); hs = 0.5·s = 0.5·3^m and the divisor 3^(m+1) as compile-time constants (DIVC > 0)
  auto level = [&](int m, float divc, float hs, float sNext) __attribute__((always_inline)) {
    if (!still) {
      V3 v = v3(p.x + off, p.y + off, p.z + off);
      V3 mv = v3(fma(-0.80f, v.z, fma(0.00f, v.y, 0.60f * v.x)), fma(0.00f, v.z, fma(1.00f, v.y, 0.00f * v.x)),
                 fma(0.60f, v.z, fma(0.00f, v.y, 0.80f * v.x)));
      p = mix(p, mv, ani);
    }
    // mod(·, 2) − 1 as 2·fract(·/2) − 1 (contract: rm_mod_pow2): 2·f is exact, so the fma rounds exactly like the subtraction
    V3 a = v3(fma(2.0f, fract_(p.x * hs), -1.0f), fma(2.0f, fract_(p.y * hs), -1.0f), fma(2.0f, fract_(p.z * hs), -1.0f));
    float rx = fabs_(fma(-3.0f, fabs_(a.x), 1.0f)), ry = fabs_(fma(-3.0f, fabs_(a.y), 1.0f)),
          rz = fabs_(fma(-3.0f, fabs_(a.z), 1.0f));
    // min(max(rx,ry), min(max(ry,rz), max(rz,rx))) is the MEDIAN of the three: one v_med3_f32 where the pairwise maxima are not
    // read, instead of three v_max + v_min3, all half-rate instructions.  Same value for every input: the median is a selection,
    // and with NaN operands v_med3 returns min3 of the others, which is what the IEEE min / max chain leaves too.  The maxima are
    // read by the trap's .y — which a MANDELBULB hit reads when a sponge comes later in the table (sdScene hands back the trap of
    // the last fractal evaluated, UB3) — so: TRAP = 1 the full trap as the shader computes it; TRAP = 2 its .z alone, for callers
    // that read nothing else and whose tables hold no bulb (the wavefront pipeline); TRAP = 0 (shadow, normal, AO) nothing.
    float da = 0.0f, db = 0.0f, dc = 0.0f, med;
    if (TRAP == 1) {
      da = __builtin_fmaxf(rx, ry); db = __builtin_fmaxf(ry, rz); dc = __builtin_fmaxf(rz, rx);
      med = __builtin_fminf(da, __builtin_fminf(db, dc));
    } else {
      med = __builtin_amdgcn_fmed3f(rx, ry, rz);
    }
    const float num = med - 1.0f;
    const float c = (divc > 0.0f) ? divc_(num, divc, 1.0f / divc) : (num / sNext);
    const bool up = c > d;
    d = up ? c : d;
    if (TRAP == 1) {
      const float t = __builtin_fminf(ty, ((0.2f * da) * db) * dc);
      ty = up ? t : ty;
    }
    if (TRAP) tz = up ? ((1.0f + (float)m) / 4.0f) : tz;
  };
  // the first eight levels written out (straight-line code, constants as immediates, one scalar test per level); deeper
  // ones — far below pixel size — in a loop with the IEEE division
  if (levels > 0) level(0, 3.0f, 0.5f, 0.0f);
  if (levels > 1) level(1, 9.0f, 1.5f, 0.0f);
  if (levels > 2) level(2, 27.0f, 4.5f, 0.0f);
  if (levels > 3) level(3, 81.0f, 13.5f, 0.0f);
  if (levels > 4) level(4, 243.0f, 40.5f, 0.0f);
  if (levels > 5) level(5, 729.0f, 121.5f, 0.0f);
  if (levels > 6) level(6, 2187.0f, 364.5f, 0.0f);
  if (levels > 7) level(7, 6561.0f, 1093.5f, 0.0f);
  float s = 6561.0f;
  for (int m = 8; m < levels; m++) {
    const float hs = 0.5f * s;
    s = s * 3.0f;
    level(m, 0.0f, hs, s);
  }
  res = v4(d, ty, tz, 0.0f);
  return d;
}
// The two level counts that occur (the shader's 4, BASELINE config 5's 5) at iTime = 0 get their own straight-line instantiation,
// chosen by a wave-uniform test; everything else takes the generic one.  Same bits.
template <int TRAP>
RM_DEV float sdMengerSponge(const SceneBlock *sb, V3 p, V4 &res) {
  const int levels = sb->s.mengerLevels;
  if (sb->mengerAni == 0.0f) {
    if (levels == 5) return mengerImpl<TRAP, 5, true>(sb, p, res);
    if (levels == 4) return mengerImpl<TRAP, 4, true>(sb, p, res);
  }
  return mengerImpl<TRAP, 0, false>(sb, p, res);
}

// ---- scene union (frag:1406-1430) ---------------------------------------------------------------------
// BULB=true is the single-Mandelbulb scene class (numObjects == 1, type MANDELBULB): same arithmetic,
// no table walk.
// SKIP (table-walk classes, march loops): an object whose bounding ball is farther from EVERY live lane's point than that
// lane's current minimum cannot lower it (its value is >= (|p_object| − boundR)·scaleFactor, and the update below is a strict
// <): the wave passes over it after the transform — 6 instructions and a wave-uniform branch instead of the shape's 20-50.
// The 8×8 pixel tile of a wave is spatially coherent, so whole waves agree often; a wave down to its last straggler lanes
// (the serial chains that end small frames) agrees almost always.
// ub: an upper bound of the minimum this call will return (+inf: none).  The march loops know one: every admitted shape's
// distance value changes by at most sb->cullLip per unit of world length (exact SDFs are 1-Lipschitz and scaleFactor undoes the
// model matrix's stretch), so the minimum at the next point is at most the minimum at this one plus cullLip × the step.  With
// it the test does not depend on the nearest object coming early in the table.
// TRACK (with SKIP): also return in `second` a lower bound of every OTHER object's value at p — the runner-up of the minimum:
// the values of the objects evaluated, and (|p_object|·(1 − ε) − boundR)·scaleFactor for the ones passed over.  The march
// loops use it for the single-object fast path (sdSceneOne below).
template <bool BULB, int COUNT, int TRAP, bool SKIP, bool TRACK>  // TRAP: 0 none, 1 the shader's, 2 the sponge's .z alone (mengerImpl)
RM_DEV SceneMin sdSceneImpl(const SceneBlock *sb, V3 p, Counters &cnt, float ub, float &second) {
  SceneMin res;
  if (TRACK) second = __builtin_inff();
  res.d = 1000000.0f;
  res.idx = -1;
  res.trap = v4(0.0f, 0.0f, 0.0f, 0.0f);
  if (COUNT) cnt.evals++;
  const int n = BULB ? 1 : sb->numObjects;
  for (int i = 0; i < n; i++) {
    // uniform index → scalar loads; the table walk reads the packed line (three loads instead of seven), the single-bulb
    // class its one RmObject (hoisted out of the march loops either way)
    const EvalRecord &o = sb->evalRec[i];
    float M[12];
#pragma unroll
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 3; r++) M[c * 3 + r] = BULB ? sb->objs[0].invModel[c * 4 + r] : o.m[c * 3 + r];
    const float scaleFactor = BULB ? sb->objs[0].scaleFactor : o.scaleFactor;
    V3 po = v3(fma(M[6], p.z, fma(M[3], p.y, fma(M[0], p.x, M[9]))),
               fma(M[7], p.z, fma(M[4], p.y, fma(M[1], p.x, M[10]))),
               fma(M[8], p.z, fma(M[5], p.y, fma(M[2], p.x, M[11]))));  // frag:1417
    if (SKIP && !BULB) {
      const float lim = fma(ub, o.invScale, o.boundR), pp = dot(po, po);
      const bool far = (lim >= 0.0f) && (pp > (lim * lim) * 1.00003f);
      if (__ballot(!far) == 0ull) {
        if (TRACK) {
          // lower bound of the value passed over, with its safety margin applied DOWNWARD whatever its sign (a lane inside
          // this object's ball while ub < 0 gives a negative bound: multiplying that by 0.9999 would raise it)
          const float v = fma(__builtin_amdgcn_sqrtf(pp), 0.9999f, -o.boundR) * scaleFactor;
          second = min_(second, fma(-fabs_(v), 1.0e-4f, v));
        }
        continue;
      }
    }
    if (COUNT) cnt.shapes++;
    float d;
    const int type = BULB ? (int)RM_MANDELBULB : o.type;
    switch (type) {  // sdMatch, frag:1262-1293 — wave-uniform branch
      case RM_CUBE: d = sdBox(po, 0.5f, 0.5f, 0.5f); break;
      case RM_CONE: d = sdCone(po, 0.5f, 0.5f); break;
      case RM_CYLINDER: d = sdCylinder(po, 0.5f, 0.5f); break;
      case RM_SPHERE: d = len(po) - 0.5f; break;
      case RM_OCTAHEDRON: d = sdOctahedron(po, 0.5f); break;
      case RM_TORUS: d = sdTorus(po, 0.5f, 0.125f); break;
      case RM_CAPSULE: d = sdCapsule(po, 0.5f, 0.1f); break;
      case RM_DEATHSTAR: d = sdDeathStar(po, 0.5f, 0.35f, 0.5f); break;
      case RM_RECTANGLE: d = sdBox(po, 0.5f, 0.5f, 0.0f); break;
      case RM_MANDELBROT: d = sdMandelBrot(sb, po.x, po.y); break;
      case RM_MANDELBULB: d = sdMandelBulb<COUNT, BULB && COUNT == 0, TRAP != 0>(sb, po, res.trap, cnt); break;
      case RM_MENGERSPONGE: d = sdMengerSponge<TRAP>(sb, po, res.trap); break;
      case RM_SIERPINSKI: d = sdSierpinski(po); break;
      default: continue;
    }
    float cur = d * scaleFactor;  // frag:1419
    if (TRACK) {
      // the runner-up: the old minimum if cur replaces it, else cur; a NaN value (it never becomes the minimum) voids the bound
      const float other = (cur < res.d) ? res.d : ((cur >= res.d) ? cur : -__builtin_inff());
      second = min_(second, other);
    }
    if (cur < res.d) { res.d = cur; res.idx = i; }
    if (SKIP && !BULB) ub = min_(ub, cur);
  }
  return res;
}
template <bool BULB, int COUNT, int TRAP = 1, bool SKIP = false>
RM_DEV SceneMin sdScene(const SceneBlock *sb, V3 p, Counters &cnt, float ub = __builtin_inff()) {
  float unused;
  return sdSceneImpl<BULB, COUNT, TRAP, SKIP, false>(sb, p, cnt, ub, unused);
}
// One object of the table alone (wave-uniform index j, a primitive): what sdScene returns when every other object is known to
// be farther than the minimum (see march()).
template <int COUNT>
RM_DEV SceneMin sdSceneOne(const SceneBlock *sb, int j, V3 p, Counters &cnt) {
  SceneMin res;
  res.trap = v4(0.0f, 0.0f, 0.0f, 0.0f);
  res.idx = j;
  if (COUNT) { cnt.evals++; cnt.shapes++; }
  const EvalRecord &o = sb->evalRec[j];
  float M[12];
#pragma unroll
  for (int c = 0; c < 12; c++) M[c] = o.m[c];
  const V3 po = v3(fma(M[6], p.z, fma(M[3], p.y, fma(M[0], p.x, M[9]))),
                   fma(M[7], p.z, fma(M[4], p.y, fma(M[1], p.x, M[10]))),
                   fma(M[8], p.z, fma(M[5], p.y, fma(M[2], p.x, M[11]))));  // frag:1417
  float d = 0.0f;
  switch (o.type) {  // primitives only (the fast path is off with a fractal in the table)
    case RM_CUBE: d = sdBox(po, 0.5f, 0.5f, 0.5f); break;
    case RM_CONE: d = sdCone(po, 0.5f, 0.5f); break;
    case RM_CYLINDER: d = sdCylinder(po, 0.5f, 0.5f); break;
    case RM_SPHERE: d = len(po) - 0.5f; break;
    case RM_OCTAHEDRON: d = sdOctahedron(po, 0.5f); break;
    case RM_TORUS: d = sdTorus(po, 0.5f, 0.125f); break;
    case RM_CAPSULE: d = sdCapsule(po, 0.5f, 0.1f); break;
    case RM_DEATHSTAR: d = sdDeathStar(po, 0.5f, 0.35f, 0.5f); break;
    default: d = sdBox(po, 0.5f, 0.5f, 0.0f); break;  // RM_RECTANGLE
  }
  res.d = d * o.scaleFactor;  // frag:1419
  return res;
}
// The bound for the next evaluation of a march that has just stepped by |d| along rd: d + cullLip·|d|·|rd|, with slack for
// the rounding of the point (lipLen = cullLip·|rd|, padded, once per march).  NaN / inf (a fractal in the table: cullLip = inf)
// disable the test.
RM_DEV float nextMinBound(float d, float lipLen, float depth) {
  const float ub = fma(fabs_(d), lipLen, d);
  return fma(fabs_(ub), 1.0e-4f, ub) + fma(depth, 1.0e-6f, 1.0e-5f);
}

// frag:1436-1444
// ub: an upper bound of sdScene at p + any tap (the taps are 0.0005 from p), +inf = none; SKIP as in sdScene.
template <bool BULB, int COUNT, bool SKIP = false>
RM_DEV V3 getNormal(const SceneBlock *sb, V3 p, Counters &cnt, float ub = __builtin_inff()) {
  const float ex = (1.0f * 0.5773f) * 0.0005f, ey = (-1.0f * 0.5773f) * 0.0005f;
  float d[4];
#pragma unroll 1
  for (int k = 0; k < 4; k++) {
    // taps e.xyy, e.yyx, e.yxy, e.xxx
    float ox = (k == 0 || k == 3) ? ex : ey;
    float oy = (k >= 2) ? ex : ey;
    float oz = (k == 1 || k == 3) ? ex : ey;
    float v = sdScene<BULB, COUNT, false, SKIP>(sb, v3(p.x + ox, p.y + oy, p.z + oz), cnt, ub).d;
    d[0] = (k == 0) ? v : d[0];
    d[1] = (k == 1) ? v : d[1];
    d[2] = (k == 2) ? v : d[2];
    d[3] = (k == 3) ? v : d[3];
  }
  V3 n;
  n.x = fma(ex, d[3], fma(ey, d[2], fma(ey, d[1], ex * d[0])));
  n.y = fma(ex, d[3], fma(ex, d[2], fma(ey, d[1], ey * d[0])));
  n.z = fma(ex, d[3], fma(ey, d[2], fma(ex, d[1], ey * d[0])));
  return normalize(n);
}

// frag:1453-1484 (side = +1 outside, −1 inside) and frag:1703-1725 (SHADOW) share one loop body.
// SHADOW=false: returns obj, d = rayDepth − minD on a hit, rayDepth on a miss (contract UB2).
// SHADOW=true : returns obj, d = penumbra factor res (contract UB1), k = 8, start depth 0.
//
// CULL (single-Mandelbulb class only; never in the counted variant, whose counters are the reference's work): a march
// whose miss distance nobody reads — primary / secondary rays of render(), hard-shadow rays — may stop as soon as the ray
// has left a ball |p_object| <= R for good, because it can no longer hit.  For power 8 and |p| = ρ the first iteration
// gives |w| >= ρ^8 − max(ρ, |seed|), the loop bails out (|w|² > 2) and the estimate is
// 0.5·ln|w|·|w| / (8ρ^7 + 1), increasing in ρ and in |w|:
//   R = 1.15 (|seed| <= 1.14, scaleFactor >= 0.05): |w| >= 1.909, estimate >= 0.0277 — times scaleFactor still 1.39× the
//            hit threshold; the set {estimate < 0.001} lies inside the ball;
//   R = 2.1  (|seed| <= 2,    scaleFactor >= 0.01): |w| >= 254, estimate >= 0.68.
// The march therefore ends at min(end, t_exit) with the same obj = −1 it would have reached some evaluations later at
// t > far.  Rays that never enter the ball stop after their first evaluation.  Soft-shadow rays also read the penumbra
// factor min(8·d/t); they use the launcher's larger ball (sceneCullEnd with cullR2Soft), which accounts for anisotropic
// object scales.
RM_DEV float bulbCullEnd(const SceneBlock *sb, V3 ro, V3 rd, float end) {
  const RmObject &o = sb->objs[0];
  const float jx = sb->g.juliaSeed[0], jy = sb->g.juliaSeed[1];
  const float seed2 = fma(jx, jx, jy * jy);
  const bool ok = (sb->g.power == 8.0f) && (o.scaleFactor >= 0.01f) && (seed2 <= 4.0f);  // wave-uniform
  if (!ok) return end;
  const bool tight = (o.scaleFactor >= 0.05f) && (seed2 <= 1.2996f);
  const float R2 = tight ? 1.3225f : 4.41f;
  const float *M = o.invModel;
  const V3 po = v3(fma(M[8], ro.z, fma(M[4], ro.y, fma(M[0], ro.x, M[12]))), fma(M[9], ro.z, fma(M[5], ro.y, fma(M[1], ro.x, M[13]))),
                   fma(M[10], ro.z, fma(M[6], ro.y, fma(M[2], ro.x, M[14]))));
  const V3 pd = v3(fma(M[8], rd.z, fma(M[4], rd.y, M[0] * rd.x)), fma(M[9], rd.z, fma(M[5], rd.y, M[1] * rd.x)),
                   fma(M[10], rd.z, fma(M[6], rd.y, M[2] * rd.x)));
  const float a = dot(pd, pd), b = dot(po, pd), c = dot(po, po) - R2;
  const float disc = fma(b, b, -(a * c));
  float tExit = (sqrt_fast_(max_(disc, 0.0f)) - b) / a;
  tExit = fma(tExit, 1.0001f, 1.0e-3f);
  if (c > 0.0f && (b >= 0.0f || disc < 0.0f)) tExit = -1.0f;  // outside and never entering
  if (!(a > 0.0f)) return end;
  return min_(end, tExit);  // a NaN tExit leaves `end` untouched
}
// The same idea for any scene: the launcher bounds every object by a world-space ball (exact SDFs are >= the distance
// to their object's ball; the bound includes a margin δ with minScale·δ >> the hit threshold), and a march whose miss
// distance is unused ends where its ray leaves the ball around all of them.  Soft-shadow rays use a larger ball, beyond
// which 8·d/t >= 1 so that the penumbra factor min(pen, 8·d/t) <= 1 can no longer change (none if the bound is too weak).
// HARD = true (every march but the soft-shadow ones): also end where the ray leaves the launcher's axis-aligned box.
template <bool HARD = true>
RM_DEV float sceneCullEnd(const SceneBlock *sb, V3 ro, V3 rd, float end, float R2) {
  if (!sb->cullOk || !(R2 > 0.0f)) return end;  // wave-uniform
  const V3 po = v3(ro.x - sb->cullC[0], ro.y - sb->cullC[1], ro.z - sb->cullC[2]);
  const float a = dot(rd, rd), b = dot(po, rd), c = dot(po, po) - R2;
  const float disc = fma(b, b, -(a * c));
  float tExit = (sqrt_fast_(max_(disc, 0.0f)) - b) / a;
  tExit = fma(tExit, 1.0001f, 1.0e-3f);
  if (c > 0.0f && (b >= 0.0f || disc < 0.0f)) tExit = -1.0f;
  if (!(a > 0.0f)) return end;
  if (HARD && sb->cullBoxOk) {  // wave-uniform
    // the parameter at which the ray passes the far plane of each slab; beyond the smallest of them it is outside the box for
    // good.  A zero component gives ±inf (or NaN on the plane itself, which the contract's min ignores); an approximate
    // reciprocal will do — the margins below absorb its error
    const float tx = ((rd.x >= 0.0f ? sb->cullHi[0] : sb->cullLo[0]) - ro.x) * __builtin_amdgcn_rcpf(rd.x);
    const float ty = ((rd.y >= 0.0f ? sb->cullHi[1] : sb->cullLo[1]) - ro.y) * __builtin_amdgcn_rcpf(rd.y);
    const float tz = ((rd.z >= 0.0f ? sb->cullHi[2] : sb->cullLo[2]) - ro.z) * __builtin_amdgcn_rcpf(rd.z);
    const float tb = min_(min_(tx, ty), tz);
    tExit = min_(tExit, fma(fabs_(tb), 1.0e-4f, tb) + 1.0e-3f);
  }
  return min_(end, tExit);
}
// ub0: an upper bound of sdScene at ro (+inf = none), for the first evaluation's skip test.
template <bool BULB, int COUNT, bool SHADOW, bool CULL = false>
RM_DEV MarchRes march(const SceneBlock *sb, V3 ro, V3 rd, float end, float side, Counters &cnt, float ub0 = __builtin_inff()) {
  if (CULL && COUNT != 1) {
    const bool softRay = SHADOW && sb->s.enableSoftShadow != 0;  // wave-uniform
    if (BULB && !softRay) end = bulbCullEnd(sb, ro, rd, end);
    else if (softRay) end = sceneCullEnd<false>(sb, ro, rd, end, sb->cullR2Soft);
    else end = sceneCullEnd<true>(sb, ro, rd, end, sb->cullR2);
  }
  float depth = 0.0f;
  float pen = 1.0f;
  SceneMin c;
  c.d = 1000000.0f; c.idx = -1; c.trap = v4(0.0f, 0.0f, 0.0f, 0.0f);
  const int steps = sb->s.maxSteps;
  const bool soft = SHADOW && sb->s.enableSoftShadow != 0;  // wave-uniform: the penumbra factor is read only then
  constexpr bool SKIP = CULL && !BULB && COUNT != 1;
  const float lipLen = SKIP ? (sb->cullLip * len(rd)) * 1.0001f : 0.0f;
  float ub = ub0;
  // Single-object fast path (SKIP classes): `second` is a lower bound of every object's value except the nearest one's at the
  // point just evaluated, carried along the ray by the same Lipschitz argument as ub (it can only have dropped by cullLip × the
  // step).  While it stays above ub for every live lane, and the lanes agree on the nearest object, no other object can be
  // the minimum at the next point — strictly — so sdScene there IS that object's value: one record, one shape, no table walk.
  float second = -__builtin_inff();
  for (int i = 0; i < steps; i++) {
    const V3 p = madd(rd, depth, ro);
    bool one = false;
    int nearU = 0;
    if (SKIP) {
      nearU = __builtin_amdgcn_readfirstlane(c.idx);
      one = sb->cullOneOk && __ballot(!(c.idx == nearU && nearU >= 0 && second > ub)) == 0ull;
    }
    if (SKIP && one) c = sdSceneOne<COUNT>(sb, nearU, p, cnt);
    else c = sdSceneImpl<BULB, COUNT, !SHADOW, SKIP, SKIP>(sb, p, cnt, ub, second);
    if (fabs_(c.d) < kSurfaceDist || depth > end) break;
    if (SHADOW) {
      if (soft) pen = min_(pen, divr_(8.0f * c.d, depth));
      depth = depth + fabs_(c.d);
    } else {
      depth = fma(c.d, side, depth);
    }
    if (SKIP) {
      ub = nextMinBound(c.d, lipLen, depth);
      second = fma(fabs_(c.d), -(lipLen * 1.0001f), second) - fma(depth, 1.0e-6f, 1.0e-5f);
    }
  }
  MarchRes r;
  bool hit = fabs_(c.d) < kSurfaceDist;
  r.obj = hit ? c.idx : -1;
  r.trap = c.trap;
  if (SHADOW) r.d = pen;
  else r.d = hit ? (depth - c.d) : depth;
  return r;
}

// ---- Perlin bump (frag:1587-1691) ---------------------------------------------------------------------
// mod(·, 289) and the /7 of pgrad with the constant-divisor sequence (rm_math.hip.h, RM_DIVC): every operand here is a
// small non-negative integer, exactly representable, so the quotients are the correctly rounded ones.
RM_DEV float permute(float x) {
  const float y = fma(x, 34.0f, 1.0f) * x;
  return fma(-289.0f, floor_(RM_DIVC(y, 289.0f)), y);
}
RM_DEV float taylorInvSqrt(float r) { return fma(-0.85373472095314f, r, 1.79284291400159f); }
RM_DEV float fade(float t) { return ((t * t) * t) * fma(t, fma(t, 6.0f, -15.0f), 10.0f); }

RM_DEV void pgrad(float ixyz, float &gx, float &gy, float &gz) {  // frag:1626-1632 / 1634-1640
  float x = RM_DIVC(ixyz, 7.0f);
  float y = fract_(RM_DIVC(floor_(x), 7.0f)) - 0.5f;
  x = fract_(x);
  float z = (0.5f - fabs_(x)) - fabs_(y);
  float sz = step_(z, 0.0f);
  gx = fma(-sz, step_(0.0f, x) - 0.5f, x);
  gy = fma(-sz, step_(0.0f, y) - 0.5f, y);
  gz = z;
}
RM_DEV float pcorner(float ixyz, float fx, float fy, float fz) {  // gradient · offset (frag:1642-1669)
  float gx, gy, gz;
  pgrad(ixyz, gx, gy, gz);
  V3 g = v3(gx, gy, gz);
  g = scale(g, taylorInvSqrt(dot(g, g)));
  return dot(g, v3(fx, fy, fz));
}
RM_DEV float pnoise(V3 p) {  // frag:1610-1676
  float i0x = floor_(p.x), i0y = floor_(p.y), i0z = floor_(p.z);
  float i1x = mod_(i0x + 1.0f, 256.0f), i1y = mod_(i0y + 1.0f, 256.0f), i1z = mod_(i0z + 1.0f, 256.0f);
  i0x = mod_(i0x, 256.0f); i0y = mod_(i0y, 256.0f); i0z = mod_(i0z, 256.0f);
  float f0x = fract_(p.x), f0y = fract_(p.y), f0z = fract_(p.z);
  float f1x = f0x - 1.0f, f1y = f0y - 1.0f, f1z = f0z - 1.0f;
  float px0 = permute(i0x), px1 = permute(i1x);
  float ixy00 = permute(px0 + i0y), ixy10 = permute(px1 + i0y);  // lanes x, y of frag:1622
  float ixy01 = permute(px0 + i1y), ixy11 = permute(px1 + i1y);  // lanes z, w
  float n000 = pcorner(permute(ixy00 + i0z), f0x, f0y, f0z);
  float n100 = pcorner(permute(ixy10 + i0z), f1x, f0y, f0z);
  float n010 = pcorner(permute(ixy01 + i0z), f0x, f1y, f0z);
  float n110 = pcorner(permute(ixy11 + i0z), f1x, f1y, f0z);
  float n001 = pcorner(permute(ixy00 + i1z), f0x, f0y, f1z);
  float n101 = pcorner(permute(ixy10 + i1z), f1x, f0y, f1z);
  float n011 = pcorner(permute(ixy01 + i1z), f0x, f1y, f1z);
  float n111 = pcorner(permute(ixy11 + i1z), f1x, f1y, f1z);
  float fx = fade(f0x), fy = fade(f0y), fz = fade(f0z);
  float nzx = mix_(n000, n001, fz), nzy = mix_(n100, n101, fz), nzz = mix_(n010, n011, fz), nzw = mix_(n110, n111, fz);
  float nyx = mix_(nzx, nzz, fy), nyy = mix_(nzy, nzw, fy);
  return 2.2f * mix_(nyx, nyy, fx);
}
RM_DEV V3 bumpNormal(V3 normal, V3 pos) {  // frag:1679-1691, BUMP_SCALE 10, BUMP_INTENSITY 2
  V3 ps = scale(pos, 10.0f);
  float nv = pnoise(ps);
  float g[3];
#pragma unroll 1
  for (int k = 0; k < 3; k++) {
    float v = pnoise(v3(ps.x + ((k == 0) ? 0.1f : 0.0f), ps.y + ((k == 1) ? 0.1f : 0.0f),
                        ps.z + ((k == 2) ? 0.1f : 0.0f))) - nv;
    g[0] = (k == 0) ? v : g[0];
    g[1] = (k == 1) ? v : g[1];
    g[2] = (k == 2) ? v : g[2];
  }
  return normalize(madd(v3(g[0], g[1], g[2]), 2.0f, normal));
}

// ---- shading --------------------------------------------------------------------------------------------
// frag:1729-1740
// ubPos: an upper bound of sdScene at pos (+inf = none); a tap lies h·|nor| from it.
template <bool BULB, int COUNT, bool SKIP = false>
RM_DEV float calcAO(const SceneBlock *sb, V3 pos, V3 nor, Counters &cnt, float ubPos = __builtin_inff()) {
  float occ = 0.0f, sca = 1.0f;
  const float lipN = SKIP ? (sb->cullLip * len(nor)) * 1.001f : 0.0f;
  for (int i = 0; i < 5; i++) {
    float h = 0.01f + ((0.12f * (float)i) / 4.0f);
    float d = sdScene<BULB, COUNT, false, SKIP>(sb, madd(nor, h, pos), cnt, SKIP ? fma(h, lipN, ubPos) : ubPos).d;
    occ = fma(h - d, sca, occ);
    sca = sca * 0.95f;
    if (occ > 0.35f) break;
  }
  return clamp_(fma(-3.0f, occ, 1.0f), 0.0f, 1.0f) * fma(0.5f, nor.y, 0.5f);
}
// frag:445-447
RM_DEV float attenuation(float d, float f0, float f1, float f2) {
  return min_(rcp_(fma(d * d, f2, fma(d, f1, f0))), 1.0f);
}
// frag:439-442, 450-461
RM_DEV float angularFalloff(const RmLight &li, V3 L) {
  V3 nd = normalize(v3(li.dir[0], li.dir[1], li.dir[2]));
  float cosalpha = dot(neg(nd), L);
  float inner = li.angle - li.penumbra;
  float t = (acos_(cosalpha) - inner) / (li.angle - inner);
  float mid = 1.0f - fma(-2.0f, pow_(t, 3.0f), 3.0f * pow_(t, 2.0f));
  float r = (cosalpha > cos_(inner)) ? 1.0f : mid;
  return (cosalpha <= cos_(li.angle)) ? 0.0f : r;
}

// Material of the hit object, fetched per lane (the index may differ across the wave).  `dif` is getDiffuse's
// result for the shaded point: kd·cDiffuse, or its blend with the texture sample (frag:1746-1781).
struct Material { V3 amb, dif, spec; float shininess; };

// ---- textured diffuse (frag:1299-1398, 1746-1781) ---------------------------------------------------------
constexpr float kTextureEps = 0.005f;  // frag:37
constexpr float kGlPi = 3.14159265f;   // frag:41
RM_DEV float uFromTheta(float theta) {  // frag:1346-1350
  return (theta < 0.0f) ? (-theta / (2.0f * kGlPi)) : (1.0f - (theta / (2.0f * kGlPi)));
}
RM_DEV void uvMap(int type, V3 p, float rU, float rV, float &su, float &sv) {
  float u, v;
  if (type == RM_CUBE) {  // frag:1299-1333
    float ax = fabs_(p.x), ay = fabs_(p.y), az = fabs_(p.z);
    float m = max_(max_(ax, ay), az);
    if (m == ax) { u = (p.x < 0.0f) ? (p.z + 0.5f) : (-p.z + 0.5f); v = p.y + 0.5f; }
    else if (m == ay) { u = p.x + 0.5f; v = (p.y < 0.0f) ? (p.z + 0.5f) : (-p.z + 0.5f); }
    else { u = (p.z < 0.0f) ? (-p.x + 0.5f) : (p.x + 0.5f); v = p.y + 0.5f; }
  } else if (type == RM_SPHERE) {  // frag:1381-1398
    u = uFromTheta(atan2_(p.z, p.x));
    float phi = asin_(p.y / 0.5f);
    v = phi / kGlPi + 0.5f;
    if (v == 0.0f || v == 1.0f) u = 0.5f;
  } else {  // cone frag:1336-1354, cylinder frag:1357-1378
    bool top = (type == RM_CYLINDER) && (fabs_(p.y - 0.5f) < kTextureEps);
    bool base = fabs_(p.y + 0.5f) < kTextureEps;
    if (top) { u = p.x + 0.5f; v = -p.z + 0.5f; }
    else if (base) { u = p.x + 0.5f; v = p.z + 0.5f; }
    else { u = uFromTheta(atan2_(p.z, p.x)); v = p.y + 0.5f; }
  }
  su = u * rU;
  sv = v * rV;
}
// frag:1746-1781.  `o` is the per-lane object record (LDS copy).
template <bool TEX>
RM_DEV V3 getDiffuse(const SceneBlock *sb, const RmObject &o, V3 p) {
  const float kd = sb->g.kd;
  V3 plain = v3(kd * o.cDiffuse[0], kd * o.cDiffuse[1], kd * o.cDiffuse[2]);
  if (!TEX) return plain;
  if (o.texLoc == -1) return plain;
  const float *M = o.invModel;
  V3 po = v3(fma(M[8], p.z, fma(M[4], p.y, fma(M[0], p.x, M[12]))), fma(M[9], p.z, fma(M[5], p.y, fma(M[1], p.x, M[13]))),
             fma(M[10], p.z, fma(M[6], p.y, fma(M[2], p.x, M[14]))));
  float su, sv;
  uvMap(o.type, po, o.repeatU, o.repeatV, su, sv);
  V3 t = sampleTexture(sb->tex[o.texLoc], su, sv);
  float k = (1.0f - o.blend) * kd;
  return v3(fma(o.blend, t.x, k * o.cDiffuse[0]), fma(o.blend, t.y, k * o.cDiffuse[1]), fma(o.blend, t.z, k * o.cDiffuse[2]));
}

// ---- area lights: linearly transformed cosines (frag:349-424, 1794-1822) ---------------------------------------
// LTC tables are GL_CLAMP_TO_EDGE; GL_LINEAR is used for every fetch (the reference's MIN = NEAREST / MAG = LINEAR
// split hangs on derivatives taken in divergent control flow, undefined in GLSL; a smooth uv over a 64² table magnifies).
constexpr float kLutScale = (64.0f - 1.0f) / 64.0f, kLutBias = 0.5f / 64.0f;  // frag:47-49
RM_DEV V3 cross(V3 a, V3 b) {
  return v3(fma(a.y, b.z, -(a.z * b.y)), fma(a.z, b.x, -(a.x * b.z)), fma(a.x, b.y, -(a.y * b.x)));
}
RM_DEV V3 integrateEdgeVec(V3 v1, V3 v2) {  // frag:349-361
  float x = dot(v1, v2), y = fabs_(x);
  float a = fma(fma(0.0145206f, y, 0.4965155f), y, 0.8543985f);
  float b = fma(4.1616724f + y, y, 3.4175940f);
  float v = a / b;
  float ts = (x > 0.0f) ? v : fma(0.5f, rcp_(sqrt_noscale_(max_(fma(-x, x, 1.0f), 1e-7f))), -v);
  return scale(cross(v1, v2), ts);
}
RM_DEV float ltcEvaluate(const SceneBlock *sb, V3 N, V3 V, V3 P, const M3 &MinvIn, const RmLight &li) {  // frag:368-424
  V3 T1 = normalize(madd(N, -dot(V, N), V));
  V3 T2 = cross(N, T1);
  M3 B;  // transpose(mat3(T1, T2, N))
  B.c[0][0] = T1.x; B.c[0][1] = T2.x; B.c[0][2] = N.x;
  B.c[1][0] = T1.y; B.c[1][1] = T2.y; B.c[1][2] = N.y;
  B.c[2][0] = T1.z; B.c[2][1] = T2.z; B.c[2][2] = N.z;
  M3 Minv = mulMM(MinvIn, B);
  V3 pts[4], L[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    pts[k] = v3(li.points[k][0], li.points[k][1], li.points[k][2]);
    L[k] = normalize(mulMV(Minv, sub(pts[k], P)));
  }
  V3 lightNormal = cross(sub(pts[1], pts[0]), sub(pts[3], pts[0]));
  bool behind = dot(sub(pts[0], P), lightNormal) < 0.0f;
  V3 vsum = integrateEdgeVec(L[0], L[1]);
  vsum = add(vsum, integrateEdgeVec(L[1], L[2]));
  vsum = add(vsum, integrateEdgeVec(L[2], L[3]));
  vsum = add(vsum, integrateEdgeVec(L[3], L[0]));
  float l = len(vsum);
  float z = (l == 0.0f) ? 0.0f : vsum.z / l;  // UB11: 0/0 when the four edge terms cancel exactly
  z = behind ? -z : z;
  float sc = sampleRGBA8<true>(sb->ltc2, RM_LTC_SIZE, RM_LTC_SIZE, fma(fma(z, 0.5f, 0.5f), kLutScale, kLutBias),
                               fma(l, kLutScale, kLutBias)).w;
  float sum = l * sc;
  return (!behind && !li.twoSided) ? 0.0f : sum;
}
RM_DEV V3 getAreaLight(const SceneBlock *sb, V3 N, V3 V, V3 P, const RmLight &li, const Material &mat) {  // frag:1795-1822
  float dotNV = clamp_(dot(N, V), 0.0f, 1.0f);
  float u = fma(0.0f, kLutScale, kLutBias), v = fma(sqrt_noscale_(1.0f - dotNV), kLutScale, kLutBias)  /* 1 − x, x in [0,1]: 0 or >= 2^-24 */;
  V4 t1 = sampleRGBA8<true>(sb->ltc1, RM_LTC_SIZE, RM_LTC_SIZE, u, v), t2 = sampleRGBA8<true>(sb->ltc2, RM_LTC_SIZE, RM_LTC_SIZE, u, v);
  M3 Minv, I;
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int r = 0; r < 3; r++) { Minv.c[c][r] = (c == 1 && r == 1) ? 1.0f : 0.0f; I.c[c][r] = (c == r) ? 1.0f : 0.0f; }
  Minv.c[0][0] = t1.x; Minv.c[0][2] = t1.y; Minv.c[2][0] = t1.z; Minv.c[2][2] = t1.w;
  float diffuse = ltcEvaluate(sb, N, V, P, I, li);
  float specular = ltcEvaluate(sb, N, V, P, Minv, li);
  float spx = specular * fma(li.intensity - mat.spec.x, t2.y, mat.spec.x * t2.x);
  float spy = specular * fma(li.intensity - mat.spec.y, t2.y, mat.spec.y * t2.x);
  float spz = specular * fma(li.intensity - mat.spec.z, t2.y, mat.spec.z * t2.x);
  return v3((li.color[0] * 1.0f) * fma(mat.dif.x, diffuse, spx), (li.color[1] * 1.0f) * fma(mat.dif.y, diffuse, spy),
            (li.color[2] * 1.0f) * fma(mat.dif.z, diffuse, spz));
}

// Per-light geometry of getPhong (frag:1864-1880): direction to the light, shadow-march range, attenuation,
// spot falloff.  `li` is read with a wave-uniform index (scalar loads).
struct LightGeom { V3 L; float maxT, fAtt, aFall; };
RM_DEV LightGeom lightSetup(const RmLight &li, V3 p, float far) {
  LightGeom g;
  g.fAtt = 1.0f; g.aFall = 1.0f;
  if (li.type == RM_LIGHT_DIRECTIONAL) {
    g.L = normalize(v3(-li.dir[0], -li.dir[1], -li.dir[2]));
    g.maxT = far;
  } else {
    V3 lpos = v3(li.pos[0], li.pos[1], li.pos[2]);
    float d = len(sub(p, lpos));
    V3 toL = sub(lpos, p);
    g.L = normalize(toL);
    g.fAtt = attenuation(d, li.func[0], li.func[1], li.func[2]);
    g.maxT = len(toL);
    if (li.type == RM_LIGHT_SPOT) g.aFall = angularFalloff(li, g.L);
  }
  return g;
}
// Shadow-ray origin p + N·SURFACE_DIST·5 (frag:1908).
RM_DEV V3 shadowOrigin(V3 p, V3 N) {
  return v3(fma(N.x * kSurfaceDist, 5.0f, p.x), fma(N.y * kSurfaceDist, 5.0f, p.y), fma(N.z * kSurfaceDist, 5.0f, p.z));
}
// One light's contribution (frag:1910-1928) given the shadow-march result; returns false if the light is
// skipped (occluded, or N·L <= 0.005).
RM_DEV bool lightTerm(const RmLight &li, const LightGeom &g, const Material &mat, V3 N, V3 V, float ks,
                      int shadowObj, float pen, bool soft, V3 &cur) {
  float NdotL = dot(N, g.L);
  bool lit = (shadowObj == -1) && !(NdotL <= 0.005f);
  NdotL = clamp_(NdotL, 0.0f, 1.0f);
  V3 lc = v3(li.color[0], li.color[1], li.color[2]);
  cur = v3((mat.dif.x * NdotL) * lc.x, (mat.dif.y * NdotL) * lc.y, (mat.dif.z * NdotL) * lc.z);
  V3 R = reflect(neg(g.L), N);
  float RdotV = clamp_(dot(R, V), 0.0f, 1.0f);
  float sp = (mat.shininess == 0.0f) ? (ks * RdotV) : (ks * pow_(RdotV, mat.shininess));
  cur = v3(fma(sp * mat.spec.x, lc.x, cur.x), fma(sp * mat.spec.y, lc.y, cur.y), fma(sp * mat.spec.z, lc.z, cur.z));
  cur = scale(cur, g.fAtt * g.aFall);
  if (soft) cur = scale(cur, pen);
  return lit;
}

// The hard-shadow marches of ONE shading point as a per-lane queue (directional lights only).  In getPhong's light loop a
// wave marches light 0's rays, then light 1's, …: every light costs the LONGEST of its rays, and lanes whose light is
// dropped (N·L <= 0.005) idle through it.  Here every lane marches its own rays back to back — bit i of `need` = light i
// wants a march — one evaluation per trip; a lane whose ray ends starts its next one in the same trip.  Each ray is the very
// same sequence of evaluations as in march<…, SHADOW = true>: same origin, direction, cull end, step cap, hit test.
// Returns the mask of lights whose ray hit.  (The schedule simulators, scripts/sim/, price this at ×0.95 of the
// instructions of the per-light loop on the bench frame and ×0.87 on the 8K Menger frame.)
template <bool BULB, int COUNT, bool CULLS>
RM_DEV uint32_t shadowQueue(const SceneBlock *sb, V3 so, uint32_t need, float far, Counters &cnt) {
  const int maxSteps = sb->s.maxSteps;
  if (maxSteps <= 0) return 0u;  // march() then evaluates nothing and reports a miss
  uint32_t hitMask = 0u, rest = need;
  int cur = -1, step = 0;
  V3 L = v3(0.0f, 0.0f, 0.0f);
  float depth = 0.0f, end = 0.0f;
  auto nextRay = [&]() {
    cur = rest ? (__builtin_ctz(rest)) : -1;
    rest &= rest - 1u;
    if (cur >= 0) {
      const RmLight &li = sb->lights[cur];  // per-lane index: a vector load, once per ray
      L = normalize(v3(-li.dir[0], -li.dir[1], -li.dir[2]));  // lightSetup's direction of a directional light
      end = far;
      if (CULLS) end = BULB ? bulbCullEnd(sb, so, L, far) : sceneCullEnd(sb, so, L, far, sb->cullR2);
      depth = 0.0f;
      step = 0;
    }
  };
  nextRay();
  while (__ballot(cur >= 0) != 0ull) {
    if (cur >= 0) {
      const SceneMin c = sdScene<BULB, COUNT, false>(sb, madd(L, depth, so), cnt);
      const bool hit = fabs_(c.d) < kSurfaceDist;
      bool fin = hit || depth > end;
      if (!fin) {
        depth = depth + fabs_(c.d);
        step++;
        fin = step >= maxSteps;  // the loop of march() runs out: a miss
      }
      if (fin) {
        hitMask |= (hit ? 1u : 0u) << cur;
        nextRay();
      }
    }
  }
  return hitMask;
}

// frag:1842-1933 with getDiffuse's untextured path (frag:1749-1752) and getSpecular (frag:1787-1792)
// RES = true adds the area-light branch (frag:1884-1905); `objs` is only read there.
// CULLS: end marches at the scene's bounds and pass over far objects in the table walk (off in the ENV instantiations, whose
// register budget it would break).
// SPLIT (table-walk kernels without samplers or secondary rays only): see LightSplit.
template <bool BULB, int COUNT, bool RES, bool CULLS, int SPLIT = 0>
RM_DEV V3 getPhong(const SceneBlock *sb, const RmObject *objs, const Material &mat, V3 N, V3 p, V3 rd, float far, Counters &cnt,
                   float ubPos = __builtin_inff(), LightSplit split = LightSplit{-1, nullptr, 0}) {
  constexpr bool SKIP = CULLS && !BULB && COUNT != 1;
  const bool partial = SPLIT == 1 && split.part >= 0;  // wave-uniform
  const float ka = sb->g.ka, ks = sb->g.ks;
  float ao = 1.0f;
  if (sb->s.enableAmbientOcclusion && !partial) ao = calcAO<BULB, COUNT, SKIP>(sb, p, N, cnt, ubPos);
  // the shadow rays start 0.005·|N| from p
  const float ubSo = SKIP ? fma(0.005f, (sb->cullLip * len(N)) * 1.001f, ubPos) : ubPos;
  V3 total = v3((mat.amb.x * ka) * ao, (mat.amb.y * ka) * ao, (mat.amb.z * ka) * ao);
  const V3 V = normalize(neg(rd));
  const V3 so = shadowOrigin(p, N);
  const int nl = sb->numLights;
  const bool soft = sb->s.enableSoftShadow != 0;
  if (BULB && !RES && COUNT != 1) {
    // hard shadows from directional lights only (wave-uniform test): the shadow marches run as a per-lane queue.  Single-bulb
    // class only: measured 2.46 → 2.37 ms on the 4K bulb frame, but 60 → 63 ms on the 8K Menger frame, whose evaluations are
    // too cheap to pay for the ray set-up inside the loop
    bool queue = !soft && nl > 0;
    for (int i = 0; i < nl; i++) queue = queue && sb->lights[i].type == RM_LIGHT_DIRECTIONAL;
    if (queue) {
      uint32_t need = 0u;
      for (int i = 0; i < nl; i++) {
        const RmLight &li = sb->lights[i];
        const V3 L = normalize(v3(-li.dir[0], -li.dir[1], -li.dir[2]));
        if (!(dot(N, L) <= 0.005f)) need |= 1u << i;  // a light that N·L drops is not marched (see below)
      }
      const uint32_t hitMask = shadowQueue<BULB, COUNT, CULLS>(sb, so, need, far, cnt);
      for (int i = 0; i < nl; i++) {
        const RmLight &li = sb->lights[i];
        const LightGeom g = lightSetup(li, p, far);
        V3 cur;
        if (lightTerm(li, g, mat, N, V, ks, ((hitMask >> i) & 1u) ? 0 : -1, 1.0f, false, cur)) total = add(total, cur);
      }
      return total;
    }
  }
  for (int i = 0; i < nl; i++) {
    const RmLight &li = sb->lights[i];  // uniform index → scalar loads
    if (RES && li.type == RM_LIGHT_AREA) {  // AREA_LIGHT_SAMPLES = 1; the "random" uv is rd.xy (frag:1889)
      V3 p1 = v3(li.points[0][0], li.points[0][1], li.points[0][2]);
      V3 side1 = sub(v3(li.points[1][0], li.points[1][1], li.points[1][2]), p1);
      V3 side2 = sub(v3(li.points[3][0], li.points[3][1], li.points[3][2]), p1);
      V3 toL = sub(madd(side2, rd.y + 0.0f, madd(side1, rd.x + 0.0f, p1)), p);
      V3 L = normalize(toL);
      if (dot(N, L) <= 0.005f) continue;
      MarchRes sh = march<BULB, COUNT, true, CULLS>(sb, so, L, len(toL), 1.0f, cnt, ubSo);
      if (sh.obj != -1 && objs[sh.obj].lightIdx != i) continue;  // only the light's own rectangle may be "in the way"
      total = add(total, getAreaLight(sb, N, V, p, li, mat));
      continue;
    }
    LightGeom g = lightSetup(li, p, far);
    // The reference marches the shadow ray first and only then drops lights with N·L <= 0.005 (frag:1908-1912);
    // the march result of such a light is never read, so it is not marched here (COUNT keeps the reference's
    // work so that the counters stay the algorithmic ones).
    MarchRes sh;
    sh.obj = -1; sh.d = 1.0f; sh.trap = v4(0.0f, 0.0f, 0.0f, 0.0f);
    const bool need = COUNT == 1 || !(dot(N, g.L) <= 0.005f);
    if (partial) {  // this workgroup's one light: the very march every other schedule runs for it; its result travels by memory
      if (i == split.part && need) {
        sh = march<BULB, COUNT, true, CULLS>(sb, so, g.L, g.maxT, 1.0f, cnt, ubSo);
        split.slot[2 * i] = __int_as_float(sh.obj);
        split.slot[2 * i + 1] = sh.d;
      }
      continue;
    }
    if (need) {
      if (SPLIT == 2) { sh.obj = __float_as_int(split.slot[2 * i]); sh.d = split.slot[2 * i + 1]; }
      else sh = march<BULB, COUNT, true, CULLS>(sb, so, g.L, g.maxT, 1.0f, cnt, ubSo);
    }
    V3 cur;
    if (lightTerm(li, g, mat, N, V, ks, sh.obj, sh.d, soft, cur)) total = add(total, cur);
  }
  return total;
}

// Orbit-trap palette of the Mandelbulb (frag:2356-2360), before the ·(phong·8).
RM_DEV V3 bulbTrapColor(float ty, float tz, float tw) {
  V3 c = v3(0.2f, 0.2f, 0.2f);
  c = mix(c, v3(0.10f, 0.20f, 0.30f), clamp_(ty, 0.0f, 1.0f));
  c = mix(c, v3(0.02f, 0.10f, 0.30f), clamp_(tz * tz, 0.0f, 1.0f));
  c = mix(c, v3(0.30f, 0.10f, 0.02f), clamp_(pow_(tw, 6.0f), 0.0f, 1.0f));
  return scale(c, 0.5f);
}

// frag:2318-2375.  `objs` is the per-lane-indexable copy of the object table (LDS).
template <bool BULB, int COUNT, bool TEX, bool CULLS, int SPLIT = 0>
RM_DEV RenderOut render(const SceneBlock *sb, const RmObject *objs, V3 ro, V3 rd, Hit &info, float side, float maxT,
                        V3 bg, Counters &cnt, LightSplit split = LightSplit{-1, nullptr, 0}) {
  RenderOut out;
  info.obj = -1;
  MarchRes res;
  if (SPLIT == 2) {
    // the finishing launch of the light split: the primary march was run — identically — by each of the tile's partial workgroups
    // and stored by the first; running it here again would put it back in series with theirs (a grazing primary ray is as long a
    // chain as a shadow march)
    const float *q = split.slot + 2 * sb->numLights;
    res.obj = __float_as_int(q[0]); res.d = q[1]; res.trap = v4(q[2], q[3], q[4], q[5]);
  } else {
    res = march<BULB, COUNT, false, CULLS>(sb, ro, rd, maxT, side, cnt);  // a miss returns maxT, not res.d
    if (SPLIT == 1 && split.part == 0) {
      float *q = split.slot + 2 * sb->numLights;
      q[0] = __int_as_float(res.obj); q[1] = res.d; q[2] = res.trap.x; q[3] = res.trap.y; q[4] = res.trap.z; q[5] = res.trap.w;
    }
  }
  if (res.obj == -1) {
    out.col = (TEX && sb->s.enableSkyBox) ? sampleCube(sb->skybox, rd) : bg;  // frag:2325-2327
    out.isEnv = 1;
    out.d = maxT;  // frag:2328
    return out;
  }
  out.isEnv = 0;
  out.d = res.d;  // frag:2332
  if (COUNT) cnt.shades++;
  V3 p = madd(rd, res.d, ro);
  // The march stopped at a point whose distance value is below SURFACE_DIST and p lies at most that far (× |rd|) from it: an
  // upper bound of sdScene at p, the seed of the skip test for the normal taps (0.0005 away), the AO taps and the shadow rays.
  constexpr bool SKIP = CULLS && !BULB && COUNT != 1;
  float ubP = __builtin_inff();
  if (SKIP) {
    const float lipLen = (sb->cullLip * len(rd)) * 1.0001f;
    ubP = fma(kSurfaceDist, lipLen, kSurfaceDist) * 1.001f + fma(fabs_(res.d), 1.0e-6f, 1.0e-5f);
  }
  V3 pn = getNormal<BULB, COUNT, SKIP>(sb, p, cnt, SKIP ? fma(0.0005f, sb->cullLip * 1.001f, ubP) : ubP);
  if (sb->s.features & RM_FEAT_PERLIN_BUMP) pn = bumpNormal(pn, p);
  const RmObject &o = objs[BULB ? 0 : res.obj];
  if (TEX && o.isEmissive) {  // frag:2339-2342: the rectangle of an area light; info.obj stays -1
    out.col = v3(o.color[0], o.color[1], o.color[2]);
    return out;
  }
  Material mat;
  mat.amb = v3(o.cAmbient[0], o.cAmbient[1], o.cAmbient[2]);
  mat.dif = getDiffuse<TEX>(sb, o, p);
  mat.spec = v3(o.cSpecular[0], o.cSpecular[1], o.cSpecular[2]);
  mat.shininess = o.shininess;
  const int type = BULB ? (int)RM_MANDELBULB : o.type;
  V3 ph = getPhong<BULB, COUNT, TEX, CULLS, SPLIT>(sb, objs, mat, pn, p, rd, maxT, cnt, ubP, split);
  V3 col = ph;
  if (type == RM_MANDELBULB) {  // frag:2354-2361
    V3 c = bulbTrapColor(res.trap.y, res.trap.z, res.trap.w);
    col = v3(c.x * (ph.x * 8.0f), c.y * (ph.y * 8.0f), c.z * (ph.z * 8.0f));
  } else if (type == RM_MENGERSPONGE) {  // frag:2362-2365
    V3 c = v3(fma(0.5f, cos_(fma(2.0f, res.trap.z, 0.0f)), 0.5f), fma(0.5f, cos_(fma(2.0f, res.trap.z, 1.0f)), 0.5f),
              fma(0.5f, cos_(fma(2.0f, res.trap.z, 2.0f)), 0.5f));
    col = mul(c, ph);
  }
  info.p = p; info.n = pn; info.rd = rd; info.obj = res.obj;
  out.col = col;
  return out;
}

// The vertex shader's outputs are varyings: raymarch.vert:18-24 is evaluated at the four corners of the full-screen quad
// (triangles TL-BL-BR and TR-TL-BR, realtimerender.cpp:225-238) and every fragment receives the affine interpolation over
// its own triangle, P0 + I·(P1 − P0) + J·(P2 − P0).  With P0 the triangle's right-angle corner: below the diagonal
// (tx + ty <= 1) P0 = BL, I = tx, J = ty; above it P0 = TR, I = 1 − tx, J = 1 − ty; (tx, ty) = the pixel centre in [0,1]².
struct QuadCoord { float I, J; bool upper; };
RM_DEV QuadCoord quadCoord(int px, int py, int W, int H) {
  const float tx = ((float)px + 0.5f) / (float)W, ty = ((float)py + 0.5f) / (float)H;
  QuadCoord q;
  q.upper = (tx + ty) > 1.0f;
  q.I = q.upper ? 1.0f - tx : tx;
  q.J = q.upper ? 1.0f - ty : ty;
  return q;
}
// twoDFragCoord = pos (vert:18): −1 + 2·I below the diagonal, 1 − 2·I above
RM_DEV void pixelNdc(int px, int py, int W, int H, float &ndcx, float &ndcy) {
  const QuadCoord q = quadCoord(px, py, W, H);
  ndcx = q.upper ? fma(q.I, -2.0f, 1.0f) : fma(q.I, 2.0f, -1.0f);
  ndcy = q.upper ? fma(q.J, -2.0f, 1.0f) : fma(q.J, 2.0f, -1.0f);
}
// nearClip / farClip (vert:23-24) interpolated from the corner values the launcher staged in sb->rayPlane, then
// frag:2388-2392: ro on the near plane, rd toward the far plane.
RM_DEV void primaryRay(const SceneBlock *sb, int px, int py, int W, int H, V3 &ro, V3 &rd) {
  const QuadCoord q = quadCoord(px, py, W, H);
  float nc[4], fc4[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {  // both triangles from scalar registers, then a per-lane select (wave-uniform almost everywhere)
    const float (*A)[3][4] = sb->rayPlane[0], (*B)[3][4] = sb->rayPlane[1];
    // readfirstlane pins each coefficient to a scalar register: otherwise the compiler may turn the select of two loads into
    // one per-lane load through a selected pointer (six global_load_dwordx4 in the wavefront pipeline's refill path)
    auto sc = [](float v) { return u2f((uint32_t)__builtin_amdgcn_readfirstlane((int)f2u(v))); };
    const float p0n = q.upper ? sc(B[0][0][k]) : sc(A[0][0][k]), p1n = q.upper ? sc(B[0][1][k]) : sc(A[0][1][k]),
                p2n = q.upper ? sc(B[0][2][k]) : sc(A[0][2][k]);
    const float p0f = q.upper ? sc(B[1][0][k]) : sc(A[1][0][k]), p1f = q.upper ? sc(B[1][1][k]) : sc(A[1][1][k]),
                p2f = q.upper ? sc(B[1][2][k]) : sc(A[1][2][k]);
    nc[k] = fma(q.J, p2n, fma(q.I, p1n, p0n));
    fc4[k] = fma(q.J, p2f, fma(q.I, p1f, p0f));
  }
  ro = v3(nc[0] / nc[3], nc[1] / nc[3], nc[2] / nc[3]);          // frag:2388 (IEEE: once per pixel, and every bit of rd counts, §2.3)
  V3 fc = v3(fc4[0] / fc4[3], fc4[1] / fc4[3], fc4[2] / fc4[3]);  // frag:2389
  rd = normalize(sub(fc, ro));                                    // frag:2392
}
RM_DEV V3 backgroundColor(const SceneBlock *sb) {  // frag:2405-2419 without SKY_BACKGROUND; later #ifdefs override earlier ones
  V3 bg = v3(0.0f, 0.0f, 0.0f);
  if (sb->s.features & RM_FEAT_WHITE_BACKGROUND) bg = v3(1.0f, 1.0f, 1.0f);
  if (sb->s.features & RM_FEAT_DARK_BACKGROUND) bg = v3(0.0f, 0.0f, 0.0f);
  return bg;
}
RM_DEV V3 backgroundColor(const SceneBlock *sb, V3 rd) {  // frag:2405-2419
  V3 bg = v3(0.0f, 0.0f, 0.0f);
  if (sb->s.features & RM_FEAT_SKY_BACKGROUND) bg = getSky(rd);
  if (sb->s.features & RM_FEAT_NIGHTSKY_BACKGROUND) bg = getMoonColor(sb->noise, sb->g.iTime, rd);
  if (sb->s.features & RM_FEAT_WHITE_BACKGROUND) bg = v3(1.0f, 1.0f, 1.0f);
  if (sb->s.features & RM_FEAT_DARK_BACKGROUND) bg = v3(0.0f, 0.0f, 0.0f);
  return bg;
}

// raymarch.vert:13-25 + frag:2383-2427 + frag:2429-2575 for the pixel centre (px, py), py = 0 at the bottom.
// ENV = false compiles the procedural layers out (the launcher picks the instantiation from the feature bits),
// so the common kernels do not carry their registers and code.
// SEC = false compiles main's secondary rays out (reflection loop, refraction): the launcher picks it when the settings or the
// materials rule them out for the whole frame, so that what render() hands over for them (hit point, normal, direction) is not
// carried across the shadow marches — fewer registers spilled around the hot loops, the same pixels.
template <bool BULB, int COUNT, bool ENV, bool TEX, bool SEC = true, int SPLIT = 0>
RM_DEV void shadePixel(const SceneBlock *sb, const RmObject *objs, int px, int py, int W, int H, V4 &fragColor,
                       V4 &bright, Counters &cnt, bool &hitFlag, LightSplit split = LightSplit{-1, nullptr, 0}) {
  float ndcx, ndcy;
  pixelNdc(px, py, W, H, ndcx, ndcy);
  bright = v4(0.0f, 0.0f, 0.0f, 1.0f);
  hitFlag = false;
  if (sb->g.isTwoD) {  // frag:2431, 2377-2380
    float s = sdMandelBrot(sb, ndcx, ndcy);
    fragColor = v4(pow_(s, 0.9f), pow_(s, 1.1f), pow_(s, 1.4f), 1.0f);
    return;
  }
  V3 ro, rd;
  primaryRay(sb, px, py, W, H, ro, rd);
  const V3 bg = ENV ? backgroundColor(sb, rd) : backgroundColor(sb);
  const uint32_t feat = sb->s.features;
  const bool env = ENV && (feat & (RM_FEAT_TERRAIN | RM_FEAT_CLOUD | RM_FEAT_SEA)) != 0;
  const float far = (ENV && (feat & RM_FEAT_CLOUD)) ? 2000.0f : sb->cam.initialFar;  // frag:2422-2426
  const float iTime = sb->g.iTime;

  Hit info;
  RenderOut ri = render<BULB, COUNT, TEX, !ENV, SPLIT>(sb, objs, ro, rd, info, 1.0f, far, bg, cnt, split);  // frag:2443
  EnvOut e;
  e.terrainHit = false; e.cloudHit = false; e.seaHit = false;
  if (env) e = envLayers(feat, sb->noise, iTime, W, ro, rd, ri.d, bg, cnt);  // frag:2444-2456
  if (ri.isEnv && !e.cloudHit && !e.terrainHit && !e.seaHit) {  // frag:2459-2465
    fragColor = v4(ri.col.x, ri.col.y, ri.col.z, 1.0f);
    return;
  }
  if (e.cloudHit || e.terrainHit || e.seaHit) {  // frag:2466-2474: cloud wins over terrain, terrain over sea
    V3 c = e.cloudHit ? e.ccol : (e.terrainHit ? e.tcol : e.scol);
    fragColor = v4(c.x, c.y, c.z, 1.0f);
    if (dot(c, v3(0.2126f, 0.7152f, 0.0722f)) > 1.0f) bright = v4(c.x, c.y, c.z, 1.0f);
    return;
  }
  hitFlag = true;
  V4 phong = v4(ri.col.x, ri.col.y, ri.col.z, 1.0f);
  V4 refl = v4(0.0f, 0.0f, 0.0f, 0.0f), refr = v4(0.0f, 0.0f, 0.0f, 0.0f);
  const Hit oi = info;  // frag:2481
  // UB5: an emissive hit leaves info.obj = -1 and the shader reads objects[-1] (frag:2341, 2483); that read is
  // taken as zeros (robust-access behaviour), i.e. no secondary rays.
  const bool noObj = TEX && info.obj < 0;
  const RmObject &o = objs[(BULB || noObj) ? 0 : info.obj];
  const V3 cRefl = noObj ? v3(0.0f, 0.0f, 0.0f) : v3(o.cReflective[0], o.cReflective[1], o.cReflective[2]);
  const V3 cRefr = noObj ? v3(0.0f, 0.0f, 0.0f) : v3(o.cTransparent[0], o.cTransparent[1], o.cTransparent[2]);
  const float ior = o.ior;
  if (SEC && sb->s.enableReflection && len(cRefl) != 0.0f) {  // frag:2491-2524
    V3 fil = v3(1.0f, 1.0f, 1.0f);
    const int nb = sb->s.numReflection;
    for (int i = 0; i < nb; i++) {
      V3 r = reflect(info.rd, info.n);
      V3 sro = v3(fma(r.x * kSurfaceDist, 3.0f, info.p.x), fma(r.y * kSurfaceDist, 3.0f, info.p.y),
                  fma(r.z * kSurfaceDist, 3.0f, info.p.z));
      fil = mul(fil, cRefl);
      RenderOut res = render<BULB, COUNT, TEX, !ENV>(sb, objs, sro, r, info, 1.0f, far, bg, cnt);
      if (env) {  // frag:2506-2518 (a sea hit sets sr.isEnv, not res.isEnv: the bounce loop goes on)
        EnvOut b = envLayers(feat, sb->noise, iTime, W, sro, r, res.d, bg, cnt);
        if (b.seaHit) res.col = b.scol;
        if (b.terrainHit) { res.col = b.tcol; res.isEnv = 1; }
        if (b.cloudHit) { res.col = b.ccol; res.isEnv = 1; }
      }
      refl.x += (sb->g.ks * fil.x) * res.col.x;
      refl.y += (sb->g.ks * fil.y) * res.col.y;
      refl.z += (sb->g.ks * fil.z) * res.col.z;
      refl.w += 1.0f;
      if (res.isEnv) break;
    }
  }
  if (SEC && sb->s.enableRefraction && len(cRefr) != 0.0f) {  // frag:2526-2570
    V3 rdIn = refract(oi.rd, oi.n, 1.0f / ior);
    V3 pEnter = v3(fma(-(oi.n.x * kSurfaceDist), 3.0f, oi.p.x), fma(-(oi.n.y * kSurfaceDist), 3.0f, oi.p.y),
                   fma(-(oi.n.z * kSurfaceDist), 3.0f, oi.p.z));
    float dIn = march<BULB, COUNT, false>(sb, pEnter, rdIn, far, -1.0f, cnt).d;
    V3 pExit = madd(rdIn, dIn, pEnter);
    V3 nExit = neg(getNormal<BULB, COUNT>(sb, pExit, cnt));
    V3 rdOut = refract(rdIn, nExit, ior);
    if (len(rdOut) != 0.0f) {
      V3 sro = v3(fma(-(nExit.x * kSurfaceDist), 5.0f, pExit.x), fma(-(nExit.y * kSurfaceDist), 5.0f, pExit.y),
                  fma(-(nExit.z * kSurfaceDist), 5.0f, pExit.z));
      RenderOut res = render<BULB, COUNT, TEX, !ENV>(sb, objs, sro, rdOut, info, 1.0f, far, bg, cnt);
      if (env) {  // frag:2555-2567
        EnvOut b = envLayers(feat, sb->noise, iTime, W, sro, rdOut, res.d, bg, cnt);
        if (b.seaHit) res.col = b.scol;
        if (b.terrainHit) res.col = b.tcol;
        if (b.cloudHit) res.col = b.ccol;
      }
      refr.x += (sb->g.kt * cRefr.x) * res.col.x;
      refr.y += (sb->g.kt * cRefr.y) * res.col.y;
      refr.z += (sb->g.kt * cRefr.z) * res.col.z;
      refr.w += 1.0f;
    }
  }
  fragColor = v4((phong.x + refl.x) + refr.x, (phong.y + refl.y) + refr.y, (phong.z + refl.z) + refr.z,
                 (phong.w + refl.w) + refr.w);  // frag:2572
  V3 c = v3(fragColor.x, fragColor.y, fragColor.z);
  float brightness = dot(c, v3(0.2126f, 0.7152f, 0.0722f));  // frag:1938-1946
  if (brightness > 1.0f) bright = v4(c.x, c.y, c.z, 1.0f);
}

}  // namespace rm
