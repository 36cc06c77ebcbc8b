/*
 * oracle/rm_oracle.c — TEST INFRASTRUCTURE.  CPU restatement of the reference's per-pixel raymarch
 * (resources/raymarch.vert + resources/raymarch.frag of KentaYoshii/Raymarcher), scalar binary32,
 * function by function, each citing the shader lines it follows ("frag:N").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.  The product
 * (raymarcher_amd/csrc) never links or calls it.
 *
 * PARITY STATUS: the reference has no tests or golden vectors for this path ("parity unpinned" by the
 * reference's own fixtures) and its desktop-GL shaders cannot be run unmodified in this container.  This
 * restatement is pinned instead by (i) outputs of the reference's OWN shaders executed here on a software
 * GLES rasteriser after a mechanical ESSL adaptation (tests/golden/glsl/: 24 frames, 14 function probes,
 * 6 post-pass cases; generator oracle/tools/gen_glsl_goldens.py), (ii) analytic known-answer tests and
 * float64 models, (iii) accuracy tests of rm_math.h against libm — DESIGN.md §2 lists what each pins and how
 * tightly.  The host tables (loader, camera) are pinned by the reference's own loader built as-is (oracle/ref).
 *
 * Numeric contract: oracle/rm_math.h (scalar built-ins) + the vector forms below.  Build with
 * -ffp-contract=off: an fma appears only where rm_fma is written.
 *
 * Decisions on the shader's undefined behaviour (SURVEY §8a "UB"), all documented in DESIGN.md §4:
 *   UB1 softshadow miss: r.d = res (the penumbra factor the author evidently meant, frag:1718-1722)
 *   UB2 raymarch miss:   res.d = rayDepth ("distance travelled along ray direction", frag:190-191)
 *   UB3 sdScene trap:    the trap of the LAST fractal object evaluated, as written (frag:1419-1428)
 *   UB4 unknown / CUSTOM type: object is skipped (never nearest)
 *   UB5 emissive hit then objects[-1] (frag:2341, 2483): the out-of-range uniform read is zeros (no secondary rays)
 *   UB7 seaRender's bare `return;` (frag:2290): returns ri as filled so far; SEA_TIME = 1 + iTime*0.5
 *   UB8 LTC table filter undefined in divergent flow: GL_LINEAR; 8-bit storage of the unsized GL_RGBA upload
 *   UB9 2-D mode BrightColor: (0,0,0,1)
 *   UB10 cloudsMap leaves `nnd` unset outside the cloud: nnd = -d always (iq's original order)
 */
#include "rm_oracle.h"
#include "rm_math.h"

#include <stdlib.h>

/* Work-trace hooks: no-ops here.  scripts/sim/wave_sim.c (an offline schedule simulator, not a test and not the
 * product) defines them before including this file to record, per pixel, the Mandelbulb iteration count of every
 * sdScene evaluation and where each march begins. */
#ifndef RMO_TRACE_EVAL
#define RMO_TRACE_EVAL(c, iters) ((void)0)
#define RMO_TRACE_MARCH(c, kind, ro, rd, endp) ((void)0) /* kind 0 = raymarch, 1 = softshadow; may lower *endp */
#define RMO_TRACE_SKIP_SHADOW(c, N, L) 0                 /* 1 = do not march a shadow ray whose light is dropped anyway */
#endif
#ifndef RMO_TRACE_SCENE_EVAL
#define RMO_TRACE_SCENE_EVAL(c) ((void)0)                /* every sdScene evaluation (any scene) */
#endif

/* ---------------------------------------------------------------- vector forms of the contract */
typedef struct { float x, y; } v2;
typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v4 V4(float x, float y, float z, float w) { v4 r = {x, y, z, w}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* a*s + b, fused per component */
static inline v3 v3_madd(v3 a, float s, v3 b) {
  return V3(rm_fma(a.x, s, b.x), rm_fma(a.y, s, b.y), rm_fma(a.z, s, b.z));
}
/* GLSL dot: products accumulated left to right, each accumulation fused */
static inline float dot2(float ax, float ay, float bx, float by) { return rm_fma(ay, by, ax * bx); }
static inline float dot3(v3 a, v3 b) { return rm_fma(a.z, b.z, rm_fma(a.y, b.y, a.x * b.x)); }
static inline float len2(float x, float y) { return rm_sqrt(dot2(x, y, x, y)); }
static inline float len3(v3 a) { return rm_sqrt(dot3(a, a)); }
/* GLSL normalize(v) = v · (1 / length(v)) */
static inline v3 normalize3(v3 a) { float inv = 1.0f / len3(a); return v3_scale(a, inv); }
/* GLSL reflect(I,N) = I − 2·dot(N,I)·N */
static inline v3 reflect3(v3 I, v3 N) { float k = 2.0f * dot3(N, I); return v3_madd(N, -k, I); }
/* GLSL refract(I,N,eta) */
static inline v3 refract3(v3 I, v3 N, float eta) {
  float d = dot3(N, I);
  float k = rm_fma(-(eta * eta), rm_fma(-d, d, 1.0f), 1.0f);
  if (k < 0.0f) return V3(0.0f, 0.0f, 0.0f);
  float t = rm_fma(eta, d, rm_sqrt(k));
  return V3(rm_fma(-t, N.x, eta * I.x), rm_fma(-t, N.y, eta * I.y), rm_fma(-t, N.z, eta * I.z));
}
static inline v3 mix3(v3 a, v3 b, float t) {
  return V3(rm_mix(a.x, b.x, t), rm_mix(a.y, b.y, t), rm_mix(a.z, b.z, t));
}
/* vec3(M * vec4(p,1)), M column-major: ((M3 + M0·x) + M1·y) + M2·z, fused */
static inline v3 xform_point(const float *M, v3 p) {
  v3 r;
  r.x = rm_fma(M[8], p.z, rm_fma(M[4], p.y, rm_fma(M[0], p.x, M[12])));
  r.y = rm_fma(M[9], p.z, rm_fma(M[5], p.y, rm_fma(M[1], p.x, M[13])));
  r.z = rm_fma(M[10], p.z, rm_fma(M[6], p.y, rm_fma(M[2], p.x, M[14])));
  return r;
}
/* M * vec4(x,y,z,w): ((M0·x + M1·y) + M2·z) + M3·w, fused */
static inline v4 mat4_mul_v4(const float *M, float x, float y, float z, float w) {
  v4 r;
  r.x = rm_fma(M[12], w, rm_fma(M[8], z, rm_fma(M[4], y, M[0] * x)));
  r.y = rm_fma(M[13], w, rm_fma(M[9], z, rm_fma(M[5], y, M[1] * x)));
  r.z = rm_fma(M[14], w, rm_fma(M[10], z, rm_fma(M[6], y, M[2] * x)));
  r.w = rm_fma(M[15], w, rm_fma(M[11], z, rm_fma(M[7], y, M[3] * x)));
  return r;
}

/* nearClip / farClip are VARYINGS (vert:9-10): the vertex shader evaluates vert:23-24 at the four corners of the full-screen
 * quad only — two triangles, TL-BL-BR and TR-TL-BR (realtimerender.cpp:225-238) — and the rasteriser hands every fragment
 * the affine interpolation over ITS triangle, P0 + I·(P1 − P0) + J·(P2 − P0) (gl_Position.w = 1: perspective-correct =
 * linear).  Restated with P0 the triangle's right-angle corner: below the diagonal (tx + ty <= 1) P0 = BL, P1 = BR, P2 = TL,
 * I = tx, J = ty; above it P0 = TR, P1 = TL, P2 = BR, I = 1 − tx, J = 1 − ty, with (tx, ty) the pixel centre in [0,1]².
 * This matters: invProjView·(x, y, ±1, 1) cancels from terms of magnitude |eye|·(far+near)/(2·far·near) down to ~far⁻¹, so
 * evaluating it PER PIXEL (what a first restatement did) puts an independent ~1e-5 relative error on every primary ray,
 * which bump-mapped reflections amplify to visible speckle (84 of 2304 pixels of lighting/reflections_complex.json more
 * than 1e-3 away from the reference shader's frame; 3 with this form).  In the real pipeline those rounding errors sit in
 * the four corner values — a camera nudge common to the whole frame — and the per-pixel part is two fused multiply-adds. */
static void rayPlanes(const float *M, v4 P[2][2][3]) {
  for (int tri = 0; tri < 2; tri++) {
    const float sg = tri ? 1.0f : -1.0f; /* P0 = (sg, sg), P1 = (−sg, sg), P2 = (sg, −sg) */
    for (int k = 0; k < 2; k++) {
      const float z = k ? 1.0f : -1.0f;
      const v4 p0 = mat4_mul_v4(M, sg, sg, z, 1.0f), p1 = mat4_mul_v4(M, -sg, sg, z, 1.0f), p2 = mat4_mul_v4(M, sg, -sg, z, 1.0f);
      P[tri][k][0] = p0;
      P[tri][k][1] = V4(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z, p1.w - p0.w);
      P[tri][k][2] = V4(p2.x - p0.x, p2.y - p0.y, p2.z - p0.z, p2.w - p0.w);
    }
  }
}
static inline v4 interpolateVarying(const v4 *P, float I, float J) {
  return V4(rm_fma(J, P[2].x, rm_fma(I, P[1].x, P[0].x)), rm_fma(J, P[2].y, rm_fma(I, P[1].y, P[0].y)),
            rm_fma(J, P[2].z, rm_fma(I, P[1].z, P[0].z)), rm_fma(J, P[2].w, rm_fma(I, P[1].w, P[0].w)));
}

/* ---------------------------------------------------------------- context */
#define SURFACE_DIST 0.001f /* frag:32 */
#define OUTSIDE 1.0f        /* frag:26 */
#define INSIDE (-1.0f)      /* frag:25 */

typedef struct {
  const RmCamera *cam;
  const RmObject *objs;
  int numObjects;
  const RmLight *lights;
  int numLights;
  RmGlobals g;
  RmSettings s;
  const RmTexture *tex;
  int numTex;
  const RmResources *res; /* noise / skybox / LTC tables (host pointers); never NULL inside the renderer */
  int W;                  /* screenDimensions.x (frag:246, realtimerender.cpp:622-629) */
  v4 rayPlane[2][2][3];   /* [triangle][near, far][P0, P1 − P0, P2 − P0] of nearClip / farClip, see rayPlanes */
  uint64_t nEval, nIter, nHit, nShade, nShape; /* per-thread work counters */
} Ctx;

typedef struct { int minObjIdx; float minD; v4 trap; } SceneMin;        /* frag:170-182 */
typedef struct { int intersectObj; float d; v4 trap; } RayMarchRes;     /* frag:184-196 */
typedef struct { v3 rd, p, n; int intersectObj; } IntersectionInfo;     /* frag:198-210 */
typedef struct { v4 fragColor; float d; int isEnv; } RenderInfo;        /* frag:232-241 */

/* ---------------------------------------------------------------- SDF primitives (frag:832-1019) */
/* frag:843-846 */
static float sdBox(v3 p, v3 b) {
  v3 q = V3(rm_abs(p.x) - b.x, rm_abs(p.y) - b.y, rm_abs(p.z) - b.z);
  v3 qm = V3(rm_max(q.x, 0.0f), rm_max(q.y, 0.0f), rm_max(q.z, 0.0f));
  return len3(qm) + rm_min(rm_max(q.x, rm_max(q.y, q.z)), 0.0f);
}
/* frag:832-834 */
static float sdSphere(v3 p, float r) { return len3(p) - r; }
/* frag:852-861 */
static float sdCone(v3 p, float r, float h) {
  float pox = len2(p.x, p.z) - r, poy = p.y + h;
  float ex = -r, ey = 2.0f * h;
  float t = rm_clamp(rm_divr(dot2(pox, poy, ex, ey), dot2(ex, ey, ex, ey)), 0.0f, 1.0f);
  float qx = rm_fma(-ex, t, pox), qy = rm_fma(-ey, t, poy);
  float d = len2(qx, qy);
  if (rm_max(qx, qy) > 0.0f) return d;
  return -rm_min(d, poy);
}
/* frag:867-870 */
static float sdCylinder(v3 p, float h, float r) {
  float dx = rm_abs(len2(p.x, p.z)) - r, dy = rm_abs(p.y) - h;
  return rm_min(rm_max(dx, dy), 0.0f) + len2(rm_max(dx, 0.0f), rm_max(dy, 0.0f));
}
/* frag:875-886 */
static float sdOctahedron(v3 p, float s) {
  p = V3(rm_abs(p.x), rm_abs(p.y), rm_abs(p.z));
  float m = ((p.x + p.y) + p.z) - s;
  v3 r = V3(rm_fma(3.0f, p.x, -m), rm_fma(3.0f, p.y, -m), rm_fma(3.0f, p.z, -m));
  v3 q;
  if (r.x < 0.0f) q = p;
  else if (r.y < 0.0f) q = V3(p.y, p.z, p.x);
  else if (r.z < 0.0f) q = V3(p.z, p.x, p.y);
  else return m * 0.57735027f;
  float k = rm_clamp(0.5f * ((q.z - q.y) + s), 0.0f, s);
  return len3(V3(q.x, (q.y - s) + k, q.z - k));
}
/* frag:891-894 */
static float sdTorus(v3 p, float tx, float ty) {
  float qx = len2(p.x, p.z) - tx;
  return len2(qx, p.y) - ty;
}
/* frag:991-994 */
static float sdCapsule(v3 p, float h, float r) {
  p.y = p.y - rm_clamp(p.y, 0.0f, h);
  return len3(p) - r;
}
/* frag:1005-1019 */
static float sdDeathStar(v3 p2, float ra, float rb, float d) {
  float px = p2.x, py = len2(p2.y, p2.z);
  float a = (((ra * ra) - (rb * rb)) + (d * d)) / (2.0f * d);
  float b = rm_sqrt(rm_max((ra * ra) - (a * a), 0.0f));
  if (rm_fma(px, b, -(py * a)) > d * rm_max(b - py, 0.0f)) {
    return len2(px - a, py - b);
  }
  return rm_max(len2(px, py) - ra, -(len2(px - d, py) - rb));
}

/* ---------------------------------------------------------------- fractals */
/* frag:751-769 */
static float sdMandelBrot(const Ctx *c, float px, float py) {
  float ltime = rm_fma(-0.5f, rm_cos(c->g.iTime * 0.06f), 0.5f);
  float zoom = rm_pow(0.9f, 50.0f * ltime);
  float k = (0.045f * zoom) * rm_fma(-ltime, 0.5f, 1.0f);
  float cx = -0.745f - k, cy = 0.186f - k;
  float ld2 = 1.0f;
  float lz2 = dot2(px, py, px, py);
  for (int i = 0; i < c->s.maxSteps; i++) {
    ld2 = ld2 * (4.0f * lz2);
    float nx = rm_fma(px, px, -(py * py)) + cx;
    float ny = rm_fma(2.0f * px, py, cy);
    px = nx; py = ny;
    lz2 = dot2(px, py, px, py);
    if (lz2 > 200.0f) break;
  }
  float d = rm_sqrt(lz2 / ld2) * rm_log(lz2);
  return rm_sqrt(rm_clamp((150.0f / zoom) * d, 0.0f, 1.0f));
}

/* frag:775-803 */
static float sdMandelBulb(Ctx *c, v3 pos, v4 *resColor) {
  const float power = c->g.power;
  v3 w = pos;
  float m = dot3(w, w);
  v4 trap = V4(rm_abs(w.x), rm_abs(w.y), rm_abs(w.z), m);
  float dz = 1.0f;
  v3 cc = pos;
  /* frag:782-784 */
  if (len2(c->g.juliaSeed[0], c->g.juliaSeed[1]) != 0.0f) cc = V3(c->g.juliaSeed[0], c->g.juliaSeed[1], 0.0f);
  const float pexp = (power - 1.0f) / 2.0f;
  if ((c->s.features & RM_FEAT_BULB_POWER8_ALGEBRAIC) && power == 8.0f) {
    /* Opt-in evaluation scheme of the same step (see RM_FEAT_BULB_POWER8_ALGEBRAIC): with θ = acos(y/r), φ = atan(x, z),
     * ρ = |w.xz|:  (y + iρ)^8 = r^8·(cos 8θ + i sin 8θ),  ((z + ix)/ρ)^8 = cos 8φ + i sin 8φ;  m^3.5 = m³·√m. */
    for (int i = 0; i < c->s.fractalIters; i++) {
      c->nIter++;
      float r = rm_sqrt(m);
      dz = rm_fma(8.0f * (((m * m) * m) * r), dz, 1.0f);
      float rho = rm_sqrt(dot2(w.x, w.z, w.x, w.z));
      float inv = 1.0f / rho;
      float cz = (rho == 0.0f) ? 1.0f : w.z * inv, sx = (rho == 0.0f) ? 0.0f : w.x * inv; /* atan(0,0) = 0 in the contract */
      float re = w.y, im = rho;
      for (int k = 0; k < 3; k++) {
        float t = rm_fma(re, re, -(im * im));
        im = 2.0f * (re * im);
        re = t;
        t = rm_fma(cz, cz, -(sx * sx));
        sx = 2.0f * (cz * sx);
        cz = t;
      }
      w = V3(rm_fma(im, sx, cc.x), re + cc.y, rm_fma(im, cz, cc.z));
      trap = V4(rm_min(trap.x, rm_abs(w.x)), rm_min(trap.y, rm_abs(w.y)), rm_min(trap.z, rm_abs(w.z)),
                rm_min(trap.w, m));
      m = dot3(w, w);
      if (m > 2.0f) break;
    }
    *resColor = V4(m, trap.y, trap.z, trap.w);
    return rm_divr((0.25f * rm_log(m)) * rm_sqrt(m), dz);
  }
  int nTrace = 0;
  for (int i = 0; i < c->s.fractalIters; i++) {
    c->nIter++;
    nTrace++;
    /* frag:787 */
    dz = rm_fma(power * rm_pow(m, pexp), dz, 1.0f);
    /* frag:789-793 */
    float r = len3(w);
    float b = power * rm_acos(rm_divr(w.y, r));
    float a = power * rm_atan2(w.x, w.z);
    float pr = rm_pow(r, power);
    float sb = rm_sin(b), cb = rm_cos(b), sa = rm_sin(a), ca = rm_cos(a);
    w = V3(rm_fma(pr, sb * sa, cc.x), rm_fma(pr, cb, cc.y), rm_fma(pr, sb * ca, cc.z));
    /* frag:795 — uses the OLD m */
    trap = V4(rm_min(trap.x, rm_abs(w.x)), rm_min(trap.y, rm_abs(w.y)), rm_min(trap.z, rm_abs(w.z)),
              rm_min(trap.w, m));
    /* frag:797-798 */
    m = dot3(w, w);
    if (m > 2.0f) break;
  }
  RMO_TRACE_EVAL(c, nTrace);
  *resColor = V4(m, trap.y, trap.z, trap.w);
  /* frag:802 */
  return rm_divr((0.25f * rm_log(m)) * rm_sqrt(m), dz);
}

/* frag:808-827 */
static float sdSierpinski(v3 p) {
  const float Scale = 1.85f, Offset = 2.0f;
  const float k = Offset * (Scale - 1.0f);
  for (int n = 0; n < 14; n++) {
    if (p.x + p.y < 0.0f) { float t = p.x; p.x = -p.y; p.y = -t; }
    if (p.x + p.z < 0.0f) { float t = p.x; p.x = -p.z; p.z = -t; }
    if (p.y + p.z < 0.0f) { float t = p.z; p.z = -p.y; p.y = -t; }
    p = V3(rm_fma(p.x, Scale, -k), rm_fma(p.y, Scale, -k), rm_fma(p.z, Scale, -k));
  }
  return len3(p) * rm_pow(Scale, -14.0f);
}

/* const mat3 ma, frag:124-126 (columns (.6,0,.8),(0,1,0),(-.8,0,.6)) */
static inline v3 mul_ma(v3 v) {
  return V3(rm_fma(-0.80f, v.z, rm_fma(0.00f, v.y, 0.60f * v.x)),
            rm_fma(0.00f, v.z, rm_fma(1.00f, v.y, 0.00f * v.x)),
            rm_fma(0.60f, v.z, rm_fma(0.00f, v.y, 0.80f * v.x)));
}
/* frag:1049-1071 */
static float sdMengerSponge(const Ctx *c, v3 p, v4 *res) {
  float d = sdBox(p, V3(1.0f, 1.0f, 1.0f));
  *res = V4(d, 1.0f, 0.0f, 0.0f);
  float ani = rm_smoothstep(-0.2f, 0.2f, -rm_cos(0.5f * c->g.iTime));
  float off = 1.5f * rm_sin(0.01f * c->g.iTime);
  float s = 1.0f;
  for (int m = 0; m < c->s.mengerLevels; m++) {
    /* frag:1057 */
    p = mix3(p, mul_ma(V3(p.x + off, p.y + off, p.z + off)), ani);
    /* frag:1058-1060 */
    v3 a = V3(rm_mod_pow2(p.x * s, 2.0f) - 1.0f, rm_mod_pow2(p.y * s, 2.0f) - 1.0f, rm_mod_pow2(p.z * s, 2.0f) - 1.0f);
    s = s * 3.0f;
    v3 r = V3(rm_abs(rm_fma(-3.0f, rm_abs(a.x), 1.0f)), rm_abs(rm_fma(-3.0f, rm_abs(a.y), 1.0f)),
              rm_abs(rm_fma(-3.0f, rm_abs(a.z), 1.0f)));
    float da = rm_max(r.x, r.y), db = rm_max(r.y, r.z), dc = rm_max(r.z, r.x);
    float cc = (rm_min(da, rm_min(db, dc)) - 1.0f) / s;
    if (cc > d) {
      d = cc;
      *res = V4(d, rm_min(res->y, ((0.2f * da) * db) * dc), (1.0f + (float)m) / 4.0f, 0.0f);
    }
  }
  return d;
}

/* ---------------------------------------------------------------- scene union (frag:1262-1293, 1406-1430) */
static SceneMin sdScene(Ctx *c, v3 p) {
  SceneMin res;
  float minD = 1000000.0f;
  int minObj = -1;
  v4 trapCol = V4(0.0f, 0.0f, 0.0f, 0.0f);
  c->nEval++;
  RMO_TRACE_SCENE_EVAL(c);
  for (int i = 0; i < c->numObjects; i++) {
    const RmObject *obj = &c->objs[i];
    v3 po = xform_point(obj->invModel, p); /* frag:1417 */
    float d;
    switch (obj->type) { /* sdMatch, frag:1262-1293 */
      case RM_CUBE: d = sdBox(po, V3(0.5f, 0.5f, 0.5f)); break;
      case RM_CONE: d = sdCone(po, 0.5f, 0.5f); break;
      case RM_CYLINDER: d = sdCylinder(po, 0.5f, 0.5f); break;
      case RM_SPHERE: d = sdSphere(po, 0.5f); break;
      case RM_OCTAHEDRON: d = sdOctahedron(po, 0.5f); break;
      case RM_TORUS: d = sdTorus(po, 0.5f, 0.125f); break;
      case RM_CAPSULE: d = sdCapsule(po, 0.5f, 0.1f); break;
      case RM_DEATHSTAR: d = sdDeathStar(po, 0.5f, 0.35f, 0.5f); break;
      case RM_RECTANGLE: d = sdBox(po, V3(0.5f, 0.5f, 0.0f)); break;
      case RM_MANDELBROT: d = sdMandelBrot(c, po.x, po.y); break;
      case RM_MANDELBULB: d = sdMandelBulb(c, po, &trapCol); break;
      case RM_MENGERSPONGE: d = sdMengerSponge(c, po, &trapCol); break;
      case RM_SIERPINSKI: d = sdSierpinski(po); break;
      default: continue; /* UB4 */
    }
    c->nShape++;
    float currD = d * obj->scaleFactor; /* frag:1419 */
    if (currD < minD) { minD = currD; minObj = i; }
  }
  res.minD = minD; res.minObjIdx = minObj; res.trap = trapCol; /* UB3 */
  return res;
}

/* frag:1436-1444 */
static v3 getNormal(Ctx *c, v3 p) {
  const float ex = (1.0f * 0.5773f) * 0.0005f, ey = (-1.0f * 0.5773f) * 0.0005f;
  float d1 = sdScene(c, V3(p.x + ex, p.y + ey, p.z + ey)).minD; /* e.xyy */
  float d2 = sdScene(c, V3(p.x + ey, p.y + ey, p.z + ex)).minD; /* e.yyx */
  float d3 = sdScene(c, V3(p.x + ey, p.y + ex, p.z + ey)).minD; /* e.yxy */
  float d4 = sdScene(c, V3(p.x + ex, p.y + ex, p.z + ex)).minD; /* e.xxx */
  /* ((e.xyy*d1 + e.yyx*d2) + e.yxy*d3) + e.xxx*d4, accumulations fused */
  v3 n;
  n.x = rm_fma(ex, d4, rm_fma(ey, d3, rm_fma(ey, d2, ex * d1)));
  n.y = rm_fma(ex, d4, rm_fma(ex, d3, rm_fma(ey, d2, ey * d1)));
  n.z = rm_fma(ex, d4, rm_fma(ey, d3, rm_fma(ex, d2, ey * d1)));
  return normalize3(n);
}

/* frag:1453-1484 */
static RayMarchRes raymarch(Ctx *c, v3 ro, v3 rd, float end, float side) {
  float rayDepth = 0.0f;
  SceneMin closest;
  closest.minD = 1000000.0f; closest.minObjIdx = -1; closest.trap = V4(0, 0, 0, 0);
  RMO_TRACE_MARCH(c, 0, ro, rd, &end);
  for (int i = 0; i < c->s.maxSteps; i++) {
    v3 p = v3_madd(rd, rayDepth, ro);
    closest = sdScene(c, p);
    if (rm_abs(closest.minD) < SURFACE_DIST || rayDepth > end) break;
    rayDepth = rm_fma(closest.minD, side, rayDepth);
  }
  RayMarchRes res;
  if (rm_abs(closest.minD) < SURFACE_DIST) {
    res.intersectObj = closest.minObjIdx;
    res.d = rayDepth - closest.minD; /* frag:1477 */
    res.trap = closest.trap;
  } else {
    res.intersectObj = -1;
    res.d = rayDepth; /* UB2 */
    res.trap = V4(0, 0, 0, 0);
  }
  return res;
}

/* ---------------------------------------------------------------- Perlin bump (frag:1587-1691) */
static inline float permute1(float x) { return rm_mod(rm_fma(x, 34.0f, 1.0f) * x, 289.0f); } /* frag:1602 */
static inline float taylorInvSqrt1(float r) { return rm_fma(-0.85373472095314f, r, 1.79284291400159f); } /* frag:1606 */
static inline float fade1(float t) { return ((t * t) * t) * rm_fma(t, rm_fma(t, 6.0f, -15.0f), 10.0f); } /* frag:1587 */

/* Lattice gradient of one corner (frag:1626-1632 / 1634-1640), always in binary32 (rmo_f32 is `float` here; the binary64
 * arbiter build, rm_oracle_f64.c, keeps it binary32 too).  In real arithmetic gz = 0.5 − |gx| − |gy| is EXACTLY 0 for 7
 * of the 49 hash classes, so step(gz, 0) — which flips the gradient — is decided by the rounding of ixyz / 7 and of the
 * two fract()s: the shader's result is defined by its binary32 evaluation, not by the real-number expression.  The input
 * is an exact small integer (a lattice hash), so the decision depends on the lattice cell only, not on the sample point. */
#ifndef RMO_F32_DEFINED
typedef float rmo_f32;
#endif
static void pgrad1(rmo_f32 ixyz, rmo_f32 *ogx, rmo_f32 *ogy, rmo_f32 *ogz) {
  rmo_f32 gx = ixyz / 7.0f;
  rmo_f32 t = floorf(gx) / 7.0f;
  rmo_f32 gy = (t - floorf(t)) - 0.5f;
  gx = gx - floorf(gx);
  rmo_f32 gz = (0.5f - fabsf(gx)) - fabsf(gy);
  rmo_f32 sz = (0.0f < gz) ? 0.0f : 1.0f;                      /* step(gz, 0) */
  gx = fmaf(-sz, ((gx < 0.0f) ? 0.0f : 1.0f) - 0.5f, gx);      /* step(0, gx) */
  gy = fmaf(-sz, ((gy < 0.0f) ? 0.0f : 1.0f) - 0.5f, gy);
  *ogx = gx; *ogy = gy; *ogz = gz;
}

/* frag:1610-1676; lanes k = 0..3 are the vec4 components */
static float pnoise(v3 p) {
  v3 Pi0 = V3(rm_floor(p.x), rm_floor(p.y), rm_floor(p.z));
  v3 Pi1 = V3(Pi0.x + 1.0f, Pi0.y + 1.0f, Pi0.z + 1.0f);
  Pi0 = V3(rm_mod(Pi0.x, 256.0f), rm_mod(Pi0.y, 256.0f), rm_mod(Pi0.z, 256.0f));
  Pi1 = V3(rm_mod(Pi1.x, 256.0f), rm_mod(Pi1.y, 256.0f), rm_mod(Pi1.z, 256.0f));
  v3 Pf0 = V3(rm_fract(p.x), rm_fract(p.y), rm_fract(p.z));
  v3 Pf1 = V3(Pf0.x - 1.0f, Pf0.y - 1.0f, Pf0.z - 1.0f);
  float ix[4] = {Pi0.x, Pi1.x, Pi0.x, Pi1.x};
  float iy[4] = {Pi0.y, Pi0.y, Pi1.y, Pi1.y};
  float gx0[4], gy0[4], gz0[4], gx1[4], gy1[4], gz1[4];
  for (int k = 0; k < 4; k++) {
    float ixy = permute1(permute1(ix[k]) + iy[k]);
    float ixy0 = permute1(ixy + Pi0.z);
    float ixy1 = permute1(ixy + Pi1.z);
    rmo_f32 gx, gy, gz;
    pgrad1((rmo_f32)ixy0, &gx, &gy, &gz); /* frag:1626-1632 */
    gx0[k] = gx; gy0[k] = gy; gz0[k] = gz;
    pgrad1((rmo_f32)ixy1, &gx, &gy, &gz); /* frag:1634-1640 */
    gx1[k] = gx; gy1[k] = gy; gz1[k] = gz;
  }
  /* lane order: x=000/001, y=100/101, z=010/011, w=110/111 (frag:1642-1649) */
  v3 g000 = V3(gx0[0], gy0[0], gz0[0]), g100 = V3(gx0[1], gy0[1], gz0[1]);
  v3 g010 = V3(gx0[2], gy0[2], gz0[2]), g110 = V3(gx0[3], gy0[3], gz0[3]);
  v3 g001 = V3(gx1[0], gy1[0], gz1[0]), g101 = V3(gx1[1], gy1[1], gz1[1]);
  v3 g011 = V3(gx1[2], gy1[2], gz1[2]), g111 = V3(gx1[3], gy1[3], gz1[3]);
  /* frag:1651-1660 */
  g000 = v3_scale(g000, taylorInvSqrt1(dot3(g000, g000)));
  g010 = v3_scale(g010, taylorInvSqrt1(dot3(g010, g010)));
  g100 = v3_scale(g100, taylorInvSqrt1(dot3(g100, g100)));
  g110 = v3_scale(g110, taylorInvSqrt1(dot3(g110, g110)));
  g001 = v3_scale(g001, taylorInvSqrt1(dot3(g001, g001)));
  g011 = v3_scale(g011, taylorInvSqrt1(dot3(g011, g011)));
  g101 = v3_scale(g101, taylorInvSqrt1(dot3(g101, g101)));
  g111 = v3_scale(g111, taylorInvSqrt1(dot3(g111, g111)));
  /* frag:1662-1669 */
  float n000 = dot3(g000, Pf0);
  float n100 = dot3(g100, V3(Pf1.x, Pf0.y, Pf0.z));
  float n010 = dot3(g010, V3(Pf0.x, Pf1.y, Pf0.z));
  float n110 = dot3(g110, V3(Pf1.x, Pf1.y, Pf0.z));
  float n001 = dot3(g001, V3(Pf0.x, Pf0.y, Pf1.z));
  float n101 = dot3(g101, V3(Pf1.x, Pf0.y, Pf1.z));
  float n011 = dot3(g011, V3(Pf0.x, Pf1.y, Pf1.z));
  float n111 = dot3(g111, Pf1);
  /* frag:1671-1675 */
  v3 f = V3(fade1(Pf0.x), fade1(Pf0.y), fade1(Pf0.z));
  float nzx = rm_mix(n000, n001, f.z), nzy = rm_mix(n100, n101, f.z);
  float nzz = rm_mix(n010, n011, f.z), nzw = rm_mix(n110, n111, f.z);
  float nyzx = rm_mix(nzx, nzz, f.y), nyzy = rm_mix(nzy, nzw, f.y);
  return 2.2f * rm_mix(nyzx, nyzy, f.x);
}

/* frag:1679-1691 (scale = BUMP_SCALE 10, intensity = BUMP_INTENSITY 2, frag:128-129) */
static v3 bumpNormal(v3 normal, v3 pos, float scale, float intensity) {
  v3 ps = v3_scale(pos, scale);
  float nv = pnoise(ps);
  v3 grad = V3(pnoise(V3(ps.x + 0.1f, ps.y + 0.0f, ps.z + 0.0f)) - nv,
               pnoise(V3(ps.x + 0.0f, ps.y + 0.1f, ps.z + 0.0f)) - nv,
               pnoise(V3(ps.x + 0.0f, ps.y + 0.0f, ps.z + 0.1f)) - nv);
  return normalize3(v3_madd(grad, intensity, normal));
}

/* ---------------------------------------------------------------- shading (frag:439-461, 1703-1933) */
/* frag:1703-1725 */
static RayMarchRes softshadow(Ctx *c, v3 ro, v3 rd, float mint, float maxt, float k) {
  float res = 1.0f;
  float rayDepth = mint;
  SceneMin closest;
  closest.minD = 1000000.0f; closest.minObjIdx = -1; closest.trap = V4(0, 0, 0, 0);
  RMO_TRACE_MARCH(c, 1, ro, rd, &maxt);
  for (int i = 0; i < c->s.maxSteps; i++) {
    closest = sdScene(c, v3_madd(rd, rayDepth, ro));
    if (rm_abs(closest.minD) < SURFACE_DIST || rayDepth > maxt) break;
    res = rm_min(res, rm_divr(k * closest.minD, rayDepth));
    rayDepth = rayDepth + rm_abs(closest.minD);
  }
  RayMarchRes r;
  r.trap = V4(0, 0, 0, 0);
  r.d = res; /* frag:1718; UB1 on the miss branch */
  r.intersectObj = (rm_abs(closest.minD) < SURFACE_DIST) ? closest.minObjIdx : -1;
  return r;
}

/* frag:1729-1740 */
static float calcAO(Ctx *c, v3 pos, v3 nor) {
  float occ = 0.0f, sca = 1.0f;
  for (int i = 0; i < 5; i++) {
    float h = 0.01f + ((0.12f * (float)i) / 4.0f);
    float d = sdScene(c, v3_madd(nor, h, pos)).minD;
    occ = rm_fma(h - d, sca, occ);
    sca = sca * 0.95f;
    if (occ > 0.35f) break;
  }
  return rm_clamp(rm_fma(-3.0f, occ, 1.0f), 0.0f, 1.0f) * rm_fma(0.5f, nor.y, 0.5f);
}

/* frag:445-447 */
static float attenuationFactor(float d, const float *func) {
  return rm_min(1.0f / rm_fma(d * d, func[2], rm_fma(d, func[1], func[0])), 1.0f);
}
/* frag:439-442, 450-461 */
static float angularFalloff(const RmLight *li, v3 L) {
  v3 nd = normalize3(V3(li->dir[0], li->dir[1], li->dir[2]));
  float cosalpha = dot3(v3_neg(nd), L);
  float inner = li->angle - li->penumbra;
  if (cosalpha <= rm_cos(li->angle)) return 0.0f;
  if (cosalpha > rm_cos(inner)) return 1.0f;
  float t = (rm_acos(cosalpha) - inner) / (li->angle - inner);
  return 1.0f - rm_fma(-2.0f, rm_pow(t, 3.0f), 3.0f * rm_pow(t, 2.0f));
}


/* ---------------------------------------------------------------- textured diffuse (frag:1299-1398, 1746-1781) */
#define TEXTURE_EPS 0.005f /* frag:37 */
#define GL_PI 3.14159265f  /* frag:41 */
static v2 uvMapCube(v3 p, float rU, float rV) { /* frag:1299-1333 */
  float u, v;
  float ax = rm_abs(p.x), ay = rm_abs(p.y), az = rm_abs(p.z);
  float m = rm_max(rm_max(ax, ay), az);
  if (m == ax) {
    if (p.x < 0.0f) { u = p.z + 0.5f; v = p.y + 0.5f; } else { u = -p.z + 0.5f; v = p.y + 0.5f; }
  } else if (m == ay) {
    if (p.y < 0.0f) { u = p.x + 0.5f; v = p.z + 0.5f; } else { u = p.x + 0.5f; v = -p.z + 0.5f; }
  } else {
    if (p.z < 0.0f) { u = -p.x + 0.5f; v = p.y + 0.5f; } else { u = p.x + 0.5f; v = p.y + 0.5f; }
  }
  v2 r = {u * rU, v * rV};
  return r;
}
static float uFromTheta(float theta) { /* frag:1346-1350 */
  if (theta < 0.0f) return -theta / (2.0f * GL_PI);
  return 1.0f - (theta / (2.0f * GL_PI));
}
static v2 uvMapCone(v3 p, float rU, float rV) { /* frag:1336-1354 */
  float u, v;
  if (rm_abs(p.y + 0.5f) < TEXTURE_EPS) { u = p.x + 0.5f; v = p.z + 0.5f; }
  else { u = uFromTheta(rm_atan2(p.z, p.x)); v = p.y + 0.5f; }
  v2 r = {u * rU, v * rV};
  return r;
}
static v2 uvMapCylinder(v3 p, float rU, float rV) { /* frag:1357-1378 */
  float u, v;
  if (rm_abs(p.y - 0.5f) < TEXTURE_EPS) { u = p.x + 0.5f; v = -p.z + 0.5f; }
  else if (rm_abs(p.y + 0.5f) < TEXTURE_EPS) { u = p.x + 0.5f; v = p.z + 0.5f; }
  else { u = uFromTheta(rm_atan2(p.z, p.x)); v = p.y + 0.5f; }
  v2 r = {u * rU, v * rV};
  return r;
}
static v2 uvMapSphere(v3 p, float rU, float rV) { /* frag:1381-1398 */
  float u = uFromTheta(rm_atan2(p.z, p.x));
  float phi = rm_asin(p.y / 0.5f);
  float v = phi / GL_PI + 0.5f;
  if (v == 0.0f || v == 1.0f) u = 0.5f;
  v2 r = {u * rU, v * rV};
  return r;
}
/* texture(sampler2D, uv) for an RGBA8 texture with GL_LINEAR filtering and GL_REPEAT wrap (GL 3.3 §3.8.11),
 * weights in binary32: mix(mix(t00,t10,a), mix(t01,t11,a), b), texel = byte / 255. */
static inline int wrapi(float f, int n) {
  if (!(rm_abs(f) < 1.0e9f)) f = 0.0f;
  int i = (int)f % n;
  return i < 0 ? i + n : i;
}
typedef struct { float c[3][3]; } m3;   /* c[col][row] */
static inline v3 m3_mul_v3(const m3 *M, v3 v) {
  return V3(rm_fma(M->c[2][0], v.z, rm_fma(M->c[1][0], v.y, M->c[0][0] * v.x)),
            rm_fma(M->c[2][1], v.z, rm_fma(M->c[1][1], v.y, M->c[0][1] * v.x)),
            rm_fma(M->c[2][2], v.z, rm_fma(M->c[1][2], v.y, M->c[0][2] * v.x)));
}
static inline m3 m3_mul_m3(const m3 *A, const m3 *B) {
  m3 R;
  for (int c = 0; c < 3; c++) {
    v3 col = m3_mul_v3(A, V3(B->c[c][0], B->c[c][1], B->c[c][2]));
    R.c[c][0] = col.x; R.c[c][1] = col.y; R.c[c][2] = col.z;
  }
  return R;
}
static inline m3 m3_scale(const m3 *A, float f) {
  m3 R;
  for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) R.c[c][r] = f * A->c[c][r];
  return R;
}
static inline v3 cross3(v3 a, v3 b) {
  return V3(rm_fma(a.y, b.z, -(a.z * b.y)), rm_fma(a.z, b.x, -(a.x * b.z)), rm_fma(a.x, b.y, -(a.y * b.x)));
}
/* One RGBA8 GL_LINEAR fetch.  clampEdge = 0: GL_REPEAT, 1: GL_CLAMP_TO_EDGE. */
static v4 sampleRGBA8(const RmTexture *t, float su, float sv, int clampEdge) {
  const int W = t->width, H = t->height;
  float u = rm_fma(su, (float)W, -0.5f), v = rm_fma(sv, (float)H, -0.5f);
  float fu = rm_floor(u), fv = rm_floor(v);
  float a = u - fu, b = v - fv;
  int i0, i1, j0, j1;
  if (clampEdge) {
    if (!(rm_abs(fu) < 1.0e9f)) fu = 0.0f;
    if (!(rm_abs(fv) < 1.0e9f)) fv = 0.0f;
    int iu = (int)fu, iv = (int)fv;
    i0 = iu < 0 ? 0 : (iu > W - 1 ? W - 1 : iu); i1 = iu + 1 < 0 ? 0 : (iu + 1 > W - 1 ? W - 1 : iu + 1);
    j0 = iv < 0 ? 0 : (iv > H - 1 ? H - 1 : iv); j1 = iv + 1 < 0 ? 0 : (iv + 1 > H - 1 ? H - 1 : iv + 1);
  } else {
    i0 = wrapi(fu, W); j0 = wrapi(fv, H);
    i1 = (i0 + 1 == W) ? 0 : i0 + 1; j1 = (j0 + 1 == H) ? 0 : j0 + 1;
  }
  const uint8_t *p00 = t->pixels + ((size_t)j0 * W + i0) * 4, *p10 = t->pixels + ((size_t)j0 * W + i1) * 4;
  const uint8_t *p01 = t->pixels + ((size_t)j1 * W + i0) * 4, *p11 = t->pixels + ((size_t)j1 * W + i1) * 4;
  float c[4];
  for (int k = 0; k < 4; k++) {
    float lo = rm_mix((float)p00[k] / 255.0f, (float)p10[k] / 255.0f, a);
    float hi = rm_mix((float)p01[k] / 255.0f, (float)p11[k] / 255.0f, a);
    c[k] = rm_mix(lo, hi, b);
  }
  return V4(c[0], c[1], c[2], c[3]);
}
static v3 sampleTexture(const RmTexture *t, v2 uv) {
  v4 c = sampleRGBA8(t, uv.x, uv.y, 0);
  return V3(c.x, c.y, c.z);
}
/* texture(samplerCube, r): major-axis face selection and (s,t) of GL 3.3 §3.8.10 table 3.19; equal magnitudes
 * resolve x before y before z (the spec leaves ties to the implementation).  Filtering stays inside the face
 * (GL_CLAMP_TO_EDGE, seamless filtering is off by default in a GL 4.1 core context and never enabled). */
static v3 sampleCube(const RmTexture *faces, v3 r) {
  float ax = rm_abs(r.x), ay = rm_abs(r.y), az = rm_abs(r.z), sc, tc, ma;
  int face;
  if (ax >= ay && ax >= az) { ma = ax; if (r.x >= 0.0f) { face = 0; sc = -r.z; tc = -r.y; } else { face = 1; sc = r.z; tc = -r.y; } }
  else if (ay >= az)        { ma = ay; if (r.y >= 0.0f) { face = 2; sc = r.x; tc = r.z; } else { face = 3; sc = r.x; tc = -r.z; } }
  else                      { ma = az; if (r.z >= 0.0f) { face = 4; sc = r.x; tc = -r.y; } else { face = 5; sc = -r.x; tc = -r.y; } }
  const float ima = 1.0f / ma; /* rm_divr */
  v4 c = sampleRGBA8(&faces[face], rm_fma(sc * ima, 0.5f, 0.5f), rm_fma(tc * ima, 0.5f, 0.5f), 1);
  return V3(c.x, c.y, c.z);
}
/* frag:1746-1781 */
static v3 getDiffuse(const Ctx *c, const RmObject *obj, v3 p) {
  const float kd = c->g.kd;
  if (obj->texLoc == -1) return V3(kd * obj->cDiffuse[0], kd * obj->cDiffuse[1], kd * obj->cDiffuse[2]);
  v3 po = xform_point(obj->invModel, p);
  v2 uv;
  if (obj->type == RM_CUBE) uv = uvMapCube(po, obj->repeatU, obj->repeatV);
  else if (obj->type == RM_CONE) uv = uvMapCone(po, obj->repeatU, obj->repeatV);
  else if (obj->type == RM_CYLINDER) uv = uvMapCylinder(po, obj->repeatU, obj->repeatV);
  else uv = uvMapSphere(po, obj->repeatU, obj->repeatV);
  v3 t = sampleTexture(&c->tex[obj->texLoc], uv);
  float k = (1.0f - obj->blend) * kd;
  return V3(rm_fma(obj->blend, t.x, k * obj->cDiffuse[0]), rm_fma(obj->blend, t.y, k * obj->cDiffuse[1]),
            rm_fma(obj->blend, t.z, k * obj->cDiffuse[2]));
}

/* ---- area lights: linearly transformed cosines (frag:349-424, 1794-1822) ----
 * LTC1/LTC2 are RM_LTC_SIZE² RGBA8 tables with GL_CLAMP_TO_EDGE.  The reference sets MIN = NEAREST, MAG = LINEAR
 * (realtimerender.cpp:910-913); which one applies depends on screen-space derivatives taken inside divergent
 * control flow (undefined in GLSL) — the smooth uv of a 64² table magnifies, so GL_LINEAR is used throughout. */
#define LUT_SCALE ((64.0f - 1.0f) / 64.0f) /* frag:48 */
#define LUT_BIAS (0.5f / 64.0f)            /* frag:49 */
static v4 sampleLTC(const uint8_t *table, float u, float v) {
  RmTexture t; t.pixels = table; t.width = RM_LTC_SIZE; t.height = RM_LTC_SIZE;
  return sampleRGBA8(&t, u, v, 1);
}
static v3 IntegrateEdgeVec(v3 v1, v3 v2) { /* frag:349-361 */
  float x = dot3(v1, v2), y = rm_abs(x);
  float a = rm_fma(rm_fma(0.0145206f, y, 0.4965155f), y, 0.8543985f);
  float b = rm_fma(4.1616724f + y, y, 3.4175940f);
  float v = a / b;
  float ts = (x > 0.0f) ? v : rm_fma(0.5f, 1.0f / rm_sqrt(rm_max(rm_fma(-x, x, 1.0f), 1e-7f)), -v);
  return v3_scale(cross3(v1, v2), ts);
}
static float LTC_Evaluate(const Ctx *c, v3 N, v3 V, v3 P, const m3 *MinvIn, const RmLight *li) { /* frag:368-424 */
  v3 T1 = normalize3(v3_madd(N, -dot3(V, N), V));
  v3 T2 = cross3(N, T1);
  m3 B = {{{T1.x, T2.x, N.x}, {T1.y, T2.y, N.y}, {T1.z, T2.z, N.z}}}; /* transpose(mat3(T1,T2,N)) */
  m3 Minv = m3_mul_m3(MinvIn, &B);
  v3 pts[4], L[4];
  for (int k = 0; k < 4; k++) {
    pts[k] = V3(li->points[k][0], li->points[k][1], li->points[k][2]);
    L[k] = m3_mul_v3(&Minv, v3_sub(pts[k], P));
  }
  v3 dir = v3_sub(pts[0], P);
  v3 lightNormal = cross3(v3_sub(pts[1], pts[0]), v3_sub(pts[3], pts[0]));
  int behind = dot3(dir, lightNormal) < 0.0f;
  for (int k = 0; k < 4; k++) L[k] = normalize3(L[k]);
  v3 vsum = IntegrateEdgeVec(L[0], L[1]);
  vsum = v3_add(vsum, IntegrateEdgeVec(L[1], L[2]));
  vsum = v3_add(vsum, IntegrateEdgeVec(L[2], L[3]));
  vsum = v3_add(vsum, IntegrateEdgeVec(L[3], L[0]));
  float len = len3(vsum);
  float z = (len == 0.0f) ? 0.0f : vsum.z / len; /* UB11: 0/0 when the four edge terms cancel exactly */
  if (behind) z = -z;
  float scale = sampleLTC(c->res->ltc2, rm_fma(rm_fma(z, 0.5f, 0.5f), LUT_SCALE, LUT_BIAS), rm_fma(len, LUT_SCALE, LUT_BIAS)).w;
  float sum = len * scale;
  if (!behind && !li->twoSided) sum = 0.0f;
  return sum;
}
static v3 getAreaLight(const Ctx *c, v3 N, v3 V, v3 P, const RmLight *li, const RmObject *obj) { /* frag:1795-1822 */
  float dotNV = rm_clamp(dot3(N, V), 0.0f, 1.0f);
  float u = rm_fma(0.0f, LUT_SCALE, LUT_BIAS), v = rm_fma(rm_sqrt(1.0f - dotNV), LUT_SCALE, LUT_BIAS);
  v4 t1 = sampleLTC(c->res->ltc1, u, v), t2 = sampleLTC(c->res->ltc2, u, v);
  const m3 Minv = {{{t1.x, 0.0f, t1.y}, {0.0f, 1.0f, 0.0f}, {t1.z, 0.0f, t1.w}}};
  const m3 I = {{{1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f}}};
  float diffuse = LTC_Evaluate(c, N, V, P, &I, li);
  float specular = LTC_Evaluate(c, N, V, P, &Minv, li);
  v3 dif = getDiffuse(c, obj, P);
  float col[3];
  const float difk[3] = {dif.x, dif.y, dif.z};
  for (int k = 0; k < 3; k++) {
    float cS = obj->cSpecular[k];
    float sp = specular * rm_fma(li->intensity - cS, t2.y, cS * t2.x);
    col[k] = (li->color[k] * 1.0f) * rm_fma(difk[k], diffuse, sp);
  }
  return V3(col[0], col[1], col[2]);
}

/* frag:1842-1933 (texLoc == -1 path of getDiffuse, frag:1749-1752; getSpecular frag:1787-1792) */
static v3 getPhong(Ctx *c, v3 N, int intersectObj, v3 p, v3 rd, float far) {
  const RmObject *obj = &c->objs[intersectObj];
  const float ka = c->g.ka, ks = c->g.ks;
  float ao = 1.0f;
  if (c->s.enableAmbientOcclusion) ao = calcAO(c, p, N);
  v3 total = V3((obj->cAmbient[0] * ka) * ao, (obj->cAmbient[1] * ka) * ao, (obj->cAmbient[2] * ka) * ao);
  for (int i = 0; i < c->numLights; i++) {
    const RmLight *li = &c->lights[i];
    float fAtt = 1.0f, aFall = 1.0f;
    v3 lpos = V3(li->pos[0], li->pos[1], li->pos[2]);
    float d = len3(v3_sub(p, lpos));
    v3 L = V3(0.0f, 0.0f, 0.0f); float maxT = 0.0f;
    if (li->type == RM_LIGHT_POINT) {
      L = normalize3(v3_sub(lpos, p));
      fAtt = attenuationFactor(d, li->func);
      maxT = len3(v3_sub(lpos, p));
    } else if (li->type == RM_LIGHT_DIRECTIONAL) {
      L = normalize3(V3(-li->dir[0], -li->dir[1], -li->dir[2]));
      maxT = far;
    } else if (li->type == RM_LIGHT_SPOT) {
      L = normalize3(v3_sub(lpos, p));
      fAtt = attenuationFactor(d, li->func);
      maxT = len3(v3_sub(lpos, p));
      aFall = angularFalloff(li, L);
    }
    v3 V = normalize3(v3_neg(rd));
    if (li->type == RM_LIGHT_AREA) { /* frag:1884-1905, AREA_LIGHT_SAMPLES = 1: the "random" uv is rd.xy */
      v3 p1 = V3(li->points[0][0], li->points[0][1], li->points[0][2]);
      v3 side1 = v3_sub(V3(li->points[1][0], li->points[1][1], li->points[1][2]), p1);
      v3 side2 = v3_sub(V3(li->points[3][0], li->points[3][1], li->points[3][2]), p1);
      v3 randomP = v3_madd(side2, rd.y + 0.0f, v3_madd(side1, rd.x + 0.0f, p1));
      L = normalize3(v3_sub(randomP, p));
      if (dot3(N, L) <= 0.005f) continue;
      maxT = len3(v3_sub(randomP, p));
      v3 so = V3(rm_fma(N.x * SURFACE_DIST, 5.0f, p.x), rm_fma(N.y * SURFACE_DIST, 5.0f, p.y),
                 rm_fma(N.z * SURFACE_DIST, 5.0f, p.z));
      RayMarchRes sh = softshadow(c, so, L, 0.0f, maxT, 8.0f);
      if (sh.intersectObj != -1 && c->objs[sh.intersectObj].lightIdx != i) continue;
      total = v3_add(total, getAreaLight(c, N, V, p, li, obj));
      continue;
    }
    /* frag:1908: origin p + N*SURFACE_DIST*5 */
    v3 so = V3(rm_fma(N.x * SURFACE_DIST, 5.0f, p.x), rm_fma(N.y * SURFACE_DIST, 5.0f, p.y),
               rm_fma(N.z * SURFACE_DIST, 5.0f, p.z));
    if (RMO_TRACE_SKIP_SHADOW(c, N, L)) continue;
    RayMarchRes sh = softshadow(c, so, L, 0.0f, maxT, 8.0f);
    if (sh.intersectObj != -1) continue;
    float NdotL = dot3(N, L);
    if (NdotL <= 0.005f) continue;
    NdotL = rm_clamp(NdotL, 0.0f, 1.0f);
    v3 lc = V3(li->color[0], li->color[1], li->color[2]);
    v3 dif = getDiffuse(c, obj, p);
    v3 cur = V3((dif.x * NdotL) * lc.x, (dif.y * NdotL) * lc.y, (dif.z * NdotL) * lc.z);
    v3 R = reflect3(v3_neg(L), N);
    float RdotV = rm_clamp(dot3(R, V), 0.0f, 1.0f);
    float sp = (obj->shininess == 0.0f) ? (ks * RdotV) : (ks * rm_pow(RdotV, obj->shininess));
    cur = V3(rm_fma(sp * obj->cSpecular[0], lc.x, cur.x), rm_fma(sp * obj->cSpecular[1], lc.y, cur.y),
             rm_fma(sp * obj->cSpecular[2], lc.z, cur.z));
    cur = v3_scale(cur, fAtt * aFall);
    if (c->s.enableSoftShadow) cur = v3_scale(cur, sh.d);
    total = v3_add(total, cur);
  }
  return total;
}


/* ---------------------------------------------------------------- procedural layers (frag:464-746, 1519-1584, 1950-2158)
 * Compile-time #defines TERRAIN / CLOUD / SKY_BACKGROUND of the shader (frag:4-15) are runtime feature bits.
 * GLSL evaluates `f*m2*x` left to right: (f*m2)*x — the scaled constant matrices are formed first. */
static const m3 kM3 = {{{0.00f, 0.80f, 0.60f}, {-0.80f, 0.36f, -0.48f}, {-0.60f, -0.48f, 0.64f}}};   /* frag:118-120 */
static const m3 kM3i = {{{0.00f, -0.80f, -0.60f}, {0.80f, 0.36f, -0.48f}, {0.60f, -0.48f, 0.64f}}};  /* frag:121-123 */

static inline float hash1f(float n) { return rm_fract((n * 17.0f) * rm_fract(n * 0.3183099f)); }  /* frag:467-469 */
static inline float hash1v2(float px, float py) {                                                     /* frag:472-475 */
  px = 50.0f * rm_fract(px * 0.3183099f);
  py = 50.0f * rm_fract(py * 0.3183099f);
  return rm_fract((px * py) * (px + py));
}
static inline float quintic(float w) { return ((w * w) * w) * rm_fma(w, rm_fma(w, 6.0f, -15.0f), 10.0f); }
/* frag:493-502 */
static float noiseT(float x, float y) {
  float px = rm_floor(x), py = rm_floor(y);
  float ux = quintic(rm_fract(x)), uy = quintic(rm_fract(y));
  float a = hash1v2(px + 0.0f, py + 0.0f), b = hash1v2(px + 1.0f, py + 0.0f);
  float c = hash1v2(px + 0.0f, py + 1.0f), d = hash1v2(px + 1.0f, py + 1.0f);
  float t = rm_fma(b - a, ux, a);
  t = rm_fma(c - a, uy, t);
  t = rm_fma((((a - b) - c) + d) * ux, uy, t);
  return rm_fma(2.0f, t, -1.0f);
}
/* frag:536-567: value noise (x) and analytic gradient (yzw) */
static v4 noised3(v3 x) {
  v3 p = V3(rm_floor(x.x), rm_floor(x.y), rm_floor(x.z));
  v3 w = V3(rm_fract(x.x), rm_fract(x.y), rm_fract(x.z));
  v3 u = V3(quintic(w.x), quintic(w.y), quintic(w.z));
  v3 du = V3(((30.0f * w.x) * w.x) * rm_fma(w.x, w.x - 2.0f, 1.0f), ((30.0f * w.y) * w.y) * rm_fma(w.y, w.y - 2.0f, 1.0f),
             ((30.0f * w.z) * w.z) * rm_fma(w.z, w.z - 2.0f, 1.0f));
  float n = rm_fma(157.0f, p.z, rm_fma(317.0f, p.y, p.x));
  float a = hash1f(n + 0.0f), b = hash1f(n + 1.0f), c = hash1f(n + 317.0f), d = hash1f(n + 318.0f);
  float e = hash1f(n + 157.0f), f = hash1f(n + 158.0f), g = hash1f(n + 474.0f), h = hash1f(n + 475.0f);
  float k0 = a, k1 = b - a, k2 = c - a, k3 = e - a;
  float k4 = ((a - b) - c) + d, k5 = ((a - c) - e) + g, k6 = ((a - b) - e) + f;
  float k7 = ((((((-a + b) + c) - d) + e) - f) - g) + h;
  float v = rm_fma(k1, u.x, k0);
  v = rm_fma(k2, u.y, v);
  v = rm_fma(k3, u.z, v);
  v = rm_fma(k4 * u.x, u.y, v);
  v = rm_fma(k5 * u.y, u.z, v);
  v = rm_fma(k6 * u.z, u.x, v);
  v = rm_fma((k7 * u.x) * u.y, u.z, v);
  float dx = rm_fma(k7 * u.y, u.z, rm_fma(k6, u.z, rm_fma(k4, u.y, k1)));
  float dy = rm_fma(k7 * u.z, u.x, rm_fma(k4, u.x, rm_fma(k5, u.z, k2)));
  float dz = rm_fma(k7 * u.x, u.y, rm_fma(k5, u.y, rm_fma(k6, u.x, k3)));
  return V4(rm_fma(2.0f, v, -1.0f), (2.0f * du.x) * dx, (2.0f * du.y) * dy, (2.0f * du.z) * dz);
}
/* work counters of the procedural layers (per thread; rmo_render sums them): fbm_9 / fbmd_8 evaluations */
static __thread uint64_t t_nFbm9, t_nFbmd8;
/* frag:630-644: f = 1.9, m2 = (0.8,0.6 | -0.6,0.8), gain .55 */
static float fbm_9(float x, float y) {
  t_nFbm9++;
  const float m00 = 1.9f * 0.80f, m01 = 1.9f * 0.60f, m10 = 1.9f * -0.60f, m11 = 1.9f * 0.80f;  /* (f*m2), c[col][row] */
  float a = 0.0f, b = 0.5f;
  for (int i = 0; i < 9; i++) {
    float n = noiseT(x, y);
    a = rm_fma(b, n, a);
    b = b * 0.55f;
    float nx = rm_fma(m10, y, m00 * x), ny = rm_fma(m11, y, m01 * x);
    x = nx; y = ny;
  }
  return a;
}
/* frag:647-667 */
static v4 fbmd_8(v3 x) {
  t_nFbmd8++;
  const m3 fm3 = m3_scale(&kM3, 2.0f), fm3i = m3_scale(&kM3i, 2.0f);
  float a = 0.0f, b = 0.5f;
  v3 d = V3(0.0f, 0.0f, 0.0f);
  m3 m = {{{1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f}}};
  for (int i = 0; i < 8; i++) {
    v4 n = noised3(x);
    a = rm_fma(b, n.x, a);
    if (i < 4) {
      m3 bm = m3_scale(&m, b);
      d = v3_add(d, m3_mul_v3(&bm, V3(n.y, n.z, n.w)));
    }
    b = b * 0.65f;
    x = m3_mul_v3(&fm3, x);
    m = m3_mul_m3(&fm3i, &m);
  }
  return V4(a, d.x, d.y, d.z);
}
/* frag:737-746 */
static v2 sdTerrain(float px, float pz) {
  float e = fbm_9(rm_divr(px, 2000.0f) + 1.0f, rm_divr(pz, 2000.0f) + -2.0f);
  float a = 1.0f - rm_smoothstep(0.12f, 0.13f, rm_abs(e + 0.12f));
  e = rm_fma(600.0f, e, 600.0f);
  e = rm_fma(90.0f, rm_smoothstep(552.0f, 594.0f, e), e);
  v2 r = {e, a};
  return r;
}
/* frag:1529-1584, timeOfDay = 0.1 */
static v3 getSunDir(void) {
  float ang = rm_mix(0.0f, 3.14f, 0.1f);
  return normalize3(V3(rm_cos(ang), rm_sin(ang), -0.577f));
}
static v3 getSkyColor(void) {
  v3 c = mix3(V3(1.0f, 0.5f, 0.2f), V3(0.8f, 0.9f, 1.1f), rm_smoothstep(0.0f, 0.2f, 0.1f));
  return mix3(c, V3(1.0f, 0.8f, 0.5f), rm_smoothstep(0.8f, 1.0f, 0.1f));
}
static v3 getSunColor(void) {
  v3 c = mix3(V3(1.0f, 0.5f, 0.2f), V3(1.0f, 1.0f, 0.8f), rm_smoothstep(0.0f, 0.2f, 0.1f));
  return mix3(c, V3(1.0f, 0.8f, 0.5f), rm_smoothstep(0.8f, 1.0f, 0.1f));
}
static v3 getSky(v3 rd) {
  v3 col = v3_scale(getSkyColor(), rm_fma(0.4f, rd.y, 0.6f));
  float s = rm_pow(rm_clamp(dot3(rd, getSunDir()), 0.0f, 1.0f), 32.0f);
  return v3_madd(getSunColor(), s, col);
}
/* frag:1519-1523 */
static v3 fog(v3 col, float t) {
  float k = (-t) * 0.00025f;
  v3 ext = V3(rm_exp2(k * 1.0f), rm_exp2(k * 1.5f), rm_exp2(k * 4.0f));
  return V3(rm_fma(1.0f - ext.x, 0.55f, col.x * ext.x), rm_fma(1.0f - ext.y, 0.55f, col.y * ext.y),
            rm_fma(1.0f - ext.z, 0.58f, col.z * ext.z));
}
/* frag:1950-1959 */
static v4 cloudsFbm(const Ctx *c, v3 pos) {
  const float it = c->g.iTime;
  v3 q = V3(rm_fma(0.07f, it, rm_fma(pos.x, 0.0015f, 2.0f)), rm_fma(0.07f, 0.5f * it, rm_fma(pos.y, 0.0015f, 1.1f)),
            rm_fma(0.07f, -0.15f * it, rm_fma(pos.z, 0.0015f, 1.0f)));
  return fbmd_8(q);
}
static float cloudsShadowFlat(const Ctx *c, v3 ro, v3 rd) {
  float t = (900.0f - ro.y) / rd.y;
  if (t < 0.0f) return 1.0f;
  return cloudsFbm(c, v3_madd(rd, t, ro)).x;
}
/* frag:1961-1974.  UB10: the reference leaves `nnd` unset when d > 0 although cloudMarch reads it for the sun
 * sample (frag:1994-1995); iq's original sets nnd = -d before the early return, which is what is done here. */
static v4 cloudsMap(const Ctx *c, v3 pos, float *nnd) {
  float d = rm_abs(pos.y - 900.0f) - 4.0f;
  float gy = rm_sign(pos.y - 900.0f);
  v4 n = cloudsFbm(c, pos);
  d = rm_fma(400.0f * n.x, rm_fma(0.3f, gy, 0.7f), d);
  *nnd = -d;
  if (d > 0.0f) return V4(-d, 0.0f, 0.0f, 0.0f);
  d = rm_min(rm_divr(-d, 100.0f), 0.25f);
  return V4(d, 0.0f, gy, 0.0f);
}
/* frag:1976-2026 */
static int cloudMarch(const Ctx *c, int steps, v3 ro, v3 rd, float minT, float maxT, v4 *sum) {
  int hasHit = 0;
  float t = minT, lastT = -1.0f, thickness = 0.0f;
  const v3 sunColor = getSunColor(), sunDir = getSunDir();
  for (int i = 0; i < steps; i++) {
    v3 pos = v3_madd(rd, t, ro);
    float nnd;
    v4 denGra = cloudsMap(c, pos, &nnd);
    float den = denGra.x;
    float dt = rm_max(0.3f, 0.011f * t);
    if (den > 0.001f) {
      hasHit = 1;
      float kk;
      cloudsMap(c, v3_madd(sunDir, 70.0f, pos), &kk);
      float sha = 1.0f - rm_smoothstep(-200.0f, 200.0f, kk);
      sha = sha * 1.5f;
      v3 nor = normalize3(V3(denGra.y, denGra.z, denGra.w));
      float dif = rm_clamp(rm_fma(0.6f, dot3(nor, sunDir), 0.4f), 0.0f, 1.0f) * sha;
      float occ = rm_fma(0.1f, 1.0f - den, rm_fma(0.7f, rm_max(1.0f - rm_divr(kk, 200.0f), 0.0f), 0.2f));
      float up = rm_fma(0.5f, nor.y, 0.5f), dn = rm_fma(-0.5f, nor.y, 0.5f);
      v3 lin = V3(0.0f, 0.0f, 0.0f);
      lin = V3(lin.x + ((0.70f * 1.0f) * up) * occ, lin.y + ((0.80f * 1.0f) * up) * occ, lin.z + ((1.00f * 1.0f) * up) * occ);
      lin = V3(lin.x + ((0.10f * 1.0f) * dn) * occ, lin.y + ((0.40f * 1.0f) * dn) * occ, lin.z + ((0.20f * 1.0f) * dn) * occ);
      lin = V3(lin.x + rm_fma((sunColor.x * 3.0f) * dif, occ, 0.1f), lin.y + rm_fma((sunColor.y * 3.0f) * dif, occ, 0.1f),
               lin.z + rm_fma((sunColor.z * 3.0f) * dif, occ, 0.1f));
      v3 col = V3(0.8f * 0.45f, 0.8f * 0.45f, 0.8f * 0.45f);
      col = v3_mul(col, lin);
      col = fog(col, t);
      float alp = rm_clamp(((den * 0.5f) * 0.125f) * dt, 0.0f, 1.0f);
      col = v3_scale(col, alp);
      float om = 1.0f - sum->w;
      *sum = V4(rm_fma(col.x, om, sum->x), rm_fma(col.y, om, sum->y), rm_fma(col.z, om, sum->z), rm_fma(alp, om, sum->w));
      thickness = rm_fma(dt, den, thickness);
      if (lastT < 0.0f) lastT = t;
    } else {
      dt = rm_abs(den) + 0.2f;
    }
    t = t + dt;
    if (sum->w > 0.995f || t > maxT) break;
  }
  float glow = rm_pow(rm_clamp(dot3(sunDir, rd), 0.0f, 1.0f), 32.0f);
  float mx = rm_max(0.0f, rm_fma(-0.0125f, thickness, 1.0f));
  sum->x = sum->x + ((mx * sunColor.x) * 0.3f) * glow;
  sum->y = sum->y + ((mx * sunColor.y) * 0.3f) * glow;
  sum->z = sum->z + ((mx * sunColor.z) * 0.3f) * glow;
  return hasHit;
}
/* frag:2031-2057.  The blue-noise texture blob is missing from the reference checkout, so the dither sample is 0
 * (incomplete texture, realtimerender.cpp:405-408); FRAME = 1 (frag:131, 2385). */
static v3 cloudRender(const Ctx *c, v3 ro, v3 rd, v3 bgCol, int *hit, float maxT) {
  float minT = 0.0f;
  float tl = (600.0f - ro.y) / rd.y, th = (1200.0f - ro.y) / rd.y;
  if (tl > 0.0f) minT = rm_max(minT, tl);
  else { *hit = 0; return bgCol; }
  if (th > 0.0f) maxT = rm_min(maxT, th);
  v4 sum = V4(0.0f, 0.0f, 0.0f, 0.0f);
  float off = (float)(1 % 64) + 0.61803398875f;
  minT = rm_fma(0.3f, rm_fract(off + 0.0f), minT);
  *hit = cloudMarch(c, 128, ro, rd, minT, maxT, &sum);
  sum = V4(rm_clamp(sum.x, 0.0f, 1.0f), rm_clamp(sum.y, 0.0f, 1.0f), rm_clamp(sum.z, 0.0f, 1.0f), rm_clamp(sum.w, 0.0f, 1.0f));
  float om = 1.0f - sum.w;
  return V3(rm_fma(bgCol.x, om, sum.x), rm_fma(bgCol.y, om, sum.y), rm_fma(bgCol.z, om, sum.z));
}
/* frag:2060-2090 */
static float raymarchTerrain(v3 ro, v3 rd, float tmin, float tmax) {
  float tp = (700.0f - ro.y) / rd.y;
  if (tp > 0.0f) tmax = rm_min(tmax, tp);
  float dis = 0.0f, th = 0.0f, t = tmin, ot = t, odis = 0.0f;
  for (int i = 0; i < 400; i++) {
    th = 0.001f * t;
    v3 pos = v3_madd(rd, t, ro);
    v2 env = sdTerrain(pos.x, pos.z);
    dis = pos.y - env.x;
    if (dis < th) break;
    ot = t;
    odis = dis;
    t = rm_fma(dis * 0.8f, rm_fma(-0.75f, env.y, 1.0f), t);
    if (t > tmax) break;
  }
  if (t > tmax) return -1.0f;
  return ot + ((th - odis) * (t - ot)) / (dis - odis);
}
/* frag:2106-2111 */
static v3 terrainNormal(float px, float pz) {
  const float e = 0.03f;
  return normalize3(V3(sdTerrain(px - e, pz - 0.0f).x - sdTerrain(px + e, pz + 0.0f).x, 2.0f * e,
                       sdTerrain(px - 0.0f, pz - e).x - sdTerrain(px + 0.0f, pz + e).x));
}
/* frag:2113-2125 */
static float terrainShadow(v3 ro, v3 rd, float mint) {
  float res = 1.0f, t = mint;
  for (int i = 0; i < 32; i++) {
    v3 pos = v3_madd(rd, t, ro);
    v2 env = sdTerrain(pos.x, pos.z);
    float hei = pos.y - env.x;
    res = rm_min(res, rm_divr(32.0f * hei, t));
    if (res < 0.0001f || pos.y > 700.0f) break;
    t = t + rm_clamp(hei, rm_fma(t, 0.1f, 2.0f), 100.0f);
  }
  return rm_clamp(res, 0.0f, 1.0f);
}
/* frag:2128-2158: returns 1 on hit and fills col, d */
static int terrainRender(const Ctx *c, v3 ro, v3 rd, float maxT, v3 bgCol, v3 *colOut, float *dOut) {
  *colOut = bgCol; *dOut = maxT;
  float res = raymarchTerrain(ro, rd, 15.0f, maxT);
  if (!(res > 0.0f)) return 0;
  *dOut = res;
  v3 p = v3_madd(rd, res, ro);
  v3 pn = terrainNormal(p.x, p.z);
  v3 epos = V3(p.x + 0.0f, p.y + 4.8f, p.z + 0.0f);
  const v3 sunColor = getSunColor(), sunDir = getSunDir();
  float sha1 = terrainShadow(V3(p.x + 0.0f, p.y + 0.02f, p.z + 0.0f), sunDir, 0.02f);
  sha1 = sha1 * rm_smoothstep(-0.325f, -0.075f, cloudsShadowFlat(c, epos, sunDir));
  v4 fb = fbmd_8(V3(((p.x - 0.0f) * 0.15f) * 1.0f, ((p.y - 600.0f) * 0.15f) * 0.2f, ((p.z - 0.0f) * 0.15f) * 1.0f));
  float k = (0.8f * (1.0f - rm_abs(pn.y))) * 0.8f;
  v3 nor = normalize3(V3(rm_fma(k, fb.y, pn.x), rm_fma(k, fb.z, pn.y), rm_fma(k, fb.w, pn.z)));
  v3 col = V3(0.18f * 0.85f, 0.12f * 0.85f, 0.10f * 0.85f);
  col = mix3(col, V3(0.1f * 0.2f, 0.1f * 0.2f, 0.0f * 0.2f), rm_smoothstep(0.7f, 0.9f, nor.y));
  float dif = rm_clamp(dot3(nor, sunDir), 0.0f, 1.0f) * sha1;
  float bac = rm_clamp(dot3(normalize3(V3(-sunDir.x, 0.0f, -sunDir.z)), nor), 0.0f, 1.0f);
  float foc = rm_clamp(rm_divr(p.y / 2.0f - 180.0f, 130.0f), 0.0f, 1.0f);
  float dom = rm_clamp(rm_fma(0.5f, nor.y, 0.5f), 0.0f, 1.0f);
  v3 lin = mix3(V3(0.1f * 0.1f, 0.1f * 0.2f, 0.1f * 0.1f), v3_scale(sunColor, 3.0f), dom);
  lin = V3((0.2f * lin.x) * foc, (0.2f * lin.y) * foc, (0.2f * lin.z) * foc);
  lin = V3(rm_fma(8.5f * sunColor.x, dif, lin.x), rm_fma(8.5f * sunColor.y, dif, lin.y), rm_fma(8.5f * sunColor.z, dif, lin.z));
  lin = V3(rm_fma((0.27f * sunColor.x) * bac, foc, lin.x), rm_fma((0.27f * sunColor.y) * bac, foc, lin.y),
           rm_fma((0.27f * sunColor.z) * bac, foc, lin.z));
  *colOut = v3_mul(col, lin);
  return 1;
}
/* ---- night sky and sea (frag:476-516, 591-598, 1562-1573, 2160-2310); both read the `noise` texture ---- */
static float hashSin2(float px, float py) { /* frag:481-483 */
  return rm_fract(rm_sin(dot2(px, py, 12.9898f, 78.233f)) * 43758.5453f);
}
static float noiseW(float px, float py) { /* frag:504-518 */
  float ix = rm_floor(px), iy = rm_floor(py), fx = rm_fract(px), fy = rm_fract(py);
  float ux = (fx * fx) * rm_fma(-2.0f, fx, 3.0f), uy = (fy * fy) * rm_fma(-2.0f, fy, 3.0f);
  float a = hashSin2(ix + 0.0f, iy + 0.0f), b = hashSin2(ix + 1.0f, iy + 0.0f);
  float cc = hashSin2(ix + 0.0f, iy + 1.0f), d = hashSin2(ix + 1.0f, iy + 1.0f);
  float result = rm_mix(rm_mix(a, b, ux), rm_mix(cc, d, ux), uy);
  return rm_fma(2.0f, result, -1.0f);
}
static float noiseV(const Ctx *c, v3 x) { /* frag:591-598; textureLod(noise, ·, 0).yx */
  v3 p = V3(rm_floor(x.x), rm_floor(x.y), rm_floor(x.z));
  v3 f = V3(rm_fract(x.x), rm_fract(x.y), rm_fract(x.z));
  f = V3((f.x * f.x) * rm_fma(-2.0f, f.x, 3.0f), (f.y * f.y) * rm_fma(-2.0f, f.y, 3.0f), (f.z * f.z) * rm_fma(-2.0f, f.z, 3.0f));
  float u = rm_fma(37.0f, p.z, p.x) + f.x, v = rm_fma(239.0f, p.z, p.y) + f.y;
  v4 t = sampleRGBA8(&c->res->noise, (u + 0.5f) / 256.0f, (v + 0.5f) / 256.0f, 0);
  return rm_fma(rm_mix(t.y, t.x, f.z), 2.0f, -1.0f);
}
static v3 getMoonColor(const Ctx *c, v3 rd) { /* frag:1562-1573, MOON = normalize(-0.4, 0.4, 0.3) (frag:107) */
  const v3 MOON = normalize3(V3(-0.4f, 0.4f, 0.3f));
  float ms = noiseV(c, v3_scale(rd, 20.0f));
  float q = (0.1f * ms) * ms; /* 0.1*ms*ms*ms associates left to right */
  v3 mCol = V3(rm_fma(-q, ms, 0.5f), rm_fma(-q, ms, 0.5f), rm_fma(-q, ms, 0.3f));
  float moonDot = dot3(MOON, rd);
  float moonA = rm_smoothstep(0.9985f, 0.999f, moonDot);
  v3 col = v3_scale(mCol, moonA);
  float halo = rm_smoothstep(0.91f, 0.9985f, moonDot);
  col = V3(rm_fma(0.15f, halo, col.x), rm_fma(0.15f, halo, col.y), rm_fma(0.15f, halo, col.z));
  float sh = 6.0f * rm_sin(c->g.iTime / 2.0f);
  float star = rm_smoothstep(0.99f, 0.999f, noiseV(c, V3(rm_floor(rm_fma(rd.x, 202.0f, -sh)), rm_floor(rm_fma(rd.y, 202.0f, -sh)),
                                                        rm_floor(rm_fma(rd.z, 202.0f, -sh)))));
  float sc = rm_clamp(star, 0.0f, 1.0f);
  return V3(rm_fma(sc, 0.4f, col.x), rm_fma(sc, 0.4f, col.y), rm_fma(sc, 0.4f, col.z));
}
#define SEA_HEIGHT 0.2f /* frag:96-99 */
#define SEA_CHOPPY 1.0f
#define SEA_SPEED 0.5f
#define SEA_FREQ 0.16f
static float sea_octave(float ux, float uy, float choppy) { /* frag:2162-2169 */
  float n = noiseW(ux, uy);
  ux += n; uy += n;
  float wx = 1.0f - rm_abs(rm_sin(ux)), wy = 1.0f - rm_abs(rm_sin(uy));
  float sx = rm_abs(rm_cos(ux)), sy = rm_abs(rm_cos(uy));
  wx = rm_mix(wx, sx, wx); wy = rm_mix(wy, sy, wy);
  return rm_pow(1.0f - rm_pow(wx * wy, 0.65f), choppy);
}
/* seaMap (ITER_GEOMETRY = 3) / seaMapD (ITER_FRAGMENT = 5), frag:2195-2241.  SEA_TIME = 1 + iTime·SEA_SPEED is a
 * global initialised from a uniform (frag:2192; not valid GLSL 3.30 — the evident meaning is taken). */
static float seaMap(const Ctx *c, v3 p, int iters) {
  const float seaTime = rm_fma(c->g.iTime, SEA_SPEED, 1.0f);
  float freq = SEA_FREQ, amp = SEA_HEIGHT, choppy = SEA_CHOPPY, ux = p.x, uy = p.z, h = 0.0f;
  for (int i = 0; i < iters; i++) {
    float d = sea_octave((ux + seaTime) * freq, (uy + seaTime) * freq, choppy);
    d += sea_octave((ux - seaTime) * freq, (uy - seaTime) * freq, choppy);
    h = rm_fma(d, amp, h);
    float nx = dot2(ux, uy, 1.6f, 1.2f), ny = dot2(ux, uy, -1.2f, 1.6f); /* uv *= octave_m (row vector × mat2, frag:103) */
    ux = nx; uy = ny;
    freq *= 2.0f; amp *= 0.2f;
    choppy = rm_mix(choppy, 1.0f, 0.2f);
  }
  return p.y - h;
}
static v3 getSeaNormal(const Ctx *c, v3 p, float eps) { /* frag:2243-2250 */
  float ny = seaMap(c, p, 5);
  float nx = seaMap(c, V3(p.x + eps, p.y, p.z), 5) - ny;
  float nz = seaMap(c, V3(p.x, p.y, p.z + eps), 5) - ny;
  return normalize3(V3(nx, eps, nz));
}
static float seaMapHeight(const Ctx *c, v3 ro, v3 rd, v3 *p, float maxT) { /* frag:2252-2282 */
  float tm = 0.0f, tx = 1000.0f;
  float hx = seaMap(c, v3_madd(rd, tx, ro), 3);
  if (hx > 0.0f) { *p = V3(0.0f, 0.0f, 0.0f); return tx; }
  float hm = seaMap(c, v3_madd(rd, tm, ro), 3);
  float tmid = 0.0f;
  for (int i = 0; i < 8; i++) {
    float f = rm_divr(hm, hm - hx);
    tmid = rm_mix(tm, tx, f);
    *p = v3_madd(rd, tmid, ro);
    if (tmid > maxT) return -1.0f;
    float hmid = seaMap(c, *p, 3);
    if (hmid < 0.0f) { tx = tmid; hx = hmid; } else { tm = tmid; hm = hmid; }
  }
  return tmid;
}
static v3 getSeaColor(const Ctx *c, v3 p, v3 n, v3 l, v3 eye, v3 dist) { /* frag:2171-2190 */
  const v3 SEA_BASE = V3(0.4f, 0.49f, 0.48f), SEA_WATER = V3(0.8f, 0.9f, 0.6f); /* frag:101-102 */
  float fresnel = rm_clamp(1.0f - dot3(n, v3_neg(eye)), 0.0f, 1.0f);
  fresnel = rm_pow(fresnel, 3.0f) * 0.65f;
  v3 refl = reflect3(eye, n);
  v3 reflected = getMoonColor(c, refl);
  float pw = rm_pow(rm_fma(dot3(n, l), 0.4f, 0.6f), 80.0f);
  v3 refracted = V3(rm_fma(pw * SEA_WATER.x, 0.12f, SEA_BASE.x), rm_fma(pw * SEA_WATER.y, 0.12f, SEA_BASE.y),
                    rm_fma(pw * SEA_WATER.z, 0.12f, SEA_BASE.z));
  v3 color = mix3(refracted, reflected, fresnel);
  float atten = rm_max(rm_fma(-dot3(dist, dist), 0.001f, 1.0f), 0.0f);
  float dh = p.y - SEA_HEIGHT;
  color = V3(rm_fma((SEA_WATER.x * dh) * 0.18f, atten, color.x), rm_fma((SEA_WATER.y * dh) * 0.18f, atten, color.y),
             rm_fma((SEA_WATER.z * dh) * 0.18f, atten, color.z));
  const float nrm = (60.0f + 8.0f) / (3.14159265f * 8.0f);
  float spec = rm_pow(rm_max(dot3(refl, l), 0.0f), 60.0f) * nrm;
  return V3(color.x + spec, color.y + spec, color.z + spec);
}
/* frag:2284-2310.  The miss path is a bare `return;` in a non-void function (frag:2290, UB7): taken as returning
 * `ri` as filled so far (colour = bgCol, d = maxT, no hit). */
static int seaRender(const Ctx *c, v3 ro, v3 rd, float maxT, v3 bgCol, v3 *colOut, float *dOut) {
  *colOut = bgCol; *dOut = maxT;
  v3 p;
  float t = seaMapHeight(c, ro, rd, &p, maxT);
  if (len3(p) == 0.0f || t == -1.0f) return 0;
  *dOut = t;
  v3 d = v3_sub(p, ro);
  v3 n = getSeaNormal(c, p, (dot3(d, d) * 0.1f) / (float)c->W);
  v3 s = getSky(rd);
  v3 sc = getSeaColor(c, p, n, getSunDir(), rd, d);
  float t2 = rm_pow(rm_smoothstep(0.0f, -0.05f, rd.y), 0.3f);
  *colOut = fog(mix3(s, sc, t2), t);
  return 1;
}
/* The env layers applied after a render() (frag:2444-2456, 2506-2518, 2555-2567): sea, terrain, then cloud. */
static void envLayers(const Ctx *c, v3 ro, v3 rd, float d, v3 bgCol, int *terrainHit, int *cloudHit, int *seaHit, v3 *tcol,
                      v3 *ccol, v3 *scol) {
  float sd = d, td = d; /* sr.d = tr.d = ri.d (frag:2444); the cloud layer is bounded by tr.d, not sr.d */
  *terrainHit = 0; *cloudHit = 0; *seaHit = 0;
  if (c->s.features & RM_FEAT_SEA) *seaHit = seaRender(c, ro, rd, d, bgCol, scol, &sd);
  if (c->s.features & RM_FEAT_TERRAIN) *terrainHit = terrainRender(c, ro, rd, sd, bgCol, tcol, &td);
  if (c->s.features & RM_FEAT_CLOUD) *ccol = cloudRender(c, ro, rd, bgCol, cloudHit, td);
}

/* ---------------------------------------------------------------- render (frag:2318-2375) */
static RenderInfo render(Ctx *c, v3 ro, v3 rd, IntersectionInfo *info, float side, float maxT, v3 bgCol) {
  RenderInfo ri;
  info->intersectObj = -1;
  RayMarchRes res = raymarch(c, ro, rd, maxT, side);
  if (res.intersectObj == -1) {
    ri.fragColor = V4(bgCol.x, bgCol.y, bgCol.z, 1.0f);
    if (c->s.enableSkyBox) { /* frag:2327 */
      v3 sk = sampleCube(c->res->skybox, rd);
      ri.fragColor = V4(sk.x, sk.y, sk.z, 1.0f);
    }
    ri.isEnv = 1; ri.d = maxT;
    return ri;
  }
  ri.isEnv = 0; ri.d = res.d;
  c->nShade++;
  v3 p = v3_madd(rd, res.d, ro);
  v3 pn = getNormal(c, p);
  if (c->s.features & RM_FEAT_PERLIN_BUMP) pn = bumpNormal(pn, p, 10.0f, 2.0f);
  const RmObject *obj = &c->objs[res.intersectObj];
  v3 col;
  if (obj->isEmissive) { /* frag:2339-2342: the rectangle of an area light; `info` keeps intersectObj = -1 */
    ri.fragColor = V4(obj->color[0], obj->color[1], obj->color[2], 1.0f);
    return ri;
  }
  if (obj->type == RM_MANDELBULB) { /* frag:2354-2361 */
    col = V3(0.2f, 0.2f, 0.2f);
    col = mix3(col, V3(0.10f, 0.20f, 0.30f), rm_clamp(res.trap.y, 0.0f, 1.0f));
    col = mix3(col, V3(0.02f, 0.10f, 0.30f), rm_clamp(res.trap.z * res.trap.z, 0.0f, 1.0f));
    col = mix3(col, V3(0.30f, 0.10f, 0.02f), rm_clamp(rm_pow(res.trap.w, 6.0f), 0.0f, 1.0f));
    col = v3_scale(col, 0.5f);
    v3 ph = getPhong(c, pn, res.intersectObj, p, rd, maxT);
    col = V3(col.x * (ph.x * 8.0f), col.y * (ph.y * 8.0f), col.z * (ph.z * 8.0f));
  } else if (obj->type == RM_MENGERSPONGE) { /* frag:2362-2365 */
    col = V3(rm_fma(0.5f, rm_cos(rm_fma(2.0f, res.trap.z, 0.0f)), 0.5f),
             rm_fma(0.5f, rm_cos(rm_fma(2.0f, res.trap.z, 1.0f)), 0.5f),
             rm_fma(0.5f, rm_cos(rm_fma(2.0f, res.trap.z, 2.0f)), 0.5f));
    col = v3_mul(col, getPhong(c, pn, res.intersectObj, p, rd, maxT));
  } else {
    col = getPhong(c, pn, res.intersectObj, p, rd, maxT);
  }
  info->p = p; info->n = pn; info->rd = rd; info->intersectObj = res.intersectObj;
  ri.fragColor = V4(col.x, col.y, col.z, 1.0f);
  return ri;
}

/* frag:1938-1946 */
static v4 brightOf(v3 color) {
  float brightness = dot3(color, V3(0.2126f, 0.7152f, 0.0722f));
  if (brightness > 1.0f) return V4(color.x, color.y, color.z, 1.0f);
  return V4(0.0f, 0.0f, 0.0f, 1.0f);
}

/* raymarch.vert:13-25 + frag:2383-2427 + frag:2429-2575 for pixel (px,py), py = 0 at the bottom */
static void shadePixel(Ctx *c, int px, int py, int W, int H, float *outColor, float *outBright) {
  /* The pixel centre in the full-screen quad: which triangle, and the weights of that triangle's vertices 1 and 2
   * (see rayPlanes).  twoDFragCoord = pos (vert:18) is a varying too: −1 + 2·I + 0·J below the diagonal, 1 − 2·I above. */
  const float tx = ((float)px + 0.5f) / (float)W, ty = ((float)py + 0.5f) / (float)H;
  const int upper = (tx + ty) > 1.0f;
  const float I = upper ? 1.0f - tx : tx, J = upper ? 1.0f - ty : ty;
  const float ndcx = upper ? rm_fma(I, -2.0f, 1.0f) : rm_fma(I, 2.0f, -1.0f);
  const float ndcy = upper ? rm_fma(J, -2.0f, 1.0f) : rm_fma(J, 2.0f, -1.0f);
  v4 fragColor, bright = V4(0.0f, 0.0f, 0.0f, 1.0f);
  if (c->g.isTwoD) { /* frag:2431, 2377-2380 */
    float scol = sdMandelBrot(c, ndcx, ndcy);
    fragColor = V4(rm_pow(scol, 0.9f), rm_pow(scol, 1.1f), rm_pow(scol, 1.4f), 1.0f);
    goto done;
  }
  {
    /* vert:23-24 */
    /* vert:23-24 at the corners (c->rayPlane), interpolated by the rasteriser */
    v4 nearClip = interpolateVarying(c->rayPlane[upper][0], I, J);
    v4 farClip = interpolateVarying(c->rayPlane[upper][1], I, J);
    /* frag:2388-2392 */
    v3 ro = V3(nearClip.x / nearClip.w, nearClip.y / nearClip.w, nearClip.z / nearClip.w);
    v3 farC = V3(farClip.x / farClip.w, farClip.y / farClip.w, farClip.z / farClip.w);
    v3 rd = normalize3(v3_sub(farC, ro));
    /* frag:2405-2419 (later #ifdefs override earlier ones) */
    v3 bgCol = V3(0.0f, 0.0f, 0.0f);
    if (c->s.features & RM_FEAT_SKY_BACKGROUND) bgCol = getSky(rd);
    if (c->s.features & RM_FEAT_NIGHTSKY_BACKGROUND) bgCol = getMoonColor(c, rd);
    if (c->s.features & RM_FEAT_WHITE_BACKGROUND) bgCol = V3(1.0f, 1.0f, 1.0f);
    if (c->s.features & RM_FEAT_DARK_BACKGROUND) bgCol = V3(0.0f, 0.0f, 0.0f);
    const int env = (c->s.features & (RM_FEAT_TERRAIN | RM_FEAT_CLOUD | RM_FEAT_SEA)) != 0;
    float far = (c->s.features & RM_FEAT_CLOUD) ? 2000.0f : c->cam->initialFar; /* frag:2422-2426 */

    IntersectionInfo info, oi;
    RenderInfo ri = render(c, ro, rd, &info, OUTSIDE, far, bgCol); /* frag:2443 */
    int terrainHit = 0, cloudHit = 0, seaHit = 0;
    v3 tcol = bgCol, ccol = bgCol, scol = bgCol;
    if (env) envLayers(c, ro, rd, ri.d, bgCol, &terrainHit, &cloudHit, &seaHit, &tcol, &ccol, &scol); /* frag:2444-2456 */
    if (ri.isEnv && !cloudHit && !terrainHit && !seaHit) { /* frag:2459-2465 */
      fragColor = ri.fragColor;
      goto done;
    } else if (cloudHit) { /* frag:2466-2468 */
      fragColor = V4(ccol.x, ccol.y, ccol.z, 1.0f);
      bright = brightOf(ccol);
      goto done;
    } else if (terrainHit) { /* frag:2469-2471 */
      fragColor = V4(tcol.x, tcol.y, tcol.z, 1.0f);
      bright = brightOf(tcol);
      goto done;
    } else if (seaHit) { /* frag:2472-2474 */
      fragColor = V4(scol.x, scol.y, scol.z, 1.0f);
      bright = brightOf(scol);
      goto done;
    }
    c->nHit++;
    v4 phong = ri.fragColor;
    v4 refl = V4(0, 0, 0, 0), refr = V4(0, 0, 0, 0);
    oi = info; /* frag:2481 */
    /* UB5: an emissive hit leaves info.intersectObj = -1 and the shader reads objects[-1] (frag:2341, 2483); an
     * out-of-range uniform read is taken as zeros (robust-access behaviour): no secondary rays. */
    static const RmObject kZeroObject;
    const RmObject *obj = info.intersectObj >= 0 ? &c->objs[info.intersectObj] : &kZeroObject;
    v3 cRefl = V3(obj->cReflective[0], obj->cReflective[1], obj->cReflective[2]);
    v3 cRefr = V3(obj->cTransparent[0], obj->cTransparent[1], obj->cTransparent[2]);
    if (c->s.enableReflection && len3(cRefl) != 0.0f) { /* frag:2491-2524 */
      v3 fil = V3(1.0f, 1.0f, 1.0f);
      for (int i = 0; i < c->s.numReflection; i++) {
        v3 r = reflect3(info.rd, info.n);
        v3 sro = V3(rm_fma(r.x * SURFACE_DIST, 3.0f, info.p.x), rm_fma(r.y * SURFACE_DIST, 3.0f, info.p.y),
                    rm_fma(r.z * SURFACE_DIST, 3.0f, info.p.z));
        fil = v3_mul(fil, cRefl);
        RenderInfo res = render(c, sro, r, &info, OUTSIDE, far, bgCol);
        if (env) { /* frag:2506-2518: terrain, then cloud, override the bounce colour and end the loop */
          int th, ch, sh; v3 tc, cc, sc;
          envLayers(c, sro, r, res.d, bgCol, &th, &ch, &sh, &tc, &cc, &sc);
          if (sh) res.fragColor = V4(sc.x, sc.y, sc.z, 1.0f); /* frag:2509 sets sr.isEnv, not res.isEnv: the loop goes on */
          if (th) { res.fragColor = V4(tc.x, tc.y, tc.z, 1.0f); res.isEnv = 1; }
          if (ch) { res.fragColor = V4(cc.x, cc.y, cc.z, 1.0f); res.isEnv = 1; }
        }
        refl.x += (c->g.ks * fil.x) * res.fragColor.x;
        refl.y += (c->g.ks * fil.y) * res.fragColor.y;
        refl.z += (c->g.ks * fil.z) * res.fragColor.z;
        refl.w += 1.0f;
        if (res.isEnv) break;
      }
    }
    if (c->s.enableRefraction && len3(cRefr) != 0.0f) { /* frag:2526-2570 */
      const RmObject *o2 = &c->objs[oi.intersectObj];
      float ior = o2->ior;
      v3 ct = V3(o2->cTransparent[0], o2->cTransparent[1], o2->cTransparent[2]);
      v3 rdIn = refract3(oi.rd, oi.n, 1.0f / ior);
      v3 pEnter = V3(rm_fma(-(oi.n.x * SURFACE_DIST), 3.0f, oi.p.x), rm_fma(-(oi.n.y * SURFACE_DIST), 3.0f, oi.p.y),
                     rm_fma(-(oi.n.z * SURFACE_DIST), 3.0f, oi.p.z));
      float dIn = raymarch(c, pEnter, rdIn, far, INSIDE).d;
      v3 pExit = v3_madd(rdIn, dIn, pEnter);
      v3 nExit = v3_neg(getNormal(c, pExit));
      v3 rdOut = refract3(rdIn, nExit, ior);
      if (len3(rdOut) == 0.0f) {
        refr = V4(0, 0, 0, 0);
      } else {
        v3 sro = V3(rm_fma(-(nExit.x * SURFACE_DIST), 5.0f, pExit.x), rm_fma(-(nExit.y * SURFACE_DIST), 5.0f, pExit.y),
                    rm_fma(-(nExit.z * SURFACE_DIST), 5.0f, pExit.z));
        RenderInfo res = render(c, sro, rdOut, &info, OUTSIDE, far, bgCol);
        if (env) { /* frag:2555-2567 */
          int th, ch, sh; v3 tc, cc, sc;
          envLayers(c, sro, rdOut, res.d, bgCol, &th, &ch, &sh, &tc, &cc, &sc);
          if (sh) res.fragColor = V4(sc.x, sc.y, sc.z, 1.0f);
          if (th) res.fragColor = V4(tc.x, tc.y, tc.z, 1.0f);
          if (ch) res.fragColor = V4(cc.x, cc.y, cc.z, 1.0f);
        }
        refr.x += (c->g.kt * ct.x) * res.fragColor.x;
        refr.y += (c->g.kt * ct.y) * res.fragColor.y;
        refr.z += (c->g.kt * ct.z) * res.fragColor.z;
        refr.w += 1.0f;
      }
    }
    /* frag:2572-2574 */
    fragColor = V4((phong.x + refl.x) + refr.x, (phong.y + refl.y) + refr.y, (phong.z + refl.z) + refr.z,
                   (phong.w + refl.w) + refr.w);
    bright = brightOf(V3(fragColor.x, fragColor.y, fragColor.z));
  }
done:
  outColor[0] = fragColor.x; outColor[1] = fragColor.y; outColor[2] = fragColor.z; outColor[3] = fragColor.w;
  if (outBright) { outBright[0] = bright.x; outBright[1] = bright.y; outBright[2] = bright.z; outBright[3] = bright.w; }
}


/* ---------------------------------------------------------------- post passes (blur.frag, hdr.frag, fxaa.frag) */
#include <immintrin.h>
/* storage in an RGBA16F target: round to binary16 (nearest even) and back */
static inline float q16(float v) { return _cvtsh_ss(_cvtss_sh(v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)); }
/* storage in an RGBA8 target: clamp, ×255, round half up, /255 */
static inline float q8(float v) {
  v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
  return (float)(int)rm_fma(v, 255.0f, 0.5f) / 255.0f;
}
static inline int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

/* blur.frag:9-31: one pass; src/dst are H·W·3 binary16-valued floats; CLAMP_TO_EDGE; taps hit texel centres */
static void blurPass(const float *src, float *dst, int W, int H, int horizontal) {
  static const float w[5] = {0.2270270270f, 0.1945945946f, 0.1216216216f, 0.0540540541f, 0.0162162162f};
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++)
      for (int k = 0; k < 3; k++) {
        float r = src[((size_t)y * W + x) * 3 + k] * w[0];
        for (int i = 1; i < 5; i++) {
          int xp = horizontal ? clampi(x + i, W) : x, yp = horizontal ? y : clampi(y + i, H);
          int xm = horizontal ? clampi(x - i, W) : x, ym = horizontal ? y : clampi(y - i, H);
          r = rm_fma(src[((size_t)yp * W + xp) * 3 + k], w[i], r);
          r = rm_fma(src[((size_t)ym * W + xm) * 3 + k], w[i], r);
        }
        dst[((size_t)y * W + x) * 3 + k] = q16(r);
      }
}

/* bilinear GL_LINEAR / GL_REPEAT fetch of an RGB float image at normalised (u, v) — the FXAA source texture */
static v3 fetchLinearRepeat(const float *img, int W, int H, float u, float v) {
  float fx = rm_fma(u, (float)W, -0.5f), fy = rm_fma(v, (float)H, -0.5f);
  float x0 = rm_floor(fx), y0 = rm_floor(fy);
  float a = fx - x0, b = fy - y0;
  int i0 = wrapi(x0, W), j0 = wrapi(y0, H);
  int i1 = (i0 + 1 == W) ? 0 : i0 + 1, j1 = (j0 + 1 == H) ? 0 : j0 + 1;
  const float *p00 = img + ((size_t)j0 * W + i0) * 3, *p10 = img + ((size_t)j0 * W + i1) * 3;
  const float *p01 = img + ((size_t)j1 * W + i0) * 3, *p11 = img + ((size_t)j1 * W + i1) * 3;
  return V3(rm_mix(rm_mix(p00[0], p10[0], a), rm_mix(p01[0], p11[0], a), b),
            rm_mix(rm_mix(p00[1], p10[1], a), rm_mix(p01[1], p11[1], a), b),
            rm_mix(rm_mix(p00[2], p10[2], a), rm_mix(p01[2], p11[2], a), b));
}
static inline float rgb2luma(v3 c) { return rm_sqrt(dot3(c, V3(0.299f, 0.587f, 0.114f))); } /* fxaa.frag:18-20 */

/* fxaa.frag:22-166 for the pixel (x, y); img = RGB floats holding 8-bit values */
static v3 fxaaPixel(const float *img, int W, int H, int x, int y) {
  static const float quality[12] = {1.0f, 1.0f, 1.0f, 1.0f, 1.0f, 1.5f, 2.0f, 2.0f, 2.0f, 2.0f, 4.0f, 8.0f};
  const float invW = 1.0f / (float)W, invH = 1.0f / (float)H; /* configureFXAAUniforms, realtimerender.cpp:815-821 */
  const float tu = ((float)x + 0.5f) / (float)W, tv = ((float)y + 0.5f) / (float)H;
#define TEXOFF(dx, dy) fetchLinearRepeat(img, W, H, tu + (float)(dx) * invW, tv + (float)(dy) * invH)
  v3 colorCenter = TEXOFF(0, 0);
  float lumaCenter = rgb2luma(colorCenter);
  float lumaDown = rgb2luma(TEXOFF(0, -1)), lumaUp = rgb2luma(TEXOFF(0, 1));
  float lumaLeft = rgb2luma(TEXOFF(-1, 0)), lumaRight = rgb2luma(TEXOFF(1, 0));
  float lumaMin = rm_min(lumaCenter, rm_min(rm_min(lumaDown, lumaUp), rm_min(lumaLeft, lumaRight)));
  float lumaMax = rm_max(lumaCenter, rm_max(rm_max(lumaDown, lumaUp), rm_max(lumaLeft, lumaRight)));
  float lumaRange = lumaMax - lumaMin;
  if (lumaRange < rm_max(0.0312f, lumaMax * 0.125f)) return colorCenter;
  float lumaDownLeft = rgb2luma(TEXOFF(-1, -1)), lumaUpRight = rgb2luma(TEXOFF(1, 1));
  float lumaUpLeft = rgb2luma(TEXOFF(-1, 1)), lumaDownRight = rgb2luma(TEXOFF(1, -1));
#undef TEXOFF
  float lumaDownUp = lumaDown + lumaUp, lumaLeftRight = lumaLeft + lumaRight;
  float lumaLeftCorners = lumaDownLeft + lumaUpLeft, lumaDownCorners = lumaDownLeft + lumaDownRight;
  float lumaRightCorners = lumaDownRight + lumaUpRight, lumaUpCorners = lumaUpRight + lumaUpLeft;
  float edgeHorizontal = (rm_abs(rm_fma(-2.0f, lumaLeft, lumaLeftCorners)) + rm_abs(rm_fma(-2.0f, lumaCenter, lumaDownUp)) * 2.0f) +
                         rm_abs(rm_fma(-2.0f, lumaRight, lumaRightCorners));
  float edgeVertical = (rm_abs(rm_fma(-2.0f, lumaUp, lumaUpCorners)) + rm_abs(rm_fma(-2.0f, lumaCenter, lumaLeftRight)) * 2.0f) +
                       rm_abs(rm_fma(-2.0f, lumaDown, lumaDownCorners));
  int isHorizontal = edgeHorizontal >= edgeVertical;
  float luma1 = isHorizontal ? lumaDown : lumaLeft, luma2 = isHorizontal ? lumaUp : lumaRight;
  float gradient1 = luma1 - lumaCenter, gradient2 = luma2 - lumaCenter;
  int is1Steepest = rm_abs(gradient1) >= rm_abs(gradient2);
  float gradientScaled = 0.25f * rm_max(rm_abs(gradient1), rm_abs(gradient2));
  float stepLength = isHorizontal ? invH : invW;
  float lumaLocalAverage;
  if (is1Steepest) { stepLength = -stepLength; lumaLocalAverage = 0.5f * (luma1 + lumaCenter); }
  else lumaLocalAverage = 0.5f * (luma2 + lumaCenter);
  float cu = tu, cv = tv;
  if (isHorizontal) cv = rm_fma(stepLength, 0.5f, cv); else cu = rm_fma(stepLength, 0.5f, cu);
  float ox = isHorizontal ? invW : 0.0f, oy = isHorizontal ? 0.0f : invH;
  float u1 = cu - ox, v1 = cv - oy, u2 = cu + ox, v2 = cv + oy;
  float lumaEnd1 = rgb2luma(fetchLinearRepeat(img, W, H, u1, v1)) - lumaLocalAverage;
  float lumaEnd2 = rgb2luma(fetchLinearRepeat(img, W, H, u2, v2)) - lumaLocalAverage;
  int reached1 = rm_abs(lumaEnd1) >= gradientScaled, reached2 = rm_abs(lumaEnd2) >= gradientScaled;
  int reachedBoth = reached1 && reached2;
  if (!reached1) { u1 -= ox; v1 -= oy; }
  if (!reached2) { u2 += ox; v2 += oy; }
  if (!reachedBoth) {
    for (int i = 2; i < 12; i++) {
      if (!reached1) lumaEnd1 = rgb2luma(fetchLinearRepeat(img, W, H, u1, v1)) - lumaLocalAverage;
      if (!reached2) lumaEnd2 = rgb2luma(fetchLinearRepeat(img, W, H, u2, v2)) - lumaLocalAverage;
      reached1 = rm_abs(lumaEnd1) >= gradientScaled;
      reached2 = rm_abs(lumaEnd2) >= gradientScaled;
      reachedBoth = reached1 && reached2;
      if (!reached1) { u1 = rm_fma(-ox, quality[i], u1); v1 = rm_fma(-oy, quality[i], v1); }
      if (!reached2) { u2 = rm_fma(ox, quality[i], u2); v2 = rm_fma(oy, quality[i], v2); }
      if (reachedBoth) break;
    }
  }
  float distance1 = isHorizontal ? (tu - u1) : (tv - v1), distance2 = isHorizontal ? (u2 - tu) : (v2 - tv);
  int isDirection1 = distance1 < distance2;
  float distanceFinal = rm_min(distance1, distance2);
  float edgeThickness = distance1 + distance2;
  float pixelOffset = -distanceFinal / edgeThickness + 0.5f;
  int isLumaCenterSmaller = lumaCenter < lumaLocalAverage;
  int correctVariation = ((isDirection1 ? lumaEnd1 : lumaEnd2) < 0.0f) != isLumaCenterSmaller;
  float finalOffset = correctVariation ? pixelOffset : 0.0f;
  float lumaAverage = (1.0f / 12.0f) * ((rm_fma(2.0f, lumaDownUp + lumaLeftRight, lumaLeftCorners)) + lumaRightCorners);
  float sub1 = rm_clamp(rm_abs(lumaAverage - lumaCenter) / lumaRange, 0.0f, 1.0f);
  float sub2 = (rm_fma(-2.0f, sub1, 3.0f) * sub1) * sub1;
  float subFinal = (sub2 * sub2) * 0.875f;
  finalOffset = rm_max(finalOffset, subFinal);
  float fu = tu, fv = tv;
  if (isHorizontal) fv = rm_fma(finalOffset * stepLength, 1.0f, fv); else fu = rm_fma(finalOffset * stepLength, 1.0f, fu);
  return fetchLinearRepeat(img, W, H, fu, fv);
}

int rmo_post_process(const float *frag, const float *bright, float *out, int W, int H, const RmPostSettings *ps) {
  if (!frag || !out || !ps || W <= 0 || H <= 0) return RM_ERR_INVALID_ARGUMENT;
  const size_t n = (size_t)W * H;
  const int light = ps->enableHDR || ps->enableGammaCorrection || ps->enableBloom;
  if (ps->enableBloom && !bright) return RM_ERR_INVALID_ARGUMENT;
  float *stage = (float *)malloc(n * 3 * sizeof(float)); /* RGB of the stage that feeds FXAA / the screen */
  if (!stage) return RM_ERR_IO;
  if (light) {
    float *bl = NULL;
    if (ps->enableBloom) { /* applyBloom, realtimerender.cpp:92-108: the reference composites pass 9 of 10 */
      float *a = (float *)malloc(n * 3 * sizeof(float)), *b = (float *)malloc(n * 3 * sizeof(float));
      for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) a[i * 3 + k] = q16(bright[i * 4 + k]);
      int horizontal = 1;
      float *src = a, *dst = b;
      for (int i = 0; i < 9; i++) {
        blurPass(src, dst, W, H, horizontal);
        float *t = src; src = dst; dst = t;
        horizontal = !horizontal;
      }
      bl = src;
      free(dst);
    }
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) { /* hdr.frag:13-35 */
        float hc = q16(frag[i * 4 + k]);
        float r;
        if (!ps->enableHDR && !ps->enableBloom) r = rm_pow(hc, 1.0f / 2.2f);
        else {
          if (ps->enableBloom) hc = hc + bl[i * 3 + k];
          r = 1.0f - rm_exp((-hc) * ps->exposure);
        }
        stage[i * 3 + k] = r;
      }
    free(bl);
  } else {
    for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) stage[i * 3 + k] = frag[i * 4 + k];
  }
  if (ps->enableFXAA) { /* the FXAA source is the RGBA8 m_customFBOColorTexture */
    for (size_t i = 0; i < n * 3; i++) stage[i] = q8(stage[i]);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++) {
        v3 c = fxaaPixel(stage, W, H, x, y);
        float *o = out + ((size_t)y * W + x) * 4;
        o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = 1.0f;
      }
  } else {
    for (size_t i = 0; i < n; i++) {
      out[i * 4] = stage[i * 3]; out[i * 4 + 1] = stage[i * 3 + 1]; out[i * 4 + 2] = stage[i * 3 + 2];
      out[i * 4 + 3] = light ? 1.0f : frag[i * 4 + 3];
    }
  }
  free(stage);
  return RM_OK;
}

/* ---------------------------------------------------------------- public oracle API */
static int texOk(const RmTexture *t) { return t->pixels && t->width > 0 && t->height > 0; }
static int validate(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights,
                    int numLights, const RmGlobals *g, const RmSettings *s, const RmResources *res) {
  if (!cam || !g || !s || (numObjects > 0 && !objs) || (numLights > 0 && !lights)) return RM_ERR_INVALID_ARGUMENT;
  if (numObjects < 0 || numLights < 0) return RM_ERR_INVALID_ARGUMENT;
  if (numObjects > RM_MAX_OBJECTS || numLights > RM_MAX_LIGHTS) return RM_ERR_CAPACITY;
  if (res->numTextures < 0 || res->numTextures > RM_MAX_TEXTURES) return RM_ERR_CAPACITY;
  if ((s->features & (RM_FEAT_NIGHTSKY_BACKGROUND | RM_FEAT_SEA)) && !texOk(&res->noise)) return RM_ERR_UNSUPPORTED;
  if (s->enableSkyBox)
    for (int f = 0; f < 6; f++)
      if (!texOk(&res->skybox[f])) return RM_ERR_UNSUPPORTED;
  for (int i = 0; i < numObjects; i++) {
    if (objs[i].type < 0 || objs[i].type >= RM_CUSTOM) return RM_ERR_UNSUPPORTED;
    if (objs[i].texLoc != -1) {
      if (objs[i].texLoc < 0 || objs[i].texLoc >= res->numTextures || !res->textures) return RM_ERR_UNSUPPORTED;
      if (objs[i].type != RM_CUBE && objs[i].type != RM_CONE && objs[i].type != RM_CYLINDER && objs[i].type != RM_SPHERE)
        return RM_ERR_UNSUPPORTED;
      if (!texOk(&res->textures[objs[i].texLoc])) return RM_ERR_INVALID_ARGUMENT;
    }
  }
  for (int i = 0; i < numLights; i++) {
    if (lights[i].type < 0 || lights[i].type > RM_LIGHT_AREA) return RM_ERR_UNSUPPORTED;
    if (lights[i].type == RM_LIGHT_AREA && (!res->ltc1 || !res->ltc2)) return RM_ERR_UNSUPPORTED;
  }
  return RM_OK;
}

int rmo_render(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
               const RmGlobals *g, const RmSettings *s, int W, int H, int rowBegin, int rowEnd, float *rgba,
               float *bright, RmCounters *counters, int threads) {
  return rmo_render_res(cam, objs, numObjects, lights, numLights, g, s, NULL, W, H, rowBegin, rowEnd, rgba, bright,
                        counters, threads);
}

int rmo_render_tex(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                   const RmGlobals *g, const RmSettings *s, const RmTexture *textures, int numTextures, int W, int H,
                   int rowBegin, int rowEnd, float *rgba, float *bright, RmCounters *counters, int threads) {
  RmResources r;
  memset(&r, 0, sizeof r);
  r.textures = textures; r.numTextures = numTextures;
  return rmo_render_res(cam, objs, numObjects, lights, numLights, g, s, &r, W, H, rowBegin, rowEnd, rgba, bright,
                        counters, threads);
}

int rmo_render_res(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                   const RmGlobals *g, const RmSettings *s, const RmResources *resIn, int W, int H, int rowBegin,
                   int rowEnd, float *rgba, float *bright, RmCounters *counters, int threads) {
  RmResources none;
  memset(&none, 0, sizeof none);
  const RmResources *res = resIn ? resIn : &none;
  int st = validate(cam, objs, numObjects, lights, numLights, g, s, res);
  if (st != RM_OK) return st;
  if (W <= 0 || H <= 0 || rowBegin < 0 || rowEnd > H || rowBegin > rowEnd || !rgba) return RM_ERR_INVALID_ARGUMENT;
  uint64_t nEval = 0, nIter = 0, nHit = 0, nShade = 0, nShape = 0, nFbm9 = 0, nFbmd8 = 0;
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : nEval, nIter, nHit, nShade, nShape, nFbm9, nFbmd8)
  for (int y = rowBegin; y < rowEnd; y++) {
    Ctx c;
    c.cam = cam; c.objs = objs; c.numObjects = numObjects; c.lights = lights; c.numLights = numLights;
    c.g = *g; c.s = *s; c.nEval = c.nIter = c.nHit = c.nShade = c.nShape = 0; c.tex = res->textures; c.numTex = res->numTextures;
    c.res = res; c.W = W;
    rayPlanes(cam->invProjView, c.rayPlane);
    t_nFbm9 = t_nFbmd8 = 0;
    for (int x = 0; x < W; x++) {
      size_t o = ((size_t)(y - rowBegin) * W + x) * 4;
      shadePixel(&c, x, y, W, H, rgba + o, bright ? bright + o : NULL);
    }
    nEval += c.nEval; nIter += c.nIter; nHit += c.nHit; nShade += c.nShade; nShape += c.nShape; nFbm9 += t_nFbm9; nFbmd8 += t_nFbmd8;
  }
  if (counters) {
    counters->sceneEvals = nEval; counters->bulbIters = nIter; counters->hitPixels = nHit;
    counters->shadedPoints = nShade; counters->terrainEvals = nFbm9; counters->cloudEvals = nFbmd8; counters->shapeEvals = nShape;
  }
  return RM_OK;
}

void rmo_ltc_quantise(const float *table, uint8_t *out, int texels) {
  for (int i = 0; i < texels * 4; i++) {
    float v = table[i];
    v = (v < 0.0f) ? 0.0f : ((v > 1.0f) ? 1.0f : v);
    if (v != v) v = 0.0f;
    out[i] = (uint8_t)rintf(v * 255.0f);
  }
}

/* element-wise probes of the numeric contract (same fn ids as rm_probe_math) */
int rmo_probe_math(int fn, const float *x, const float *y, const float *z, float *out, int n) {
  for (int i = 0; i < n; i++) {
    switch (fn) {
      case RM_FN_SIN: out[i] = rm_sin(x[i]); break;
      case RM_FN_COS: out[i] = rm_cos(x[i]); break;
      case RM_FN_ACOS: out[i] = rm_acos(x[i]); break;
      case RM_FN_ATAN2: out[i] = rm_atan2(x[i], y[i]); break;
      case RM_FN_LOG2: out[i] = rm_log2(x[i]); break;
      case RM_FN_EXP2: out[i] = rm_exp2(x[i]); break;
      case RM_FN_POW: out[i] = rm_pow(x[i], y[i]); break;
      case RM_FN_SQRT: out[i] = rm_sqrt(x[i]); break;
      case RM_FN_DIV: out[i] = x[i] / y[i]; break;
      case RM_FN_PNOISE3: out[i] = pnoise(V3(x[i], y[i], z[i])); break;
      case RM_FN_ASIN: out[i] = rm_asin(x[i]); break;
      case RM_FN_Q16: out[i] = q16(x[i]); break;
      case RM_FN_SQRT_FAST: out[i] = rm_sqrt(x[i]); break;
      case RM_FN_DIVR: out[i] = rm_divr(x[i], y[i]); break;
      case RM_FN_RCP: out[i] = 1.0f / x[i]; break;
      case RM_FN_SMOOTHSTEP: out[i] = rm_smoothstep(x[i], y[i], z[i]); break;
      case RM_FN_MIN: out[i] = rm_min(x[i], y[i]); break;
      case RM_FN_MAX: out[i] = rm_max(x[i], y[i]); break;
      case RM_FN_FRACT: out[i] = rm_fract(x[i]); break;
      default: return RM_ERR_INVALID_ARGUMENT;
    }
  }
  return RM_OK;
}

int rmo_probe_sdscene(const RmObject *objs, int numObjects, const RmGlobals *g, const RmSettings *s,
                      const float *pts, float *out, int n) {
  Ctx c;
  c.cam = NULL; c.objs = objs; c.numObjects = numObjects; c.lights = NULL; c.numLights = 0;
  c.g = *g; c.s = *s; c.nEval = c.nIter = c.nHit = 0; c.tex = NULL; c.numTex = 0;
  for (int i = 0; i < n; i++) {
    SceneMin m = sdScene(&c, V3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    out[4 * i] = m.minD; out[4 * i + 1] = (float)m.minObjIdx; out[4 * i + 2] = m.trap.y; out[4 * i + 3] = m.trap.z;
  }
  return RM_OK;
}

/* procedural-layer probes: kind 0 cloudsFbm → (value, gradient), 1 cloudsMap → (density, gra.y, nnd, 0),
 * 2 sdTerrain(p.xz) → (height, slope flag, 0, 0) */
int rmo_probe_env(int kind, float iTime, const float *pts, float *out, int n) {
  Ctx c;
  memset(&c, 0, sizeof c);
  c.g.iTime = iTime;
  for (int i = 0; i < n; i++) {
    v3 p = V3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    v4 r = V4(0, 0, 0, 0);
    if (kind == 0) r = cloudsFbm(&c, p);
    else if (kind == 1) { float nnd; v4 m = cloudsMap(&c, p, &nnd); r = V4(m.x, m.z, nnd, 0.0f); }
    else if (kind == 2) { v2 t = sdTerrain(p.x, p.z); r = V4(t.x, t.y, 0.0f, 0.0f); }
    else return RM_ERR_INVALID_ARGUMENT;
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
  }
  return RM_OK;
}
/* kind 3: (seaMap, seaMapD, noiseW(p.xz), sea_octave(p.xz, 1)); 4: (getMoonColor(normalize(p)), noiseV(p));
 * 5: (sin(p.x), cos(p.x), hash(p.xy), 0).  `noise` (host pixels) is needed by kind 4. */
int rmo_probe_env2(int kind, float iTime, const RmTexture *noise, const float *pts, float *out, int n) {
  Ctx c;
  RmResources res;
  memset(&c, 0, sizeof c);
  memset(&res, 0, sizeof res);
  if (noise) res.noise = *noise;
  c.res = &res;
  c.g.iTime = iTime;
  for (int i = 0; i < n; i++) {
    v3 p = V3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    v4 r = V4(0, 0, 0, 0);
    if (kind == 3) r = V4(seaMap(&c, p, 3), seaMap(&c, p, 5), noiseW(p.x, p.z), sea_octave(p.x, p.z, 1.0f));
    else if (kind == 4) { v3 m = getMoonColor(&c, normalize3(p)); r = V4(m.x, m.y, m.z, noiseV(&c, p)); }
    else if (kind == 5) r = V4(rm_sin(p.x), rm_cos(p.x), hashSin2(p.x, p.y), 0.0f);
    else return RM_ERR_INVALID_ARGUMENT;
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
  }
  return RM_OK;
}

/* the 48 coefficients of rayPlanes ([triangle][near, far][P0, P1 − P0, P2 − P0][xyzw]) for tests */
void rmo_ray_planes(const RmCamera *cam, float *out) {
  v4 P[2][2][3];
  rayPlanes(cam->invProjView, P);
  for (int t = 0; t < 2; t++)
    for (int k = 0; k < 2; k++)
      for (int j = 0; j < 3; j++) {
        float *o = out + ((t * 2 + k) * 3 + j) * 4;
        o[0] = P[t][k][j].x; o[1] = P[t][k][j].y; o[2] = P[t][k][j].z; o[3] = P[t][k][j].w;
      }
}

/* bits of the contract's constants, for tests */
/* Exhaustive check behind the product's constant-divisor sequence (rm_math.hip.h, RM_DIVC): for EVERY binary32 mantissa
 * of x, q = x·fl(1/c) followed by one fma-residual correction equals the correctly rounded x / c.  Rounding commutes with
 * scaling by powers of two inside the normal range, so one binade of x covers them all.  Returns the mismatch count. */
long rmo_check_const_div(float c) {
  const float rc = 1.0f / c;
  long bad = 0;
  for (uint32_t m = 0; m < (1u << 23); m++) {
    const float x = rm_u2f(0x4b000000u | m);
    const float q0 = x * rc;
    const float q = rm_fma(rm_fma(-c, q0, x), rc, q0);
    if (q != x / c) bad++;
    const float xn = -x, qn0 = xn * rc;
    if (rm_fma(rm_fma(-c, qn0, xn), rc, qn0) != xn / c) bad++;
  }
  return bad;
}

uint32_t rmo_const_bits(int which) {
  switch (which) {
    case 0: return rm_f2u(RM_PI);
    case 1: return rm_f2u(RM_PIO2_HI);
    case 2: return rm_f2u(RM_PIO2_MID);
    case 3: return 0u; /* (the third Cody–Waite term, no longer part of the contract) */
    case 4: return rm_f2u(RM_2OPI);
    case 5: return rm_f2u(RM_LN2);
    case 6: return rm_f2u(RM_LOG2E);
    default: return 0;
  }
}
