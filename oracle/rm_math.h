/*
 * oracle/rm_math.h — TEST INFRASTRUCTURE (CPU oracle).  Never linked into the product library.
 *
 * The "rm_math" numeric contract restated in plain C99.  GLSL leaves the precision
 * of its built-ins (sin, cos, acos, atan, pow, exp2, log, …) implementation-defined, so two GPUs
 * running resources/raymarch.frag do not agree bit-for-bit, and the Mandelbulb/Menger silhouettes
 * amplify 1-ulp differences into different hit/miss decisions.  To make "identical results"
 * checkable, this project FIXES one legal implementation of every built-in, made only of binary32
 * operations that x86-64 and gfx950 both compute to the same bits: +, −, ×, ÷, reciprocal, sqrt, fma (correctly rounded),
 * floor, rint, fract, min / max (with the NaN and zero rules written out below), compares, bit moves.  The HIP kernels
 * implement the same contract independently (raymarcher_amd/csrc/rm_math.hip.h); tests require bit equality.
 *
 * Rules of the contract
 *   - every operation is binary32, round-to-nearest-even, no flush of results;
 *   - a*b+c is fused ONLY where written as rm_fma(); build with -ffp-contract=off;
 *   - min / max follow the rule of the hardware instruction (stated at rm_min), fract stays below 1, the quotients on
 *     hot paths are x · RN(1/y) (rm_divr): each is one legal reading of GLSL's text and one GPU instruction (or three);
 *   - out-of-domain inputs give the documented finite/inf value, never "undefined".
 * Polynomial coefficients come from oracle/tools/fit_coeffs.py (max approximation error in the
 * comments); measured end-to-end accuracy is asserted in tests/test_oracle_math.py.
 */
#ifndef RM_ORACLE_MATH_H
#define RM_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float rm_fma(float a, float b, float c) { return fmaf(a, b, c); }
static inline uint32_t rm_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float rm_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* min / max: GLSL leaves them undefined for NaN operands; the contract takes the hardware's rule (v_min_f32 / v_max_f32 in
 * IEEE mode, one instruction instead of compare + select): a signalling NaN operand is returned quieted (first operand
 * first); otherwise a quiet NaN operand is ignored (the other operand is returned; the first one if both are NaN); −0 orders
 * below +0; otherwise the smaller / larger operand.  (Observed on gfx950 for every pair of a set of special values, and held
 * by tests/test_gpu_parity.py::test_min_max_fract_…) */
static inline int rm__is_snan(float f) { uint32_t u = rm_f2u(f); return (u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu) != 0u && !(u & 0x00400000u); }
static inline float rm__quiet(float f) { return rm_u2f(rm_f2u(f) | 0x00400000u); }
static inline float rm_min(float x, float y) {
  if (rm__is_snan(x)) return rm__quiet(x);
  if (rm__is_snan(y)) return rm__quiet(y);
  if (x != x) return (y != y) ? x : y;
  if (y != y) return x;
  if (x == 0.0f && y == 0.0f) return (rm_f2u(x) >> 31) ? x : y;
  return (x < y) ? x : y;
}
static inline float rm_max(float x, float y) {
  if (rm__is_snan(x)) return rm__quiet(x);
  if (rm__is_snan(y)) return rm__quiet(y);
  if (x != x) return (y != y) ? x : y;
  if (y != y) return x;
  if (x == 0.0f && y == 0.0f) return (rm_f2u(x) >> 31) ? y : x;
  return (x > y) ? x : y;
}
static inline float rm_clamp(float x, float lo, float hi) { return rm_min(rm_max(x, lo), hi); }
static inline float rm_abs(float x) { return fabsf(x); }
static inline float rm_floor(float x) { return floorf(x); }
/* GLSL fract(x) = x − floor(x) (may round to 1.0 for tiny negative x; kept as specified). */
/* GLSL fract(x) = x − floor(x), kept inside [0, 1): for a tiny negative x the difference rounds to 1.0, which is returned as
 * 1 − 2^-24 — what the hardware instruction does (v_fract_f32 equals this for every one of the 2^32 inputs: checked on the
 * device, rm_debug_check_math) and what the hash-type uses of fract (noise lattices) expect.  inf, NaN → NaN. */
static inline float rm_fract(float x) { float f = x - floorf(x); return (f >= 1.0f) ? 0.99999994f : f; }
/* GLSL mod(x,y) = x − y·floor(x/y). */
static inline float rm_mod(float x, float y) { return rm_fma(-y, floorf(x / y), x); }
/* mod(x, 2^k) through fract: y·fract(x/y).  x/y and the multiplication by y are exact, so this IS x − y·floor(x/y) — except
 * for a tiny negative x, where that expression rounds to y itself and this one stays below it (fract's rule).  The Menger
 * sponge's mod(p·s, 2): multiply, v_fract_f32, and the ·2 − 1 that follows in one fma. */
static inline float rm_mod_pow2(float x, float y) { return y * rm_fract(x * (1.0f / y)); }
static inline float rm_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
/* GLSL step(edge,x) = x < edge ? 0 : 1. */
static inline float rm_step(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
/* GLSL mix(x,y,a) = x·(1−a) + y·a, the second product fused. */
static inline float rm_mix(float x, float y, float a) { return rm_fma(y, a, x * (1.0f - a)); }
/* x / y as x · RN(1/y): the correctly rounded reciprocal, then one multiplication (error < 1.5 ulp instead of 1/2; GLSL allows
 * 2.5).  Used for the quotients on hot paths — inside the Mandelbulb iteration (acos argument, atan ratio) and its distance estimate,
 * smoothstep, the soft-shadow and terrain-shadow penumbra, the cone's projection, the terrain / cloud scalings by constants,
 * the cube-map projection: on the GPU RN(1/y) is three instructions (v_rcp_f32 +
 * one Newton step, exact over the whole normal range — checked for every input) or a compile-time constant, the IEEE
 * quotient ten.  Same special values as x / y except where 1/y over- or underflows (|y| < 2^-128 or > 2^126). */
static inline float rm_divr(float x, float y) { return x * (1.0f / y); }
/* GLSL smoothstep: t = clamp((x−e0)/(e1−e0),0,1); t·t·(3−2t). */
static inline float rm_smoothstep(float e0, float e1, float x) {
  float t = rm_clamp(rm_divr(x - e0, e1 - e0), 0.0f, 1.0f);
  return (t * t) * rm_fma(-2.0f, t, 3.0f);
}
static inline float rm_sqrt(float x) { return sqrtf(x); }

/* ---- sin / cos ------------------------------------------------------------------------------ */
#define RM_PI 3.14159274f       /* fl(pi)   0x40490fdb */
#define RM_PIO2 1.57079637f     /* fl(pi/2) 0x3fc90fdb */
#define RM_2OPI 0.636619747f    /* fl(2/pi) 0x3f22f983 */
/* pi/2 = HI + MID − 1.7e-15, each a full binary32 (two-term Cody–Waite with fma: the neglected tail times the
 * largest k of the contract range, 2.7e6, is 4.6e-9 — far below the 1.5e-7 the polynomial kernels deliver). */
#define RM_PIO2_HI 1.57079637f          /* 0x3fc90fdb */
#define RM_PIO2_MID (-4.37113883e-08f)  /* 0xb33bbd2e */

/* r = x − k·pi/2 with k = the integer nearest to x·2/pi; valid contract range |x| < 2^22, outside it (and NaN) the
 * reduction returns r = 0, q = 0 (sin → 0, cos → 1).  k by the shifter trick: t = fma(x, 2/pi, 1.5·2^23) lies in [2^23, 2^24),
 * where the unit in the last place is 1, so the fma's single rounding IS round-to-nearest-even of x·2/pi; k = t − 1.5·2^23
 * is exact and the two low bits of t's significand are k mod 4 (also for negative k).  One fma and one subtraction
 * instead of multiply, rint and convert. */
#define RM_SHIFTER 12582912.0f /* 1.5·2^23, 0x4b400000 */
static inline float rm__reduce_pio2(float x, int *q) {
  if (!(fabsf(x) < 4194304.0f)) { *q = 0; return 0.0f; }
  float t = rm_fma(x, RM_2OPI, RM_SHIFTER);
  float k = t - RM_SHIFTER;
  float r = rm_fma(-k, RM_PIO2_HI, x);
  r = rm_fma(-k, RM_PIO2_MID, r);
  *q = (int)(rm_f2u(t) & 3u);
  return r;
}
/* sin(r), |r| <= pi/4 : r + r·z·S(z)   (max rel approx err 6.8e-9) */
static inline float rm__sin_poly(float r) {
  float z = r * r;
  float s = rm_fma(z, -1.950213627e-04f, 8.332063444e-03f);
  s = rm_fma(z, s, -1.666665375e-01f);
  return rm_fma(s * z, r, r);
}
/* cos(r), |r| <= pi/4 : 1 − z/2 + z²·C(z)  (max rel approx err 7.0e-10) */
static inline float rm__cos_poly(float r) {
  float z = r * r;
  float c = rm_fma(z, 2.441812649e-05f, -1.388718490e-03f);
  c = rm_fma(z, c, 4.166664183e-02f);
  return rm_fma(z, c * z, rm_fma(z, -0.5f, 1.0f));
}
static inline float rm_sin(float x) {
  int q; float r = rm__reduce_pio2(x, &q);
  float v = (q & 1) ? rm__cos_poly(r) : rm__sin_poly(r);
  return (q & 2) ? -v : v;
}
static inline float rm_cos(float x) {
  int q; float r = rm__reduce_pio2(x, &q);
  float v = (q & 1) ? rm__sin_poly(r) : rm__cos_poly(r);
  return ((q + 1) & 2) ? -v : v;
}

/* ---- acos ----------------------------------------------------------------------------------- */
/* asin(x) = x + x·z·P(z), z = x² ∈ [0, 0.25]  (max rel approx err 6.3e-9) */
static inline float rm__asin_p(float z) {
  float p = rm_fma(z, 4.277068377e-02f, 2.384351753e-02f);
  p = rm_fma(z, p, 4.553402960e-02f);
  p = rm_fma(z, p, 7.494829595e-02f);
  p = rm_fma(z, p, 1.666676253e-01f);
  return p;
}
/* acos(x) = sqrt(1 − |x|)·P(|x|) on [0, 1) — the form of Abramowitz & Stegun 4.4.46 with a degree-7 minimax P fitted
 * by oracle/tools/fit_coeffs.py (max rel approx err 5.6e-8; measured < 2.9 ulp) — and pi − that for x <= 0.
 * |x| >= 1 or NaN clamps to acos(±1): x > 0 → 0, otherwise pi.  One polynomial, one square root, no branch on |x|. */
static inline float rm_acos(float x) {
  float ax = rm_min(1.0f, fabsf(x)); /* the clamp of the domain; a quiet NaN gives 1 (rm_min), i.e. acos(NaN) = pi */
  float p = rm_fma(ax, -1.253449009e-03f, 6.638590246e-03f);
  p = rm_fma(ax, p, -1.704506390e-02f);
  p = rm_fma(ax, p, 3.086272627e-02f);
  p = rm_fma(ax, p, -5.016417801e-02f);
  p = rm_fma(ax, p, 8.897730708e-02f);
  p = rm_fma(ax, p, -2.145987004e-01f);
  p = rm_fma(ax, p, 1.570796251e+00f);
  float v = sqrtf(1.0f - ax) * p; /* |x| >= 1: sqrt(0)·P(1) = 0 */
  return (x > 0.0f) ? v : (RM_PI - v);
}

/* asin(x) = sign(x)·(pi/2 − acos-branch); |x| >= 1 or NaN clamps: x > 0 → pi/2, otherwise −pi/2. */
static inline float rm_asin(float x) {
  float ax = fabsf(x);
  if (ax <= 0.5f) {
    float z = x * x;
    return rm_fma(x * z, rm__asin_p(z), x);
  } else if (ax < 1.0f) {
    float z = (1.0f - ax) * 0.5f;
    float s = sqrtf(z);
    float as = rm_fma(s * z, rm__asin_p(z), s);
    float r = RM_PIO2 - 2.0f * as;
    return (x < 0.0f) ? -r : r;
  }
  return (x > 0.0f) ? RM_PIO2 : -RM_PIO2;
}

/* ---- atan(y, x) ----------------------------------------------------------------------------- */
/* atan(t) = t + t·s·P(s), s = t², t ∈ [0,1]  (max rel approx err 2.1e-8) */
static inline float rm__atan_p(float s) {
  float p = rm_fma(s, 2.920665313e-03f, -1.636782475e-02f);
  p = rm_fma(s, p, 4.321170226e-02f);
  p = rm_fma(s, p, -7.552202046e-02f);
  p = rm_fma(s, p, 1.066599935e-01f);
  p = rm_fma(s, p, -1.421105415e-01f);
  p = rm_fma(s, p, 1.999377310e-01f);
  p = rm_fma(s, p, -3.333315253e-01f);
  return p;
}

/* GLSL atan(y,x).  0/0 → 0; inf/inf and NaN ratios are treated as t = 1, as is a ratio that overflows (denormal x and y).  x is "negative" iff
 * x < 0 (so −0 counts as +0); the result carries the sign bit of y. */
static inline float rm_atan2(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = rm_max(ax, ay);
  float mn = rm_min(ax, ay);
  float t = rm_divr(mn, mx);
  if (!(t <= 1.0f)) t = 1.0f; /* NaN, +inf */
  if (mx == 0.0f) t = 0.0f;
  float s = t * t;
  float a = rm_fma(t * s, rm__atan_p(s), t);
  if (ay > ax) a = RM_PIO2 - a;
  if (x < 0.0f) a = RM_PI - a;
  return copysignf(a, y);
}

/* ---- log2 / exp2 / pow / log / exp ------------------------------------------------------------ */
/* log2(x): x < FLT_MIN (zero, denormal, negative) or NaN → −inf; +inf → 128 (bit pattern path). */
static inline float rm_log2(float x) {
  if (!(x >= 1.17549435e-38f)) return -INFINITY;
  uint32_t ux = rm_f2u(x) - 0x3f3504f3u;                 /* bits of sqrt(1/2) */
  int32_t e = (int32_t)ux >> 23;                         /* arithmetic shift */
  float m = rm_u2f((ux & 0x007fffffu) + 0x3f3504f3u);    /* m ∈ [sqrt(.5), sqrt(2)) */
  float f = m - 1.0f;
  /* log2(1+f) = f·L(f)  (max rel approx err 4.3e-8) */
  float l = rm_fma(f, 1.258333027e-01f, -2.072679251e-01f);
  l = rm_fma(f, l, 2.157161385e-01f);
  l = rm_fma(f, l, -2.389451116e-01f);
  l = rm_fma(f, l, 2.879162133e-01f);
  l = rm_fma(f, l, -3.607036769e-01f);
  l = rm_fma(f, l, 4.809106290e-01f);
  l = rm_fma(f, l, -7.213473320e-01f);
  l = rm_fma(f, l, 1.442695022e+00f);
  return rm_fma(f, l, (float)e);
}
/* exp2(x): x <= −125 or NaN → 0; x >= 128 → +inf. */
static inline float rm_exp2(float x) {
  if (!(x > -125.0f)) return 0.0f;
  if (x >= 128.0f) return INFINITY;
  float n = rintf(x);
  float f = x - n;
  /* 2^f = 1 + f·E(f), |f| <= 0.5  (max rel approx err 1.6e-8) */
  float p = rm_fma(f, 1.535335905e-04f, 1.339887502e-03f);
  p = rm_fma(f, p, 9.618436918e-03f);
  p = rm_fma(f, p, 5.550332367e-02f);
  p = rm_fma(f, p, 2.402264774e-01f);
  p = rm_fma(f, p, 6.931471825e-01f);
  p = rm_fma(f, p, 1.0f);
  return p * rm_u2f((uint32_t)((int32_t)n + 127) << 23);
}
/* GLSL pow(x,y).  General case: exp2(y·log2(x)), the form the GLSL spec itself names.  Integer and half-integer
 * exponents with |y| <= 128 (the shader's pow(r, 8), pow(m, 3.5), pow(·, shininess), pow(t, 3) …) are evaluated the way
 * libm implementations do, by binary exponentiation (least-significant bit first: p *= b on a set bit, b *= b while bits
 * remain), times sqrt(x) for a half-integer, reciprocal for y < 0 — at most 14 roundings; measured against float64 it is
 * as accurate as the exp2/log2 route or better (max 4.8 vs 6.8 ulp at y = 8, 69 vs 93 at y = 100) at a fraction of the cost.  Consequences outside GLSL's domain (x < 0 is undefined there): an
 * integer exponent keeps the sign of the product, pow(x, 0) = 1 for every x. */
static inline float rm_pow(float x, float y) {
  float ay = fabsf(y), two = ay + ay;
  if (two <= 256.0f && two == floorf(two)) {
    int n = (int)ay;
    float p = 1.0f, b = x;
    for (int e = n; e != 0; e >>= 1) {
      if (e & 1) p = p * b;
      if (e > 1) b = b * b;
    }
    if (ay != (float)n) p = p * sqrtf(x);
    return (y < 0.0f) ? 1.0f / p : p;
  }
  return rm_exp2(y * rm_log2(x));
}
#define RM_LN2 0.693147182f    /* 0x3f317218 */
#define RM_LOG2E 1.44269502f   /* 0x3fb8aa3b */
static inline float rm_log(float x) { return rm_log2(x) * RM_LN2; }
static inline float rm_exp(float x) { return rm_exp2(x * RM_LOG2E); }

#endif /* RM_ORACLE_MATH_H */
