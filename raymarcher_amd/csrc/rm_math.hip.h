// rm_math.hip.h — device implementation of the "rm_math" numeric contract for gfx950.
//
// GLSL leaves the precision of sin/cos/acos/atan/pow/exp2/log implementation-defined
// (the reference calls them all over resources/raymarch.frag, e.g. frag:787-793).  This project fixes
// one legal implementation built only from operations that a CPU reproduces bit for bit — v_add/v_mul/v_fma, correctly
// rounded reciprocal / division / sqrt, v_floor, v_fract, v_rndne, v_min/v_max with their documented NaN and zero rules,
// integer bit moves — so that a frame is reproducible; the CPU oracle implements the same contract separately and the
// GPU tests compare bits.  Compile with -ffp-contract=off and without fast-math: a fused multiply-add
// exists only where fma() is written.  Branch-free select forms are used so a wave never diverges
// inside a built-in.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rm {

#define RM_DEV __device__ __forceinline__

RM_DEV float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RM_DEV uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
RM_DEV float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
RM_DEV float fabs_(float x) { return __builtin_fabsf(x); }
// contract (oracle rm_min / rm_max): the hardware's rule — a signalling NaN operand comes back quieted, a quiet NaN operand
// is ignored, −0 < +0.  Inline asm: clang's fminf/fmaxf would put a canonicalising v_max in front (IEEE mode), and the
// compare + select form of GLSL's text is two instructions.
RM_DEV float min_(float x, float y) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
RM_DEV float max_(float x, float y) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
RM_DEV float clamp_(float x, float lo, float hi) { return min_(max_(x, lo), hi); }
RM_DEV float floor_(float x) { return __builtin_floorf(x); }
// contract: x − floor(x) kept below 1 (1 − 2^-24 where the difference rounds to 1.0) = v_fract_f32, for every input
// (rm_debug_check_math; scripts/microbench/fract_exhaustive.hip)
RM_DEV float fract_(float x) { return __builtin_amdgcn_fractf(x); }
RM_DEV float mod_(float x, float y) { return fma(-y, __builtin_floorf(x / y), x); }
RM_DEV float step_(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
RM_DEV float mix_(float x, float y, float a) { return fma(y, a, x * (1.0f - a)); }
RM_DEV float sqrt_(float x) { return __builtin_sqrtf(x); }
// 1.0f / y, bit for bit.  v_rcp_f32 followed by ONE Newton step is the correctly rounded reciprocal of every y with
// 2^-126 <= |y| < 2^126 — all 2·253·2^23 of them, checked exhaustively on the device against the IEEE quotient
// (scripts/microbench/rcp_exhaustive.hip; rm_debug_check_math in the test suite) — so the ten-instruction IEEE expansion runs
// only when some lane of the wave holds zero, a denormal, |y| >= 2^126, an infinity or a NaN (wave-uniform branch).
RM_DEV float rcp_(float y) {
  const float ay = fabs_(y);
  if (__builtin_expect(__ballot(!(ay >= 1.17549435e-38f) || !(ay < 8.50705917e37f)) != 0, 0)) return 1.0f / y;
  const float r = __builtin_amdgcn_rcpf(y);
  return fma(fma(-y, r, 1.0f), r, r);
}
// the fast form alone, for callers that have done the range check themselves (one check for several reciprocals)
RM_DEV float rcp_raw_(float y) {
  const float r = __builtin_amdgcn_rcpf(y);
  return fma(fma(-y, r, 1.0f), r, r);
}
// x / y of the contract's hot quotients (oracle rm_divr): x · RN(1/y).
RM_DEV float divr_(float x, float y) { return x * rcp_(y); }
// ... by a literal constant: the reciprocal is folded at compile time (the same RN(1/c) the oracle computes)
#define RM_DIVR_CONST(x, c) ((x) * (1.0f / (c)))
RM_DEV float smoothstep_(float e0, float e1, float x) {
  float t = clamp_(divr_(x - e0, e1 - e0), 0.0f, 1.0f);
  return (t * t) * fma(-2.0f, t, 3.0f);
}
// Correctly rounded sqrt for x == ±0, |x| >= 2^-96 (NaN for the negative ones), +inf and NaN: the refinement hipcc itself emits for
// sqrtf (v_sqrt_f32, then pick among s−1ulp, s, s+1ulp by the sign of the fma residuals) without the 2^32
// pre-scaling that only inputs below 2^-96 need.  Correct rounding is unique, so the bits equal sqrt_().
RM_DEV float sqrt_noscale_(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  float sd = u2f(f2u(s) - 1u), su = u2f(f2u(s) + 1u);
  float rd = fma(-sd, s, x), ru = fma(-su, s, x);
  s = (rd <= 0.0f) ? sd : s;
  return (ru > 0.0f) ? su : s;
}
// sqrt_() with a wave-uniform choice of the cheap form: the scaled form runs only if some lane holds a positive
// input below 2^-96 (practically never).
RM_DEV float sqrt_fast_(float x) {
  const float ax = fabs_(x);  // negative denormals too: v_sqrt_f32 takes them for −0, the IEEE result is NaN (found by the exhaustive check)
  if (__builtin_expect(__ballot((ax > 0.0f) && (ax < 1.262177448e-29f)) != 0, 0)) return sqrt_(x);
  return sqrt_noscale_(x);
}

// min_() with a literal operand or a source modifier folded into the instruction (the same contract).
#ifdef RM_X_BUILTIN_MIN
RM_DEV float hwmin_(float a, float b) { return __builtin_fminf(a, b); }
RM_DEV float hwmin1_(float a) { return __builtin_fminf(a, 1.0f); }
RM_DEV float hwmin_abs_(float a, float b) { return __builtin_fminf(a, __builtin_fabsf(b)); }
#else
RM_DEV float hwmin_(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
RM_DEV float hwmin1_(float a) { float r; asm("v_min_f32 %0, 1.0, %1" : "=v"(r) : "v"(a)); return r; }
RM_DEV float hwmin_abs_(float a, float b) { float r; asm("v_min_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b)); return r; }
#endif
// x / c for a compile-time constant c in three instructions instead of the eleven of the IEEE division expansion:
// q = x·fl(1/c), then one fma-residual correction.  For the constants used with it (7, 289, 3^k, 100, 200, 130, 2000)
// the result equals the correctly rounded quotient for EVERY binary32 mantissa of x — checked exhaustively by
// tests/test_oracle_math.py::test_constant_division_sequence_is_exact — as long as x/c and the residual stay in the
// normal range (|x| >= 2^-100 or x == 0; callers pass integers or differences of numbers near 1).
RM_DEV float divc_(float x, float c, float rc) { float q = x * rc; float r = fma(-c, q, x); return fma(r, rc, q); }
#define RM_DIVC(x, c) ::rm::divc_((x), (c), 1.0f / (c))

constexpr float kPi = 3.14159274f;          // 0x40490fdb
constexpr float kPio2 = 1.57079637f;        // 0x3fc90fdb
constexpr float k2oPi = 0.636619747f;       // 0x3f22f983
constexpr float kShifter = 12582912.0f;     // 1.5·2^23, 0x4b400000
constexpr float kPio2Mid = -4.37113883e-08f;  // 0xb33bbd2e

// sin and cos of one argument sharing the Cody–Waite reduction (bits equal the separate calls).
RM_DEV void sincos_(float x, float &sn, float &cs) {
  bool ok = fabs_(x) < 4194304.0f;
  // k = nearest integer to x·2/pi by the shifter 1.5·2^23 (oracle rm__reduce_pio2): one fma, one subtraction; the two low
  // bits of t's significand are k mod 4
  float t = fma(x, k2oPi, kShifter);
  float k = t - kShifter;
  float r = fma(-k, kPio2, x);
  r = fma(-k, kPio2Mid, r);
  uint32_t q = ok ? f2u(t) : 0u;
  r = ok ? r : 0.0f;
  float z = r * r;
  float s = fma(z, -1.950213627e-04f, 8.332063444e-03f);
  s = fma(z, s, -1.666665375e-01f);
  float sp = fma(s * z, r, r);
  float c = fma(z, 2.441812649e-05f, -1.388718490e-03f);
  c = fma(z, c, 4.166664183e-02f);
  float cp = fma(z, c * z, fma(z, -0.5f, 1.0f));
  float sv = (q & 1) ? cp : sp;
  float cv = (q & 1) ? sp : cp;
  // sign flips as integer xor of the sign bit (exact negation, also of zero)
  sn = u2f(f2u(sv) ^ ((uint32_t)(q & 2) << 30));
  cs = u2f(f2u(cv) ^ ((uint32_t)((q + 1) & 2) << 30));
}
// Same bits as sincos_ for finite |x| < 2^22 (the guard of the contract is dead there); used by the Mandelbulb
// iteration, whose angles are power·acos(·) and power·atan(·,·): finite and bounded by power·pi.
RM_DEV void sincos_inrange_(float x, float &sn, float &cs) {
  float t = fma(x, k2oPi, kShifter);
  float k = t - kShifter;
  float r = fma(-k, kPio2, x);
  r = fma(-k, kPio2Mid, r);
  uint32_t q = f2u(t);
  float z = r * r;
  float s = fma(z, -1.950213627e-04f, 8.332063444e-03f);
  s = fma(z, s, -1.666665375e-01f);
  float sp = fma(s * z, r, r);
  float c = fma(z, 2.441812649e-05f, -1.388718490e-03f);
  c = fma(z, c, 4.166664183e-02f);
  float cp = fma(z, c * z, fma(z, -0.5f, 1.0f));
  float sv = (q & 1) ? cp : sp;
  float cv = (q & 1) ? sp : cp;
  sn = u2f(f2u(sv) ^ ((uint32_t)(q & 2) << 30));
  cs = u2f(f2u(cv) ^ ((uint32_t)((q + 1) & 2) << 30));
}
RM_DEV float sin_(float x) { float s, c; sincos_(x, s, c); return s; }
RM_DEV float cos_(float x) { float s, c; sincos_(x, s, c); return c; }

RM_DEV float asin_p(float z) {
  float p = fma(z, 4.277068377e-02f, 2.384351753e-02f);
  p = fma(z, p, 4.553402960e-02f);
  p = fma(z, p, 7.494829595e-02f);
  p = fma(z, p, 1.666676253e-01f);
  return p;
}
// acos(x) = sqrt(1 − |x|)·P(|x|) for x > 0, pi − that otherwise: one polynomial, one square root, no
// branch on |x|.  1 − |x| is 0, >= 2^-24, negative or NaN — never a tiny positive number — so the unscaled correctly
// rounded square root applies; its NaN for |x| > 1 is discarded by the clamp.
RM_DEV float acos_(float x) {
  // ax = min(1, |x|) in ONE instruction (the contract's min: a quiet NaN gives 1): the clamp of the domain and |x| in a
  // register of its own — folded into the fma operands as a source modifier it would force the VOP3 encoding, which takes
  // no literal, and the seven coefficients would each occupy a scalar register across the march loops
  float ax;
  asm("v_min_f32 %0, 1.0, |%1|" : "=v"(ax) : "v"(x));
  float p = fma(ax, -1.253449009e-03f, 6.638590246e-03f);
  p = fma(ax, p, -1.704506390e-02f);
  p = fma(ax, p, 3.086272627e-02f);
  p = fma(ax, p, -5.016417801e-02f);
  p = fma(ax, p, 8.897730708e-02f);
  p = fma(ax, p, -2.145987004e-01f);
  p = fma(ax, p, 1.570796251e+00f);
  float v = sqrt_noscale_(1.0f - ax) * p;  // 1 − ax is 0 or >= 2^-24; |x| >= 1 or NaN: sqrt(0)·P(1) = 0
  return (x > 0.0f) ? v : (kPi - v);
}

RM_DEV float asin_(float x) {
  float ax = fabs_(x);
  bool small = ax <= 0.5f;
  float z = small ? (x * x) : ((1.0f - ax) * 0.5f);
  float s = small ? x : sqrt_(z);
  float as = fma(s * z, asin_p(z), s);
  float big = kPio2 - 2.0f * as;
  big = (x < 0.0f) ? -big : big;
  float r = small ? as : big;
  float edge = (x > 0.0f) ? kPio2 : -kPio2;
  return (small || ax < 1.0f) ? r : edge;
}

RM_DEV float atan_p(float s) {
  float p = fma(s, 2.920665313e-03f, -1.636782475e-02f);
  p = fma(s, p, 4.321170226e-02f);
  p = fma(s, p, -7.552202046e-02f);
  p = fma(s, p, 1.066599935e-01f);
  p = fma(s, p, -1.421105415e-01f);
  p = fma(s, p, 1.999377310e-01f);
  p = fma(s, p, -3.333315253e-01f);
  return p;
}
// RAW = true: the caller has checked, for the whole wave, that max(|x|, |y|) lies in the reciprocal's fast range
// (2^-126 <= · < 2^126): the bare v_rcp_f32 + Newton form, no guard of its own.  Same bits as RAW = false there.
template <bool RAW = false>
RM_DEV float atan2_(float y, float x) {
  float ax = fabs_(x), ay = fabs_(y);
  float mx, mn;  // the contract's max / min of the magnitudes, |·| folded into the instructions
  asm("v_max_f32 %0, |%1|, |%2|" : "=v"(mx) : "v"(x), "v"(y));
  asm("v_min_f32 %0, |%1|, |%2|" : "=v"(mn) : "v"(x), "v"(y));
  float t = RAW ? mn * rcp_raw_(mx) : divr_(mn, mx);
  // contract: a NaN quotient (0·inf, inf·0, NaN operand) or one that overflows (denormal operands) is 1, and 0 if mx == 0.
  // mn <= mx, so any other quotient is <= 1 and v_min_f32(t, 1) — which ignores a NaN operand — is t itself.
  t = hwmin1_(t);
  t = (mx == 0.0f) ? 0.0f : t;
  float s = t * t;
  float a = fma(t * s, atan_p(s), t);
  a = (ay > ax) ? (kPio2 - a) : a;
  a = (x < 0.0f) ? (kPi - a) : a;
  return u2f((f2u(a) & 0x7fffffffu) | (f2u(y) & 0x80000000u));
}

RM_DEV float log2_(float x) {
  uint32_t ux = f2u(x) - 0x3f3504f3u;
  int32_t e = (int32_t)ux >> 23;
  float m = u2f((ux & 0x007fffffu) + 0x3f3504f3u);
  float f = m - 1.0f;
  float l = fma(f, 1.258333027e-01f, -2.072679251e-01f);
  l = fma(f, l, 2.157161385e-01f);
  l = fma(f, l, -2.389451116e-01f);
  l = fma(f, l, 2.879162133e-01f);
  l = fma(f, l, -3.607036769e-01f);
  l = fma(f, l, 4.809106290e-01f);
  l = fma(f, l, -7.213473320e-01f);
  l = fma(f, l, 1.442695022e+00f);
  float r = fma(f, l, (float)e);
  return (x >= 1.17549435e-38f) ? r : -__builtin_inff();
}
RM_DEV float exp2_(float x) {
  float n = __builtin_rintf(x);
  float f = x - n;
  float p = fma(f, 1.535335905e-04f, 1.339887502e-03f);
  p = fma(f, p, 9.618436918e-03f);
  p = fma(f, p, 5.550332367e-02f);
  p = fma(f, p, 2.402264774e-01f);
  p = fma(f, p, 6.931471825e-01f);
  p = fma(f, p, 1.0f);
  // 2^n without a float→int conversion: n + 1.5·2^23 holds the integer n in its low mantissa bits (exact for
  // |n| < 2^22), so ((bits + 127) << 23) = (bits << 23) + 0x3f800000 is the exponent field of 2^n for every n the
  // contract does not override below (−125 < x < 128); for other x the value is replaced.
  uint32_t scale = (f2u(n + 12582912.0f) << 23) + 0x3f800000u;
  float r = p * u2f(scale);
  r = (x >= 128.0f) ? __builtin_inff() : r;
  return (x > -125.0f) ? r : 0.0f;
}
// pow(x, y): integer and half-integer exponents with |y| <= 128 by binary exponentiation (·sqrt(x) for the half,
// reciprocal for y < 0), everything else exp2(y·log2(x)) — the contract of DESIGN.md §3.  The
// exponents on the hot path (power, (power−1)/2, shininess) are wave-uniform: the loop is scalar-controlled and the
// classification of y is loop-invariant in the Mandelbulb iteration.
struct PowPlan { int n; bool fast, half, neg; };  // how pow(·, y) is evaluated for one exponent y
RM_DEV PowPlan powPlan(float y) {
  float ay = fabs_(y), two = ay + ay;
  PowPlan pl;
  pl.fast = (two <= 256.0f) && (two == floor_(two));
  pl.n = pl.fast ? (int)ay : 0;
  pl.half = ay != (float)pl.n;
  pl.neg = y < 0.0f;
  return pl;
}
// The same plan for a WAVE-UNIFORM exponent (a uniform such as `power`): held in scalar registers, so the bit loop
// and the branches of powApply are scalar control flow.  Hoist it out of loops that call pow with a fixed exponent.
RM_DEV PowPlan powPlanUniform(float y) {
  PowPlan pl = powPlan(y);
  pl.n = __builtin_amdgcn_readfirstlane(pl.n);
  pl.fast = __builtin_amdgcn_readfirstlane((int)pl.fast) != 0;
  pl.half = __builtin_amdgcn_readfirstlane((int)pl.half) != 0;
  pl.neg = __builtin_amdgcn_readfirstlane((int)pl.neg) != 0;
  return pl;
}
RM_DEV float powApply(float x, float y, const PowPlan &pl) {
  if (pl.fast) {
    float p = 1.0f, b = x;
    for (int e = pl.n; e != 0; e >>= 1) {
      if (e & 1) p = p * b;
      if (e > 1) b = b * b;
    }
    if (pl.half) p = p * sqrt_fast_(x);
    return pl.neg ? rcp_(p) : p;
  }
  return exp2_(y * log2_(x));
}
RM_DEV float pow_(float x, float y) { return powApply(x, y, powPlan(y)); }
constexpr float kLn2 = 0.693147182f;   // 0x3f317218
constexpr float kLog2e = 1.44269502f;  // 0x3fb8aa3b
RM_DEV float log_(float x) { return log2_(x) * kLn2; }
RM_DEV float exp_(float x) { return exp2_(x * kLog2e); }

}  // namespace rm
