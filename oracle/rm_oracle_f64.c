/*
 * oracle/rm_oracle_f64.c — TEST INFRASTRUCTURE.  The ARBITER: the very same restatement (rm_oracle.c, same operation
 * order, same constants) compiled with every `float` a binary64 and every built-in taken from libm's double functions
 * instead of the rm_math polynomials.
 *
 * Why: the reference has no golden vectors for the shader, and the reference shader run on SwiftShader differs from the
 * binary32 oracle on chaotic pixels (Mandelbulb, Menger: 4-tap normals over 2.9e-4, pow(·,100)).  Which of the two is
 * "off"?  Neither knows the true value of the shader's real-number semantics; this build approximates it 2^29 times
 * better than either, so |oracle32 − f64| and |SwiftShader − f64| can be compared pixel by pixel
 * (tests/test_oracle_vs_glsl.py::test_oracle32_is_as_close_to_the_f64_arbiter_as_the_reference_on_swiftshader).
 *
 * One part stays binary32 on purpose: the Perlin lattice gradients (pgrad1 in rm_oracle.c) contain a comparison that is
 * an exact tie in real arithmetic for 7 of 49 hash classes — the shader's result there is DEFINED by binary32 rounding
 * (binary32 oracle and SwiftShader agree on every such pixel; a binary64 evaluation would pick other gradients and
 * "disagree" with both on 10–30 % of bump-mapped pixels for no meaningful reason).
 *
 * Mechanics: system headers first; then `#define float double`; then the public ABI header (so RmObject … hold doubles —
 * tests convert the binary32 tables field by field) and a libm-backed rm_math; then rm_oracle.c itself.
 * Not the contract, not bit-comparable with anything; only ever compared with tolerances.
 */
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef float rmo_f32; /* the one place that stays binary32: the tie decision of the Perlin lattice gradients (pgrad1) */
#define RMO_F32_DEFINED
#define float double
#define RM_ORACLE_MATH_H /* keep the binary32 contract out of this build */

static inline double rm_fma(double a, double b, double c) { return fma(a, b, c); }
static inline uint32_t rm_f2u(double f) { float g = (float)0; (void)f; (void)g; return 0u; } /* bit tricks: unused here */
static inline double rm_u2f(uint32_t u) { (void)u; return 0.0; }
static inline double rm_min(double x, double y) { return fmin(x, y); } /* a NaN operand is ignored, as in the contract */
static inline double rm_max(double x, double y) { return fmax(x, y); }
static inline double rm_clamp(double x, double lo, double hi) { return rm_min(rm_max(x, lo), hi); }
static inline double rm_abs(double x) { return fabs(x); }
static inline double rm_floor(double x) { return floor(x); }
static inline double rm_fract(double x) { double f = x - floor(x); return (f >= 1.0) ? 0.99999999999999989 : f; }
static inline double rm_mod(double x, double y) { return x - y * floor(x / y); }
static inline double rm_mod_pow2(double x, double y) { return y * rm_fract(x / y); }
static inline double rm_sign(double x) { return (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : 0.0); }
static inline double rm_step(double edge, double x) { return (x < edge) ? 0.0 : 1.0; }
static inline double rm_mix(double x, double y, double a) { return x * (1.0 - a) + y * a; }
static inline double rm_smoothstep(double e0, double e1, double x) {
  double t = rm_clamp((x - e0) / (e1 - e0), 0.0, 1.0);
  return (t * t) * (3.0 - 2.0 * t);
}
static inline double rm_sqrt(double x) { return sqrt(x); }
#define RM_PI 3.14159265358979323846
#define RM_PIO2 1.57079632679489661923
#define RM_PIO2_HI RM_PIO2
#define RM_PIO2_MID 0.0
#define RM_2OPI 0.63661977236758134308
#define RM_LN2 0.69314718055994530942
#define RM_LOG2E 1.44269504088896340736
static inline double rm_sin(double x) { return sin(x); }
static inline double rm_cos(double x) { return cos(x); }
/* the documented out-of-domain values of the contract, so that the same pixels are defined */
static inline double rm_acos(double x) { return (fabs(x) < 1.0) ? acos(x) : ((x > 0.0) ? 0.0 : RM_PI); }
static inline double rm_asin(double x) { return (fabs(x) < 1.0) ? asin(x) : ((x > 0.0) ? RM_PIO2 : -RM_PIO2); }
static inline double rm_atan2(double y, double x) { return atan2(y, x); }
static inline double rm_divr(double x, double y) { return x / y; }
static inline double rm_log2(double x) { return (x > 0.0) ? log2(x) : -INFINITY; }
static inline double rm_exp2(double x) { return exp2(x); }
static inline double rm_pow(double x, double y) { return pow(x, y); }
static inline double rm_log(double x) { return (x > 0.0) ? log(x) : -INFINITY; }
static inline double rm_exp(double x) { return exp(x); }

#include "rm_oracle.c"
