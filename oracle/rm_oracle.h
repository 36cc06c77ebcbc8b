/*
 * oracle/rm_oracle.h — TEST INFRASTRUCTURE.  API of the CPU oracle (see rm_oracle.c).
 * Uses the product's public PODs (include/raymarcher_amd.h) so tests feed both sides the same bytes.
 * Host buffers only; nothing here touches a GPU.
 */
#ifndef RM_ORACLE_H
#define RM_ORACLE_H
#include "../include/raymarcher_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Render rows [rowBegin,rowEnd) into host buffers (row 0 = bottom). `threads` OpenMP threads. */
int rmo_render(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
               const RmGlobals *g, const RmSettings *s, int W, int H, int rowBegin, int rowEnd, float *rgba,
               float *bright, RmCounters *counters, int threads);
/* Same with object textures (HOST pixel pointers in RmTexture.pixels). */
int rmo_render_tex(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                   const RmGlobals *g, const RmSettings *s, const RmTexture *textures, int numTextures, int W, int H,
                   int rowBegin, int rowEnd, float *rgba, float *bright, RmCounters *counters, int threads);
/* Same with every sampler (HOST pointers throughout RmResources); res may be NULL. */
int rmo_render_res(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                   const RmGlobals *g, const RmSettings *s, const RmResources *res, int W, int H, int rowBegin,
                   int rowEnd, float *rgba, float *bright, RmCounters *counters, int threads);
void rmo_ltc_quantise(const float *table, uint8_t *out, int texels);
/* Post passes on host buffers (see rm_post_process). */
int rmo_post_process(const float *frag, const float *bright, float *out, int W, int H, const RmPostSettings *ps);
int rmo_probe_math(int fn, const float *x, const float *y, const float *z, float *out, int n);
int rmo_probe_sdscene(const RmObject *objs, int numObjects, const RmGlobals *g, const RmSettings *s,
                      const float *pts, float *out, int n);
int rmo_probe_env(int kind, float iTime, const float *pts, float *out, int n);
int rmo_probe_env2(int kind, float iTime, const RmTexture *noise, const float *pts, float *out, int n);
uint32_t rmo_const_bits(int which);
/* Mismatches of the three-instruction constant-divisor sequence against IEEE division over all mantissas (0 = exact). */
long rmo_check_const_div(float c);
void rmo_ray_planes(const RmCamera *cam, float *out48);
#ifdef __cplusplus
}
#endif
#endif
