/*
 * raymarcher_amd.h — C-ABI of the MI355X-native sphere-tracing renderer.
 *
 * This is the drop-in boundary for ONE path of KentaYoshii/Raymarcher: the per-pixel raymarch that
 * the reference runs as a GLSL fragment shader behind one draw call.  The reference has no FFI for
 * that path; its de-facto operator interface is the uniform block of resources/raymarch.frag:245-286
 * plus glDrawArrays in Realtime::rayMarch() (src/realtimerender.cpp:53-87).  Every entry point below
 * names the reference interface it replaces.  All paths are relative to the reference checkout.
 *
 * Conventions
 *  - plain C, no C++/torch types; all structs are PODs with 4-byte members only (no padding surprises);
 *  - matrices are column-major float[16] exactly as glm / glUniformMatrix4fv(…, GL_FALSE, …) hand them over;
 *  - output frames are row-major float RGBA, **row 0 = bottom of the image** (GL convention,
 *    frag:2572-2574); rm_frame_to_rgba8() applies the vertical flip of Realtime::saveViewportImage
 *    (src/realtime.cpp:337-338);
 *  - d_* pointers are DEVICE pointers owned by the caller (hipMalloc / torch tensor storage); render calls are
 *    asynchronous on the caller's stream and act on the calling thread's current device.  Library state is per device
 *    (a ring of 7 KB scene-table slots that grows instead of blocking, timing records) and, for scratch memory, per
 *    (device, stream): rm_post_process and the experimental Mandelbulb pipelines keep a grow-only workspace for each
 *    stream they are called on (≤ 20 B/pixel resp. ≤ 52 B/pixel + 8 B per pixel·light), (re)allocated — with a
 *    synchronise of that stream — only when a larger frame than any before is processed on it.  Calls on different
 *    streams or devices may be issued concurrently from different host threads; calls on ONE stream must come from one
 *    thread at a time (as with any HIP stream);
 *  - every function returns an rm_status; no exception crosses this boundary (the reference throws
 *    std::runtime_error on shader failure, src/utils/shaderloader.h:39,83, and prints + returns on
 *    scene errors, src/raymarch/raymarchscene.cpp:111).
 */
#ifndef RAYMARCHER_AMD_H
#define RAYMARCHER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RM_ABI_VERSION 4

/* Capacity limits — src/realtime.h:17-27 (MAX_NUM_LIGHTS 10, MAX_NUM_SHAPES 30). */
#define RM_MAX_LIGHTS 10
#define RM_MAX_OBJECTS 30

/* Primitive type tags — src/utils/scenedata.h:18-33 == frag:53-68. */
enum {
  RM_CUBE = 0, RM_CONE = 1, RM_CYLINDER = 2, RM_SPHERE = 3, RM_OCTAHEDRON = 4, RM_TORUS = 5,
  RM_CAPSULE = 6, RM_DEATHSTAR = 7, RM_RECTANGLE = 8, RM_MANDELBROT = 9, RM_MANDELBULB = 10,
  RM_MENGERSPONGE = 11, RM_SIERPINSKI = 12, RM_CUSTOM = 13
};
/* Light type tags — src/utils/scenedata.h:10-15 == frag:72-75. */
enum { RM_LIGHT_POINT = 0, RM_LIGHT_DIRECTIONAL = 1, RM_LIGHT_SPOT = 2, RM_LIGHT_AREA = 3 };

/* Compile-time #defines of the reference shader (frag:4-15) as a runtime feature mask. */
enum {
  RM_FEAT_SKY_BACKGROUND = 1u << 0,      /* frag:5  */
  RM_FEAT_NIGHTSKY_BACKGROUND = 1u << 1, /* frag:6  (samples RmResources.noise) */
  RM_FEAT_DARK_BACKGROUND = 1u << 2,     /* frag:8  */
  RM_FEAT_WHITE_BACKGROUND = 1u << 3,    /* frag:9  */
  RM_FEAT_CLOUD = 1u << 4,               /* frag:12 */
  RM_FEAT_TERRAIN = 1u << 5,             /* frag:13 */
  RM_FEAT_SEA = 1u << 6,                 /* frag:14 (samples RmResources.noise) */
  RM_FEAT_PERLIN_BUMP = 1u << 7,         /* frag:15 */
  /* Not a reference #define — an opt-in evaluation scheme.  When set and power == 8 exactly, the Mandelbulb step
   * w ← c + r^8·(sin 8θ sin 8φ, cos 8θ, sin 8θ cos 8φ) (frag:789-793) is evaluated by three complex squarings of
   * (y + iρ) and of (z + ix)/ρ, ρ = |w.xz|, instead of acos/atan/sin/cos/pow, and m^3.5 as m³·√m: the same function
   * (angle-multiplication identities), different rounding (≈1e-6 relative per step), ≈3.5× fewer instructions.
   * Oracle and kernels implement it identically, so CPU/GPU parity stays bit-exact.  Any other power ignores the bit. */
  RM_FEAT_BULB_POWER8_ALGEBRAIC = 1u << 8
};
/* The checked-in shader's state: WHITE_BACKGROUND + PERLIN_BUMP (frag:9,15). */
#define RM_FEAT_REFERENCE_DEFAULT (RM_FEAT_WHITE_BACKGROUND | RM_FEAT_PERLIN_BUMP)

typedef enum rm_status {
  RM_OK = 0,
  RM_ERR_INVALID_ARGUMENT = 1, /* null pointer, bad size, rows out of range */
  RM_ERR_CAPACITY = 2,         /* > RM_MAX_OBJECTS / RM_MAX_LIGHTS (reference silently drops: realtimerender.cpp:662,737) */
  RM_ERR_UNSUPPORTED = 3,      /* CUSTOM objects; a feature whose resource (texture, noise, skybox, LTC table) was not supplied */
  RM_ERR_DEVICE = 4,           /* HIP runtime error; see rm_last_error() */
  RM_ERR_IO = 5,               /* file missing / unreadable */
  RM_ERR_PARSE = 6             /* scenefile schema violation */
} rm_status;

/* struct RayMarchObject — frag:135-168; uploaded by configureShapesUniforms, realtimerender.cpp:732-811. */
typedef struct RmObject {
  int32_t type;          /* RM_CUBE … RM_CUSTOM */
  float invModel[16];    /* world → object, column-major (obj.m_ctmInv) */
  float scaleFactor;     /* min diag of accumulated scale (realtimerender.cpp:749-751) */
  float shininess;
  float blend;
  float ior;
  float cAmbient[3];
  float cDiffuse[3];
  float cSpecular[3];
  float cReflective[3];
  float cTransparent[3];
  int32_t texLoc;        /* -1 = untextured; 0..numTextures-1 = index into RmResources.textures (rm_render_ex / rm_render_res) */
  float repeatU;
  float repeatV;
  int32_t isEmissive;    /* the rectangle drawn for an area light (raymarchscene.cpp:121-133): rendered as `color` */
  float color[3];
  int32_t lightIdx;
} RmObject;

/* struct LightSource — frag:212-230; uploaded by configureLightsUniforms, realtimerender.cpp:651-704. */
typedef struct RmLight {
  int32_t type;          /* RM_LIGHT_* */
  float color[3];
  float dir[3];          /* ctm·dir, NOT normalised (shader normalises) */
  float pos[3];
  float func[3];         /* attenuation (c0, c1, c2) */
  float angle;           /* spot outer angle, radians */
  float penumbra;        /* radians */
  float points[4][3];    /* area light corners tl,tr,br,bl in world space (realtimerender.cpp:688-693) */
  float intensity;
  int32_t twoSided;
} RmLight;

/* One object texture — objTextures[i] (frag:265), uploaded by initShapesTextures (realtimerender.cpp:267-303):
 * RGBA8, GL_LINEAR min/mag filter, GL_REPEAT wrap, no mipmaps; rows bottom-up (the reference mirrors the image
 * at load, raymarchscene.cpp:208).  `pixels` is a DEVICE pointer for rm_render_ex. */
#define RM_MAX_TEXTURES 10 /* src/realtime.h:17-27 */
typedef struct RmTexture {
  const uint8_t *pixels;
  int32_t width, height;
} RmTexture;

/*
 * Every sampler the shader reads, as the reference leaves them in GPU memory.  All images are RGBA8 with
 * GL_LINEAR filtering in binary32 weights; `pixels` are DEVICE pointers, rows in glTexImage2D order (row 0 = t 0).
 * A member with pixels == NULL is "not supplied"; rendering a scene that needs it fails with RM_ERR_UNSUPPORTED.
 */
#define RM_LTC_SIZE 64 /* LUT_SIZE, frag:47 */
typedef struct RmResources {
  const RmTexture *textures; /* objTextures[] (frag:265), GL_REPEAT */
  int32_t numTextures;
  RmTexture noise;           /* `noise` (frag:270; realtimerender.cpp:378-395: noise_texture_1.png, 256×256), GL_REPEAT;
                              * read by noiseV (frag:591-598) for NIGHTSKY_BACKGROUND and SEA */
  RmTexture skybox[6];       /* `skybox` cube map faces +X,−X,+Y,−Y,+Z,−Z (frag:267; initCubeMap,
                              * realtimerender.cpp:557-589), GL_CLAMP_TO_EDGE, filtered within a face
                              * (GL_TEXTURE_CUBE_MAP_SEAMLESS is never enabled); used when enableSkyBox */
  const uint8_t *ltc1;       /* LTC1 / LTC2 (frag:268-269): RM_LTC_SIZE² RGBA8 texels each.  The reference uploads   */
  const uint8_t *ltc2;       /* float tables with the unsized GL_RGBA internal format (realtimerender.cpp:908, 925), */
                             /* i.e. clamped to [0,1] and stored as 8-bit; rm_ltc_quantise() does that conversion.   */
} RmResources;
/* clamp(x,0,1)·255 rounded to nearest: float RGBA table → the 8-bit texels the reference's upload leaves. */
void rm_ltc_quantise(const float *table, uint8_t *out, int texels);

/* Camera uniforms — configureCameraUniforms, realtimerender.cpp:596-615. */
typedef struct RmCamera {
  float invProjView[16]; /* inverse(proj·view), column-major */
  float initialFar;      /* far plane (frag:247, 2425) */
  float eyePosition[4];  /* declared by the shader, unused by it (frag:245) */
} RmCamera;

/* Scalar uniforms — frag:248-255, 274, 282-283; realtimerender.cpp:621-645, 651-661, 809-810. */
typedef struct RmGlobals {
  float ka, kd, ks, kt;
  float power;           /* Mandelbulb power (settings.h:47, default 8) */
  float juliaSeed[2];
  float iTime;           /* seconds; 0 for offline frames */
  int32_t isTwoD;        /* 2-D Mandelbrot mode (frag:2431) */
} RmGlobals;

/* Option uniforms (frag:277-281) + the shader's compile-time constants as runtime knobs. */
typedef struct RmSettings {
  int32_t enableSoftShadow;
  int32_t enableReflection;
  int32_t enableRefraction;
  int32_t enableAmbientOcclusion;
  int32_t enableSkyBox;  /* frag:281, 2327: rays that miss every object sample RmResources.skybox */
  int32_t maxSteps;      /* MAX_STEPS, frag:28 (reference 256) */
  int32_t fractalIters;  /* MAX_STEPS_FRACTALS, frag:29 (reference 20) */
  int32_t mengerLevels;  /* loop bound of frag:1056 (reference 4) */
  int32_t numReflection; /* NUM_REFLECTION, frag:45 (reference 1) */
  uint32_t features;     /* RM_FEAT_* mask (reference: RM_FEAT_REFERENCE_DEFAULT) */
} RmSettings;

/* Fill *s with the reference's constants (256 steps, 20 fractal iterations, 4 Menger levels, 1 bounce,
 * WHITE_BACKGROUND|PERLIN_BUMP, all options off). */
void rm_settings_default(RmSettings *s);

/* ---- library / device ---------------------------------------------------------------------- */
int rm_abi_version(void);
/* sizeof() of ABI struct `which` as compiled into the library (0 RmObject, 1 RmLight, 2 RmCamera, 3 RmGlobals,
 * 4 RmSettings, 5 RmCounters, 6 RmHostSettings, 7 RmCameraData, 8 RmTexture, 9 RmPostSettings, 10 RmResources; -1 otherwise) so bindings can verify layout. */
int rm_abi_sizeof(int which);
const char *rm_status_string(int status);
/* Thread-local text of the last failure in this thread ("" if none). */
const char *rm_last_error(void);
/* Number of HIP devices; <0 on error. */
int rm_device_count(void);
/* hipSetDevice for the calling thread. */
int rm_set_device(int device);

/* ---- the hot path -------------------------------------------------------------------------- */
/*
 * rm_render — replaces Realtime::rayMarch(): the five configure*Uniforms calls + glDrawArrays
 * (src/realtimerender.cpp:53-87) and everything resources/raymarch.{vert,frag} do per pixel.
 * Renders rows [rowBegin,rowEnd) of a W×H frame.  d_rgba receives (rowEnd-rowBegin)·W float4
 * (fragColor, frag:18), d_bright the same for BrightColor (frag:19) or NULL.  Asynchronous on
 * `stream` (a hipStream_t, NULL = default stream).  Host structs are copied before return.
 */
int rm_render(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights,
              int numLights, const RmGlobals *g, const RmSettings *s, int W, int H, int rowBegin,
              int rowEnd, float *d_rgba, float *d_bright, void *stream);

/*
 * rm_render_ex — rm_render with object textures: objects whose texLoc is 0..numTextures-1 take their diffuse
 * colour from textures[texLoc] through the reference's uv maps (cube, cone, cylinder, sphere: frag:1299-1398) and
 * getDiffuse's blend (frag:1746-1781).  Other textured primitive types are rejected (the reference indexes
 * customTextures[texLoc-15] out of bounds for them).
 */
int rm_render_ex(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                 const RmGlobals *g, const RmSettings *s, const RmTexture *textures, int numTextures, int W, int H,
                 int rowBegin, int rowEnd, float *d_rgba, float *d_bright, void *stream);

/*
 * rm_render_res — rm_render with every sampler the shader can read (RmResources): object textures, the noise
 * texture of the night sky / sea, the sky-box cube map and the LTC tables of area lights.  `res` may be NULL.
 */
int rm_render_res(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                  const RmGlobals *g, const RmSettings *s, const RmResources *res, int W, int H, int rowBegin,
                  int rowEnd, float *d_rgba, float *d_bright, void *stream);

/*
 * rm_render_tiles — the multi-GPU shard of the same frame (no reference counterpart; the reference
 * renders whole frames on one GPU).  The frame is cut into tiles of `tileRows` rows; this call renders
 * tiles t with t % numShards == shard, packed contiguously in tile order into d_rgba
 * (rm_shard_rows() rows × W float4).  Interleaving balances the centre-heavy cost across GPUs.
 */
int rm_render_tiles(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights,
                    int numLights, const RmGlobals *g, const RmSettings *s, int W, int H, int tileRows,
                    int shard, int numShards, float *d_rgba, float *d_bright, void *stream);
/* rm_render_tiles with the samplers of rm_render_res (`res` may be NULL): every shard passes the same resources, each
 * GPU holding its own copy of the images. */
int rm_render_tiles_res(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                        const RmGlobals *g, const RmSettings *s, const RmResources *res, int W, int H, int tileRows,
                        int shard, int numShards, float *d_rgba, float *d_bright, void *stream);
/* The row-tile partition.  Default (root relief 0): tile t belongs to shard t mod numShards.  rm_set_root_relief(K), K in 2..64:
 * the deal runs in cycles of numShards·K − 1 tiles — K − 1 full rounds, then one round that leaves shard 0 out — so shard 0, the
 * gather's root (which also receives numShards − 1 slots and de-interleaves the whole frame every frame), renders (K − 1)/K of a
 * peer's tiles.  The setting is PROCESS-WIDE and every entry point that deals tiles reads it (rm_render_tiles*, rm_shard_rows,
 * rm_shard_row_to_frame, rm_deinterleave*, rm_gather_*): every rank of a job sets the same value before rendering.  With relief
 * the largest shard is shard 1, not shard 0: size gather slots by rm_gather_slot_rows.  0 switches it off. */
int rm_set_root_relief(int K);
int rm_get_root_relief(void);
/* Rows owned by `shard` under the rm_render_tiles partition. */
int rm_shard_rows(int H, int tileRows, int shard, int numShards);
/* Frame row of the shard's packed row `localRow` (inverse map used when de-interleaving a gather). */
int rm_shard_row_to_frame(int H, int tileRows, int shard, int numShards, int localRow);
/*
 * rm_deinterleave — scatter the concatenation of all shards' packed rows (shard 0 first, the layout an
 * RCCL gather produces) into frame order.  Shard k's rows start at row k·shardStrideRows of d_gathered
 * (equal-sized gather slots, padded at the end); shardStrideRows = 0 means tightly packed.
 * d_frame: H·W float4, device, distinct from d_gathered.
 */
int rm_deinterleave(const float *d_gathered, float *d_frame, int W, int H, int tileRows, int numShards,
                    int shardStrideRows, void *stream);

/*
 * rm_gather_* — the gather of a sharded frame, for a host that drives all GPUs of a node from ONE process (no reference
 * counterpart).  devices[k] renders shard k of numDevices (rm_render_tiles on a stream of that device);
 * rm_gather_tiles then moves every shard's packed rows into slot k of d_gathered on devices[root] — one grouped
 * ncclSend / ncclRecv pair per peer over RCCL (xGMI: each peer has its own link to the root), the root's own tiles by a
 * device copy — asynchronously: the send of shard k is enqueued on streams[k] (behind its render), the receives on
 * streams[root], where rm_deinterleave(d_gathered, d_frame, W, H, tileRows, numDevices, rm_gather_slot_rows(...),
 * streams[root]) follows.  Slots are rm_gather_slot_rows(H, tileRows, numDevices) rows each (the largest shard), so
 * d_gathered holds numDevices · slotRows · W float4.  librccl is loaded on first use (RM_ERR_UNSUPPORTED if absent);
 * with one device no communicator is created.  One process per GPU (torch.distributed / MPI hosts) does not need this:
 * raymarcher_amd/dist.py gathers with the process group's own RCCL.
 */
typedef struct RmGather RmGather;
int rm_gather_create(const int *devices, int numDevices, RmGather **out);
/* flags: RM_GATHER_FORCE_COMM builds the RCCL communicator(s) even for ONE device and sends the root's own tiles to itself
 * through ncclSend / ncclRecv instead of a device copy — the whole RCCL path (library load, ncclCommInitAll, a grouped
 * send / receive pair) on a one-GPU box; tests and bring-up.  A grouped call that fails aborts the communicators
 * (ncclCommAbort: no unmatched send is left on a stream) and the object refuses further use. */
enum { RM_GATHER_FORCE_COMM = 1u };
int rm_gather_create_ex(const int *devices, int numDevices, unsigned flags, RmGather **out);
void rm_gather_destroy(RmGather *g);
int rm_gather_slot_rows(int H, int tileRows, int numShards);
int rm_gather_tiles(RmGather *g, const float *const *d_tiles, float *d_gathered, int W, int H, int tileRows, int root,
                    void *const *streams);
/* The same gather with 4 bytes per pixel instead of 16, for hosts that only need the 8-bit image (saveViewportImage writes
 * a PNG, src/realtime.cpp:284-350): every shard converts its packed tiles with rm_tiles_to_rgba8 (clamp → ×255 → round, no
 * flip) on its own stream, rm_gather_tiles_rgba8 moves them (a quarter of the xGMI traffic into the root), and
 * rm_deinterleave_rgba8 writes the frame — rows in frame order, or flipped top-down like rm_frame_to_rgba8 (flip != 0). */
int rm_tiles_to_rgba8(const float *d_tiles, uint8_t *d_tiles8, int W, int rows, void *stream);
int rm_gather_tiles_rgba8(RmGather *g, const uint8_t *const *d_tiles8, uint8_t *d_gathered8, int W, int H, int tileRows, int root,
                          void *const *streams);
int rm_deinterleave_rgba8(const uint8_t *d_gathered8, uint8_t *d_frame8, int W, int H, int tileRows, int numShards,
                          int shardStrideRows, int flip, void *stream);

/* Fractal / shading work counters of the last counted render (debug/roofline accounting). */
typedef struct RmCounters {
  uint64_t sceneEvals;    /* sdScene evaluations (frag:1406) */
  uint64_t bulbIters;     /* Mandelbulb inner iterations (frag:785-799) */
  uint64_t hitPixels;     /* pixels whose primary ray hit */
  uint64_t shadedPoints;  /* surface points shaded by render() (frag:2333-2373): primary hits and reflection / refraction hits */
  uint64_t terrainEvals;  /* fbm_9 evaluations (frag:630-644): terrain height samples of the TERRAIN layer */
  uint64_t cloudEvals;    /* fbmd_8 evaluations (frag:647-667): cloud density samples (CLOUD) and the terrain's bump / cloud shadow */
  uint64_t shapeEvals;    /* sdMatch evaluations (frag:1262-1293): objects evaluated, summed over the sdScene evaluations — numObjects
                             per evaluation as the shader is written; fewer in RM_COUNT_EXECUTED where the table walk passes over objects */
} RmCounters;
/* Same as rm_render but also accumulates counters with device atomics (slower; synchronises).  Counts the REFERENCE's
 * work: every evaluation the shader as written performs, i.e. without the bit-identical shortcuts of the production
 * kernels (marches ended at the scene's bounding ball, no shadow march for a light that N·L <= 0.005 drops anyway). */
int rm_render_counted(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights,
                      int numLights, const RmGlobals *g, const RmSettings *s, int W, int H, int rowBegin,
                      int rowEnd, float *d_rgba, float *d_bright, RmCounters *out);
/* mode RM_COUNT_REFERENCE = rm_render_counted; RM_COUNT_EXECUTED counts the work the production kernel really executes
 * (shortcuts honoured) — the pair gives the algorithmic and the executed figure of the roofline. */
enum { RM_COUNT_REFERENCE = 1, RM_COUNT_EXECUTED = 2 };
/* rm_render_counted_ex with samplers (res may be NULL): scenes with procedural layers or samplers are counted in mode
 * RM_COUNT_REFERENCE only (their kernels have no bit-identical shortcuts to tell apart). */
int rm_render_counted_res(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights,
                          int numLights, const RmGlobals *g, const RmSettings *s, const RmResources *res, int W, int H,
                          int rowBegin, int rowEnd, float *d_rgba, float *d_bright, int mode, RmCounters *out);
int rm_render_counted_ex(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights,
                         int numLights, const RmGlobals *g, const RmSettings *s, int W, int H, int rowBegin,
                         int rowEnd, float *d_rgba, float *d_bright, int mode, RmCounters *out);
/* Diagnostic build of the single-Mandelbulb kernel and of the plain table-walk kernel (no samplers, no procedural layers;
 * RM_ERR_UNSUPPORTED otherwise): production code + s_memtime / s_memrealtime stamps per wave, written to a buffer of their own: renders the whole frame once, synchronises and returns the shader clock the chip held under
 * this kernel's own load, in MHz (Σ cycle spans ÷ Σ 100 MHz-tick spans over all waves).  Call it after a few back-to-back
 * renders so that the clock has settled.  d_waveSpans (device, may be NULL): 2 words per wave — its first and last
 * s_memrealtime stamp (100 MHz ticks) — indexed tile·w + wave with w = waves per workgroup (1 unless RM_WAVES_PER_BLOCK
 * overrides) and ceil(W / (8·w)) tiles per tile row: size it for (ceil(W/8) + 3)·ceil(H/8) waves; for occupancy timelines. */
int rm_render_clocked(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                      const RmGlobals *g, const RmSettings *s, int W, int H, float *d_rgba, double *shaderMHz,
                      unsigned long long *d_waveSpans);

/* Average device time in ms of the `rm_render*` launches made on the CURRENT device since rm_set_timing(1), timed with
 * hipEvents on their own stream; rm_get_* reads and resets that device's records (the on/off switch is process-wide). */
int rm_set_timing(int on);
int rm_get_timing(double *avgKernelMs, int *launches);
/* Same, split by role, both averaged over ALL the launches (total = stage 0 + stage 1): stage 1 = the render (the one-lane-per-pixel
 * kernel, or all kernels of the wavefront pipeline), stage 0 = the tile-ordering launches that preceded it in the launches that had
 * them (rm_set_tile_order: a new picture and the first repeats of one; a settled picture, a small frame or raster order has none).
 * Stages 2-3 are zero. */
int rm_get_stage_timing(double *avgTotalMs, double avgStageMs[4], int *launches);
/* Which schedule renders a frame: 0 = the measured-fastest one of the scene's class (default), 1 = one lane per pixel
 * (rm::render_kernel, every class), 5 = the wavefront pipeline of rm_wavefront.hip.h for the table-walk classes (primitives,
 * Menger sponge, Sierpinski; no samplers, procedural layers, refraction, Mandelbulb or 2-D Mandelbrot in the scene) — per
 * generation of rays (primary, then each reflection bounce of frag:2491-2524) a persistent march kernel whose lanes are rays
 * refilled from a queue as they end, a dense surface kernel, the same march kernel over the shadow rays, a dense light /
 * bounce kernel.  A request that does not apply to the scene (5 with a Mandelbulb, samplers, layers or refraction, or on a
 * frame its 32-bit ray ids cannot cover) runs 1.  Both produce identical bits; the switch exists for A/B measurement and
 * tests.  (Paths 2-4, three multi-kernel pipelines of the single-Mandelbulb class, were measured slower than path 1 on every
 * frame incl. frames without tile-order history — profiles/r04_b_bulb_paths.md — and removed in round 4; the numbers are refused.) */
int rm_set_kernel_path(int path);
/* Scratch memory the library owns.  Everything the schedules need beyond the caller's frame lives in grow-only buffers per
 * (device, stream): 8 B per tile for the tile-order feedback, the post passes' ping-pong images, and — by far the largest —
 * the wavefront pipeline's ray / hit / path records, ≈(160 + 4·numLights) bytes per pixel of the launch (5.8 GB for a
 * 7680×4320 frame, once per stream that renders such frames).  rm_set_workspace_limit caps the size of any ONE such buffer
 * (0 = no limit, the default; the environment variable RM_WF_MAX_WORKSPACE_BYTES sets the initial value).  When the wavefront
 * workspace exceeds the limit or the device cannot allocate it, a launch that chose the pipeline by itself (kernel path 0)
 * renders with rm::render_kernel instead — identical pixels, no workspace — and remembers the refusal for that stream; only
 * an explicit rm_set_kernel_path(5) reports RM_ERR_DEVICE.  An allocation failure never leaves HIP's error state set.
 * rm_release_workspaces drains the current device and frees all of its buffers (freedBytes may be NULL); the next launch
 * that needs one allocates it again, and the next frame on each stream runs in raster tile order. */
int rm_set_workspace_limit(unsigned long long bytes);
int rm_release_workspaces(unsigned long long *freedBytes);
/* Tests: the schedule (numbering above; never 0) the most recent render launch on the current device ran, -1 on error. */
int rm_debug_last_path(void);
/* Tests: how many tiles the most recent render launch on the current device rendered one light per workgroup ("light split": the
 * heaviest tiles of a SETTLED picture of the plain table-walk class with two or more lights are rendered by numLights workgroups
 * each, one shadow march per pixel apiece, and finished — by whichever of them arrives last — from the stored results: the same
 * marches and the same sums in the same order, so the same pixels; it shortens the longest waves of latency-bound frames, C2 0.79 →
 * 0.51 ms.  Whether it pays is measured per picture.  RM_LIGHT_SPLIT=0 turns it off, =n makes the heaviest 1/n of the tiles the
 * candidates; default 256).  0: none; -1 on error. */
int rm_debug_last_split(void);
/* Tests / experiments: the divisor above for the process — n >= 1 also splits WITHOUT the measurement that normally decides per
 * picture whether the split pays (settled frames 0-1 plain, 2-3 split, the better of the two from then on); 0 = off, -1 = back to
 * RM_LIGHT_SPLIT / the default, measured. */
int rm_debug_set_light_split(int div);
/* Launch order of a frame's tiles (workgroups).  Tile costs span three orders of magnitude and a single ray that never
 * converges is a sequential chain of ~1 ms, so a kernel whose heaviest tiles start late ends in a tail of a few lonely
 * waves; starting heavy tiles first removes it.  The order never changes a pixel.  mode 1 (default): every frame records each
 * tile's shader-cycle cost; the next frame of the same size on the same stream starts its tiles heaviest-first by those costs
 * when it is the SAME picture (scene tables, camera, settings, rows) — a picture that keeps repeating settles: from its fifth
 * frame on the order of the fourth is reused, with no ordering launches and no cost recording (RM_TILE_ORDER_SETTLE=0: re-sort
 * every frame) — and otherwise — the first frame, a moved camera, a
 * changed scene — by a geometric classification of the tiles (centre ray against the objects' bounding balls: silhouette rings
 * first, interiors next, background last), combined with the stale costs where a frame of that size was rendered before;
 * scenes with procedural layers or objects without a bound start new pictures in raster order.  mode 0: always raster order;
 * -1: back to the default / the RM_TILE_ORDER environment variable.  Applies to every launch of the one-lane-per-pixel kernel
 * with at least 2048 tiles (not to the 2-D Mandelbrot path or the wavefront pipeline, whose persistent waves balance
 * themselves). */
int rm_set_tile_order(int mode);
/* Shape of the pixel tile a wave renders: 8×8 by default; for table-walk, sampler and layer scenes the launcher measures, per
 * stream and picture, whether 4 wide × 16 tall tiles are faster (frames 0-1 and 4-5 of a picture run 8×8, frames 2-3 and 6-7 4×16,
 * the second of each pair timed with HIP events) and keeps the shape with the smaller best time from the ninth frame on.  The shape never changes a pixel.  Tests /
 * experiments: mode 0 = tune (default), 3 = always 8×8, 2 = always 4×16, -1 = back to the RM_TILE_SHAPE environment variable. */
int rm_debug_set_tile_shape(int mode);
/* Experiments: force a given launch order (d_order: a device permutation of 0..tileCount-1, or NULL) and / or collect the
 * tiles' costs (d_cost: tileCount device words, accumulated, or NULL) for subsequent launches on the current device. */
int rm_debug_set_tile_order(const int32_t *d_order, uint32_t *d_cost, int tileCount);
/* Tests: the 48 coefficients [triangle][near, far][P0, P1 − P0, P2 − P0][xyzw] from which the kernels interpolate nearClip /
 * farClip (raymarch.vert:23-24 evaluated at the corners of the full-screen quad, realtimerender.cpp:225-238; DESIGN.md
 * §2.3), computed on the host exactly as the launcher stages them.  No GPU needed. */
int rm_debug_ray_planes(const RmCamera *cam, float *out48);
/* Tests: the bounds the launcher stages for ending marches whose miss distance nobody reads: out14 = { ok,
 * centre xyz, R² of the ball, R² of the soft-shadow ball (0 = none), boxOk, box lo xyz, box hi xyz, lip }.  Outside the ball — and,
 * where boxOk, outside the box — every object's distance value exceeds the hit threshold (0.001) by a wide factor, so a ray
 * that has left ball ∩ box for good can only miss.  The box is staged only where it is much tighter than the ball (volume
 * ratio < 0.3).  lip: no object's distance value changes by more than lip per unit of world length (+inf with a fractal in
 * the table): the seed of the table walk's skip test.  A pure function of the object table and the globals; no GPU needed. */
int rm_debug_cull_bounds(const RmObject *objs, int numObjects, const RmGlobals *g, float *out14);
/* Tests: the kernels' cheap exact forms against the IEEE operations for every one of the 2^32 inputs, on the current device
 * (≈2 s).  mismatches5[0]: the reciprocal (v_rcp_f32 + one Newton step inside 2^-126 <= |y| < 2^126, the IEEE expansion
 * outside) vs 1.0f / y; [1]: the bare fast form over its range; [2]: the square root (v_sqrt_f32 + residual selection, the
 * scaled expansion only below 2^-96) vs sqrtf; [3]: the unscaled form over its domain; [4]: fract (v_fract_f32) vs
 * x − floor(x) kept below 1.  All must be 0. */
int rm_debug_check_math(unsigned long long *mismatches5);

/*
 * rm_frame_to_rgba8 — clamp→×255→round and vertical flip, the read-back of
 * Realtime::saveViewportImage (src/realtime.cpp:284-350).  d_rgba: H·W float4 (row 0 = bottom);
 * d_out: H·W·4 bytes, row 0 = top.
 */
int rm_frame_to_rgba8(const float *d_rgba, uint8_t *d_out, int W, int H, void *stream);

/* ---- post passes (src/realtimerender.cpp:78-165; resources/blur.frag, hdr.frag, fxaa.frag) ------------------ */
/* Settings surface — src/settings.h:37-41. */
typedef struct RmPostSettings {
  int32_t enableFXAA, enableGammaCorrection, enableHDR, enableBloom;
  float exposure;
} RmPostSettings;
/*
 * rm_post_process — what Realtime::rayMarch() does after the draw call: applyLightEffects() (bloom = 10 ping-pong
 * passes of a separable 9-tap Gaussian over BrightColor, of which the reference composites the 9th; then gamma
 * 1/2.2, or 1−exp(−(colour+bloom)·exposure)) and applyFXAA().  Storage formats are the reference's: the HDR /
 * bright / ping-pong targets are RGBA16F (values are rounded to binary16 between passes), the FXAA source is
 * RGBA8 sampled with GL_LINEAR / GL_REPEAT.  d_frag, d_bright (may be NULL without bloom) and d_out are H·W
 * float4, row 0 = bottom; d_out receives the colour the 8-bit default framebuffer would quantise
 * (rm_frame_to_rgba8 does that).  Uses a grow-only per-device workspace of 24 B/pixel.
 */
int rm_post_process(const float *d_frag, const float *d_bright, float *d_out, int W, int H, const RmPostSettings *ps,
                    void *stream);

/* ---- math spec probes (tests only: evaluate the device implementation of one rm_math function
 *      element-wise so it can be compared bit-for-bit with the oracle) ---------------------------- */
enum { RM_FN_SIN = 0, RM_FN_COS, RM_FN_ACOS, RM_FN_ATAN2, RM_FN_LOG2, RM_FN_EXP2, RM_FN_POW, RM_FN_SQRT,
       RM_FN_DIV, RM_FN_PNOISE3, RM_FN_ASIN, RM_FN_Q16 /* round to binary16 and back */,
       RM_FN_SQRT_FAST /* device-only variant of sqrt, must equal RM_FN_SQRT bit for bit */,
       RM_FN_DIVR /* x · RN(1/y), the contract's hot-path quotient */, RM_FN_RCP /* device form of 1.0f / x */,
       RM_FN_SMOOTHSTEP /* smoothstep(x, y, z) */, RM_FN_MIN, RM_FN_MAX, RM_FN_FRACT,
       RM_FN_MEDIAN_ABS /* device only: v_med3_f32(|x|, |y|, |z|), the Menger level's spelling of min(max(x,y), min(max(y,z), max(z,x))) */,
       RM_FN_COUNT };
int rm_probe_math(int fn, const float *d_x, const float *d_y, const float *d_z, float *d_out, int n,
                  void *stream);
/* Evaluate sdScene (frag:1406-1430) at n world-space points: d_out[4n] = (minD, minObjIdx, trap.y, trap.z). */
int rm_probe_sdscene(const RmObject *objs, int numObjects, const RmGlobals *g, const RmSettings *s,
                     const float *d_pts, float *d_out, int n, void *stream);

/* ---- host side kept from the reference: scenefile loader, camera, Settings -------------------- */
/* Settings surface — src/settings.h:19-55 (render-relevant fields only). */
typedef struct RmHostSettings {
  int32_t screenWidth, screenHeight; /* settings.h:21-22 */
  float nearPlane, farPlane;         /* settings.h:28-29 (GUI defaults 0.1 / 100, mainwindow.cpp:129-130) */
  int32_t twoDSpace;
  int32_t enableSoftShadow, enableReflection, enableRefraction, enableAmbientOcculusion;
  float power;                       /* settings.h:47 */
  float juliaSeed[2];                /* settings.h:48 */
} RmHostSettings;
void rm_host_settings_default(RmHostSettings *s);

/* Camera — src/camera/camera.cpp:8-34 (initializeCamera), :74-97 (view), :105-133 (proj). */
typedef struct RmCameraData {
  float pos[4], look[4], up[4]; /* SceneCameraData, scenedata.h:110-120 */
  float heightAngle;            /* radians */
} RmCameraData;
int rm_camera_build(const RmCameraData *cd, int W, int H, float nearPlane, float farPlane,
                    float view[16], float proj[16], RmCamera *out);

/* Opaque parsed scene — SceneParser::parse (src/utils/sceneparser.cpp:117-133) +
 * RayMarchScene::initScene (src/raymarch/raymarchscene.cpp:104-134). */
typedef struct RmScene RmScene;
int rm_scene_load(const char *path, RmScene **out);
int rm_scene_load_string(const char *json, RmScene **out);
void rm_scene_free(RmScene *scene);
int rm_scene_num_objects(const RmScene *scene);
int rm_scene_num_lights(const RmScene *scene);
/* Pointers stay valid until rm_scene_free. */
const RmObject *rm_scene_objects(const RmScene *scene);
const RmLight *rm_scene_lights(const RmScene *scene);
int rm_scene_globals(const RmScene *scene, const RmHostSettings *hs, RmGlobals *out);
int rm_scene_camera_data(const RmScene *scene, RmCameraData *out);
/* Texture file referenced by object i, or NULL; load it with rm_image_load(path, 1, …) into RmResources.textures[texLoc]. */
const char *rm_scene_object_texture(const RmScene *scene, int i);

/* Image file → RGBA8 (stands in for QImage::load + convertToFormat(RGBA8888) + mirrored(), raymarchscene.cpp:198-209).
 * PNG (8/16-bit grey, grey+alpha, RGB, RGBA, palette; non-interlaced) and baseline JPEG (grayscale or YCbCr, 4:4:4 /
 * 4:2:2 / 4:2:0; decoded with libjpeg's default arithmetic — islow IDCT, fancy upsampling — so the pixels equal
 * QImage's) and the first frame of a GIF.  Other formats, progressive JPEG: RM_ERR_UNSUPPORTED.  flipVertical = 1 gives the bottom-up
 * rows the renderer expects.  *outPixels is malloc'ed host memory of w·h·4 bytes; free with rm_image_free. */
int rm_image_load(const char *path, int flipVertical, uint8_t **outPixels, int *w, int *h);
void rm_image_free(uint8_t *pixels);

/* Sky-box selection of the GUI (settings.idxSkyBox → RayMarchScene::getCubeMapWithType, raymarchscene.cpp:50-86;
 * enum CUBEMAP, scenedata.h:43-48): path of face `face` (0..5, the order of RmResources.skybox) of cube map `which`
 * (1 BEACH, 2 NIGHTSKY, 3 ISLAND) relative to the scenefiles directory, or NULL.  Reproduced as written, including the
 * NIGHTSKY list naming −x before +x and −y before +y.  Load each with rm_image_load(path, 1, …) as initCubeMap does. */
const char *rm_skybox_face_path(int which, int face);

/* PNG writer for RGBA8 rows (top row first) — stands in for QImage::save (realtime.cpp:346). */
int rm_write_png(const char *path, const uint8_t *rgba, int W, int H);

#ifdef __cplusplus
}
#endif
#endif /* RAYMARCHER_AMD_H */
