/*
 * scripts/sim/wave_sim.c — OFFLINE schedule simulator for the single-Mandelbulb kernel (not a test, not the product).
 *
 * Includes the CPU oracle with its work-trace hooks enabled, renders sampled 32×8-pixel workgroups of a frame, records
 * for every pixel the Mandelbulb iteration count of every sdScene evaluation of every march (primary, 4 normal taps,
 * one shadow ray per light that is not dropped), applies the production kernel's bit-identical culls (bounding-ball
 * march end, no shadow march for N·L <= 0.005), and then prices several wave schedules in WAVE-LEVEL VALU instructions:
 *
 *   nested      the shipped schedule: one lane per pixel, march loop around the iteration loop; a wave pays, per march
 *               step, the evaluation overhead once and the iteration body max-over-lanes times
 *   flat        the iteration loop and the march loop flattened into one loop whose body is ONE Mandelbulb iteration;
 *               lanes that finish an evaluation run the epilogue/prologue block in the same trip (paid when any lane does)
 *   park(T)     as flat, but finished lanes park and the epilogue block runs when >= T lanes wait or nobody iterates
 *   wg-*        work items compacted across the 256 lanes of the workgroup before they are marched
 *   queue       each lane marches its own shadow rays back to back (no wave-level sync between lights)
 *
 * Output: per schedule, total wave-instructions and lane utilisation (lane-level useful instructions / 64·wave-level).
 * Build: see scripts/sim/run_wave_sim.py.
 */
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#define SIM_MAXS 256
#define SIM_MAXL 4
typedef struct {
  int nPrimary;
  uint8_t primary[SIM_MAXS + 8];
  int hit;
  uint8_t normal[4];
  int nShadow[SIM_MAXL]; /* 0 = light dropped / pixel missed */
  uint8_t shadow[SIM_MAXL][SIM_MAXS + 8];
} PixTrace;

static __thread PixTrace *t_pix;
static __thread int t_light;      /* index of the light whose shadow march comes next */
static __thread int t_phase;      /* 0 primary, 2 shadow */
static __thread float t_cullR2;   /* object-space cull radius² (0 = no cull) */

static void sim_eval(int iters) {
  PixTrace *p = t_pix;
  if (!p) return;
  if (t_phase == 0) { if (p->nPrimary < SIM_MAXS + 8) p->primary[p->nPrimary++] = (uint8_t)iters; }
  else { int l = t_light - 1; if (l >= 0 && l < SIM_MAXL && p->nShadow[l] < SIM_MAXS + 8) p->shadow[l][p->nShadow[l]++] = (uint8_t)iters; }
}
static int sim_skip_shadow(float ndotl) {
  PixTrace *p = t_pix;
  if (!p) return 0;
  if (t_light == 0) { /* first light of a hit pixel: the last four primary-phase evaluations were the normal taps */
    p->hit = 1;
    for (int k = 0; k < 4; k++) p->normal[k] = p->primary[p->nPrimary - 4 + k];
    p->nPrimary -= 4;
  }
  t_light++;
  t_phase = 2;
  return ndotl <= 0.005f;
}

/* ---- generic scenes (constant-cost evaluations): per pixel the sequence of marches with their evaluation counts */
#define SIM_MAXSEG 40
typedef struct { int n; short evals[SIM_MAXSEG]; unsigned char kind[SIM_MAXSEG]; } SegTrace;
static __thread SegTrace *t_seg;
static __thread int t_segKind;
static void seg_begin(int kind) {
  SegTrace *s = t_seg;
  if (!s) return;
  if (s->n < SIM_MAXSEG) { s->kind[s->n] = (unsigned char)kind; s->evals[s->n] = 0; s->n++; }
}
static void seg_eval(void) {
  SegTrace *s = t_seg;
  if (!s) return;
  if (s->n == 0) seg_begin(0);
  s->evals[s->n - 1]++;
}
#define RMO_TRACE_SCENE_EVAL(c) seg_eval()
#define RMO_TRACE_EVAL(c, iters) sim_eval(iters)
#define RMO_TRACE_MARCH(c, kind, ro, rd, endp) \
  t_segKind = (kind), sim_march_begin((const void *)(c), (ro).x, (ro).y, (ro).z, (rd).x, (rd).y, (rd).z, (endp))
static void sim_march_begin(const void *ctx, float rox, float roy, float roz, float rdx, float rdy, float rdz, float *endp);
#define RMO_TRACE_SKIP_SHADOW(c, N, L) sim_skip_shadow(dot3((N), (L)))
#include "../../oracle/rm_oracle.c"

/* the production kernel's bulbCullEnd (rm_device.hip.h): end the march where the ray leaves |p_obj| <= R for good */
static void sim_march_begin(const void *ctx, float rox, float roy, float roz, float rdx, float rdy, float rdz, float *endp) {
  const Ctx *c = (const Ctx *)ctx;
  seg_begin(t_segKind);
  const v3 ro = V3(rox, roy, roz), rd = V3(rdx, rdy, rdz);
  if (!(t_cullR2 > 0.0f)) return;
  const float *M = c->objs[0].invModel;
  v3 po = xform_point(M, ro);
  v3 pd = V3(rm_fma(M[8], rd.z, rm_fma(M[4], rd.y, M[0] * rd.x)), rm_fma(M[9], rd.z, rm_fma(M[5], rd.y, M[1] * rd.x)),
             rm_fma(M[10], rd.z, rm_fma(M[6], rd.y, M[2] * rd.x)));
  float a = dot3(pd, pd), b = dot3(po, pd), cc = dot3(po, po) - t_cullR2;
  float disc = rm_fma(b, b, -(a * cc));
  float tExit = (rm_sqrt(rm_max(disc, 0.0f)) - b) / a;
  tExit = rm_fma(tExit, 1.0001f, 1.0e-3f);
  if (cc > 0.0f && (b >= 0.0f || disc < 0.0f)) tExit = -1.0f;
  if (!(a > 0.0f)) return;
  if (tExit < *endp) *endp = tExit;
}

/* ---------------------------------------------------------------- cost model */
typedef struct {
  double cIt, cEv;     /* iteration body, per-evaluation overhead (prologue + DE epilogue + march bookkeeping) */
  double cItF, cEvF;   /* the same inside a flattened state-machine loop (state handling on top) */
  double cRay;         /* per-ray setup (light geometry, cull end) */
  double cHit;         /* bump + material + Phong terms per wave that holds a hit pixel (same in every schedule) */
  int hist;            /* fill the histograms */
} Cost;

typedef struct { const uint8_t *it; int n; } Item; /* one march: n evaluations with it[k] iterations each */

typedef struct { double wave, lane; } Acc; /* wave-level instructions, lane-level useful instructions */

static double item_lane_cost(const Item *x, const Cost *k) {
  double s = 0;
  for (int e = 0; e < x->n; e++) s += k->cEv + k->cIt * x->it[e];
  return s;
}
/* histogram of the shipped schedule's march cost by the number of lanes still marching (filled by sched_nested) */
static double g_costByActive[65];
static double g_costByStep[257];
/* nested loops over <= 64 items in one wave */
static void sched_nested(const Item *it, int n, const Cost *k, Acc *a) {
  int steps = 0;
  for (int i = 0; i < n; i++) { if (it[i].n > steps) steps = it[i].n; a->lane += item_lane_cost(&it[i], k); }
  for (int s = 0; s < steps; s++) {
    int mx = 0, act = 0;
    for (int i = 0; i < n; i++) if (it[i].n > s) { act++; if (it[i].it[s] > mx) mx = it[i].it[s]; }
    a->wave += k->cEv + k->cIt * mx;
    if (k->hist) {
#pragma omp atomic
      g_costByActive[act] += k->cEv + k->cIt * mx;
#pragma omp atomic
      g_costByStep[s < 256 ? s : 256] += k->cEv + k->cIt * mx;
    }
  }
}
/* flattened loop with parking threshold T (T = 1: the epilogue runs in the trip in which a lane finishes).
 * Each lane owns a QUEUE of items (qn[i] items starting at q[i]) that it marches back to back. */
static void sched_flat(const Item *const *q, const int *qn, int n, int T, const Cost *k, Acc *a) {
  int cur[64], e[64], rem[64], parked[64], done[64];
  int live = 0;
  for (int i = 0; i < n; i++) {
    cur[i] = 0; e[i] = 0; parked[i] = 0; done[i] = 0; rem[i] = 0;
    while (cur[i] < qn[i] && q[i][cur[i]].n == 0) cur[i]++;
    if (cur[i] >= qn[i]) { done[i] = 1; continue; }
    for (int j = 0; j < qn[i]; j++) a->lane += item_lane_cost(&q[i][j], k) + k->cRay;
    rem[i] = q[i][cur[i]].it[0];
    live++;
  }
  if (!live) return;
  a->wave += k->cEvF; /* prologue of the first evaluation */
  while (live) {
    int iterating = 0, nparked = 0;
    for (int i = 0; i < n; i++) if (!done[i]) { if (parked[i]) nparked++; else iterating++; }
    if (iterating) {
      a->wave += k->cItF;
      for (int i = 0; i < n; i++)
        if (!done[i] && !parked[i]) { if (--rem[i] <= 0) { parked[i] = 1; nparked++; iterating--; } }
    }
    if (nparked && (nparked >= T || iterating == 0)) {
      a->wave += k->cEvF;
      int newRay = 0;
      for (int i = 0; i < n; i++)
        if (!done[i] && parked[i]) {
          parked[i] = 0;
          e[i]++;
          if (e[i] >= q[i][cur[i]].n) { /* march over: next ray of the lane's queue */
            e[i] = 0; cur[i]++;
            while (cur[i] < qn[i] && q[i][cur[i]].n == 0) cur[i]++;
            if (cur[i] >= qn[i]) { done[i] = 1; live--; continue; }
            newRay = 1;
          }
          rem[i] = q[i][cur[i]].it[e[i]];
        }
      if (newRay) a->wave += k->cRay;
    }
  }
}

/* nested loops where every lane marches its OWN queue of rays back to back: a wave step evaluates each live lane's next
 * point; lanes whose ray ended set up their next ray in the same step (paid once per step if any lane does) */
static void sched_nested_queue(const Item *const *q, const int *qn, int n, const Cost *k, Acc *a) {
  int cur[64], e[64], live = 0;
  for (int i = 0; i < n; i++) {
    cur[i] = 0; e[i] = 0;
    while (cur[i] < qn[i] && q[i][cur[i]].n == 0) cur[i]++;
    if (cur[i] < qn[i]) live++;
    for (int j = 0; j < qn[i]; j++) if (q[i][j].n) a->lane += item_lane_cost(&q[i][j], k) + k->cRay;
  }
  if (live) a->wave += k->cRay;
  while (live) {
    int mx = 0, sw = 0;
    for (int i = 0; i < n; i++) {
      if (cur[i] >= qn[i]) continue;
      const Item *it = &q[i][cur[i]];
      if (it->it[e[i]] > mx) mx = it->it[e[i]];
      if (++e[i] >= it->n) {
        e[i] = 0; cur[i]++;
        while (cur[i] < qn[i] && q[i][cur[i]].n == 0) cur[i]++;
        if (cur[i] >= qn[i]) live--; else sw = 1;
      }
    }
    a->wave += k->cEv + k->cIt * mx + (sw ? k->cRay : 0);
  }
}

#define NSCHED 11
static const char *kSchedNames[NSCHED] = {
  "nested (shipped)", "flat T=1", "park T=8", "park T=16", "park T=32",
  "nested, shadow rays wg-compacted", "flat T=1, shadow rays wg-compacted", "park T=16, shadow rays wg-compacted",
  "park T=16 primary + per-lane ray queue", "park T=16, hit pixels + rays wg-compacted",
  "nested, per-lane shadow-ray queue",
};

#define SIM_MAXWAVES (1 << 18)
static float g_waveEst[SIM_MAXWAVES];  /* lane-level cost of the wave's centre pixel (a cheap predictor) */
static float g_waveCost[SIM_MAXWAVES]; /* shipped-schedule cost of every simulated wave, in dispatch order */
static int g_waveIdx[SIM_MAXWAVES];
static int g_nWaves;
typedef struct {
  double wave[NSCHED], lane[NSCHED];
  double primWave[NSCHED], shadWave[NSCHED], normWave[NSCHED];
  double pixels, hits, rays, evals, iters;
} SimOut;

/* schedule one workgroup (4 waves × 64 lanes; px[w*64 + l]) under every schedule */
static void sim_workgroup(const PixTrace *px, int nLights, const Cost *k, SimOut *o) {
  Item prim[256], norm[256], shad[SIM_MAXL][256];
  for (int i = 0; i < 256; i++) {
    prim[i].it = px[i].primary; prim[i].n = px[i].nPrimary;
    norm[i].it = px[i].normal; norm[i].n = px[i].hit ? 4 : 0;
    for (int l = 0; l < nLights; l++) { shad[l][i].it = px[i].shadow[l]; shad[l][i].n = px[i].nShadow[l]; }
    o->pixels += 1; o->hits += px[i].hit;
    for (int e = 0; e < px[i].nPrimary; e++) { o->evals++; o->iters += px[i].primary[e]; }
    if (px[i].hit) for (int e = 0; e < 4; e++) { o->evals++; o->iters += px[i].normal[e]; }
    for (int l = 0; l < nLights; l++) { if (px[i].nShadow[l]) o->rays++; for (int e = 0; e < px[i].nShadow[l]; e++) { o->evals++; o->iters += px[i].shadow[l][e]; } }
  }
  const int Ts[5] = {0, 1, 8, 16, 32};
  for (int s = 0; s < NSCHED; s++) {
    Acc P = {0, 0}, N = {0, 0}, S = {0, 0};
    double hitCost = 0;
    /* ---- primary ---- */
    int primFlatT = (s >= 1 && s <= 4) ? Ts[s] : (s == 6 ? 1 : ((s >= 7 && s <= 9) ? 16 : 0));
    for (int w = 0; w < 4; w++) {
      if (!primFlatT) sched_nested(prim + 64 * w, 64, k, &P);
      else {
        const Item *q[64]; int qn[64];
        for (int i = 0; i < 64; i++) { q[i] = &prim[64 * w + i]; qn[i] = 1; }
        sched_flat(q, qn, 64, primFlatT, k, &P);
      }
    }
    /* ---- normals (+ per-hit shading cost) ---- */
    if (s == 9) { /* hit pixels compacted across the workgroup */
      Item hitList[256]; int nh = 0;
      for (int i = 0; i < 256; i++) if (norm[i].n) hitList[nh++] = norm[i];
      for (int b = 0; b < nh; b += 64) { sched_nested(hitList + b, nh - b < 64 ? nh - b : 64, k, &N); hitCost += k->cHit; }
    } else {
      for (int w = 0; w < 4; w++) {
        int any = 0;
        for (int i = 0; i < 64; i++) any |= norm[64 * w + i].n;
        if (any) { sched_nested(norm + 64 * w, 64, k, &N); hitCost += k->cHit; }
      }
    }
    /* ---- shadows ---- */
    if (s <= 4) {
      for (int w = 0; w < 4; w++)
        for (int l = 0; l < nLights; l++) {
          int any = 0;
          for (int i = 0; i < 64; i++) any |= shad[l][64 * w + i].n;
          if (!any) continue;
          S.wave += k->cRay;
          if (s == 0) sched_nested(shad[l] + 64 * w, 64, k, &S);
          else {
            const Item *q[64]; int qn[64];
            for (int i = 0; i < 64; i++) { q[i] = &shad[l][64 * w + i]; qn[i] = 1; }
            Acc t = {0, 0};
            sched_flat(q, qn, 64, Ts[s], k, &t);
            S.wave += t.wave; S.lane += t.lane - 0; /* lane cost counted inside */
          }
        }
    } else if (s == 5 || s == 6 || s == 7 || s == 9) {
      Item rays[256 * SIM_MAXL]; int nr = 0;
      /* compaction order: pixel-major, so that a pixel's rays and neighbouring pixels stay together */
      for (int i = 0; i < 256; i++) for (int l = 0; l < nLights; l++) if (shad[l][i].n) rays[nr++] = shad[l][i];
      for (int b = 0; b < nr; b += 64) {
        int m = nr - b < 64 ? nr - b : 64;
        S.wave += k->cRay;
        if (s == 5) sched_nested(rays + b, m, k, &S);
        else {
          const Item *q[64]; int qn[64];
          for (int i = 0; i < m; i++) { q[i] = &rays[b + i]; qn[i] = 1; }
          sched_flat(q, qn, m, s == 6 ? 1 : 16, k, &S);
        }
      }
    } else if (s == 10) {
      for (int w = 0; w < 4; w++) {
        Item lq[64][SIM_MAXL]; const Item *q[64]; int qn[64];
        for (int i = 0; i < 64; i++) { qn[i] = nLights; q[i] = lq[i]; for (int l = 0; l < nLights; l++) lq[i][l] = shad[l][64 * w + i]; }
        sched_nested_queue(q, qn, 64, k, &S);
      }
    } else if (s == 8) {
      for (int w = 0; w < 4; w++) {
        Item lq[64][SIM_MAXL]; const Item *q[64]; int qn[64];
        for (int i = 0; i < 64; i++) { qn[i] = nLights; q[i] = lq[i]; for (int l = 0; l < nLights; l++) lq[i][l] = shad[l][64 * w + i]; }
        sched_flat(q, qn, 64, 16, k, &S);
      }
    }
    /* sched_nested counts lane cost without cRay; add it for comparability */
    if (s == 0 || s == 5) for (int i = 0; i < 256; i++) for (int l = 0; l < nLights; l++) if (shad[l][i].n) S.lane += k->cRay;
    o->wave[s] += P.wave + N.wave + S.wave + hitCost;
    o->lane[s] += P.lane + N.lane + S.lane;
    o->primWave[s] += P.wave; o->normWave[s] += N.wave + hitCost; o->shadWave[s] += S.wave;
  }
}

/* Unified per-lane state machine at the MARCH-STEP level: every lane runs its own sequence primary march → (hit) 4 normal
 * taps → surface work (bump, material, light setup: cSurf, once) → shadow ray 1 → shadow ray 2 …; a wave trip evaluates
 * the next point of every lane that is in a march (cost cEv + cIt·max iterations); lanes that reach the surface work park
 * until >= T lanes wait there or nobody is marching, then the block runs once for all of them (cost cSurf); a lane that
 * starts a new shadow ray pays cRay in the trip it starts (once per trip if any lane does). */
static double wave_cost_unified(const PixTrace *px, int nLights, const Cost *k, int T, double cSurf) {
  int phase[64];   /* 0 primary, 1 normals, 2 parked for surface work, 3.. shadow ray (phase-3), 99 done */
  int e[64];
  double c = 0;
  int live = 0;
  for (int i = 0; i < 64; i++) { phase[i] = px[i].nPrimary > 0 ? 0 : 99; e[i] = 0; if (phase[i] != 99) live++; }
  while (live) {
    int mx = -1, marching = 0, parked = 0, newRay = 0;
    for (int i = 0; i < 64; i++) {
      const PixTrace *q = &px[i];
      if (phase[i] == 99) continue;
      if (phase[i] == 2) { parked++; continue; }
      marching++;
      const uint8_t *it; int n;
      if (phase[i] == 0) { it = q->primary; n = q->nPrimary; }
      else if (phase[i] == 1) { it = q->normal; n = 4; }
      else { it = q->shadow[phase[i] - 3]; n = q->nShadow[phase[i] - 3]; }
      if (it[e[i]] > mx) mx = it[e[i]];
      if (++e[i] >= n) { /* this march ends with this evaluation */
        e[i] = 0;
        if (phase[i] == 0) phase[i] = q->hit ? 1 : 99;
        else if (phase[i] == 1) phase[i] = 2;
        else {
          int l = phase[i] - 3 + 1;
          while (l < nLights && q->nShadow[l] == 0) l++;
          phase[i] = l < nLights ? 3 + l : 99;
          if (phase[i] != 99) newRay = 1;
        }
        if (phase[i] == 99) live--;
      }
    }
    if (marching) c += k->cEv + k->cIt * mx + (newRay ? k->cRay : 0);
    /* surface block */
    parked = 0; marching = 0;
    for (int i = 0; i < 64; i++) { if (phase[i] == 2) parked++; else if (phase[i] != 99) marching++; }
    if (parked && (parked >= T || marching == 0)) {
      c += cSurf;
      for (int i = 0; i < 64; i++)
        if (phase[i] == 2) {
          int l = 0;
          while (l < nLights && px[i].nShadow[l] == 0) l++;
          phase[i] = l < nLights ? 3 + l : 99;
          if (phase[i] == 99) live--;
        }
    }
  }
  return c;
}

/* shipped-schedule cost of one wave (64 pixels) */
static double wave_cost_shipped(const PixTrace *px, int nLights, const Cost *k) {
  Item it[64];
  Acc a = {0, 0};
  int anyHit = 0;
  for (int i = 0; i < 64; i++) { it[i].it = px[i].primary; it[i].n = px[i].nPrimary; anyHit |= px[i].hit; }
  sched_nested(it, 64, k, &a);
  if (anyHit) {
    for (int i = 0; i < 64; i++) { it[i].it = px[i].normal; it[i].n = px[i].hit ? 4 : 0; }
    sched_nested(it, 64, k, &a);
    a.wave += k->cHit;
    for (int l = 0; l < nLights; l++) {
      int any = 0;
      for (int i = 0; i < 64; i++) { it[i].it = px[i].shadow[l]; it[i].n = px[i].nShadow[l]; any |= it[i].n; }
      if (any) { a.wave += k->cRay; sched_nested(it, 64, k, &a); }
    }
  }
  return a.wave;
}
static double pixel_cost(const PixTrace *p, int nLights, const Cost *k) {
  double c = 0;
  Item it;
  it.it = p->primary; it.n = p->nPrimary; c += item_lane_cost(&it, k);
  if (p->hit) { it.it = p->normal; it.n = 4; c += item_lane_cost(&it, k) + k->cHit; }
  for (int l = 0; l < nLights; l++) { it.it = p->shadow[l]; it.n = p->nShadow[l]; c += item_lane_cost(&it, k); }
  return c;
}
/* cost of one wave when the shadow rays of all lights are spread over the lanes and marched together (<= 64 rays: one pass;
 * more: passes of 64 rays in pixel-major order) */
static double wave_cost_spread(const PixTrace *px, int nLights, const Cost *k, int onlyIfFit) {
  Item it[64];
  Acc a = {0, 0};
  int anyHit = 0;
  for (int i = 0; i < 64; i++) { it[i].it = px[i].primary; it[i].n = px[i].nPrimary; anyHit |= px[i].hit; }
  sched_nested(it, 64, k, &a);
  if (!anyHit) return a.wave;
  for (int i = 0; i < 64; i++) { it[i].it = px[i].normal; it[i].n = px[i].hit ? 4 : 0; }
  sched_nested(it, 64, k, &a);
  a.wave += k->cHit;
  Item rays[64 * SIM_MAXL];
  int nr = 0;
  for (int i = 0; i < 64; i++) for (int l = 0; l < nLights; l++) if (px[i].nShadow[l]) { rays[nr].it = px[i].shadow[l]; rays[nr].n = px[i].nShadow[l]; nr++; }
  if (nr <= 64 || !onlyIfFit) {
    for (int b = 0; b < nr; b += 64) { a.wave += k->cRay + 40; sched_nested(rays + b, nr - b < 64 ? nr - b : 64, k, &a); }
  } else {
    for (int l = 0; l < nLights; l++) {
      int any = 0;
      for (int i = 0; i < 64; i++) { it[i].it = px[i].shadow[l]; it[i].n = px[i].nShadow[l]; any |= it[i].n; }
      if (any) { a.wave += k->cRay; sched_nested(it, 64, k, &a); }
    }
  }
  return a.wave;
}
static double g_unified[4];
void sim_unified(double *out) { for (int i = 0; i < 4; i++) out[i] = g_unified[i]; }
static float g_waveCostSpread[2][SIM_MAXWAVES];
static float g_wavePrimCost[SIM_MAXWAVES];
static short g_waveHits[SIM_MAXWAVES], g_waveMaxPrimSteps[SIM_MAXWAVES], g_waveMaxShadSteps[SIM_MAXWAVES];
int sim_wave_detail(float *prim, short *hits, short *mp, short *ms, int max) {
  int n = g_nWaves < max ? g_nWaves : max;
  for (int i = 0; i < n; i++) { prim[i] = g_wavePrimCost[i]; hits[i] = g_waveHits[i]; mp[i] = g_waveMaxPrimSteps[i]; ms[i] = g_waveMaxShadSteps[i]; }
  return n;
}
int sim_wave_costs_spread(float *c0, float *c1, int max) {
  int n = g_nWaves < max ? g_nWaves : max;
  for (int i = 0; i < n; i++) { c0[i] = g_waveCostSpread[0][i]; c1[i] = g_waveCostSpread[1][i]; }
  return n;
}
int sim_wave_costs(float *cost, int *idx, float *est, int max) {
  int n = g_nWaves < max ? g_nWaves : max;
  for (int i = 0; i < n; i++) { cost[i] = g_waveCost[i]; idx[i] = g_waveIdx[i]; est[i] = g_waveEst[i]; }
  return n;
}

/* ---- tail deferral: a march whose live lanes drop to <= T hands its remaining evaluations to a pool; pools are
 * marched later in dense waves of 64 rays (a second kernel).  Pools are per simulator thread (arbitrary packing order). */
typedef struct { uint8_t it[SIM_MAXS + 8]; int n; } PoolItem;
typedef struct { PoolItem items[64]; int n; double wave, lane; long rays; } Pool;
static void pool_flush(Pool *p, const Cost *k) {
  if (!p->n) return;
  Item v[64];
  for (int i = 0; i < p->n; i++) { v[i].it = p->items[i].it; v[i].n = p->items[i].n; }
  Acc a = {0, 0};
  sched_nested(v, p->n, k, &a);
  p->wave += a.wave + k->cRay;
  p->lane += a.lane;
  p->n = 0;
}
static void pool_push(Pool *p, const uint8_t *it, int n, const Cost *k) {
  if (n <= 0) return;
  PoolItem *q = &p->items[p->n];
  memcpy(q->it, it, (size_t)n);
  q->n = n;
  p->rays++;
  if (++p->n == 64) pool_flush(p, k);
}
/* nested march of one wave with deferral; defer[i] is set for lanes whose remainder went to the pool */
static double nested_defer(const Item *it, int n, int T, Pool *pool, const Cost *k, int *defer) {
  double c = 0;
  int steps = 0;
  for (int i = 0; i < n; i++) if (it[i].n > steps) steps = it[i].n;
  for (int s = 0; s < steps; s++) {
    int mx = 0, act = 0;
    for (int i = 0; i < n; i++) if (it[i].n > s) { act++; if (it[i].it[s] > mx) mx = it[i].it[s]; }
    if (act <= T) {
      for (int i = 0; i < n; i++) if (it[i].n > s) { pool_push(pool, it[i].it + s, it[i].n - s, k); if (defer) defer[i] = 1; }
      c += 20; /* writing the continuation records */
      break;
    }
    c += k->cEv + k->cIt * mx;
  }
  return c;
}
typedef struct { double k1, k1max; Pool prim, norm, shad; } DeferAcc;
static void wave_cost_defer(const PixTrace *px, int nLights, int T, const Cost *k, DeferAcc *d) {
  Item it[64];
  int defer[64];
  memset(defer, 0, sizeof defer);
  int anyHit = 0;
  for (int i = 0; i < 64; i++) { it[i].it = px[i].primary; it[i].n = px[i].nPrimary; }
  double c = nested_defer(it, 64, T, &d->prim, k, defer);
  for (int i = 0; i < 64; i++) {
    if (defer[i] && px[i].hit) { /* the rest of a deferred pixel runs in the tail kernels */
      pool_push(&d->norm, px[i].normal, 4, k);
      for (int l = 0; l < nLights; l++) pool_push(&d->shad, px[i].shadow[l], px[i].nShadow[l], k);
    }
    if (!defer[i]) anyHit |= px[i].hit;
  }
  if (anyHit) {
    for (int i = 0; i < 64; i++) { it[i].it = px[i].normal; it[i].n = (px[i].hit && !defer[i]) ? 4 : 0; }
    Acc a = {0, 0};
    sched_nested(it, 64, k, &a);
    c += a.wave + k->cHit;
    for (int l = 0; l < nLights; l++) {
      int any = 0;
      for (int i = 0; i < 64; i++) { it[i].it = px[i].shadow[l]; it[i].n = defer[i] ? 0 : px[i].nShadow[l]; any |= it[i].n; }
      if (any) c += k->cRay + nested_defer(it, 64, T, &d->shad, k, NULL);
    }
  }
  d->k1 += c;
  if (c > d->k1max) d->k1max = c;
}
#define NDEFER 5
static const int kDeferT[NDEFER] = {0, 4, 8, 12, 16};
static double g_deferOut[NDEFER][6]; /* k1 total, k1 max wave, tail-kernel wave cost, rays deferred (prim, shad), pixels */
void sim_defer_results(double *out) { memcpy(out, g_deferOut, sizeof g_deferOut); }

void sim_histograms(double *byActive, double *byStep) {
  for (int i = 0; i < 65; i++) byActive[i] = g_costByActive[i];
  for (int i = 0; i < 257; i++) byStep[i] = g_costByStep[i];
}
/* Generic scene (any object table): every pixel's sequence of marches; prices, per 8×8 wave, (a) the shipped code-position
 * schedule — lanes meet at every march of the code (k-th march of every lane together; the normal taps ride with the march
 * before them) — and (b) a per-lane queue where every lane runs its own marches back to back (one evaluation per trip).
 * out: [0] lane-level evaluations, [1] wave-level evaluation trips (a), [2] trips (b), [3] waves, [4] pixels. */
/* optional: 2 doubles per 8×8 wave (whole-wave length, longest quarter), set by the caller before sim_generic */
static double *g_waveLen = NULL;
static double g_dumpAbove = 1e300;
static double *g_packLen = NULL;
static int g_packStride = 0;
void sim_set_pack_len_buffer(double *buf, int stride) { g_packLen = buf; g_packStride = stride; }
void sim_set_dump_above(double v) { g_dumpAbove = v; }
void sim_set_wave_len_buffer(double *buf) { g_waveLen = buf; }
int sim_generic(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
                const RmGlobals *g, const RmSettings *s, int W, int H, int stride, double *out, int threads) {
  RmResources none;
  memset(&none, 0, sizeof none);
  double laneEv = 0, tripsA = 0, tripsB = 0, tripsC = 0, nw = 0, npx = 0, tripsA2 = 0;
  double tripsD[3] = {0, 0, 0};
  const double cSurf = 3.3, cLight = 0.7; /* in evaluations of a Menger-class scene (≈300 instructions each) */
  const int gx = (W + 7) / 8, gy = (H + 7) / 8;
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads) reduction(+ : laneEv, tripsA, tripsB, tripsC, nw, npx, tripsA2, tripsD[:3])
  for (int b = 0; b < gx * gy; b++) {
    const int bx = b % gx, by = b / gx;
    if ((bx + 5 * by) % stride != 0) continue;
    Ctx c;
    c.cam = cam; c.objs = objs; c.numObjects = numObjects; c.lights = lights; c.numLights = numLights;
    c.g = *g; c.s = *s; c.nEval = c.nIter = c.nHit = 0; c.tex = NULL; c.numTex = 0; c.res = &none; c.W = W;
      rayPlanes(cam->invProjView, c.rayPlane);  /* since round 2 the primary rays are interpolated from the quad corners */
    SegTrace st[64];
    memset(st, 0, sizeof st);
    t_cullR2 = 0.0f;
    for (int l = 0; l < 64; l++) {
      int x = bx * 8 + (l % 8), y = by * 8 + l / 8;
      if (x >= W || y >= H) continue;
      float col[4], br[4];
      t_seg = &st[l]; t_pix = NULL;
      shadePixel(&c, x, y, W, H, col, br);
      t_seg = NULL;
      npx += 1;
    }
    int maxSeg = 0;
    for (int l = 0; l < 64; l++) if (st[l].n > maxSeg) maxSeg = st[l].n;
    for (int k = 0; k < maxSeg; k++) {
      int mx = 0;
      for (int l = 0; l < 64; l++) if (k < st[l].n && st[l].evals[k] > mx) mx = st[l].evals[k];
      tripsA += mx;
    }
    int mxTot = 0;
    for (int l = 0; l < 64; l++) {
      int tot = 0;
      for (int k = 0; k < st[l].n; k++) tot += st[l].evals[k];
      laneEv += tot;
      if (tot > mxTot) mxTot = tot;
    }
    tripsB += mxTot;
    /* (c) lanes still meet at every raymarch of the code (primary, bounces), but the shadow marches that follow it run as a
     * per-lane queue: group g of a lane = its g-th raymarch segment, then the sum of the shadow segments up to the next one */
    {
      int pos[64], done = 0;
      for (int l = 0; l < 64; l++) pos[l] = 0;
      while (!done) {
        int mxR = 0, mxS = 0;
        done = 1;
        for (int l = 0; l < 64; l++) {
          int k = pos[l];
          if (k >= st[l].n) continue;
          done = 0;
          if (st[l].evals[k] > mxR) mxR = st[l].evals[k]; /* the raymarch (kind 0) — or an orphan shadow segment */
          k++;
          int sh = 0;
          while (k < st[l].n && st[l].kind[k] == 1) sh += st[l].evals[k++];
          if (sh > mxS) mxS = sh;
          pos[l] = k;
        }
        tripsC += mxR + mxS;
      }
    }
    /* (e) the wave's own length under schedule (c) — whole, and cut into four 16-lane parts that run as waves of their own (what
     * a launcher could do with the tiles the previous frame measured heaviest): the part with the longest chain. */
    if (g_waveLen) {
      double best = 0;
      for (int part = -1; part < 4; part++) {
        int pos[64], done = 0;
        double len = 0;
        for (int l = 0; l < 64; l++) pos[l] = 0;
        while (!done) {
          int mxR = 0, mxS = 0;
          done = 1;
          for (int l = 0; l < 64; l++) {
            if (part >= 0 && (l >> 4) != part) continue;
            int k = pos[l];
            if (k >= st[l].n) continue;
            done = 0;
            if (st[l].evals[k] > mxR) mxR = st[l].evals[k];
            k++;
            int sh = 0;
            while (k < st[l].n && st[l].kind[k] == 1) sh += st[l].evals[k++];
            if (sh > mxS) mxS = sh;
            pos[l] = k;
          }
          len += mxR + mxS;
        }
        if (part < 0) g_waveLen[2 * b] = len;
        else if (len > best) best = len;
      }
      g_waveLen[2 * b + 1] = best;
      if (g_packLen) {
        /* (f) the shadow rays of one shading round PACKED: lanes still meet at every raymarch, but the shadow rays that follow it —
         * (pixel, light) pairs in pixel-major order — are dealt to the 64 lanes 64 at a time, one march per pass */
        int pos[64], done = 0;
        double len = 0;
        for (int l = 0; l < 64; l++) pos[l] = 0;
        while (!done) {
          int mxR = 0, nray = 0, passMax = 0;
          double sh = 0;
          done = 1;
          for (int l = 0; l < 64; l++) {
            int k = pos[l];
            if (k >= st[l].n) continue;
            done = 0;
            if (st[l].evals[k] > mxR) mxR = st[l].evals[k];
            k++;
            while (k < st[l].n && st[l].kind[k] == 1) {
              if (st[l].evals[k] > passMax) passMax = st[l].evals[k];
              if (++nray == 64) { sh += passMax; nray = 0; passMax = 0; }
              k++;
            }
            pos[l] = k;
          }
          if (nray) sh += passMax;
          len += mxR + sh;
        }
        g_packLen[b] = len;
        /* (g) the same rays as a WAVE-level queue: a lane that finishes its ray takes the next one of the list (list scheduling
         * on 64 lanes): the round's shadow phase lasts as long as its busiest lane */
        {
          int pos2[64], fin = 0;
          double len2 = 0;
          for (int l = 0; l < 64; l++) pos2[l] = 0;
          while (!fin) {
            int mxR = 0;
            double busy[64];
            for (int l = 0; l < 64; l++) busy[l] = 0;
            fin = 1;
            for (int l = 0; l < 64; l++) {
              int k = pos2[l];
              if (k >= st[l].n) continue;
              fin = 0;
              if (st[l].evals[k] > mxR) mxR = st[l].evals[k];
              k++;
              while (k < st[l].n && st[l].kind[k] == 1) {
                int best = 0;
                for (int q = 1; q < 64; q++) if (busy[q] < busy[best]) best = q;
                busy[best] += st[l].evals[k] + 2; /* + the hand-over */
                k++;
              }
              pos2[l] = k;
            }
            double mk = 0;
            for (int q = 0; q < 64; q++) if (busy[q] > mk) mk = busy[q];
            len2 += mxR + mk;
          }
          g_packLen[g_packStride + b] = len2;
        }
      }
      if (g_waveLen[2 * b] >= g_dumpAbove) {  /* the wave's longest lane: its marches (kind:evaluations) */
        int bl = 0, bt = -1;
        for (int l = 0; l < 64; l++) { int tot = 0; for (int k = 0; k < st[l].n; k++) tot += st[l].evals[k]; if (tot > bt) { bt = tot; bl = l; } }
#pragma omp critical
        {
          fprintf(stderr, "wave %d: %.0f trips; longest lane %d (%d evaluations):", b, g_waveLen[2 * b], bl, bt);
          for (int k = 0; k < st[bl].n; k++) fprintf(stderr, " %d:%d", st[bl].kind[k], st[bl].evals[k]);
          fprintf(stderr, "\n");
        }
      }
    }
    /* (d) per-lane queue WITH the price of the blocks between marches, in units of one evaluation: after a raymarch that is
     * followed by shadow rays (= a hit) the surface block (normal finalisation, bump, material: cSurf), after every shadow
     * march the light term + next-ray set-up (cLight).  A block runs once per trip in which any lane needs it; the surface
     * block is parked until >= T lanes wait or nobody marches.  The shipped schedule pays the same blocks once per code
     * position: added to (a) as tripsA2 for a like-for-like ratio. */
    for (int v = 0; v < 3; v++) {
      const int T = (v == 0) ? 1 : (v == 1 ? 8 : 16);
      int seg[64], e[64], parked[64], live = 0;
      double cost = 0;
      for (int l = 0; l < 64; l++) { seg[l] = 0; e[l] = 0; parked[l] = 0; if (st[l].n > 0) live++; else seg[l] = -1; }
      while (live) {
        int marching = 0, needLight = 0, nparked = 0;
        for (int l = 0; l < 64; l++) {
          if (seg[l] < 0 || parked[l]) { if (seg[l] >= 0) nparked++; continue; }
          marching++;
          if (st[l].evals[seg[l]] == 0 || ++e[l] >= st[l].evals[seg[l]]) { /* this march ends with this evaluation */
            const int k = seg[l];
            e[l] = 0;
            if (k + 1 >= st[l].n) { seg[l] = -1; live--; if (st[l].kind[k] == 1) needLight = 1; continue; }
            if (st[l].kind[k] == 1) needLight = 1;
            else if (st[l].kind[k + 1] == 1) { parked[l] = 1; nparked++; }
            seg[l] = k + 1;
          }
        }
        if (marching) cost += 1.0 + (needLight ? cLight : 0.0);
        marching = 0;
        for (int l = 0; l < 64; l++) if (seg[l] >= 0 && !parked[l]) marching++;
        if (nparked && (nparked >= T || marching == 0)) {
          cost += cSurf;
          for (int l = 0; l < 64; l++) parked[l] = 0;
        }
      }
      tripsD[v] += cost;
    }
    {
      /* blocks of the shipped schedule: one surface block per raymarch position where any lane hit, one light block per shadow position */
      int pos[64], done = 0;
      for (int l = 0; l < 64; l++) pos[l] = 0;
      for (int k = 0; k < maxSeg; k++) {
        int anyShadow = 0, anyHit = 0;
        for (int l = 0; l < 64; l++) if (k < st[l].n) { if (st[l].kind[k] == 1) anyShadow = 1; else if (k + 1 < st[l].n && st[l].kind[k + 1] == 1) anyHit = 1; }
        tripsA2 += (anyShadow ? cLight : 0.0) + (anyHit ? cSurf : 0.0);
      }
      (void)pos; (void)done;
    }
    nw += 1;
  }
  out[0] = laneEv; out[1] = tripsA; out[2] = tripsB; out[3] = nw; out[4] = npx; out[5] = tripsC;
  out[6] = tripsA2; out[7] = tripsD[0]; out[8] = tripsD[1]; out[9] = tripsD[2];
  return RM_OK;
}

int sim_num_schedules(void) { return NSCHED; }
const char *sim_schedule_name(int s) { return (s >= 0 && s < NSCHED) ? kSchedNames[s] : ""; }

/* Trace and price every `stride`-th workgroup (32×8 pixels) of the frame.  cost[6] = cIt, cEv, cItF, cEvF, cRay, cHit.
 * out: wave[NSCHED], lane[NSCHED], primWave, normWave, shadWave, then pixels, hits, rays, evals, iters. */
int sim_run(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
            const RmGlobals *g, const RmSettings *s, int W, int H, int stride, float cullRadius, const double *cost,
            double *out, int threads) {
  if (numObjects != 1 || objs[0].type != RM_MANDELBULB || numLights > SIM_MAXL) return RM_ERR_UNSUPPORTED;
  RmResources none;
  memset(&none, 0, sizeof none);
  const Cost k = {cost[0], cost[1], cost[2], cost[3], cost[4], cost[5], 0};
  Cost kh = k;
  kh.hist = 1;
  memset(g_costByActive, 0, sizeof g_costByActive);
  memset(g_costByStep, 0, sizeof g_costByStep);
  memset(g_deferOut, 0, sizeof g_deferOut);
  memset(g_unified, 0, sizeof g_unified);
  const int gx = (W + 31) / 32, gy = (H + 7) / 8;
  SimOut tot;
  memset(&tot, 0, sizeof tot);
  g_nWaves = 0;
#pragma omp parallel num_threads(threads)
  {
    SimOut loc;
    memset(&loc, 0, sizeof loc);
    PixTrace *px = (PixTrace *)malloc(256 * sizeof(PixTrace));
    DeferAcc *da = (DeferAcc *)calloc(NDEFER, sizeof(DeferAcc));
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < gx * gy; b++) {
      const int bx = b % gx, by = b / gx;
      if ((bx + 5 * by) % stride != 0) continue;
      Ctx c;
      c.cam = cam; c.objs = objs; c.numObjects = numObjects; c.lights = lights; c.numLights = numLights;
      c.g = *g; c.s = *s; c.nEval = c.nIter = c.nHit = 0; c.tex = NULL; c.numTex = 0; c.res = &none; c.W = W;
      rayPlanes(cam->invProjView, c.rayPlane);  /* since round 2 the primary rays are interpolated from the quad corners */
      memset(px, 0, 256 * sizeof(PixTrace));
      for (int w = 0; w < 4; w++)
        for (int l = 0; l < 64; l++) {
          int x = bx * 32 + w * 8 + (l % 8), y = by * 8 + l / 8;
          if (x >= W || y >= H) continue;
          float col[4], br[4];
          t_pix = &px[w * 64 + l]; t_light = 0; t_phase = 0; t_cullR2 = cullRadius * cullRadius;
          shadePixel(&c, x, y, W, H, col, br);
          t_pix = NULL;
        }
      sim_workgroup(px, numLights, &k, &loc);
      for (int w = 0; w < 4; w++) {
        const double c = wave_cost_shipped(px + 64 * w, numLights, &kh);
        for (int t = 0; t < NDEFER; t++) wave_cost_defer(px + 64 * w, numLights, kDeferT[t], &k, &da[t]);
        {
          static const int uT[4] = {1, 8, 16, 32};
          for (int t = 0; t < 4; t++) {
            const double cu = wave_cost_unified(px + 64 * w, numLights, &k, uT[t], k.cHit);
#pragma omp atomic
            g_unified[t] += cu;
          }
        }
        int slot;
#pragma omp atomic capture
        slot = g_nWaves++;
        if (slot < SIM_MAXWAVES) {
          g_waveCost[slot] = (float)c; g_waveIdx[slot] = b * 4 + w;
          {
            Item it[64]; Acc a = {0, 0}; int nh = 0, mp = 0, ms = 0;
            for (int i = 0; i < 64; i++) {
              const PixTrace *q = &px[64 * w + i];
              it[i].it = q->primary; it[i].n = q->nPrimary; nh += q->hit;
              if (q->nPrimary > mp) mp = q->nPrimary;
              for (int l = 0; l < numLights; l++) if (q->nShadow[l] > ms) ms = q->nShadow[l];
            }
            sched_nested(it, 64, &k, &a);
            g_wavePrimCost[slot] = (float)a.wave; g_waveHits[slot] = (short)nh; g_waveMaxPrimSteps[slot] = (short)mp; g_waveMaxShadSteps[slot] = (short)ms;
          }
          g_waveCostSpread[0][slot] = (float)wave_cost_spread(px + 64 * w, numLights, &k, 1);
          g_waveCostSpread[1][slot] = (float)wave_cost_spread(px + 64 * w, numLights, &k, 0);
          /* predictor: the 4 pixels around the tile centre (lanes (3,3),(4,3),(3,4),(4,4)) */
          g_waveEst[slot] = (float)(pixel_cost(&px[64 * w + 27], numLights, &k) + pixel_cost(&px[64 * w + 28], numLights, &k) +
                                    pixel_cost(&px[64 * w + 35], numLights, &k) + pixel_cost(&px[64 * w + 36], numLights, &k));
        }
      }
    }
    free(px);
    /* (da is read in the critical section below and leaked at exit of the region: a few KB per thread) */
#pragma omp critical
    {
      for (int t = 0; t < NDEFER; t++) {
        pool_flush(&da[t].prim, &k); pool_flush(&da[t].norm, &k); pool_flush(&da[t].shad, &k);
        g_deferOut[t][0] += da[t].k1;
        if (da[t].k1max > g_deferOut[t][1]) g_deferOut[t][1] = da[t].k1max;
        g_deferOut[t][2] += da[t].prim.wave + da[t].norm.wave + da[t].shad.wave;
        g_deferOut[t][3] += (double)da[t].prim.rays;
        g_deferOut[t][4] += (double)da[t].shad.rays;
      }
      for (int i = 0; i < NSCHED; i++) {
        tot.wave[i] += loc.wave[i]; tot.lane[i] += loc.lane[i]; tot.primWave[i] += loc.primWave[i];
        tot.normWave[i] += loc.normWave[i]; tot.shadWave[i] += loc.shadWave[i];
      }
      tot.pixels += loc.pixels; tot.hits += loc.hits; tot.rays += loc.rays; tot.evals += loc.evals; tot.iters += loc.iters;
    }
  }
  int o = 0;
  for (int i = 0; i < NSCHED; i++) out[o++] = tot.wave[i];
  for (int i = 0; i < NSCHED; i++) out[o++] = tot.lane[i];
  for (int i = 0; i < NSCHED; i++) out[o++] = tot.primWave[i];
  for (int i = 0; i < NSCHED; i++) out[o++] = tot.normWave[i];
  for (int i = 0; i < NSCHED; i++) out[o++] = tot.shadWave[i];
  out[o++] = tot.pixels; out[o++] = tot.hits; out[o++] = tot.rays; out[o++] = tot.evals; out[o++] = tot.iters;
  return RM_OK;
}

/* ---- cost-sorted lane assignment: the 1024 pixels of a 32x32 super-tile dealt to its 16 waves in the order
 * of a per-pixel cost — the pixel's own cost of THIS frame (the bound: a perfect predictor), or the cost of the pixel (dx, dy)
 * away (a previous frame under motion) — against the shipped 8x8 tiles.  out[0] = shipped wave cost, out[1] = sorted by own
 * cost, out[2] / out[3] = sorted by the cost 2 / 6 pixels away, out[4] = pixels. */
typedef struct { float c; int i; } SortKey;
static int cmp_key(const void *a, const void *b) {
  const float x = ((const SortKey *)a)->c, y = ((const SortKey *)b)->c;
  return x < y ? 1 : (x > y ? -1 : 0);
}
int sim_sorted(const RmCamera *cam, const RmObject *objs, int numObjects, const RmLight *lights, int numLights,
               const RmGlobals *g, const RmSettings *s, int W, int H, int stride, float cullRadius, const double *cost,
               double *out, int threads) {
  if (numObjects != 1 || objs[0].type != RM_MANDELBULB || numLights > SIM_MAXL) return RM_ERR_UNSUPPORTED;
  RmResources none;
  memset(&none, 0, sizeof none);
  const Cost k = {cost[0], cost[1], cost[2], cost[3], cost[4], cost[5], 0};
  const int gx = (W + 31) / 32, gy = (H + 31) / 32;
  double tot[5] = {0, 0, 0, 0, 0};
#pragma omp parallel num_threads(threads)
  {
    PixTrace *px = (PixTrace *)malloc(1024 * sizeof(PixTrace));
    PixTrace *grp = (PixTrace *)malloc(64 * sizeof(PixTrace));
    float *c = (float *)malloc(1024 * sizeof(float));
    SortKey *key = (SortKey *)malloc(1024 * sizeof(SortKey));
    double loc[5] = {0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 1)
    for (int b = 0; b < gx * gy; b++) {
      const int bx = b % gx, by = b / gx;
      if ((bx + 3 * by) % stride != 0) continue;
      Ctx cx;
      cx.cam = cam; cx.objs = objs; cx.numObjects = numObjects; cx.lights = lights; cx.numLights = numLights;
      cx.g = *g; cx.s = *s; cx.nEval = cx.nIter = cx.nHit = 0; cx.tex = NULL; cx.numTex = 0; cx.res = &none; cx.W = W;
      rayPlanes(cam->invProjView, cx.rayPlane);
      memset(px, 0, 1024 * sizeof(PixTrace));
      for (int ly = 0; ly < 32; ly++)
        for (int lx = 0; lx < 32; lx++) {
          const int x = bx * 32 + lx, y = by * 32 + ly;
          if (x >= W || y >= H) continue;
          float col[4], br[4];
          t_pix = &px[ly * 32 + lx]; t_light = 0; t_phase = 0; t_cullR2 = cullRadius * cullRadius;
          shadePixel(&cx, x, y, W, H, col, br);
          t_pix = NULL;
        }
      for (int i = 0; i < 1024; i++) c[i] = (float)pixel_cost(&px[i], numLights, &k);
      /* shipped: 16 tiles of 8x8 */
      for (int ty = 0; ty < 4; ty++)
        for (int tx = 0; tx < 4; tx++) {
          for (int l = 0; l < 64; l++) grp[l] = px[(ty * 8 + l / 8) * 32 + tx * 8 + (l % 8)];
          loc[0] += wave_cost_shipped(grp, numLights, &k);
        }
      static const int shift[3][2] = {{0, 0}, {2, 1}, {6, 3}};
      for (int v = 0; v < 3; v++) {
        for (int i = 0; i < 1024; i++) {
          int lx = (i % 32) + shift[v][0], ly = (i / 32) + shift[v][1];
          if (lx > 31) lx = 31;
          if (ly > 31) ly = 31;
          key[i].c = c[ly * 32 + lx]; key[i].i = i;
        }
        qsort(key, 1024, sizeof(SortKey), cmp_key);
        for (int w = 0; w < 16; w++) {
          for (int l = 0; l < 64; l++) grp[l] = px[key[w * 64 + l].i];
          loc[1 + v] += wave_cost_shipped(grp, numLights, &k);
        }
      }
      loc[4] += 1024;
    }
#pragma omp critical
    for (int i = 0; i < 5; i++) tot[i] += loc[i];
    free(px); free(grp); free(c); free(key);
  }
  for (int i = 0; i < 5; i++) out[i] = tot[i];
  return RM_OK;
}
