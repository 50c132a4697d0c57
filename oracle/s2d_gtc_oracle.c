/*
 * s2d_gtc_oracle.c -- CPU ORACLE of the GoToCenter surrogate task.  TEST INFRASTRUCTURE ONLY.
 * Restates GoToCenterEnv of the reference (python_sample_soccer_env.py:17-255), read as text:
 * the module cannot be imported (it needs stable_baselines3 and runs argparse at import,
 * :355-372), so this restatement is pinned by hand-derived known answers only
 * (tests/test_gtc.py) -- "parity unpinned" against an executed reference.
 * fp32 build = deterministic spec shared with the HIP kernel; fp64 build = numpy semantics.
 */
#include "s2d_oracle_common.h"
#include "../include/s2d_gtc.h"
#define API __attribute__((visibility("default")))

typedef struct GP { REAL x_min, x_max, y_min, y_max, min_dist; int max_steps, continuous, auto_reset; uint64_t seed; int64_t off; } GP;
typedef struct GEnv { REAL x, y, body, prev_distance, prev_angle_diff; int32_t step_count, episode; } GEnv;
typedef struct GEngine { GP p; int64_t n; GEnv *e; REAL *obs, *terminal_obs, *reward; uint8_t *done, *result; unsigned long long stats[8]; } GEngine;

/* wrap_angle_deg :17-24  ((a + 180) % 360) - 180 with Python's floor-mod -> [-180, 180) */
static REAL wrap_deg(REAL a) {
  REAL t = a + R(180.0);
#ifdef S2DO_F64
  REAL m = fmod(t, 360.0); if (m < 0.0) m += 360.0;
#else
  REAL m = t - R(360.0) * floorf(t * 0.002777777777777778f);
#endif
  return m - R(180.0);
}
/* angle_to_point_deg :26-36 towards the centre (0,0) */
static REAL angle_to_center(REAL x, REAL y) { return wrap_deg(atan2_deg(R(0.0) - y, R(0.0) - x)); }
/* diff_angle_deg_abs :38-44 */
static REAL diff_abs(REAL a, REAL b) { return R(fabs)(wrap_deg(a - b)); }

static void g_obs(const GEnv *e, REAL *o) {                       /* _get_obs :236-255 */
  REAL diff = wrap_deg(angle_to_center(e->x, e->y) - e->body);
  o[0] = DIVC(diff, R(180.0), 0.005555555555555556f); o[1] = DIVC(e->body, R(180.0), 0.005555555555555556f);
  o[2] = DIVC(e->x, R(52.5), 0.01904761904761905f); o[3] = DIVC(e->y, R(34.0), 0.029411764705882353f);
}
static void g_reset(const GP *p, GEnv *e, uint64_t gid) {          /* reset :115-134 */
  uint32_t w[4];
  draw(p->seed, gid, (uint32_t)e->episode, ST_RESET, 0, w);
  e->x = p->x_min + rnd_u01(w[0]) * (p->x_max - p->x_min);        /* :119 */
  e->y = p->y_min + rnd_u01(w[1]) * (p->y_max - p->y_min);        /* :120 */
  e->body = R(-180.0) + rnd_u01(w[2]) * R(360.0);                 /* :122 */
  e->step_count = 0; e->episode += 1;
  e->prev_distance = hypot2(e->x, e->y);                          /* :127 */
  e->prev_angle_diff = diff_abs(e->body, angle_to_center(e->x, e->y));   /* :128-129 */
}
static void g_step(const GP *p, GEnv *e, REAL action, REAL *reward, int *done, int *result) {   /* step :136-234 */
  REAL dash_r;
  if (p->continuous) dash_r = action < R(-1.0) ? R(-1.0) : (action > R(1.0) ? R(1.0) : action);   /* :169-170 */
  else dash_r = (DIVC((REAL)(int)action, R(16.0), 0.0625f) - R(0.5)) * R(2.0);                       /* :173-174 */
  REAL dir = wrap_deg(e->body + dash_r * R(180.0));              /* :177 */
  REAL sn, cs;
  sincos_deg(dir, &sn, &cs);                                      /* :181-183 */
  e->x += cs; e->y += sn;                                         /* :186-187 */
  REAL d = hypot2(e->x, e->y);                                    /* :194 */
  REAL adiff = diff_abs(e->body, angle_to_center(e->x, e->y));    /* :195-196 */
  REAL r = (e->prev_distance - d) + DIVC(e->prev_angle_diff - adiff, R(180.0), 0.005555555555555556f);   /* :199-202 */
  e->step_count += 1;                                             /* :204 */
  int dn = 0, res = S2D_RESULT_NONE;
  if (e->x < p->x_min || e->x > p->x_max || e->y < p->y_min || e->y > p->y_max) { dn = 1; r -= R(10.0); res = S2D_RESULT_OUT; }   /* :211-215 */
  else if (d < p->min_dist) { dn = 1; r += R(10.0); res = S2D_RESULT_GOAL; }                           /* :217-220 */
  else if (e->step_count >= p->max_steps) { dn = 1; r -= R(5.0); res = S2D_RESULT_TIMEOUT; }            /* :222-225 */
  e->prev_distance = d; e->prev_angle_diff = adiff;               /* :231-232 */
  *reward = r; *done = dn; *result = res;
}

API GEngine *s2dgo_create(const S2DGtcConfig *c, int64_t n) {
  GEngine *h = (GEngine *)calloc(1, sizeof *h);
  h->p = (GP){(REAL)c->x_min, (REAL)c->x_max, (REAL)c->y_min, (REAL)c->y_max, (REAL)c->min_distance_to_center,
              c->max_steps, c->continuous, c->auto_reset, c->seed, c->env_id_offset};
  h->n = n; h->e = (GEnv *)calloc((size_t)n, sizeof(GEnv));
  h->obs = (REAL *)calloc((size_t)n * 4, sizeof(REAL)); h->terminal_obs = (REAL *)calloc((size_t)n * 4, sizeof(REAL));
  h->reward = (REAL *)calloc((size_t)n, sizeof(REAL)); h->done = (uint8_t *)calloc((size_t)n, 1); h->result = (uint8_t *)calloc((size_t)n, 1);
  return h;
}
API void s2dgo_destroy(GEngine *h) { if (h) { free(h->e); free(h->obs); free(h->terminal_obs); free(h->reward); free(h->done); free(h->result); free(h); } }
API void s2dgo_reset(GEngine *h, const uint8_t *mask) {
  for (int64_t i = 0; i < h->n; ++i) if (!mask || mask[i]) {
    g_reset(&h->p, &h->e[i], (uint64_t)(h->p.off + i)); g_obs(&h->e[i], &h->obs[i * 4]);
    h->reward[i] = 0; h->done[i] = 0; h->result[i] = 0;
  }
}
/* actions: double[n] or NULL (random policy: POLICY stream keyed (gid, episode, step_count)) */
API void s2dgo_step(GEngine *h, const double *actions) {
  for (int64_t i = 0; i < h->n; ++i) {
    GEnv *e = &h->e[i]; uint64_t gid = (uint64_t)(h->p.off + i);
    REAL a;
    if (actions) a = (REAL)actions[i];
    else {
      uint32_t w[4];
      draw(h->p.seed, gid, (uint32_t)e->episode, ST_POLICY, (uint32_t)e->step_count, w);
      a = h->p.continuous ? rnd_u01(w[0]) * R(2.0) - R(1.0) : (REAL)rnd_below(w[0], 16);
    }
    REAL rw; int dn, res;
    g_step(&h->p, e, a, &rw, &dn, &res);
    g_obs(e, &h->obs[i * 4]);
    h->reward[i] = rw; h->done[i] = (uint8_t)dn; h->result[i] = (uint8_t)res;
    h->stats[0]++; h->stats[res] += res ? 1 : 0;
    if (dn && h->p.auto_reset) { memcpy(&h->terminal_obs[i * 4], &h->obs[i * 4], 4 * sizeof(REAL)); g_reset(&h->p, e, gid); g_obs(e, &h->obs[i * 4]); }
  }
}
API void s2dgo_set(GEngine *h, int64_t i, double x, double y, double body, int step_count) {
  GEnv *e = &h->e[i]; e->x = (REAL)x; e->y = (REAL)y; e->body = (REAL)body; e->step_count = step_count;
  e->prev_distance = hypot2(e->x, e->y); e->prev_angle_diff = diff_abs(e->body, angle_to_center(e->x, e->y));
  g_obs(e, &h->obs[i * 4]);
}
/* field: 0 x 1 y 2 body 3 prev_distance 4 prev_angle_diff 5 step_count 6 episode 7 reward 8 done 9 result */
API void s2dgo_get(const GEngine *h, int field, double *out) {
  for (int64_t i = 0; i < h->n; ++i) {
    const GEnv *e = &h->e[i];
    double v = field == 0 ? e->x : field == 1 ? e->y : field == 2 ? e->body : field == 3 ? e->prev_distance : field == 4 ? e->prev_angle_diff
             : field == 5 ? e->step_count : field == 6 ? e->episode : field == 7 ? h->reward[i] : field == 8 ? h->done[i] : h->result[i];
    out[i] = v;
  }
}
API void s2dgo_obs(const GEngine *h, int terminal, double *out) {
  const REAL *src = terminal ? h->terminal_obs : h->obs;
  for (int64_t i = 0; i < h->n * 4; ++i) out[i] = src[i];
}
API const unsigned long long *s2dgo_stats(const GEngine *h) { return h->stats; }
