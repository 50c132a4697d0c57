/*
 * s2d_gtc_oracle.c -- CPU ORACLE of the GoToCenter surrogate task.  TEST INFRASTRUCTURE ONLY.
 * Restates GoToCenterEnv of the reference (python_sample_soccer_env.py:17-255).  PINNED: the fp64 build is
 * checked to 1e-12 against tests/golden/gtc.json, which tests/golden/make_golden.py produces by RUNNING the
 * reference's own GoToCenterEnv class in the build container (resets with injected draws, step sequences of
 * all four action modes incl. the turn / use_turn 4-output mode with the selection uniform injected, every
 * Out / Goal / Timeout priority case) -- tests/test_gtc.py::test_oracle_pinned_by_reference_fixture.
 * fp32 build = deterministic spec shared with the HIP kernel; fp64 build = numpy semantics.
 */
#include "s2d_oracle_common.h"
#include "../include/s2d_gtc.h"
#define API __attribute__((visibility("default")))

typedef struct GP { REAL x_min, x_max, y_min, y_max, min_dist; int max_steps, continuous, auto_reset, turn, use_turn, adim; uint64_t seed; int64_t off; } GP;
typedef struct GEnv { REAL x, y, body, prev_distance, prev_angle_diff; int32_t step_count, episode; } GEnv;
typedef struct GEngine { GP p; int64_t n; GEnv *e; REAL *obs, *terminal_obs, *reward; uint8_t *done, *result; unsigned long long stats[8]; } GEngine;

/* wrap_angle_deg :18-25  ((a + 180) % 360) - 180 with Python's floor-mod -> [-180, 180) */
static REAL wrap_deg(REAL a) {
  REAL t = a + R(180.0);
#ifdef S2DO_F64
  REAL m = fmod(t, 360.0); if (m < 0.0) m += 360.0;
#else
  REAL m = t - R(360.0) * floorf(t * 0.002777777777777778f);
#endif
  return m - R(180.0);
}
/* angle_to_point_deg :27-37 towards the centre (0,0) */
static REAL angle_to_center(REAL x, REAL y) { return wrap_deg(atan2_deg(R(0.0) - y, R(0.0) - x)); }
/* diff_angle_deg_abs :39-44 */
static REAL diff_abs(REAL a, REAL b) { return R(fabs)(wrap_deg(a - b)); }

static void g_obs(const GEnv *e, REAL *o) {                       /* _get_obs :235-255 */
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
static REAL clip1(REAL v) { return v < R(-1.0) ? R(-1.0) : (v > R(1.0) ? R(1.0) : v); }
/* a[0..adim): the action row; u: the selection uniform of :151 (used by the use_turn mode only) */
static void g_step(const GP *p, GEnv *e, const REAL *a, REAL u, REAL *reward, int *done, int *result) {   /* step :136-234 */
  REAL dash_r, turn_r = R(0.0);
  int dash_selected = 1, turn_selected = 0;
  if (p->turn && p->continuous) {                                 /* :142-158 */
    dash_r = clip1(a[0]);                                         /* :143-144 */
    if (p->use_turn) {
      turn_r = clip1(a[1]);                                       /* :146 */
      REAL dash_p = clip1(a[2]), turn_p = clip1(a[3]);            /* :147-148 */
      REAL et = exp_r(turn_p), ed = exp_r(dash_p);                /* :149-150  p = softmax([turn_p, dash_p]) */
      REAL p0 = et / (et + ed);
      turn_selected = u < p0;                                     /* :151  p[0] is the TURN probability here */
      dash_selected = !turn_selected;                             /* :152 */
    }                                                             /* else :153-158: always dash with actions[0] */
  } else if (p->continuous) dash_r = clip1(a[0]);                 /* :159-162 */
  else dash_r = (DIVC((REAL)(int)a[0], R(16.0), 0.0625f) - R(0.5)) * R(2.0);                       /* :163-166 */
  if (dash_selected) {
    REAL dir = wrap_deg(e->body + dash_r * R(180.0));            /* :169 */
    REAL sn, cs;
    sincos_deg(dir, &sn, &cs);                                    /* :173-175 */
    e->x += cs; e->y += sn;                                       /* :178-179 */
  }
  if (turn_selected) e->body = wrap_deg(e->body + turn_r * R(180.0));   /* :181-183 */
  REAL d = hypot2(e->x, e->y);                                    /* :186 */
  REAL adiff = diff_abs(e->body, angle_to_center(e->x, e->y));    /* :187-188 */
  REAL r = (e->prev_distance - d) + DIVC(e->prev_angle_diff - adiff, R(180.0), 0.005555555555555556f);   /* :191-194 */
  e->step_count += 1;                                             /* :196 */
  int dn = 0, res = S2D_RESULT_NONE;
  if (e->x < p->x_min || e->x > p->x_max || e->y < p->y_min || e->y > p->y_max) { dn = 1; r -= R(10.0); res = S2D_RESULT_OUT; }   /* :203-207 */
  else if (d < p->min_dist) { dn = 1; r += R(10.0); res = S2D_RESULT_GOAL; }                           /* :209-212 */
  else if (e->step_count >= p->max_steps) { dn = 1; r -= R(5.0); res = S2D_RESULT_TIMEOUT; }            /* :214-217 */
  e->prev_distance = d; e->prev_angle_diff = adiff;               /* :223-224 */
  *reward = r; *done = dn; *result = res;
}

API GEngine *s2dgo_create(const S2DGtcConfig *c, int64_t n) {
  GEngine *h = (GEngine *)calloc(1, sizeof *h);
  h->p = (GP){(REAL)c->x_min, (REAL)c->x_max, (REAL)c->y_min, (REAL)c->y_max, (REAL)c->min_distance_to_center,
              c->max_steps, c->continuous, c->auto_reset, c->turn, c->use_turn,
              (c->turn && c->continuous) ? c->actor_out_size : 1, c->seed, c->env_id_offset};
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
/* actions: double[n][adim] or NULL (random policy: POLICY stream keyed (gid, episode, step_count));
 * select_u: double[n] or NULL (SELECT stream, same key) -- the uniform of :151, injectable for the fixtures */
API void s2dgo_step_u(GEngine *h, const double *actions, const double *select_u) {
  const int adim = h->p.adim;
  for (int64_t i = 0; i < h->n; ++i) {
    GEnv *e = &h->e[i]; uint64_t gid = (uint64_t)(h->p.off + i);
    REAL a[4] = {R(0.0), R(0.0), R(0.0), R(0.0)}, u = R(0.0);
    uint32_t w[4];
    if (actions) { for (int k = 0; k < adim; ++k) a[k] = (REAL)actions[i * adim + k]; }
    else {
      draw(h->p.seed, gid, (uint32_t)e->episode, ST_POLICY, (uint32_t)e->step_count, w);
      if (h->p.continuous) { for (int k = 0; k < adim; ++k) a[k] = rnd_u01(w[k]) * R(2.0) - R(1.0); }
      else a[0] = (REAL)rnd_below(w[0], 16);
    }
    if (h->p.turn && h->p.continuous && h->p.use_turn) {
      if (select_u) u = (REAL)select_u[i];
      else { draw(h->p.seed, gid, (uint32_t)e->episode, ST_SELECT, (uint32_t)e->step_count, w); u = rnd_u01(w[0]); }
    }
    REAL rw; int dn, res;
    g_step(&h->p, e, a, u, &rw, &dn, &res);
    g_obs(e, &h->obs[i * 4]);
    h->reward[i] = rw; h->done[i] = (uint8_t)dn; h->result[i] = (uint8_t)res;
    h->stats[0]++; h->stats[res] += res ? 1 : 0;
    if (dn && h->p.auto_reset) { memcpy(&h->terminal_obs[i * 4], &h->obs[i * 4], 4 * sizeof(REAL)); g_reset(&h->p, e, gid); g_obs(e, &h->obs[i * 4]); }
  }
}
API void s2dgo_step(GEngine *h, const double *actions) { s2dgo_step_u(h, actions, NULL); }
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
