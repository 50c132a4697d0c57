/*
 * s2d_oracle.c -- CPU ORACLE for the reach_ball hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's `cpu_baseline` leg may load it.  The product path (gym-soccer-2d-env_amd/)
 * never imports, links or calls anything here and fails loudly without its HIP library.
 *
 * It is a plain-C scalar restatement of the reference's algorithm for the path
 * (SURVEY.md section 8a).  Two builds of this one source:
 *     -DS2DO_F64  REAL=double, libm sin/cos/atan2/exp  -- "what the reference's Python
 *                 (float64) computes"; pinned against tests/golden/ JSON files, which were
 *                 produced by running the reference's own ReachBallEnv methods.
 *     (default)   REAL=float, the deterministic fp32 math spec of DESIGN.md section 4
 *                 (explicit fmaf polynomials, correctly-rounded / and sqrt, no
 *                 contraction) -- what the HIP kernels must reproduce BIT FOR BIT.
 *
 * PARITY PINNING.  Rows A2-A5 (action map, observation, reward/done/result, reset
 * sampler) are pinned by the golden fixtures.  Rows S/P (dash / turn / stamina /
 * integrate / decay / collision = rcssserver's arithmetic) live in third-party binaries
 * that are NOT under /root/reference (clsframework/rcssserver, GitHub releases/latest,
 * no version pinned: scripts/download-rcssserver.sh:30) and the reference holds no test
 * or golden vector at that boundary: for those rows this oracle restates rcssserver's
 * published model and is "parity unpinned" against a real
 * rcssserver; it is pinned only by the hand-derived known answers of appendix B.
 *
 * Each function cites the reference file:line it follows.
 */
#include "s2d_oracle_common.h"

API void s2do_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { s2do_philox4x32_10_impl(ctr, key, out); }

API void s2do_sincos_deg(double deg, double *s, double *c) { REAL a, b; sincos_deg((REAL)deg, &a, &b); *s = a; *c = b; }
API double s2do_atan2_deg(double y, double x) { return atan2_deg((REAL)y, (REAL)x); }
API double s2do_exp(double x) { return exp_r((REAL)x); }
API double s2do_hypot(double x, double y) { return hypot2((REAL)x, (REAL)y); }
API double s2do_norm_deg(double d) { return norm_deg((REAL)d); }
API int s2do_real_bytes(void) { return (int)sizeof(REAL); }

/* ======================================================================================
 * Parameters rounded once to REAL.
 * ==================================================================================== */
typedef struct P {
  REAL half_l, half_w;
  REAL player_size, player_decay, player_rand, player_speed_max, player_accel_max, inertia_moment;
  REAL stamina_max, stamina_inc_max, stamina_capacity, extra_stamina;
  REAL recover_init, recover_dec_thr_value, recover_min, recover_dec;
  REAL effort_init, effort_dec_thr_value, effort_min, effort_dec, effort_inc_thr_value, effort_inc;
  REAL dash_power_rate, max_dash_power, min_dash_power, max_dash_angle, min_dash_angle;
  REAL dash_angle_step, side_dash_rate, back_dash_rate, max_moment, min_moment;
  REAL ball_size, ball_decay, ball_rand, ball_speed_max;
  REAL collision_vel_rate;
  /* derived once (host, double -> REAL) */
  REAL inv_half_l, inv_half_w, inv_dash_angle_step, player_accel_max2, player_speed_max2, ball_speed_max2;
  REAL rsum, rsum2, act_scale;
  /* task */
  int change_ball_position, change_ball_velocity, max_steps, use_continuous, n_actions, use_turning;
  REAL ball_position_x, ball_position_y, ball_speed, ball_direction, min_distance_to_ball;
  REAL travel_factor; /* (1 - 0.96^max_steps) / (1 - 0.96), reach_ball_env.py:207 */
  uint64_t seed;
  int64_t env_id_offset;
  int auto_reset, noise;
} P;

static void params_from_config(const S2DConfig *c, P *p) {
  const S2DServerParams *s = &c->sp;
  const S2DReachBallParams *t = &c->task;
  p->half_l = (REAL)s->pitch_half_length; p->half_w = (REAL)s->pitch_half_width;
  p->player_size = (REAL)s->player_size; p->player_decay = (REAL)s->player_decay;
  p->player_rand = (REAL)s->player_rand; p->player_speed_max = (REAL)s->player_speed_max;
  p->player_accel_max = (REAL)s->player_accel_max; p->inertia_moment = (REAL)s->inertia_moment;
  p->stamina_max = (REAL)s->stamina_max; p->stamina_inc_max = (REAL)s->stamina_inc_max;
  p->stamina_capacity = (REAL)s->stamina_capacity; p->extra_stamina = (REAL)s->extra_stamina;
  p->recover_init = (REAL)s->recover_init;
  p->recover_dec_thr_value = (REAL)(s->recover_dec_thr * s->stamina_max);
  p->recover_min = (REAL)s->recover_min; p->recover_dec = (REAL)s->recover_dec;
  p->effort_init = (REAL)s->effort_init;
  p->effort_dec_thr_value = (REAL)(s->effort_dec_thr * s->stamina_max);
  p->effort_min = (REAL)s->effort_min; p->effort_dec = (REAL)s->effort_dec;
  p->effort_inc_thr_value = (REAL)(s->effort_inc_thr * s->stamina_max);
  p->effort_inc = (REAL)s->effort_inc;
  p->dash_power_rate = (REAL)s->dash_power_rate; p->max_dash_power = (REAL)s->max_dash_power;
  p->min_dash_power = (REAL)s->min_dash_power; p->max_dash_angle = (REAL)s->max_dash_angle;
  p->min_dash_angle = (REAL)s->min_dash_angle; p->dash_angle_step = (REAL)s->dash_angle_step;
  p->side_dash_rate = (REAL)s->side_dash_rate; p->back_dash_rate = (REAL)s->back_dash_rate;
  p->max_moment = (REAL)s->max_moment; p->min_moment = (REAL)s->min_moment;
  p->ball_size = (REAL)s->ball_size; p->ball_decay = (REAL)s->ball_decay;
  p->ball_rand = (REAL)s->ball_rand; p->ball_speed_max = (REAL)s->ball_speed_max;
  p->collision_vel_rate = (REAL)s->collision_vel_rate;
  p->inv_half_l = (REAL)(1.0 / s->pitch_half_length); p->inv_half_w = (REAL)(1.0 / s->pitch_half_width);
  p->inv_dash_angle_step = s->dash_angle_step > 0 ? (REAL)(1.0 / s->dash_angle_step) : R(0.0);
  p->player_accel_max2 = p->player_accel_max * p->player_accel_max;
  p->player_speed_max2 = p->player_speed_max * p->player_speed_max;
  p->ball_speed_max2 = p->ball_speed_max * p->ball_speed_max;
  p->rsum = p->player_size + p->ball_size; p->rsum2 = p->rsum * p->rsum;
  p->act_scale = (REAL)(360.0 / (double)(t->action_space_size > 0 ? t->action_space_size : 1));
  p->change_ball_position = t->change_ball_position; p->change_ball_velocity = t->change_ball_velocity;
  p->max_steps = t->max_steps; p->use_continuous = t->use_continuous_action;
  p->n_actions = t->action_space_size; p->use_turning = t->use_turning;
  p->ball_position_x = (REAL)t->ball_position_x; p->ball_position_y = (REAL)t->ball_position_y;
  p->ball_speed = (REAL)t->ball_speed; p->ball_direction = (REAL)t->ball_direction;
  p->min_distance_to_ball = (REAL)t->min_distance_to_ball;
  p->travel_factor = (REAL)((1.0 - pow(t->reset_ball_decay, (double)t->max_steps)) / (1.0 - t->reset_ball_decay));
  p->seed = c->seed; p->env_id_offset = c->env_id_offset;
  p->auto_reset = c->auto_reset; p->noise = c->noise;
}

/* ======================================================================================
 * A2  ReachBallEnv.action_to_rpc_actions   reach_ball_env.py:53-85
 * `a` points at 1 (discrete/continuous) or 4 (turning) values; `u` is the uniform draw of
 * line 71.  step_number += 1 (line 55) is done by the caller (env_step).
 * ==================================================================================== */
static void action_map(const P *p, const REAL *a, REAL u, int *cmd, REAL *power, REAL *dir) {
  if (p->use_continuous) {
    if (p->use_turning) {
      REAL v[4];
      for (int k = 0; k < 4; ++k) v[k] = a[k] < R(-1.0) ? R(-1.0) : (a[k] > R(1.0) ? R(1.0) : a[k]); /* :64 */
      REAL turn_prob = v[0], turn_angle = v[1], dash_prob = v[2], dash_angle = v[3];             /* :65-68 */
      REAL e0 = exp_r(dash_prob), e1 = exp_r(turn_prob);                                         /* :69-70 */
      REAL p0 = e0 / (e0 + e1);
      if (u < p0) {            /* :71-75  (quirk: "turn" is chosen with the DASH probability) */
        *cmd = S2D_CMD_TURN; *power = R(0.0); *dir = turn_angle * R(180.0);
      } else {                 /* :76-79 */
        *cmd = S2D_CMD_DASH; *power = R(100.0); *dir = dash_angle * R(180.0);
      }
    } else {                   /* :81-82  not clipped */
      *cmd = S2D_CMD_DASH; *power = R(100.0); *dir = a[0] * R(180.0);
    }
  } else {                     /* :84-85  Python float modulo: result has the sign of the divisor */
#ifdef S2DO_F64
    REAL t = a[0] * R(360.0) / (REAL)p->n_actions;
    REAL m = R(fmod)(t, R(360.0));
    if (m < R(0.0)) m += R(360.0);
#else
    REAL t = a[0] * p->act_scale;                                 /* act_scale = 360 / n */
    REAL m = t - R(360.0) * floorf(t * 0.002777777777777778f);   /* floor-mod; identity for 0 <= a < n */
#endif
    *cmd = S2D_CMD_DASH; *power = R(100.0); *dir = m - R(180.0);
  }
}

API void s2do_action_map(const S2DConfig *cfg, const double *a, double u, int *cmd, double *power, double *dir) {
  P p; params_from_config(cfg, &p);
  REAL v[4] = {(REAL)a[0], 0, 0, 0};
  if (p.use_continuous && p.use_turning) for (int k = 1; k < 4; ++k) v[k] = (REAL)a[k];
  REAL pw, d;
  action_map(&p, v, (REAL)u, cmd, &pw, &d);
  *power = pw; *dir = d;
}

/* ======================================================================================
 * A3  ReachBallEnv.state_to_observation    reach_ball_env.py:87-111
 * ==================================================================================== */
static void observation(const P *p, REAL bx, REAL by, REAL bvx, REAL bvy, REAL px, REAL py, REAL body,
                        REAL *obs) {
  REAL ball_speed = hypot2(bvx, bvy);                      /* :91 */
  REAL ball_direction = atan2_deg(bvy, bvx);               /* :92 */
  REAL player_body = norm_deg(body);                       /* :94 */
  REAL player_to_ball = atan2_deg(by - py, bx - px);       /* :95 */
  REAL rel = norm_deg(player_to_ball - player_body);       /* :96 */
  obs[0] = DIVC(rel, R(180.0), 0.005555555555555556f);           /* :98 */
  obs[1] = DIVC(player_body, R(180.0), 0.005555555555555556f);
  obs[2] = DIVC(px, p->half_l, p->inv_half_l);                    /* 52.5 */
  obs[3] = DIVC(py, p->half_w, p->inv_half_w);                    /* 34.0 */
  obs[4] = DIVC(bx, p->half_l, p->inv_half_l);
  obs[5] = DIVC(by, p->half_w, p->inv_half_w);
  obs[6] = DIVC(ball_speed, R(3.0), 0.3333333333333333f);
  obs[7] = DIVC(ball_direction, R(360.0), 0.002777777777777778f);
  obs[8] = DIVC(bvx, R(3.0), 0.3333333333333333f);
  obs[9] = DIVC(bvy, R(3.0), 0.3333333333333333f);               /* :107 */
}

API void s2do_observation(const S2DConfig *cfg, const double *in7, double *obs10) {
  P p; params_from_config(cfg, &p);
  REAL o[10];
  observation(&p, (REAL)in7[0], (REAL)in7[1], (REAL)in7[2], (REAL)in7[3], (REAL)in7[4], (REAL)in7[5], (REAL)in7[6], o);
  for (int k = 0; k < 10; ++k) obs10[k] = o[k];
}

/* ======================================================================================
 * A4  ReachBallEnv.check_trainer_observation   reach_ball_env.py:113-161
 * ==================================================================================== */
static void check_trainer(const P *p, REAL bx, REAL by, REAL px, REAL py, REAL body, int step_number,
                          REAL *carry_dist, REAL *carry_angle, int *done, REAL *reward, int *result) {
  REAL dx = bx - px, dy = by - py;
  REAL distance_to_ball = hypot2(dx, dy);                               /* :121 */
  REAL player_body = norm_deg(body);                                    /* :122 */
  REAL ball_direction = atan2_deg(dy, dx);                              /* :123 */
  REAL diff = norm_deg(ball_direction - player_body);                   /* :124 */
  int d = 0, res = S2D_RESULT_NONE;
  REAL distance_reward = *carry_dist - distance_to_ball;                /* :130 */
  REAL r = distance_reward;                                             /* :128,131  0.0 + x */
  REAL angle_reward = DIVC(R(fabs)(norm_deg(*carry_angle)) - R(fabs)(diff), R(180.0), 0.005555555555555556f); /* :133 */
  r += angle_reward;
  if (distance_to_ball < p->min_distance_to_ball) { d = 1; r += R(10.0); res = S2D_RESULT_GOAL; }   /* :137-140 */
  if (R(fabs)(px) > p->half_l || R(fabs)(py) > p->half_w) { d = 1; r -= R(-10.0); res = S2D_RESULT_OUT; } /* :142-145 (+10, quirk) */
  if (step_number > p->max_steps) { d = 1; r -= R(5.0); res = S2D_RESULT_TIMEOUT; }                 /* :147-150 strict > */
  *carry_dist = distance_to_ball;                                       /* :158 */
  *carry_angle = diff;                                                  /* :159 */
  *done = d; *reward = r; *result = res;
}

API void s2do_check_trainer(const S2DConfig *cfg, const double *in5, int step_number, double *carry_dist,
                            double *carry_angle, int *done, double *reward, int *result) {
  P p; params_from_config(cfg, &p);
  REAL cd = (REAL)*carry_dist, ca = (REAL)*carry_angle, rw;
  check_trainer(&p, (REAL)in5[0], (REAL)in5[1], (REAL)in5[2], (REAL)in5[3], (REAL)in5[4], step_number, &cd, &ca, done, &rw, result);
  *carry_dist = cd; *carry_angle = ca; *reward = rw;
}

/* ======================================================================================
 * A5  ReachBallEnv.trainer_reset_actions + get_ball_velocity   reach_ball_env.py:170-218
 * The sampler is written against an abstract draw source so that the SAME code consumes
 * (a) the recorded draws of tests/golden/reset.json and (b) the engine's Philox stream.
 * ==================================================================================== */
typedef struct DrawSrc {
  /* (a) recorded */
  const double *rec; int n_rec, i_rec, underflow;
  /* (b) philox */
  uint64_t seed, gid; uint32_t cycle; int pos; /* pos: sequential word index in the RESET stream */
} DrawSrc;

static uint32_t src_word(DrawSrc *s) {
  uint32_t w[4];
  draw(s->seed, s->gid, s->cycle, ST_RESET, (uint32_t)(s->pos >> 2), w);
  return w[(s->pos++) & 3];
}
/* random.randint(lo, hi), inclusive */
static int src_randint(DrawSrc *s, int lo, int hi) {
  if (s->rec) { if (s->i_rec >= s->n_rec) { s->underflow = 1; return lo; } return (int)s->rec[s->i_rec++]; }
  return lo + rnd_below(src_word(s), (uint32_t)(hi - lo + 1));
}
/* random.random() */
static REAL src_random(DrawSrc *s) {
  if (s->rec) { if (s->i_rec >= s->n_rec) { s->underflow = 1; return 0; } return (REAL)s->rec[s->i_rec++]; }
  return rnd_u01(src_word(s));
}

#define S2DO_MAX_VEL_TRIES 255

typedef struct ResetDraw { REAL px, py, body, bx, by, bvx, bvy; int tries; } ResetDraw;

static void reset_sample(const P *p, DrawSrc *s, ResetDraw *o) {
  /* Philox layout: block 0 = {player x, player y, body, ball x}, block 1 = {ball y, speed0,
   * dir0, -}, later velocity attempts two per block.  Same ORDER as the reference's draws. */
  o->px = (REAL)src_randint(s, -50, 50);                 /* :173 */
  o->py = (REAL)src_randint(s, -30, 30);                 /* :174 */
  o->body = (REAL)src_randint(s, 0, 360);                /* :175 */
  if (p->change_ball_position) {                         /* :176-181 */
    o->bx = (REAL)src_randint(s, -50, 50);
    o->by = (REAL)src_randint(s, -30, 30);
  } else {
    o->bx = p->ball_position_x;
    o->by = p->ball_position_y;
  }
  o->tries = 0;
  if (p->change_ball_velocity) {                         /* :202-212 */
    REAL vx = R(0.0), vy = R(0.0);
    int ok = 0;
    for (int k = 0; k < S2DO_MAX_VEL_TRIES && !ok; ++k) {
      /* Philox word positions: try 0 = block 1 words {1,2}; try k>=1 = block 2+(k-1)/2,
       * words {0,1} (odd k) or {2,3} (even k): two candidates per Philox call */
      if (!s->rec) s->pos = k == 0 ? 5 : 4 * (2 + (k - 1) / 2) + 2 * ((k - 1) & 1);
      REAL speed = src_random(s) * R(3.0);               /* :204 */
      REAL dir = (REAL)src_randint(s, 0, 360);           /* :205 */
      if (s->rec && s->underflow) break;
      REAL sn, cs;
      sincos_deg(dir, &sn, &cs);
      vx = speed * cs; vy = speed * sn;                  /* :206 from_polar */
      REAL travel = speed * p->travel_factor;            /* :207 */
      REAL tx = o->bx + travel * cs, ty = o->by + travel * sn; /* :208-209 */
      o->tries = k + 1;
      if (R(fabs)(tx) <= p->half_l && R(fabs)(ty) <= p->half_w) ok = 1; /* :211 */
    }
    if (!ok) { vx = R(0.0); vy = R(0.0); }               /* cap reached (p < 1e-70): ball at rest */
    o->bvx = vx; o->bvy = vy;
  } else {                                               /* :213-216 */
    REAL sn, cs;
    sincos_deg(p->ball_direction, &sn, &cs);
    o->bvx = p->ball_speed * cs; o->bvy = p->ball_speed * sn;
  }
}

/* fixture entry: consume recorded draws, return the trainer actions' fields */
API int s2do_reset_from_draws(const S2DConfig *cfg, const double *draws, int n_draws, double *out7, int *n_used) {
  P p; params_from_config(cfg, &p);
  DrawSrc s; memset(&s, 0, sizeof s);
  s.rec = draws; s.n_rec = n_draws;
  ResetDraw o;
  reset_sample(&p, &s, &o);
  out7[0] = o.bx; out7[1] = o.by; out7[2] = o.bvx; out7[3] = o.bvy; out7[4] = o.px; out7[5] = o.py; out7[6] = o.body;
  *n_used = s.i_rec;
  return s.underflow ? -1 : 0;
}

/* ======================================================================================
 * S  rcssserver dynamics for one player + one ball (EXT).
 * ==================================================================================== */
typedef struct Env {
  REAL px, py, vx, vy, body, stamina, effort, recovery, capacity;
  REAL bx, by, bvx, bvy, prev_dist, prev_angle;
  int32_t step_number, cycle;
  uint32_t policy_step;   /* steps that consumed an in-engine POLICY / SELECT draw */
  uint32_t episode;       /* number of resets so far = index of the current episode (0 before the first reset) */
  int32_t last_tries;     /* velocity candidates the last reset drew (diagnostic for the distribution tests) */
} Env;

static REAL clampr(REAL v, REAL lo, REAL hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Player::dash -- appendix A "Dash(power, dir)" */
static void cmd_dash(const P *p, Env *e, REAL power, REAL dir, REAL *ax, REAL *ay) {
  power = clampr(power, p->min_dash_power, p->max_dash_power);
  dir = clampr(dir, p->min_dash_angle, p->max_dash_angle);
  if (p->dash_angle_step > R(0.0)) dir = p->dash_angle_step * R(rint)(DIVC(dir, p->dash_angle_step, p->inv_dash_angle_step));
  int back = power < R(0.0);
  REAL need = back ? power * R(-2.0) : power;
  REAL avail = e->stamina + p->extra_stamina;
  if (need > avail) need = avail;
  REAL st = e->stamina - need;
  e->stamina = st > R(0.0) ? st : R(0.0);
  power = back ? need / R(-2.0) : need;
  REAL ad = R(fabs)(dir);
  REAL dir_rate = ad > R(90.0)
      ? p->back_dash_rate - ((p->back_dash_rate - p->side_dash_rate) * (R(1.0) - DIVC(ad - R(90.0), R(90.0), 0.011111111111111112f)))
      : p->side_dash_rate + ((R(1.0) - p->side_dash_rate) * (R(1.0) - DIVC(ad, R(90.0), 0.011111111111111112f)));
  dir_rate = clampr(dir_rate, R(0.0), R(1.0));
  REAL acc = R(fabs)(e->effort * power * dir_rate * p->dash_power_rate);
  if (back) dir += R(180.0);
  REAL sn, cs;
  sincos_deg(norm_deg(e->body + dir), &sn, &cs);
  *ax = acc * cs;
  *ay = acc * sn;
}

/* Player::turn -- appendix A "Turn(moment)"; `noise_u` in [0,1) or 0.5 for none */
static void cmd_turn(const P *p, Env *e, REAL moment, REAL noise_u) {
  moment = clampr(moment, p->min_moment, p->max_moment);
  REAL speed = hypot2(e->vx, e->vy);
  REAL f = R(1.0);
  if (p->noise) f = R(1.0) + (noise_u * R(2.0) - R(1.0)) * p->player_rand;
  e->body = norm_deg(e->body + f * moment / (R(1.0) + p->inertia_moment * speed));
}

/* MPObject::_inc for one object (accel clamp, vel += accel, speed clamp, noise, pos += vel).
 * The magnitude tests compare squares (|a|^2 > max^2), the sqrt is taken only when clamping. */
static void obj_inc(REAL *x, REAL *y, REAL *vx, REAL *vy, int has_accel, REAL ax, REAL ay, REAL accel_max,
                    REAL accel_max2, REAL speed_max, REAL speed_max2, int noise, REAL rnd, uint32_t noise_word) {
  if (has_accel) {
    REAL a2 = sq2(ax, ay);
    if (a2 > accel_max2) { REAL k = accel_max / R(sqrt)(a2); ax *= k; ay *= k; }
    *vx += ax; *vy += ay;
  }
  REAL s2 = sq2(*vx, *vy);
  if (s2 > speed_max2) { REAL k = speed_max / R(sqrt)(s2); *vx *= k; *vy *= k; }
  if (noise) {   /* polar(U(0, rand * |vel|), U(-180, 180)): magnitude from the word's high half, direction = a WHOLE degree from its low half */
    REAL s = hypot2(*vx, *vy);
    REAL u_mag = (REAL)(noise_word >> 16) * R(1.52587890625e-05);
    REAL mag = u_mag * (rnd * s);
    REAL sn, cs;
    sincos_deg((REAL)(int32_t)(((noise_word & 0xffffu) * 360u) >> 16) - R(180.0), &sn, &cs);
    *vx += mag * cs; *vy += mag * sn;
  }
  *x += *vx; *y += *vy;
}

/* Stadium::collisions for the single player-ball pair: symmetric separation about the
 * midpoint to exact contact, both velocities *= collision_vel_rate (-0.1). */
static void collide(const P *p, Env *e) {
  REAL dx = e->bx - e->px, dy = e->by - e->py;
  REAL d2 = sq2(dx, dy);
  REAL rsum = p->rsum;
  if (d2 < p->rsum2) {
    REAL d = R(sqrt)(d2);
    REAL ux, uy;
    if (d > R(0.0)) { ux = dx / d; uy = dy / d; } else { ux = R(1.0); uy = R(0.0); }
    REAL mx = (e->px + e->bx) * R(0.5), my = (e->py + e->by) * R(0.5);
    REAL h = rsum * R(0.5);
    e->px = mx - ux * h; e->py = my - uy * h;
    e->bx = mx + ux * h; e->by = my + uy * h;
    e->vx *= p->collision_vel_rate; e->vy *= p->collision_vel_rate;
    e->bvx *= p->collision_vel_rate; e->bvy *= p->collision_vel_rate;
  }
}

/* Player::updateStamina -- appendix A */
static void update_stamina(const P *p, Env *e) {
  if (e->stamina <= p->recover_dec_thr_value) {
    if (e->recovery > p->recover_min) { REAL r = e->recovery - p->recover_dec; e->recovery = r > p->recover_min ? r : p->recover_min; }
  }
  if (e->stamina <= p->effort_dec_thr_value) {
    if (e->effort > p->effort_min) { REAL f = e->effort - p->effort_dec; e->effort = f > p->effort_min ? f : p->effort_min; }
  }
  if (e->stamina >= p->effort_inc_thr_value) {
    if (e->effort < p->effort_init) { REAL f = e->effort + p->effort_inc; e->effort = f < p->effort_init ? f : p->effort_init; }
  }
  REAL inc = e->recovery * p->stamina_inc_max;
  REAL room = p->stamina_max - e->stamina;
  if (inc > room) inc = room;
  if (p->stamina_capacity >= R(0.0)) { if (inc > e->capacity) inc = e->capacity; }
  e->stamina += inc;
  if (e->stamina > p->stamina_max) e->stamina = p->stamina_max;
  if (p->stamina_capacity >= R(0.0)) { REAL c = e->capacity - inc; e->capacity = c > R(0.0) ? c : R(0.0); }
}

/* One simulator cycle (appendix A "Per-cycle order", play_on, referee off in coach mode:
 * soccer_2d_env.py:363-366 starts rcssserver with coach=true and no coach_w_referee). */
/* Noise draws (player_rand / ball_rand / turn) of a commanded cycle are keyed by the env's policy_step
 * (stream NOISE), those of the command-less cycle a reset consumes by the reset's own key (stream NOISE_RESET,
 * the RESET stream's counter word): like the policy draws they must not depend on whether an episode ended
 * earlier, and the state a reset leaves behind is a function of (env id, key) alone. */
/* One Philox block serves the movement noise of TWO commanded cycles: block 0 at counter k >> 1, words (x, y) for even k, (z, w)
 * for odd k -- one word per object (player, ball); the reset's command-less cycle takes (x, y) of the block at its own key.
 * The turn noise is word x of block 1 at counter k. */
static void sim_cycle(const P *p, Env *e, uint64_t gid, int cmd, REAL power, REAL dir, uint32_t noise_ctr,
                      uint32_t noise_stream) {
  uint32_t nz[4] = {0, 0, 0, 0}, nz2[4] = {0, 0, 0, 0};
  uint32_t wp = 0, wb = 0;
  if (p->noise) {
    const int paired = noise_stream == ST_NOISE;
    draw(p->seed, gid, paired ? noise_ctr >> 1 : noise_ctr, noise_stream, 0, nz);
    if (cmd == S2D_CMD_TURN) draw(p->seed, gid, noise_ctr, noise_stream, 1, nz2);
    const int odd = paired && (noise_ctr & 1u);
    wp = odd ? nz[2] : nz[0]; wb = odd ? nz[3] : nz[1];
  }
  REAL ax = R(0.0), ay = R(0.0);
  if (cmd == S2D_CMD_DASH) cmd_dash(p, e, power, dir, &ax, &ay);
  else if (cmd == S2D_CMD_TURN) cmd_turn(p, e, dir, rnd_u01(nz2[0]));
  obj_inc(&e->px, &e->py, &e->vx, &e->vy, cmd == S2D_CMD_DASH, ax, ay, p->player_accel_max, p->player_accel_max2,
          p->player_speed_max, p->player_speed_max2, p->noise, p->player_rand, wp);
  obj_inc(&e->bx, &e->by, &e->bvx, &e->bvy, 0, R(0.0), R(0.0), R(0.0), R(0.0),
          p->ball_speed_max, p->ball_speed_max2, p->noise, p->ball_rand, wb);
  collide(p, e);
  e->cycle = (int32_t)((uint32_t)e->cycle + 1u);        /* referee: time += 1 (wraps, never UB) */
  e->vx *= p->player_decay; e->vy *= p->player_decay;   /* _turn */
  e->bvx *= p->ball_decay; e->bvy *= p->ball_decay;
  update_stamina(p, e);
}

/* A5 + A6: trainer moves ball & player, recovers the player, then ONE cycle passes with
 * no body command (soccer_2d_env.py:186-197); carry is seeded (reach_ball_env.py:166). */
static void env_reset(const P *p, Env *e, uint64_t gid, REAL *obs) {
  DrawSrc s; memset(&s, 0, sizeof s);
  /* RESET stream counter word: the index of the episode the reset starts (1, 2, ...).  The state a reset leaves
   * behind is therefore a function of (env id, episode index) alone -- independent of how long earlier episodes
   * lasted -- so an engine may prepare any number of future episodes ahead of the simulation. */
  e->episode += 1u;
  s.seed = p->seed; s.gid = gid; s.cycle = e->episode;
  ResetDraw o;
  reset_sample(p, &s, &o);
  e->last_tries = o.tries;
  e->step_number = 0;                                    /* :172 */
  e->bx = o.bx; e->by = o.by; e->bvx = o.bvx; e->bvy = o.bvy;      /* (move (ball) x y 0 vx vy) */
  e->px = o.px; e->py = o.py; e->body = norm_deg(o.body); e->vx = R(0.0); e->vy = R(0.0); /* (move (player..)) */
  e->stamina = p->stamina_max; e->recovery = p->recover_init;     /* (recover) */
  e->effort = p->effort_init; e->capacity = p->stamina_capacity;
  sim_cycle(p, e, gid, S2D_CMD_NONE, R(0.0), R(0.0), s.cycle, ST_NOISE_RESET);   /* noise keyed like the sample */
  observation(p, e->bx, e->by, e->bvx, e->bvy, e->px, e->py, e->body, obs);
  int d, res; REAL rw;
  check_trainer(p, e->bx, e->by, e->px, e->py, e->body, e->step_number, &e->prev_dist, &e->prev_angle, &d, &rw, &res);
}

/* ======================================================================================
 * Vectorised engine (struct-of-arrays host memory), mirror of the s2d_* C ABI.
 * ==================================================================================== */
typedef struct S2DOEngine {
  P p; S2DConfig cfg; int64_t n;
  Env *env;
  REAL *obs, *terminal_obs, *reward, *action_dir;
  uint8_t *done, *result, *action_cmd;
  unsigned long long stats[8];
} S2DOEngine;

API S2DOEngine *s2do_create(const S2DConfig *cfg, int64_t n) {
  if (!cfg || n <= 0) return NULL;
  S2DOEngine *h = (S2DOEngine *)calloc(1, sizeof *h);
  params_from_config(cfg, &h->p);
  h->cfg = *cfg; h->n = n;
  h->env = (Env *)calloc((size_t)n, sizeof(Env));
  h->obs = (REAL *)calloc((size_t)n * 10, sizeof(REAL));
  h->terminal_obs = (REAL *)calloc((size_t)n * 10, sizeof(REAL));
  h->reward = (REAL *)calloc((size_t)n, sizeof(REAL));
  h->action_dir = (REAL *)calloc((size_t)n, sizeof(REAL));
  h->done = (uint8_t *)calloc((size_t)n, 1);
  h->result = (uint8_t *)calloc((size_t)n, 1);
  h->action_cmd = (uint8_t *)calloc((size_t)n, 1);
  for (int64_t i = 0; i < n; ++i) {                      /* initial state = after (recover) */
    Env *e = &h->env[i];
    e->stamina = h->p.stamina_max; e->recovery = h->p.recover_init;
    e->effort = h->p.effort_init; e->capacity = h->p.stamina_capacity;
  }
  return h;
}
API void s2do_destroy(S2DOEngine *h) {
  if (!h) return;
  free(h->env); free(h->obs); free(h->terminal_obs); free(h->reward); free(h->action_dir);
  free(h->done); free(h->result); free(h->action_cmd); free(h);
}

API void s2do_reset(S2DOEngine *h, const uint8_t *mask) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < h->n; ++i) {
    if (mask && !mask[i]) continue;
    env_reset(&h->p, &h->env[i], (uint64_t)(h->p.env_id_offset + i), &h->obs[i * 10]);
    h->reward[i] = R(0.0); h->done[i] = 0; h->result[i] = 0;
  }
}

/* In-engine policy randomness (not part of the reference: its policies live in the caller).
 * Keyed by the env's `policy_step` k, NOT by the cycle, so that the draws of step t do not depend
 * on whether an episode ended before t:  discrete / 1-D continuous policies use word k&3 of the
 * POLICY block at counter k>>2 (one Philox call serves four steps), the 4-D turning policy uses
 * the four words of POLICY block 1 at counter k; the turn/dash selection uniform of
 * reach_ball_env.py:71 is word k&3 of the SELECT block at counter k>>2. */
static uint32_t quad_word(uint64_t seed, uint64_t gid, uint32_t k, uint32_t stream) {
  uint32_t w[4];
  draw(seed, gid, k >> 2, stream, 0, w);
  return w[k & 3u];
}

/* decode the action of env i for this cycle (caller layouts of include/s2d.h, or the
 * in-engine random policy) */
static void fetch_action(const S2DOEngine *h, int64_t i, const void *actions, int kind, uint32_t k,
                         REAL a[4], void *rollout_action_out) {
  const P *p = &h->p;
  uint64_t gid = (uint64_t)(p->env_id_offset + i);
  a[0] = a[1] = a[2] = a[3] = R(0.0);
  switch (kind) {
    case S2D_ACT_DISCRETE_I32: a[0] = (REAL)((const int32_t *)actions)[i]; break;
    case S2D_ACT_DISCRETE_I64: a[0] = (REAL)((const int64_t *)actions)[i]; break;
    case S2D_ACT_CONTINUOUS: a[0] = (REAL)((const float *)actions)[i]; break;
    case S2D_ACT_TURNING: for (int j = 0; j < 4; ++j) a[j] = (REAL)((const float *)actions)[i * 4 + j]; break;
    case S2D_ACT_COMMAND: break;                         /* decoded by step_one */
    default: {
      if (!p->use_continuous) a[0] = (REAL)rnd_below(quad_word(p->seed, gid, k, ST_POLICY), (uint32_t)p->n_actions);
      else if (!p->use_turning) a[0] = rnd_u01(quad_word(p->seed, gid, k, ST_POLICY)) * R(2.0) - R(1.0);
      else {
        uint32_t w[4];
        draw(p->seed, gid, k, ST_POLICY, 1, w);
        for (int j = 0; j < 4; ++j) a[j] = rnd_u01(w[j]) * R(2.0) - R(1.0);
      }
    }
  }
  if (rollout_action_out) {
    if (!p->use_continuous) ((int32_t *)rollout_action_out)[i] = (int32_t)a[0];
    else if (!p->use_turning) ((float *)rollout_action_out)[i] = (float)a[0];
    else for (int j = 0; j < 4; ++j) ((float *)rollout_action_out)[i * 4 + j] = (float)a[j];
  }
}

/* A1  Soccer2DEnv.step   soccer_2d_env.py:226-269 */
/* S2D_ACT_COMMAND: command word -> S2D_CMD_* (include/s2d.h: the nearest of FREEZE -1, NONE 0, DASH 1, TURN 2; anything else,
 * NaN included, is no command) */
static int command_code(float x) {
  int c = S2D_CMD_NONE;
  if (x >= -1.5f && x <= -0.5f) c = S2D_CMD_FREEZE;
  if (x >= 0.5f && x < 1.5f) c = S2D_CMD_DASH;
  if (x >= 1.5f && x < 2.5f) c = S2D_CMD_TURN;
  return c;
}

static void step_one(S2DOEngine *h, int64_t i, const void *actions, int kind, void *rollout_action_out,
                     unsigned long long local_stats[4]) {
  const P *p = &h->p;
  Env *e = &h->env[i];
  uint64_t gid = (uint64_t)(p->env_id_offset + i);
  REAL a[4];
  const int turning = p->use_continuous && p->use_turning;
  if (kind == S2D_ACT_COMMAND && command_code(((const float *)actions)[4 * i]) == S2D_CMD_FREEZE) return;   /* not part of this cycle */
  fetch_action(h, i, actions, kind, e->policy_step, a, rollout_action_out);
  e->step_number += 1;                                   /* reach_ball_env.py:55 */
  const uint32_t k = e->policy_step;
  REAL u = R(0.0);
  if (turning) u = rnd_u01(quad_word(p->seed, gid, k, ST_SELECT));
  if (turning || kind == S2D_ACT_RANDOM || p->noise) e->policy_step += 1u;
  int cmd; REAL power, dir;
  if (kind == S2D_ACT_COMMAND) {                         /* a decoded PlayerAction body command, executed as it is (server.py:64) */
    const float *c4 = (const float *)actions + 4 * i;
    cmd = command_code(c4[0]); power = (REAL)c4[1]; dir = (REAL)c4[2];
  } else
  action_map(p, a, u, &cmd, &power, &dir);               /* :238 */
  h->action_cmd[i] = (uint8_t)cmd; h->action_dir[i] = dir;
  sim_cycle(p, e, gid, cmd, power, dir, k, ST_NOISE);    /* rcssserver cycle; trainer forces PlayOn :242 */
  REAL *obs = &h->obs[i * 10];
  observation(p, e->bx, e->by, e->bvx, e->bvy, e->px, e->py, e->body, obs);   /* :249 */
  int d, res; REAL rw;
  check_trainer(p, e->bx, e->by, e->px, e->py, e->body, e->step_number, &e->prev_dist, &e->prev_angle, &d, &rw, &res); /* :266 */
  h->reward[i] = rw; h->done[i] = (uint8_t)d; h->result[i] = (uint8_t)res;
  local_stats[res] += 1;
  if (d && p->auto_reset) {
    memcpy(&h->terminal_obs[i * 10], obs, 10 * sizeof(REAL));
    env_reset(p, e, gid, obs);
  }
}

API void s2do_step(S2DOEngine *h, const void *actions, int kind) {
  unsigned long long s1 = 0, s2 = 0, s3 = 0;
#pragma omp parallel for schedule(static) reduction(+ : s1, s2, s3)
  for (int64_t i = 0; i < h->n; ++i) {
    unsigned long long ls[4] = {0, 0, 0, 0};
    step_one(h, i, actions, kind, NULL, ls);
    s1 += ls[1]; s2 += ls[2]; s3 += ls[3];
  }
  h->stats[0] += (unsigned long long)h->n; h->stats[1] += s1; h->stats[2] += s2; h->stats[3] += s3;
}

/* T steps; outputs time-major like S2DRollout but in REAL for obs/reward */
API void s2do_rollout(S2DOEngine *h, int n_steps, const void *actions, int kind, REAL *obs, void *action,
                      REAL *reward, uint8_t *done, uint8_t *result) {
  const P *p = &h->p;
  size_t aw = !p->use_continuous ? 4 : (p->use_turning ? 16 : 4);
  size_t in_w = kind == S2D_ACT_DISCRETE_I64 ? 8 : (kind == S2D_ACT_TURNING ? 16 : 4);
  for (int t = 0; t < n_steps; ++t) {
    const void *act_t = actions ? (const char *)actions + (size_t)t * (size_t)h->n * in_w : NULL;
    void *act_out = action ? (char *)action + (size_t)t * (size_t)h->n * aw : NULL;
    unsigned long long s1 = 0, s2 = 0, s3 = 0;
#pragma omp parallel for schedule(static) reduction(+ : s1, s2, s3)
    for (int64_t i = 0; i < h->n; ++i) {
      unsigned long long ls[4] = {0, 0, 0, 0};
      step_one(h, i, act_t, kind, act_out, ls);
      s1 += ls[1]; s2 += ls[2]; s3 += ls[3];
    }
    h->stats[0] += (unsigned long long)h->n; h->stats[1] += s1; h->stats[2] += s2; h->stats[3] += s3;
    size_t off = (size_t)t * (size_t)h->n;
    if (obs) memcpy(obs + off * 10, h->obs, (size_t)h->n * 10 * sizeof(REAL));
    if (reward) memcpy(reward + off, h->reward, (size_t)h->n * sizeof(REAL));
    if (done) memcpy(done + off, h->done, (size_t)h->n);
    if (result) memcpy(result + off, h->result, (size_t)h->n);
  }
}

/* field ids for s2do_get_state: order of S2DBuffers' state pointers */
API int s2do_get_state(const S2DOEngine *h, int field, double *out) {
  for (int64_t i = 0; i < h->n; ++i) {
    const Env *e = &h->env[i];
    double v;
    switch (field) {
      case 0: v = e->px; break; case 1: v = e->py; break; case 2: v = e->vx; break; case 3: v = e->vy; break;
      case 4: v = e->body; break; case 5: v = e->stamina; break; case 6: v = e->effort; break;
      case 7: v = e->recovery; break; case 8: v = e->capacity; break; case 9: v = e->bx; break;
      case 10: v = e->by; break; case 11: v = e->bvx; break; case 12: v = e->bvy; break;
      case 13: v = e->prev_dist; break; case 14: v = e->prev_angle; break;
      case 15: v = e->step_number; break; case 16: v = e->cycle; break; case 17: v = e->policy_step; break;
      case 18: v = e->episode; break;
      default: return -1;
    }
    out[i] = v;
  }
  return 0;
}
/* overwrite one env's state (tests: hand-placed scenarios) -- 19 values in field order */
API int s2do_set_env(S2DOEngine *h, int64_t i, const double *v17) {
  if (i < 0 || i >= h->n) return -1;
  Env *e = &h->env[i];
  e->px = (REAL)v17[0]; e->py = (REAL)v17[1]; e->vx = (REAL)v17[2]; e->vy = (REAL)v17[3]; e->body = (REAL)v17[4];
  e->stamina = (REAL)v17[5]; e->effort = (REAL)v17[6]; e->recovery = (REAL)v17[7]; e->capacity = (REAL)v17[8];
  e->bx = (REAL)v17[9]; e->by = (REAL)v17[10]; e->bvx = (REAL)v17[11]; e->bvy = (REAL)v17[12];
  e->prev_dist = (REAL)v17[13]; e->prev_angle = (REAL)v17[14];
  e->step_number = (int32_t)v17[15]; e->cycle = (int32_t)v17[16]; e->policy_step = (uint32_t)v17[17];
  e->episode = (uint32_t)v17[18];
  return 0;
}
API void s2do_last_tries(const S2DOEngine *h, int32_t *out) { for (int64_t i = 0; i < h->n; ++i) out[i] = h->env[i].last_tries; }
API const REAL *s2do_obs(const S2DOEngine *h) { return h->obs; }
API const REAL *s2do_terminal_obs(const S2DOEngine *h) { return h->terminal_obs; }
API const REAL *s2do_reward(const S2DOEngine *h) { return h->reward; }
API const REAL *s2do_action_dir(const S2DOEngine *h) { return h->action_dir; }
API const uint8_t *s2do_done(const S2DOEngine *h) { return h->done; }
API const uint8_t *s2do_result(const S2DOEngine *h) { return h->result; }
API const uint8_t *s2do_action_cmd(const S2DOEngine *h) { return h->action_cmd; }
API const unsigned long long *s2do_stats(const S2DOEngine *h) { return h->stats; }
