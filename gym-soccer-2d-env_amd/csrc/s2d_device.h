// s2d_device.h -- device-side building blocks of the MI355X (gfx950) 2D-soccer engine:
// the deterministic fp32 math spec, Philox4x32-10 (section 5) and
// the one-player/one-ball dynamics of the reach_ball path.
//
// fp32 contract: every operation below is an IEEE-754 binary32 add / mul / fma / div /
// sqrt / rint / compare executed in the written order (the library is compiled with
// -ffp-contract=off; hipcc's default correctly-rounded divide and sqrt stay on).  The
// results are therefore a pure function of the inputs -- the same on every CU, for every
// launch geometry and every shard layout -- and equal, bit for bit, to any other
// implementation of the same spec.
//
// Reference semantics (file:line under /root/reference) are cited per function; the
// rcssserver arithmetic (dash/turn/stamina/integrate/collide) is EXT (SURVEY.md appx A).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/s2d.h"

#define S2D_DEV __device__ __forceinline__

// ------------------------------------------------------------------ parameters (kernarg)
struct S2DDevParams {
  float half_l, half_w;
  float player_size, player_decay, player_rand, player_speed_max, player_accel_max, inertia_moment;
  float stamina_max, stamina_inc_max, stamina_capacity, extra_stamina;
  float recover_init, recover_dec_thr_value, recover_min, recover_dec;
  float effort_init, effort_dec_thr_value, effort_min, effort_dec, effort_inc_thr_value, effort_inc;
  float dash_power_rate, max_dash_power, min_dash_power, max_dash_angle, min_dash_angle;
  float dash_angle_step, side_dash_rate, back_dash_rate, max_moment, min_moment;
  float ball_size, ball_decay, ball_rand, ball_speed_max;
  float collision_vel_rate;
  // task (ReachBallEnv kwargs, reach_ball_env.py:26-36)
  float ball_position_x, ball_position_y, ball_speed, ball_direction, min_distance_to_ball;
  float travel_factor;  // (1 - 0.96^max_steps) / (1 - 0.96), reach_ball_env.py:207
  int change_ball_position, change_ball_velocity, max_steps, use_continuous, n_actions, use_turning;
  int auto_reset, noise;
  uint32_t seed_lo, seed_hi;
  uint32_t gid_lo, gid_hi;  // env_id_offset
};

// ------------------------------------------------------------------ Philox4x32-10
enum { S2D_ST_RESET = 0, S2D_ST_POLICY = 1, S2D_ST_SELECT = 2, S2D_ST_NOISE = 3 };

struct U4 { uint32_t x, y, z, w; };

S2D_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
// ctr = { gid_lo, gid_hi, cycle, (stream << 16) | block }, key = seed
S2D_DEV U4 s2d_draw(const S2DDevParams& p, uint32_t gid_lo, uint32_t gid_hi, uint32_t cycle, uint32_t stream,
                    uint32_t block) {
  return philox4x32_10(gid_lo, gid_hi, cycle, (stream << 16) | block, p.seed_lo, p.seed_hi);
}
S2D_DEV int rnd_below(uint32_t w, uint32_t span) { return (int)__umulhi(w, span); }
S2D_DEV float rnd_u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-8f; }

// ------------------------------------------------------------------ fp32 math spec
S2D_DEV void sincos_deg(float deg, float& s, float& c) {
  float q = rintf(deg * 0.011111111111111112f);
  float r = fmaf(-q, 90.0f, deg);
  float x = r * 0.017453292519943295f;
  float z = x * x;
  float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(z, ps, -1.6666654611e-1f);
  ps = fmaf(x * z, ps, x);
  float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(z, pc, 4.166664568298827e-2f);
  pc = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  int n = ((int)q) & 3;
  float ss = (n & 1) ? pc : ps;
  float cc = (n & 1) ? ps : pc;
  s = (n & 2) ? -ss : ss;
  c = ((n + 1) & 2) ? -cc : cc;
}
S2D_DEV float atan2_deg(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  if (mx == 0.0f) return 0.0f;
  float t = mn / mx;
  float base = 0.0f;
  if (t > 0.41421356237f) {
    t = (t - 1.0f) / (t + 1.0f);
    base = 45.0f;
  }
  float z = t * t;
  float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
  p = fmaf(z, p, 1.99777106478e-1f);
  p = fmaf(z, p, -3.33329491539e-1f);
  float a = fmaf(p * z, t, t);
  a = fmaf(a, 57.29577951308232f, base);
  if (ay > ax) a = 90.0f - a;
  if (x < 0.0f) a = 180.0f - a;
  if (y < 0.0f) a = -a;
  return a;
}
S2D_DEV float hypot2(float x, float y) { return sqrtf(fmaf(x, x, y * y)); }
S2D_DEV float exp_spec(float x) {
  float k = rintf(x * 1.44269504088896341f);
  float r = fmaf(-k, 0.693359375f, x);
  r = fmaf(-k, -2.12194440e-4f, r);
  float z = r * r;
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float y = fmaf(p, z, r) + 1.0f;
  return ldexpf(y, (int)k);
}
// pyrusgeom AngleDeg normalisation (reach_ball_env.py:94-96, 122-124 rely on it)
S2D_DEV float norm_deg(float d) {
  if (d < -360.0f || 360.0f < d) d = fmodf(d, 360.0f);
  if (d < -180.0f) d += 360.0f;
  if (d > 180.0f) d -= 360.0f;
  return d;
}
S2D_DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------------------ one env in registers
struct Env {
  float px, py, vx, vy, body, stamina, effort, recovery, capacity;
  float bx, by, bvx, bvy, prev_dist, prev_angle;
  int step_number, cycle;
};
enum {  // SoA field order == S2DBuffers state pointers
  F_PX, F_PY, F_VX, F_VY, F_BODY, F_STAMINA, F_EFFORT, F_RECOVERY, F_CAPACITY,
  F_BX, F_BY, F_BVX, F_BVY, F_PREV_DIST, F_PREV_ANGLE, F_STEP, F_CYCLE, F_COUNT
};

S2D_DEV void env_load(Env& e, const float* __restrict__ S, int64_t stride, int64_t i) {
  e.px = S[F_PX * stride + i]; e.py = S[F_PY * stride + i];
  e.vx = S[F_VX * stride + i]; e.vy = S[F_VY * stride + i];
  e.body = S[F_BODY * stride + i];
  e.stamina = S[F_STAMINA * stride + i]; e.effort = S[F_EFFORT * stride + i];
  e.recovery = S[F_RECOVERY * stride + i]; e.capacity = S[F_CAPACITY * stride + i];
  e.bx = S[F_BX * stride + i]; e.by = S[F_BY * stride + i];
  e.bvx = S[F_BVX * stride + i]; e.bvy = S[F_BVY * stride + i];
  e.prev_dist = S[F_PREV_DIST * stride + i]; e.prev_angle = S[F_PREV_ANGLE * stride + i];
  e.step_number = __float_as_int(S[F_STEP * stride + i]);
  e.cycle = __float_as_int(S[F_CYCLE * stride + i]);
}
S2D_DEV void env_store(const Env& e, float* __restrict__ S, int64_t stride, int64_t i) {
  S[F_PX * stride + i] = e.px; S[F_PY * stride + i] = e.py;
  S[F_VX * stride + i] = e.vx; S[F_VY * stride + i] = e.vy;
  S[F_BODY * stride + i] = e.body;
  S[F_STAMINA * stride + i] = e.stamina; S[F_EFFORT * stride + i] = e.effort;
  S[F_RECOVERY * stride + i] = e.recovery; S[F_CAPACITY * stride + i] = e.capacity;
  S[F_BX * stride + i] = e.bx; S[F_BY * stride + i] = e.by;
  S[F_BVX * stride + i] = e.bvx; S[F_BVY * stride + i] = e.bvy;
  S[F_PREV_DIST * stride + i] = e.prev_dist; S[F_PREV_ANGLE * stride + i] = e.prev_angle;
  S[F_STEP * stride + i] = __int_as_float(e.step_number);
  S[F_CYCLE * stride + i] = __int_as_float(e.cycle);
}

// ------------------------------------------------------------------ A2 action map
// ReachBallEnv.action_to_rpc_actions, reach_ball_env.py:53-85 (step_number++ by caller)
S2D_DEV void action_map(const S2DDevParams& p, float a0, float a1, float a2, float a3, float u, int& cmd,
                        float& power, float& dir) {
  if (p.use_continuous) {
    if (p.use_turning) {
      float turn_prob = clampf(a0, -1.0f, 1.0f), turn_angle = clampf(a1, -1.0f, 1.0f);   // :64-68
      float dash_prob = clampf(a2, -1.0f, 1.0f), dash_angle = clampf(a3, -1.0f, 1.0f);
      float e0 = exp_spec(dash_prob), e1 = exp_spec(turn_prob);                          // :69-70
      float p0 = e0 / (e0 + e1);
      if (u < p0) { cmd = S2D_CMD_TURN; power = 0.0f; dir = turn_angle * 180.0f; }       // :71-75 (quirk kept)
      else { cmd = S2D_CMD_DASH; power = 100.0f; dir = dash_angle * 180.0f; }            // :76-79
    } else {
      cmd = S2D_CMD_DASH; power = 100.0f; dir = a0 * 180.0f;                             // :81-82, not clipped
    }
  } else {                                                                               // :84-85
    float t = a0 * 360.0f / (float)p.n_actions;
    float m = fmodf(t, 360.0f);
    if (m < 0.0f) m += 360.0f;
    cmd = S2D_CMD_DASH; power = 100.0f; dir = m - 180.0f;
  }
}

// ------------------------------------------------------------------ A3 + A4 fused
// state_to_observation (reach_ball_env.py:87-111) and check_trainer_observation (:113-161)
// read the same full-state truth, so the player->ball vector, its angle and the distance
// are computed once and shared.
struct ObsOut { float o[S2D_OBS_DIM]; };

S2D_DEV void observe_and_check(const S2DDevParams& p, Env& e, ObsOut& ob, int& done, float& reward, int& result) {
  float dx = e.bx - e.px, dy = e.by - e.py;
  float ball_speed = hypot2(e.bvx, e.bvy);               // :91
  float ball_direction = atan2_deg(e.bvy, e.bvx);        // :92
  float player_body = norm_deg(e.body);                  // :94 / :122
  float player_to_ball = atan2_deg(dy, dx);              // :95 / :123
  float rel = norm_deg(player_to_ball - player_body);    // :96 / :124
  ob.o[0] = rel / 180.0f;                                // :98-107
  ob.o[1] = player_body / 180.0f;
  ob.o[2] = e.px / p.half_l;
  ob.o[3] = e.py / p.half_w;
  ob.o[4] = e.bx / p.half_l;
  ob.o[5] = e.by / p.half_w;
  ob.o[6] = ball_speed / 3.0f;
  ob.o[7] = ball_direction / 360.0f;
  ob.o[8] = e.bvx / 3.0f;
  ob.o[9] = e.bvy / 3.0f;
  float distance_to_ball = hypot2(dx, dy);               // :121
  int d = 0, res = S2D_RESULT_NONE;
  float r = 0.0f;
  r += e.prev_dist - distance_to_ball;                   // :130-131
  r += (fabsf(norm_deg(e.prev_angle)) - fabsf(rel)) / 180.0f;   // :133-134
  if (distance_to_ball < p.min_distance_to_ball) { d = 1; r += 10.0f; res = S2D_RESULT_GOAL; }      // :137-140
  if (fabsf(e.px) > p.half_l || fabsf(e.py) > p.half_w) { d = 1; r -= -10.0f; res = S2D_RESULT_OUT; }  // :142-145
  if (e.step_number > p.max_steps) { d = 1; r -= 5.0f; res = S2D_RESULT_TIMEOUT; }                  // :147-150
  e.prev_dist = distance_to_ball;                        // :158
  e.prev_angle = rel;                                    // :159
  done = d; reward = r; result = res;
}

// ------------------------------------------------------------------ S: rcssserver cycle (EXT)
S2D_DEV void cmd_dash(const S2DDevParams& p, Env& e, float power, float dir, float& ax, float& ay) {
  power = clampf(power, p.min_dash_power, p.max_dash_power);
  dir = clampf(dir, p.min_dash_angle, p.max_dash_angle);
  if (p.dash_angle_step > 0.0f) dir = p.dash_angle_step * rintf(dir / p.dash_angle_step);
  bool back = power < 0.0f;
  float need = back ? power * -2.0f : power;
  float avail = e.stamina + p.extra_stamina;
  if (need > avail) need = avail;
  float st = e.stamina - need;
  e.stamina = st > 0.0f ? st : 0.0f;
  power = back ? need / -2.0f : need;
  float ad = fabsf(dir);
  float dir_rate = ad > 90.0f
      ? p.back_dash_rate - ((p.back_dash_rate - p.side_dash_rate) * (1.0f - (ad - 90.0f) / 90.0f))
      : p.side_dash_rate + ((1.0f - p.side_dash_rate) * (1.0f - ad / 90.0f));
  dir_rate = clampf(dir_rate, 0.0f, 1.0f);
  float acc = fabsf(e.effort * power * dir_rate * p.dash_power_rate);
  if (back) dir += 180.0f;
  float sn, cs;
  sincos_deg(norm_deg(e.body + dir), sn, cs);
  ax += acc * cs;
  ay += acc * sn;
}
S2D_DEV void cmd_turn(const S2DDevParams& p, Env& e, float moment, float noise_u) {
  moment = clampf(moment, p.min_moment, p.max_moment);
  float speed = hypot2(e.vx, e.vy);
  float f = 1.0f;
  if (p.noise) f = 1.0f + (noise_u * 2.0f - 1.0f) * p.player_rand;
  e.body = norm_deg(e.body + f * moment / (1.0f + p.inertia_moment * speed));
}
S2D_DEV void obj_inc(float& x, float& y, float& vx, float& vy, float ax, float ay, float accel_max,
                     float speed_max, int noise, float rnd, float u_mag, float u_ang) {
  if (ax != 0.0f || ay != 0.0f) {
    float a = hypot2(ax, ay);
    if (a > accel_max) { float k = accel_max / a; ax *= k; ay *= k; }
    vx += ax; vy += ay;
  }
  if (vx != 0.0f || vy != 0.0f) {
    float s = hypot2(vx, vy);
    if (s > speed_max) { float k = speed_max / s; vx *= k; vy *= k; }
  }
  if (noise) {
    float s = hypot2(vx, vy);
    float mag = u_mag * (rnd * s);
    float sn, cs;
    sincos_deg(u_ang * 360.0f - 180.0f, sn, cs);
    vx += mag * cs; vy += mag * sn;
  }
  x += vx; y += vy;
}
S2D_DEV void collide(const S2DDevParams& p, Env& e) {
  float dx = e.bx - e.px, dy = e.by - e.py;
  float d = hypot2(dx, dy);
  float rsum = p.player_size + p.ball_size;
  if (d < rsum) {
    float ux, uy;
    if (d > 0.0f) { ux = dx / d; uy = dy / d; } else { ux = 1.0f; uy = 0.0f; }
    float mx = (e.px + e.bx) * 0.5f, my = (e.py + e.by) * 0.5f;
    float h = rsum * 0.5f;
    e.px = mx - ux * h; e.py = my - uy * h;
    e.bx = mx + ux * h; e.by = my + uy * h;
    e.vx *= p.collision_vel_rate; e.vy *= p.collision_vel_rate;
    e.bvx *= p.collision_vel_rate; e.bvy *= p.collision_vel_rate;
  }
}
S2D_DEV void update_stamina(const S2DDevParams& p, Env& e) {
  if (e.stamina <= p.recover_dec_thr_value) {
    if (e.recovery > p.recover_min) { float r = e.recovery - p.recover_dec; e.recovery = r > p.recover_min ? r : p.recover_min; }
  }
  if (e.stamina <= p.effort_dec_thr_value) {
    if (e.effort > p.effort_min) { float f = e.effort - p.effort_dec; e.effort = f > p.effort_min ? f : p.effort_min; }
  }
  if (e.stamina >= p.effort_inc_thr_value) {
    if (e.effort < p.effort_init) { float f = e.effort + p.effort_inc; e.effort = f < p.effort_init ? f : p.effort_init; }
  }
  float inc = e.recovery * p.stamina_inc_max;
  float room = p.stamina_max - e.stamina;
  if (inc > room) inc = room;
  if (p.stamina_capacity >= 0.0f) { if (inc > e.capacity) inc = e.capacity; }
  e.stamina += inc;
  if (e.stamina > p.stamina_max) e.stamina = p.stamina_max;
  if (p.stamina_capacity >= 0.0f) { float c = e.capacity - inc; e.capacity = c > 0.0f ? c : 0.0f; }
}
// one cycle, play_on, referee off (coach mode: soccer_2d_env.py:363-366)
S2D_DEV void sim_cycle(const S2DDevParams& p, Env& e, uint32_t gid_lo, uint32_t gid_hi, int cmd, float power,
                       float dir) {
  U4 nz{0, 0, 0, 0}, nz2{0, 0, 0, 0};
  if (p.noise) {
    nz = s2d_draw(p, gid_lo, gid_hi, (uint32_t)e.cycle, S2D_ST_NOISE, 0);
    nz2 = s2d_draw(p, gid_lo, gid_hi, (uint32_t)e.cycle, S2D_ST_NOISE, 1);
  }
  float ax = 0.0f, ay = 0.0f;
  if (cmd == S2D_CMD_DASH) cmd_dash(p, e, power, dir, ax, ay);
  else if (cmd == S2D_CMD_TURN) cmd_turn(p, e, dir, rnd_u01(nz2.x));
  obj_inc(e.px, e.py, e.vx, e.vy, ax, ay, p.player_accel_max, p.player_speed_max, p.noise, p.player_rand,
          rnd_u01(nz.x), rnd_u01(nz.y));
  obj_inc(e.bx, e.by, e.bvx, e.bvy, 0.0f, 0.0f, 0.0f, p.ball_speed_max, p.noise, p.ball_rand, rnd_u01(nz.z),
          rnd_u01(nz.w));
  collide(p, e);
  e.cycle += 1;
  e.vx *= p.player_decay; e.vy *= p.player_decay;
  e.bvx *= p.ball_decay; e.bvy *= p.ball_decay;
  update_stamina(p, e);
}

// ------------------------------------------------------------------ A5 + A6 reset
// trainer_reset_actions + get_ball_velocity (reach_ball_env.py:170-218), then ONE cycle
// with no body command (soccer_2d_env.py:186-197).  Philox RESET stream at (gid, cycle):
// block 0 = {player x, player y, body, ball x}, block 1 = {ball y}, attempt k = block 2+k.
#define S2D_MAX_VEL_TRIES 256
S2D_DEV void env_reset(const S2DDevParams& p, Env& e, uint32_t gid_lo, uint32_t gid_hi) {
  uint32_t cyc = (uint32_t)e.cycle;
  U4 w = s2d_draw(p, gid_lo, gid_hi, cyc, S2D_ST_RESET, 0);
  float px = (float)(-50 + rnd_below(w.x, 101));         // :173
  float py = (float)(-30 + rnd_below(w.y, 61));          // :174
  float body = (float)rnd_below(w.z, 361);               // :175
  float bx, by;
  if (p.change_ball_position) {                          // :176-181
    bx = (float)(-50 + rnd_below(w.w, 101));
    U4 w1 = s2d_draw(p, gid_lo, gid_hi, cyc, S2D_ST_RESET, 1);
    by = (float)(-30 + rnd_below(w1.x, 61));
  } else {
    bx = p.ball_position_x; by = p.ball_position_y;
  }
  float bvx = 0.0f, bvy = 0.0f;
  if (p.change_ball_velocity) {                          // :202-212
    bool ok = false;
    for (int k = 0; k < S2D_MAX_VEL_TRIES && !ok; ++k) {  // bounded: every lane leaves the loop
      U4 wv = s2d_draw(p, gid_lo, gid_hi, cyc, S2D_ST_RESET, 2 + k);
      float speed = rnd_u01(wv.x) * 3.0f;
      float dir = (float)rnd_below(wv.y, 361);
      float sn, cs;
      sincos_deg(dir, sn, cs);
      bvx = speed * cs; bvy = speed * sn;
      float travel = speed * p.travel_factor;
      float tx = bx + travel * cs, ty = by + travel * sn;
      if (fabsf(tx) <= p.half_l && fabsf(ty) <= p.half_w) ok = true;
    }
    if (!ok) { bvx = 0.0f; bvy = 0.0f; }
  } else {                                               // :213-216
    float sn, cs;
    sincos_deg(p.ball_direction, sn, cs);
    bvx = p.ball_speed * cs; bvy = p.ball_speed * sn;
  }
  e.step_number = 0;                                     // :172
  e.bx = bx; e.by = by; e.bvx = bvx; e.bvy = bvy;
  e.px = px; e.py = py; e.body = norm_deg(body); e.vx = 0.0f; e.vy = 0.0f;
  e.stamina = p.stamina_max; e.recovery = p.recover_init;
  e.effort = p.effort_init; e.capacity = p.stamina_capacity;
  sim_cycle(p, e, gid_lo, gid_hi, S2D_CMD_NONE, 0.0f, 0.0f);
}
