// s2d_device.h -- device-side building blocks of the MI355X (gfx950) 2D-soccer engine:
// the deterministic fp32 math spec, Philox4x32-10 (section 5) and
// the one-player/one-ball dynamics of the reach_ball path.
//
// fp32 contract: every operation below is an IEEE-754 binary32 add / mul / fma / div /
// sqrt / rint / floor / compare executed in the written order (the library is compiled with
// -ffp-contract=off; hipcc's default correctly-rounded divide and sqrt stay on).  The
// results are therefore a pure function of the inputs -- the same on every CU, for every
// launch geometry and every shard layout.
//
// Shape of the code (the path is VALU-issue-bound, not HBM-bound, until it is lean):
//   * the always-taken path is straight-line selects; real branches are kept only for
//     rare events (speed/accel clamps, collisions, resets) so that waves skip them;
//   * divisions by constants are multiplications by the rounded reciprocal, magnitude
//     tests compare squares, atan2 uses ONE division;
//   * loop-invariant scalars of the common path travel by value in the kernarg segment
//     (S2DHot, SGPRs); parameters only the rare paths read are fetched through a pointer
//     (S2DRare) inside those branches, so they do not occupy SGPRs in the hot loop.
//
// Reference semantics (file:line under /root/reference) are cited per function; the
// rcssserver arithmetic (dash/turn/stamina/integrate/collide) is EXT (SURVEY.md appx A).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/s2d.h"

#define S2D_DEV __device__ __forceinline__

// action decoding mode (template parameter): reach_ball_env.py:39-47
enum { S2D_MODE_DISCRETE = 0, S2D_MODE_CONT1 = 1, S2D_MODE_TURN4 = 2 };

// ------------------------------------------------------------------ parameters
struct S2DHot {  // by value (kernarg -> SGPRs): everything the always-taken path reads
  float inv_half_l, inv_half_w, half_l, half_w;
  float player_decay, ball_decay;
  float player_accel_max2, player_speed_max2, ball_speed_max2, rsum2;
  float stamina_max, stamina_inc_max, extra_stamina, stamina_capacity;
  float recover_init, recover_dec_thr_value, recover_min, recover_dec;
  float effort_init, effort_dec_thr_value, effort_min, effort_dec, effort_inc_thr_value, effort_inc;
  float dash_power_rate, max_dash_power, min_dash_power, max_dash_angle, min_dash_angle;
  float dash_angle_step, inv_dash_angle_step, side_dash_rate, back_dash_rate;
  float min_distance_to_ball, min_dist2_thr, act_scale;
  int max_steps, n_actions, auto_reset;
  uint32_t seed_lo, seed_hi, gid_lo, gid_hi;
  // read only by the TURN4 / NOISE instantiations
  float max_moment, min_moment, inertia_moment, player_rand, ball_rand;
};
struct S2DRare {  // device memory, read inside rare branches only
  float player_accel_max, player_speed_max, ball_speed_max, rsum, collision_vel_rate;
  float recover_init;
  float ball_position_x, ball_position_y, ball_speed, ball_direction, travel_factor;
  int change_ball_position, change_ball_velocity;
  int tab_len;        // entries of S2DTables (0: the dash-only fast path is off for this configuration)
  float tab_power;    // the clamped power of every Dash(100, .) the tables were built with
  int pad[1];
};

// The loop-invariant float parameters arrive in SGPRs (kernarg).  A rollout loop keeps ~45 of
// them live next to pointers, exec masks and loop state, which exceeds the 102 SGPRs of a wave:
// the compiler then spills to VGPR lanes or re-loads kernargs inside the loop (s_load +
// s_waitcnt, ~200 cycles each).  VGPRs are plentiful here (60 of 512), so the floats are moved
// into VGPRs once, behind an opaque v_mov the compiler cannot fold back into scalar form.
S2D_DEV float to_vgpr(float x) {
  float y;
  asm("v_mov_b32 %0, %1" : "=v"(y) : "s"(x));
  return y;
}
S2D_DEV S2DHot hot_in_vgprs(const S2DHot& p) {
  S2DHot v = p;   // ints / seeds stay scalar (they feed scalar branches and Philox keys)
  v.inv_half_l = to_vgpr(p.inv_half_l); v.inv_half_w = to_vgpr(p.inv_half_w);
  v.half_l = to_vgpr(p.half_l); v.half_w = to_vgpr(p.half_w);
  v.player_decay = to_vgpr(p.player_decay); v.ball_decay = to_vgpr(p.ball_decay);
  v.player_accel_max2 = to_vgpr(p.player_accel_max2); v.player_speed_max2 = to_vgpr(p.player_speed_max2);
  v.ball_speed_max2 = to_vgpr(p.ball_speed_max2); v.rsum2 = to_vgpr(p.rsum2);
  v.stamina_max = to_vgpr(p.stamina_max); v.stamina_inc_max = to_vgpr(p.stamina_inc_max);
  v.extra_stamina = to_vgpr(p.extra_stamina);
  v.recover_dec_thr_value = to_vgpr(p.recover_dec_thr_value); v.recover_min = to_vgpr(p.recover_min);
  v.recover_dec = to_vgpr(p.recover_dec);
  v.effort_init = to_vgpr(p.effort_init); v.effort_dec_thr_value = to_vgpr(p.effort_dec_thr_value);
  v.effort_min = to_vgpr(p.effort_min); v.effort_dec = to_vgpr(p.effort_dec);
  v.effort_inc_thr_value = to_vgpr(p.effort_inc_thr_value); v.effort_inc = to_vgpr(p.effort_inc);
  v.dash_power_rate = to_vgpr(p.dash_power_rate); v.max_dash_power = to_vgpr(p.max_dash_power);
  v.min_dash_power = to_vgpr(p.min_dash_power); v.max_dash_angle = to_vgpr(p.max_dash_angle);
  v.min_dash_angle = to_vgpr(p.min_dash_angle);
  v.inv_dash_angle_step = to_vgpr(p.inv_dash_angle_step);
  v.side_dash_rate = to_vgpr(p.side_dash_rate); v.back_dash_rate = to_vgpr(p.back_dash_rate);
  v.min_distance_to_ball = to_vgpr(p.min_distance_to_ball); v.act_scale = to_vgpr(p.act_scale);
  v.min_dist2_thr = to_vgpr(p.min_dist2_thr);
  return v;
}

// ------------------------------------------------------------------ Philox4x32-10
enum { S2D_ST_RESET = 0, S2D_ST_POLICY = 1, S2D_ST_SELECT = 2, S2D_ST_NOISE = 3, S2D_ST_NOISE_RESET = 5 };   // 4 = tackle (match)

struct U4 { uint32_t x, y, z, w; };

S2D_DEV U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;   // one v_mad_u64_u32 each
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}
// ctr = { gid_lo, gid_hi, cycle, (stream << 16) | block }, key = seed
S2D_DEV U4 s2d_draw(const S2DHot& p, uint32_t gid_lo, uint32_t gid_hi, uint32_t cycle, uint32_t stream,
                    uint32_t block) {
  return philox4x32_10(gid_lo, gid_hi, cycle, (stream << 16) | block, p.seed_lo, p.seed_hi);
}
// In-engine policy randomness is keyed by the env's policy_step k (not by its cycle), so the
// draw of step t does not depend on whether an episode ended before t: a rollout kernel can
// draw ahead of the simulation, and one Philox call serves four steps (word k & 3 of the block
// at counter k >> 2).  The 4-D turning policy uses the four words of POLICY block 1 at counter k.
S2D_DEV U4 policy_quad(const S2DHot& p, uint32_t gid_lo, uint32_t gid_hi, uint32_t k, uint32_t stream) {
  return s2d_draw(p, gid_lo, gid_hi, k >> 2, stream, 0);
}
S2D_DEV uint32_t quad_word(const U4& q, uint32_t k) {
  uint32_t j = k & 3u;
  uint32_t lo = (j & 1u) ? q.y : q.x, hi = (j & 1u) ? q.w : q.z;
  return (j & 2u) ? hi : lo;
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
// one division: atan(mn/mx) = 45deg + atan((mn-mx)/(mn+mx)) above tan(pi/8)
S2D_DEV float atan2_deg(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  bool hi = mn > 0.41421356237f * mx;
  float num = hi ? mn - mx : mn;
  float den = hi ? mn + mx : mx;
  float base = hi ? 45.0f : 0.0f;
  den = (mx == 0.0f) ? 1.0f : den;      // zero vector: 0/1 -> angle 0 (pyrusgeom convention)
  float t = num / den;
  float z = t * t;
  float p = fmaf(z, 8.05374449538e-2f, -1.38776856032e-1f);
  p = fmaf(z, p, 1.99777106478e-1f);
  p = fmaf(z, p, -3.33329491539e-1f);
  float a = fmaf(p * z, t, t);
  a = fmaf(a, 57.29577951308232f, base);
  a = (ay > ax) ? 90.0f - a : a;
  a = (x < 0.0f) ? 180.0f - a : a;
  a = (y < 0.0f) ? -a : a;
  return (mx == 0.0f) ? 0.0f : a;
}
S2D_DEV float sq2(float x, float y) { return fmaf(x, x, y * y); }
// Correctly rounded square root (== sqrtf) in 9 instructions instead of the compiler's 16: v_sqrt_f32 (1 ulp), then the
// neighbour test of the compiler's own expansion -- residuals of r - 1ulp and r + 1ulp, both exact in an fma -- without
// its scaling of tiny arguments and its class test.  That core is exact for x = 0 and for x >= 2^-96; every distance and
// speed the engine forms lies there (nonzero squares are >= 1e-15), and the general sequence stays as a rare branch for
// 0 < x < 2^-96 so that the function equals sqrtf on every non-negative finite input.
#ifndef S2D_FAST_SQRT
#define S2D_FAST_SQRT 1
#endif
S2D_DEV float sqrt_cr(float x) {
#if S2D_FAST_SQRT
  float r = __builtin_amdgcn_sqrtf(x);
  const float rm = __int_as_float(__float_as_int(r) - 1), rp = __int_as_float(__float_as_int(r) + 1);
  const float em = fmaf(-rm, r, x), ep = fmaf(-rp, r, x);
  r = (em <= 0.0f) ? rm : r;
  r = (ep > 0.0f) ? rp : r;
  if (__builtin_expect((uint32_t)__float_as_int(x) - 1u < 0x0f800000u - 1u, 0)) r = sqrtf(x);   // 0 < x < 2^-96
  return r;
#else
  return sqrtf(x);
#endif
}
S2D_DEV float hypot2(float x, float y) { return sqrt_cr(fmaf(x, x, y * y)); }
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
// pyrusgeom AngleDeg normalisation for |d| <= 540 (every angle the engine forms: sums and
// differences of two angles in [-180,180], plus 180 for a back dash); general form with
// fmod kept for the diagnostic entry point.
S2D_DEV float norm_deg(float d) {
  d = (d < -180.0f) ? d + 360.0f : d;
  d = (d > 180.0f) ? d - 360.0f : d;
  return d;
}
S2D_DEV float norm_deg_any(float d) {
  if (d < -360.0f || 360.0f < d) d = fmodf(d, 360.0f);
  return norm_deg(d);
}
S2D_DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------------------ one env in registers
struct Env {
  float px, py, vx, vy, body, stamina, effort, recovery, capacity;
  float bx, by, bvx, bvy, prev_dist, prev_angle;
  int step_number, cycle;
  int episode;   // resets so far = index of the current episode: the RESET stream's counter word
};
enum {  // SoA plane order == S2DBuffers state pointers
  F_PX, F_PY, F_VX, F_VY, F_BODY, F_STAMINA, F_EFFORT, F_RECOVERY, F_CAPACITY,
  F_BX, F_BY, F_BVX, F_BVY, F_PREV_DIST, F_PREV_ANGLE, F_STEP, F_CYCLE,
  F_POLICY,   // policy_step: touched only by launches that draw in-engine policy randomness
  F_EPISODE,  // episode index (read by every launch that can reset, written when it did)
  F_COUNT
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
  e.episode = __float_as_int(S[F_EPISODE * stride + i]);
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
  S[F_EPISODE * stride + i] = __int_as_float(e.episode);
}

// ------------------------------------------------------------------ A2 action map
// ReachBallEnv.action_to_rpc_actions, reach_ball_env.py:53-85 (step_number++ by caller)
struct Action4 { float a0, a1, a2, a3; };

template <int MODE>
S2D_DEV void action_map(const S2DHot& p, const Action4& a, float u, int& cmd, float& power, float& dir) {
  if (MODE == S2D_MODE_TURN4) {
    float turn_prob = clampf(a.a0, -1.0f, 1.0f), turn_angle = clampf(a.a1, -1.0f, 1.0f);   // :64-68
    float dash_prob = clampf(a.a2, -1.0f, 1.0f), dash_angle = clampf(a.a3, -1.0f, 1.0f);
    float e0 = exp_spec(dash_prob), e1 = exp_spec(turn_prob);                              // :69-70
    float p0 = e0 / (e0 + e1);
    bool turn = u < p0;                                                                     // :71 (quirk kept)
    cmd = turn ? S2D_CMD_TURN : S2D_CMD_DASH;
    power = turn ? 0.0f : 100.0f;
    dir = (turn ? turn_angle : dash_angle) * 180.0f;                                        // :73, :77
  } else if (MODE == S2D_MODE_CONT1) {
    cmd = S2D_CMD_DASH; power = 100.0f; dir = a.a0 * 180.0f;                                // :81-82, not clipped
  } else {                                                                                  // :84-85
    float t = a.a0 * p.act_scale;                                  // act_scale = 360 / n
    float m = t - 360.0f * floorf(t * 0.002777777777777778f);      // floor-mod; identity for 0 <= a < n
    cmd = S2D_CMD_DASH; power = 100.0f; dir = m - 180.0f;
  }
}

// S2D_ACT_COMMAND: the command word of a decoded body command (float[N][4], include/s2d.h) -> S2D_CMD_*.  One definition for the kernels
// and (restated) for the CPU checker: the nearest of FREEZE (-1), NONE (0), DASH (1), TURN (2); anything else -- other numbers, NaN --
// is no command.
S2D_DEV int command_code(float x) {
  int c = S2D_CMD_NONE;
  c = (x >= -1.5f && x <= -0.5f) ? (int)S2D_CMD_FREEZE : c;
  c = (x >= 0.5f && x < 1.5f) ? (int)S2D_CMD_DASH : c;
  c = (x >= 1.5f && x < 2.5f) ? (int)S2D_CMD_TURN : c;
  return c;
}

// ------------------------------------------------------------------ A3 + A4 fused
// state_to_observation (reach_ball_env.py:87-111) and check_trainer_observation (:113-161)
// read the same full-state truth, so the player->ball vector, its angle and the distance
// are computed once and shared.  body / prev_angle are stored normalised, so AngleDeg(x) of
// lines 94, 122 and 133 is the identity on them.
struct ObsOut { float o[S2D_OBS_DIM]; };

// The three pieces below are the same arithmetic in the same order whether they run in one
// thread (observe_and_check) or are split between a simulating wave and an observing wave
// (wave-specialised rollout kernel).
enum { S2D_FLAG_GOAL = 1, S2D_FLAG_OUT = 2, S2D_FLAG_TIMEOUT = 4 };

// distance + done conditions, reach_ball_env.py:121, 137, 142, 147 (d2 = |ball - player|^2)
S2D_DEV int judge(const S2DHot& p, float px, float py, float d2, int step_number, float& dist) {
  dist = sqrt_cr(d2);                                    // :121  == hypot2(bx - px, by - py)
  int f = (dist < p.min_distance_to_ball) ? S2D_FLAG_GOAL : 0;                       // :137
  f |= (fabsf(px) > p.half_l || fabsf(py) > p.half_w) ? S2D_FLAG_OUT : 0;            // :142
  f |= (step_number > p.max_steps) ? S2D_FLAG_TIMEOUT : 0;                           // :147 strict >
  return f;
}
// The same decision without the square root: sqrt is correctly rounded and monotone, so
// sqrtf(d2) < m  <=>  d2 < T  with T = the smallest float whose rounded root reaches m (found on
// the host).  Lets the simulating wave decide "done" from d2 and leaves the root to the
// observing wave, which needs the distance for the reward anyway.
S2D_DEV int judge_sq(const S2DHot& p, float px, float py, float d2, int step_number) {
  int f = (d2 < p.min_dist2_thr) ? S2D_FLAG_GOAL : 0;
  f |= (fabsf(px) > p.half_l || fabsf(py) > p.half_w) ? S2D_FLAG_OUT : 0;
  f |= (step_number > p.max_steps) ? S2D_FLAG_TIMEOUT : 0;
  return f;
}
// state_to_observation, reach_ball_env.py:87-111, in two halves that share nothing (so two waves can
// evaluate them side by side): the player half (o[0..3], returns the body->ball angle difference) and the
// ball half (o[4..9]).
S2D_DEV float observe_player(const S2DHot& p, float px, float py, float body, float bx, float by, float* o) {
  float dx = bx - px, dy = by - py;
  float player_to_ball = atan2_deg(dy, dx);              // :95 / :123
  float rel = norm_deg(player_to_ball - body);           // :96 / :124
  o[0] = rel * 0.005555555555555556f;                    // :98-101  (x/180, x/52.5, x/34)
  o[1] = body * 0.005555555555555556f;
  o[2] = px * p.inv_half_l;
  o[3] = py * p.inv_half_w;
  return rel;
}
S2D_DEV void observe_ball(const S2DHot& p, float bx, float by, float bvx, float bvy, float* o) {
  float ball_speed = hypot2(bvx, bvy);                   // :91
  float ball_direction = atan2_deg(bvy, bvx);            // :92
  o[4] = bx * p.inv_half_l;                              // :102-107  (x/52.5, x/34, x/3, x/360)
  o[5] = by * p.inv_half_w;
  o[6] = ball_speed * 0.3333333333333333f;
  o[7] = ball_direction * 0.002777777777777778f;
  o[8] = bvx * 0.3333333333333333f;
  o[9] = bvy * 0.3333333333333333f;
}
S2D_DEV float observe(const S2DHot& p, float px, float py, float body, float bx, float by, float bvx, float bvy,
                      ObsOut& ob) {
  observe_ball(p, bx, by, bvx, bvy, ob.o);
  return observe_player(p, px, py, body, bx, by, ob.o);
}
// reward and label of check_trainer_observation, reach_ball_env.py:128-150
S2D_DEV float reward_of(float prev_dist, float prev_angle, float dist, float rel, int flags, int& result) {
  float r = prev_dist - dist;                            // :130-131
  r += (fabsf(prev_angle) - fabsf(rel)) * 0.005555555555555556f;   // :133-134
  r = (flags & S2D_FLAG_GOAL) ? r + 10.0f : r;           // :139
  r = (flags & S2D_FLAG_OUT) ? r - -10.0f : r;           // :144 (+10, quirk kept)
  r = (flags & S2D_FLAG_TIMEOUT) ? r - 5.0f : r;         // :149
  int res = (flags & S2D_FLAG_GOAL) ? S2D_RESULT_GOAL : S2D_RESULT_NONE;   // later label overwrites earlier
  res = (flags & S2D_FLAG_OUT) ? S2D_RESULT_OUT : res;
  res = (flags & S2D_FLAG_TIMEOUT) ? S2D_RESULT_TIMEOUT : res;
  result = res;
  return r;
}

// d2 = |ball - player|^2 of the state in `e` (sim_cycle returns it)
S2D_DEV void observe_and_check(const S2DHot& p, Env& e, float d2, ObsOut& ob, int& done, float& reward, int& result) {
  float rel = observe(p, e.px, e.py, e.body, e.bx, e.by, e.bvx, e.bvy, ob);
  float dist;
  int flags = judge(p, e.px, e.py, d2, e.step_number, dist);
  reward = reward_of(e.prev_dist, e.prev_angle, dist, rel, flags, result);
  e.prev_dist = dist;                                    // :158
  e.prev_angle = rel;                                    // :159
  done = flags ? 1 : 0;
}

// ------------------------------------------------------------------ S: rcssserver cycle (EXT)
// Dash(power, dir), appendix A.  Split in two so that the part that depends on the command
// alone (clamps, direction discretisation, direction rate) can be evaluated ahead of the
// simulation by another wave; the arithmetic and its order are those of the one-piece form.
struct CmdPrep { float power, dir, dir_rate; };   // TURN: dir = the raw moment, the rest unused
S2D_DEV CmdPrep dash_prepare(const S2DHot& p, float power, float dir) {
  power = clampf(power, p.min_dash_power, p.max_dash_power);
  dir = clampf(dir, p.min_dash_angle, p.max_dash_angle);
  float disc = p.dash_angle_step * rintf(dir * p.inv_dash_angle_step);
  dir = (p.dash_angle_step > 0.0f) ? disc : dir;
  float ad = fabsf(dir);
  float r_back = p.back_dash_rate - ((p.back_dash_rate - p.side_dash_rate) * (1.0f - (ad - 90.0f) * 0.011111111111111112f));
  float r_fwd = p.side_dash_rate + ((1.0f - p.side_dash_rate) * (1.0f - ad * 0.011111111111111112f));
  float dir_rate = clampf(ad > 90.0f ? r_back : r_fwd, 0.0f, 1.0f);
  return CmdPrep{power, dir, dir_rate};
}
S2D_DEV void dash_apply(const S2DHot& p, Env& e, const CmdPrep& c, float& ax, float& ay) {
  float power = c.power;
  bool back = power < 0.0f;
  float need = back ? power * -2.0f : power;
  float avail = e.stamina + p.extra_stamina;
  need = (need > avail) ? avail : need;
  float st = e.stamina - need;
  e.stamina = st > 0.0f ? st : 0.0f;
  power = back ? need / -2.0f : need;
  float acc = fabsf(e.effort * power * c.dir_rate * p.dash_power_rate);
  float dir = back ? c.dir + 180.0f : c.dir;
  float sn, cs;
  sincos_deg(norm_deg(e.body + dir), sn, cs);
  ax = acc * cs;
  ay = acc * sn;
}
S2D_DEV CmdPrep cmd_prepare(const S2DHot& p, int cmd, float power, float dir) {
  if (cmd == S2D_CMD_DASH) return dash_prepare(p, power, dir);
  return CmdPrep{power, dir, 0.0f};
}
S2D_DEV void cmd_turn(const S2DHot& p, Env& e, float moment, bool noise, float noise_u) {
  moment = clampf(moment, p.min_moment, p.max_moment);
  float speed = hypot2(e.vx, e.vy);
  float f = noise ? 1.0f + (noise_u * 2.0f - 1.0f) * p.player_rand : 1.0f;
  e.body = norm_deg(e.body + f * moment / (1.0f + p.inertia_moment * speed));
}
// Velocity noise of MPObject::_inc: polar(U(0, rand * |vel|), U(-180, 180)).  The draws do not depend on the state, so -- like
// the policy draw -- they are keyed by the env's policy_step k (stream NOISE; the command-less cycle of a reset uses stream
// NOISE_RESET with the reset's own key) and can be prepared ahead of the simulation by another wave.  Round 3 respecified the
// draw (in the spec, i.e. here and in the CPU checker alike) so that it costs a quarter: ONE word per object and cycle -- magnitude uniform
// from its high 16 bits, direction = a WHOLE degree -180 .. 179 from its low 16 bits (bias < 0.6 % between degrees), whose sine /
// cosine are entries of the whole-degree table the dash fast path already keeps in LDS -- and one Philox block per TWO cycles
// (block 0 at counter k >> 1: words x, y for even k, z, w for odd k).  rcssserver's own generator cannot be matched anyway
// (SURVEY section 7): what is tested is the distribution (tests/test_gpu_distributions.py).
struct NoiseIn { float pm, ps, pc, bm, bs, bc, tu; };   // player: magnitude uniform, sin, cos; ball: same; turn uniform
struct NoiseWords { uint32_t wp, wb; float tu; };        // the raw words: player, ball; turn uniform
S2D_DEV float noise_mag(uint32_t w) { return (float)(w >> 16) * 1.52587890625e-05f; }
S2D_DEV int noise_dir_index(uint32_t w) { return (int)(((w & 0xffffu) * 360u) >> 16); }   // 0 .. 359: whole degree + 180
// `block` caches Philox block 0 across the two cycles it serves (`refresh` = draw it now: first cycle of a launch, or even k)
S2D_DEV NoiseWords noise_words(const S2DHot& p, uint32_t gid_lo, uint32_t gid_hi, uint32_t ctr, uint32_t stream, bool turn,
                               U4& block, bool refresh) {
  const bool paired = stream == S2D_ST_NOISE;
  if (refresh) block = s2d_draw(p, gid_lo, gid_hi, paired ? ctr >> 1 : ctr, stream, 0);
  const bool odd = paired && (ctr & 1u);
  NoiseWords n{odd ? block.z : block.x, odd ? block.w : block.y, 0.0f};
  if (turn) n.tu = rnd_u01(s2d_draw(p, gid_lo, gid_hi, ctr, stream, 1).x);
  return n;
}
S2D_DEV NoiseIn noise_prepare(const S2DHot& p, uint32_t gid_lo, uint32_t gid_hi, uint32_t ctr, uint32_t stream,
                              bool turn) {
  U4 blk;
  const NoiseWords w = noise_words(p, gid_lo, gid_hi, ctr, stream, turn, blk, true);
  NoiseIn n;
  n.pm = noise_mag(w.wp); sincos_deg((float)(noise_dir_index(w.wp) - 180), n.ps, n.pc);
  n.bm = noise_mag(w.wb); sincos_deg((float)(noise_dir_index(w.wb) - 180), n.bs, n.bc);
  n.tu = w.tu;
  return n;
}
S2D_DEV void add_noise(float& vx, float& vy, float rnd, float u_mag, float sn, float cs) {
  float s = hypot2(vx, vy);
  float mag = u_mag * (rnd * s);
  vx += mag * cs; vy += mag * sn;
}
S2D_DEV void update_stamina(const S2DHot& p, Env& e) {
  float st = e.stamina;
  float rdec = e.recovery - p.recover_dec;
  rdec = rdec > p.recover_min ? rdec : p.recover_min;
  e.recovery = (st <= p.recover_dec_thr_value && e.recovery > p.recover_min) ? rdec : e.recovery;
  float fdec = e.effort - p.effort_dec;
  fdec = fdec > p.effort_min ? fdec : p.effort_min;
  e.effort = (st <= p.effort_dec_thr_value && e.effort > p.effort_min) ? fdec : e.effort;
  float finc = e.effort + p.effort_inc;
  finc = finc < p.effort_init ? finc : p.effort_init;
  e.effort = (st >= p.effort_inc_thr_value && e.effort < p.effort_init) ? finc : e.effort;
  float inc = e.recovery * p.stamina_inc_max;
  float room = p.stamina_max - st;
  inc = inc > room ? room : inc;
  bool capped = p.stamina_capacity >= 0.0f;
  inc = (capped && inc > e.capacity) ? e.capacity : inc;
  st += inc;
  e.stamina = st > p.stamina_max ? p.stamina_max : st;
  float c = e.capacity - inc;
  c = c > 0.0f ? c : 0.0f;
  e.capacity = capped ? c : e.capacity;
}
// MPObject::_inc for player and ball + Stadium::collisions for the single pair, in rcssserver's
// order: accel clamp, vel += accel, speed clamp, noise, pos += vel (player, then ball), collision.
// Returns |ball - player|^2 of the final positions.
template <bool NOISE>
S2D_DEV float move_sequential(const S2DHot& p, const S2DRare* __restrict__ rp, Env& e, bool accel, float ax, float ay,
                              const NoiseIn& nz) {
  if (accel) {
    float a2 = sq2(ax, ay);
    if (a2 > p.player_accel_max2) { float k = rp->player_accel_max / sqrtf(a2); ax *= k; ay *= k; }
    e.vx += ax; e.vy += ay;
  }
  float s2 = sq2(e.vx, e.vy);
  if (s2 > p.player_speed_max2) { float k = rp->player_speed_max / sqrtf(s2); e.vx *= k; e.vy *= k; }
  if (NOISE) add_noise(e.vx, e.vy, p.player_rand, nz.pm, nz.ps, nz.pc);
  e.px += e.vx; e.py += e.vy;
  float b2 = sq2(e.bvx, e.bvy);
  if (b2 > p.ball_speed_max2) { float k = rp->ball_speed_max / sqrtf(b2); e.bvx *= k; e.bvy *= k; }
  if (NOISE) add_noise(e.bvx, e.bvy, p.ball_rand, nz.bm, nz.bs, nz.bc);
  e.bx += e.bvx; e.by += e.bvy;
  float dx = e.bx - e.px, dy = e.by - e.py;
  float d2 = sq2(dx, dy);
  if (d2 < p.rsum2) {                          // Stadium::collisions, single pair
    float d = sqrtf(d2);
    float ux, uy;
    if (d > 0.0f) { ux = dx / d; uy = dy / d; } else { ux = 1.0f; uy = 0.0f; }
    float mx = (e.px + e.bx) * 0.5f, my = (e.py + e.by) * 0.5f;
    float h = rp->rsum * 0.5f, cv = rp->collision_vel_rate;
    e.px = mx - ux * h; e.py = my - uy * h;
    e.bx = mx + ux * h; e.by = my + uy * h;
    e.vx *= cv; e.vy *= cv; e.bvx *= cv; e.bvy *= cv;
    d2 = sq2(e.bx - e.px, e.by - e.py);
  }
  return d2;
}
// one cycle, play_on, referee off (coach mode: soccer_2d_env.py:363-366).
// HAS_CMD=false is the command-less cycle a reset consumes (soccer_2d_env.py:190).
// The three clamps and the collision are rare events (with stock parameters a dashing player
// never exceeds accel/speed max, the ball starts below its speed max, and a collision needs the
// ball within 0.385 m).  Without noise the common path therefore evaluates all four conditions
// on the unclamped values -- which are the sequential conditions as long as none fires -- and
// only a wave with a lane that trips one re-runs those lanes through move_sequential: one
// branch per cycle instead of four.  Returns |ball - player|^2 after the cycle (judge_sq).
// the integration of one cycle for given player acceleration: MPObject::_inc for player and ball + the collision
template <bool NOISE>
S2D_DEV float sim_move(const S2DHot& p, const S2DRare* __restrict__ rp, Env& e, bool accel, float ax, float ay,
                       const NoiseIn& nz) {
  // (with noise the same holds: the clamps test the velocities BEFORE this cycle's noise is added -- a dashing player's 0.4 * v + 0.6
  // stays below speed_max for any v the noise can leave behind -- so the noisy cycle, too, runs unclamped with one test at the end;
  // the speeds the noise magnitudes scale with are the roots of the squares the tests need anyway)
  const float vx0 = e.vx, vy0 = e.vy, px0 = e.px, py0 = e.py, bx0 = e.bx, by0 = e.by, bvx0 = e.bvx, bvy0 = e.bvy;
  const float a2 = sq2(ax, ay);
  if (accel) { e.vx += ax; e.vy += ay; }
  const float s2 = sq2(e.vx, e.vy);
  if (NOISE) { const float mag = nz.pm * (p.player_rand * sqrt_cr(s2)); e.vx += mag * nz.pc; e.vy += mag * nz.ps; }   // add_noise
  e.px += e.vx; e.py += e.vy;
  const float b2 = sq2(e.bvx, e.bvy);
  if (NOISE) { const float mag = nz.bm * (p.ball_rand * sqrt_cr(b2)); e.bvx += mag * nz.bc; e.bvy += mag * nz.bs; }
  e.bx += e.bvx; e.by += e.bvy;
  float d2 = sq2(e.bx - e.px, e.by - e.py);
  const bool rare = (accel && a2 > p.player_accel_max2) || s2 > p.player_speed_max2 || b2 > p.ball_speed_max2 ||
                    d2 < p.rsum2;
  if (rare) {
    e.vx = vx0; e.vy = vy0; e.px = px0; e.py = py0; e.bx = bx0; e.by = by0;
    if (NOISE) { e.bvx = bvx0; e.bvy = bvy0; }
    d2 = move_sequential<NOISE>(p, rp, e, accel, ax, ay, nz);
  }
  return d2;
}
// referee tick + MPObject::_turn (decay) of both objects
S2D_DEV void sim_tick_decay(const S2DHot& p, Env& e) {
  e.cycle = (int)((uint32_t)e.cycle + 1u);               // wraps after 2^31 cycles (~47 min of fused rollouts) without UB
  e.vx *= p.player_decay; e.vy *= p.player_decay;
  e.bvx *= p.ball_decay; e.bvy *= p.ball_decay;
}
template <bool NOISE, bool HAS_CMD>
S2D_DEV float sim_cycle(const S2DHot& p, const S2DRare* __restrict__ rp, Env& e, int cmd, const CmdPrep& c,
                        const NoiseIn& nz) {
  float ax = 0.0f, ay = 0.0f;
  bool accel = false;
  if (HAS_CMD) {
    if (cmd == S2D_CMD_DASH) {
      dash_apply(p, e, c, ax, ay);
      accel = true;
    } else if (cmd == S2D_CMD_TURN) {
      cmd_turn(p, e, c.dir, NOISE, nz.tu);
    }
  }
  const float d2 = sim_move<NOISE>(p, rp, e, accel, ax, ay, nz);
  sim_tick_decay(p, e);
  update_stamina(p, e);
  return d2;
}

// ---- the dash-only fast path of the rollout pipeline ------------------------------------------------------------------
// In the discrete and 1-D continuous action modes every command is Dash(100, dir) (reach_ball_env.py:79-85), so stamina,
// effort, recovery and capacity are functions of the episode's step number alone, and -- with an integer dash_angle_step
// and the integer body angles the reset grid produces (:175; nothing but a Turn changes the body) -- the dash direction is
// a whole number of degrees.  S2DTables (built once per engine by the very functions above) holds, per step number s:
// the stamina words BEFORE the step and ep[s] = effort * spent power of the dash (dash_apply's operands, in its order).
// A group whose envs all sit on that table (checked when the launch starts) simulates a cycle without the stamina model
// and reads sine / cosine from a 361-entry table of sincos_deg at whole degrees: same values, ~110 instructions fewer.
#define S2D_TAB_MAX 256
struct S2DTables { float ep[S2D_TAB_MAX], stamina[S2D_TAB_MAX], effort[S2D_TAB_MAX], recovery[S2D_TAB_MAX], capacity[S2D_TAB_MAX]; };
S2D_DEV void tables_build(const S2DHot& p, float recover_init, float cmd_power, int len, S2DTables& t) {
  Env e{};
  e.stamina = p.stamina_max; e.recovery = recover_init; e.effort = p.effort_init; e.capacity = p.stamina_capacity;   // (recover)
  update_stamina(p, e);                                  // the command-less cycle a reset consumes
  for (int s = 0; s < len; ++s) {
    t.stamina[s] = e.stamina; t.effort[s] = e.effort; t.recovery[s] = e.recovery; t.capacity[s] = e.capacity;
    const float avail = e.stamina + p.extra_stamina;     // dash_apply with power = cmd_power > 0 (never a back dash)
    const float need = (cmd_power > avail) ? avail : cmd_power;
    const float st = e.stamina - need;
    t.ep[s] = e.effort * need;
    e.stamina = st > 0.0f ? st : 0.0f;
    update_stamina(p, e);
  }
}
template <bool NOISE>
S2D_DEV float sim_cycle_dash_fast(const S2DHot& p, const S2DRare* __restrict__ rp, Env& e, float ep, float dir_rate,
                                  float sn, float cs, const NoiseIn& nz) {
  const float acc = fabsf(ep * dir_rate * p.dash_power_rate);   // dash_apply: |effort * power * dir_rate * dash_power_rate|
  const float d2 = sim_move<NOISE>(p, rp, e, true, acc * cs, acc * sn, nz);
  sim_tick_decay(p, e);
  return d2;
}

// ------------------------------------------------------------------ A5 + A6 reset
// trainer_reset_actions + get_ball_velocity (reach_ball_env.py:170-218), then ONE cycle
// with no body command (soccer_2d_env.py:186-197).  Philox RESET stream at (gid, cycle):
// block 0 = {player x, player y, body, ball x}, block 1 = {ball y, speed0, dir0, -}, then two
// velocity candidates per block.
#define S2D_MAX_VEL_TRIES 255
struct ResetSample { float px, py, body, bx, by, bvx, bvy; };

// RESET stream counter word = the index of the episode the reset STARTS (1, 2, ...; Env::episode counts the resets so
// far).  The state a reset leaves behind is therefore a function of (env id, episode index) alone -- it does not depend on
// how long earlier episodes lasted -- so the rollout kernels prepare the next few episodes of every env ahead of the
// simulation, off the simulating wave, and a reset becomes a copy.
S2D_DEV uint32_t reset_key(const Env& e) { return (uint32_t)e.episode + 1u; }

// one velocity candidate of get_ball_velocity (reach_ball_env.py:202-212): speed, direction -> velocity; accepted when the ball
// would come to rest inside the pitch
S2D_DEV bool vel_candidate(const S2DHot& p, const S2DRare& r, float bx, float by, uint32_t ws, uint32_t wd, float& bvx, float& bvy) {
  float speed = rnd_u01(ws) * 3.0f;
  float dir = (float)rnd_below(wd, 361);
  float sn, cs;
  sincos_deg(dir, sn, cs);
  bvx = speed * cs; bvy = speed * sn;
  float travel = speed * r.travel_factor;
  float tx = bx + travel * cs, ty = by + travel * sn;
  return fabsf(tx) <= p.half_l && fabsf(ty) <= p.half_w;
}
// candidate k >= 1 of an env: two candidates per Philox block (block 2 + (k - 1) / 2; try 0 rides in block 1)
S2D_DEV bool vel_try(const S2DHot& p, const S2DRare& r, uint32_t gid_lo, uint32_t gid_hi, uint32_t c0, float bx, float by, int k,
                     float& bvx, float& bvy) {
  const U4 wb = s2d_draw(p, gid_lo, gid_hi, c0, S2D_ST_RESET, 2 + ((k - 1) >> 1));
  const bool odd = ((k - 1) & 1) != 0;
  return vel_candidate(p, r, bx, by, odd ? wb.z : wb.x, odd ? wb.w : wb.y, bvx, bvy);
}
S2D_DEV ResetSample reset_sample(const S2DHot& p, const S2DRare& r, uint32_t gid_lo, uint32_t gid_hi, uint32_t c0) {
  U4 w = s2d_draw(p, gid_lo, gid_hi, c0, S2D_ST_RESET, 0);
  U4 w1 = s2d_draw(p, gid_lo, gid_hi, c0, S2D_ST_RESET, 1);
  ResetSample o;
  o.px = (float)(-50 + rnd_below(w.x, 101));             // :173
  o.py = (float)(-30 + rnd_below(w.y, 61));              // :174
  o.body = norm_deg((float)rnd_below(w.z, 361));         // :175 (move normalises the angle)
  float bx = r.ball_position_x, by = r.ball_position_y;
  if (r.change_ball_position) {                          // :176-181
    bx = (float)(-50 + rnd_below(w.w, 101));
    by = (float)(-30 + rnd_below(w1.x, 61));
  }
  float bvx = 0.0f, bvy = 0.0f;
  if (r.change_ball_velocity) {                          // :202-212
    bool ok = vel_candidate(p, r, bx, by, w1.y, w1.z, bvx, bvy);   // try 0 rides in block 1
    U4 wb{0, 0, 0, 0};
    for (int k = 1; k < S2D_MAX_VEL_TRIES && !ok; ++k) {  // bounded: every lane leaves the loop
      uint32_t ws, wd;                                   // two candidates per Philox call
      if ((k - 1) & 1) { ws = wb.z; wd = wb.w; }
      else { wb = s2d_draw(p, gid_lo, gid_hi, c0, S2D_ST_RESET, 2 + ((k - 1) >> 1)); ws = wb.x; wd = wb.y; }
      ok = vel_candidate(p, r, bx, by, ws, wd, bvx, bvy);
    }
    if (!ok) { bvx = 0.0f; bvy = 0.0f; }
  } else {                                               // :213-216
    float sn, cs;
    sincos_deg(r.ball_direction, sn, cs);
    bvx = r.ball_speed * cs; bvy = r.ball_speed * sn;
  }
  o.bx = bx; o.by = by; o.bvx = bvx; o.bvy = bvy;
  return o;
}
// The same sample drawn by a whole wave together (call it from wave-uniform control flow, with ALL 64 lanes; `need` = this lane
// wants one).  The sequential loop above runs as long as the unluckiest of its lanes keeps being rejected -- the acceptance rate of
// the stock task is ~1/2, so a wave of 64 needs ~8 rounds and the unluckiest of a launch's 200 000 draws ~17.  Here, after
// everybody's try 0, the lanes of the wave are handed to the envs still pending: h = 64 / pending (a power of two) lanes evaluate
// tries k .. k + h - 1 of one env side by side, the env takes the FIRST accepted one -- what the sequential loop would have
// reached -- and k advances by h: three rounds for almost every wave.  `scratch` = 64 wave-private LDS words.
S2D_DEV void coop_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
S2D_DEV ResetSample reset_sample_coop(const S2DHot& p, const S2DRare& r, uint32_t gid_lo, uint32_t gid_hi, uint32_t c0, bool need,
                                      int lane, uint32_t* scratch) {
  U4 w = s2d_draw(p, gid_lo, gid_hi, c0, S2D_ST_RESET, 0);
  U4 w1 = s2d_draw(p, gid_lo, gid_hi, c0, S2D_ST_RESET, 1);
  ResetSample o;
  o.px = (float)(-50 + rnd_below(w.x, 101));             // :173
  o.py = (float)(-30 + rnd_below(w.y, 61));              // :174
  o.body = norm_deg((float)rnd_below(w.z, 361));         // :175
  float bx = r.ball_position_x, by = r.ball_position_y;
  if (r.change_ball_position) {                          // :176-181
    bx = (float)(-50 + rnd_below(w.w, 101));
    by = (float)(-30 + rnd_below(w1.x, 61));
  }
  float bvx = 0.0f, bvy = 0.0f;
  if (r.change_ball_velocity) {                          // :202-212 (uniform: r is the same for every lane)
    bool ok = vel_candidate(p, r, bx, by, w1.y, w1.z, bvx, bvy);
    bool pending = need && !ok;
    int k = 1;                                           // next try of every pending lane (they advance together)
    unsigned long long P = __ballot(pending);
    while (P != 0ull && k < S2D_MAX_VEL_TRIES) {
      const int np = (int)__popcll(P);
      int hs = 0;                                        // h = 2^hs helpers per pending env, h * np <= 64
      while ((np << (hs + 1)) <= 64) ++hs;
      const int mine = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(P >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)P, 0u));   // my rank among the pending
      if (pending) scratch[mine] = (uint32_t)lane;
      coop_fence();
      const int q = lane >> hs, t = lane & ((1 << hs) - 1);
      const bool helper = q < np;
      const int owner = helper ? (int)scratch[q] : lane;
      coop_fence();
      const uint32_t ogl = (uint32_t)__shfl((int)gid_lo, owner), ogh = (uint32_t)__shfl((int)gid_hi, owner);
      const uint32_t oc0 = (uint32_t)__shfl((int)c0, owner);
      const float obx = __shfl(bx, owner), oby = __shfl(by, owner);
      float cvx = 0.0f, cvy = 0.0f;
      bool ok_t = false;
      if (helper && k + t < S2D_MAX_VEL_TRIES) ok_t = vel_try(p, r, ogl, ogh, oc0, obx, oby, k + t, cvx, cvy);
      const unsigned long long A = __ballot(ok_t);
      int src = lane;
      bool got = false;
      if (pending) {
        const unsigned long long grp = (A >> (mine << hs)) & (hs == 6 ? ~0ull : ((1ull << (1 << hs)) - 1ull));
        if (grp != 0ull) { got = true; src = (mine << hs) + (int)__ffsll((long long)grp) - 1; }
      }
      const float nvx = __shfl(cvx, src), nvy = __shfl(cvy, src);
      if (got) { bvx = nvx; bvy = nvy; ok = true; pending = false; }
      k += 1 << hs;
      P = __ballot(pending);
    }
    if (!ok) { bvx = 0.0f; bvy = 0.0f; }
  } else {                                               // :213-216
    float sn, cs;
    sincos_deg(r.ball_direction, sn, cs);
    bvx = r.ball_speed * cs; bvy = r.ball_speed * sn;
  }
  o.bx = bx; o.by = by; o.bvx = bvx; o.bvy = bvy;
  return o;
}
// trainer (move ball) (move player) (recover), then the command-less cycle (soccer_2d_env.py:186-197)
template <bool NOISE>
S2D_DEV float reset_apply(const S2DHot& p, const S2DRare* __restrict__ rp, Env& e, uint32_t gid_lo, uint32_t gid_hi,
                          const ResetSample& o, float recover_init, uint32_t key) {
  e.step_number = 0;                                     // :172
  e.bx = o.bx; e.by = o.by; e.bvx = o.bvx; e.bvy = o.bvy;
  e.px = o.px; e.py = o.py; e.body = o.body; e.vx = 0.0f; e.vy = 0.0f;
  e.stamina = p.stamina_max; e.recovery = recover_init;
  e.effort = p.effort_init; e.capacity = p.stamina_capacity;
  NoiseIn nz{0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  if (NOISE) nz = noise_prepare(p, gid_lo, gid_hi, key, S2D_ST_NOISE_RESET, false);   // keyed like the sample itself
  return sim_cycle<NOISE, false>(p, rp, e, S2D_CMD_NONE, CmdPrep{0.0f, 0.0f, 0.0f}, nz);
}
// The env a reset leaves behind -- trainer moves + recover + the command-less cycle, whose noise is keyed by the
// reset's own key -- is a pure function of (env id, key), so it can be prepared together with the sample (batched,
// off the critical path) and a reset becomes a register copy plus the cycle's tick.  NextEpisode = that prepared state.
struct NextEpisode { float px, py, vx, vy, body, stamina, effort, recovery, capacity, bx, by, bvx, bvy; };
template <bool NOISE>
S2D_DEV NextEpisode episode_prepare(const S2DHot& p, const S2DRare* __restrict__ rp, const S2DRare& r, uint32_t gid_lo,
                                    uint32_t gid_hi, uint32_t key) {
  const ResetSample o = reset_sample(p, r, gid_lo, gid_hi, key);
  Env t{};
  reset_apply<NOISE>(p, rp, t, gid_lo, gid_hi, o, r.recover_init, key);
  return NextEpisode{t.px, t.py, t.vx, t.vy, t.body, t.stamina, t.effort, t.recovery, t.capacity, t.bx, t.by, t.bvx, t.bvy};
}
template <bool NOISE>
S2D_DEV NextEpisode episode_prepare_coop(const S2DHot& p, const S2DRare* __restrict__ rp, const S2DRare& r, uint32_t gid_lo,
                                         uint32_t gid_hi, uint32_t key, bool need, int lane, uint32_t* scratch) {
  const ResetSample o = reset_sample_coop(p, r, gid_lo, gid_hi, key, need, lane, scratch);
  Env t{};
  reset_apply<NOISE>(p, rp, t, gid_lo, gid_hi, o, r.recover_init, key);
  return NextEpisode{t.px, t.py, t.vx, t.vy, t.body, t.stamina, t.effort, t.recovery, t.capacity, t.bx, t.by, t.bvx, t.bvy};
}
S2D_DEV void episode_begin(Env& e, const NextEpisode& q) {
  e.px = q.px; e.py = q.py; e.vx = q.vx; e.vy = q.vy; e.body = q.body;
  e.stamina = q.stamina; e.effort = q.effort; e.recovery = q.recovery; e.capacity = q.capacity;
  e.bx = q.bx; e.by = q.by; e.bvx = q.bvx; e.bvy = q.bvy;
  e.step_number = 0;                                     // reach_ball_env.py:172
  e.cycle = (int)((uint32_t)e.cycle + 1u);               // the command-less cycle (soccer_2d_env.py:190)
  e.episode = (int)((uint32_t)e.episode + 1u);
}
// The observation a reset returns and the carry it seeds (reach_ball_env.py:163-168) are functions of that
// prepared state alone, so they are prepared with it: o[0..9] = the new episode's first row, dist / rel = the carry.
struct FirstObs { float o[S2D_OBS_DIM], dist, rel; };
S2D_DEV FirstObs first_obs(const S2DHot& p, const NextEpisode& q) {
  FirstObs f;
  observe_ball(p, q.bx, q.by, q.bvx, q.bvy, f.o);
  f.rel = observe_player(p, q.px, q.py, q.body, q.bx, q.by, f.o);
  f.dist = hypot2(q.bx - q.px, q.by - q.py);
  return f;
}
template <bool NOISE>
S2D_DEV float env_reset(const S2DHot& p, const S2DRare* __restrict__ rp, Env& e, uint32_t gid_lo,
                        uint32_t gid_hi) {
  const S2DRare r = *rp;                                 // one bulk scalar load for the whole path
  const uint32_t key = reset_key(e);
  ResetSample o = reset_sample(p, r, gid_lo, gid_hi, key);
  e.episode = (int)key;
  return reset_apply<NOISE>(p, rp, e, gid_lo, gid_hi, o, r.recover_init, key);
}
