// s2d_match.hip -- 11v11 full-match engine for MI355X (gfx950): kernels + C ABI of
// include/s2d_match.h.  Rules = rcssserver's, restated (EXT); the tests
// hold an independent CPU restatement of the same rules which this file matches bit for bit.
//
// Mapping: ONE MATCH PER HALF-WAVE.  Lanes 0..21 of a 32-lane half are the players
// (0..10 left team, 11..21 right team), lane 22 is the ball, lanes 23..31 idle; a wave runs
// two matches, a 256-thread workgroup eight.  All per-object work (commands, movement,
// stamina) is lane-local; the cross-object steps use half-wave shuffles (ds_bpermute, width
// 32) and 64-bit ballots split per half:
//   * kick / tackle impulses are summed into the ball in player order (fixed order = fixed
//     fp32 result),
//   * collisions are Jacobi passes: every lane scans the 23 objects of its match, proposes
//     the symmetric contact position for each overlap and takes the mean (no ordering, no
//     atomics), up to 10 passes,
//   * referee decisions (goal, ball out, restarts, offside, half time) are evaluated
//     redundantly by every lane of the half from broadcast values, so there is no divergence
//     inside a match and no serial "referee lane",
//   * nearest-player-to-ball per team is a 22-step scan in index order.
// Every lane stays active for the whole kernel (shuffles need their source lanes), matches
// beyond N only skip their loads and stores.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>

#include "s2d_device.h"
#include "../../include/s2d_match.h"

#define S2D_API extern "C" __attribute__((visibility("default")))

static constexpr int kMBlock = 256;
static constexpr int kHalf = 32;
static constexpr int kEnvsPerBlock = kMBlock / kHalf;
static constexpr int NP = S2D_MATCH_PLAYERS;
static constexpr int BALL = S2D_MATCH_BALL;
static constexpr int SLOTS = S2D_MATCH_SLOTS;

enum { S2D_ST_TACKLE = 4, S2D_ST_CATCH = 5, S2D_ST_TYPES = 6 };
enum { SIDE_NONE = 0, SIDE_LEFT = 1, SIDE_RIGHT = 2 };
enum { MF_X, MF_Y, MF_VX, MF_VY, MF_BODY, MF_STAMINA, MF_EFFORT, MF_RECOVERY, MF_CAPACITY, MF_TACKLE, MF_CATCH_BAN, MF_CARD, MF_OBJ_PLANES };
// Per-slot parameters (heterogeneous PlayerTypes, idl/service.proto:1697-1732): a [PT_WORDS][32] table,
// one column per lane of the half-wave.  Column 22 (the ball) holds ball_size / ball_decay in the
// size / decay rows, so the collision scan and the decay treat players and ball alike.
enum { PT_SPEED_MAX, PT_SPEED_MAX2, PT_STAMINA_INC, PT_DECAY, PT_INERTIA, PT_DASH_RATE, PT_SIZE, PT_INV_KICK_MARGIN,
       PT_KICKABLE_AREA2, PT_KICK_RAND, PT_EXTRA_STAMINA, PT_EFFORT_MAX, PT_EFFORT_MIN, PT_KICK_RATE, PT_CATCH_LEN, PT_WORDS };
enum { ME_CYCLE, ME_MODE, ME_MODE_SIDE, ME_SCORE_L, ME_SCORE_R, ME_LAST_TOUCH, ME_TIMER, ME_OFFSIDE, ME_REWARD, ME_NEAREST_L,
       ME_NEAREST_R, ME_HOLDER, ME_MOVES, ME_TAKER, ME_LAST_KICKER, ME_STOPPED, ME_TICK, ME_ENV_PLANES };

struct MParams {   // every field rounded once on the host (double -> float); per-PlayerType values live in the PT table
  float half_l, half_w, ball_size, player_rand, ball_rand;
  float player_accel_max, player_accel_max2, ball_speed_max, ball_speed_max2;
  float ball_accel_max, ball_accel_max2;
  float stamina_max, stamina_capacity;
  float recover_init, recover_dec_thr_value, recover_min, recover_dec;
  float effort_dec_thr_value, effort_dec, effort_inc_thr_value, effort_inc;
  float max_dash_power, min_dash_power, max_dash_angle, min_dash_angle;
  float dash_angle_step, inv_dash_angle_step, side_dash_rate, back_dash_rate, max_moment, min_moment;
  float collision_vel_rate;
  float max_power, min_power, inv_max_power;
  float tackle_dist, tackle_back_dist, tackle_width, tackle_power_rate, max_tackle_power, max_back_tackle_power;
  float tackle_reach2;   // beyond this squared distance a tackle fails for certain (see m_tackle)
  float goal_half_width, offside_area2, free_kick_distance, inv_speed_decay;
  float catch_half_w, catch_probability, max_catch_angle, min_catch_angle, pen_x, pen_half_w;
  int tackle_cycles, half_time_cycles, nr_normal_halfs, drop_ball_time, use_offside, catch_ban_cycle, goalie_max_moves;
  int after_goal_wait, kick_off_wait, back_passes, free_kick_faults;
  int stopped_clock, announce_wait, foul_cycles; float foul_detect_probability;
  int nr_extra_halfs, extra_half_cycles, golden_goal;
  int penalty_shoot_outs, pen_before_setup_wait, pen_ready_wait, pen_taken_wait, pen_nr_kicks, pen_max_extra_kicks; float pen_spot_x;
  int illegal_defense_number, illegal_defense_duration; float ill_x, ill_half_w;   // the strip: beyond ill_x on the own side, |y| < ill_half_w
  int pen_allow_mult_kicks, pen_random_winner;
  int total_cycles, end_cycles;   // derived: end of the normal time, end of the last period (= total_cycles without extra halves)
  int auto_reset, noise;
  uint32_t seed_lo, seed_hi, gid_lo, gid_hi;
};
typedef float PTab[kHalf];   // one row of the per-slot table

// The stock configuration (s2d_match_default_config) as compile-time constants, with the derivations of mparams_from_config restated
// as constant expressions: the kernels are instantiated once over MParams (any configuration; ~70 words read from LDS where they
// are used) and once over MStock, where every use is an immediate -- no LDS reads or waits for them, dead branches (no dash-angle
// quantisation off, no unlimited stamina capacity, ...) compiled out.  m_is_stock() compares an engine's derived MParams with
// these bit for bit; anything else runs the general instantiation.  Per-engine words (seed, env ids, switches) stay variables.
#ifndef S2D_STOCK_SHOOT_OUTS
#define S2D_STOCK_SHOOT_OUTS 1   // rcssserver's penalty_shoot_outs (experiment builds: 0)
#endif
#ifndef S2D_STOCK_EXTRA_HALFS
#define S2D_STOCK_EXTRA_HALFS 2   // rcssserver's nr_extra_halfs
#endif
#define M_STOCK_PHYSICS \
  static constexpr float half_l = (float)52.5, half_w = (float)34.0, ball_size = (float)0.085, player_rand = (float)0.1, ball_rand = (float)0.05; \
  static constexpr float player_accel_max = (float)1.0, player_accel_max2 = player_accel_max * player_accel_max; \
  static constexpr float ball_speed_max = (float)3.0, ball_speed_max2 = ball_speed_max * ball_speed_max; \
  static constexpr float ball_accel_max = (float)2.7, ball_accel_max2 = ball_accel_max * ball_accel_max; \
  static constexpr float stamina_max = (float)8000.0, stamina_capacity = (float)130600.0; \
  static constexpr float recover_init = (float)1.0, recover_dec_thr_value = (float)(0.3 * 8000.0), recover_min = (float)0.5, recover_dec = (float)0.002; \
  static constexpr float effort_dec_thr_value = (float)(0.3 * 8000.0), effort_dec = (float)0.005; \
  static constexpr float effort_inc_thr_value = (float)(0.6 * 8000.0), effort_inc = (float)0.01; \
  static constexpr float max_dash_power = (float)100.0, min_dash_power = (float)0.0, max_dash_angle = (float)180.0, min_dash_angle = (float)-180.0; \
  static constexpr float dash_angle_step = (float)1.0, inv_dash_angle_step = (float)(1.0 / 1.0); \
  static constexpr float side_dash_rate = (float)0.4, back_dash_rate = (float)0.6, max_moment = (float)180.0, min_moment = (float)-180.0; \
  static constexpr float collision_vel_rate = (float)-0.1; \
  static constexpr float max_power = (float)100.0, min_power = (float)-100.0, inv_max_power = (float)(1.0 / 100.0); \
  static constexpr float tackle_dist = (float)2.0, tackle_back_dist = (float)0.0, tackle_width = (float)1.25, tackle_power_rate = (float)0.027; \
  static constexpr float max_tackle_power = (float)100.0, max_back_tackle_power = (float)0.0; \
  static constexpr float tackle_reach2 = (float)(1.01 * (2.0 * 2.0 + 1.25 * 1.25)); \
  static constexpr float goal_half_width = (float)(14.02 * 0.5), offside_area2 = (float)(2.5 * 2.5), free_kick_distance = (float)9.15; \
  static constexpr float inv_speed_decay = (float)(1.0 / (3.0 * 0.94)); \
  static constexpr float catch_half_w = (float)(1.0 * 0.5), catch_probability = (float)1.0, max_catch_angle = (float)90.0, min_catch_angle = (float)-90.0; \
  static constexpr float pen_x = (float)(52.5 - 16.5), pen_half_w = (float)20.16;
struct MStock {
  M_STOCK_PHYSICS
  static constexpr int tackle_cycles = 10, half_time_cycles = 3000, nr_normal_halfs = 2, drop_ball_time = 100, use_offside = 1, catch_ban_cycle = 5;
  static constexpr int goalie_max_moves = 2, after_goal_wait = 50, kick_off_wait = 0, back_passes = 1, free_kick_faults = 1;
  static constexpr int stopped_clock = 1, announce_wait = 30, foul_cycles = 5;
  static constexpr float foul_detect_probability = (float)0.5;
  static constexpr int nr_extra_halfs = S2D_STOCK_EXTRA_HALFS, extra_half_cycles = 1000, golden_goal = 0;
  static constexpr int pen_before_setup_wait = 10, pen_ready_wait = 10, pen_taken_wait = 150, pen_nr_kicks = 5, pen_max_extra_kicks = 5;
  static constexpr float pen_spot_x = (float)(52.5 - 42.5);
  static constexpr int illegal_defense_number = 0, illegal_defense_duration = 20;   // (off, as in the stock server: the rule's code folds away)
  static constexpr int pen_allow_mult_kicks = 1, pen_random_winner = 0;
  static constexpr float ill_x = (float)(52.5 - 16.5), ill_half_w = (float)(40.32 * 0.5);
  static constexpr int total_cycles = half_time_cycles * nr_normal_halfs, end_cycles = total_cycles + extra_half_cycles * nr_extra_halfs;
  int auto_reset, noise;
  int penalty_shoot_outs;   // per engine like the two above: an engine that differs from the stock rules only in this word keeps this kernel
  uint32_t seed_lo, seed_hi, gid_lo, gid_hi;
};
// The general parameter block with the IllegalDefense_ rule compiled out (number = 0: the stock server's setting and nearly every
// engine's): read through this type, `p.illegal_defense_number` is the constant below and the rule's per-cycle count folds away (as a
// run-time test on an LDS word in front of the quick exit it cost the general kernel 4 %).  Same layout as MParams: no data members.
struct MParamsNoIll : MParams { static constexpr int illegal_defense_number = 0; };
static_assert(sizeof(MParamsNoIll) == sizeof(MParams), "MParamsNoIll adds no data");
// The same physics and rules with the SCHEDULE of the match -- how long things last, how many there are of them -- as per-engine
// words: a learner's engine with short halves, no extra time or other waits differs from the stock configuration in these words
// only and would otherwise run the general instantiation (1.79 G against 2.00 G, profiles/r04/match_schedule_words.txt).  They sit in
// tests of rare branches and in one compare per cycle -- scalar registers -- but cost the fully constant kernel 2-6 %, so both exist.
#define M_SCHEDULE_INTS(X) X(penalty_shoot_outs) X(half_time_cycles) X(nr_normal_halfs) X(drop_ball_time) X(after_goal_wait) X(kick_off_wait) \
  X(announce_wait) X(nr_extra_halfs) X(extra_half_cycles) X(golden_goal) X(total_cycles) X(end_cycles) X(pen_before_setup_wait) \
  X(pen_ready_wait) X(pen_taken_wait) X(pen_nr_kicks) X(pen_max_extra_kicks)
struct MStockSched {
  M_STOCK_PHYSICS
  static constexpr int tackle_cycles = 10, use_offside = 1, catch_ban_cycle = 5, goalie_max_moves = 2, back_passes = 1, free_kick_faults = 1;
  static constexpr int stopped_clock = 1, foul_cycles = 5;
  static constexpr float foul_detect_probability = (float)0.5;
  static constexpr float pen_spot_x = (float)(52.5 - 42.5);
  static constexpr int illegal_defense_number = 0, illegal_defense_duration = 20;   // (off, as in the stock server: the rule's code folds away)
  static constexpr int pen_allow_mult_kicks = 1, pen_random_winner = 0;
  static constexpr float ill_x = (float)(52.5 - 16.5), ill_half_w = (float)(40.32 * 0.5);
  int auto_reset, noise;
  uint32_t seed_lo, seed_hi, gid_lo, gid_hi;
#define X(name) int name;
  M_SCHEDULE_INTS(X)
#undef X
};
#define M_CONFIG_FLOATS(X) X(half_l) X(half_w) X(ball_size) X(player_rand) X(ball_rand) X(player_accel_max) X(player_accel_max2) \
  X(ball_speed_max) X(ball_speed_max2) X(ball_accel_max) X(ball_accel_max2) X(stamina_max) X(stamina_capacity) X(recover_init) \
  X(recover_dec_thr_value) X(recover_min) X(recover_dec) X(effort_dec_thr_value) X(effort_dec) X(effort_inc_thr_value) X(effort_inc) \
  X(max_dash_power) X(min_dash_power) X(max_dash_angle) X(min_dash_angle) X(dash_angle_step) X(inv_dash_angle_step) X(side_dash_rate) \
  X(back_dash_rate) X(max_moment) X(min_moment) X(collision_vel_rate) X(max_power) X(min_power) X(inv_max_power) X(tackle_dist) \
  X(tackle_back_dist) X(tackle_width) X(tackle_power_rate) X(max_tackle_power) X(max_back_tackle_power) X(tackle_reach2) \
  X(goal_half_width) X(offside_area2) X(free_kick_distance) X(inv_speed_decay) X(catch_half_w) X(catch_probability) \
  X(max_catch_angle) X(min_catch_angle) X(pen_x) X(pen_half_w) X(foul_detect_probability) X(pen_spot_x) X(ill_x) X(ill_half_w)
#define M_CONFIG_INTS(X) X(tackle_cycles) X(half_time_cycles) X(nr_normal_halfs) X(drop_ball_time) X(use_offside) X(catch_ban_cycle) \
  X(goalie_max_moves) X(after_goal_wait) X(kick_off_wait) X(back_passes) X(free_kick_faults) X(stopped_clock) X(announce_wait) X(foul_cycles) \
  X(nr_extra_halfs) X(extra_half_cycles) X(golden_goal) X(total_cycles) X(end_cycles) X(pen_before_setup_wait) \
  X(pen_ready_wait) X(pen_taken_wait) X(pen_nr_kicks) X(pen_max_extra_kicks) X(illegal_defense_number) X(illegal_defense_duration) X(pen_allow_mult_kicks) X(pen_random_winner)
// every configuration word of MParams is in one of the two lists (the remaining seven are the per-engine words)
static_assert(sizeof(MParams) == 4 * (56 + 28 + 7), "a field was added to MParams: list it in M_CONFIG_FLOATS / M_CONFIG_INTS and in MStock");

// The per-slot table of an engine whose 22 players are all of the stock PlayerType (the default: s2d_match_default_config), with the
// same spelling as the LDS table -- types[ROW][lane] -- but every entry an immediate: a cycle reads about ten of them per lane, each
// an LDS load and a wait in the general kernel.  Column 22 (the ball) differs in the size / decay rows; pad lanes (23..31) get the
// players' values (the LDS table holds zeros there; nothing reads them).  kickable_area2 is the one derived word that is not a
// constant expression (the host searches the largest float whose root does not exceed the kickable area): the engine's own value.
template <class T> struct TypesAreConst { static constexpr bool value = false; };
struct MStockTypes {
  static constexpr float speed_max = (float)1.05, speed_max2 = speed_max * speed_max, stamina_inc = (float)45.0, decay = (float)0.4;
  static constexpr float inertia = (float)5.0, dash_rate = (float)0.006, size = (float)0.3, inv_kick_margin = (float)(1.0 / 0.7);
  static constexpr float kick_rand = (float)0.1, extra_stamina = (float)50.0, effort_max = (float)1.0, effort_min = (float)0.6;
  static constexpr float kick_rate = (float)0.027, catch_len = (float)(1.2 * 1.0), ball_decay = (float)0.94;
  float kickable_area2;
  struct Row {
    int row; float ka2;
    S2D_DEV float operator[](int l) const {
      switch (row) {
        case PT_SPEED_MAX: return speed_max; case PT_SPEED_MAX2: return speed_max2; case PT_STAMINA_INC: return stamina_inc;
        case PT_DECAY: return l == BALL ? ball_decay : decay; case PT_INERTIA: return inertia; case PT_DASH_RATE: return dash_rate;
        case PT_SIZE: return l == BALL ? MStock::ball_size : size; case PT_INV_KICK_MARGIN: return inv_kick_margin;
        case PT_KICKABLE_AREA2: return ka2; case PT_KICK_RAND: return kick_rand; case PT_EXTRA_STAMINA: return extra_stamina;
        case PT_EFFORT_MAX: return effort_max; case PT_EFFORT_MIN: return effort_min; case PT_KICK_RATE: return kick_rate;
        default: return catch_len;
      }
    }
  };
  S2D_DEV Row operator[](int row) const { return Row{row, kickable_area2}; }
};
template <> struct TypesAreConst<MStockTypes> { static constexpr bool value = true; };

struct MObj { float x, y, vx, vy, body, stamina, effort, recovery, capacity; int tackle, catch_ban, card; };
// Per-match words every cycle reads (registers; the same value in the 32 lanes of the match's half-wave) ...
struct MGame { int cycle, mode, mode_side, last_touch, offside; float reward; int done, nearest_l, nearest_r;
               int tick;               /* cycles since the reset, stopped ones included (the Philox counter) */
               int to_half;            /* cycles until the clock reaches the next multiple of half_time_cycles (derived at load: not a
                                          state word) -- a countdown instead of an integer modulo by a run-time divisor every cycle */
               int timer;              /* set-play / dead-ball timer: read and counted in every cycle of a waiting match, whose wave
                                          is the one the launch waits for -- a register, not an LDS round trip */ };
// ... and the ones only events touch (goals, set plays, catches, kicks by the taker, a standing clock): they live in LDS, one row
// per match, so that they do not occupy registers in a kernel that sits on its 128-VGPR cap (4 resident waves per SIMD).  All
// 32 lanes of a half read and write the same word with the same value; LDS operations of one wave execute in order.
struct MRare { int score_l, score_r;
               int holder, moves; /* 1 + index of the goalie holding a caught ball (0 = nobody), his remaining moves */
               int taker, last_kicker; /* 1 + index (0 = nobody): set-play taker not yet followed by another touch; last Kick-command kicker */
               int stopped;            /* WorldModel.stoped_cycle */ };
// cycles until `cycle` is the next multiple of h (> 0): the value of the countdown for a clock that reads `cycle`
S2D_DEV int cycles_to_half(int cycle, int h) {
  const int rem = cycle % h;                             // C remainder: negative for a clock that has wrapped
  return rem < 0 ? -rem : h - rem;
}
// ... and to the end of the period the clock is in: a normal half, or (from the end of the normal time on, when extra halves
// exist) an extra half
template <class P> S2D_DEV int cycles_to_period_end(const P& p, int cycle) {
  const int total = p.total_cycles;
  if (p.nr_extra_halfs > 0 && cycle >= total) return cycles_to_half(cycle - total, p.extra_half_cycles);
  return cycles_to_half(cycle, p.half_time_cycles);
}

__constant__ float kFormX[11] = {-50.0f, -35.0f, -35.0f, -35.0f, -35.0f, -20.0f, -20.0f, -20.0f, -20.0f, -10.5f, -10.5f};
__constant__ float kFormY[11] = {0.0f, -20.0f, -7.0f, 7.0f, 20.0f, -22.0f, -8.0f, 8.0f, 22.0f, -6.0f, 6.0f};

template <class P> S2D_DEV U4 m_draw(const P& p, uint32_t gl, uint32_t gh, uint32_t cyc, uint32_t stream, uint32_t block) {
  // The match id and the seed do not change during a launch, so the compiler would compute the loop-invariant part of the first
  // round (two 64-bit products) and all ten round keys once per launch and keep them -- 30 registers this kernel does not have:
  // they were spilled to scratch and reloaded at every draw.  Opaque copies keep the whole block function inside the loop.
  uint32_t k0 = __builtin_amdgcn_readfirstlane(p.seed_lo), k1 = __builtin_amdgcn_readfirstlane(p.seed_hi);
  asm volatile("" : "+v"(gl), "+v"(gh), "+s"(k0), "+s"(k1));
  return philox4x32_10(gl, gh, cyc, (stream << 16) | block, k0, k1);
}
S2D_DEV int side_of(int i) { return i < 11 ? SIDE_LEFT : SIDE_RIGHT; }
S2D_DEV int other_side(int s) { return s == SIDE_LEFT ? SIDE_RIGHT : SIDE_LEFT; }
// TimeOver, and the two modes only an operator sets (Pause, Human: idl/service.proto:280-281 -- write them into the mode plane to hold
// a match, PlayOn or a set play to let it go on): nobody acts, nothing is decided, the clock stands
constexpr uint32_t kHaltedModes = (1u << S2D_GM_TIME_OVER) | (1u << S2D_GM_PAUSE) | (1u << S2D_GM_HUMAN);
S2D_DEV bool is_halted(int mode) { return ((kHaltedModes >> (mode & 31)) & 1u) != 0u; }
S2D_DEV bool is_setplay(int mode) { return mode != S2D_GM_PLAY_ON && !is_halted(mode); }
// Mode classes as bit masks over GameModeType (every value used is < 32): one shift + and instead of a chain of compares.
// announcements: a dead ball named after the offending side; after announce_wait cycles the referee awards the restart
constexpr uint32_t kAnnounceModes = (1u << S2D_GM_OFF_SIDE) | (1u << S2D_GM_BACK_PASS) | (1u << S2D_GM_FREE_KICK_FAULT) |
                                    (1u << S2D_GM_CATCH_FAULT) | (1u << S2D_GM_FOUL_CHARGE) | (1u << S2D_GM_ILLEGAL_DEFENSE) |
                                    (1u << S2D_GM_FOUL_PUSH) | (1u << S2D_GM_FOUL_MULTIPLE_ATTACKER) | (1u << S2D_GM_FOUL_BALL_OUT);
// modes in which nobody may play the ball
constexpr uint32_t kPeriodEndModes = (1u << S2D_GM_FIRST_HALF_OVER) | (1u << S2D_GM_EXTEND_HALF);   // "half_time", "time_extended"
// the shoot-out's modes (idl/service.proto:290-297)
constexpr uint32_t kPenaltyModes = (1u << S2D_GM_PENALTY_SETUP) | (1u << S2D_GM_PENALTY_READY) | (1u << S2D_GM_PENALTY_TAKEN) |
                                   (1u << S2D_GM_PENALTY_MISS) | (1u << S2D_GM_PENALTY_SCORE) | (1u << S2D_GM_PENALTY_ONFIELD) |
                                   (1u << S2D_GM_PENALTY_FOUL);
constexpr uint32_t kDeadBallModes = kAnnounceModes | (1u << S2D_GM_AFTER_GOAL) | (1u << S2D_GM_BEFORE_KICK_OFF) |
                                    kPeriodEndModes | (1u << S2D_GM_GOALIE_CATCH) |
                                    (kPenaltyModes & ~((1u << S2D_GM_PENALTY_READY) | (1u << S2D_GM_PENALTY_TAKEN)));
// modes in which the clock stands still (with stopped_clock): WorldModel.cycle keeps its value, stoped_cycle counts
constexpr uint32_t kClockStandsModes = kAnnounceModes | (1u << S2D_GM_BEFORE_KICK_OFF) | (1u << S2D_GM_AFTER_GOAL) |
                                       kPeriodEndModes | kHaltedModes | kPenaltyModes;
static_assert(S2D_GM_EXTEND_HALF < 32, "mode masks are 32 bits wide");
S2D_DEV bool in_modes(int mode, uint32_t mask) { return ((mask >> (mode & 31)) & 1u) != 0u; }
S2D_DEV bool is_announcement(int mode) { return in_modes(mode, kAnnounceModes); }
S2D_DEV bool ball_dead(int mode) { return in_modes(mode, kDeadBallModes); }
S2D_DEV bool clock_stands(int mode) { return in_modes(mode, kClockStandsModes); }
S2D_DEV bool is_period_end(int mode) { return in_modes(mode, kPeriodEndModes); }
#ifdef S2D_NO_SHOOTOUT   // experiment builds: the cycle without the shoot-out's code (the mode is never entered)
S2D_DEV bool is_penalty(int) { return false; }
constexpr bool kShootOut = false;
#else
S2D_DEV bool is_penalty(int mode) { return in_modes(mode, kPenaltyModes); }
constexpr bool kShootOut = true;
#endif
// the shoot-out's state in the set-play word (include/s2d_match.h): kicks taken / goals of a side
S2D_DEV int pen_kicks(int w, int side) { return (w >> (side == SIDE_LEFT ? 12 : 16)) & 15; }
S2D_DEV int pen_goals(int w, int side) { return (w >> (side == SIDE_LEFT ? 20 : 24)) & 15; }
// is the shoot-out decided (or used up) after the kicks counted in w?  (oracle: pen_over)
template <class P> S2D_DEV bool pen_over(const P& p, int w) {
  const int kl = pen_kicks(w, SIDE_LEFT), kr = pen_kicks(w, SIDE_RIGHT), gl = pen_goals(w, SIDE_LEFT), gr = pen_goals(w, SIDE_RIGHT);
  const int nr = p.pen_nr_kicks;
  if (kl <= nr && kr <= nr) {                            // the regular kicks: over as soon as one side cannot catch up
    if (gl > gr + (nr - kr) || gr > gl + (nr - kl)) return true;
    if (kl == nr && kr == nr) return gl != gr || p.pen_max_extra_kicks <= 0;
    return false;
  }
  if (kl == kr) return gl != gr || kl >= nr + p.pen_max_extra_kicks;   // pairs of extra kicks
  return false;
}
S2D_DEV float hbcast(float v, int src) { return __shfl(v, src, kHalf); }
S2D_DEV int hbcasti(int v, int src) { return __shfl(v, src, kHalf); }
// The same broadcast from a lane known at compile time (the ball's): two v_readlane and a select -- a few cycles -- where the
// general shuffle is a ds_bpermute through the LDS pipeline (~100 cycles of latency a wave of this kernel cannot hide).  Called
// in uniform control flow only.
template <int SRC> S2D_DEV int hbcasti_c(int v, int half) {
  const int lo = __builtin_amdgcn_readlane(v, SRC), hi = __builtin_amdgcn_readlane(v, SRC + kHalf);
  return half ? hi : lo;
}
template <int SRC> S2D_DEV float hbcast_c(float v, int half) { return __int_as_float(hbcasti_c<SRC>(__float_as_int(v), half)); }
// 32-bit ballot of this lane's half
S2D_DEV uint32_t hballot(bool pred, int half) { return (uint32_t)(__ballot(pred) >> (half * kHalf)); }

S2D_DEV void m_place(MObj& o, int l, int kickoff_side) {   // place_formation() for lane l
  if (l < NP && o.card >= S2D_CARD_RED) return;            // sent off: stays where he was parked
  if (l < NP) {
    int k = l % 11; bool left = l < 11;
    o.x = left ? kFormX[k] : -kFormX[k]; o.y = kFormY[k];
    o.vx = 0.0f; o.vy = 0.0f; o.body = left ? 0.0f : 180.0f; o.tackle = 0; o.catch_ban = 0;
    if (kickoff_side == SIDE_LEFT && l == 10) { o.x = -0.4f; o.y = 0.0f; }
    if (kickoff_side == SIDE_RIGHT && l == 21) { o.x = 0.4f; o.y = 0.0f; }
  } else if (l == BALL) {
    o.x = 0.0f; o.y = 0.0f; o.vx = 0.0f; o.vy = 0.0f;
  }
}
template <class P> S2D_DEV void m_recover(const P& p, float effort_max, MObj& o, bool with_capacity) {
  o.stamina = p.stamina_max; o.effort = effort_max; o.recovery = p.recover_init;
  if (with_capacity) o.capacity = p.stamina_capacity;
}
template <class P> S2D_DEV void m_reset(const P& p, float effort_max, MObj& o, MGame& g, MRare& r, int l) {
  o = MObj{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  g = MGame{0, p.kick_off_wait > 0 ? S2D_GM_BEFORE_KICK_OFF : S2D_GM_KICK_OFF, SIDE_LEFT, 0, 0, 0.0f, 0, 10, 20, 0, p.half_time_cycles, 0};
  r = MRare{0, 0, 0, 0, 0, 0, 0};
  if (l < NP) m_recover(p, effort_max, o, true);
  m_place(o, l, SIDE_LEFT);
}
template <class P, class TY> S2D_DEV void m_dash(const P& p, const TY& pt, int l, MObj& o, float power, float dir, float& ax, float& ay) {
  power = clampf(power, p.min_dash_power, p.max_dash_power);
  dir = clampf(dir, p.min_dash_angle, p.max_dash_angle);
  if (p.dash_angle_step > 0.0f) dir = p.dash_angle_step * rintf(dir * p.inv_dash_angle_step);
  bool back = power < 0.0f;
  float need = back ? power * -2.0f : power;
  float avail = o.stamina + pt[PT_EXTRA_STAMINA][l];
  if (need > avail) need = avail;
  float st = o.stamina - need;
  o.stamina = st > 0.0f ? st : 0.0f;
  power = back ? need / -2.0f : need;
  float ad = fabsf(dir);
  float dir_rate = ad > 90.0f
      ? p.back_dash_rate - ((p.back_dash_rate - p.side_dash_rate) * (1.0f - (ad - 90.0f) * 0.011111111111111112f))
      : p.side_dash_rate + ((1.0f - p.side_dash_rate) * (1.0f - ad * 0.011111111111111112f));
  dir_rate = clampf(dir_rate, 0.0f, 1.0f);
  float acc = fabsf(o.effort * power * dir_rate * pt[PT_DASH_RATE][l]);
  if (back) dir += 180.0f;
  float sn, cs;
  sincos_deg(norm_deg_any(o.body + dir), sn, cs);
  ax = acc * cs; ay = acc * sn;
}
template <class P> S2D_DEV void m_turn(const P& p, float inertia_moment, MObj& o, float moment, float noise_u) {
  moment = clampf(moment, p.min_moment, p.max_moment);
  float speed = hypot2(o.vx, o.vy);
  float f = 1.0f;
  if (p.noise) f = 1.0f + (noise_u * 2.0f - 1.0f) * p.player_rand;
  o.body = norm_deg_any(o.body + f * moment / (1.0f + inertia_moment * speed));
}
template <class P, class TY> S2D_DEV bool m_kick(const P& p, const TY& pt, int l, const MObj& o, float bx, float by, float bvx, float bvy,
                    float power, float dir, float u_mag, float u_ang, float& kx, float& ky) {
  float dx = bx - o.x, dy = by - o.y;
  // dist <= kickable_area decided on the square: sqrt is correctly rounded and monotone, and the table holds
  // the largest float whose root does not exceed the kickable area (found on the host)
  float d2 = sq2(dx, dy);
  if (!(d2 <= pt[PT_KICKABLE_AREA2][l])) return false;
  float dist = sqrtf(d2);
  const float inv_margin = pt[PT_INV_KICK_MARGIN][l];
  power = clampf(power, p.min_power, p.max_power);
  dir = clampf(dir, -180.0f, 180.0f);
  float dir_diff = fabsf(norm_deg_any(atan2_deg(dy, dx) - o.body));
  float dist_ball = dist - pt[PT_SIZE][l] - p.ball_size;
  float eff = power * pt[PT_KICK_RATE][l] * (1.0f - 0.25f * (dir_diff * 0.005555555555555556f)
                                             - 0.25f * (dist_ball * inv_margin));
  float sn, cs;
  sincos_deg(norm_deg_any(o.body + dir), sn, cs);
  float ax = eff * cs, ay = eff * sn;
  if (p.noise) {
    float pos_rate = 0.5f + 0.25f * (dir_diff * 0.005555555555555556f + dist_ball * inv_margin);
    float speed_rate = 0.5f + 0.5f * (hypot2(bvx, bvy) * p.inv_speed_decay);
    float max_rand = pt[PT_KICK_RAND][l] * (power * p.inv_max_power) * (pos_rate + speed_rate);
    float mag = u_mag * max_rand;
    float s2, c2;
    sincos_deg(u_ang * 360.0f - 180.0f, s2, c2);
    ax += mag * c2; ay += mag * s2;
  }
  kx = ax; ky = ay;
  return true;
}
// A ball farther away than sqrt(tackle_dist^2 + tackle_width^2) has |x| > tackle_dist or |y| > tackle_width in
// the body frame, i.e. fail > 1 > u: tackle_reach2 is that bound with a 1 % margin (far more than the rounding
// of the rotation), so those lanes skip the rotation and -- see the caller -- the Philox draw.
template <class P> S2D_DEV bool m_tackle_in_reach(const P& p, const MObj& o, float bx, float by) {
  return sq2(bx - o.x, by - o.y) <= p.tackle_reach2;
}
template <class P> S2D_DEV bool m_tackle(const P& p, const MObj& o, float bx, float by, float dir, float u, bool foul, float& kx, float& ky) {
  float dx = bx - o.x, dy = by - o.y;
  float sn, cs;
  sincos_deg(o.body, sn, cs);
  float rx = dx * cs + dy * sn;
  float ry = dy * cs - dx * sn;
  float d = rx > 0.0f ? p.tackle_dist : p.tackle_back_dist;
  float tx = d > 0.0f ? fabsf(rx) / d : (rx == 0.0f ? 0.0f : 1.0e9f);
  float ty = fabsf(ry) / p.tackle_width;
  float tx2 = tx * tx, ty2 = ty * ty;
  float fail = tx2 * tx2 * tx2 + ty2 * ty2 * ty2;                        // tackle_exponent 6
  if (foul) { const float tx4 = tx2 * tx2, ty4 = ty2 * ty2; fail = tx4 * tx4 * tx2 + ty4 * ty4 * ty2; }   // Tackle.foul: foul_exponent 10
  if (!(u >= fail)) return false;
  dir = clampf(dir, -180.0f, 180.0f);
  float ang_ball = fabsf(norm_deg_any(atan2_deg(dy, dx) - o.body));
  float eff = (p.max_back_tackle_power + (p.max_tackle_power - p.max_back_tackle_power) * (1.0f - fabsf(dir) * 0.005555555555555556f))
              * p.tackle_power_rate * (1.0f - 0.5f * (ang_ball * 0.005555555555555556f));
  float s2, c2;
  sincos_deg(norm_deg_any(o.body + dir), s2, c2);
  kx = eff * c2; ky = eff * s2;
  return true;
}
// Player::goalieCatch: the ball must lie in the catch rectangle (catch_len long, catch_area_w wide) rooted
// at the goalie and turned to body + dir; u = uniform draw (used when catch_probability < 1)
template <class P> S2D_DEV bool m_catch(const P& p, float catch_len, const MObj& o, float bx, float by, float dir, float u) {
  dir = clampf(dir, p.min_catch_angle, p.max_catch_angle);
  float sn, cs;
  sincos_deg(norm_deg_any(o.body + dir), sn, cs);
  float dx = bx - o.x, dy = by - o.y;
  float rx = dx * cs + dy * sn, ry = dy * cs - dx * sn;
  if (!(rx >= 0.0f && rx <= catch_len && fabsf(ry) <= p.catch_half_w)) return false;
  return u < p.catch_probability;
}
template <class P, class TY> S2D_DEV void m_update_stamina(const P& p, const TY& pt, int l, MObj& e) {
  const float effort_min = pt[PT_EFFORT_MIN][l], effort_max = pt[PT_EFFORT_MAX][l];
  if (e.stamina <= p.recover_dec_thr_value) {
    if (e.recovery > p.recover_min) { float r = e.recovery - p.recover_dec; e.recovery = r > p.recover_min ? r : p.recover_min; }
  }
  if (e.stamina <= p.effort_dec_thr_value) {
    if (e.effort > effort_min) { float f = e.effort - p.effort_dec; e.effort = f > effort_min ? f : effort_min; }
  }
  if (e.stamina >= p.effort_inc_thr_value) {
    if (e.effort < effort_max) { float f = e.effort + p.effort_inc; e.effort = f < effort_max ? f : effort_max; }
  }
  float inc = e.recovery * pt[PT_STAMINA_INC][l];
  float room = p.stamina_max - e.stamina;
  if (inc > room) inc = room;
  if (p.stamina_capacity >= 0.0f) { if (inc > e.capacity) inc = e.capacity; }
  e.stamina += inc;
  if (e.stamina > p.stamina_max) e.stamina = p.stamina_max;
  if (p.stamina_capacity >= 0.0f) { float c = e.capacity - inc; e.capacity = c > 0.0f ? c : 0.0f; }
}
S2D_DEV void m_noise(float& vx, float& vy, float rnd, float u_mag, float u_ang) {
  float s = hypot2(vx, vy);
  float mag = u_mag * (rnd * s);
  float sn, cs;
  sincos_deg(u_ang * 360.0f - 180.0f, sn, cs);
  vx += mag * cs; vy += mag * sn;
}

// Event counters (stats[1..7]: goals left / right, matches finished, kicks + catches, tackles, offsides, ball-outs).  A lane marks
// the events of ITS cycle in one word (`ev`); at the end of the cycle the wave adds the bits up (ballot + popcount) into the
// workgroup's LDS counters -- seven per-lane counters carried through the loop cost seven of the kernel's 128 registers.
enum { EV_GOAL_L = 1 << 1, EV_GOAL_R = 1 << 2, EV_FINISHED = 1 << 3, EV_KICK = 1 << 4, EV_OFFSIDE = 1 << 6, EV_OUT = 1 << 7 };
struct MCounts { unsigned int* lds; bool valid; unsigned int tackles; };   // the workgroup's counters; this lane's match exists; tackles
// are the one frequent event (a quarter of the benchmark policy's commands): they keep a per-lane counter, flushed after the loop
S2D_DEV void m_count_events(const MCounts& c, int ev) {     // wave-uniform call
  const unsigned long long any = __ballot(c.valid && ev != 0);
  if (any == 0ull) return;
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    if (k == 5) continue;                                  // tackles: MCounts::tackles
    const unsigned long long mk = __ballot(c.valid && ((ev >> k) & 1));
    if (mk != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&c.lds[k], (unsigned int)__popcll(mk));
  }
}

// One cycle of the match held by this half-wave.  l = lane within the half, half = 0/1.
// cmd/a/b = this lane's command (players).  All 64 lanes execute every shuffle.
S2D_DEV void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Collision tile of one match: kTileSlots float4 (x, y, size, -).  Object l lives in slot l and again in slot l + 23;
// pad lanes (l > 22) put both of their writes behind the copies (slots 46 ..), so no two lanes ever write one slot.
constexpr int kTileSlots = 2 * kHalf;
S2D_DEV int tile_slot(int l) { return l <= BALL ? l : l + (BALL + 1); }
S2D_DEV void tile_put(float4* row, int l, float x, float y) {
  float2 v = make_float2(x, y);
  *reinterpret_cast<float2*>(row + tile_slot(l)) = v;
  *reinterpret_cast<float2*>(row + l + (BALL + 1)) = v;
}
S2D_DEV void tile_init(float4* row, int l, float size) {
  row[tile_slot(l)] = make_float4(0.0f, 0.0f, size, 0.0f);
  row[l + (BALL + 1)] = make_float4(0.0f, 0.0f, size, 0.0f);
}
template <class P, class TY> S2D_DEV void match_cycle(const P& p, const TY& pt, MObj& o, MGame& g, MRare& gr, int l, int half, uint32_t gl, uint32_t gh,
                         int cmd, float a, float bb, MCounts& cnt, float4* pos) {
  int ev = 0;                                              // this lane's events of this cycle (EV_*)
  const bool is_player = l < NP, is_ball = l == BALL;
  const uint32_t cyc = (uint32_t)g.tick;                   // Philox counter: cycles since the reset, stopped ones included
  const int mode0 = g.mode, side0 = g.mode_side;
  const float x0 = o.x, y0 = o.y;
  const float bx0 = hbcast_c<BALL>(o.x, half), by0 = hbcast_c<BALL>(o.y, half);
  float bvx0 = 0.0f, bvy0 = 0.0f;                         // the ball's velocity: read by the kick noise only
  if (p.noise) { bvx0 = hbcast_c<BALL>(o.vx, half); bvy0 = hbcast_c<BALL>(o.vy, half); }
  g.reward = 0.0f; g.done = 0;

  // ---- 1. commands + player movement (lane-local)
  float ax = 0.0f, ay = 0.0f, kx = 0.0f, ky = 0.0f;
  bool kicked = false, by_kick = false;                    // by_kick: the impulse came from a Kick command (not a tackle)
  if (!is_player || o.tackle > 0 || is_halted(mode0) || o.card >= S2D_CARD_RED) cmd = S2D_MCMD_NONE;
  // the shoot-out: the taker acts once the kick is ready, the defending goalie once it is taken, nobody else at all
  const bool pen_goalie_acts = mode0 == S2D_GM_PENALTY_TAKEN && l == (side0 == SIDE_LEFT ? S2D_MATCH_GOALIE_RIGHT : S2D_MATCH_GOALIE_LEFT);
  if (is_penalty(mode0)) {                                 // (uniform per match)
    const bool taker_acts = l == (gr.taker & 0xff) - 1 && (mode0 == S2D_GM_PENALTY_READY || mode0 == S2D_GM_PENALTY_TAKEN);
    if (!(taker_acts || pen_goalie_acts)) cmd = S2D_MCMD_NONE;
  }
  // one noise block per object and cycle: x, y = movement noise; z, w = the command's own noise (a player sends ONE body
  // command per cycle: Turn uses z, Kick uses z and w)
  U4 nz{0, 0, 0, 0};
  if (p.noise) nz = m_draw(p, gl, gh, cyc, S2D_ST_NOISE, (uint32_t)l);
  const bool may_touch = !is_setplay(mode0) || (side_of(l) == side0 && !ball_dead(mode0)) || pen_goalie_acts;   // announcements, after a goal, before the kick-off: the ball is dead
  bool foul_try = false, foul_seen_l = false;              // this lane's intentional tackle succeeded; the referee would see a foul of his
  bool caught = false, hold_moved = false;
  if (cmd == S2D_MCMD_DASH) m_dash(p, pt, l, o, a, bb, ax, ay);
  else if (cmd == S2D_MCMD_TURN) m_turn(p, pt[PT_INERTIA][l], o, a, rnd_u01(nz.z));
  else if (cmd == S2D_MCMD_CATCH) {
    // goalies only, play_on only, not while banned; every attempt starts the ban
    if ((l == S2D_MATCH_GOALIE_LEFT || l == S2D_MATCH_GOALIE_RIGHT) && (mode0 == S2D_GM_PLAY_ON || mode0 == S2D_GM_PENALTY_TAKEN) && o.catch_ban == 0) {
      float u = 0.0f;
      if (p.catch_probability < 1.0f) u = rnd_u01(m_draw(p, gl, gh, cyc, S2D_ST_CATCH, (uint32_t)l).x);
      o.catch_ban = p.catch_ban_cycle + 1;
      caught = m_catch(p, pt[PT_CATCH_LEN][l], o, bx0, by0, a, u);
    }
  } else if (cmd == S2D_MCMD_MOVE) {
    // Move(x, y) in the team's own frame (right team mirrored): before a kick-off anywhere in the own half;
    // while holding a caught ball, goalie_max_moves times inside the own penalty area
    const float sgn = side_of(l) == SIDE_LEFT ? 1.0f : -1.0f;
    const bool holds = mode0 == S2D_GM_FREE_KICK && gr.holder == l + 1 && gr.moves > 0;
    if (mode0 == S2D_GM_KICK_OFF || mode0 == S2D_GM_AFTER_GOAL || mode0 == S2D_GM_BEFORE_KICK_OFF || holds) {
      float tx = clampf(a, -p.half_l, holds ? -p.pen_x : 0.0f);
      float ty = holds ? clampf(bb, -p.pen_half_w, p.pen_half_w) : clampf(bb, -p.half_w, p.half_w);
      o.x = sgn * tx; o.y = sgn * ty; o.vx = 0.0f; o.vy = 0.0f;
      hold_moved = holds;
    }
  } else if (cmd == S2D_MCMD_KICK) {
    bool ok = m_kick(p, pt, l, o, bx0, by0, bvx0, bvy0, a, bb, rnd_u01(nz.z), rnd_u01(nz.w), kx, ky);
    if (ok && may_touch) { kicked = true; by_kick = true; ev |= EV_KICK; } else { kx = 0.0f; ky = 0.0f; }
  } else if (cmd == S2D_MCMD_TACKLE) {
    bool ok = false;
    const bool foul = bb != 0.0f;                          // Tackle.foul
    if (m_tackle_in_reach(p, o, bx0, by0)) {               // rare: most tackles are nowhere near the ball
      U4 w = m_draw(p, gl, gh, cyc, S2D_ST_TACKLE, (uint32_t)l);
      ok = m_tackle(p, o, bx0, by0, a, rnd_u01(w.x), foul, kx, ky);
      foul_try = ok && foul && mode0 == S2D_GM_PLAY_ON;
      foul_seen_l = rnd_u01(w.y) < p.foul_detect_probability;
    }
    o.tackle = p.tackle_cycles + 1;
    cnt.tackles += 1u;
    if (ok && may_touch) kicked = true; else { kx = 0.0f; ky = 0.0f; }
  }
  if (is_player) {
    if (cmd == S2D_MCMD_DASH) {
      float a2 = sq2(ax, ay);
      if (a2 > p.player_accel_max2) { float k = p.player_accel_max / sqrtf(a2); ax *= k; ay *= k; }
      o.vx += ax; o.vy += ay;
    }
    float s2 = sq2(o.vx, o.vy);
    if (s2 > pt[PT_SPEED_MAX2][l]) { float k = pt[PT_SPEED_MAX][l] / sqrtf(s2); o.vx *= k; o.vy *= k; }
    if (p.noise) m_noise(o.vx, o.vy, p.player_rand, rnd_u01(nz.x), rnd_u01(nz.y));
    o.x += o.vx; o.y += o.vy;
  }
  // Three copies of the rest of the cycle.  In most cycles nothing was caught, moved by hand, fouled or kicked (CALM), and mostly the
  // mode is play_on with no offside flag up as well (PLAY); the copies for those cycles have these facts as constants, so the event
  // blocks below and every later test on them fold away instead of being evaluated and skipped.  CALM without PLAY is the copy of
  // the WAITING matches (a set play not yet taken, a dead ball's countdown): their waves are the ones a launch waits for, so their
  // cycle has to be as short as a quiet one.  A wave takes the most general copy one of its two matches needs.
  const int mode_start = mode0;
  const bool cmd_events = __ballot(caught || hold_moved || foul_try || kicked) != 0ull;
  const bool mode_events = __ballot(mode0 != S2D_GM_PLAY_ON || g.offside != 0) != 0ull;
  auto rest_of_cycle = [&](auto calm_tag, auto play_tag) {
  constexpr bool CALM = decltype(calm_tag)::value, PLAY = decltype(play_tag)::value;
  const int mode0 = PLAY ? (int)S2D_GM_PLAY_ON : mode_start;
  if (PLAY) g.offside = 0;                                 // (it is)
  // a successful catch wins the cycle: every kick / tackle impulse of this cycle is dropped
  // ... and so does a move of the goalie who holds the ball (the holder is a goalie: bits 0 / 11)
  int caught_by = -1, hold_move = -1;
  if (!CALM && __ballot(caught || hold_moved) != 0ull) {   // wave-uniform, rare
    const uint32_t goalies = (1u << S2D_MATCH_GOALIE_LEFT) | (1u << S2D_MATCH_GOALIE_RIGHT);
    const uint32_t cmask = hballot(caught, half) & goalies, hmask = hballot(hold_moved, half) & goalies;
    caught_by = cmask ? __ffs((int)cmask) - 1 : -1;
    hold_move = hmask ? __ffs((int)hmask) - 1 : -1;
    if (caught_by >= 0 && l == caught_by) ev |= EV_KICK;
    if (caught_by >= 0 || hold_move >= 0) { kicked = false; by_kick = false; kx = 0.0f; ky = 0.0f; }
  }
  // FoulCharge_ (idl/service.proto:282): a successful INTENTIONAL tackle through an opponent who has the ball (kickable) inside the
  // tackler's tackle area brings him down for foul_cycles; the referee sees it with foul_detect_probability.  Positions of the
  // start of the cycle; the first such tackler (lowest index) counts.  Wave-uniform and rare: only tackles with foul set get here.
  int foul_call = 0;                                       // 1 + tackler if the referee saw a foul, else 0
  if (!CALM && __ballot(foul_try) != 0ull) {
    const bool has_ball = is_player && o.card < S2D_CARD_RED && sq2(bx0 - x0, by0 - y0) <= pt[PT_KICKABLE_AREA2][l];
    const uint32_t hb = hballot(has_ball, half) & 0x3FFFFFu;
    float sn, cs;
    sincos_deg(o.body, sn, cs);
    int victim = -1;
    const int o0 = side_of(l) == SIDE_LEFT ? 11 : 0;
    for (int j = 0; j < 11; ++j) {
      const int jj = o0 + j;
      const float dx = hbcast(x0, jj) - x0, dy = hbcast(y0, jj) - y0;
      const float rx = dx * cs + dy * sn, ry = dy * cs - dx * sn;
      if (victim < 0 && ((hb >> jj) & 1u) && rx >= 0.0f && rx <= p.tackle_dist && fabsf(ry) <= p.tackle_width) victim = jj;
    }
    const uint32_t fm = hballot(foul_try && victim >= 0, half) & 0x3FFFFFu;
    const int foul_by = fm ? __ffs((int)fm) - 1 : -1;
    const int src = foul_by >= 0 ? foul_by : 0;
    const int foul_victim = hbcasti(victim, src);
    const bool foul_seen = hbcasti(foul_seen_l ? 1 : 0, src) != 0 && foul_by >= 0;
    if (foul_by >= 0) {                                    // the victim goes down; a foul the referee saw is a card
      if (l == foul_victim && o.tackle < p.foul_cycles + 1) o.tackle = p.foul_cycles + 1;
      if (l == foul_by && foul_seen && o.card < S2D_CARD_RED) o.card += 1;
    }
    foul_call = foul_seen ? foul_by + 1 : 0;
  }
  // ---- 2. ball: impulses summed in player order
  float bax = 0.0f, bay = 0.0f;
  bool any_kick = false, fk_fault = false;
  int last_kicker = -1, taker0 = 0;
  const bool wave_kick = !CALM && __ballot(kicked) != 0ull; // wave-uniform: kicks are rare events
  if (wave_kick) {
    const uint32_t kmask = hballot(kicked, half) & 0x3FFFFFu;
    any_kick = kmask != 0u;
    last_kicker = any_kick ? 31 - __clz(kmask) : -1;
    // the impulses in player order, kickers only: the union of the two matches' kicker masks is wave-uniform, so each kicker's
    // (kx, ky) comes through v_readlane (the lane number in an SGPR) instead of 44 ds_bpermute per cycle with a kick in it
    const unsigned long long kfull = __ballot(kicked);
    uint32_t um = ((uint32_t)kfull | (uint32_t)(kfull >> kHalf)) & 0x3FFFFFu;
    while (um != 0u) {
      const int j = __ffs((int)um) - 1;
      um &= um - 1u;
      const int xl = __builtin_amdgcn_readlane(__float_as_int(kx), j), xh = __builtin_amdgcn_readlane(__float_as_int(kx), j + kHalf);
      const int yl = __builtin_amdgcn_readlane(__float_as_int(ky), j), yh = __builtin_amdgcn_readlane(__float_as_int(ky), j + kHalf);
      const float kxj = __int_as_float(half ? xh : xl), kyj = __int_as_float(half ? yh : yl);
      if ((kmask >> j) & 1u) { bax += kxj; bay += kyj; }
    }
    if (any_kick) g.last_touch = side_of(last_kicker);
    // free-kick fault / back-pass bookkeeping (oracle: match_step, same decisions from the same masks)
    taker0 = gr.taker & 0xff;                                                    // (bit 8: his set play was an INDIRECT free kick)
    const uint32_t cmask2 = hballot(by_kick, half) & 0x3FFFFFu;                  // Kick-command kickers
    const uint32_t taker_bit = taker0 > 0 ? (1u << (taker0 - 1)) : 0u;
    const bool other_touch = (kmask & ~taker_bit) != 0u;
    fk_fault = p.free_kick_faults && mode0 == S2D_GM_PLAY_ON && taker0 != 0 && any_kick && !other_touch;
    if (any_kick) {
      if (is_penalty(mode0)) { /* the word carries the shoot-out's state */ }
      else if (is_setplay(mode0)) gr.taker = (last_kicker + 1) | (mode0 == S2D_GM_IND_FREE_KICK ? 0x100 : 0);   // this kick puts the ball into play
      else if (other_touch) gr.taker = 0;
      const int last_kick_cmd = cmask2 ? 31 - __clz(cmask2) : -1;
      gr.last_kicker = (last_kick_cmd == last_kicker) ? last_kick_cmd + 1 : 0;
    }
  }
  const bool ball_live = !is_setplay(mode0) || any_kick || mode0 == S2D_GM_PENALTY_TAKEN;
  if (caught_by >= 0) {                                   // held: the ball rests where it was caught
    g.last_touch = side_of(caught_by);
    if (is_ball) { o.vx = 0.0f; o.vy = 0.0f; }
  } else if (hold_move >= 0) {                            // the holding goalie moved: the ball goes with him, in front of his body
    const float gx = hbcast(o.x, hold_move), gy = hbcast(o.y, hold_move), gb = hbcast(o.body, hold_move);
    const float r = pt[PT_SIZE][hold_move] + p.ball_size + 0.1f;   // clear of the collision radius, inside the kickable area
    float sn, cs;
    sincos_deg(gb, sn, cs);
    if (is_ball) { o.x = gx + r * cs; o.y = gy + r * sn; o.vx = 0.0f; o.vy = 0.0f; }
    gr.moves -= 1;
  } else if (is_ball && ball_live) {
    if (any_kick) {
      float a2 = sq2(bax, bay);
      if (a2 > p.ball_accel_max2) { float k = p.ball_accel_max / sqrtf(a2); bax *= k; bay *= k; }
      o.vx += bax; o.vy += bay;
    }
    float s2 = sq2(o.vx, o.vy);
    if (s2 > p.ball_speed_max2) { float k = p.ball_speed_max / sqrtf(s2); o.vx *= k; o.vy *= k; }
    if (p.noise) m_noise(o.vx, o.vy, p.ball_rand, rnd_u01(nz.x), rnd_u01(nz.y));   // block 22 = the ball lane's own draw
    o.x += o.vx; o.y += o.vy;
  }
  // ---- 3. collisions (Jacobi passes; loop bounds are wave-uniform)
  bool collided = false;
  int touch_player = -1;
  const float ri = pt[PT_SIZE][l];
  // Overlaps are rare (two of 23 objects within ~0.6 m), so the 23-step scan is preceded by an exact
  // detection pass that costs half of it: 23 is odd, so "lane l against objects l+1 .. l+11 (mod 23)"
  // visits every unordered pair exactly once.  Same comparison (d2 < (ri + rj)^2, both symmetric in the
  // pair), so a wave skips the scan only when the scan would have found nothing.
  // The tile row of a match holds (x, y, size) of its 23 objects TWICE (slots l and l + 23, tile_put), so "object
  // l + m" is slot l + m without a wrap and every read of the unrolled loop is one ds_read_b128 at a constant offset
  // from the lane's own slot; the sizes are written once per launch (tile_init).  Pad lanes (23 .. 31) compare
  // whatever their slots hold and are masked out at the end.
  bool overlap = false;
  // the ball after its move (nothing but a collision moves it again before the referee looks)
  float bx = hbcast_c<BALL>(o.x, half), by = hbcast_c<BALL>(o.y, half);
  if constexpr (TypesAreConst<TY>::value) {
    // One player size: every player-player pair has the same r^2, so the ring keeps the smallest d2 and compares once; the ball
    // leaves the ring (its slot holds a far-away point for this pass -- the scan below puts the real one back) and every player
    // tests it directly with the other radius.  Exactly the pairs and comparisons of the general form, ~30 instructions fewer.
    constexpr float rr_pp = (MStockTypes::size + MStockTypes::size) * (MStockTypes::size + MStockTypes::size);
    constexpr float r_pb = MStockTypes::size + MStock::ball_size, rr_pb = r_pb * r_pb;
    tile_put(pos, l, is_ball ? 1.0e9f : o.x, is_ball ? 1.0e9f : o.y);
    wave_fence();
    const float4* mine = pos + l;
    float nearest2 = 1.0e30f;
#pragma unroll
    for (int m = 1; m <= 11; ++m) {
      const float4 pj = mine[m];
      nearest2 = fminf(nearest2, sq2(o.x - pj.x, o.y - pj.y));
    }
    overlap = is_player && (nearest2 < rr_pp || sq2(o.x - bx, o.y - by) < rr_pb);
  } else {
    tile_put(pos, l, o.x, o.y);
    wave_fence();
    const float4* mine = pos + l;
    // d2 < r^2 for some partner  <=>  min over the partners of (d2 - r^2) < 0 (the sign of a float difference is exact): one running
    // minimum instead of eleven compare results gathered bit by bit
    float worst = 1.0f;
#pragma unroll
    for (int m = 1; m <= 11; ++m) {
      const float4 pj = mine[m];
      float dx = o.x - pj.x, dy = o.y - pj.y;
      float r = ri + pj.z;
      worst = fminf(worst, sq2(dx, dy) - r * r);
    }
    overlap = worst < 0.0f && l <= BALL;
  }
  wave_fence();
  const bool wave_overlap = __ballot(overlap) != 0ull;     // wave-uniform: everything about collisions hangs on it
  for (int pass = 0; pass < 10 && wave_overlap; ++pass) {
    float sx = 0.0f, sy = 0.0f; int c = 0; int tp = -1;
    tile_put(pos, l, o.x, o.y);                           // wave-private tile: LDS ops of a wave are in order
    wave_fence();
    for (int j = 0; j <= BALL; ++j) {
      const float4 pj = pos[j];                           // same address for the whole half: broadcast read
      float xj = pj.x, yj = pj.y;
      float rj = pj.z;
      float dx = o.x - xj, dy = o.y - yj;
      float d2 = sq2(dx, dy), r = ri + rj;
      if (l <= BALL && j != l && d2 < r * r) {
        float d = sqrtf(d2), ux, uy;
        if (d > 0.0f) { ux = dx / d; uy = dy / d; } else { ux = l < j ? -1.0f : 1.0f; uy = 0.0f; }
        float mx = (o.x + xj) * 0.5f, my = (o.y + yj) * 0.5f, h = r * 0.5f;
        sx += mx + ux * h; sy += my + uy * h; c++;
        tp = j;
      }
    }
    wave_fence();
    if (__ballot(c > 0) == 0ull) break;
    bool moved = false;
    if (c > 0) {
      const float nx = sx / (float)c, ny = sy / (float)c;
      moved = nx != o.x || ny != o.y;
      o.x = nx; o.y = ny; collided = true; if (is_ball) touch_player = tp;
    }
    // A pass is a function of the positions alone: when it moved nobody (two players left exactly in contact, neither of them moving --
    // rounding calls that an overlap again in every pass and every cycle), the remaining passes would repeat it word for word.  Such a
    // pair cost its wave all ten passes per cycle for as long as both stood still: one match of 8 192 made a 64-cycle launch last
    // 960 us instead of 270 (profiles/r04/match_slow_launch_bisect.txt).
    if (__ballot(moved) == 0ull) break;
  }
  int coll_touch_side = SIDE_NONE;
  if (wave_overlap) {
    if (collided) { o.vx *= p.collision_vel_rate; o.vy *= p.collision_vel_rate; }
    touch_player = hbcasti_c<BALL>(touch_player, half);
    if (touch_player >= 0 && (!is_setplay(mode0) || side_of(touch_player) == side0)) {
      coll_touch_side = side_of(touch_player);
      g.last_touch = coll_touch_side;
      if (!is_penalty(mode0) && touch_player + 1 != (gr.taker & 0xff)) gr.taker = 0;
      if (touch_player + 1 != gr.last_kicker) gr.last_kicker = 0;
    }
  }
  if (wave_overlap) { bx = hbcast_c<BALL>(o.x, half); by = hbcast_c<BALL>(o.y, half); }   // ... which may just have happened
  // ---- 4. set play: opponents keep their distance
  if (__ballot(mode0 != S2D_GM_PLAY_ON) != 0ull) {         // wave-uniform
    const float bxn = bx, byn = by;
    // the side that does not take the set play keeps its distance; during an announcement that is the offending side (side0)
    const int kept_away = is_announcement(mode0) ? side0 : other_side(side0);
    if (is_setplay(mode0) && mode0 != S2D_GM_AFTER_GOAL && !is_period_end(mode0) && !is_penalty(mode0) && is_player &&
        side_of(l) == kept_away && o.card < S2D_CARD_RED) {
      float dx = o.x - bxn, dy = o.y - byn, d = hypot2(dx, dy);
      if (d < p.free_kick_distance) {
        float ux, uy;
        if (d > 0.0f) { ux = dx / d; uy = dy / d; } else { ux = side_of(l) == SIDE_LEFT ? -1.0f : 1.0f; uy = 0.0f; }
        o.x = bxn + ux * p.free_kick_distance; o.y = byn + uy * p.free_kick_distance;
      }
    }
  }
  // ---- 5. referee (every lane of the half evaluates the same decisions).  The clock: WorldModel.cycle advances unless the mode
  // of this cycle is one in which it stands still
  g.tick = (int)((uint32_t)g.tick + 1u);
  const bool advanced = !(p.stopped_clock && clock_stands(mode0));
  if (advanced) { g.cycle = (int)((uint32_t)g.cycle + 1u); gr.stopped = 0; }   // wrap-defined
  else gr.stopped += 1;
  // Most cycles are quiet: play_on, nobody touched the ball, the ball is on the pitch, no offside flag is up and the clock is not at
  // a half's end.  In such a cycle every decision below comes out "nothing happens", so a wave whose two matches are both quiet
  // skips them (some forty branches that each fall through); the test is the union of the conditions those decisions read.
  // The same holds for a match that WAITS: a set play nobody has taken yet (up to drop_ball_time cycles), a dead ball whose wait
  // is not over (after a goal, an announcement, before a kick-off), a finished match without auto-restart.  Its timer counts and
  // nothing else is due -- and it waits for tens of cycles on end, so a wave holding such a match would take the referee's full
  // pass in every cycle of a launch and finish long after the others (the launch lasts as long as its slowest wave: 305 us against
  // 215 us for a batch in which nothing happens, profiles/r03/match_quiet_rate.txt, with 95 % of all wave-cycles quiet).
  const int total_cycles = p.total_cycles, end_cycles = p.end_cycles;      // end of the normal time, end of the last period
  // IllegalDefense_ (idl/service.proto:295, 1637-1640; off when number = 0 -- the stock server, and then all of this folds away): the
  // counters of a cycle played in PlayOn (oracle: match_step).  They live in the set-play timer, which PlayOn does not use.
  int ill_timer = 0; bool ill_call = false;
  if (p.illegal_defense_number > 0) {
    const bool in_strip = is_player && o.card < S2D_CARD_RED && fabsf(o.y) < p.ill_half_w && (l < 11 ? o.x < -p.ill_x : o.x > p.ill_x);
    const uint32_t zm = hballot(in_strip, half) & 0x3FFFFFu;
    const int nl = __popc(zm & 0x7FFu), nr = __popc(zm >> 11);
    int cl = g.timer & 0xff, cr = (g.timer >> 8) & 0xff;
    cl = (g.last_touch == SIDE_RIGHT && nl >= p.illegal_defense_number) ? (cl < 255 ? cl + 1 : 255) : 0;
    cr = (g.last_touch == SIDE_LEFT && nr >= p.illegal_defense_number) ? (cr < 255 ? cr + 1 : 255) : 0;
    ill_timer = cl | (cr << 8);
    ill_call = mode0 == S2D_GM_PLAY_ON && (cl >= p.illegal_defense_duration || cr >= p.illegal_defense_duration);
  }
  const bool calm = !(any_kick || caught_by >= 0 || hold_move >= 0 || foul_call != 0) && g.offside == 0 && !ill_call &&
                    !(advanced && (g.cycle >= end_cycles || g.to_half == 1));
  bool idle = mode0 == S2D_GM_PLAY_ON && fabsf(bx) <= p.half_l && fabsf(by) <= p.half_w;   // play goes on, the ball is on the pitch
  if (!PLAY && mode0 != S2D_GM_PLAY_ON) {
    if (is_halted(mode0)) {
      idle = true;
    } else {                                               // the value the timer has to stay below after this cycle's increment
      const int pen_limit = mode0 == S2D_GM_PENALTY_READY ? p.pen_ready_wait : mode0 == S2D_GM_PENALTY_TAKEN ? p.pen_taken_wait + 1 :
                            mode0 == S2D_GM_PENALTY_SETUP ? 0 : p.pen_before_setup_wait;
      const int limit = is_penalty(mode0) ? pen_limit : !ball_dead(mode0) ? p.drop_ball_time + 1 : mode0 == S2D_GM_AFTER_GOAL ? p.after_goal_wait :
                        is_announcement(mode0) ? p.announce_wait : mode0 == S2D_GM_BEFORE_KICK_OFF ? p.kick_off_wait : 0;
      idle = g.timer + 1 < limit && (mode0 != S2D_GM_PENALTY_TAKEN || (fabsf(bx) <= p.half_l && fabsf(by) <= p.half_w));
    }
  }
  const bool busy = !(calm && idle);
  if (__ballot(busy) == 0ull) {
    if (advanced) g.to_half -= 1;                          // the clock moved, and not onto a half's end
    if (mode0 != S2D_GM_PLAY_ON && !is_halted(mode0)) g.timer += 1;
    if (p.illegal_defense_number > 0 && mode0 == S2D_GM_PLAY_ON) g.timer = ill_timer;
    if (!is_halted(mode0) && mode0 != S2D_GM_FREE_KICK) { gr.holder = 0; gr.moves = 0; }
  } else {
  float first = -1.0e9f, second = -1.0e9f;      // two largest dirS*x0 among the kicker's opponents
  const int S = any_kick ? side_of(last_kicker) : SIDE_LEFT;
  const float dirS = S == SIDE_LEFT ? 1.0f : -1.0f;
  if (wave_kick) {
    const int o0 = S == SIDE_LEFT ? 11 : 0;
    for (int j = 0; j < 11; ++j) {
      float v = dirS * hbcast(x0, o0 + j);
      if (v > first) { second = first; first = v; } else if (v > second) second = v;
    }
  }
  bool restart_form = false; int form_side = SIDE_LEFT;      // formation placement requested
  bool place_ball = false; float pbx = 0.0f, pby = 0.0f;     // ball placement requested
  bool recover_half = false;
  if (!is_halted(mode0)) {
    if (mode0 == S2D_GM_AFTER_GOAL) {                      // dead ball until the wait is over, then the conceding side kicks off
      g.timer += 1;
      if (g.timer >= p.after_goal_wait) {
        const int ks = other_side(side0);
        restart_form = true; form_side = ks;
        g.mode = S2D_GM_KICK_OFF; g.mode_side = ks; g.timer = 0; g.offside = 0; g.last_touch = SIDE_NONE;
        gr.taker = 0; gr.last_kicker = 0;
      }
    } else if (mode0 == S2D_GM_BEFORE_KICK_OFF) {          // nobody plays the ball, players may Move
      g.timer += 1;
      if (g.timer >= p.kick_off_wait) { g.mode = S2D_GM_KICK_OFF; g.timer = 0; }
    } else if (is_period_end(mode0)) {                     // one cycle of "half time" / "time extended", then the next kick-off
      g.mode = p.kick_off_wait > 0 ? S2D_GM_BEFORE_KICK_OFF : S2D_GM_KICK_OFF; g.timer = 0;
    } else if (mode0 == S2D_GM_GOALIE_CATCH) {             // one cycle of "goalie_catch_ball", then his free kick
      g.mode = S2D_GM_FREE_KICK; g.timer = 0;
    } else if (is_announcement(mode0)) {                   // offside_l, back_pass_l, ...: after the wait, the restart for the other side
      g.timer += 1;
      if (g.timer >= p.announce_wait) {
        g.mode = (mode0 == S2D_GM_BACK_PASS || mode0 == S2D_GM_FREE_KICK_FAULT) ? S2D_GM_IND_FREE_KICK : S2D_GM_FREE_KICK;
        g.mode_side = other_side(side0); g.timer = 0;
        // PenaltyKick_ (idl/service.proto:278): the foul was called inside the offender's own penalty area -- the other side restarts
        // from the penalty spot of that half (11 m from the goal line: a constant of the pitch, like rcssserver's) instead of the foul's
        const bool own_area = fabsf(by) <= p.pen_half_w && (side0 == SIDE_LEFT ? bx <= -p.pen_x : bx >= p.pen_x);
        if (mode0 == S2D_GM_FOUL_CHARGE && own_area) {
          g.mode = S2D_GM_PENALTY_KICK; g.offside = 0;
          place_ball = true; pbx = (side0 == SIDE_LEFT ? -1.0f : 1.0f) * (p.half_l - 11.0f); pby = 0.0f;
        }
      }
    } else if (is_penalty(mode0)) {                        // the shoot-out's own sequence (oracle: match_step, same order)
      int result = -1;                                     // 0 / 1 / 2: this cycle ends the kick with a miss / a goal / a foul of the kicker (a miss)
      if (mode0 == S2D_GM_PENALTY_ONFIELD) {
        g.timer += 1;
        if (g.timer >= p.pen_before_setup_wait) { g.mode = S2D_GM_PENALTY_SETUP; g.mode_side = SIDE_LEFT; }   // the left team kicks first (placed below)
      } else if (mode0 == S2D_GM_PENALTY_SETUP) {          // one cycle: everybody was placed on entering it
        g.mode = S2D_GM_PENALTY_READY; g.timer = 0;
      } else if (mode0 == S2D_GM_PENALTY_READY) {
        if (any_kick) { g.mode = S2D_GM_PENALTY_TAKEN; g.timer = 0; }
        else { g.timer += 1; if (g.timer >= p.pen_ready_wait) result = 0; }
      } else if (mode0 == S2D_GM_PENALTY_TAKEN) {
        if (caught_by >= 0) result = 0;
        else if (!p.pen_allow_mult_kicks && ((hballot(kicked, half) >> ((gr.taker & 0xff) - 1)) & 1u)) result = 2;   // PenaltyFoul_: a second touch
        else if (bx > p.half_l && fabsf(by) < p.goal_half_width) result = 1;
        else if (fabsf(bx) > p.half_l || fabsf(by) > p.half_w) result = 0;
        else { g.timer += 1; if (g.timer > p.pen_taken_wait) result = 0; }
      } else {                                             // PenaltyScore_ / PenaltyMiss_ / PenaltyFoul_: the verdict stands for a while
        g.timer += 1;
        if (g.timer >= p.pen_before_setup_wait) {
          if (pen_over(p, gr.taker)) {
            // pen_random_winner: a level shoot-out is decided by a coin (oracle: same draw, TACKLE stream, the ball's block)
            if (p.pen_random_winner && pen_goals(gr.taker, SIDE_LEFT) == pen_goals(gr.taker, SIDE_RIGHT))
              gr.taker |= (rnd_u01(m_draw(p, gl, gh, cyc, S2D_ST_TACKLE, (uint32_t)BALL).x) < 0.5f ? 1 : 2) << 28;
            g.mode = S2D_GM_TIME_OVER; g.mode_side = SIDE_NONE; g.done = 1; if (is_ball) ev |= EV_FINISHED;
          } else { g.mode = S2D_GM_PENALTY_SETUP; g.mode_side = other_side(side0); }
        }
      }
      if (result >= 0) {                                   // counted, announced; the ball is dead
        int w = gr.taker + (1 << (side0 == SIDE_LEFT ? 12 : 16));
        if (result == 1) { w += 1 << (side0 == SIDE_LEFT ? 20 : 24); g.reward = side0 == SIDE_LEFT ? 1.0f : -1.0f; }
        gr.taker = w;
        g.mode = result == 1 ? S2D_GM_PENALTY_SCORE : result == 2 ? S2D_GM_PENALTY_FOUL : S2D_GM_PENALTY_MISS; g.timer = 0;
        if (is_ball) { o.vx = 0.0f; o.vy = 0.0f; }
      }
    } else if (is_setplay(mode0)) {
      if (any_kick) { g.mode = S2D_GM_PLAY_ON; g.timer = 0; }
      else { g.timer += 1; if (g.timer > p.drop_ball_time) { g.mode = S2D_GM_PLAY_ON; g.timer = 0; } }
    }
    // offside candidates: each lane tests itself, the mask is assembled by ballot
    float line = 0.0f;
    if (second > line) line = second;
    { float bl = dirS * bx0; if (bl > line) line = bl; }
    const bool cand = is_player && side_of(l) == S && l != last_kicker && dirS * x0 > line;
    const uint32_t cand_mask = hballot(cand, half) & 0x3FFFFFu;
    // flagged players near the ball (for the offside call), tested on the post-move positions
    const bool near_ball = is_player && sq2(o.x - bx, o.y - by) < p.offside_area2;
    const uint32_t near_mask = hballot(near_ball, half) & 0x3FFFFFu;
    if (g.mode == S2D_GM_PLAY_ON) {
      if (any_kick) {
        const bool exempt = mode0 == S2D_GM_KICK_IN || mode0 == S2D_GM_GOAL_KICK || mode0 == S2D_GM_CORNER_KICK;
        g.offside = (p.use_offside && !exempt) ? (int)cand_mask : 0;
      } else if (coll_touch_side != SIDE_NONE && g.offside) {
        int flagged_side = (g.offside & 0x7FF) ? SIDE_LEFT : SIDE_RIGHT;
        if (coll_touch_side != flagged_side) g.offside = 0;
      }
      if (caught_by >= 0) {                                // goalie holds the ball
        const int gs = side_of(caught_by);
        const bool in_area = fabsf(by) <= p.pen_half_w && (gs == SIDE_LEFT ? bx <= -p.pen_x : bx >= p.pen_x);
        // back pass: the goalie catches a ball a team-mate kicked to him -> indirect free kick for the other side from the
        // nearer front corner of the penalty area; otherwise, inside the own penalty area: free kick for the goalie's side;
        // outside: catch fault
        const int lk = gr.last_kicker - 1;
        const bool back_pass = p.back_passes && in_area && lk >= 0 && lk != caught_by && side_of(lk) == gs;
        place_ball = true; g.timer = 0; g.offside = 0; gr.taker = 0; gr.last_kicker = 0;
        if (back_pass) {                                    // back_pass_l / _r: named after the offending side
          pbx = gs == SIDE_LEFT ? -p.pen_x : p.pen_x; pby = by > 0.0f ? p.pen_half_w : -p.pen_half_w;
          g.mode = S2D_GM_BACK_PASS; g.mode_side = gs;
        } else {                                            // GoalieCatch_, then his free kick -- or CatchFault_ outside the area
          pbx = bx; pby = by;
          g.mode = in_area ? S2D_GM_GOALIE_CATCH : S2D_GM_CATCH_FAULT; g.mode_side = gs;
          if (in_area) { gr.holder = caught_by + 1; gr.moves = p.goalie_max_moves; }
        }
      } else if (foul_call != 0) {                          // the referee saw the foul: FoulCharge_ where the ball is
        place_ball = true; pbx = clampf(bx, -p.half_l, p.half_l); pby = clampf(by, -p.half_w, p.half_w);
        g.mode = S2D_GM_FOUL_CHARGE; g.mode_side = side_of(foul_call - 1); g.timer = 0; g.offside = 0;
      } else if (fk_fault) {                                // the taker touched the ball twice
        place_ball = true; pbx = clampf(bx, -p.half_l, p.half_l); pby = clampf(by, -p.half_w, p.half_w);
        g.mode = S2D_GM_FREE_KICK_FAULT; g.mode_side = side_of(taker0 - 1); g.timer = 0; g.offside = 0;
        gr.taker = 0; gr.last_kicker = 0;
      // (no goal directly from an indirect free kick: while nobody but its taker has touched the ball -- bit 8 of the taker word --, a ball
      // in the net is a ball over the goal line: a goal kick, by the branch below)
      } else if (!(gr.taker & 0x100) && bx > p.half_l && fabsf(by) < p.goal_half_width) {
        gr.score_l += 1; g.reward = 1.0f; if (is_ball) ev |= EV_GOAL_L;
        g.timer = 0; g.offside = 0; g.last_touch = SIDE_NONE;
        if (p.after_goal_wait > 0) { place_ball = true; pbx = bx; pby = by; g.mode = S2D_GM_AFTER_GOAL; g.mode_side = SIDE_LEFT; }
        else { restart_form = true; form_side = SIDE_RIGHT; g.mode = S2D_GM_KICK_OFF; g.mode_side = SIDE_RIGHT; }
      } else if (!(gr.taker & 0x100) && bx < -p.half_l && fabsf(by) < p.goal_half_width) {
        gr.score_r += 1; g.reward = -1.0f; if (is_ball) ev |= EV_GOAL_R;
        g.timer = 0; g.offside = 0; g.last_touch = SIDE_NONE;
        if (p.after_goal_wait > 0) { place_ball = true; pbx = bx; pby = by; g.mode = S2D_GM_AFTER_GOAL; g.mode_side = SIDE_RIGHT; }
        else { restart_form = true; form_side = SIDE_LEFT; g.mode = S2D_GM_KICK_OFF; g.mode_side = SIDE_LEFT; }
      } else if (fabsf(bx) > p.half_l || fabsf(by) > p.half_w) {
        if (is_ball) ev |= EV_OUT;
        int toucher = g.last_touch == SIDE_NONE ? SIDE_LEFT : g.last_touch;
        float sy = by < 0.0f ? -1.0f : 1.0f, sxn = bx < 0.0f ? -1.0f : 1.0f;
        place_ball = true; g.timer = 0; g.offside = 0;
        if (fabsf(bx) <= p.half_l) {
          g.mode = S2D_GM_KICK_IN; g.mode_side = other_side(toucher);
          pbx = clampf(bx, -p.half_l, p.half_l); pby = sy * p.half_w;
        } else {
          int defender = bx > 0.0f ? SIDE_RIGHT : SIDE_LEFT;
          if (toucher == defender) {
            g.mode = S2D_GM_CORNER_KICK; g.mode_side = other_side(defender);
            pbx = sxn * (p.half_l - 1.0f); pby = sy * (p.half_w - 1.0f);
          } else {
            g.mode = S2D_GM_GOAL_KICK; g.mode_side = defender;
            pbx = sxn * (p.half_l - 5.5f); pby = sy * 9.16f;
          }
        }
      } else if (g.offside) {
        uint32_t hit = (uint32_t)g.offside & near_mask;
        if (hit) {
          int t = __ffs((int)hit) - 1;            // first flagged player in index order
          if (is_ball) ev |= EV_OFFSIDE;
          place_ball = true; pbx = 0.0f; pby = 0.0f;   // coordinates fetched below (needs a shuffle)
          g.mode = S2D_GM_OFF_SIDE; g.mode_side = side_of(t); g.timer = 0;          // offside_l / _r: named after the offender
          g.offside = -1 - t;                       // marker: ball goes to player t (resolved below)
        }
      }
    }
    if (p.illegal_defense_number > 0 && mode0 == S2D_GM_PLAY_ON && g.mode == S2D_GM_PLAY_ON) {   // no other call in this cycle
      g.timer = ill_timer;
      if (ill_call) {                                      // named after the offender; the ball on that half's penalty spot
        const int offender = (ill_timer & 0xff) >= p.illegal_defense_duration ? SIDE_LEFT : SIDE_RIGHT;
        g.mode = S2D_GM_ILLEGAL_DEFENSE; g.mode_side = offender; g.timer = 0; g.offside = 0;
        place_ball = true; pbx = (offender == SIDE_LEFT ? -1.0f : 1.0f) * (p.half_l - 11.0f); pby = 0.0f;
      }
    }
    // half time / extra time / time over (ServerParam.nr_extra_halfs, extra_half_time, golden_goal: idl/service.proto:1601, 1622, 1635):
    // decided when the clock has just moved.  A draw at the end of the normal time is extended by nr_extra_halfs halves of
    // extra_half_cycles, played in full unless golden_goal; the periods alternate the kick-off side.  (No penalty shoot-out.)
    bool at_half = false;                                  // the clock has just reached the end of a period
    if (advanced) { const int left = g.to_half - 1; at_half = left == 0; g.to_half = left; }
    const bool extra = p.nr_extra_halfs > 0;
    bool over = advanced && g.cycle >= end_cycles;
    if (extra && advanced && g.cycle == total_cycles) over = gr.score_l != gr.score_r;      // a draw is extended
    if (extra && p.golden_goal && g.cycle > total_cycles && g.reward != 0.0f) over = true;   // a goal (this cycle) in extra time
    if (is_penalty(mode0)) { over = false; at_half = false; }   // (the shoot-out ends by its own count; its clock stands)
    if (kShootOut && over && advanced && g.cycle == end_cycles && p.penalty_shoot_outs && gr.score_l == gr.score_r) {
      // a draw after the last period: PenaltyOnfield_, named after the half the kicks are taken in (the right one)
      g.mode = S2D_GM_PENALTY_ONFIELD; g.mode_side = SIDE_RIGHT; g.timer = 0; g.offside = 0; g.last_touch = SIDE_NONE;
      gr.taker = 0; gr.last_kicker = 0;
      if (is_ball) { o.vx = 0.0f; o.vy = 0.0f; }
    } else if (over) {
      g.mode = S2D_GM_TIME_OVER; g.mode_side = SIDE_NONE; g.done = 1; if (g.offside > 0) g.offside = 0; if (is_ball) ev |= EV_FINISHED;
    } else if (at_half) {
      const bool in_extra = extra && g.cycle >= total_cycles;
      const int k = in_extra ? p.nr_normal_halfs + (g.cycle - total_cycles) / p.extra_half_cycles : g.cycle / p.half_time_cycles;
      const int ks = (k & 1) ? SIDE_RIGHT : SIDE_LEFT;
      recover_half = true; restart_form = true; form_side = ks; place_ball = false;
      g.mode = g.cycle == total_cycles ? S2D_GM_EXTEND_HALF : S2D_GM_FIRST_HALF_OVER;
      g.mode_side = ks; g.timer = 0; g.offside = 0; g.last_touch = SIDE_NONE;
    }
    if (at_half) g.to_half = (extra && g.cycle >= total_cycles) ? p.extra_half_cycles : p.half_time_cycles;
    if (place_ball || restart_form) { gr.taker = 0; gr.last_kicker = 0; }   // every restart ends the double-touch / back-pass bookkeeping
    if (g.mode != S2D_GM_FREE_KICK && g.mode != S2D_GM_GOALIE_CATCH) { gr.holder = 0; gr.moves = 0; }   // nobody holds the ball any more
  }
  // resolve the offside spot (uniform shuffle, then apply)
  {
    int t = g.offside < 0 ? -1 - g.offside : 0;
    float tx = hbcast(o.x, t), ty = hbcast(o.y, t);
    if (g.offside < 0) {
      if (place_ball) { pbx = tx; pby = ty; }
      g.offside = 0;
    }
  }
  if (recover_half && is_player) m_recover(p, pt[PT_EFFORT_MAX][l], o, false);
  if (g.mode == S2D_GM_PENALTY_SETUP && mode0 != S2D_GM_PENALTY_SETUP) {   // entering PenaltySetup_: the referee places everybody (oracle: pen_setup)
    const int pen_next = g.mode_side;
    const int w = gr.taker;
    const int taker = (pen_next == SIDE_LEFT ? 0 : 11) + 10 - pen_kicks(w, pen_next) % 11;
    const int goalie = pen_next == SIDE_LEFT ? S2D_MATCH_GOALIE_RIGHT : S2D_MATCH_GOALIE_LEFT;
    if (is_player && o.card < S2D_CARD_RED) {              // (sent off: stays parked; a kick that falls to him is missed)
      if (l == taker) { o.x = p.pen_spot_x - 0.7f; o.y = 0.0f; o.body = 0.0f; }
      else if (l == goalie) { o.x = p.half_l - 1.0f; o.y = 0.0f; o.body = 180.0f; }
      else { o.x = l < 11 ? -3.0f : 3.0f; o.y = -7.5f + 1.5f * (float)(l % 11); }
      o.vx = 0.0f; o.vy = 0.0f;
    }
    if (is_ball) { o.x = p.pen_spot_x; o.y = 0.0f; o.vx = 0.0f; o.vy = 0.0f; }
    g.timer = 0; g.offside = 0;
    gr.last_kicker = 0; gr.taker = (w & ~0xff) | (taker + 1);
  }
  if (restart_form) m_place(o, l, form_side);
  else if (place_ball && is_ball) { o.x = pbx; o.y = pby; o.vx = 0.0f; o.vy = 0.0f; }
  }                                                        // busy
  // a second card is a red one: the player waits beside the halfway line, outside the pitch, one spot per uniform number
  if (is_player && o.card >= S2D_CARD_RED) {
    o.x = 0.0f; o.y = (side_of(l) == SIDE_LEFT ? -1.0f : 1.0f) * (p.half_w + 6.0f + 1.5f * (float)(l % 11));
    o.vx = 0.0f; o.vy = 0.0f;
  }
  };
  if (cmd_events) rest_of_cycle(std::false_type{}, std::false_type{});
  else if (mode_events) rest_of_cycle(std::true_type{}, std::false_type{});
  else rest_of_cycle(std::true_type{}, std::true_type{});
  // ---- 6. decay, tackle timers, stamina
  if (l <= BALL) { const float decay = pt[PT_DECAY][l]; o.vx *= decay; o.vy *= decay; }   // column 22 = ball_decay
  if (is_player) {
    if (o.tackle > 0) o.tackle -= 1;
    if (o.catch_ban > 0) o.catch_ban -= 1;
    m_update_stamina(p, pt, l, o);
  }
  m_count_events(cnt, ev);
  if (g.done && p.auto_reset) {
    int d = g.done; float rw = g.reward; int tk = g.tick;
    m_reset(p, pt[PT_EFFORT_MAX][l], o, g, gr, l);
    g.done = d; g.reward = rw; g.tick = tk;                // the draws of the next match continue the sequence
  }
}

// Nearest player to the ball per team (ties -> lowest index): butterfly min-reduction over the half-wave on the 64-bit key
// (bits(d2) << 8 | index); d2 >= 0, so its bit pattern orders like the value and the key orders like (d2, index) -- the same
// winner as a scan in index order.  Outputs only (nothing in the dynamics reads them): evaluated once per launch, after its
// last cycle.
S2D_DEV void match_nearest(const MObj& o, MGame& g, int l) {
  const bool is_player = l < NP;
  float bxn = hbcast(o.x, BALL), byn = hbcast(o.y, BALL);
  float d2 = sq2(o.x - bxn, o.y - byn);
  const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 8) | (unsigned long long)l;
  unsigned long long kl = (is_player && l < 11) ? key : ~0ull;
  unsigned long long kr = (is_player && l >= 11) ? key : ~0ull;
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    unsigned long long ol = __shfl_xor(kl, off, kHalf), orr = __shfl_xor(kr, off, kHalf);
    kl = ol < kl ? ol : kl; kr = orr < kr ? orr : kr;
  }
  g.nearest_l = (int)(kl & 0xFFull); g.nearest_r = (int)(kr & 0xFFull);
}

// benchmark policy: Philox POLICY stream, block = player, counter = cycle / 4 -- one call serves four cycles, one word each:
// command = the two top bits, magnitude = the 15 bits below them, direction = the low 15 bits (both exact in float: integers
// below 2^15 times a power of two).  `w` caches the block between the cycles of a fused rollout (`fresh` = draw it).
template <class P> S2D_DEV void m_random_action(const P& p, uint32_t gl, uint32_t gh, uint32_t cyc, int l, bool fresh, U4& w, int& cmd,
                             float& a, float& b) {
  if (fresh || (cyc & 3u) == 0u) w = m_draw(p, gl, gh, cyc >> 2, S2D_ST_POLICY, (uint32_t)l);
  const uint32_t lo = (cyc & 1u) ? w.y : w.x, hi = (cyc & 1u) ? w.w : w.z;
  const uint32_t wd = (cyc & 2u) ? hi : lo;
  cmd = 1 + (int)(wd >> 30);
  const float u = (float)((wd >> 15) & 0x7FFFu) * 3.0517578125e-05f;             // [0, 1)
  const float s = (float)(wd & 0x7FFFu) * 6.103515625e-05f - 1.0f;               // [-1, 1)
  // selects, not conditional stores through the references: those made {a, b} a stack array in scratch memory
  const bool two = cmd == S2D_MCMD_DASH || cmd == S2D_MCMD_KICK;
  const float ang = s * 180.0f;
  a = two ? u * 100.0f : ang;
  b = two ? ang : 0.0f;
}

// ------------------------------------------------------------------------------------------
// memory <-> registers
// ------------------------------------------------------------------------------------------
struct MPtrs { float* obj; int32_t* env; float* reward; uint8_t* done; unsigned long long* stats; int64_t obj_stride; int64_t env_stride;
               const float* ptab; /* [PT_WORDS][32] per-slot PlayerType parameters */ };

S2D_DEV void m_load(const MPtrs& q, int64_t e, int l, MObj& o, MGame& g, MRare& r) {
  o = MObj{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (l < SLOTS) {
    int64_t k = e * SLOTS + l;
    o.catch_ban = __float_as_int(q.obj[MF_CATCH_BAN * q.obj_stride + k]);
    o.card = __float_as_int(q.obj[MF_CARD * q.obj_stride + k]);
    o.x = q.obj[MF_X * q.obj_stride + k]; o.y = q.obj[MF_Y * q.obj_stride + k];
    o.vx = q.obj[MF_VX * q.obj_stride + k]; o.vy = q.obj[MF_VY * q.obj_stride + k];
    o.body = q.obj[MF_BODY * q.obj_stride + k];
    o.stamina = q.obj[MF_STAMINA * q.obj_stride + k]; o.effort = q.obj[MF_EFFORT * q.obj_stride + k];
    o.recovery = q.obj[MF_RECOVERY * q.obj_stride + k]; o.capacity = q.obj[MF_CAPACITY * q.obj_stride + k];
    o.tackle = __float_as_int(q.obj[MF_TACKLE * q.obj_stride + k]);
  }
  g.cycle = q.env[ME_CYCLE * q.env_stride + e]; g.mode = q.env[ME_MODE * q.env_stride + e];
  g.mode_side = q.env[ME_MODE_SIDE * q.env_stride + e]; r.score_l = q.env[ME_SCORE_L * q.env_stride + e];
  r.score_r = q.env[ME_SCORE_R * q.env_stride + e]; g.last_touch = q.env[ME_LAST_TOUCH * q.env_stride + e];
  g.timer = q.env[ME_TIMER * q.env_stride + e]; g.offside = q.env[ME_OFFSIDE * q.env_stride + e];
  g.reward = 0.0f; g.done = 0; g.nearest_l = 0; g.nearest_r = 0;        // outputs: match_nearest() after the launch's last cycle
  r.holder = q.env[ME_HOLDER * q.env_stride + e]; r.moves = q.env[ME_MOVES * q.env_stride + e];
  r.taker = q.env[ME_TAKER * q.env_stride + e]; r.last_kicker = q.env[ME_LAST_KICKER * q.env_stride + e];
  r.stopped = q.env[ME_STOPPED * q.env_stride + e]; g.tick = q.env[ME_TICK * q.env_stride + e];
}
template <class P> S2D_DEV void m_derive(const P& p, MGame& g) { g.to_half = cycles_to_period_end(p, g.cycle); }
S2D_DEV void m_store(const MPtrs& q, int64_t e, int l, const MObj& o, const MGame& g, const MRare& r) {
  if (l < SLOTS) {
    int64_t k = e * SLOTS + l;
    q.obj[MF_X * q.obj_stride + k] = o.x; q.obj[MF_Y * q.obj_stride + k] = o.y;
    q.obj[MF_VX * q.obj_stride + k] = o.vx; q.obj[MF_VY * q.obj_stride + k] = o.vy;
    q.obj[MF_BODY * q.obj_stride + k] = o.body;
    q.obj[MF_STAMINA * q.obj_stride + k] = o.stamina; q.obj[MF_EFFORT * q.obj_stride + k] = o.effort;
    q.obj[MF_RECOVERY * q.obj_stride + k] = o.recovery; q.obj[MF_CAPACITY * q.obj_stride + k] = o.capacity;
    q.obj[MF_TACKLE * q.obj_stride + k] = __int_as_float(o.tackle);
    q.obj[MF_CATCH_BAN * q.obj_stride + k] = __int_as_float(o.catch_ban);
    q.obj[MF_CARD * q.obj_stride + k] = __int_as_float(o.card);
  }
  if (l == BALL) {
    q.env[ME_CYCLE * q.env_stride + e] = g.cycle; q.env[ME_MODE * q.env_stride + e] = g.mode;
    q.env[ME_MODE_SIDE * q.env_stride + e] = g.mode_side; q.env[ME_SCORE_L * q.env_stride + e] = r.score_l;
    q.env[ME_SCORE_R * q.env_stride + e] = r.score_r; q.env[ME_LAST_TOUCH * q.env_stride + e] = g.last_touch;
    q.env[ME_TIMER * q.env_stride + e] = g.timer; q.env[ME_OFFSIDE * q.env_stride + e] = g.offside;
    q.env[ME_NEAREST_L * q.env_stride + e] = g.nearest_l; q.env[ME_NEAREST_R * q.env_stride + e] = g.nearest_r;
    q.env[ME_HOLDER * q.env_stride + e] = r.holder; q.env[ME_MOVES * q.env_stride + e] = r.moves;
    q.env[ME_TAKER * q.env_stride + e] = r.taker; q.env[ME_LAST_KICKER * q.env_stride + e] = r.last_kicker;
    q.env[ME_STOPPED * q.env_stride + e] = r.stopped; q.env[ME_TICK * q.env_stride + e] = g.tick;
    q.reward[e] = g.reward; q.done[e] = (uint8_t)g.done;
  }
}
// workgroup sums (LDS, m_count_events) -> ONE striped atomic per workgroup and counter (thousands of waves adding to one
// address cost ~12 ns each, serially)
S2D_DEV void m_flush_counts(unsigned long long* stats, const unsigned int* lds_cnt) {
  __syncthreads();
  if (threadIdx.x >= 1 && threadIdx.x < 8 && lds_cnt[threadIdx.x])
    atomicAdd(&stats[(blockIdx.x % S2D_STATS_STRIPES) * 8 + threadIdx.x], (unsigned long long)lds_cnt[threadIdx.x]);
}

__global__ __launch_bounds__(kMBlock) void s2d_match_reset_kernel(MParams p, MPtrs q, int64_t n, const uint8_t* __restrict__ mask) {
  const int l = threadIdx.x & (kHalf - 1);
  const int64_t e = (int64_t)blockIdx.x * kEnvsPerBlock + threadIdx.x / kHalf;
  if (e >= n) return;
  if (mask && !mask[e]) return;
  MObj o; MGame g; MRare r;
  m_reset(p, q.ptab[PT_EFFORT_MAX * kHalf + l], o, g, r, l);
  m_store(q, e, l, o, g, r);
}

struct MRoll { float* obs; float* reward; int32_t* mode; uint8_t* done; };

// n_steps cycles; actions = [T][N][22][3] or NULL (random policy).  n_steps = 1 with ro = {} is the per-step API.
struct MShared {                                      // the workgroup's LDS (declared by the kernel)
  float4 (*pos_tile)[kTileSlots]; PTab* pt; unsigned int* lds_cnt; MRare* rare; float (*obs_tile)[2 * SLOTS * S2D_MATCH_OBJ_WORDS];
};
template <class P, class TY>
S2D_DEV void match_rollout_body(const P& p, const TY& pt, const MShared& sh, const MPtrs& q, int64_t n, int n_steps,
                                const float* __restrict__ actions, const MRoll& ro) {
  const int l = threadIdx.x & (kHalf - 1), l_launch = l;
  const int half = (threadIdx.x >> 5) & 1, half_launch = half;
  const int64_t e = (int64_t)blockIdx.x * kEnvsPerBlock + threadIdx.x / kHalf;
  const bool valid = e < n;
  const int64_t ec = valid ? e : n - 1;              // lanes of out-of-range matches shadow the last match (no stores)
  MObj o; MGame g;
  MRare& r = sh.rare[threadIdx.x / kHalf];
  m_load(q, ec, l, o, g, r);
  m_derive(p, g);
  tile_init(sh.pos_tile[threadIdx.x / kHalf], l, pt[PT_SIZE][l]);
  const uint64_t gid = (((uint64_t)p.gid_hi << 32) | p.gid_lo) + (uint64_t)ec;
  const uint32_t gl = (uint32_t)gid, gh = (uint32_t)(gid >> 32);
  MCounts cnt{sh.lds_cnt, valid, 0u};
  U4 pol{0, 0, 0, 0};                                     // the policy block of the current pair of cycles
  // The state loaded above is first used inside the loop, and that is where the compiler would wait for it -- with s_waitcnt
  // vmcnt(k), a counter that on gfx9 also counts STORES: from the second cycle on those waits would hold the wave until the
  // previous cycles' record stores had been acknowledged by memory.  Waiting for the loads here leaves no wait in the loop.
  __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0)
  // The record rows of cycle t lie n matches behind those of cycle t - 1: one uniform pointer per array, advanced once per cycle
  // (scalar adds), plus a small per-lane offset inside the workgroup's stretch of the row -- 32 bits, so the stores take their base
  // from scalar registers.  (Formed per cycle from t, n and the match index, the four addresses were 64-bit vector arithmetic: ~45
  // of the cycle's ~550 vector instructions.)
  constexpr int kVecPerMatch = SLOTS * S2D_MATCH_OBJ_WORDS / 4;   // 30 float4 per match
  const int64_t eb = (int64_t)blockIdx.x * kEnvsPerBlock;  // the workgroup's first match
  char* obs_row = reinterpret_cast<char*>(ro.obs) + eb * (int64_t)(kVecPerMatch * 16);
  float* reward_row = ro.reward + eb; int32_t* mode_row = ro.mode + eb; uint8_t* done_row = ro.done + eb;
  const uint32_t lane = threadIdx.x & 63u, m_in_wg = threadIdx.x / kHalf;
  const uint32_t obs_off = ((threadIdx.x >> 6) * 2u * kVecPerMatch + lane) * 16u;
  const int64_t e0 = e - half;                             // first match of this wave (matches of a wave: e0, e0 + 1)
  const bool obs_lane = (int)lane < (e0 + 1 < n ? 2 * kVecPerMatch : (e0 < n ? kVecPerMatch : 0));
  float* const obs_slot = sh.obs_tile[threadIdx.x >> 6] + (half * SLOTS + l) * S2D_MATCH_OBJ_WORDS;
  const float4* const obs_vec = reinterpret_cast<const float4*>(sh.obs_tile[threadIdx.x >> 6]) + lane;
  // Fair shares of the SIMD.  Among waves of equal priority the issue arbiter prefers the OLDEST: of the four waves a SIMD holds,
  // the first one dispatched ran its 64 cycles in 322 k clocks, the fourth in 470 k (profiles/r03/match_wave_durations.txt) -- same
  // work, and the launch waits for the fourth.  The waves therefore take turns at the four priority levels, a new turn every four
  // cycles, starting from their slot number on the SIMD (HW_ID.wave_id): at any time the four hold four different levels, and over
  // a launch every wave spends the same time at each.
  const int simd_slot = (int)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);   // HW_REG_HW_ID bits 3:0
  for (int t = 0; t < n_steps; ++t) {
    if ((t & 3) == 0) {
      switch ((simd_slot + (t >> 2)) & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
      }
    }
    // Everything that depends only on the lane number -- masks such as "is a player", "is the ball", bit positions, Philox block
    // words -- is loop-invariant, and the compiler computes it all once per launch and keeps it: ~90 scalar and ~20 vector
    // registers more than there are, spilled and reloaded inside the loop.  One instruction each to recompute: opaque copies of
    // the lane number and the half make them per-cycle values.
    int l = l_launch, half = half_launch;
    asm volatile("" : "+v"(l), "+v"(half));
    int cmd = S2D_MCMD_NONE; float a = 0.0f, b = 0.0f;
    if (l < NP) {
      if (actions) {
        const float* ap = actions + (((int64_t)t * n + ec) * NP + l) * 3;
        cmd = (int)ap[0]; a = ap[1]; b = ap[2];
      } else {
        m_random_action(p, gl, gh, (uint32_t)g.tick, l, t == 0, pol, cmd, a, b);
      }
    }
    match_cycle(p, pt, o, g, r, l, half, gl, gh, cmd, a, b, cnt, sh.pos_tile[threadIdx.x / kHalf]);
    if (ro.obs) {                                          // wave-uniform
      if (l < SLOTS) { obs_slot[0] = o.x; obs_slot[1] = o.y; obs_slot[2] = o.vx; obs_slot[3] = o.vy; obs_slot[4] = o.body; }
      wave_fence();
      if (obs_lane) *reinterpret_cast<float4*>(obs_row + obs_off) = *obs_vec;
      wave_fence();
      obs_row += n * (int64_t)(kVecPerMatch * 16);
    }
    if (valid && l == BALL) {
      if (ro.reward) reward_row[m_in_wg] = g.reward;
      if (ro.mode) mode_row[m_in_wg] = g.mode;
      if (ro.done) done_row[m_in_wg] = (uint8_t)g.done;
    }
    reward_row += n; mode_row += n; done_row += n;         // (never dereferenced when the array is absent)
  }
  match_nearest(o, g, l);
  if (valid) m_store(q, e, l, o, g, r);
  {                                                     // tackles: per-lane counters -> wave sum -> LDS
    unsigned int tk = valid ? cnt.tackles : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tk += __shfl_down(tk, off);
    if ((threadIdx.x & 63) == 0 && tk) atomicAdd(&sh.lds_cnt[5], tk);
  }
  m_flush_counts(q.stats, sh.lds_cnt);
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&q.stats[0], (unsigned long long)n * (unsigned long long)n_steps);
}

// STOCK: the configuration words are MStock's constants (m_is_stock() said they equal this engine's); else they are read from LDS.
// STOCK_TYPES (with STOCK): all 22 players of the stock PlayerType, the table's entries are constants too.
// SCHED (with STOCK and STOCK_TYPES): the schedule words are the engine's own (MStockSched).  ILL (general only): the engine has
// IllegalDefense_ switched on.
template <bool STOCK, bool STOCK_TYPES, bool SCHED = false, bool ILL = false>
__global__ __launch_bounds__(kMBlock, 4) void s2d_match_rollout_kernel(MParams p_arg, MPtrs q, int64_t n, int n_steps,
                                                                     const float* __restrict__ actions, MRoll ro) {
  __shared__ float4 pos_tile[kEnvsPerBlock][kTileSlots];
  __shared__ PTab pt[PT_WORDS];                       // per-slot PlayerType parameters, shared by the 8 matches
  __shared__ unsigned int lds_cnt[8];
  __shared__ MRare rare[kEnvsPerBlock];                // per-match words only events touch (see MRare)
  // rollout observations: the two matches of a wave are neighbours in [T][N][24][5], i.e. 960 contiguous bytes per cycle.  Each lane
  // puts its five words into a wave-private tile (stride 5: no bank conflicts) and the wave stores the block as 60 x 16 bytes --
  // instead of five 20-byte-strided dword stores per lane (partial lines: what the reach kernels' store-pattern study priced)
  __shared__ __attribute__((aligned(16))) float obs_tile[kMBlock / 64][2 * SLOTS * S2D_MATCH_OBJ_WORDS];
  const MShared sh{pos_tile, pt, lds_cnt, rare, obs_tile};
  static_assert(STOCK || !STOCK_TYPES, "constant types come with constant rules");
  if constexpr (!STOCK_TYPES)
    for (int k = threadIdx.x; k < PT_WORDS * kHalf; k += kMBlock) (&pt[0][0])[k] = q.ptab[k];
  if (threadIdx.x < 8) lds_cnt[threadIdx.x] = 0u;
  static_assert(!SCHED || (STOCK && STOCK_TYPES), "the engine's own schedule comes with constant rules and types");
  if constexpr (SCHED) {
    __syncthreads();
#define X(name) , p_arg.name
    const MStockSched p{p_arg.auto_reset, p_arg.noise, p_arg.seed_lo, p_arg.seed_hi, p_arg.gid_lo, p_arg.gid_hi M_SCHEDULE_INTS(X)};
#undef X
    const MStockTypes types{__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q.ptab[PT_KICKABLE_AREA2 * kHalf])))};
    match_rollout_body(p, types, sh, q, n, n_steps, actions, ro);
  } else if constexpr (STOCK) {
    __syncthreads();
    const MStock p{p_arg.auto_reset, p_arg.noise, p_arg.penalty_shoot_outs, p_arg.seed_lo, p_arg.seed_hi, p_arg.gid_lo, p_arg.gid_hi};
    if constexpr (STOCK_TYPES) {
      const MStockTypes types{__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(q.ptab[PT_KICKABLE_AREA2 * kHalf])))};
      match_rollout_body(p, types, sh, q, n, n_steps, actions, ro);
    } else {
      match_rollout_body(p, static_cast<const PTab*>(pt), sh, q, n, n_steps, actions, ro);
    }
  } else {
    // The ~70 uniform parameters are read from LDS (broadcast reads) where they are used instead of
    // occupying SGPRs for the whole kernel: as kernargs they cost 142 SGPR spills and 48 B of scratch at
    // the 128-VGPR cap (26 spills / 12 B this way, +14 % throughput).
    static_assert(!ILL || !STOCK, "the stock configurations have the rule off");
    using PBlock = std::conditional_t<ILL, MParams, MParamsNoIll>;
    __shared__ PBlock p_lds;
    static_assert(sizeof(MParams) / 4 <= kMBlock, "one thread per parameter word");
    if (threadIdx.x < sizeof(MParams) / 4)
      reinterpret_cast<uint32_t*>(&p_lds)[threadIdx.x] = reinterpret_cast<const uint32_t*>(&p_arg)[threadIdx.x];
    __syncthreads();
    match_rollout_body(p_lds, static_cast<const PTab*>(pt), sh, q, n, n_steps, actions, ro);
  }
}

// Relative tables (Player.dist_from_self / angle_from_self of every agent's WorldModel): lane p scans the
// 23 objects of its match through a broadcast-read LDS tile and writes one row of 23 values.
__global__ __launch_bounds__(kMBlock) void s2d_match_relative_kernel(MPtrs q, int64_t n, float* __restrict__ dist,
                                                                      float* __restrict__ angle) {
  __shared__ float2 pos_tile[kEnvsPerBlock][kHalf];
  const int l = threadIdx.x & (kHalf - 1);
  const int64_t e = (int64_t)blockIdx.x * kEnvsPerBlock + threadIdx.x / kHalf;
  const bool valid = e < n;
  const int64_t ec = valid ? e : n - 1;
  float x = 0.0f, y = 0.0f;
  if (l <= BALL) { x = q.obj[MF_X * q.obj_stride + ec * SLOTS + l]; y = q.obj[MF_Y * q.obj_stride + ec * SLOTS + l]; }
  float2* pos = pos_tile[threadIdx.x / kHalf];
  pos[l] = make_float2(x, y);
  wave_fence();
  if (valid && l < NP) {
    float* drow = dist + ((e * NP + l) * (int64_t)(BALL + 1));
    float* arow = angle + ((e * NP + l) * (int64_t)(BALL + 1));
    for (int j = 0; j <= BALL; ++j) {
      const float2 pj = pos[j];
      float dx = pj.x - x, dy = pj.y - y;
      drow[j] = j == l ? 0.0f : hypot2(dx, dy);
      arow[j] = j == l ? 0.0f : atan2_deg(dy, dx);
    }
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct S2DMatchEngine {
  S2DMatchConfig cfg; MParams mp; float ptab[PT_WORDS][kHalf]; int64_t n, stride; int device;
  bool stock = false;                                  // mp's configuration words equal MStock: launches use the constant-folded kernels
  bool stock_types = false;                            // ... and every player is of the stock PlayerType (ptab's entries equal MStockTypes)
  bool stock_sched = false;                            // stock rules, physics and types, the engine's own schedule (MStockSched)
  char* arena; size_t arena_bytes; bool owns_arena;
  S2DMatchBuffers buf; MPtrs ptrs;
};

// errors share the thread-local text of s2d_last_error() (defined in s2d_engine.hip)
extern "C" void s2d_internal_set_error(const char* msg);
static int mfail(int code, const std::string& msg) { s2d_internal_set_error(msg.c_str()); return code; }
#define MHIP_TRY(expr)                                                                          \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) return mfail(S2D_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

static size_t m_align(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct MLayout { size_t obj, env, reward, done, stats, ptab, total; int64_t stride; };
static MLayout m_layout(int64_t n) {
  MLayout L; L.stride = (int64_t)m_align((size_t)n, 64);
  size_t off = 0;
  L.obj = off; off += m_align((size_t)MF_OBJ_PLANES * (size_t)L.stride * SLOTS * 4, 256);
  L.env = off; off += m_align((size_t)ME_ENV_PLANES * (size_t)L.stride * 4, 256);
  L.reward = off; off += m_align((size_t)L.stride * 4, 256);
  L.done = off; off += m_align((size_t)L.stride, 256);
  L.stats = off; off += (size_t)S2D_STATS_STRIPES * 8 * sizeof(unsigned long long);
  L.ptab = off; off += m_align((size_t)PT_WORDS * kHalf * 4, 256);
  L.total = off;
  return L;
}

// PlayerType 0 = the ServerParam values
static S2DPlayerType m_default_type(const S2DServerParams& s, const S2DMatchParams& m) {
  S2DPlayerType t;
  t.player_speed_max = s.player_speed_max; t.stamina_inc_max = s.stamina_inc_max; t.player_decay = s.player_decay;
  t.inertia_moment = s.inertia_moment; t.dash_power_rate = s.dash_power_rate; t.player_size = s.player_size;
  t.kickable_margin = m.kickable_margin; t.kick_rand = m.kick_rand; t.extra_stamina = s.extra_stamina;
  t.effort_max = s.effort_init; t.effort_min = s.effort_min; t.kick_power_rate = m.kick_power_rate;
  t.catchable_area_l_stretch = 1.0;
  return t;
}

S2D_API void s2d_match_default_player_params(S2DPlayerParams* q) {   // rcssserver stock player.conf (EXT)
  if (!q) return;
  std::memset(q, 0, sizeof *q);
  q->player_speed_max_delta_min = 0.0; q->player_speed_max_delta_max = 0.0; q->stamina_inc_max_delta_factor = 0.0;
  q->player_decay_delta_min = -0.1; q->player_decay_delta_max = 0.1; q->inertia_moment_delta_factor = 25.0;
  q->dash_power_rate_delta_min = 0.0; q->dash_power_rate_delta_max = 0.0; q->player_size_delta_factor = -100.0;
  q->kickable_margin_delta_min = -0.1; q->kickable_margin_delta_max = 0.1; q->kick_rand_delta_factor = 1.0;
  q->extra_stamina_delta_min = 0.0; q->extra_stamina_delta_max = 50.0;
  q->effort_max_delta_factor = -0.004; q->effort_min_delta_factor = -0.004;
  q->new_dash_power_rate_delta_min = -0.0012; q->new_dash_power_rate_delta_max = 0.0008;
  q->new_stamina_inc_max_delta_factor = -6000.0;
  q->kick_power_rate_delta_min = 0.0; q->kick_power_rate_delta_max = 0.0;
  q->catchable_area_l_stretch_min = 1.0; q->catchable_area_l_stretch_max = 1.3;
}

// host-side Philox (same function as the device's) for the type generator
static void h_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// rcssserver's HeteroPlayer recipe (EXT, as published): each non-default type draws one delta per
// trade-off pair from the PlayerParam ranges -- faster decay <-> more inertia, stronger dash <-> less
// stamina income, wider kickable margin <-> noisier kicks, more extra stamina <-> lower effort range --
// and is re-drawn until its top speed (effort_max * dash_power_rate * max_dash_power / (1 - decay))
// lies in (0.75, player_speed_max].  Uniforms: Philox stream S2D_ST_TYPES at counter (type id, try).
S2D_API int s2d_match_generate_player_types(S2DMatchConfig* c, const S2DPlayerParams* pp, uint64_t seed) {
  if (!c) return mfail(S2D_EINVAL, "config is NULL");
  S2DPlayerParams stock;
  if (!pp) { s2d_match_default_player_params(&stock); pp = &stock; }
  const S2DServerParams& s = c->sp; const S2DMatchParams& m = c->mp;
  const S2DPlayerType base = m_default_type(s, m);
  c->player_types[0] = base;
  for (int id = 1; id < S2D_MATCH_PLAYER_TYPES; ++id) {
    S2DPlayerType t = base;
    for (int attempt = 0; attempt < 1000; ++attempt) {
      double u[12];
      for (int b = 0; b < 3; ++b) {
        uint32_t w[4];
        h_philox((uint32_t)id, (uint32_t)attempt, 0u, (S2D_ST_TYPES << 16) | (uint32_t)b, (uint32_t)seed, (uint32_t)(seed >> 32), w);
        for (int k = 0; k < 4; ++k) u[b * 4 + k] = (double)(w[k] >> 8) * 5.9604644775390625e-8;
      }
      auto U = [&](int k, double lo, double hi) { return lo + u[k] * (hi - lo); };
      t = base;
      double d = U(0, pp->player_speed_max_delta_min, pp->player_speed_max_delta_max);
      t.player_speed_max += d; t.stamina_inc_max += d * pp->stamina_inc_max_delta_factor;
      d = U(1, pp->player_decay_delta_min, pp->player_decay_delta_max);
      t.player_decay += d; t.inertia_moment += d * pp->inertia_moment_delta_factor;
      d = U(2, pp->dash_power_rate_delta_min, pp->dash_power_rate_delta_max);
      t.dash_power_rate += d; t.player_size += d * pp->player_size_delta_factor;
      d = U(3, pp->new_dash_power_rate_delta_min, pp->new_dash_power_rate_delta_max);
      t.dash_power_rate += d; t.stamina_inc_max += d * pp->new_stamina_inc_max_delta_factor;
      d = U(4, pp->kickable_margin_delta_min, pp->kickable_margin_delta_max);
      t.kickable_margin += d; t.kick_rand += d * pp->kick_rand_delta_factor;
      d = U(5, pp->extra_stamina_delta_min, pp->extra_stamina_delta_max);
      t.extra_stamina += d; t.effort_max += d * pp->effort_max_delta_factor; t.effort_min += d * pp->effort_min_delta_factor;
      d = U(6, pp->kick_power_rate_delta_min, pp->kick_power_rate_delta_max);
      t.kick_power_rate += d;
      t.catchable_area_l_stretch = U(7, pp->catchable_area_l_stretch_min, pp->catchable_area_l_stretch_max);
      if (!(t.player_decay > 0.0 && t.player_decay < 1.0) || !(t.kickable_margin > 0.0) || !(t.player_size > 0.0)) continue;
      const double real_speed_max = t.effort_max * t.dash_power_rate * s.max_dash_power / (1.0 - t.player_decay);
      if (real_speed_max > 0.75 && real_speed_max <= t.player_speed_max) break;
      t = base;                                           // exhausted tries fall back to the default type
    }
    c->player_types[id] = t;
  }
  return S2D_OK;
}

S2D_API void s2d_match_default_config(S2DMatchConfig* c) {
  if (!c) return;
  S2DConfig base; s2d_default_config(&base);
  std::memset(c, 0, sizeof *c);
  c->abi_version = S2D_ABI_VERSION; c->struct_bytes = (uint32_t)sizeof(S2DMatchConfig);
  c->sp = base.sp;
  S2DMatchParams& m = c->mp;   // rcssserver stock values (EXT)
  m.kick_power_rate = 0.027; m.kickable_margin = 0.7; m.kick_rand = 0.1; m.max_power = 100.0; m.min_power = -100.0;
  m.tackle_dist = 2.0; m.tackle_back_dist = 0.0; m.tackle_width = 1.25; m.tackle_power_rate = 0.027;
  m.max_tackle_power = 100.0; m.max_back_tackle_power = 0.0;
  m.goal_width = 14.02; m.offside_active_area_size = 2.5; m.free_kick_distance = 9.15;
  m.tackle_cycles = 10; m.half_time_cycles = 3000; m.nr_normal_halfs = 2; m.drop_ball_time = 100; m.use_offside = 1;
  m.catch_ban_cycle = 5; m.catchable_area_l = 1.2; m.catch_area_w = 1.0; m.catch_probability = 1.0;
  m.max_catch_angle = 90.0; m.min_catch_angle = -90.0; m.penalty_area_length = 16.5; m.penalty_area_half_width = 20.16;
  m.goalie_max_moves = 2; m.after_goal_wait = 50;
  m.kick_off_wait = 0; m.back_passes = 1; m.free_kick_faults = 1;
  m.stopped_clock = 1; m.announce_wait = 30; m.foul_cycles = 5; m.foul_detect_probability = 0.5;
  m.nr_extra_halfs = S2D_STOCK_EXTRA_HALFS; m.extra_half_cycles = 1000; m.golden_goal = 0;
  m.penalty_shoot_outs = S2D_STOCK_SHOOT_OUTS; m.pen_before_setup_wait = 10; m.pen_ready_wait = 10; m.pen_taken_wait = 150; m.pen_nr_kicks = 5;
  m.pen_max_extra_kicks = 5; m.pen_dist_x = 42.5;
  m.pen_allow_mult_kicks = 1; m.pen_random_winner = 0;
  m.illegal_defense_number = 0; m.illegal_defense_duration = 20; m.illegal_defense_dist_x = 16.5; m.illegal_defense_width = 40.32;
  c->seed = 0x5EEDull; c->env_id_offset = 0; c->auto_reset = 1; c->noise = 0;
  for (int t = 0; t < S2D_MATCH_PLAYER_TYPES; ++t) c->player_types[t] = m_default_type(c->sp, m);   // homogeneous
}

S2D_API int s2d_match_validate_config(const S2DMatchConfig* c) {
  if (!c) return mfail(S2D_EINVAL, "config is NULL");
  if (c->abi_version != S2D_ABI_VERSION) return mfail(S2D_EINVAL, "config.abi_version mismatch");
  if (c->struct_bytes != sizeof(S2DMatchConfig)) return mfail(S2D_EINVAL, "config.struct_bytes != sizeof(S2DMatchConfig)");
  if (!(c->sp.pitch_half_length > 0) || !(c->sp.pitch_half_width > 0)) return mfail(S2D_EINVAL, "pitch extents must be > 0");
  if (!(c->mp.kickable_margin > 0) || !(c->mp.max_power > 0) || !(c->mp.tackle_width > 0))
    return mfail(S2D_EINVAL, "kickable_margin, max_power and tackle_width must be > 0");
  if (c->mp.half_time_cycles < 1 || c->mp.nr_normal_halfs < 1) return mfail(S2D_EINVAL, "half_time_cycles and nr_normal_halfs must be >= 1");
  if (c->mp.tackle_cycles < 0 || c->mp.drop_ball_time < 0) return mfail(S2D_EINVAL, "tackle_cycles / drop_ball_time must be >= 0");
  if (c->env_id_offset < 0) return mfail(S2D_EINVAL, "env_id_offset must be >= 0");
  if (c->mp.goalie_max_moves < 0 || c->mp.after_goal_wait < 0 || c->mp.kick_off_wait < 0)
    return mfail(S2D_EINVAL, "goalie_max_moves / after_goal_wait / kick_off_wait must be >= 0");
  if (c->mp.catch_ban_cycle < 0 || !(c->mp.catch_area_w > 0) || !(c->mp.catchable_area_l > 0))
    return mfail(S2D_EINVAL, "catch_ban_cycle must be >= 0, catch_area_w and catchable_area_l > 0");
  if (c->mp.announce_wait < 0 || c->mp.foul_cycles < 0 || !(c->mp.foul_detect_probability >= 0 && c->mp.foul_detect_probability <= 1))
    return mfail(S2D_EINVAL, "announce_wait / foul_cycles must be >= 0, foul_detect_probability in [0, 1]");
  if (c->mp.nr_extra_halfs < 0 || (c->mp.nr_extra_halfs > 0 && c->mp.extra_half_cycles < 1))
    return mfail(S2D_EINVAL, "nr_extra_halfs must be >= 0 and extra_half_cycles >= 1 when extra halves are played");
  if (c->mp.pen_before_setup_wait < 0 || c->mp.pen_ready_wait < 0 || c->mp.pen_taken_wait < 0 || c->mp.pen_nr_kicks < 1 ||
      c->mp.pen_max_extra_kicks < 0 || c->mp.pen_nr_kicks + c->mp.pen_max_extra_kicks > 15)
    return mfail(S2D_EINVAL, "pen_*_wait must be >= 0, pen_nr_kicks >= 1, pen_max_extra_kicks >= 0 and their sum <= 15");
  if (c->mp.illegal_defense_number < 0 || c->mp.illegal_defense_duration < 1 || c->mp.illegal_defense_duration > 255)
    return mfail(S2D_EINVAL, "illegal_defense_number must be >= 0 and illegal_defense_duration in [1, 255]");
  for (int i = 0; i < S2D_MATCH_PLAYERS; ++i)
    if (c->player_type_id[i] < 0 || c->player_type_id[i] >= S2D_MATCH_PLAYER_TYPES)
      return mfail(S2D_EINVAL, "player_type_id entries must be in [0, 18)");
  for (int t = 0; t < S2D_MATCH_PLAYER_TYPES; ++t) {
    const S2DPlayerType& y = c->player_types[t];
    if (!(y.player_decay >= 0 && y.player_decay <= 1) || !(y.kickable_margin > 0) || !(y.player_size > 0) ||
        !(y.player_speed_max > 0))
      return mfail(S2D_EINVAL, "player_types: decay must be in [0,1], kickable_margin / player_size / player_speed_max > 0");
  }
  return S2D_OK;
}

static void mparams_from_config(const S2DMatchConfig& c, MParams& p, float (*ptab)[kHalf]) {
  const S2DServerParams& s = c.sp; const S2DMatchParams& m = c.mp;
  std::memset(&p, 0, sizeof p);
  p.half_l = (float)s.pitch_half_length; p.half_w = (float)s.pitch_half_width;
  p.ball_size = (float)s.ball_size;
  p.player_rand = (float)s.player_rand; p.ball_rand = (float)s.ball_rand;
  p.player_accel_max = (float)s.player_accel_max; p.player_accel_max2 = p.player_accel_max * p.player_accel_max;
  p.ball_speed_max = (float)s.ball_speed_max; p.ball_speed_max2 = p.ball_speed_max * p.ball_speed_max;
  p.ball_accel_max = (float)s.ball_accel_max; p.ball_accel_max2 = p.ball_accel_max * p.ball_accel_max;
  p.stamina_max = (float)s.stamina_max; p.stamina_capacity = (float)s.stamina_capacity;
  p.recover_init = (float)s.recover_init; p.recover_dec_thr_value = (float)(s.recover_dec_thr * s.stamina_max);
  p.recover_min = (float)s.recover_min; p.recover_dec = (float)s.recover_dec;
  p.effort_dec_thr_value = (float)(s.effort_dec_thr * s.stamina_max); p.effort_dec = (float)s.effort_dec;
  p.effort_inc_thr_value = (float)(s.effort_inc_thr * s.stamina_max); p.effort_inc = (float)s.effort_inc;
  p.max_dash_power = (float)s.max_dash_power;
  p.min_dash_power = (float)s.min_dash_power; p.max_dash_angle = (float)s.max_dash_angle;
  p.min_dash_angle = (float)s.min_dash_angle; p.dash_angle_step = (float)s.dash_angle_step;
  p.inv_dash_angle_step = s.dash_angle_step > 0 ? (float)(1.0 / s.dash_angle_step) : 0.0f;
  p.side_dash_rate = (float)s.side_dash_rate; p.back_dash_rate = (float)s.back_dash_rate;
  p.max_moment = (float)s.max_moment; p.min_moment = (float)s.min_moment;
  p.collision_vel_rate = (float)s.collision_vel_rate;
  p.max_power = (float)m.max_power; p.min_power = (float)m.min_power;
  p.inv_max_power = (float)(1.0 / m.max_power);
  p.tackle_dist = (float)m.tackle_dist; p.tackle_back_dist = (float)m.tackle_back_dist;
  p.tackle_width = (float)m.tackle_width; p.tackle_power_rate = (float)m.tackle_power_rate;
  p.max_tackle_power = (float)m.max_tackle_power; p.max_back_tackle_power = (float)m.max_back_tackle_power;
  {
    const double td = m.tackle_dist > m.tackle_back_dist ? m.tackle_dist : m.tackle_back_dist;
    p.tackle_reach2 = (float)(1.01 * (td * td + m.tackle_width * m.tackle_width));
  }
  p.goal_half_width = (float)(m.goal_width * 0.5);
  p.offside_area2 = (float)(m.offside_active_area_size * m.offside_active_area_size);
  p.free_kick_distance = (float)m.free_kick_distance;
  p.inv_speed_decay = (float)(1.0 / (s.ball_speed_max * s.ball_decay));
  p.catch_half_w = (float)(m.catch_area_w * 0.5); p.catch_probability = (float)m.catch_probability;
  p.max_catch_angle = (float)m.max_catch_angle; p.min_catch_angle = (float)m.min_catch_angle;
  p.pen_x = (float)(s.pitch_half_length - m.penalty_area_length); p.pen_half_w = (float)m.penalty_area_half_width;
  p.tackle_cycles = m.tackle_cycles; p.half_time_cycles = m.half_time_cycles;
  p.nr_normal_halfs = m.nr_normal_halfs; p.drop_ball_time = m.drop_ball_time; p.use_offside = m.use_offside;
  p.catch_ban_cycle = m.catch_ban_cycle; p.goalie_max_moves = m.goalie_max_moves; p.after_goal_wait = m.after_goal_wait;
  p.kick_off_wait = m.kick_off_wait; p.back_passes = m.back_passes; p.free_kick_faults = m.free_kick_faults;
  p.stopped_clock = m.stopped_clock; p.announce_wait = m.announce_wait; p.foul_cycles = m.foul_cycles;
  p.foul_detect_probability = (float)m.foul_detect_probability;
  p.nr_extra_halfs = m.nr_extra_halfs; p.extra_half_cycles = m.extra_half_cycles; p.golden_goal = m.golden_goal != 0;
  p.penalty_shoot_outs = m.penalty_shoot_outs != 0; p.pen_before_setup_wait = m.pen_before_setup_wait; p.pen_ready_wait = m.pen_ready_wait;
  p.pen_taken_wait = m.pen_taken_wait; p.pen_nr_kicks = m.pen_nr_kicks; p.pen_max_extra_kicks = m.pen_max_extra_kicks;
  p.pen_spot_x = (float)(s.pitch_half_length - m.pen_dist_x);
  p.illegal_defense_number = m.illegal_defense_number; p.illegal_defense_duration = m.illegal_defense_duration;
  p.pen_allow_mult_kicks = m.pen_allow_mult_kicks != 0; p.pen_random_winner = m.pen_random_winner != 0;
  p.ill_x = (float)(s.pitch_half_length - m.illegal_defense_dist_x); p.ill_half_w = (float)(m.illegal_defense_width * 0.5);
  p.total_cycles = m.half_time_cycles * m.nr_normal_halfs;
  p.end_cycles = p.total_cycles + (m.nr_extra_halfs > 0 ? m.extra_half_cycles * m.nr_extra_halfs : 0);
  p.auto_reset = c.auto_reset; p.noise = c.noise;
  p.seed_lo = (uint32_t)c.seed; p.seed_hi = (uint32_t)(c.seed >> 32);
  p.gid_lo = (uint32_t)(uint64_t)c.env_id_offset; p.gid_hi = (uint32_t)((uint64_t)c.env_id_offset >> 32);
  // per-slot table: column i = the PlayerType of player i, column 22 = the ball (size / decay rows)
  std::memset(ptab, 0, sizeof(float) * PT_WORDS * kHalf);
  for (int i = 0; i < NP; ++i) {
    const S2DPlayerType& t = c.player_types[c.player_type_id[i]];
    const float size = (float)t.player_size, margin = (float)t.kickable_margin;
    ptab[PT_SPEED_MAX][i] = (float)t.player_speed_max; ptab[PT_SPEED_MAX2][i] = ptab[PT_SPEED_MAX][i] * ptab[PT_SPEED_MAX][i];
    ptab[PT_STAMINA_INC][i] = (float)t.stamina_inc_max; ptab[PT_DECAY][i] = (float)t.player_decay;
    ptab[PT_INERTIA][i] = (float)t.inertia_moment; ptab[PT_DASH_RATE][i] = (float)t.dash_power_rate;
    ptab[PT_SIZE][i] = size; ptab[PT_INV_KICK_MARGIN][i] = (float)(1.0 / t.kickable_margin);
    {  // largest float T with sqrtf(T) <= kickable_area  (dist <= ka  <=>  dist^2 <= T)
      const float ka = size + p.ball_size + margin;
      float T = ka * ka;
      while (std::sqrt(T) > ka) T = std::nextafter(T, 0.0f);
      while (std::sqrt(std::nextafter(T, INFINITY)) <= ka) T = std::nextafter(T, INFINITY);
      ptab[PT_KICKABLE_AREA2][i] = T;
    }
    ptab[PT_KICK_RAND][i] = (float)t.kick_rand; ptab[PT_EXTRA_STAMINA][i] = (float)t.extra_stamina;
    ptab[PT_EFFORT_MAX][i] = (float)t.effort_max; ptab[PT_EFFORT_MIN][i] = (float)t.effort_min;
    ptab[PT_KICK_RATE][i] = (float)t.kick_power_rate;
    ptab[PT_CATCH_LEN][i] = (float)(m.catchable_area_l * t.catchable_area_l_stretch);
  }
  ptab[PT_SIZE][BALL] = p.ball_size; ptab[PT_DECAY][BALL] = (float)s.ball_decay;
}

// the configuration words of p, bit for bit, against MStock's constants
static bool m_is_stock(const MParams& p) {
  auto same = [](float a, float b) { return std::memcmp(&a, &b, sizeof a) == 0; };
  bool ok = true;
#define X(name) ok = ok && same(p.name, (float)MStock::name);
  M_CONFIG_FLOATS(X)
#undef X
#define X(name) ok = ok && p.name == (int)MStock::name;
  M_CONFIG_INTS(X)
#undef X
  return ok;
}

// ... all of them but the schedule words (MStockSched keeps those as variables)
static bool m_rules_are_stock(const MParams& p) {
  auto same = [](float a, float b) { return std::memcmp(&a, &b, sizeof a) == 0; };
  bool ok = true;
#define X(name) ok = ok && same(p.name, (float)MStockSched::name);
  M_CONFIG_FLOATS(X)
#undef X
  return ok && p.tackle_cycles == MStockSched::tackle_cycles && p.use_offside == MStockSched::use_offside &&
         p.catch_ban_cycle == MStockSched::catch_ban_cycle && p.goalie_max_moves == MStockSched::goalie_max_moves &&
         p.back_passes == MStockSched::back_passes && p.free_kick_faults == MStockSched::free_kick_faults &&
         p.stopped_clock == MStockSched::stopped_clock && p.foul_cycles == MStockSched::foul_cycles &&
         p.illegal_defense_number == MStockSched::illegal_defense_number && p.illegal_defense_duration == MStockSched::illegal_defense_duration &&
         p.pen_allow_mult_kicks == MStockSched::pen_allow_mult_kicks && p.pen_random_winner == MStockSched::pen_random_winner;
}

static bool m_types_are_stock(const float (*t)[kHalf]) {
  auto same = [](float a, float b) { return std::memcmp(&a, &b, sizeof a) == 0; };
  bool ok = same(t[PT_SIZE][BALL], MStock::ball_size) && same(t[PT_DECAY][BALL], MStockTypes::ball_decay);
  for (int i = 0; i < NP && ok; ++i) {
    ok = same(t[PT_SPEED_MAX][i], MStockTypes::speed_max) && same(t[PT_SPEED_MAX2][i], MStockTypes::speed_max2) &&
         same(t[PT_STAMINA_INC][i], MStockTypes::stamina_inc) && same(t[PT_DECAY][i], MStockTypes::decay) &&
         same(t[PT_INERTIA][i], MStockTypes::inertia) && same(t[PT_DASH_RATE][i], MStockTypes::dash_rate) &&
         same(t[PT_SIZE][i], MStockTypes::size) && same(t[PT_INV_KICK_MARGIN][i], MStockTypes::inv_kick_margin) &&
         same(t[PT_KICKABLE_AREA2][i], t[PT_KICKABLE_AREA2][0]) && same(t[PT_KICK_RAND][i], MStockTypes::kick_rand) &&
         same(t[PT_EXTRA_STAMINA][i], MStockTypes::extra_stamina) && same(t[PT_EFFORT_MAX][i], MStockTypes::effort_max) &&
         same(t[PT_EFFORT_MIN][i], MStockTypes::effort_min) && same(t[PT_KICK_RATE][i], MStockTypes::kick_rate) &&
         same(t[PT_CATCH_LEN][i], MStockTypes::catch_len);
  }
  return ok;
}

S2D_API size_t s2d_match_arena_bytes(const S2DMatchConfig* cfg, int64_t n_envs) {
  if (!cfg || n_envs <= 0) return 0;
  return m_layout(n_envs).total;
}

struct MDeviceGuard {
  int prev = -1; bool ok = false;
  explicit MDeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess) ok = (prev == dev) || (hipSetDevice(dev) == hipSuccess); }
  ~MDeviceGuard() { if (ok && prev >= 0) (void)hipSetDevice(prev); }
};
static unsigned m_grid(int64_t n) { return (unsigned)((n + kEnvsPerBlock - 1) / kEnvsPerBlock); }

S2D_API int s2d_match_reset(S2DMatchHandle h, const uint8_t* mask_dev, void* stream) {
  if (!h) return mfail(S2D_EINVAL, "NULL handle");
  MDeviceGuard guard(h->device);
  hipLaunchKernelGGL(s2d_match_reset_kernel, dim3(m_grid(h->n)), dim3(kMBlock), 0, static_cast<hipStream_t>(stream), h->mp,
                     h->ptrs, h->n, mask_dev);
  MHIP_TRY(hipGetLastError());
  return S2D_OK;
}

S2D_API int s2d_match_create(const S2DMatchConfig* cfg, int64_t n_envs, int device, void* arena_dev, size_t arena_bytes,
                             void* stream, S2DMatchHandle* out) {
  if (!out) return mfail(S2D_EINVAL, "out handle is NULL");
  *out = nullptr;
  int rc = s2d_match_validate_config(cfg);
  if (rc != S2D_OK) return rc;
  if (n_envs <= 0 || n_envs > (int64_t)1 << 28) return mfail(S2D_EINVAL, "n_envs must be in [1, 2^28]");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return mfail(S2D_ENODEV, "no HIP device visible");
  if (device < 0 || device >= ndev) return mfail(S2D_EINVAL, "device index out of range");
  MDeviceGuard guard(device);
  if (!guard.ok) return mfail(S2D_EHIP, "hipSetDevice failed");
  MLayout L = m_layout(n_envs);
  S2DMatchEngine* h = new (std::nothrow) S2DMatchEngine();
  if (!h) return mfail(S2D_ENOMEM, "host allocation failed");
  h->cfg = *cfg; h->n = n_envs; h->stride = L.stride; h->device = device;
  mparams_from_config(*cfg, h->mp, h->ptab);
  {                                                     // S2D_MATCH_GENERAL_KERNEL=1: the general instantiation whatever the configuration (tests, A/B)
    const char* general = std::getenv("S2D_MATCH_GENERAL_KERNEL");
    h->stock = m_is_stock(h->mp) && !(general && general[0] == '1');
    h->stock_types = h->stock && m_types_are_stock(h->ptab);
    h->stock_sched = !h->stock && m_rules_are_stock(h->mp) && m_types_are_stock(h->ptab) && !(general && general[0] == '1');
  }
  if (arena_dev) {
    if (arena_bytes < L.total) { delete h; return mfail(S2D_ENOMEM, "arena smaller than s2d_match_arena_bytes()"); }
    if (reinterpret_cast<uintptr_t>(arena_dev) & 255u) { delete h; return mfail(S2D_EINVAL, "arena must be 256-byte aligned"); }
    h->arena = static_cast<char*>(arena_dev); h->owns_arena = false;
  } else {
    void* pmem = nullptr;
    if (hipMalloc(&pmem, L.total) != hipSuccess) { delete h; return mfail(S2D_ENOMEM, "hipMalloc of the arena failed"); }
    h->arena = static_cast<char*>(pmem); h->owns_arena = true;
  }
  h->arena_bytes = L.total;
  float* obj = reinterpret_cast<float*>(h->arena + L.obj);
  int32_t* env = reinterpret_cast<int32_t*>(h->arena + L.env);
  const size_t os = (size_t)L.stride * SLOTS, es = (size_t)L.stride;
  S2DMatchBuffers& b = h->buf;
  b.n_envs = n_envs;
  b.x = obj + MF_X * os; b.y = obj + MF_Y * os; b.vx = obj + MF_VX * os; b.vy = obj + MF_VY * os; b.body = obj + MF_BODY * os;
  b.stamina = obj + MF_STAMINA * os; b.effort = obj + MF_EFFORT * os; b.recovery = obj + MF_RECOVERY * os;
  b.stamina_capacity = obj + MF_CAPACITY * os; b.tackle_cycles = reinterpret_cast<int32_t*>(obj + MF_TACKLE * os);
  b.catch_ban = reinterpret_cast<int32_t*>(obj + MF_CATCH_BAN * os);
  b.cycle = env + ME_CYCLE * es; b.mode = env + ME_MODE * es; b.mode_side = env + ME_MODE_SIDE * es;
  b.score_left = env + ME_SCORE_L * es; b.score_right = env + ME_SCORE_R * es; b.last_touch_side = env + ME_LAST_TOUCH * es;
  b.setplay_timer = env + ME_TIMER * es; b.offside_mask = env + ME_OFFSIDE * es;
  b.ball_holder = env + ME_HOLDER * es; b.goalie_moves = env + ME_MOVES * es;
  b.set_play_taker = env + ME_TAKER * es; b.last_kicker = env + ME_LAST_KICKER * es;
  b.stopped_cycle = env + ME_STOPPED * es; b.tick = env + ME_TICK * es;
  b.card = reinterpret_cast<int32_t*>(obj + MF_CARD * os);
  b.reward_left = reinterpret_cast<float*>(h->arena + L.reward);
  b.done = reinterpret_cast<uint8_t*>(h->arena + L.done);
  b.nearest_left = env + ME_NEAREST_L * es; b.nearest_right = env + ME_NEAREST_R * es;
  b.stats = reinterpret_cast<unsigned long long*>(h->arena + L.stats);
  h->ptrs = MPtrs{obj, env, b.reward_left, b.done, b.stats, (int64_t)os, (int64_t)es,
                  reinterpret_cast<const float*>(h->arena + L.ptab)};
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(h->arena, 0, L.total, st);
  // h->ptab lives as long as the handle, so the (possibly staged) copy may complete later
  if (e == hipSuccess) e = hipMemcpyAsync(h->arena + L.ptab, h->ptab, sizeof h->ptab, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(s2d_match_reset_kernel, dim3(m_grid(n_envs)), dim3(kMBlock), 0, st, h->mp, h->ptrs, h->n,
                       (const uint8_t*)nullptr);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    std::string msg = std::string("arena initialisation: ") + hipGetErrorString(e);
    if (h->owns_arena) (void)hipFree(h->arena);
    delete h;
    return mfail(S2D_EHIP, msg);
  }
  *out = h;
  return S2D_OK;
}

S2D_API void s2d_match_destroy(S2DMatchHandle h) {
  if (!h) return;
  if (h->owns_arena && h->arena) { MDeviceGuard guard(h->device); (void)hipFree(h->arena); }
  delete h;
}
S2D_API int s2d_match_buffers(S2DMatchHandle h, S2DMatchBuffers* out) {
  if (!h || !out) return mfail(S2D_EINVAL, "NULL argument");
  *out = h->buf;
  return S2D_OK;
}
S2D_API int s2d_match_buffer_offsets(S2DMatchHandle h, int64_t* offsets, int n_offsets) {
  if (!h || !offsets) return mfail(S2D_EINVAL, "NULL argument");
  const S2DMatchBuffers& b = h->buf;
  const void* ptrs[] = {b.x, b.y, b.vx, b.vy, b.body, b.stamina, b.effort, b.recovery, b.stamina_capacity, b.tackle_cycles,
                        b.catch_ban, b.cycle, b.mode, b.mode_side, b.score_left, b.score_right, b.last_touch_side, b.setplay_timer,
                        b.offside_mask, b.ball_holder, b.goalie_moves, b.set_play_taker, b.last_kicker, b.stopped_cycle, b.tick, b.card,
                        b.reward_left, b.done, b.nearest_left, b.nearest_right, b.stats};
  const int count = 1 + (int)(sizeof ptrs / sizeof ptrs[0]);
  if (n_offsets < count) return mfail(S2D_EINVAL, "offsets array too small (need 32)");
  offsets[0] = (int64_t)h->arena_bytes;
  for (int k = 1; k < count; ++k) offsets[k] = (int64_t)(static_cast<const char*>(ptrs[k - 1]) - h->arena);
  return S2D_OK;
}

static int m_launch(S2DMatchHandle h, int n_steps, const float* actions, const S2DMatchRollout* out, void* stream) {
  MRoll ro{nullptr, nullptr, nullptr, nullptr};
  if (out) ro = MRoll{out->obs, out->reward, out->mode, out->done};
  MDeviceGuard guard(h->device);
  if (h->stock_sched)
    hipLaunchKernelGGL((s2d_match_rollout_kernel<true, true, true>), dim3(m_grid(h->n)), dim3(kMBlock), 0, static_cast<hipStream_t>(stream), h->mp,
                       h->ptrs, h->n, n_steps, actions, ro);
  else if (h->stock_types)
    hipLaunchKernelGGL((s2d_match_rollout_kernel<true, true>), dim3(m_grid(h->n)), dim3(kMBlock), 0, static_cast<hipStream_t>(stream), h->mp,
                       h->ptrs, h->n, n_steps, actions, ro);
  else if (h->stock)
    hipLaunchKernelGGL((s2d_match_rollout_kernel<true, false>), dim3(m_grid(h->n)), dim3(kMBlock), 0, static_cast<hipStream_t>(stream), h->mp,
                       h->ptrs, h->n, n_steps, actions, ro);
  else if (h->mp.illegal_defense_number > 0)
    hipLaunchKernelGGL((s2d_match_rollout_kernel<false, false, false, true>), dim3(m_grid(h->n)), dim3(kMBlock), 0, static_cast<hipStream_t>(stream),
                       h->mp, h->ptrs, h->n, n_steps, actions, ro);
  else
    hipLaunchKernelGGL((s2d_match_rollout_kernel<false, false>), dim3(m_grid(h->n)), dim3(kMBlock), 0, static_cast<hipStream_t>(stream), h->mp,
                       h->ptrs, h->n, n_steps, actions, ro);
  MHIP_TRY(hipGetLastError());
  return S2D_OK;
}
S2D_API const char* s2d_match_kernel_name(S2DMatchHandle h) {
  if (!h) return "";
  if (h->stock_sched) return "s2d_match_rollout_kernel<stock rules, own schedule>";
  return h->stock_types ? "s2d_match_rollout_kernel<stock, stock types>" : h->stock ? "s2d_match_rollout_kernel<stock>" :
         h->mp.illegal_defense_number > 0 ? "s2d_match_rollout_kernel<general, illegal defense>" : "s2d_match_rollout_kernel<general>";
}
S2D_API int s2d_match_relative(S2DMatchHandle h, float* dist_dev, float* angle_dev, void* stream) {
  if (!h || !dist_dev || !angle_dev) return mfail(S2D_EINVAL, "NULL argument");
  MDeviceGuard guard(h->device);
  hipLaunchKernelGGL(s2d_match_relative_kernel, dim3(m_grid(h->n)), dim3(kMBlock), 0, static_cast<hipStream_t>(stream), h->ptrs,
                     h->n, dist_dev, angle_dev);
  MHIP_TRY(hipGetLastError());
  return S2D_OK;
}
S2D_API int s2d_match_step(S2DMatchHandle h, const float* actions_dev, void* stream) {
  if (!h) return mfail(S2D_EINVAL, "NULL handle");
  return m_launch(h, 1, actions_dev, nullptr, stream);
}
S2D_API int s2d_match_rollout(S2DMatchHandle h, int n_steps, const float* actions_dev, const S2DMatchRollout* out, void* stream) {
  if (!h) return mfail(S2D_EINVAL, "NULL handle");
  if (n_steps < 0) return mfail(S2D_EINVAL, "n_steps must be >= 0");
  if (out && out->obs && (reinterpret_cast<uintptr_t>(out->obs) & 15u)) return mfail(S2D_EINVAL, "rollout obs buffer must be 16-byte aligned");
  if (n_steps == 0) return S2D_OK;
  return m_launch(h, n_steps, actions_dev, out, stream);
}
