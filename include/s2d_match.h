/*
 * s2d_match.h -- C ABI of the 11v11 full-match engine (BASELINE.json
 * configs[3]: 22 players, kick / tackle / catch / offside / stamina, heterogeneous player
 * types, thousands of lockstep matches).
 * Same library (libs2d_hip.so), same conventions as s2d.h: plain C, device pointers,
 * hipStream_t as void*, stream-ordered asynchronous launches, 0 / negative error codes.
 *
 * What it replaces in the reference: the rcssserver match itself behind Soccer2DEnv
 * (soccer_2d_env.py:356-383 starts it; every State read at server.py:49-103 comes from it).
 * The reference holds NO Python for an 11v11 task; the boundary mirrored here is therefore the
 * protobuf schema: commands = PlayerAction {Dash, Turn, Kick, Tackle} (idl/service.proto:380-402),
 * observations = WorldModel {ball, teammates[11], opponents[11], game_mode_type, scores, cycle}
 * (idl/service.proto:144-175, 306-349), play modes = GameModeType (267-301).
 * All match rules are rcssserver's (EXT): parity-unpinned.
 */
#ifndef S2D_MATCH_H_
#define S2D_MATCH_H_

#include "s2d.h"

#ifdef __cplusplus
extern "C" {
#endif

#define S2D_MATCH_PLAYERS 22      /* 0..10 left team (attacks +x), 11..21 right team */
#define S2D_MATCH_SLOTS 24        /* per-env object slots in memory: 22 players, slot 22 = ball, 23 = pad */
#define S2D_MATCH_BALL 22
#define S2D_MATCH_OBJ_WORDS 5     /* observation row of one object: x, y, vx, vy, body */

#define S2D_MATCH_PLAYER_TYPES 18 /* PlayerParam.player_types, idl/service.proto:1666 */
#define S2D_MATCH_GOALIE_LEFT 0   /* the goalie is the first player of each team */
#define S2D_MATCH_GOALIE_RIGHT 11

/* body commands = the PlayerAction oneof members 1..6 (idl/service.proto:380-411, 1291-1298).  TurnNeck and
 * ChangeView only steer the vision model, which a full-state engine does not have: send S2D_MCMD_NONE. */
enum { S2D_MCMD_NONE = 0, S2D_MCMD_DASH = 1, S2D_MCMD_TURN = 2, S2D_MCMD_KICK = 3, S2D_MCMD_TACKLE = 4,
       S2D_MCMD_CATCH = 5, S2D_MCMD_MOVE = 6 };
/* GameModeType values used (idl/service.proto:267-301).  mode_side: for the restarts (KickOff_, KickIn_, FreeKick_, CornerKick_,
 * GoalKick_, IndFreeKick_, GoalieCatch_, PenaltyKick_ -- a FoulCharge_ called inside the offender's own penalty area, restarted from
 * that half's penalty spot) the side that takes it; for the ANNOUNCEMENTS (AfterGoal_, OffSide_, BackPass_,
 * FreeKickFault_, CatchFault_, FoulCharge_: rcssserver's goal_l, offside_l, back_pass_l, ...) the side the call is named after --
 * the scorer, the offender.  An announcement is a dead ball with the clock stopped; after announce_wait (after_goal_wait) cycles
 * the referee turns it into the restart for the other side.  FirstHalfOver / ExtendHalf: one stopped cycle at the end of a half /
 * of a drawn normal time (side = who kicks off next).  The shoot-out's modes: side = the team of the current taker
 * (PenaltyOnfield_: the half the kicks are taken in). */
enum {
  S2D_GM_BEFORE_KICK_OFF = 0, S2D_GM_TIME_OVER = 1, S2D_GM_PLAY_ON = 2, S2D_GM_KICK_OFF = 3, S2D_GM_KICK_IN = 4,
  S2D_GM_FREE_KICK = 5, S2D_GM_CORNER_KICK = 6, S2D_GM_GOAL_KICK = 7, S2D_GM_AFTER_GOAL = 8, S2D_GM_OFF_SIDE = 9,
  S2D_GM_PENALTY_KICK = 10, S2D_GM_FIRST_HALF_OVER = 11, S2D_GM_FOUL_CHARGE = 14, S2D_GM_BACK_PASS = 18, S2D_GM_FREE_KICK_FAULT = 19,
  S2D_GM_CATCH_FAULT = 20, S2D_GM_IND_FREE_KICK = 21, S2D_GM_GOALIE_CATCH = 30, S2D_GM_EXTEND_HALF = 31,
  /* the penalty shoot-out after a drawn extra time (idl/service.proto:290-297).  PenaltyFoul_: the kicker played the ball a second
   * time although pen_allow_mult_kicks is off -- the kick counts as missed (the defending goalie cannot infringe: he stands still
   * until the kick) */
  S2D_GM_PENALTY_SETUP = 22, S2D_GM_PENALTY_READY = 23, S2D_GM_PENALTY_TAKEN = 24, S2D_GM_PENALTY_MISS = 25,
  S2D_GM_PENALTY_SCORE = 26, S2D_GM_PENALTY_ONFIELD = 28, S2D_GM_PENALTY_FOUL = 29,
  S2D_GM_FOUL_PUSH = 15, S2D_GM_FOUL_MULTIPLE_ATTACKER = 16, S2D_GM_FOUL_BALL_OUT = 17,   /* announcements the engine's own referee never
                                             makes (rcssserver defines them and calls only FoulCharge_): written into the mode plane with the
                                             offending side they are played like one -- dead ball, then a FreeKick_ for the other side */
  S2D_GM_PAUSE = 12, S2D_GM_HUMAN = 13,   /* never entered by the engine: an operator writes them into the mode plane to HOLD a match
                                             (nobody acts, nothing is decided, the clock stands) and another mode to let it go on */
  S2D_GM_ILLEGAL_DEFENSE = 27   /* an announcement like OffSide_: named after the offending side (see illegal_defense_number) */
};
/* During the shoot-out the set-play word (S2DMatchBuffers.set_play_taker) carries its state -- PenaltyKickState of the proto
 * (idl/service.proto:130-138): bits 0-7 = 1 + index of the current taker, 12-15 / 16-19 = kicks taken by the left / right team,
 * 20-23 / 24-27 = their shoot-out goals; the mode side is the current taker's team. */
#define S2D_PEN_TAKER(w) (((w) & 0xff) - 1)
#define S2D_PEN_KICKS_LEFT(w) (((w) >> 12) & 15)
#define S2D_PEN_KICKS_RIGHT(w) (((w) >> 16) & 15)
#define S2D_PEN_SCORE_LEFT(w) (((w) >> 20) & 15)
#define S2D_PEN_SCORE_RIGHT(w) (((w) >> 24) & 15)
/* cards (rcssserver's yellow_card / red_card referee messages; no field of the proto's Player carries them) */
enum { S2D_CARD_NONE = 0, S2D_CARD_YELLOW = 1, S2D_CARD_RED = 2 /* sent off: parked beside the pitch, commands ignored */ };

/* ServerParam fields the match needs beyond S2DServerParams (same names as idl/service.proto:
 * 1435-1662); defaults = rcssserver stock values. */
typedef struct S2DMatchParams {
  double kick_power_rate, kickable_margin, kick_rand, max_power, min_power;        /* .027 .7 .1 100 -100 */
  double tackle_dist, tackle_back_dist, tackle_width, tackle_power_rate;           /* 2 0 1.25 .027 */
  double max_tackle_power, max_back_tackle_power;                                  /* 100 0 */
  double goal_width, offside_active_area_size, free_kick_distance;                 /* 14.02 2.5 9.15 */
  int32_t tackle_cycles, half_time_cycles, nr_normal_halfs, drop_ball_time;        /* 10 3000 2 100 */
  int32_t use_offside, catch_ban_cycle;                                            /* 1 5 */
  /* goalie catch (idl/service.proto:1488-1490, 1643-1644); the catch rectangle is catchable_area_l *
   * PlayerType.catchable_area_l_stretch long, catch_area_w wide, rooted at the goalie */
  double catchable_area_l, catch_area_w, catch_probability, max_catch_angle, min_catch_angle;  /* 1.2 1 1 90 -90 */
  double penalty_area_length, penalty_area_half_width;                             /* 16.5 20.16 */
  int32_t goalie_max_moves;               /* 2: Move commands a goalie may issue while he holds a caught ball */
  int32_t after_goal_wait;                /* 50: cycles of AfterGoal_ (mode side = the scorer) between a goal and the
                                             kick-off formation; 0 = kick-off at once */
  int32_t kick_off_wait;                  /* 0: cycles of BeforeKickOff (idl/service.proto:268) before the kick-off of each
                                             half -- nobody may play the ball, players may Move inside their own half
                                             (rcssserver's auto_mode waits kick_off_wait = 100); 0 = KickOff_ at once */
  int32_t back_passes;                    /* 1: a goalie catching a ball a team-mate kicked concedes an indirect free kick
                                             (BackPass_, idl/service.proto:286; ServerParam.back_passes :1579) */
  int32_t free_kick_faults;               /* 1: the taker of a set play may not play the ball twice in a row
                                             (FreeKickFault_, :287; ServerParam.free_kick_faults :1578) */
  int32_t stopped_clock;                  /* 1: the clock (WorldModel.cycle) stands still in BeforeKickOff, AfterGoal_, the
                                             announcements, FirstHalfOver and TimeOver; WorldModel.stoped_cycle (:333) counts those
                                             cycles.  0 = the round-1/2 behaviour: time runs in every mode */
  int32_t announce_wait;                  /* 30: cycles an announcement (OffSide_, BackPass_, FreeKickFault_, CatchFault_,
                                             FoulCharge_) lasts before the restart it awards (rcssserver's AFTER_*_WAIT) */
  int32_t foul_cycles;                    /* 5: cycles a fouled player stays down (ServerParam.foul_cycles, :1633) */
  int32_t nr_extra_halfs;                 /* 2: extra halves played when the normal time ends in a draw (ServerParam.nr_extra_halfs,
                                             idl/service.proto:1601): one stopped cycle of ExtendHalf (:299, rcssserver's
                                             "time_extended"), then a kick-off; 0 = the match ends with the normal time */
  double foul_detect_probability;         /* .5: chance that the referee sees an intentional foul (:1632) */
  int32_t extra_half_cycles;              /* 1000: length of an extra half (ServerParam.extra_half_time :1622, 100 s of 10 cycles);
                                             FirstHalfOver between extra halves; after the last one a draw goes to the shoot-out below */
  int32_t golden_goal;                    /* 0: a goal in extra time ends the match at once (ServerParam.golden_goal :1635) */
  /* The penalty shoot-out (ServerParam.penalty_shoot_outs, pen_*: idl/service.proto:1602-1613) after a draw that the last period
   * leaves: PenaltyOnfield_ (side = the half the kicks are taken in: the right one) for pen_before_setup_wait cycles; then per
   * kick PenaltySetup_ (one cycle: the ball on the spot pen_dist_x from the right goal line, the taker behind it, the other team's
   * goalie on the line, everybody else inside the centre circle, all placed by the referee = pen_coach_moves_players), PenaltyReady_
   * (the taker has pen_ready_wait cycles to play the ball), PenaltyTaken_ (taker against goalie, at most pen_taken_wait cycles:
   * ball in the goal = PenaltyScore_; out, caught or out of time = PenaltyMiss_; a second touch with pen_allow_mult_kicks off =
   * PenaltyFoul_), the verdict for pen_before_setup_wait cycles.
   * The left team kicks first, takers from index 10 downwards; pen_nr_kicks each (decided early when one side cannot catch up),
   * then pairs of kicks until one pair decides or pen_max_extra_kicks more are used up (then the draw stands, unless pen_random_winner
   * below tosses a coin).  The clock stands throughout. */
  int32_t penalty_shoot_outs;             /* 1 */
  int32_t pen_before_setup_wait, pen_ready_wait, pen_taken_wait;   /* 10 10 150 */
  int32_t pen_nr_kicks, pen_max_extra_kicks;                       /* 5 5 (their sum <= 15) */
  double pen_dist_x;                      /* 42.5 */
  /* IllegalDefense_ (idl/service.proto:295; ServerParam.illegal_defense_number / _duration / _dist_x / _width :1637-1640; OFF in the
   * stock server: number = 0).  While the ball is in play and the OTHER team was the last to play it, a team that keeps at least
   * `number` players inside the strip of dist_x in front of its own goal line, width wide, for `duration` cycles on end is called:
   * IllegalDefense_ named after it, the ball on that half's penalty spot, after announce_wait a FreeKick_ for the other team.
   * (In PlayOn the two counters live in setplay_timer, bits 0-7 left / 8-15 right: the word is otherwise unused there.) */
  int32_t illegal_defense_number, illegal_defense_duration;   /* 0 20 (duration <= 255) */
  double illegal_defense_dist_x, illegal_defense_width;       /* 16.5 40.32 */
  int32_t pen_allow_mult_kicks;           /* 1 (ServerParam.pen_allow_mult_kicks, idl/service.proto:1611): the taker may play the ball
                                             again during PenaltyTaken_; 0 = a second touch of his is PenaltyFoul_, the kick is missed */
  int32_t pen_random_winner;              /* 0 (ServerParam.pen_random_winner, idl/service.proto:1610): a shoot-out that ends level is decided
                                             by a coin (one Philox draw in the cycle the match ends): bits 28-29 of the set-play word =
                                             1 left / 2 right won the toss; the score is not touched.  (The field took the place of
                                             reserved_mp2: the struct's size and every other offset are unchanged.) */
} S2DMatchParams;

/* PlayerType (idl/service.proto:1697-1732): the members that enter the dynamics.  Type 0 is the
 * default type (= the ServerParam / S2DMatchParams values). */
typedef struct S2DPlayerType {
  double player_speed_max, stamina_inc_max, player_decay, inertia_moment, dash_power_rate, player_size;
  double kickable_margin, kick_rand, extra_stamina, effort_max, effort_min, kick_power_rate;
  double catchable_area_l_stretch;
} S2DPlayerType;

/* PlayerParam (idl/service.proto:1664-1695): the ranges rcssserver draws its heterogeneous types
 * from (stock values in s2d_match_default_player_params). */
typedef struct S2DPlayerParams {
  double player_speed_max_delta_min, player_speed_max_delta_max, stamina_inc_max_delta_factor;
  double player_decay_delta_min, player_decay_delta_max, inertia_moment_delta_factor;
  double dash_power_rate_delta_min, dash_power_rate_delta_max, player_size_delta_factor;
  double kickable_margin_delta_min, kickable_margin_delta_max, kick_rand_delta_factor;
  double extra_stamina_delta_min, extra_stamina_delta_max, effort_max_delta_factor, effort_min_delta_factor;
  double new_dash_power_rate_delta_min, new_dash_power_rate_delta_max, new_stamina_inc_max_delta_factor;
  double kick_power_rate_delta_min, kick_power_rate_delta_max;
  double catchable_area_l_stretch_min, catchable_area_l_stretch_max;
} S2DPlayerParams;

typedef struct S2DMatchConfig {
  uint32_t abi_version;   /* S2D_ABI_VERSION */
  uint32_t struct_bytes;  /* sizeof(S2DMatchConfig) */
  S2DServerParams sp;
  S2DMatchParams mp;
  uint64_t seed;
  int64_t env_id_offset;
  int32_t auto_reset;     /* 1: a finished match (TimeOver) restarts inside the same step */
  int32_t noise;          /* 0: player_rand/ball_rand/kick_rand off; tackle success is always drawn */
  int32_t reserved[4];
  /* heterogeneous players: the type table and the type of every player slot (DoChangePlayerType,
   * idl/service.proto:1393-1433).  s2d_match_default_config: 18 copies of the default type, all ids 0. */
  S2DPlayerType player_types[S2D_MATCH_PLAYER_TYPES];
  int32_t player_type_id[S2D_MATCH_SLOTS];   /* [0..21]; the goalies (0, 11) keep type 0 in rcssserver */
} S2DMatchConfig;

/* Device buffers.  Per-object planes are [N][24] (slot = lane of the env's half-wave);
 * per-env arrays are [N]. */
typedef struct S2DMatchBuffers {
  int64_t n_envs;
  float *x, *y, *vx, *vy, *body;                      /* players + ball (ball: body unused) */
  float *stamina, *effort, *recovery, *stamina_capacity;
  int32_t *tackle_cycles;                             /* >0: player frozen after a tackle */
  int32_t *catch_ban;                                 /* >0: goalie may not catch (catch_ban_cycle after every attempt) */
  int32_t *cycle, *mode, *mode_side, *score_left, *score_right;
  int32_t *last_touch_side, *setplay_timer, *offside_mask;     /* bit i = player i flagged */
  int32_t *ball_holder, *goalie_moves;  /* 1 + index of the goalie holding a caught ball (0 = nobody), his remaining moves */
  int32_t *set_play_taker, *last_kicker;   /* 1 + index (0 = nobody): who put the ball into play from the last set play and has
                                            not been followed by another touch; who last moved it with a Kick command */
  int32_t *stopped_cycle;  /* [N] WorldModel.stoped_cycle (idl/service.proto:333): cycles the clock has been standing still */
  int32_t *tick;           /* [N] simulator cycles since s2d_match_reset, stopped ones included: the Philox counter of every draw */
  int32_t *card;           /* [N][24] S2D_CARD_* per player */
  float *reward_left;      /* [N] +1 left goal, -1 right goal this cycle */
  uint8_t *done;           /* [N] 1 when the match reached TimeOver this cycle */
  int32_t *nearest_left, *nearest_right;              /* [N] index of the player closest to the ball, per team */
  unsigned long long *stats;  /* [S2D_STATS_STRIPES][8], sum over stripes: [0] env-steps [1] goals left
                                 [2] goals right [3] matches finished [4] kicks + catches [5] tackles
                                 [6] offsides [7] ball-outs */
} S2DMatchBuffers;

typedef struct S2DMatchRollout {
  float *obs;        /* [T][N][24][5] x,y,vx,vy,body after each cycle (slot 22 = ball) or NULL; 16-byte aligned */
  float *reward;     /* [T][N] or NULL */
  int32_t *mode;     /* [T][N] or NULL */
  uint8_t *done;     /* [T][N] or NULL */
} S2DMatchRollout;

typedef struct S2DMatchEngine *S2DMatchHandle;

void s2d_match_default_config(S2DMatchConfig *cfg);
void s2d_match_default_player_params(S2DPlayerParams *pp);
/* Fill cfg->player_types[1..17] the way rcssserver's HeteroPlayer does (trade-off pairs drawn from
 * the PlayerParam ranges; EXT), deterministically from `seed` (Philox).
 * pp == NULL: stock ranges.  Type 0 stays the default type; player_type_id is not touched. */
int s2d_match_generate_player_types(S2DMatchConfig *cfg, const S2DPlayerParams *pp, uint64_t seed);
int s2d_match_validate_config(const S2DMatchConfig *cfg);
size_t s2d_match_arena_bytes(const S2DMatchConfig *cfg, int64_t n_envs);
int s2d_match_create(const S2DMatchConfig *cfg, int64_t n_envs, int device, void *arena_dev, size_t arena_bytes,
                     void *stream, S2DMatchHandle *out);
void s2d_match_destroy(S2DMatchHandle h);
int s2d_match_buffers(S2DMatchHandle h, S2DMatchBuffers *out);
/* byte offsets of every S2DMatchBuffers pointer from the arena base (slot 0 = arena size) */
int s2d_match_buffer_offsets(S2DMatchHandle h, int64_t *offsets, int n_offsets);
/* kick-off formation, full stamina, score 0-0, cycle 0, KickOff for the left side */
int s2d_match_reset(S2DMatchHandle h, const uint8_t *mask_dev, void *stream);
/* actions_dev: float[N][22][3] = {command, a, b}: Dash(power=a, dir=b) Turn(moment=a)
 * Kick(power=a, dir=b) Tackle(power_or_dir=a, foul = b != 0; idl/service.proto:399-402) Catch(dir=a, goalies only) Move(x=a, y=b in the team's own frame: the
 * right team's coordinates are mirrored; legal before a kick-off into the own half, and for a goalie
 * holding a caught ball inside his penalty area); NULL = uniform random policy drawn in-kernel */
int s2d_match_step(S2DMatchHandle h, const float *actions_dev, void *stream);
int s2d_match_rollout(S2DMatchHandle h, int n_steps, const float *actions_dev /* [T][N][22][3] or NULL */,
                      const S2DMatchRollout *out, void *stream);

/* Per-agent relative tables of the WorldModel every player receives: for agent p (0..21) and object
 * j (0..21 players, 22 = ball) dist[N][22][23] = Player.dist_from_self / Ball.dist_from_self and
 * angle[N][22][23] = Player.angle_from_self / Ball.angle_from_self (absolute direction of j seen from
 * p, degrees; idl/service.proto:84-85, 155-156).  The diagonal (j = p) is 0. */
int s2d_match_relative(S2DMatchHandle h, float *dist_dev, float *angle_dev, void *stream);

/* Which instantiation of the cycle kernel this engine launches: "...<stock>" when its configuration equals
 * s2d_match_default_config() in every rule / physics word (those are compile-time constants there), "...<general>" otherwise
 * (same arithmetic, parameters read at run time; S2D_MATCH_GENERAL_KERNEL=1 in the environment selects it regardless).  Seed,
 * env_id_offset, auto_reset, noise and the PlayerTypes do not matter for the choice. */
const char *s2d_match_kernel_name(S2DMatchHandle h);

#ifdef __cplusplus
}
#endif
#endif /* S2D_MATCH_H_ */
